// The FAST pass of a 129 .. 512-frame kappa-sigma stack on gfx950: two lanes per pixel, the default engine's running sums.
//
// sigma_clip_combine (core/stacking/combine.rs:14-92) for the frame counts of deep light stacks and master calibration stacks.
// Round 5 kept 129 .. 256 samples in ONE lane's registers (VGPRs + AGPRs: one wave per SIMD, 8 .. 9 cycles per instruction because a
// lone wave cannot issue back to back, and nothing to hide the 256 loads behind: 10.6 ms for 256 x 4096^2) and ran 257 .. 512 through
// stack_pair.hip's oracle arithmetic (two f64 chains of 512 additions per iteration: 58 ms).  Here, as in stack_pair.hip, lanes 2k
// and 2k + 1 share pixel k of the wave's 32 and hold H = 128 (129 .. 256 frames: 256 registers, TWO waves per SIMD) or H = 256
// (257 .. 512 frames) samples each:
//   * gather: the even lane takes frames 0 .. R - 1, the odd lane frames R .. 2R - 1 (R = the frame-count class, a multiple of 16 or
//     32: wires R .. H - 1 are +inf pads known at compile time -- no loads, and SortNet<H>::sort_fused_n<R> drops their operations;
//     frames n .. 2R - 1 read a plane of +inf); each lane sorts its samples, one cross step + an in-lane bitonic merge leave sorted
//     ranks 0 .. H - 1 in the even lane and H .. 2H - 1 in the odd lane (stack_pair.hpp);
//   * median and MAD at a compile-time position: every pixel of this kernel holds all n samples finite (the others are handed over,
//     see below), so M = n / 2 is one number per launch and a binary dispatch reaches the instance written for it -- the window
//     formula MAD = min_p max(med - V[p], V[p + M] - med) with constant register indices, the partner's rank through DPP;
//   * clipping: the default engine of stack_sigma_clip.hip (clip_fast) in pair form.  Survivors are a rank interval; the low end
//     lives in the even lane's first registers, the high end in the odd lane's last real ones (a launch-uniform position), each
//     walk looks at eight samples; iteration >= 1 gets mean and sigma from running moments about the median, E = sum (x - med),
//     Q = sum (x - med)^2 in f64 -- one pass over the lane's registers, the pair's halves added through DPP -- minus what the walks
//     shaved.  Same contract as the <= 64-frame fast engine: 1e-5 relative, at most 1e-4 of the pixels may differ from the oracle
//     at all (measured: none).
//   * whatever does not fit -- a pixel with a non-finite sample, more than eight rejected samples at one end in total, nothing
//     surviving -- is appended to one of 2048 lists and redone by stack_pair.hip's kernel (the oracle's arithmetic, bit for bit)
//     in LIST mode: AB_FB_STACK_GENERAL_PIXELS counts them under AB_TRACE.
// ~5 700 instructions per wave of 32 pixels at H = 128.
#include "stack_pair.hpp"

#include <algorithm>
#include <cmath>

using namespace abpair;

namespace {

template <int OP>  // 1 min, 2 max (signed)
__device__ __forceinline__ int wave_reduce_i32(int x) {
    constexpr int id = OP == 1 ? 0x7fffffff : (int)0x80000000;
    auto op = [](int a, int b) { return OP == 1 ? min(a, b) : max(a, b); };
    x = op(x, __builtin_amdgcn_update_dpp(id, x, 0x111, 0xf, 0xf, false));
    x = op(x, __builtin_amdgcn_update_dpp(id, x, 0x112, 0xf, 0xf, false));
    x = op(x, __builtin_amdgcn_update_dpp(id, x, 0x114, 0xf, 0xf, false));
    x = op(x, __builtin_amdgcn_update_dpp(id, x, 0x118, 0xf, 0xf, false));
    x = op(x, __builtin_amdgcn_update_dpp(id, x, 0x142, 0xa, 0xf, false));
    x = op(x, __builtin_amdgcn_update_dpp(id, x, 0x143, 0xc, 0xf, false));
    return __builtin_amdgcn_readlane(x, 63);
}

// sqrt(v) for the iteration's sigma, of which only the f32 rounding is used (stack_sigma_clip.hip: sqrt_for_sigma)
__device__ __forceinline__ double sqrt_for_sigma(double v) {
    const double y = __builtin_amdgcn_rsq(v);
    const double g = v * y;
    const double e = __builtin_fma(-g, g, v);
    const double r = __builtin_fma(e, 0.5 * y, g);
    return v > 0.0 ? r : 0.0;
}

__device__ __forceinline__ void nop_fence(float &x) { asm volatile("s_nop 1" : "+v"(x)); }

// median (combine.rs:38-40) and MAD (combine.rs:42-46) of a pair that holds n finite samples with n / 2 = M: ranks 0 .. H - 1 in the
// even lane, H .. n - 1 in the odd lane.  MAD = the (M + 1)-th smallest deviation = min over the windows [p, p + M] of sorted samples
// that contain the median of the larger end deviation: p = 0 .. n - 1 - M (stack_sigma_clip.hip: med_mad_at).
template <int H, int M>
__device__ __forceinline__ void med_mad_at(const float (&v)[H], bool odd, bool n_is_odd, float &med_out, float &mad_out) {
    static_assert(M >= H / 2 && M <= H, "a pair holds more than H samples");
    constexpr int S = H - M;  // rank p + M is the odd lane's register p - S
    float med;
    if constexpr (M < H) {
        const float o = swapf(v[M]);
        med = odd ? o : v[M];
    } else {
        const float o = swapf(v[0]);
        med = odd ? v[0] : o;
    }
    const float inf = __builtin_inff();
    float best_even = inf, best_pair = inf;
    // windows that end inside the even lane (its result counts; the odd lane computes along on its own registers and is ignored)
#pragma unroll
    for (int p = 0; p < (S < M ? S : M); ++p) best_even = ab_v_min(best_even, ab_v_max(med - v[p], v[p + M] - med));
    // windows that end in the odd lane: the even lane brings med - V[p], the odd lane V[p + M] - med
#pragma unroll
    for (int p = S; p < M; ++p) {
        const float d = odd ? v[p - S] - med : med - v[p];
        best_pair = ab_v_min(best_pair, ab_v_max(d, swapf(d)));
    }
    if (n_is_odd) {  // (launch-uniform) n = 2M + 1: the window [M, 2M], whose larger deviation is V[2M] - med
        constexpr int T = 2 * M - H;  // V[2M] in the odd lane
        if constexpr (T >= 0 && T < H) {
            const float e = v[T] - med;
            const float o = swapf(e);
            best_pair = ab_v_min(best_pair, odd ? e : o);
        }
    }
    nop_fence(best_even);  // (written inside an asm statement: see dpp_fence)
    const float be = swapf(best_even);
    med_out = med;
    mad_out = ab_v_min(best_pair, odd ? be : best_even);
}
template <int H, int LO, int HI>
__device__ __forceinline__ void med_mad_dispatch(const float (&v)[H], bool odd, bool n_is_odd, int M /* launch-uniform */, float &med, float &mad) {
    if constexpr (LO == HI) {
        med_mad_at<H, LO>(v, odd, n_is_odd, med, mad);
    } else {
        constexpr int MID = (LO + HI) / 2;
        if (M <= MID)
            med_mad_dispatch<H, LO, MID>(v, odd, n_is_odd, M, med, mad);
        else
            med_mad_dispatch<H, MID + 1, HI>(v, odd, n_is_odd, M, med, mad);
    }
}

// the median alone (median_combine_row_major, calibration.rs:106-124: sorted[len / 2]): rank M
template <int H, int LO, int HI>
__device__ __forceinline__ float median_dispatch(const float (&v)[H], bool odd, int M /* launch-uniform */) {
    if constexpr (LO == HI) {
        if constexpr (LO < H) {
            const float o = swapf(v[LO]);
            return odd ? o : v[LO];
        } else {
            const float o = swapf(v[0]);
            return odd ? v[0] : o;
        }
    } else {
        constexpr int MID = (LO + HI) / 2;
        return M <= MID ? median_dispatch<H, LO, MID>(v, odd, M) : median_dispatch<H, MID + 1, HI>(v, odd, M);
    }
}

// hand the pixel to the list pass: one atomic per wave, kListSlots counters (stack_sigma_clip.hip)
__device__ __forceinline__ void hand_over(const PairArgs &a, bool d, int lane, int64_t g) {
    const unsigned long long m = __ballot(d);
    if (m) {
        const int leader = (int)__builtin_ctzll(m);
        const unsigned int w = blockIdx.x;
        const unsigned int slot = (w + (w / kListSlots) * 977u) & (kListSlots - 1);
        unsigned int base = 0;
        if (lane == leader) base = atomicAdd(&a.list_count[slot], (unsigned int)__builtin_popcountll(m));
        base = __shfl(base, leader, 64);
        if (d) a.list[(size_t)slot * a.list_cap + base + (unsigned int)__builtin_popcountll(m & ((1ull << lane) - 1ull))] = (int)g;
    }
}

// One clipping pass over the two ends of the pair's rank interval (combine.rs:65-82).  [la, lb]: this lane's part of it in its own
// register indices.  The low end is the even lane's registers 0 .. 7; the high end the odd lane's last real registers (chunks ct and
// ct - 1, ct launch-uniform) and, when the odd lane holds at most eight samples, the even lane's top registers as well.  A sorted end
// fails the test as a prefix, so the counts of the two lanes simply add.  `decided`: both ends saw a surviving sample within their
// eight -- otherwise the pixel is not this kernel's.  UPDATE: the shaved samples are folded into e_rem / q_rem (moments about c0).
template <int H, bool UPDATE>
__device__ __forceinline__ void clip_walk(const float (&v)[H], bool odd, bool go, int la, int lb, int ct, float center, float lo, float hi, float c0,
                                          double c0d, int &cl_own, int &ch_own, bool &decided, double &e_rem, double &q_rem) {
    constexpr int NC = H / 4;
    int cl = 0, ch = 0;
    bool found_lo = false, found_hi = false;
    auto fold = [&](const bool (&r)[4], int base) {
        if constexpr (UPDATE) {
            if (__any(r[0] || r[1] || r[2] || r[3])) {
                asm volatile("" ::: "memory");  // keeps this a branch (stack_sigma_clip.hip: clip_ends)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float xm = r[j] ? v[base + j] : c0;
                    const double e = (double)xm - c0d;  // 0 for lanes that keep the sample
                    e_rem += e;
                    q_rem = __builtin_fma(e, e, q_rem);
                }
            }
        }
    };
    const bool lo_lane = go && !odd, hi_lane = go && odd;
#pragma unroll
    for (int c = 0; c < 2; ++c) {
        bool r[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int i = 4 * c + j;
            const bool in = (i >= la) && (i <= lb);
            const bool ok = (v[i] - center) >= lo;
            r[j] = lo_lane && in && !ok;
            found_lo = found_lo || (in && ok);
            cl += r[j] ? 1 : 0;
        }
        fold(r, 4 * c);
        if (c == 1 || !__any(lo_lane && !found_lo)) break;
    }
    auto high_chunk = [&](int c, bool mine) {  // registers 4c + 3 down to 4c
        bool r[4];
#pragma unroll
        for (int j = 3; j >= 0; --j) {
            const int i = 4 * c + j;
            const bool in = (i >= la) && (i <= lb);
            const bool ok = (v[i] - center) <= hi;
            r[j] = mine && in && !ok;
            found_hi = found_hi || (mine && in && ok);
            ch += r[j] ? 1 : 0;
        }
        fold(r, 4 * c);
    };
#pragma unroll
    for (int c = NC - 1; c >= 0; --c) {
        if (c != ct && c != ct - 1) continue;                        // (uniform)
        if (c == ct - 1 && !__any(hi_lane && !found_hi)) continue;  // the first chunk settled it for every pixel of the wave
        high_chunk(c, hi_lane);
    }
    if (ct <= 1) {  // (uniform) the odd lane holds at most eight samples and the two chunks above were all of them
        if (__any(hi_lane && !found_hi)) {
            high_chunk(NC - 1, lo_lane);
            if (__any(lo_lane && !found_hi)) high_chunk(NC - 2, lo_lane);
        }
    }
    const int f_lo = (!odd && found_lo) ? 1 : 0, f_hi = found_hi ? 1 : 0;
    decided = ((f_lo | swapi(f_lo)) & (f_hi | swapi(f_hi))) != 0;
    cl_own = cl;
    ch_own = ch;
}

template <int H, int R, bool MEDIAN = false>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(H == 128 ? 2 : 1, H == 128 ? 2 : 1))) void stack_duo_fast_kernel(const PairArgs a) {
    static_assert(R <= H && 2 * R > H && R % 8 == 0, "frame-count class");
    const int lane = threadIdx.x;
    const bool odd = lane & 1;
    const int pix = lane >> 1;
    int64_t g = (int64_t)blockIdx.x * 32 + pix;
    const bool valid = g < a.total;
    if (!valid) g = a.total - 1;

    // ---- gather (combine.rs:170-175): BOTH lanes of a pair fetch their pixel from frame f and from frame R + f (wave-uniform planes:
    // scalar base + one shared 32-bit byte offset, one 128-byte line per instruction) and each keeps its half's sample ----
    float v[H];
    const int have = odd ? a.n - R : R;  // frames of this half
    const uint32_t boff = (uint32_t)g * 4u;
#pragma unroll
    for (int f = 0; f < H; ++f) {
        if (f < R) {
            const float xe = *(const float *)((const char *)a.p[f] + boff);
            const float xo = *(const float *)((const char *)a.p[f + R] + boff);  // (table entries past n point at a plane of +inf)
            v[f] = odd ? xo : xe;
        } else {
            v[f] = __builtin_inff();  // a pad of the class: never loaded, never moved by the network
        }
    }
    float nf = 0.0f;  // fma(x, 0, nf) stays 0 for finite x and turns NaN for inf / NaN
#pragma unroll
    for (int f = 0; f < R; ++f) nf = __builtin_fmaf(f < have ? v[f] : 0.0f, 0.0f, nf);
    int full = 1;
    if (__any(nf != nf)) {  // rare: some lane of this wave met a non-finite sample -- its pixel goes to the list (the network takes no NaN)
        int cnt = 0;
#pragma unroll
        for (int f = 0; f < R; ++f) {
            const bool fin = __builtin_isfinite(v[f]);
            v[f] = fin ? v[f] : __builtin_inff();
            cnt += fin ? 1 : 0;
        }
        full = cnt == have ? 1 : 0;
    }
    full &= swapi(full);
    bool defer = !full;

    // ---- sort: R per lane, cross step, in-lane bitonic merge ----
    if constexpr (R < H)
        SortNet<H>::template sort_fused_n<R>(v, [](auto) {});
    else
        SortNet<H>::sort_fused(v);
    dpp_fence<H>(v);
    cross_step<H>(v, odd);
    bitonic_merge<H>(v);
    dpp_fence<H>(v);

    // ---- median / MAD: n / 2 is one number per launch ----
    constexpr int kClassWidth = H / 8;  // frames per lane between two classes
    constexpr int kMLo = R - kClassWidth > H / 2 ? R - kClassWidth : H / 2;
    if constexpr (MEDIAN) {  // median_combine_row_major (calibration.rs:84-125): the pixels with a non-finite sample go to the list pass
        const float m = median_dispatch<H, kMLo, R>(v, odd, a.n >> 1);
        const bool writer = valid && !odd;
        if (writer && !defer) a.out[g] = m;
        hand_over(a, writer && defer, lane, g);
        return;
    }
    float med, mad;
    med_mad_dispatch<H, kMLo, R>(v, odd, (a.n & 1) != 0, a.n >> 1, med, mad);

    // ---- iteration 0: clip about the median with the MAD sigma (combine.rs:37-48,63-82) ----
    const int t_top = a.n - 1 - H;  // the odd lane's highest real register
    const int ct = t_top >> 2;
    int la = 0, lb = odd ? t_top : H - 1;  // this lane's part of the survivors
    int len = a.n;
    uint32_t rej = 0;
    float last_center = __builtin_nanf("");
    bool active = !defer;
    const float c0 = med;
    const double c0d = (double)med;
    double e_rem = 0.0, q_rem = 0.0;
    auto apply = [&](bool go, int cl_own, int ch_own, bool decided) {
        const int cl = cl_own + swapi(cl_own), ch = ch_own + swapi(ch_own);
        if (go && !decided) defer = true;
        const bool take = go && decided;
        const int removed = (cl + ch > len) ? len : (cl + ch);
        if (take) {
            rej += (uint32_t)removed;
            len -= removed;
            la += cl_own;
            lb -= ch_own;
        }
        active = take && (removed != 0);
    };
    if (a.max_iter >= 1) {
        const float sigma = (float)fmax((double)mad * kMadToSigma, 1e-10);
        const bool go = active;
        if (go) last_center = med;
        int cl_own, ch_own;
        bool decided;
        clip_walk<H, false>(v, odd, go, la, lb, ct, med, -a.sigma_low * sigma, a.sigma_high * sigma, c0, c0d, cl_own, ch_own, decided, e_rem, q_rem);
        apply(go, cl_own, ch_own, decided);
    }

    // ---- one pass over the survivors: E = sum e_i, Q = sum e_i^2 with e_i = x_i - med (f64), the pair's halves added ----
    double E, Q;
    {
        double E1 = 0.0, Q1 = 0.0;
        const int a_hi = wave_reduce_i32<2>(la), b_lo = wave_reduce_i32<1>(lb);
#pragma unroll
        for (int c = 0; c < H / 4; ++c) {
            if ((4 * c >= a_hi) && (4 * c + 3 <= b_lo)) {  // (uniform) every lane keeps all four
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const double e = (double)v[4 * c + j] - c0d;
                    E1 += e;
                    Q1 = __builtin_fma(e, e, Q1);
                }
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int i = 4 * c + j;
                    const bool in = (i >= la) && (i <= lb);
                    float xm = in ? v[i] : c0;
                    asm volatile("" : "+v"(xm));  // select the f32 sample (or the compiler selects the two halves of the f64)
                    const double e = (double)xm - c0d;
                    E1 += e;
                    Q1 = __builtin_fma(e, e, Q1);
                }
            }
        }
        E = E1 + swapd(E1);
        Q = Q1 + swapd(Q1);
    }

    // ---- iterations >= 1: mean / sigma from the running sums (combine.rs:50-82; stack_sigma_clip.hip: clip_fast_tail) ----
    for (uint32_t it = 1; it < a.max_iter; ++it) {
        if (!__any(active)) break;
        launder<H>(v);  // stop LICM from hoisting the f32->f64 conversions out of this loop
        const double er = e_rem + swapd(e_rem), qr = q_rem + swapd(q_rem);
        const double nn = (double)len;
        const double sum = __builtin_fma(nn, c0d, E - er);
        const double mean = sum / (double)(len > 0 ? len : 1);
        const double dlt = mean - c0d;
        double ss = (Q - qr) - nn * (dlt * dlt);
        ss = ss > 0.0 ? ss : 0.0;
        const double variance = ss / (double)(len > 1 ? len - 1 : 1);
        const float center = (float)mean;
        const float sigma = (float)fmax(sqrt_for_sigma(variance), 1e-10);
        const bool go = active && (len >= 2);
        if (go) last_center = center;
        int cl_own, ch_own;
        bool decided;
        clip_walk<H, true>(v, odd, go, la, lb, ct, center, -a.sigma_low * sigma, a.sigma_high * sigma, c0, c0d, cl_own, ch_own, decided, e_rem, q_rem);
        apply(go, cl_own, ch_own, decided);
    }

    // ---- result (combine.rs:85-91) ----
    const double er = e_rem + swapd(e_rem);
    const double S = __builtin_fma((double)len, c0d, E - er);
    const float mean_f = (float)(S / (double)(len > 0 ? len : 1));
    const float value = len > 0 ? mean_f : (__builtin_isfinite(last_center) ? last_center : 0.0f);
    const bool writer = valid && !odd;
    if (writer && !defer) a.out[g] = value;

    hand_over(a, writer && defer, lane, g);

    // rejection count: one atomic per wave, spread over kRejSlots counters (summed by the host)
    int r = (writer && !defer) ? (int)rej : 0;
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) r += __shfl_xor(r, off, 64);
    if (lane == 0 && r != 0) atomicAdd(&a.rejected[blockIdx.x & (kRejSlots - 1)], (unsigned long long)r);
}

}  // namespace

int ab_stack_duo_launch(ab_ctx *ctx, int H, int R, const PairArgs &args) {
    const dim3 grid((unsigned)((args.total + 31) / 32)), block(64);
#define AB_DUO_CASE(HV, RV)                                                                              \
    if (H == HV && R == RV) {                                                                            \
        if (args.median_only)                                                                            \
            hipLaunchKernelGGL((stack_duo_fast_kernel<HV, RV, true>), grid, block, 0, ctx->stream, args); \
        else                                                                                             \
            hipLaunchKernelGGL((stack_duo_fast_kernel<HV, RV>), grid, block, 0, ctx->stream, args);       \
        AB_HIP(ctx, hipGetLastError());                                                                  \
        return AB_OK;                                                                                    \
    }
#ifndef AB_DUO_ONE_CLASS  // (tests/test_abi_cpu.py walks the listing of ONE instance: the others are the same code at other constants)
    AB_DUO_CASE(128, 80)
    AB_DUO_CASE(128, 96)
    AB_DUO_CASE(128, 128)
#ifdef AB_DEV_ABLATION  // (257 .. 512 frames take four lanes per pixel, stack_quad.hip; AB_STACK_NO_QUAD=1 on a developer build: 36 ms against 15 for 512 frames)
    AB_DUO_CASE(256, 160)
    AB_DUO_CASE(256, 192)
    AB_DUO_CASE(256, 224)
    AB_DUO_CASE(256, 256)
#endif
#endif
    AB_DUO_CASE(128, 112)
#undef AB_DUO_CASE
    return ab_set_error(ctx, AB_ERR_INVALID, "internal: no two-lane kernel for %d samples per lane in class %d", H, R);
}
