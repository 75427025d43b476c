// The FAST pass of a 257 .. 512-frame kappa-sigma stack on gfx950: FOUR lanes per pixel, 128 samples each, two waves per SIMD.
//
// sigma_clip_combine (core/stacking/combine.rs:14-92).  stack_duo.hip's engine with the pair widened to a quad: with 256 samples per
// lane (two lanes per pixel) the samples fill the unified register file, ONE wave fits a SIMD, and a lone wave issues an instruction
// every 8 .. 9 cycles -- 36 ms for 512 x 4096^2.  Lanes 4k .. 4k + 3 share pixel k of the wave's 16:
//   * gather: lane q takes frames q R .. (q + 1) R - 1 (R = the frame-count class, a multiple of 16; wires R .. 127 are +inf pads
//     known at compile time, frames n .. 4R - 1 read a plane of +inf); every lane loads ITS plane's pointer from the table (a vector
//     load of 8 bytes) and then its sample: two memory instructions and a 64-bit add per sample, nothing fetched twice;
//   * sort: SortNet<128>::sort_fused_n<R> per lane; level 1 merges lanes (0, 1) and (2, 3) as stack_duo.hip does (cross step
//     against the partner's reversed registers + an in-lane bitonic merge); level 2 merges the two sorted runs of 256: a cross
//     step against lane q ^ 3's reversed registers, one half-cleaner stage between lanes q and q ^ 1, the in-lane merge again.
//     An exchange keeps v_med3(x, partner, -inf / +inf): the minimum or the maximum by a per-lane constant, one instruction.
//     Lane q then holds sorted ranks 128 q .. 128 q + 127;
//   * median / MAD at a compile-time position (n / 2 is one number per launch): lanes 0 and 1 hold the windows' low ends V[p], the
//     high ends V[p + M] come from lane q + 1 or q + 2 through DPP at a constant register index;
//   * clipping, running moments, the list of pixels handed to stack_pair.hip's oracle-arithmetic kernel: as in stack_duo.hip, the
//     low end in lane 0's first registers, the high end in the lane that holds rank n - 1.
// Same contract as the <= 64-frame fast engine: 1e-5 relative, at most 1e-4 of the pixels may differ from the oracle at all.
//
// 513 .. 1024 frames: the same kernel with EIGHT lanes per pixel (L = 8; lanes 8k .. 8k + 7, pixel k of the wave's 8): a third
// merge level (cross step against lane 7 - q's reversed registers = DPP row_half_mirror, half-cleaner stages between lanes q ^ 2
// and q ^ 1, the in-lane merge), sums over three DPP steps, a window's far end `row_shl` 2 .. 5 lanes on, the median's lane through
// ds_bpermute.  The wave-per-pixel kernel (stack_wide.hip: a bitonic sort through 64-lane shuffles) took 43.7 ms for 513 x 2048^2
// where 512 frames take 5: its pixels now are only the ones this pass hands over.
#include "stack_pair.hpp"

#include <algorithm>
#include <cmath>

using namespace abpair;

namespace {

constexpr int HQ = 128;  // samples per lane

template <int CTRL>
__device__ __forceinline__ float dppf(float x) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), CTRL, 0xf, 0xf, false));
}
template <int CTRL>
__device__ __forceinline__ int dppi(int x) {
    return __builtin_amdgcn_update_dpp(0, x, CTRL, 0xf, 0xf, false);
}
constexpr int kSwap1 = 0xB1;  // quad_perm [1,0,3,2]: lane q ^ 1
constexpr int kSwap2 = 0x4E;  // quad_perm [2,3,0,1]: lane q ^ 2
constexpr int kRev = 0x1B;    // quad_perm [3,2,1,0]: lane q ^ 3
constexpr int kHalfMirror = 0x141;  // row_half_mirror: lane 7 - i of every 8
// sums / ors / minima over the L = 4 or 8 lanes of a pixel: after the steps inside a quad every lane of it holds the quad's value,
// and the half mirror pairs a lane with one of the other quad
template <int L>
__device__ __forceinline__ int grp_sum(int x) {
    x += dppi<kSwap1>(x);
    x += dppi<kSwap2>(x);
    if constexpr (L == 8) x += dppi<kHalfMirror>(x);
    return x;
}
template <int L>
__device__ __forceinline__ int grp_or(int x) {
    x |= dppi<kSwap1>(x);
    x |= dppi<kSwap2>(x);
    if constexpr (L == 8) x |= dppi<kHalfMirror>(x);
    return x;
}
template <int CTRL>
__device__ __forceinline__ double dppd(double x) {
    const unsigned long long u = __builtin_bit_cast(unsigned long long, x);
    const unsigned int lo = (unsigned int)dppi<CTRL>((int)(unsigned int)u), hi = (unsigned int)dppi<CTRL>((int)(unsigned int)(u >> 32));
    return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
}
// (every step adds the same two numbers in both lanes of a pair: all lanes of the group end with the same bits)
template <int L>
__device__ __forceinline__ double grp_sum(double x) {
    x += dppd<kSwap1>(x);
    x += dppd<kSwap2>(x);
    if constexpr (L == 8) x += dppd<kHalfMirror>(x);
    return x;
}
template <int L>
__device__ __forceinline__ float grp_min(float x) {
    x = fminf(x, dppf<kSwap1>(x));
    x = fminf(x, dppf<kSwap2>(x));
    if constexpr (L == 8) x = fminf(x, dppf<kHalfMirror>(x));
    return x;
}
// lane k's register for every lane of the group
template <int L, int KLANE>
__device__ __forceinline__ float grp_bcast(float x, int lane) {
    if constexpr (L == 4) {
        return dppf<KLANE * 0x55>(x);  // quad_perm [k,k,k,k]
    } else {
        return __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(((lane & ~7) | KLANE) << 2, __builtin_bit_cast(int, x)));
    }
}

__device__ __forceinline__ double sqrt_for_sigma(double v) {  // stack_sigma_clip.hip
    const double y = __builtin_amdgcn_rsq(v);
    const double g = v * y;
    const double e = __builtin_fma(-g, g, v);
    const double r = __builtin_fma(e, 0.5 * y, g);
    return v > 0.0 ? r : 0.0;
}
__device__ __forceinline__ void nop_fence(float &x) { asm volatile("s_nop 1" : "+v"(x)); }

// v[i] against the partner lane's v[127 - i]: a lane keeps the smaller (sel = -inf) or the larger (sel = +inf) of each pair
template <int CTRL>
__device__ __forceinline__ void cross_rev(float (&v)[HQ], float sel) {
#pragma unroll
    for (int i = 0; i < HQ / 2; ++i) {
        const float t1 = dppf<CTRL>(v[HQ - 1 - i]), t2 = dppf<CTRL>(v[i]);
        const float a = ab_v_med3(v[i], t1, sel), b = ab_v_med3(v[HQ - 1 - i], t2, sel);
        v[i] = a;
        v[HQ - 1 - i] = b;
    }
}
// v[i] against the partner lane's v[i]: the half-cleaner stage of distance 128 of a bitonic merge over two lanes
template <int CTRL>
__device__ __forceinline__ void cross_same(float (&v)[HQ], float sel) {
#pragma unroll
    for (int i = 0; i < HQ; ++i) {
        const float t = dppf<CTRL>(v[i]);
        v[i] = ab_v_med3(v[i], t, sel);
    }
}

// median (combine.rs:38-40) and MAD (combine.rs:42-46) of a group of L lanes that holds n finite samples, n / 2 = M: rank r in lane
// r >> 7, register r & 127.  MAD = min over p = 0 .. n - 1 - M of max(med - V[p], V[p + M] - med) (stack_duo.hip).  The window p
// belongs to the lane that holds V[p] (lanes below M >> 7: every register; lane M >> 7: registers below M & 127); its high end is
// register (i + M) & 127 of the lane (M >> 7) or (M >> 7) + 1 places on: DPP row_shl.
template <int L, int M>
__device__ __forceinline__ void med_mad_at(const float (&v)[HQ], int q, int lane, bool n_is_odd, float &med_out, float &mad_out) {
    static_assert(M >= 32 * L && M <= 64 * L, "a group of L lanes holds 64 L + 1 .. 128 L samples");
    constexpr int Mq = M >> 7, Mr = M & 127;
    const float med = grp_bcast<L, Mq>(v[Mr], lane);
    const float inf = __builtin_inff();
    float best_lo = inf /* windows from registers below Mr */, best_hi = inf /* from the others */;
#pragma unroll
    for (int i = 0; i < HQ; ++i) {
        const int s2 = (i + Mr) & 127;
        const float hi_end = ((i + Mr) >> 7) ? dppf<0x100 + Mq + 1>(v[s2]) : dppf<0x100 + Mq>(v[s2]);  // row_shl: lane q + Mq (+ 1)
        const float t = ab_v_max(med - v[i], hi_end - med);
        if (i < Mr)
            best_lo = ab_v_min(best_lo, t);
        else
            best_hi = ab_v_min(best_hi, t);
    }
    float x = q < Mq ? ab_v_min(best_lo, best_hi) : (q == Mq ? best_lo : inf);
    if (n_is_odd) {  // (launch-uniform) n = 2M + 1: the window [M, 2M], whose larger deviation is V[2M] - med
        constexpr int T = 2 * M;
        if constexpr (T < L * HQ) {
            const float e = grp_bcast<L, (T >> 7)>(v[T & 127], lane) - med;
            x = q == 0 ? fminf(x, e) : x;
        }
    }
    x = x + 0.0f;  // (a compiler-visible VALU write: the DPP reads below are then the hazard recogniser's business)
    med_out = med;
    mad_out = grp_min<L>(x);
}
template <int L, int LO, int HI>
__device__ __forceinline__ void med_mad_dispatch(const float (&v)[HQ], int q, int lane, bool n_is_odd, int M /* launch-uniform */, float &med, float &mad) {
    if constexpr (LO == HI) {
        med_mad_at<L, LO>(v, q, lane, n_is_odd, med, mad);
    } else {
        constexpr int MID = (LO + HI) / 2;
        if (M <= MID)
            med_mad_dispatch<L, LO, MID>(v, q, lane, n_is_odd, M, med, mad);
        else
            med_mad_dispatch<L, MID + 1, HI>(v, q, lane, n_is_odd, M, med, mad);
    }
}

// the median alone (median_combine_row_major, calibration.rs:106-124: sorted[len / 2]): rank M = lane M >> 7, register M & 127
template <int L, int LO, int HI>
__device__ __forceinline__ float median_dispatch(const float (&v)[HQ], int lane, int M /* launch-uniform */) {
    if constexpr (LO == HI) {
        return grp_bcast<L, (LO >> 7)>(v[LO & 127], lane);
    } else {
        constexpr int MID = (LO + HI) / 2;
        return M <= MID ? median_dispatch<L, LO, MID>(v, lane, M) : median_dispatch<L, MID + 1, HI>(v, lane, M);
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

// One clipping pass over the two ends of the quad's rank interval (combine.rs:65-82; stack_duo.hip: clip_walk).  [la, lb]: this
// lane's part of it in its own register indices.  The low end is lane 0's registers 0 .. 7; the high end lane qt's last real
// registers (chunks ct and ct - 1; qt, ct launch-uniform) and, when that lane holds at most eight samples, lane qt - 1's top.
template <int L, bool UPDATE>
__device__ __forceinline__ void clip_walk(const float (&v)[HQ], int q, bool go, int la, int lb, int qt, int ct, float center, float lo, float hi, float c0,
                                          double c0d, int &cl_own, int &ch_own, bool &decided, double &e_rem, double &q_rem) {
    constexpr int NC = HQ / 4;
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
    const bool lo_lane = go && q == 0, hi_lane = go && q == qt, hi2_lane = go && q == qt - 1;
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
    // (whether some pixel of the wave still looks for its first surviving sample at the high end: the quad's lanes share the flag)
    auto open_hi = [&]() { return __any(go && grp_or<L>(found_hi ? 1 : 0) == 0); };
#pragma unroll
    for (int c = NC - 1; c >= 0; --c) {
        if (c != ct && c != ct - 1) continue;  // (uniform)
        if (c == ct - 1 && !open_hi()) continue;  // the first chunk settled it for every pixel of the wave
        high_chunk(c, hi_lane);
    }
    if (ct <= 1) {  // (uniform) lane qt holds at most eight samples and the two chunks above were all of them
        if (open_hi()) {
            high_chunk(NC - 1, hi2_lane);
            if (open_hi()) high_chunk(NC - 2, hi2_lane);
        }
    }
    const int f_lo = (q == 0 && found_lo) ? 1 : 0, f_hi = found_hi ? 1 : 0;
    decided = (grp_or<L>(f_lo) & grp_or<L>(f_hi)) != 0;
    cl_own = cl;
    ch_own = ch;
}

template <int L, int R, bool MEDIAN = false>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2, 2))) void stack_quad_fast_kernel(const PairArgs a) {
    static_assert((L == 4 || L == 8) && R <= HQ && R > HQ / 2 && R % 8 == 0, "frame-count class");
    constexpr int PX = 64 / L;  // pixels per wave
    const int lane = threadIdx.x;
    const int q = lane & (L - 1);
    const int pix = lane / L;
    // A wave reads 16 (8) pixels = 64 (32) bytes of every plane: a HALF (quarter) of a cache line, the rest being the next waves'.
    // Workgroup ids go round the eight XCDs (id % 8), so consecutive ids would fetch every line into several XCDs' L2; ids id, id + 8,
    // ... -- same XCD, dispatched together -- take neighbouring pixel groups instead (the grid is a multiple of 8).
    const unsigned int per_xcd = gridDim.x >> 3;
    const unsigned int group = (blockIdx.x & 7u) * per_xcd + (blockIdx.x >> 3);
    int64_t g = (int64_t)group * PX + pix;
    const bool valid = g < a.total;
    if (!valid) g = a.total - 1;

    // ---- gather (combine.rs:170-175): lane q's frames are table entries q R .. q R + R - 1 (entries past n: a plane of +inf) ----
    float v[HQ];
    const int have = min(max(a.n - q * R, 0), R);  // real frames of this lane
    const uint32_t boff = (uint32_t)g * 4u;
    const float *const *tab = a.p + q * R;
#pragma unroll
    for (int f = 0; f < HQ; ++f) {
        if (f < R)
            v[f] = *(const float *)((const char *)tab[f] + boff);
        else
            v[f] = __builtin_inff();  // a pad of the class: never loaded, never moved by the network
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
    bool defer = grp_sum<L>(full) != L;

    // ---- sort: R per lane, then log2 L merge levels ----
    const float inf = __builtin_inff();
    const float sel1 = (q & 1) ? inf : -inf, sel2 = (q & 2) ? inf : -inf;
    if constexpr (R < HQ)
        SortNet<HQ>::template sort_fused_n<R>(v, [](auto) {});
    else
        SortNet<HQ>::sort_fused(v);
    dpp_fence<HQ>(v);
    cross_rev<kSwap1>(v, sel1);  // lanes (0, 1), (2, 3), ...: runs of 256
    bitonic_merge<HQ>(v);
    dpp_fence<HQ>(v);
    cross_rev<kRev>(v, sel2);  // quads: runs of 512
    dpp_fence<HQ>(v);
    cross_same<kSwap1>(v, sel1);
    bitonic_merge<HQ>(v);
    dpp_fence<HQ>(v);
    if constexpr (L == 8) {  // the two quads: one run of 1024
        const float sel4 = (q & 4) ? inf : -inf;
        cross_rev<kHalfMirror>(v, sel4);
        dpp_fence<HQ>(v);
        cross_same<kSwap2>(v, sel2);
        dpp_fence<HQ>(v);
        cross_same<kSwap1>(v, sel1);
        bitonic_merge<HQ>(v);
        dpp_fence<HQ>(v);
    }

    // ---- median / MAD: n / 2 is one number per launch ----
    constexpr int kMHi = L * R / 2, kMLo = kMHi - 8 * L > 32 * L ? kMHi - 8 * L : 32 * L;  // n / 2 over the class's frame counts
    if constexpr (MEDIAN) {  // median_combine_row_major (calibration.rs:84-125): the pixels with a non-finite sample go to the list pass
        const float m = median_dispatch<L, kMLo, kMHi>(v, lane, a.n >> 1);
        const bool writer = valid && q == 0;
        if (writer && !defer) a.out[g] = m;
        hand_over(a, writer && defer, lane, g);
        return;
    }
    float med, mad;
    med_mad_dispatch<L, kMLo, kMHi>(v, q, lane, (a.n & 1) != 0, a.n >> 1, med, mad);

    // ---- iteration 0: clip about the median with the MAD sigma (combine.rs:37-48,63-82) ----
    const int qt = (a.n - 1) >> 7, t_top = (a.n - 1) & 127;  // rank n - 1: lane qt, register t_top
    const int ct = t_top >> 2;
    int la = 0, lb = q < qt ? HQ - 1 : (q == qt ? t_top : -1);  // this lane's part of the survivors
    int len = a.n;
    uint32_t rej = 0;
    float last_center = __builtin_nanf("");
    bool active = !defer;
    const float c0 = med;
    const double c0d = (double)med;
    double e_rem = 0.0, q_rem = 0.0;
    auto apply = [&](bool go, int cl_own, int ch_own, bool decided) {
        const int cl = grp_sum<L>(cl_own), ch = grp_sum<L>(ch_own);
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
        clip_walk<L, false>(v, q, go, la, lb, qt, ct, med, -a.sigma_low * sigma, a.sigma_high * sigma, c0, c0d, cl_own, ch_own, decided, e_rem, q_rem);
        apply(go, cl_own, ch_own, decided);
    }

    // ---- one pass over the survivors: E = sum e_i, Q = sum e_i^2 with e_i = x_i - med (f64), the quad's parts added ----
    double E, Q;
    {
        double E1 = 0.0, Q1 = 0.0;
#pragma unroll
        for (int i = 0; i < HQ; ++i) {
            const bool in = (i >= la) && (i <= lb);
            float xm = in ? v[i] : c0;
            asm volatile("" : "+v"(xm));  // select the f32 sample (or the compiler selects the two halves of the f64)
            const double e = (double)xm - c0d;
            E1 += e;
            Q1 = __builtin_fma(e, e, Q1);
        }
        E = grp_sum<L>(E1);
        Q = grp_sum<L>(Q1);
    }

    // ---- iterations >= 1: mean / sigma from the running sums (combine.rs:50-82; stack_sigma_clip.hip: clip_fast_tail) ----
    for (uint32_t it = 1; it < a.max_iter; ++it) {
        if (!__any(active)) break;
        launder<HQ>(v);  // stop LICM from hoisting the f32->f64 conversions out of this loop
        const double er = grp_sum<L>(e_rem), qr = grp_sum<L>(q_rem);
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
        clip_walk<L, true>(v, q, go, la, lb, qt, ct, center, -a.sigma_low * sigma, a.sigma_high * sigma, c0, c0d, cl_own, ch_own, decided, e_rem, q_rem);
        apply(go, cl_own, ch_own, decided);
    }

    // ---- result (combine.rs:85-91) ----
    const double er = grp_sum<L>(e_rem);
    const double S = __builtin_fma((double)len, c0d, E - er);
    const float mean_f = (float)(S / (double)(len > 0 ? len : 1));
    const float value = len > 0 ? mean_f : (__builtin_isfinite(last_center) ? last_center : 0.0f);
    const bool writer = valid && q == 0;
    if (writer && !defer) a.out[g] = value;

    hand_over(a, writer && defer, lane, g);

    // rejection count: one atomic per wave, spread over kRejSlots counters (summed by the host)
    int r = (writer && !defer) ? (int)rej : 0;
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) r += __shfl_xor(r, off, 64);
    if (lane == 0 && r != 0) atomicAdd(&a.rejected[blockIdx.x & (kRejSlots - 1)], (unsigned long long)r);
}

}  // namespace

// the fast pass of a 257 .. 512-frame (L = 4 lanes per pixel) or 513 .. 1024-frame (L = 8) stack; the arguments' table holds L R pointers
int ab_stack_quad_launch(ab_ctx *ctx, int L, int R, const PairArgs &args) {
    const int64_t px = 64 / L;
    const dim3 grid((unsigned)(((args.total + px - 1) / px + 7) / 8 * 8)), block(64);  // (a multiple of 8: see the kernel's pixel-group order)
#define AB_QUAD_CASE(LV, RV)                                                                              \
    if (L == LV && R == RV) {                                                                             \
        if (args.median_only)                                                                             \
            hipLaunchKernelGGL((stack_quad_fast_kernel<LV, RV, true>), grid, block, 0, ctx->stream, args); \
        else                                                                                              \
            hipLaunchKernelGGL((stack_quad_fast_kernel<LV, RV>), grid, block, 0, ctx->stream, args);       \
        AB_HIP(ctx, hipGetLastError());                                                                   \
        return AB_OK;                                                                                     \
    }
#ifndef AB_QUAD_ONE_CLASS  // (tests/test_abi_cpu.py walks the listing of ONE instance of each lane count)
    AB_QUAD_CASE(4, 80)
    AB_QUAD_CASE(4, 96)
    AB_QUAD_CASE(4, 128)
    AB_QUAD_CASE(8, 80)
    AB_QUAD_CASE(8, 96)
    AB_QUAD_CASE(8, 128)
#endif
    AB_QUAD_CASE(4, 112)
    AB_QUAD_CASE(8, 112)
#undef AB_QUAD_CASE
    return ab_set_error(ctx, AB_ERR_INVALID, "internal: no %d-lane kernel for class %d", L, R);
}
