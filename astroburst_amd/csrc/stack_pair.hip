// Per-pixel kappa-sigma stacking of 257 .. 512 contiguous frames on gfx950: TWO LANES PER PIXEL.
//
// sigma_clip_combine (core/stacking/combine.rs:14-92) and median_combine_row_major (core/stacking/calibration.rs:84-125) for the
// frame counts of master calibration stacks (calibration.rs:47-125).  stack_sigma_clip.hip keeps a pixel's samples in ONE lane
// (up to 256: the whole unified register file); stack_wide.hip gives a pixel a whole wave (3000 instructions per pixel, 180 ms
// for 4096^2 x 512).  Here lanes 2k and 2k+1 share pixel k of the wave's 32:
//   * the even lane gathers frames 0 .. 255, the odd lane frames 256 .. 511 (non-finite samples and absent frames become
//     +inf pads), and each sorts its 256 samples in registers with the same network as the one-lane kernel (SortNet<256>);
//   * one cross step -- v[i] against the partner's v[255 - i], fetched by DPP quad_perm [1,0,3,2] -- leaves the 256 smallest
//     in the even lane and the 256 largest in the odd lane, each a bitonic sequence, and an in-lane bitonic merge (8 half-
//     cleaner stages on constant register indices) sorts them: the pair now holds sorted ranks 0 .. 511, pads on top;
//   * the median and the MAD pairing (term(p) = max(med - V[p], V[p + n/2] - med), the k-th element of the merge of the
//     two deviation runs, as in stack_sigma_clip.hip) read ranks at a per-pixel offset: with all 512 samples finite the
//     offset is 256 and the partner's register p IS V[p + 256] (DPP); otherwise the ranks travel through LDS, one half at
//     a time (32 KB per wave);
//   * the clipping iterations are the oracle's arithmetic word for word (two-pass mean / squared deviations, f64, ascending):
//     the survivors are a rank interval, a sum runs over the even lane's part first and is handed to the odd lane, which
//     continues it -- the same sequence of f64 additions as one lane walking all 512 ranks, so the result is BIT-IDENTICAL to
//     the wave-per-pixel kernel and to the CPU restatement (ORC_ORDER_ASCENDING), not merely within the 1e-5 contract.
// ~25 000 instructions per wave of 32 pixels, 512 load instructions of 128 B each; one wave per SIMD (the samples fill the
// register file), 32 KB of LDS per wave.  Measured (4096^2, round 4): 512 frames 59.5 ms, 320 frames 64 ms (the LDS paths), where
// the wave-per-pixel kernel takes 180 .. 375 ms -- 117 us per wave for ~32 000 executed instructions = 8 .. 9 cycles per
// instruction: ONE wave per SIMD cannot issue faster (tools/valu_rate.hip's rates need two), so fewer instructions do not buy
// time here.  Tried and backed out: sums taken eight samples at a time with wave-uniform inside / outside tests and counts that
// stop at the first surviving sample (-35 % of the clip's instructions: 60.4 ms, and 82 ms for 320 frames -- the uniform masks
// spill scalar registers); the maximum of an exchange as a ^ b ^ min (v_bitop3_b32, full rate) -- no change; four-wave
// workgroups kept in step by a barrier every 128 exchanges so that the 60 KB of straight-line sort is fetched once per
// workgroup (70.7 ms: the instruction cache was not the limit).  What would help is two waves per SIMD: four lanes per pixel
// with 128 samples each.
#include "stack_pair.hpp"

#include <algorithm>
#include <cmath>

using namespace abpair;

namespace {

// One f64 accumulation over the global rank interval [a, b] (ranks 0 .. 511 over the pair), ascending: the even lane's part
// first, then the odd lane continues from the even lane's sum.  la / lb: this lane's part of the interval in local indices
// (la = lb = -1: none).  MODE 0: sum of v; MODE 1: sum of (v - mean)^2.  Both lanes return the pair's total.
template <int H, int MODE>
__device__ __forceinline__ double pair_sum(float (&v)[H], int la, int lb, bool odd, double mean) {
    double S = 0.0;
#pragma unroll 1
    for (int phase = 0; phase < 2; ++phase) {
        launder<H>(v);  // (or the H conversions, the same in both phases, are hoisted out of this loop: 512 registers, all spilled)
        // phase 0: even lanes add their samples, odd lanes add zeros; phase 1: the odd lane continues the even lane's sum
        if (phase == 1) {
            const double from_even = swapd(S);  // (evaluated by ALL lanes: a DPP read of a lane that is switched off returns nothing)
            S = odd ? from_even : S;
        }
        const bool mine = odd == (phase == 1);
        const int pa = mine ? la : -1, pb = mine ? lb : -1;
#pragma unroll
        for (int i = 0; i < H; ++i) {
            const bool in = (unsigned)(i - pa) <= (unsigned)(pb - pa);
            if (MODE == 0) {
                const float xs = in ? v[i] : 0.0f;  // (x + 0.0 is exact)
                S += (double)xs;
            } else {
                const double dd = (double)v[i] - mean;
                const double sq = dd * dd;
                S += in ? sq : 0.0;
            }
        }
    }
    const double from_odd = swapd(S);
    return odd ? S : from_odd;
}

// ranks through LDS: `buf` holds one half (256 ranks) of each of the wave's 32 pixels, rank-major (conflict-free: a wave's
// accesses to one rank are 32 consecutive words)
template <int H>
__device__ __forceinline__ void put_half(float *buf, const float (&v)[H], bool writer, int pix) {
    if (writer) {
#pragma unroll
        for (int i = 0; i < H; ++i) buf[i * 32 + pix] = v[i];
    }
}

// one wave's 32 pixels (g: this pair's pixel; valid: it exists).  Returns the pixel's rejected-sample count in the even lane.
template <int H, bool MEDIAN_ONLY>
__device__ __forceinline__ uint32_t pair_pixel(const PairArgs &a, float *buf, int64_t g, bool valid) {
    const int lane = threadIdx.x;
    const bool odd = lane & 1;
    const int pix = lane >> 1;

    // ---- gather (combine.rs:170-175 / calibration.rs:95-104): only finite samples take part; the rest are +inf pads ----
    // lane-parity halves: two exec-masked load streams of H instructions, 32 lanes x 4 B = one 128-byte line each
    // Every load instruction has a wave-uniform plane (scalar base + one shared 32-bit byte offset, no vector address work): BOTH
    // lanes of a pair fetch their pixel from frame f and from frame f + H (the two lanes read the same word: one 128-byte
    // line per instruction either way) and each keeps its half's sample.
    float v[H];
    const int have = odd ? a.n - H : H;  // frames of this half (n > H: the even half is full)
    const uint32_t boff = (uint32_t)g * 4u;
#pragma unroll
    for (int f = 0; f < H; ++f) {
        const float xe = *(const float *)((const char *)a.p[f] + boff);
        const float xo = *(const float *)((const char *)a.p[f + H] + boff);  // (table entries past n point at a plane of +inf: pads)
        v[f] = odd ? xo : xe;
    }
    float nf = 0.0f;  // fma(x, 0, nf) stays 0 for finite x and turns NaN for inf / NaN
#pragma unroll
    for (int f = 0; f < H; ++f) nf = __builtin_fmaf(f < have ? v[f] : 0.0f, 0.0f, nf);
    int n_lane = have;
    if (__any(nf != nf)) {  // rare: some lane of this wave met a non-finite sample
        n_lane = 0;
#pragma unroll
        for (int f = 0; f < H; ++f) {
            const bool fin = __builtin_isfinite(v[f]);
            v[f] = fin ? v[f] : __builtin_inff();
            n_lane += fin ? 1 : 0;
        }
    }
    const int n = n_lane + swapi(n_lane);  // the pixel's finite samples: sorted ranks [0, n)

    // ---- sort: H per lane, cross step, in-lane bitonic merge ----
    SortNet<H>::sort_fused(v);  // the rewrite over min3 / med3 / max3 (tools/gen_sortnet.py): 5493 instructions for 256, not 7486
    dpp_fence<H>(v);
    cross_step<H>(v, odd);
    if constexpr (MEDIAN_ONLY) {
        // calibration.rs:106-124: 0 for no finite sample, else sorted[len / 2].  All 2H finite: rank H = the smallest of the odd
        // lane's half, no merge needed
        if (__all(n == 2 * H)) {
            float m = v[0];
#pragma unroll
            for (int i = 1; i < H; ++i) m = ab_v_min(m, v[i]);
            if (valid && odd) a.out[g] = m;
            return 0;
        }
    }
    bitonic_merge<H>(v);
    dpp_fence<H>(v);

    // ---- median (combine.rs:38-40) and MAD (combine.rs:42-46) ----
    const int M = n >> 1;
    float med, mad;
    if (__all(n == 2 * H)) {
        const float o0 = swapf(v[0]);
        med = odd ? v[0] : o0;  // rank H
        // term(p) = max(med - V[p], V[p + H] - med), p = 0 .. H - 1: V[p] is the even lane's v[p], V[p + H] the odd lane's
        float best = __builtin_inff();
#pragma unroll
        for (int p = 0; p < H; ++p) {
            const float d = odd ? v[p] - med : med - v[p];
            best = ab_v_min(best, ab_v_max(d, swapf(d)));
        }
        mad = best;
    } else {
        // per-pixel offsets: the ranks travel through LDS, the odd lane's half first.  The even lane evaluates
        //   med = V[M];  best = med - V[0];  for p = 1 .. M with p + M < 2H: best = min(best, max(med - V[p], V[p + M] - med))
        // (ranks >= n are +inf pads and drop out by themselves; p <= H - 1 because a term needs p + M <= n - 1 <= 2H - 1)
        float medv = 0.0f, best = __builtin_inff();
        __syncthreads();  // (LIST mode: the previous pixels' reads of buf are done)
        put_half<H>(buf, v, odd, pix);
        __syncthreads();
        if (M >= H && M < 2 * H) medv = buf[(M - H) * 32 + pix];
        if (__any(M < H)) {  // (wave-uniform) some pixel's median and near partners lie in the EVEN half
            __syncthreads();
            put_half<H>(buf, v, !odd, pix);
            __syncthreads();
            if (M < H) medv = buf[M * 32 + pix];
            if constexpr (!MEDIAN_ONLY) {
                if (!odd) {
#pragma unroll
                    for (int p = 1; p < H; ++p) {  // (branch-free: the read is clamped, the term selected)
                        const int r = p + M;
                        const float w = buf[min(r, H - 1) * 32 + pix];
                        const float t = fminf(best, fmaxf(medv - v[p], w - medv));
                        best = (p <= M && r < H) ? t : best;
                    }
                }
            }
            __syncthreads();
            put_half<H>(buf, v, odd, pix);
            __syncthreads();
        }
        if constexpr (!MEDIAN_ONLY) {
            if (!odd) {  // the terms whose partner rank lies in the ODD half (which is what buf holds now)
#pragma unroll
                for (int p = 1; p < H; ++p) {
                    const int r = p + M - H;
                    const float w = buf[min(max(r, 0), H - 1) * 32 + pix];
                    const float t = fminf(best, fmaxf(medv - v[p], w - medv));
                    best = (p <= M && r >= 0 && r < H) ? t : best;
                }
                best = fminf(best, medv - v[0]);
            }
        }
        const float em = swapf(medv), eb = swapf(best);  // (the even lane's)
        med = odd ? em : medv;
        mad = odd ? eb : best;
    }

    if constexpr (MEDIAN_ONLY) {
        if (valid && !odd) a.out[g] = (n == 0) ? 0.0f : med;
        return 0;
    }

    // ---- clipping iterations (combine.rs:31-83), the oracle's arithmetic: two-pass mean / variance over the rank interval ----
    const int base = odd ? H : 0;
    float sigma = (float)fmax((double)mad * kMadToSigma, 1e-10);
    float center = med;
    int ra = 0, rb = n - 1, len = n;  // survivors: global ranks ra .. rb
    uint32_t rej = 0;
    float last_center = __builtin_nanf("");
    bool active = n >= 2;
    for (uint32_t it = 0; it < a.max_iter; ++it) {
        if (!__any(active)) break;
        launder<H>(v);
        int la = max(ra - base, 0), lb = min(rb - base, H - 1);  // this lane's part of the interval
        if (ra > rb || la > lb) la = lb = -1;                    // none: (unsigned)(i + 1) <= 0 holds for no i >= 0
        if (it > 0) {
            const double S = pair_sum<H, 0>(v, la, lb, odd, 0.0);
            const double nn = (double)len;
            const double mean = S / nn;
            opaque(la, lb);
            const double Q = pair_sum<H, 1>(v, la, lb, odd, mean);
            const double variance = Q / fmax(nn - 1.0, 1.0);
            center = (float)mean;
            sigma = (float)fmax(sqrt(variance), 1e-10);
            opaque(la, lb);
        }
        const bool go = active && (len >= 2);  // `if len < 2 { break }` (combine.rs:33-35)
        if (go) last_center = center;          // combine.rs:63
        const float lo = -a.sigma_low * sigma;  // combine.rs:65-66
        const float hi = a.sigma_high * sigma;
        int cl = 0, ch = 0;
#pragma unroll
        for (int i = 0; i < H; ++i) {
            const bool in = (unsigned)(i - la) <= (unsigned)(lb - la);
            const float dev = v[i] - center;
            cl += (in && !(dev >= lo)) ? 1 : 0;
            ch += (in && !(dev <= hi)) ? 1 : 0;
        }
        cl += swapi(cl);
        ch += swapi(ch);
        // a sample can fail both tests only if nothing survives (lo > hi or NaN thresholds)
        const int removed = (cl + ch > len) ? len : (cl + ch);
        if (go) {
            rej += (uint32_t)removed;  // combine.rs:76-78
            len -= removed;
            if (len > 0) {
                ra += cl;
                rb -= ch;
            } else {
                ra = 1;
                rb = 0;
            }
        }
        active = go && (removed != 0);  // combine.rs:80-82
    }

    // ---- result (combine.rs:20-26,85-91) ----
    launder<H>(v);
    int la = max(ra - base, 0), lb = min(rb - base, H - 1);
    if (ra > rb || la > lb) la = lb = -1;
    opaque(la, lb);
    const double S = pair_sum<H, 0>(v, la, lb, odd, 0.0);  // an empty interval sums to 0
    float value;
    if (n == 0)
        value = 0.0f;
    else if (n == 1)
        value = med;  // the single finite sample is V[0] = V[n / 2]
    else if (len == 0)
        value = __builtin_isfinite(last_center) ? last_center : 0.0f;
    else
        value = (float)(S / (double)len);
    if (valid && !odd) a.out[g] = value;
    return (valid && !odd) ? rej : 0u;
}

// H = 256: one wave per SIMD (the samples fill the register file); H = 128 (the list pass of a 129 .. 256-frame stack): two
template <int H, bool MEDIAN_ONLY>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(H == 128 ? 2 : 1, H == 128 ? 2 : 1))) void stack_pair_kernel(const PairArgs a) {
    __shared__ float buf[H * 32];
    const int pix = threadIdx.x >> 1;
    int r = 0;
    if (a.walk_lists) {
        // LIST mode: the pixels stack_duo.hip's fast pass handed over, kListWaves one-wave workgroups per list
        const unsigned int slot = blockIdx.x / kListWaves, sub = blockIdx.x % kListWaves;
        const unsigned int cnt = a.list_count[slot];
        const int *list = a.list + (size_t)slot * a.list_cap;
#pragma unroll 1
        for (unsigned int base = sub * 32u; base < cnt; base += 32u * kListWaves) {
            const unsigned int k = base + (unsigned int)pix;
            const bool valid = k < cnt;
            r += (int)pair_pixel<H, MEDIAN_ONLY>(a, buf, (int64_t)list[valid ? k : cnt - 1], valid);
        }
        // the last of the list's workgroups to get here leaves the list empty for the next launch (stack_sigma_clip.hip: no fence)
        if (threadIdx.x == 0) {
            if (atomicAdd(&a.list_ticket[slot], 1u) == kListWaves - 1) {
                a.list_ticket[slot] = 0;
                a.list_count[slot] = 0;
            }
        }
    } else {
        int64_t g = (int64_t)blockIdx.x * 32 + pix;
        const bool valid = g < a.total;
        if (!valid) g = a.total - 1;
        r = (int)pair_pixel<H, MEDIAN_ONLY>(a, buf, g, valid);
    }
    // rejection count: one atomic per wave, spread over kRejSlots counters (summed by the host)
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) r += __shfl_xor(r, off, 64);
    if (threadIdx.x == 0 && r != 0) atomicAdd(&a.rejected[blockIdx.x & (kRejSlots - 1)], (unsigned long long)r);
}

}  // namespace

// dplanes: HOST array of n device pointers (128 < n <= 1024), contiguous planes of rows x cols; counters already cleared by the caller.
// The default engine (and its median combine): a fast multi-lane pass -- two lanes per pixel for 129 .. 256 frames (stack_duo.hip), four
// for 257 .. 512, eight for 513 .. 1024 (stack_quad.hip) -- then the oracle's arithmetic over the pixels it handed over (this file's
// kernel in list mode up to 512 frames, stack_wide.hip's wave-per-pixel kernel beyond).  AB_STACK_EXACT=1 (129 .. 512 frames): this
// file's kernel over all pixels.
int ab_stack_pair_device(ab_ctx *ctx, const float *const *dplanes, size_t n, int64_t rows, int64_t cols, const ab_stack_config *cfg,
                         float *out_dev, bool median_only) {
    AB_CHECK(ctx, n > 128 && n <= 1024, "the multi-lane stack takes 129 .. 1024 frames (got %zu)", n);
    const int H = n > 256 ? 256 : 128;  // samples per lane of THIS file's kernel (129 .. 512 frames)
    const bool fast = !ctx->stack_exact;  // (the median combine too: a pixel with every sample finite needs the sort and one register)
    AB_CHECK(ctx, fast || n <= 512, "internal: the exact engine of %zu frames is the wave-per-pixel kernel's", n);
    constexpr size_t kTab = 1024;  // [0, kTab): the fast pass's table (frames, then the +inf plane); [kTab, kTab + 512): this file's kernel's
    void *ws = nullptr;
    AB_TRY(ab_workspace(ctx, AB_WS_STACK_WIDE, (kTab + 512) * sizeof(float *), &ws));
    // the tables are tiny; a blocking copy keeps the host array's lifetime out of the picture
    AB_HIP(ctx, hipStreamSynchronize(ctx->stream));
    // frames the stack is short of: ONE plane of +inf stands in for all of them (a non-finite sample is exactly what the
    // algorithm ignores, combine.rs:170-175; the pad reads stay in L2) -- every load of the kernel is unconditional
    const int64_t total = rows * cols;
    const float *inf_plane = nullptr;
    {
        float *ip = nullptr;
        const void *before = ctx->ws[AB_WS_STACK_INF];
        const size_t had = ctx->ws_bytes[AB_WS_STACK_INF];
        AB_TRY(ab_workspace(ctx, AB_WS_STACK_INF, (size_t)total * sizeof(float), (void **)&ip));
        if (ip != before || had < (size_t)total * sizeof(float))
            AB_HIP(ctx, hipMemsetD32Async((hipDeviceptr_t)ip, 0x7f800000, ctx->ws_bytes[AB_WS_STACK_INF] / sizeof(float), ctx->stream));
        inf_plane = ip;
    }
    // frame-count classes of the fast pass: R frames per lane, a multiple of 16 (H / 8 with two lanes): the pads' loads and network
    // operations vanish.  (AB_STACK_NO_QUAD=1 on a developer build: 257 .. 512 frames on two lanes with 256 samples each.)
    static const bool no_quad = ab_dev_env("AB_STACK_NO_QUAD") != nullptr;
    const bool quad = fast && n > 256 && (n > 512 || !no_quad);
    const size_t lanes = n > 512 ? 8 : (quad ? 4 : 2);
    const int cw = quad ? 16 : H / 8;
    const int R = fast ? (int)(((n + lanes - 1) / lanes + (size_t)cw - 1) / (size_t)cw) * cw : H;
    std::vector<const float *> table(kTab + 512);
    for (size_t i = 0; i < kTab; ++i) table[i] = i < n ? dplanes[i] : inf_plane;
    for (size_t i = 0; i < 512; ++i) table[kTab + i] = table[i];
    AB_HIP(ctx, hipMemcpy(ws, table.data(), table.size() * sizeof(float *), hipMemcpyHostToDevice));
    PairArgs a;
    memset(&a, 0, sizeof a);
    a.p = (const float *const *)ws + kTab;
    a.n = (int)n;
    a.half = H;
    a.total = total;
    a.sigma_low = cfg->sigma_low;
    a.sigma_high = cfg->sigma_high;
    a.max_iter = cfg->max_iterations;
    a.out = out_dev;
    a.rejected = ctx->counters;
    a.median_only = median_only ? 1 : 0;
    const dim3 grid((unsigned)((a.total + 31) / 32)), block(64);
    if (!fast) {  // the exact engine (AB_STACK_EXACT=1) and the median combine: every pixel through the oracle's arithmetic
        if (H == 128 && median_only)
            hipLaunchKernelGGL((stack_pair_kernel<128, true>), grid, block, 0, ctx->stream, a);
        else if (H == 128)
            hipLaunchKernelGGL((stack_pair_kernel<128, false>), grid, block, 0, ctx->stream, a);
        else if (median_only)
            hipLaunchKernelGGL((stack_pair_kernel<256, true>), grid, block, 0, ctx->stream, a);
        else
            hipLaunchKernelGGL((stack_pair_kernel<256, false>), grid, block, 0, ctx->stream, a);
        AB_HIP(ctx, hipGetLastError());
        return AB_OK;
    }
    // the lists: slot = wave index & (kListSlots - 1) (rotated), so a slot holds at most ceil(waves / kListSlots) waves' worth of pixels
    // (the fast pass's waves: 32 pixels each with two lanes per pixel, 16 with four and the grid rounded up to a multiple of 8)
    const int64_t px_wave = 64 / (int64_t)lanes;
    const int64_t waves = quad ? ((total + px_wave - 1) / px_wave + 7) / 8 * 8 : (total + 31) / 32;
    const unsigned int cap = (unsigned int)(((waves + kListSlots - 1) / kListSlots) * px_wave);
    char *lw = nullptr;
    const void *before = ctx->ws[AB_WS_STACK_PAIR_LISTS];
    AB_TRY(ab_workspace(ctx, AB_WS_STACK_PAIR_LISTS, (size_t)2 * kListSlots * sizeof(unsigned int) + (size_t)kListSlots * cap * sizeof(int), (void **)&lw));
    a.list_count = (unsigned int *)lw;
    a.list_ticket = a.list_count + kListSlots;
    a.list = (int *)(lw + (size_t)2 * kListSlots * sizeof(unsigned int));
    a.list_cap = cap;
    // (the list pass leaves every counter at zero again, but a call that failed between the two passes would not have: 16 KB, in
    // stream order, in front of a kernel of milliseconds)
    (void)before;
    AB_HIP(ctx, hipMemsetAsync(a.list_count, 0, 2 * kListSlots * sizeof(unsigned int), ctx->stream));
    PairArgs f = a;
    f.p = (const float *const *)ws;
    f.half = R;
    if (quad)
        AB_TRY(ab_stack_quad_launch(ctx, (int)lanes, R, f));
    else
        AB_TRY(ab_stack_duo_launch(ctx, H, R, f));
    if (ab_env("AB_TRACE")) {  // developer aid: how many pixels the fast pass handed over
        std::vector<unsigned int> cnt(kListSlots, 0);
        AB_HIP(ctx, hipMemcpyAsync(cnt.data(), a.list_count, kListSlots * sizeof(unsigned int), hipMemcpyDeviceToHost, ctx->stream));
        AB_HIP(ctx, hipStreamSynchronize(ctx->stream));
        unsigned long long tot = 0, mx = 0;
        for (unsigned int c : cnt) tot += c, mx = c > mx ? c : mx;
        ab_count_fallback(ctx, AB_FB_STACK_GENERAL_PIXELS, tot);
        fprintf(stderr, "[ab_trace] two-lane stack: %llu of %lld pixels handed to the list pass (%.2f %%), fullest list %llu of %u\n", tot, (long long)total,
                100.0 * (double)tot / (double)total, mx, cap);
    }
    if (n > 512)  // the wave-per-pixel kernel walks the lists
        return ab_stack_wide_list_device(ctx, (const float *const *)ws, n, rows, cols, cfg, out_dev, median_only, a.list_count, a.list, cap);
    a.walk_lists = 1;
    const dim3 lgrid(kListSlots * kListWaves);
    if (H == 128 && median_only)
        hipLaunchKernelGGL((stack_pair_kernel<128, true>), lgrid, block, 0, ctx->stream, a);
    else if (H == 128)
        hipLaunchKernelGGL((stack_pair_kernel<128, false>), lgrid, block, 0, ctx->stream, a);
    else if (median_only)
        hipLaunchKernelGGL((stack_pair_kernel<256, true>), lgrid, block, 0, ctx->stream, a);
    else
        hipLaunchKernelGGL((stack_pair_kernel<256, false>), lgrid, block, 0, ctx->stream, a);
    AB_HIP(ctx, hipGetLastError());
    return AB_OK;
}
