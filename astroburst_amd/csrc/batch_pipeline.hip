// Batch calibration pipeline on gfx950: core/imaging/calibration_pipeline.rs.
//
//   calibrate_light (:74-118), normalize_frames (:309-319), sigma_clipped_mean_stack (:321-378), the per-channel body
//   of run_batch_pipeline (:157-190), normalize_channel / apply_luminance / compose_rgb_from_masters (:201-307).
//
// The reference materialises every calibrated frame, then every normalised frame, then stacks them: 6 full passes over
// the light frames.  Here a channel costs TWO reads of the lights and nothing else of that size:
//   1. cal_means_kernel   -- per-frame f64 sum of the calibrated samples (masters read once per pixel, 64 frames deep)
//   2. scms_kernel<CAL>   -- the stack kernel re-applies bias / dark / flat / inv_mean in its gather, in the reference's
//                            f32 operation order, so the samples it clips are bit-identical to the reference's frames.
//
// The clipping differs from combine.rs (stack_sigma_clip.hip): median and MAD are recomputed on EVERY iteration, the
// z-test is strict on both sides, NaN samples take part (and are always rejected by a retain pass), rejections are
// counted per frame, and the result is an f32 mean summed in FRAME order.  Mapping:
//   * one lane = one pixel; its n samples stay in VGPRs in frame order (u[]) for the final frame-order sum;
//   * a sorted copy (Batcher network, constant indices) is parked in LDS as [rank][lane] -- per-lane dynamic ranks then
//     cost one conflict-free ds_read (bank = lane) instead of a scratch access;
//   * survivors of every retain pass are an interval [lo, hi) of the sorted order (z is monotone in v), found by
//     walking in from both ends with the reference's own f32 z arithmetic;
//   * median = S[lo + len/2]; MAD = the len/2-th smallest of the two sorted deviation runs left / right of the median,
//     by binary search on the split (<= 6 steps of two LDS reads);
//   * the kept set is {f : S[lo] <= u_f <= S[hi-1]} (ties share their z, so they are kept or dropped together).
// One wave per block (16 KB LDS at n = 64, ten waves per CU), persistent blocks striding over 64-pixel chunks, so the
// per-frame rejection counters stay in a register (lane f owns frame f) until the block retires.
#include "ab_common.hpp"
#include "block_sort.hpp"
#include "wave_sort.hpp"

#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstring>
#include <strings.h>

#define AB_CE(a, b)                         \
    {                                       \
        T lo_ = fminf(v[a], v[b]);          \
        T hi_ = fmaxf(v[a], v[b]);          \
        v[a] = lo_;                         \
        v[b] = hi_;                         \
    }
// four unsorted samples through the three-input operations: 7 half-rate instructions instead of 10 (see stack_sigma_clip.hip;
// NaN samples have become +inf before the network runs)
#define AB_SORT4(a, b, c, d)                                                         \
    {                                                                                \
        const T x0_ = v[a], x1_ = v[b], x2_ = v[c], x3_ = v[d];                      \
        const T s0_ = fminf(fminf(x0_, x1_), x2_), s1_ = __builtin_amdgcn_fmed3f(x0_, x1_, x2_), \
                s2_ = fmaxf(fmaxf(x0_, x1_), x2_);                                   \
        v[a] = __builtin_amdgcn_fmed3f(-__builtin_inff(), s0_, x3_); /* = min: no canonicalising v_max of x3 first */ \
        v[b] = __builtin_amdgcn_fmed3f(s0_, s1_, x3_);                               \
        v[c] = __builtin_amdgcn_fmed3f(s1_, s2_, x3_);                               \
        v[d] = __builtin_amdgcn_fmed3f(__builtin_inff(), s2_, x3_);  /* = max */          \
    }
#define AB_STACK_CE_XOR 1
#include "sort_ops.hpp"
#include "sortnet_gen.hpp"

namespace {

constexpr int kMaxFrames = 128;  // lane-per-pixel kernels; 65 .. 128: two lane-resident registers per table, no prefetch
constexpr int kWave = 64;
constexpr int kWavesPerBlock = 4;  // the waves of a block take consecutive chunks: 1 KB contiguous per frame per block step
constexpr double kMadToSigma = 1.4826;  // types/constants.rs:7
constexpr int kSumBlock = 256;

struct Masters {  // calibrate_light's three optional planes; a null pointer = absent or of the wrong length (:87-89)
    const float *bias, *dark, *flat;
};

struct BatchArgs {
    const float *p[kMaxFrames];
    float scale[kMaxFrames];  // normalize_frames' inv_mean (:313-314); 1 where the mean is <= 0 or NaN (x * 1.0f == x bit for bit)
    Masters m;
    int n;
    uint32_t npix;
    float sigma_low, sigma_high;
    int max_iter;
    float *out;
    uint32_t *rej;  // [waves][64]
    int stage;           // developer ablation (AB_BATCH_STAGE): 1 no sort, 2 no LDS copy, 3 no epilogue loop
    uint32_t per_block;  // 0: block b takes chunks b, b + grid, ...; else chunks [b * per_block, (b + 1) * per_block)
    // the retain test without its division (retain_thresholds below): z = RN32(d / sigma) < sigma_high  <=>  d < mid_hi * sigma
    int retain_fast;
    int guess_step;  // 0: every MAD search starts from the whole range (AB_BATCH_GUESS: the bracket's half width, default 1)
    double mid_lo, mid_hi;
};

// `(x - median) / sigma` is compared with two constants (:359-360), so the IEEE division can be replaced by a comparison of d =
// RN32(x - median) with a threshold: RN32(q) < h  <=>  q < m, m the midpoint of h and its predecessor, and q < m  <=>  d < m sigma
// for sigma > 0.  m is an odd 25-bit integer times a power of two, sigma has 24 bits, d 24: m sigma and d are exact doubles and the
// comparison is exact -- and a tie q = m cannot happen at all (m sigma has at least 25 significant bits, d has 24), so the rounding
// rule at the midpoint never matters.  Mirror image for the lower bound.  Only for bounds in the normal range (the midpoint must
// not touch the subnormals) and a finite sigma; everything else keeps the division (tests/test_retain_threshold_math.py restates
// the claim in numpy on values one ulp either side of every boundary).
bool retain_thresholds(BatchArgs *a) {
    const float nsl = -a->sigma_low, sh = a->sigma_high;
    auto usable = [](float v) { return std::isfinite(v) && std::fabs(v) >= 1e-30f && std::fabs(v) <= 1e30f; };
    a->retain_fast = 0;
    a->mid_lo = a->mid_hi = 0.0;
    if (!usable(nsl) || !usable(sh) || ab_dev_env("AB_BATCH_DIVIDE")) return false;
    a->mid_hi = ((double)std::nextafterf(sh, -INFINITY) + (double)sh) * 0.5;
    a->mid_lo = ((double)nsl + (double)std::nextafterf(nsl, INFINITY)) * 0.5;
    a->retain_fast = 1;
    return true;
}

// calibrate_light's per-pixel chain (:93-113) in its f32 operation order.  (Tried: sharing the Newton-refined reciprocal
// of the flat across a pixel's n frames and finishing each quotient with the four fma of the hardware-assisted expansion --
// bit-exact on 50 M random operand pairs, but the per-sample range check and wave vote it needs made both kernels slower.)
struct CalPx {
    float b, d, f;
    bool has_b, has_d, div;
};
__device__ __forceinline__ CalPx cal_load(const Masters &m, uint32_t i) {
    CalPx c;
    c.has_b = m.bias != nullptr;
    c.has_d = m.dark != nullptr;
    c.b = c.has_b ? m.bias[i] : 0.0f;
    c.d = c.has_d ? m.dark[i] : 0.0f;
    c.f = m.flat ? m.flat[i] : 1.0f;
    c.div = m.flat != nullptr && __builtin_isfinite(c.f) && fabsf(c.f) > 1e-4f;
    return c;
}
__device__ __forceinline__ float cal_apply(float v, const CalPx &c) {
    if (c.has_b) v -= c.b;
    if (c.has_d) v -= c.d;
    if (c.div) v = v / c.f;
    return v < 0.0f ? 0.0f : v;  // NaN stays NaN
}

// The plane pointers and frame scales live in VGPRs, one frame per lane (64 uniform pointers would overflow the SGPR file
// and get spilled lane by lane); each load pulls its base out with two v_readlane into a buffer descriptor and issues
// `buffer_load_dword v, voffset, s[rsrc], 0 offen` -- no 64-bit address arithmetic, no flat-address aperture check.
template <int NP, int R>
__device__ __forceinline__ void gather(float (&u)[NP], const uint32_t (&plo_in)[R], const uint32_t (&phi_in)[R], uint32_t gi, uint32_t plane_bytes) {
    static_assert(NP <= 64 * R, "one pointer register per 64 frames");
    const uint32_t off = gi * 4u;  // < 2^32: planes hold fewer than 2^30 pixels
    // opaque to loop-invariant code motion: hoisted out of the chunk loop, the 64 bases would be spilled right back
    uint32_t plo[R], phi[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        plo[r] = plo_in[r];
        phi[r] = phi_in[r];
        asm volatile("" : "+v"(plo[r]), "+v"(phi[r]));
    }
#pragma unroll
    for (int f = 0; f < NP; ++f) {  // slots past n read the host's pad plane (see scms_kernel)
        const uint64_t base = ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)phi[f >> 6], f & 63) << 32) |
                              (uint32_t)__builtin_amdgcn_readlane((int)plo[f >> 6], f & 63);
        const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)base, 0, (int)plane_bytes, 0x00020000);
        u[f] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, (int)off, 0, 0));
    }
}

// The same gather with the plane pointers through the SCALAR cache (<= 64 slots): the table is the first member of the kernel
// argument block, `s_load_dwordx16` fetches eight pointers at a time and every sample is `global_load_dword v, voffset,
// s[base:base+1]` -- no v_readlane pair, no descriptor, no s_nop per sample (that was 15 % of the chunk loop's instructions).
// The kernarg pointer goes through an opaque copy on every call: hoisted out of the chunk loop, the 64 bases would overflow
// the SGPR file and be spilled lane by lane, which is how the VGPR-resident table above came about.  gi < npix (clamped by
// the callers), so nothing is out of range without the descriptor's bounds check.
template <int NP>
__device__ __forceinline__ void gather_scalar(float (&u)[NP], uint32_t gi) {
    static_assert(NP <= 64, "the 128-slot kernel keeps its table in two VGPRs");
    const uint32_t off = gi * 4u;
    // (explicit address spaces: behind the opaque copy the compiler no longer knows that this is the kernarg segment -- constant,
    // uniform -- nor that what it holds are global pointers, and would fall back to 64-bit flat loads for both)
    typedef const uint64_t __attribute__((address_space(4))) *KernargTable;
    typedef const float __attribute__((address_space(1))) *GlobalF32;
    KernargTable kp = (KernargTable)__builtin_amdgcn_kernarg_segment_ptr();  // BatchArgs::p is at offset 0
    asm volatile("" : "+s"(kp));
#pragma unroll
    for (int f = 0; f < NP; ++f) u[f] = *(GlobalF32)(kp[f] + off);
}

// slot f of a ragged stack (n < NP) is a pad.  The frame count goes through an opaque scalar copy at every use: left to
// itself LLVM evaluates all NP `f >= n` up front as 64-bit lane masks, keeps them for the whole chunk loop and spills the
// SGPR file (157 spills, one wave per SIMD, 2x the time of the full kernel).
__device__ __forceinline__ bool pad_slot(int n, int f) {
    asm volatile("" : "+s"(n));
    return f >= n;
}

// FULL: n == NP.  Otherwise the slots past n read a plane of FLT_MAX (set up by the host): finite, so the fast path needs
// no per-slot predicate -- they sort above every sample that can be inside the window [0, n), which is all the clipping
// loop ever looks at.  (When the wave meets a non-finite sample, the slow path turns them into +inf like NaN stand-ins.)
template <int NP, bool CAL, bool FULL>
__global__ __launch_bounds__(kWave *kWavesPerBlock) void scms_kernel(const BatchArgs a) {
    extern __shared__ float S_[];  // [NP][64] sorted samples of this wave's 64 pixels
    const int lane = threadIdx.x & 63;
    const uint32_t wid = blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6), nwaves = gridDim.x * kWavesPerBlock;  // global wave id
    constexpr int R = (NP + 63) / 64;  // lane-resident tables: lane l of register r belongs to frame 64 r + l
    constexpr bool PF = NP <= 64;      // 128 slots: u[], v[] alone fill the register file, the next chunk is not prefetched
    uint32_t mycount[R];  // rejected samples of this lane's frames, over every chunk of this block (< 2^30 per block)
    uint32_t plo[R], phi[R];
    float myscale[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const uint64_t myptr = (uint64_t)a.p[64 * r + lane];
        plo[r] = (uint32_t)myptr;
        phi[r] = (uint32_t)(myptr >> 32);
        myscale[r] = a.scale[64 * r + lane];
        mycount[r] = 0;
    }
    const uint32_t nchunks = (a.npix + kWave - 1) / kWave;
#define S(i) S_[(i) * (kWave * kWavesPerBlock) + threadIdx.x]

    // software pipeline: the next chunk's samples are in flight while this one is sorted and clipped (a wave computes
    // for ~5 us per chunk and only two waves fit a SIMD, so nothing else would hide the HBM latency)
    float nxt[PF ? NP : 1];
    CalPx cnxt{};
    const uint32_t first = a.per_block ? wid * a.per_block : wid, step = a.per_block ? 1u : nwaves;
    const uint32_t end = a.per_block ? min(first + a.per_block, nchunks) : nchunks;
    if constexpr (PF) {
        if (first < end) {
            const uint32_t g0 = first * kWave + lane, gi0 = g0 < a.npix ? g0 : a.npix - 1;
            if constexpr (NP <= 64) gather_scalar<NP>(nxt, gi0); else gather<NP>(nxt, plo, phi, gi0, a.npix * 4u);
            if constexpr (CAL) cnxt = cal_load(a.m, gi0);
        }
    }
    for (uint32_t chunk = first; chunk < end; chunk += step) {
        const uint32_t g = chunk * kWave + lane;
        const bool valid = g < a.npix;
#define PAD(f) (!FULL && pad_slot(a.n, (f)))

        // ---- gather, frame order (:344-348) ----
        float u[NP];
        CalPx c{};
        if constexpr (PF) {
#pragma unroll
            for (int f = 0; f < NP; ++f) u[f] = nxt[f];
            c = cnxt;
            if (chunk + step < end) {
                const uint32_t g1 = (chunk + step) * kWave + lane, gi1 = g1 < a.npix ? g1 : a.npix - 1;
                if constexpr (NP <= 64) gather_scalar<NP>(nxt, gi1); else gather<NP>(nxt, plo, phi, gi1, a.npix * 4u);
                if constexpr (CAL) cnxt = cal_load(a.m, gi1);
            }
        } else {
            const uint32_t gi = valid ? g : a.npix - 1;
            if constexpr (NP <= 64) gather_scalar<NP>(u, gi); else gather<NP>(u, plo, phi, gi, a.npix * 4u);
            if constexpr (CAL) c = cal_load(a.m, gi);
        }
        if constexpr (CAL) {
#pragma unroll
            for (int f = 0; f < NP; ++f) {
                const float cv = cal_apply(u[f], c) * __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, myscale[f >> 6]), f & 63));
                u[f] = PAD(f) ? __FLT_MAX__ : cv;  // (a pad must not be calibrated: a negative flat would clamp it to 0)
            }
        }

        // ---- sorted copy: NaN sorts last in f32_cmp (math/median.rs:4-13); it travels as +inf and is told apart by count ----
        float v[NP];
        int cnan = 0;
        float nf = 0.0f;  // fma(x, 0, nf) stays 0 for finite x and turns NaN for inf / NaN
#pragma unroll
        for (int f = 0; f < NP; ++f) {
            v[f] = u[f];
            nf = __builtin_fmaf(u[f], 0.0f, nf);
        }
        if (__any(nf != nf)) {  // rare: some lane of this wave met a non-finite sample
#pragma unroll
            for (int f = 0; f < NP; ++f) {
                const bool isn = u[f] != u[f];
                v[f] = (isn || PAD(f)) ? __builtin_inff() : u[f];  // pads go above real +inf samples again
                cnan += isn ? 1 : 0;
            }
        }
        if (a.stage != 1) {
            if constexpr (NP >= 8 && NP <= 128)
                SortNet<NP>::sort_fused(v);  // the rewrite over min3 / med3 / max3 (tools/gen_sortnet.py): 789 instructions for 64 samples, not 1038
            else
                SortNet<NP>::sort(v);
        }
        if (a.stage != 2) {
#pragma unroll
            for (int i = 0; i < NP; ++i) S(i) = v[i];
        }

        // ---- the clipping loop (:350-367) on the window [lo, hi) of the sorted order ----
        int lo = 0, hi = a.n;
        int prev_i0 = 0, prev_k = 0;  // the last iteration's MAD split (see the search below)
        for (int it = 0; it < a.max_iter; ++it) {
            const int len = hi - lo;
            if (len < 3) break;
            const int k = len >> 1, c = lo + k;
            // one LDS round trip: the median and the two outermost samples of either end (all the retain pass reads for
            // a pixel that loses at most two samples per end; len >= 3 keeps the four indices inside the window)
            const float med = S(c), e0 = S(lo), e1 = S(lo + 1), f0 = S(hi - 1), f1 = S(hi - 2);
            // a NaN or infinite median makes every z NaN: the retain pass empties the pixel
            if ((cnan > 0 && c >= a.n - cnan) || !__builtin_isfinite(med)) {
                lo = hi = 0;
                break;
            }
            // MAD = the k-th smallest of the deviation runs A[j] = med - S[c-j] (j < nA), B[j] = S[c+1+j] - med (j < nB):
            // the number i of A's among the k + 1 smallest is the first i with A[i] >= B[k-i]; 4-ary search, three
            // independent probes (six reads) per round trip
            const int nA = c - lo + 1, nB = hi - c - 1;
            int i0 = max(0, k + 1 - nB), i1 = min(k + 1, nA);
            // from the second iteration on the search starts at the previous split, moved by half of what the window lost: a
            // retain pass shaves a sample or two, the split moves by one or two, and a bracket of three probes around the guess
            // settles most lanes in ONE round trip instead of three (lanes it does not settle go on with the 4-ary rounds)
            bool guessed = it > 0 && a.guess_step > 0;
            while (i0 < i1) {
                const int w = i1 - i0;
                int m1 = i0 + (w >> 2), m2 = i0 + (w >> 1), m3 = i0 + ((3 * w) >> 2);  // all within [i0, i1 - 1]
                if (guessed) {
                    const int g = prev_i0 + ((k - prev_k) >> 1);
                    m2 = min(max(g, i0), i1 - 1);
                    m1 = max(m2 - a.guess_step, i0);
                    m3 = min(m2 + a.guess_step, i1 - 1);
                    guessed = false;
                }
                const float a1 = S(c - m1), b1 = S(c + k + 1 - m1), a2 = S(c - m2), b2 = S(c + k + 1 - m2), a3 = S(c - m3), b3 = S(c + k + 1 - m3);
                const bool p1 = (med - a1) < (b1 - med), p2 = (med - a2) < (b2 - med), p3 = (med - a3) < (b3 - med);
                const int n0 = !p1 ? i0 : (!p2 ? m1 + 1 : (!p3 ? m2 + 1 : m3 + 1));
                const int n1 = !p1 ? m1 : (!p2 ? m2 : (!p3 ? m3 : i1));
                i0 = n0;
                i1 = n1;
            }
            prev_i0 = i0;
            prev_k = k;
            const int j0 = k + 1 - i0;
            const float ma = i0 > 0 ? med - S(c - (i0 - 1)) : -__builtin_inff();
            const float mb = j0 > 0 ? S(c + j0) - med : -__builtin_inff();
            float mad = fmaxf(ma, mb);
            if (cnan > 0 && k >= len - cnan) mad = __builtin_nanf("");  // the rank falls among the NaN deviations
            const float sigma = (float)((double)mad * kMadToSigma);
            if (sigma < 1e-10f) break;
            const float nsl = -a.sigma_low, sh = a.sigma_high;
            int nlo = lo, nhi = hi;
            auto shave = [&](auto kept) {
                if (!kept(e0)) {
                    nlo = lo + 1;
                    if (!kept(e1)) {
                        nlo = lo + 2;
                        while (nlo < nhi && !kept(S(nlo))) ++nlo;
                    }
                }
                if (nhi > nlo && !kept(f0)) {
                    nhi = hi - 1;
                    if (nhi > nlo && !kept(f1)) {
                        nhi = hi - 2;
                        while (nhi > nlo && !kept(S(nhi - 1))) --nhi;
                    }
                }
            };
            if (a.retain_fast && !__any(!__builtin_isfinite(sigma))) {  // wave-uniform; see retain_thresholds
                const double sd = (double)sigma, tlo = a.mid_lo * sd, thi = a.mid_hi * sd;
                shave([&](float x) {
                    const double d = (double)(x - med);
                    return d > tlo && d < thi;
                });
            } else {
                shave([&](float x) {
                    const float z = (x - med) / sigma;
                    return z > nsl && z < sh;
                });
            }
            cnan = 0;  // a retain pass never keeps a NaN
            if (nlo == lo && nhi == hi) break;
            lo = nlo;
            hi = nhi;
        }

        // ---- frame-order f32 mean of the survivors (:369), per-frame rejection counts (:361) ----
        const int len = hi - lo;
        const bool all = lo == 0 && hi == a.n;
        const float lov = len > 0 ? S(lo) : __builtin_inff(), hiv = len > 0 ? S(hi - 1) : -__builtin_inff();
        // u_f survives <=> clamp(u_f, S[lo], S[hi-1]) == u_f (NaN never equals its clamp; an untouched pixel keeps
        // everything, an emptied one nothing).  One v_med3 + one v_cmp per frame; the compare IS the wave's reject mask,
        // so the per-frame count is a scalar popcount.
        const unsigned long long live = __builtin_amdgcn_ballot_w64(valid & !all);   // lanes whose compare decides
        const unsigned long long force = __builtin_amdgcn_ballot_w64(valid & !all & (len == 0));
        float sum = 0.0f;
        float lo_s = lov;
        uint32_t cnts[R];  // lane f & 63 of register f >> 6 <- this chunk's rejections of frame f (one v_writelane each)
#pragma unroll
        for (int r = 0; r < R; ++r) cnts[r] = 0;
        if (a.stage == 3) sum = u[0] + v[NP - 1];
        else
#pragma unroll
        for (int f = 0; f < NP; ++f) {
            // a sequencing point per frame: without it the scheduler evaluates all 64 masks first and spills them
            asm volatile("" : "+v"(lo_s), "+v"(sum));
            const float clamped = __builtin_amdgcn_fmed3f(u[f], lo_s, hiv);
            unsigned long long rej = (__builtin_amdgcn_ballot_w64(clamped != u[f]) & live) | force;
            if constexpr (!FULL) rej = PAD(f) ? ~0ull : rej;  // a pad never enters the sum (its count lands on a lane >= n)
            const bool mine = __builtin_amdgcn_inverse_ballot_w64(rej);  // the scalar mask back as a v_cndmask selector
            sum += mine ? 0.0f : u[f];
            int cnt = __popcll(rej);
            asm volatile("" : "+s"(cnt));  // keep it a register operand even where the compiler can fold it to a constant
            asm volatile("v_writelane_b32 %0, %1, %2" : "+v"(cnts[f >> 6]) : "s"(cnt), "n"(f & 63));
        }
#pragma unroll
        for (int r = 0; r < R; ++r) mycount[r] += cnts[r];
        if (valid) a.out[g] = len == 0 ? 0.0f : sum / (float)len;
    }
#undef S
#undef PAD
#pragma unroll
    for (int r = 0; r < R; ++r) a.rej[(size_t)wid * kMaxFrames + 64 * r + lane] = mycount[r];
}

// [blocks][64] -> [64]: one block per frame
__global__ __launch_bounds__(256) void rej_reduce_kernel(const uint32_t *__restrict__ part, int blocks, unsigned long long *__restrict__ out) {
    const int f = blockIdx.x;
    unsigned long long s = 0;
    for (int b = threadIdx.x; b < blocks; b += 256) s += part[(size_t)b * kMaxFrames + f];
    __shared__ unsigned long long red[256];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int k = 128; k > 0; k >>= 1) {
        if ((int)threadIdx.x < k) red[threadIdx.x] += red[threadIdx.x + k];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[f] = red[0];
}

// per-frame f64 sums of the calibrated samples: part[block][f].  blockIdx.y selects a group of FR <= 32 frames (64
// frames deep the accumulators alone would take 128 VGPRs and leave no room to prefetch); the next pixels' samples are
// in flight while the current ones are converted and added.
template <int FR>
__global__ __launch_bounds__(kSumBlock) void cal_means_kernel(const BatchArgs a, double *__restrict__ part) {
    double acc[FR];
#pragma unroll
    for (int f = 0; f < FR; ++f) acc[f] = 0.0;
    const int first = blockIdx.y * FR;
    const uint32_t stride = gridDim.x * kSumBlock, bytes = a.npix * 4u;
    const uint64_t myptr = (uint64_t)a.p[first + ((threadIdx.x & 63) % FR)];
    const uint32_t plo[1] = {(uint32_t)myptr}, phi[1] = {(uint32_t)(myptr >> 32)};
    // wave-uniform trip count: gather() moves the lane-resident pointers through a register copy, which only the ACTIVE
    // lanes take part in -- a lane that had left the loop would hand v_readlane a stale base
    float nxt[FR];
    CalPx cnxt{};
    uint32_t g0 = blockIdx.x * kSumBlock;
    if (g0 < a.npix) {
        const uint32_t g = g0 + threadIdx.x, gi = g < a.npix ? g : a.npix - 1;
        gather<FR>(nxt, plo, phi, gi, bytes);
        cnxt = cal_load(a.m, gi);
    }
    for (; g0 < a.npix; g0 += stride) {
        const bool valid = g0 + threadIdx.x < a.npix;
        float u[FR];
#pragma unroll
        for (int f = 0; f < FR; ++f) u[f] = nxt[f];
        const CalPx c = cnxt;
        if (g0 + stride < a.npix) {
            const uint32_t g = g0 + stride + threadIdx.x, gi = g < a.npix ? g : a.npix - 1;
            gather<FR>(nxt, plo, phi, gi, bytes);
            cnxt = cal_load(a.m, gi);
        }
#pragma unroll
        for (int f = 0; f < FR; ++f) acc[f] += valid ? (double)cal_apply(u[f], c) : 0.0;
    }
    __shared__ double red[kSumBlock / kWave][FR];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int f = 0; f < FR; ++f) {
        double s = acc[f];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
        if (lane == 0) red[wave][f] = s;
    }
    __syncthreads();
    if (threadIdx.x < FR) {
        double s = 0.0;
        for (int w = 0; w < kSumBlock / kWave; ++w) s += red[w][threadIdx.x];
        part[(size_t)blockIdx.x * kMaxFrames + first + threadIdx.x] = s;
    }
}

// mode 0: sum of v; mode 1: sum of (v - center)^2   (run_batch_pipeline's mean / variance, :173-179)
__global__ __launch_bounds__(kSumBlock) void sum_f64_kernel(const float *__restrict__ data, int64_t n, int mode, double center,
                                                            double *__restrict__ part) {
    double s = 0.0;
    const int64_t stride = (int64_t)gridDim.x * kSumBlock;
    for (int64_t i = (int64_t)blockIdx.x * kSumBlock + threadIdx.x; i < n; i += stride) {
        const double v = (double)data[i];
        s += mode ? (v - center) * (v - center) : v;
    }
    __shared__ double red[kSumBlock];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int k = kSumBlock / 2; k > 0; k >>= 1) {
        if ((int)threadIdx.x < k) red[threadIdx.x] += red[threadIdx.x + k];
        __syncthreads();
    }
    if (threadIdx.x == 0) part[blockIdx.x] = red[0];
}

__global__ __launch_bounds__(256) void calibrate_kernel(const float *__restrict__ light, Masters m, uint32_t npix, float *__restrict__ out) {
    const uint32_t stride = gridDim.x * 256;
    for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < npix; i += stride) out[i] = cal_apply(light[i], cal_load(m, i));
}

__global__ __launch_bounds__(256) void scale_kernel(const float *__restrict__ in, int64_t n, float k, float *__restrict__ out) {
    const int64_t stride = (int64_t)gridDim.x * 256;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) out[i] = in[i] * k;
}

// normalize_channel's scan (:293-299): `v < min` / `v > max` skip NaN and see the infinities
__global__ __launch_bounds__(256) void minmax_kernel(const float *__restrict__ src, int rows, int cols, int64_t ld, float2 *__restrict__ part) {
    float mn = __builtin_inff(), mx = -__builtin_inff();
    const int64_t n = (int64_t)rows * cols, stride = (int64_t)gridDim.x * 256;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
        const int64_t y = i / cols, x = i - y * cols;
        const float v = src[y * ld + x];
        mn = v < mn ? v : mn;
        mx = v > mx ? v : mx;
    }
    __shared__ float smn[256], smx[256];
    smn[threadIdx.x] = mn;
    smx[threadIdx.x] = mx;
    __syncthreads();
    for (int k = 128; k > 0; k >>= 1) {
        if ((int)threadIdx.x < k) {
            smn[threadIdx.x] = fminf(smn[threadIdx.x], smn[threadIdx.x + k]);
            smx[threadIdx.x] = fmaxf(smx[threadIdx.x], smx[threadIdx.x + k]);
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) part[blockIdx.x] = make_float2(smn[0], smx[0]);
}

struct ChanNorm {  // normalize_channel (:301-306) folded into the compose kernel
    const float *p;
    int64_t ld;
    float mn, inv_range;
    int zero;  // range < 1e-10: the channel becomes zeros
};
__device__ __forceinline__ float norm_px(const ChanNorm &c, int64_t y, int64_t x) {
    if (c.zero) return 0.0f;
    const float t = (c.p[y * c.ld + x] - c.mn) * c.inv_range;
    return t < 0.0f ? 0.0f : (t > 1.0f ? 1.0f : t);
}
struct ComposeArgs {
    ChanNorm r, g, b, l;
    int with_l;
    int rows, cols;
    float *out;
};
__global__ __launch_bounds__(256) void compose_masters_kernel(const ComposeArgs a) {  // :214-266, apply_luminance :269-289
    const int64_t n = (int64_t)a.rows * a.cols, stride = (int64_t)gridDim.x * 256;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
        const int64_t y = i / a.cols, x = i - y * a.cols;
        float r = norm_px(a.r, y, x), g = norm_px(a.g, y, x), b = norm_px(a.b, y, x);
        if (a.with_l) {
            const float l = norm_px(a.l, y, x);
            const float rgb_lum = 0.2126f * r + 0.7152f * g + 0.0722f * b;
            const float scale = rgb_lum > 1e-10f ? l / rgb_lum : 1.0f;
            r = r * scale;
            g = g * scale;
            b = b * scale;
            r = r < 0.0f ? 0.0f : (r > 1.0f ? 1.0f : r);
            g = g < 0.0f ? 0.0f : (g > 1.0f ? 1.0f : g);
            b = b < 0.0f ? 0.0f : (b > 1.0f ? 1.0f : b);
        }
        a.out[i * 3] = r;
        a.out[i * 3 + 1] = g;
        a.out[i * 3 + 2] = b;
    }
}

// ---- more than 64 frames (65 .. 512): one WAVE per pixel --------------------------------------------------------------
// Lane l holds frames l, l + 64, ... (K = 2, 4 or 8 registers, in frame order by construction), a sorted copy comes from
// wave_sort, ranks are v_readlane lookups, and every iteration re-sorts the window's deviations for its MAD.  Each lane
// counts the rejections of ITS frames in registers across all the pixels of the wave.  Same definition as scms_kernel,
// bit for bit; ~40x slower per sample.
struct WideBatchArgs {
    const float *const *p;  // n plane pointers (device array)
    const float *scale;     // n inv_mean factors (1 where a frame is not scaled)
    unsigned long long *rej;  // n per-frame counters (zeroed by the caller)
    Masters m;
    int n;
    uint32_t npix;
    float sigma_low, sigma_high;
    int max_iter;
    float *out;
};

template <int K, bool CAL>
__global__ __launch_bounds__(256) void scms_wide_kernel(const WideBatchArgs a) {
    const int lane = threadIdx.x & 63;
    const uint32_t wave_id = blockIdx.x * 4u + (threadIdx.x >> 6), nwaves = gridDim.x * 4u;
    uint32_t myrej[K];
    float myscale[K];
#pragma unroll
    for (int k = 0; k < K; ++k) {
        myrej[k] = 0;
        myscale[k] = (CAL && lane + 64 * k < a.n) ? a.scale[lane + 64 * k] : 1.0f;
    }
    for (uint32_t g = wave_id; g < a.npix; g += nwaves) {  // wave-uniform pixel
        float u[K], s[K];
        int nan_here = 0;
        CalPx c{};
        if constexpr (CAL) c = cal_load(a.m, g);
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const int f = lane + 64 * k;
            const bool present = f < a.n;
            float v = __builtin_inff();
            if (present) {
                v = a.p[f][g];
                if constexpr (CAL) v = cal_apply(v, c) * myscale[k];
            }
            u[k] = v;
            const bool isn = present && v != v;  // NaN sorts last in f32_cmp: it travels as +inf and is told apart by count
            s[k] = isn ? __builtin_inff() : v;
            nan_here += isn ? 1 : 0;
        }
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) nan_here += __shfl_xor(nan_here, off, 64);
        int cnan = __builtin_amdgcn_readfirstlane(nan_here);
        wave_sort<K>(s, lane);  // ranks [0, n) hold the n samples (absent slots are +inf pads above them)

        int lo = 0, hi = a.n;
        for (int it = 0; it < a.max_iter; ++it) {  // calibration_pipeline.rs:350-367; every condition is wave-uniform
            const int len = hi - lo;
            if (len < 3) break;
            const int k2 = len >> 1, cpos = lo + k2;
            const float med = elem<K>(s, cpos);
            if ((cnan > 0 && cpos >= a.n - cnan) || !__builtin_isfinite(med)) {  // every z is NaN: the pass empties the pixel
                lo = hi = 0;
                break;
            }
            float d[K];
#pragma unroll
            for (int k = 0; k < K; ++k) {
                const int e = lane * K + k;
                d[k] = (e >= lo && e < hi) ? fabsf(s[k] - med) : __builtin_inff();
            }
            wave_sort<K>(d, lane);
            float mad = elem<K>(d, k2);
            if (cnan > 0 && k2 >= len - cnan) mad = __builtin_nanf("");  // the rank falls among the NaN deviations
            const float sigma = (float)((double)mad * kMadToSigma);
            if (sigma < 1e-10f) break;
            const float nsl = -a.sigma_low, sh = a.sigma_high;
            int drop_lo = 0, drop_hi = 0;
#pragma unroll
            for (int k = 0; k < K; ++k) {
                const int e = lane * K + k;
                const bool in = e >= lo && e < hi;
                const float z = (s[k] - med) / sigma;
                const bool keep = z > nsl && z < sh;
                const bool low_side = s[k] < med;
                drop_lo += (int)__builtin_popcountll(__ballot(in && !keep && low_side));
                drop_hi += (int)__builtin_popcountll(__ballot(in && !keep && !low_side));
            }
            cnan = 0;  // a retain pass never keeps a NaN
            if (drop_lo == 0 && drop_hi == 0) break;
            lo += drop_lo;
            hi -= drop_hi;
            if (hi < lo) hi = lo;
        }

        const int len = hi - lo;
        const bool all = lo == 0 && hi == a.n;
        const float lov = len > 0 ? elem<K>(s, lo) : __builtin_inff(), hiv = len > 0 ? elem<K>(s, hi - 1) : -__builtin_inff();
        float w[K];
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const bool present = lane + 64 * k < a.n;
            const bool keep = present && (all || (u[k] >= lov && u[k] <= hiv));
            w[k] = keep ? u[k] : 0.0f;
            myrej[k] += (present && !keep) ? 1u : 0u;
        }
        float sum = 0.0f;  // frame order: frame f lives in lane f % 64, register f / 64
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const int cnt = min(64, a.n - 64 * k);  // uniform
            for (int l = 0; l < cnt; ++l) sum += __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, w[k]), l));
        }
        if (lane == 0) a.out[g] = len == 0 ? 0.0f : sum / (float)len;
    }
#pragma unroll
    for (int k = 0; k < K; ++k)
        if (lane + 64 * k < a.n && myrej[k]) atomicAdd(&a.rej[lane + 64 * k], (unsigned long long)myrej[k]);
}

// ---- more than 2048 frames (any count): one WORKGROUP per pixel, samples in global scratch --------------------------------
// The same definition once more (calibration_pipeline.rs:317-378) with a pixel's samples in a per-workgroup segment of three
// arrays: U = the calibrated samples in frame order, S = sorted, D = the window's deviations sorted (block_sort.hpp).  Frames
// past n are +inf pads above the order, NaN samples travel as +inf and are told apart by count, exactly as in scms_wide_kernel.
// A fallback for frame counts no real set of megapixel frames reaches; bit-identical to the narrower kernels where they overlap.
struct DeepBatchArgs {
    WideBatchArgs w;
    int np2;
    float *scratch;  // gridDim.x segments of 3 * np2 floats
};

template <bool CAL>
__global__ __launch_bounds__(256) void scms_deep_kernel(const DeepBatchArgs b) {
    const WideBatchArgs &a = b.w;
    __shared__ int sh_cnt[2];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    float *U = b.scratch + (size_t)blockIdx.x * 3 * b.np2, *S = U + b.np2, *D = S + b.np2;
    for (uint32_t g = blockIdx.x; g < a.npix; g += gridDim.x) {
        if (tid < 2) sh_cnt[tid] = 0;
        __syncthreads();
        CalPx c{};
        if constexpr (CAL) c = cal_load(a.m, g);
        int nan_here = 0;
        for (int f = tid; f < b.np2; f += 256) {
            float v = __builtin_inff();
            const bool present = f < a.n;
            if (present) {
                v = a.p[f][g];
                if constexpr (CAL) v = cal_apply(v, c) * a.scale[f];
            }
            U[f] = v;
            const bool isn = present && v != v;
            S[f] = isn ? __builtin_inff() : v;
            nan_here += isn ? 1 : 0;
        }
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) nan_here += __shfl_xor(nan_here, off, 64);
        if (lane == 0 && nan_here) atomicAdd(&sh_cnt[0], nan_here);
        __syncthreads();
        int cnan = sh_cnt[0];
        block_bitonic_sort(S, b.np2);

        int lo = 0, hi = a.n;
        for (int it = 0; it < a.max_iter; ++it) {  // calibration_pipeline.rs:350-367; every condition is workgroup-uniform
            const int len = hi - lo;
            if (len < 3) break;
            const int k2 = len >> 1, cpos = lo + k2;
            const float med = S[cpos];
            if ((cnan > 0 && cpos >= a.n - cnan) || !__builtin_isfinite(med)) {  // every z is NaN: the pass empties the pixel
                lo = hi = 0;
                break;
            }
            for (int e = tid; e < b.np2; e += 256) D[e] = (e >= lo && e < hi) ? fabsf(S[e] - med) : __builtin_inff();
            block_bitonic_sort(D, b.np2);
            float mad = D[k2];
            if (cnan > 0 && k2 >= len - cnan) mad = __builtin_nanf("");  // the rank falls among the NaN deviations
            const float sigma = (float)((double)mad * kMadToSigma);
            if (sigma < 1e-10f) break;
            const float nsl = -a.sigma_low, sh = a.sigma_high;
            __syncthreads();
            if (tid < 2) sh_cnt[tid] = 0;
            __syncthreads();
            int drop_lo = 0, drop_hi = 0;
            for (int e = lo + tid; e < hi; e += 256) {
                const float sv = S[e];
                const float z = (sv - med) / sigma;
                const bool keep = z > nsl && z < sh;
                const bool low_side = sv < med;
                drop_lo += (!keep && low_side) ? 1 : 0;
                drop_hi += (!keep && !low_side) ? 1 : 0;
            }
#pragma unroll
            for (int off = 32; off >= 1; off >>= 1) {
                drop_lo += __shfl_xor(drop_lo, off, 64);
                drop_hi += __shfl_xor(drop_hi, off, 64);
            }
            if (lane == 0) {
                if (drop_lo) atomicAdd(&sh_cnt[0], drop_lo);
                if (drop_hi) atomicAdd(&sh_cnt[1], drop_hi);
            }
            __syncthreads();
            drop_lo = sh_cnt[0];
            drop_hi = sh_cnt[1];
            cnan = 0;  // a retain pass never keeps a NaN
            if (drop_lo == 0 && drop_hi == 0) break;
            lo += drop_lo;
            hi -= drop_hi;
            if (hi < lo) hi = lo;
        }

        const int len = hi - lo;
        const bool all = lo == 0 && hi == a.n;
        const float lov = len > 0 ? S[lo] : __builtin_inff(), hiv = len > 0 ? S[hi - 1] : -__builtin_inff();
        __syncthreads();  // (D is rewritten below: everybody is past its last read of it)
        for (int f = tid; f < a.n; f += 256) {  // D <- the kept samples in frame order, 0 where rejected
            const float u = U[f];
            const bool keep = all || (u >= lov && u <= hiv);
            D[f] = keep ? u : 0.0f;
            if (!keep) atomicAdd(&a.rej[f], 1ull);
        }
        __syncthreads();
        if (wave == 0) {
            const float sum = wave_serial_sum_f32(D, 0, a.n - 1, lane);
            if (lane == 0) a.out[g] = len == 0 ? 0.0f : sum / (float)len;
        }
        __syncthreads();  // the next pixel reuses the segment and the shared words
    }
}

// f64 sum of the calibrated samples of ONE frame per blockIdx.y (deep stacks only: the masters are re-read per frame)
__global__ __launch_bounds__(kSumBlock) void cal_frame_sum_kernel(const float *const *__restrict__ p, Masters m, uint32_t npix, double *__restrict__ part) {
    const float *frame = p[blockIdx.y];
    double s = 0.0;
    const uint32_t stride = gridDim.x * kSumBlock;
    for (uint32_t i = blockIdx.x * kSumBlock + threadIdx.x; i < npix; i += stride) s += (double)cal_apply(frame[i], cal_load(m, i));
    __shared__ double red[kSumBlock];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int k = kSumBlock / 2; k > 0; k >>= 1) {
        if ((int)threadIdx.x < k) red[threadIdx.x] += red[threadIdx.x + k];
        __syncthreads();
    }
    if (threadIdx.x == 0) part[(size_t)blockIdx.y * gridDim.x + blockIdx.x] = red[0];
}

int cu_of(ab_ctx *ctx) { return ctx->cu_count > 0 ? ctx->cu_count : 256; }
int np_for(int n) { return n <= 2 ? 2 : (n <= 4 ? 4 : (n <= 8 ? 8 : (n <= 16 ? 16 : (n <= 32 ? 32 : (n <= 64 ? 64 : 128))))); }

int download(ab_ctx *ctx, void *dst, const void *src, size_t bytes) {
    void *pin = nullptr;
    AB_TRY(ab_pinned(ctx, bytes, &pin));
    AB_HIP(ctx, hipMemcpyAsync(pin, src, bytes, hipMemcpyDeviceToHost, ctx->stream));
    AB_HIP(ctx, hipStreamSynchronize(ctx->stream));
    memcpy(dst, pin, bytes);
    return AB_OK;
}

// persistent kernels: exactly as many blocks as the chip keeps resident (a partial second round of 100-chunk blocks would
// double the kernel time), but never more than there are work items
template <class K>
int resident_grid(ab_ctx *ctx, K kernel, int block, size_t lds, int64_t items) {
    int per_cu = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, block, lds) != hipSuccess || per_cu < 1) per_cu = 1;
    return (int)std::max<int64_t>(1, std::min<int64_t>(items, (int64_t)cu_of(ctx) * per_cu));
}

template <int NP, bool CAL, bool FULL>
int launch_scms_np(ab_ctx *ctx, const BatchArgs &a, int64_t nchunks, uint32_t **rej_out) {
    constexpr int kThreads = kWave * kWavesPerBlock;
    const size_t lds = (size_t)NP * kThreads * sizeof(float);
    if constexpr (NP > 64) {  // 128 KiB of the CU's 160: above the default dynamic-LDS limit
        // (per device, and a process may hold contexts on several: set it on every launch, it is a table write)
        AB_HIP(ctx, hipFuncSetAttribute((const void *)scms_kernel<NP, CAL, FULL>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    }
    const int grid = resident_grid(ctx, scms_kernel<NP, CAL, FULL>, kThreads, lds, (nchunks + kWavesPerBlock - 1) / kWavesPerBlock);
    const int waves = grid * kWavesPerBlock;
    void *rej = nullptr;
    AB_TRY(ab_workspace(ctx, AB_WS_BATCH_REJ, ((size_t)waves + 2) * kMaxFrames * sizeof(unsigned long long), &rej));
    BatchArgs b = a;
    retain_thresholds(&b);
    b.guess_step = ab_dev_env("AB_BATCH_GUESS") ? atoi(ab_dev_env("AB_BATCH_GUESS")) : 1;
#ifdef AB_DEV_ABLATION  // (stage cuts for tools/time_batch.py: only in a -DAB_DEV_ABLATION build)
    b.stage = ab_dev_env("AB_BATCH_STAGE") ? atoi(ab_dev_env("AB_BATCH_STAGE")) : 0;
#else
    b.stage = 0;
#endif
    b.per_block = 0;  // strided chunks; a contiguous range per wave measured the same
    b.rej = (uint32_t *)((unsigned long long *)rej + kMaxFrames);  // [0, 64) u64 totals, then the per-block u32 partials
    hipLaunchKernelGGL((scms_kernel<NP, CAL, FULL>), dim3(grid), dim3(kThreads), lds, ctx->stream, b);
    AB_HIP(ctx, hipGetLastError());
    hipLaunchKernelGGL(rej_reduce_kernel, dim3(kMaxFrames), dim3(256), 0, ctx->stream, b.rej, waves, (unsigned long long *)rej);
    AB_HIP(ctx, hipGetLastError());
    *rej_out = (uint32_t *)rej;
    return AB_OK;
}

template <bool CAL, bool FULL>
int launch_scms(ab_ctx *ctx, int np, const BatchArgs &a, int64_t nchunks, uint32_t **rej_out) {
    switch (np) {
    case 2: return launch_scms_np<2, CAL, FULL>(ctx, a, nchunks, rej_out);
    case 4: return launch_scms_np<4, CAL, FULL>(ctx, a, nchunks, rej_out);
    case 8: return launch_scms_np<8, CAL, FULL>(ctx, a, nchunks, rej_out);
    case 16: return launch_scms_np<16, CAL, FULL>(ctx, a, nchunks, rej_out);
    case 32: return launch_scms_np<32, CAL, FULL>(ctx, a, nchunks, rej_out);
    case 64: return launch_scms_np<64, CAL, FULL>(ctx, a, nchunks, rej_out);
    default: return launch_scms_np<128, CAL, FULL>(ctx, a, nchunks, rej_out);
    }
}

template <int NP>
int launch_means_np(ab_ctx *ctx, const BatchArgs &a, std::vector<double> *part, int *grid_out) {
    constexpr int FR = NP < 32 ? NP : 32;
    const int kGroups = (a.n + FR - 1) / FR;  // groups that hold a real frame (the slots past n would only re-read frame 0)
    const int grid = std::max(1, resident_grid(ctx, cal_means_kernel<FR>, kSumBlock, 0, ((int64_t)a.npix + kSumBlock - 1) / kSumBlock * kGroups) / kGroups);
    void *d = nullptr;
    AB_TRY(ab_scratch(ctx, (size_t)grid * kMaxFrames * sizeof(double), &d));
    hipLaunchKernelGGL(cal_means_kernel<FR>, dim3(grid, kGroups), dim3(kSumBlock), 0, ctx->stream, a, (double *)d);
    AB_HIP(ctx, hipGetLastError());
    part->resize((size_t)grid * kMaxFrames);
    *grid_out = grid;
    return download(ctx, part->data(), d, part->size() * sizeof(double));
}

int launch_means(ab_ctx *ctx, int np, const BatchArgs &a, std::vector<double> *part, int *grid_out) {
    switch (np) {
    case 2: return launch_means_np<2>(ctx, a, part, grid_out);
    case 4: return launch_means_np<4>(ctx, a, part, grid_out);
    case 8: return launch_means_np<8>(ctx, a, part, grid_out);
    case 16: return launch_means_np<16>(ctx, a, part, grid_out);
    case 32: return launch_means_np<32>(ctx, a, part, grid_out);
    case 64: return launch_means_np<64>(ctx, a, part, grid_out);
    default: return launch_means_np<128>(ctx, a, part, grid_out);
    }
}

// sum (mode 0) or sum of squared deviations (mode 1) of a device plane: fixed-shape tree, partials added in block order
int plane_sum(ab_ctx *ctx, const float *data, int64_t n, int mode, double center, double *out) {
    const int grid = (int)std::max<int64_t>(1, std::min<int64_t>((n + kSumBlock - 1) / kSumBlock, (int64_t)cu_of(ctx) * 8));
    void *d = nullptr;
    AB_TRY(ab_scratch(ctx, grid * sizeof(double), &d));
    hipLaunchKernelGGL(sum_f64_kernel, dim3(grid), dim3(kSumBlock), 0, ctx->stream, data, n, mode, center, (double *)d);
    AB_HIP(ctx, hipGetLastError());
    std::vector<double> part(grid);
    AB_TRY(download(ctx, part.data(), d, grid * sizeof(double)));
    double s = 0.0;
    for (double p : part) s += p;
    *out = s;
    return AB_OK;
}

struct StagedMasters {
    StagedPlane p[3];
    bool staged[3] = {false, false, false};
    Masters m{nullptr, nullptr, nullptr};
};
// a master of another LENGTH is skipped, not an error (:87-89 compare slice lengths only)
int stage_masters(ab_ctx *ctx, const ab_calibration_masters *masters, int64_t npix, StagedMasters *out) {
    if (!masters) return AB_OK;
    const ab_plane *src[3] = {masters->bias, masters->dark, masters->flat};
    const float **dst[3] = {&out->m.bias, &out->m.dark, &out->m.flat};
    for (int k = 0; k < 3; ++k) {
        if (!src[k] || !src[k]->data || src[k]->rows * src[k]->cols != npix) continue;
        AB_TRY(ab_stage_in(ctx, src[k], &out->p[k]));
        out->staged[k] = true;
        *dst[k] = out->p[k].dptr;
    }
    return AB_OK;
}
void release_masters(ab_ctx *ctx, StagedMasters *s) {
    for (int k = 0; k < 3; ++k)
        if (s->staged[k]) ab_stage_release(ctx, &s->p[k]);
}

ab_batch_stack_config config_or_default(const ab_batch_stack_config *cfg) {
    ab_batch_stack_config c;
    ab_batch_stack_config_default(&c);
    if (cfg) c = *cfg;
    return c;
}

// 65 .. 2048 frames: tables in a workspace, one wave per pixel; beyond (or above AB_BATCH_DEEP_FROM): one workgroup per pixel
int stack_wide_device(ab_ctx *ctx, const float *const *frames, size_t n, int64_t npix, const Masters &m, bool cal, const ab_batch_stack_config &cfg,
                      float *out, uint64_t *rejection_counts) {
    char *ws = nullptr;
    const size_t ptr_bytes = n * sizeof(float *), scale_bytes = ((n * sizeof(float) + 7) / 8) * 8, rej_bytes = n * sizeof(unsigned long long);
    AB_TRY(ab_workspace(ctx, AB_WS_BATCH_WIDE, ptr_bytes + scale_bytes + rej_bytes, (void **)&ws));
    AB_HIP(ctx, hipStreamSynchronize(ctx->stream));
    AB_HIP(ctx, hipMemcpy(ws, frames, ptr_bytes, hipMemcpyHostToDevice));
    std::vector<float> scale(n, 1.0f);
    if (cal && cfg.normalize_before_stack) {  // normalize_frames (:309-319) on the calibrated samples
        const int gx = std::max(1, std::min<int>((int)((npix + kSumBlock - 1) / kSumBlock), cu_of(ctx) * 2));
        void *d = nullptr;
        AB_TRY(ab_scratch(ctx, (size_t)gx * n * sizeof(double), &d));
        hipLaunchKernelGGL(cal_frame_sum_kernel, dim3(gx, (unsigned)n), dim3(kSumBlock), 0, ctx->stream, (const float *const *)ws, m, (uint32_t)npix,
                           (double *)d);
        AB_HIP(ctx, hipGetLastError());
        std::vector<double> part((size_t)gx * n);
        AB_TRY(download(ctx, part.data(), d, part.size() * sizeof(double)));
        for (size_t f = 0; f < n; ++f) {
            double s = 0.0;
            for (int b = 0; b < gx; ++b) s += part[f * gx + b];
            const double mean = s / (double)npix;
            if (mean > 0.0) scale[f] = 1.0f / (float)mean;
        }
    }
    AB_HIP(ctx, hipMemcpy(ws + ptr_bytes, scale.data(), n * sizeof(float), hipMemcpyHostToDevice));
    AB_HIP(ctx, hipMemsetAsync(ws + ptr_bytes + scale_bytes, 0, rej_bytes, ctx->stream));
    WideBatchArgs a;
    a.p = (const float *const *)ws;
    a.scale = (const float *)(ws + ptr_bytes);
    a.rej = (unsigned long long *)(ws + ptr_bytes + scale_bytes);
    a.m = m;
    a.n = (int)n;
    a.npix = (uint32_t)npix;
    a.sigma_low = cfg.sigma_low;
    a.sigma_high = cfg.sigma_high;
    a.max_iter = (int)std::min<uint64_t>(cfg.max_iterations, 1u << 20);
    a.out = out;
    const int grid = (int)std::max<int64_t>(1, std::min<int64_t>((npix + 3) / 4, (int64_t)cu_of(ctx) * 8));
    if (n > (size_t)ctx->batch_deep_from) {
        DeepBatchArgs d;
        d.w = a;
        d.np2 = 2;
        while ((size_t)d.np2 < n) d.np2 <<= 1;
        const int64_t seg_bytes = (int64_t)3 * d.np2 * (int64_t)sizeof(float);
        int64_t dgrid = std::min<int64_t>(npix, (int64_t)cu_of(ctx) * 4);
        dgrid = std::max<int64_t>(1, std::min<int64_t>(dgrid, ((int64_t)1 << 30) / seg_bytes));
        AB_TRY(ab_workspace(ctx, AB_WS_BATCH_DEEP, (size_t)dgrid * (size_t)seg_bytes, (void **)&d.scratch));
        if (cal) hipLaunchKernelGGL((scms_deep_kernel<true>), dim3((unsigned)dgrid), dim3(256), 0, ctx->stream, d);
        else hipLaunchKernelGGL((scms_deep_kernel<false>), dim3((unsigned)dgrid), dim3(256), 0, ctx->stream, d);
    } else {
        const int kk = n <= 128 ? 2 : (n <= 256 ? 4 : (n <= 512 ? 8 : (n <= 1024 ? 16 : 32)));
#define AB_SCMS_WIDE(K)                                                                                             \
    do {                                                                                                            \
        if (cal) hipLaunchKernelGGL((scms_wide_kernel<K, true>), dim3(grid), dim3(256), 0, ctx->stream, a);         \
        else hipLaunchKernelGGL((scms_wide_kernel<K, false>), dim3(grid), dim3(256), 0, ctx->stream, a);            \
    } while (0)
        if (kk == 2) AB_SCMS_WIDE(2);
        else if (kk == 4) AB_SCMS_WIDE(4);
        else if (kk == 8) AB_SCMS_WIDE(8);
        else if (kk == 16) AB_SCMS_WIDE(16);
        else AB_SCMS_WIDE(32);
#undef AB_SCMS_WIDE
    }
    AB_HIP(ctx, hipGetLastError());
    std::vector<unsigned long long> host(n);
    AB_TRY(download(ctx, host.data(), a.rej, rej_bytes));
    if (rejection_counts)
        for (size_t f = 0; f < n; ++f) rejection_counts[f] = host[f];
    return AB_OK;
}

// the device-resident body shared by ab_sigma_clipped_mean_stack (cal == false) and ab_run_batch_channel
int stack_device(ab_ctx *ctx, const float *const *frames, size_t n, int64_t npix, const Masters &m, bool cal, const ab_batch_stack_config &cfg,
                 float *out, uint64_t *rejection_counts) {
    AB_CHECK(ctx, n >= 1 && n <= ((size_t)1 << 24), "sigma_clipped_mean_stack: %zu frames (1 .. 2^24 per call)", n);
    AB_CHECK(ctx, npix > 0 && npix < ((int64_t)1 << 30), "sigma_clipped_mean_stack: planes of 1 .. 2^30 - 1 pixels");
    AB_HIP(ctx, hipSetDevice(ctx->device));
    if (n > (size_t)kMaxFrames) return stack_wide_device(ctx, frames, n, npix, m, cal, cfg, out, rejection_counts);
    BatchArgs a;
    memset(&a, 0, sizeof a);
    for (int f = 0; f < kMaxFrames; ++f) {
        a.p[f] = frames[(size_t)f < n ? (size_t)f : 0];
        a.scale[f] = 1.0f;
    }
    a.m = m;
    a.n = (int)n;
    a.npix = (uint32_t)npix;
    a.sigma_low = cfg.sigma_low;
    a.sigma_high = cfg.sigma_high;
    a.max_iter = (int)std::min<uint64_t>(cfg.max_iterations, 1u << 20);
    a.out = out;
    const int np = np_for((int)n);
    if (cal && cfg.normalize_before_stack) {  // normalize_frames (:309-319) on the calibrated samples, never materialised
        std::vector<double> part;
        int grid = 0;
        AB_TRY(launch_means(ctx, np, a, &part, &grid));
        for (size_t f = 0; f < n; ++f) {
            double s = 0.0;
            for (int b = 0; b < grid; ++b) s += part[(size_t)b * kMaxFrames + f];
            const double mean = s / (double)npix;
            if (mean > 0.0) a.scale[f] = 1.0f / (float)mean;
        }
    }
    const int64_t nchunks = (npix + kWave - 1) / kWave;
    const bool full = (int)n == np;
    if (!full) {  // the slots past n read FLT_MAX (see scms_kernel); one plane, L2-resident, shared by all of them
        float *pad = nullptr;
        const void *before = ctx->ws[AB_WS_BATCH_PAD];
        const size_t had = ctx->ws_bytes[AB_WS_BATCH_PAD];
        AB_TRY(ab_workspace(ctx, AB_WS_BATCH_PAD, (size_t)npix * sizeof(float), (void **)&pad));
        if (pad != before || had < (size_t)npix * sizeof(float))
            AB_HIP(ctx, hipMemsetD32Async((hipDeviceptr_t)pad, 0x7f7fffff, ctx->ws_bytes[AB_WS_BATCH_PAD] / sizeof(float), ctx->stream));
        for (int f = (int)n; f < np; ++f) a.p[f] = pad;
    }
    uint32_t *total = nullptr;
    if (cal)
        AB_TRY((full ? launch_scms<true, true>(ctx, np, a, nchunks, &total) : launch_scms<true, false>(ctx, np, a, nchunks, &total)));
    else
        AB_TRY((full ? launch_scms<false, true>(ctx, np, a, nchunks, &total) : launch_scms<false, false>(ctx, np, a, nchunks, &total)));
    unsigned long long host[kMaxFrames];
    AB_TRY(download(ctx, host, total, sizeof host));
    if (rejection_counts)
        for (size_t f = 0; f < n; ++f) rejection_counts[f] = host[f];
    return AB_OK;
}

int check_frames(ab_ctx *ctx, const ab_plane *frames, size_t n, const char *what) {
    AB_CHECK(ctx, frames && n > 0, "%s: no frames", what);
    for (size_t i = 0; i < n; ++i) {
        AB_CHECK(ctx, frames[i].data && frames[i].rows > 0 && frames[i].cols > 0, "%s: frame %zu is null or has a zero dimension", what, i);
        AB_CHECK(ctx, frames[i].rows == frames[0].rows && frames[i].cols == frames[0].cols,
                 "%s: frame %zu has shape (%lld, %lld) but frame 0 has (%lld, %lld). All frames must match.", what, i, (long long)frames[i].rows,
                 (long long)frames[i].cols, (long long)frames[0].rows, (long long)frames[0].cols);
    }
    return AB_OK;
}

struct StagedFrames {
    std::vector<StagedPlane> st;
    std::vector<const float *> ptr;
    size_t staged = 0;
};
int stage_frames(ab_ctx *ctx, const ab_plane *frames, size_t n, StagedFrames *s) {
    s->st.resize(n);
    s->ptr.resize(n);
    for (; s->staged < n; ++s->staged) {
        AB_TRY(ab_stage_in(ctx, &frames[s->staged], &s->st[s->staged]));
        s->ptr[s->staged] = s->st[s->staged].dptr;
    }
    return AB_OK;
}
void release_frames(ab_ctx *ctx, StagedFrames *s) {
    for (size_t i = 0; i < s->staged; ++i) ab_stage_release(ctx, &s->st[i]);
}

int channel_minmax(ab_ctx *ctx, const float *p, int rows, int cols, int64_t ld, ChanNorm *c) {
    const int64_t n = (int64_t)rows * cols;
    const int grid = (int)std::max<int64_t>(1, std::min<int64_t>((n + 255) / 256, (int64_t)cu_of(ctx) * 8));
    void *d = nullptr;
    AB_TRY(ab_scratch(ctx, grid * sizeof(float2), &d));
    hipLaunchKernelGGL(minmax_kernel, dim3(grid), dim3(256), 0, ctx->stream, p, rows, cols, ld, (float2 *)d);
    AB_HIP(ctx, hipGetLastError());
    std::vector<float2> part(grid);
    AB_TRY(download(ctx, part.data(), d, grid * sizeof(float2)));
    float mn = INFINITY, mx = -INFINITY;
    for (const float2 &q : part) {
        mn = q.x < mn ? q.x : mn;
        mx = q.y > mx ? q.y : mx;
    }
    const float range = mx - mn;
    c->p = p;
    c->ld = ld;
    c->mn = mn;
    c->zero = range < 1e-10f;  // NaN range (no comparable sample, or inf - inf) is not < 1e-10: the map then yields NaN, as the reference's
    c->inv_range = 1.0f / range;
    return AB_OK;
}

}  // namespace

extern "C" {

void ab_batch_stack_config_default(ab_batch_stack_config *cfg) {  // calibration_pipeline.rs:28-37
    if (!cfg) return;
    cfg->sigma_low = 2.5f;
    cfg->sigma_high = 3.0f;
    cfg->max_iterations = 5;
    cfg->normalize_before_stack = 1;
}

int ab_calibrate_light(ab_ctx *ctx, const ab_plane *light, const ab_calibration_masters *masters, ab_plane_mut *out) try {
    if (!ctx) return AB_ERR_INVALID;
    AB_CHECK(ctx, light && out && out->rows == light->rows && out->cols == light->cols, "null plane or mismatched dims");
    StagedPlane in;
    AB_TRY(ab_stage_in(ctx, light, &in));
    const int64_t npix = in.rows * in.cols;
    StagedMasters sm;
    int rc = npix < ((int64_t)1 << 32) ? AB_OK : ab_set_error(ctx, AB_ERR_INVALID, "calibrate_light: planes of less than 2^32 pixels");
    if (rc == AB_OK) rc = stage_masters(ctx, masters, npix, &sm);
    StagedOut so;
    if (rc == AB_OK) rc = ab_stage_out_begin(ctx, out, &so);
    if (rc == AB_OK) {
        const int grid = (int)std::max<int64_t>(1, std::min<int64_t>((npix + 255) / 256, (int64_t)cu_of(ctx) * 16));
        hipLaunchKernelGGL(calibrate_kernel, dim3(grid), dim3(256), 0, ctx->stream, in.dptr, sm.m, (uint32_t)npix, so.dptr);
        if (hipGetLastError() != hipSuccess) {
            ab_stage_out_abort(ctx, &so);
            rc = ab_set_error(ctx, AB_ERR_HIP, "calibrate_light launch failed");
        } else {
            rc = ab_stage_out_finish(ctx, &so);
        }
    }
    release_masters(ctx, &sm);
    ab_stage_release(ctx, &in);
    return rc;
} AB_CATCH(ctx)

int ab_normalize_frames(ab_ctx *ctx, const ab_plane *frames, size_t n, ab_plane_mut *outs) try {
    if (!ctx) return AB_ERR_INVALID;
    AB_CHECK(ctx, (frames && outs) || n == 0, "null argument");
    for (size_t f = 0; f < n; ++f) {
        AB_CHECK(ctx, outs[f].rows == frames[f].rows && outs[f].cols == frames[f].cols, "normalize_frames: output %zu differs from its frame's dims", f);
        StagedPlane in;
        AB_TRY(ab_stage_in(ctx, &frames[f], &in));
        const int64_t npix = in.rows * in.cols;
        double s = 0.0;
        int rc = plane_sum(ctx, in.dptr, npix, 0, 0.0, &s);
        StagedOut so;
        if (rc == AB_OK) rc = ab_stage_out_begin(ctx, &outs[f], &so);
        if (rc == AB_OK) {
            const double mean = s / (double)npix;
            const int grid = (int)std::max<int64_t>(1, std::min<int64_t>((npix + 255) / 256, (int64_t)cu_of(ctx) * 16));
            if (mean > 0.0)
                hipLaunchKernelGGL(scale_kernel, dim3(grid), dim3(256), 0, ctx->stream, in.dptr, npix, 1.0f / (float)mean, so.dptr);
            else if (so.dptr != in.dptr)  // frame.clone()
                (void)hipMemcpyAsync(so.dptr, in.dptr, (size_t)npix * sizeof(float), hipMemcpyDeviceToDevice, ctx->stream);
            if (hipGetLastError() != hipSuccess) {
                ab_stage_out_abort(ctx, &so);
                rc = ab_set_error(ctx, AB_ERR_HIP, "normalize_frames launch failed");
            } else {
                rc = ab_stage_out_finish(ctx, &so);
            }
        }
        ab_stage_release(ctx, &in);
        if (rc != AB_OK) return rc;
    }
    return AB_OK;
} AB_CATCH(ctx)

int ab_sigma_clipped_mean_stack(ab_ctx *ctx, const ab_plane *frames, size_t n, const ab_batch_stack_config *config, ab_plane_mut *out,
                                uint64_t *rejection_counts) try {
    if (!ctx) return AB_ERR_INVALID;
    AB_CHECK(ctx, out, "null argument");
    AB_TRY(check_frames(ctx, frames, n, "sigma_clipped_mean_stack"));
    AB_CHECK(ctx, out->rows == frames[0].rows && out->cols == frames[0].cols, "sigma_clipped_mean_stack: the output differs from the frames' dims");
    const ab_batch_stack_config cfg = config_or_default(config);
    StagedFrames sf;
    int rc = stage_frames(ctx, frames, n, &sf);
    StagedOut so;
    if (rc == AB_OK) rc = ab_stage_out_begin(ctx, out, &so);
    if (rc == AB_OK) {
        rc = stack_device(ctx, sf.ptr.data(), n, frames[0].rows * frames[0].cols, Masters{nullptr, nullptr, nullptr}, false, cfg, so.dptr, rejection_counts);
        if (rc == AB_OK)
            rc = ab_stage_out_finish(ctx, &so);
        else
            ab_stage_out_abort(ctx, &so);
    }
    release_frames(ctx, &sf);
    return rc;
} AB_CATCH(ctx)

int ab_run_batch_channel(ab_ctx *ctx, const ab_plane *lights, size_t n, const ab_calibration_masters *masters, const ab_batch_stack_config *config,
                         ab_plane_mut *out_master, uint64_t *rejection_counts, ab_batch_channel_stats *stats) try {
    if (!ctx) return AB_ERR_INVALID;
    AB_CHECK(ctx, out_master, "null argument");
    AB_TRY(check_frames(ctx, lights, n, "run_batch_pipeline"));
    AB_CHECK(ctx, out_master->rows == lights[0].rows && out_master->cols == lights[0].cols, "run_batch_pipeline: the master differs from the lights' dims");
    const ab_batch_stack_config cfg = config_or_default(config);
    const int64_t npix = lights[0].rows * lights[0].cols;
    StagedFrames sf;
    StagedMasters sm;
    int rc = stage_frames(ctx, lights, n, &sf);
    if (rc == AB_OK) rc = stage_masters(ctx, masters, npix, &sm);
    StagedOut so;
    if (rc == AB_OK) rc = ab_stage_out_begin(ctx, out_master, &so);
    if (rc == AB_OK) {
        rc = stack_device(ctx, sf.ptr.data(), n, npix, sm.m, true, cfg, so.dptr, rejection_counts);
        double s = 0.0, q = 0.0;
        if (rc == AB_OK && stats) {  // :173-179
            rc = plane_sum(ctx, so.dptr, npix, 0, 0.0, &s);
            if (rc == AB_OK) rc = plane_sum(ctx, so.dptr, npix, 1, s / (double)npix, &q);
            if (rc == AB_OK) {
                stats->lights_input = n;
                stats->mean = s / (double)npix;
                stats->stddev = std::sqrt(q / (double)npix);
            }
        }
        if (rc == AB_OK)
            rc = ab_stage_out_finish(ctx, &so);
        else
            ab_stage_out_abort(ctx, &so);
    }
    release_masters(ctx, &sm);
    release_frames(ctx, &sf);
    return rc;
} AB_CATCH(ctx)

int ab_compose_rgb_from_masters(ab_ctx *ctx, const ab_plane *r, const ab_plane *g, const ab_plane *b, const ab_plane *l, float *out_rgb,
                                int32_t out_on_device, int64_t *out_rows, int64_t *out_cols) try {
    if (!ctx) return AB_ERR_INVALID;
    AB_CHECK(ctx, r && g && b && r->data && g->data && b->data, "null argument");
    AB_CHECK(ctx, r->rows > 0 && r->cols > 0 && g->rows > 0 && g->cols > 0 && b->rows > 0 && b->cols > 0, "a master has a zero dimension");
    const bool same = g->rows == r->rows && g->cols == r->cols && b->rows == r->rows && b->cols == r->cols;
    const int64_t h = std::min(r->rows, std::min(g->rows, b->rows)), w = std::min(r->cols, std::min(g->cols, b->cols));  // :211-213
    if (out_rows) *out_rows = h;
    if (out_cols) *out_cols = w;
    if (!out_rgb) return AB_OK;  // size query
    const bool with_l = same && l && l->data && l->rows == h && l->cols == w;  // :237-238
    const ab_plane *src[4] = {r, g, b, l};
    StagedPlane st[4];
    int staged = 0, rc = AB_OK;
    ComposeArgs a;
    memset(&a, 0, sizeof a);
    ChanNorm *cn[4] = {&a.r, &a.g, &a.b, &a.l};
    for (int c = 0; c < (with_l ? 4 : 3) && rc == AB_OK; ++c) {
        rc = ab_stage_in(ctx, src[c], &st[c]);
        if (rc != AB_OK) break;
        ++staged;
        rc = channel_minmax(ctx, st[c].dptr, (int)h, (int)w, src[c]->cols, cn[c]);
    }
    float *dout = out_rgb;
    void *owned = nullptr;
    const size_t bytes = (size_t)(h * w) * 3 * sizeof(float);
    if (rc == AB_OK && !out_on_device) {
        if (hipMalloc(&owned, bytes) != hipSuccess) rc = ab_set_error(ctx, AB_ERR_HIP, "cannot allocate the RGB cube");
        dout = (float *)owned;
    }
    if (rc == AB_OK) {
        a.with_l = with_l;
        a.rows = (int)h;
        a.cols = (int)w;
        a.out = dout;
        const int grid = (int)std::max<int64_t>(1, std::min<int64_t>((h * w + 255) / 256, (int64_t)cu_of(ctx) * 16));
        hipLaunchKernelGGL(compose_masters_kernel, dim3(grid), dim3(256), 0, ctx->stream, a);
        if (hipGetLastError() != hipSuccess) rc = ab_set_error(ctx, AB_ERR_HIP, "compose_rgb_from_masters launch failed");
        if (rc == AB_OK && owned &&
            (hipMemcpyAsync(out_rgb, owned, bytes, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess || hipStreamSynchronize(ctx->stream) != hipSuccess))
            rc = ab_set_error(ctx, AB_ERR_HIP, "download of the RGB cube failed");
    }
    if (owned) {
        (void)hipStreamSynchronize(ctx->stream);
        (void)hipFree(owned);
    }
    for (int c = 0; c < staged; ++c) ab_stage_release(ctx, &st[c]);
    return rc;
} AB_CATCH(ctx)

int ab_run_batch_pipeline(ab_ctx *ctx, const ab_batch_channel_input *channels, size_t n_channels, const ab_calibration_masters *masters,
                          const ab_batch_stack_config *config, ab_plane_mut *out_masters, ab_batch_channel_stats *stats, float *out_rgb,
                          int32_t rgb_on_device, int64_t *rgb_rows, int64_t *rgb_cols) try {
    if (!ctx) return AB_ERR_INVALID;
    AB_CHECK(ctx, channels && n_channels > 0, "No channels provided");  // :125-127
    AB_CHECK(ctx, out_masters, "null argument");
    for (size_t c = 0; c < n_channels; ++c) {  // :129-142
        const char *label = channels[c].label ? channels[c].label : "";
        AB_CHECK(ctx, channels[c].lights && channels[c].n_lights > 0, "Channel '%s' has no lights", label);
        for (size_t i = 1; i < channels[c].n_lights; ++i)
            AB_CHECK(ctx, channels[c].lights[i].rows == channels[c].lights[0].rows && channels[c].lights[i].cols == channels[c].lights[0].cols,
                     "Channel '%s': frame %zu has shape (%lld, %lld) but frame 0 has (%lld, %lld). All frames must match.", label, i,
                     (long long)channels[c].lights[i].rows, (long long)channels[c].lights[i].cols, (long long)channels[c].lights[0].rows,
                     (long long)channels[c].lights[0].cols);
    }
    for (size_t c = 0; c < n_channels; ++c)
        AB_TRY(ab_run_batch_channel(ctx, channels[c].lights, channels[c].n_lights, masters, config, &out_masters[c], channels[c].rejection_counts,
                                    stats ? &stats[c] : nullptr));
    // compose_rgb_from_masters (:201-207): the first master labelled R / G / B (and L), ASCII case-insensitive
    auto find = [&](const char *want) -> const ab_plane_mut * {
        for (size_t c = 0; c < n_channels; ++c)
            if (channels[c].label && strcasecmp(channels[c].label, want) == 0) return &out_masters[c];
        return nullptr;
    };
    const ab_plane_mut *m[4] = {find("R"), find("G"), find("B"), find("L")};
    if (rgb_rows) *rgb_rows = 0;
    if (rgb_cols) *rgb_cols = 0;
    if (!m[0] || !m[1] || !m[2]) return AB_OK;  // rgb = None
    ab_plane in[4];
    for (int k = 0; k < 4; ++k)
        if (m[k]) in[k] = ab_plane{m[k]->data, m[k]->rows, m[k]->cols, m[k]->on_device};
    return ab_compose_rgb_from_masters(ctx, &in[0], &in[1], &in[2], m[3] ? &in[3] : nullptr, out_rgb, rgb_on_device, rgb_rows, rgb_cols);
} AB_CATCH(ctx)

}  // extern "C"
