// Per-pixel kappa-sigma stacking on gfx950.
//
// Replaces core/stacking/combine.rs: sigma_clip_combine (:14-92) and the per-pixel loop of
// stack_images (:160-182).
//
// Layout / mapping (MI355X-first, not a translation of the rayon row loop):
//   * frames stay as N separate row-major planes in HBM, exactly as the reference holds
//     them (combine.rs:152-155).  One LANE owns one output pixel; a wavefront therefore reads
//     64 consecutive pixels = one fully coalesced 256-byte segment per frame, and issues all
//     N loads before it consumes the first (N x 256 B in flight per wave).
//   * the N samples of a pixel live in VGPRs for the whole computation (N <= 64).  They are
//     sorted once with a Batcher merge-exchange network whose indices are compile-time
//     constants (tools/gen_sortnet.py), because a runtime-indexed private array would be
//     demoted to scratch memory.
//   * with the samples sorted, every later step of the reference algorithm becomes cheap:
//       - median  = s[n/2]                                   (combine.rs:38-40)
//       - MAD     = the n/2-th smallest of |s_i - med|.  The deviations left and right of the
//                   median are two sorted runs A and B, and the k-th element of their merge is
//                   min over splits i+j=k+1 of max(A[i-1], B[j-1]) -- ~2N min/max instead of
//                   a second selection                          (combine.rs:42-46)
//       - the survivors of each clipping pass form an INTERVAL [a, b] of the sorted order,
//         because `dev = v - center` is monotone in v           (combine.rs:65-74)
//   * iterations >= 1 sum the survivors in ascending value order in f64 (the reference's order
//     is the unspecified post-select permutation; see include/astroburst_hip.h).
//   * rejected-pixel count: per-lane u32 -> wavefront shuffle reduction -> one 64-bit atomic
//     per wave                                                  (combine.rs:158,181)
//
// HBM traffic is the algorithmic minimum: 4*N*P bytes read once, 4*P written.
#include "ab_common.hpp"

#include <cmath>

// -inf / +inf the optimiser cannot see through (one SGPR each): v_med3_f32(x, y, -inf) IS min(x, y), but spelled fminf() -- or as
// a med3 with a literal infinity, which instcombine folds back to fminf() -- it is preceded by a canonicalising v_max_f32 x, x, x
// whenever the compiler cannot prove an operand is not a signalling NaN (a freshly loaded sample, the result of an integer
// operation).  No NaN reaches the network (non-finite samples are replaced by +inf above it).
__device__ __forceinline__ float ab_ninf() {
    float x = -__builtin_inff();
    asm("" : "+s"(x));
    return x;
}
__device__ __forceinline__ float ab_pinf() {
    float x = __builtin_inff();
    asm("" : "+s"(x));
    return x;
}
// Compare-exchange.  v_min_f32 / v_max_f32 / v_med3_f32 are half-rate on gfx950 (4.3 cycles per wave64 instruction,
// tools/valu_rate.hip), and a sorting network is nothing else.  AB_STACK_CE_XOR: the larger element is not compared for a second
// time -- the minimum is one of the two inputs bit for bit (no NaN reaches the network, denormals are not flushed), so the other
// one is a ^ b ^ min, a single v_bitop3_b32.
#if !defined(AB_STACK_NO_CE_XOR) && !defined(AB_STACK_CE_XOR)
#define AB_STACK_CE_XOR 1
#endif
#ifdef AB_STACK_CE_XOR
#define AB_CE(a, b)                                                                                                   \
    {                                                                                                                 \
        const T lo_ = __builtin_amdgcn_fmed3f(v[a], v[b], ab_ninf());                                                 \
        v[b] = __uint_as_float(__builtin_amdgcn_bitop3_b32(__float_as_uint(v[a]), __float_as_uint(v[b]), __float_as_uint(lo_), 0x96)); \
        v[a] = lo_;                                                                                                   \
    }
#else
#define AB_CE(a, b)                         \
    {                                       \
        T lo_ = fminf(v[a], v[b]);          \
        T hi_ = fmaxf(v[a], v[b]);          \
        v[a] = lo_;                         \
        v[b] = hi_;                         \
    }
#endif
// The base case of the network on four unsorted samples with the three-input operations: sort three (min3 / med3 / max3), then
// place the fourth (min, med3, med3, max) -- 7 instructions where Batcher's five compare-exchanges take 10, all of them at the
// same half rate as a two-input min / max (tools/valu_rate.hip): 48 of the sort's 1086 instructions, 1.18 -> 1.14 ms on the bench
// stack (same box, AB_LIB_PATH A/B).
#ifndef AB_STACK_NO_SORT4  // (A/B switch for tools/time_stack_bench_data.py)
#define AB_SORT4(a, b, c, d)                                                         \
    {                                                                                \
        const T x0_ = v[a], x1_ = v[b], x2_ = v[c], x3_ = v[d];                      \
        const T s0_ = fminf(fminf(x0_, x1_), x2_), s1_ = __builtin_amdgcn_fmed3f(x0_, x1_, x2_), \
                s2_ = fmaxf(fmaxf(x0_, x1_), x2_);                                   \
        v[a] = __builtin_amdgcn_fmed3f(AB_SORT4_NINF, s0_, x3_); /* = min */         \
        v[b] = __builtin_amdgcn_fmed3f(s0_, s1_, x3_);                               \
        v[c] = __builtin_amdgcn_fmed3f(s1_, s2_, x3_);                               \
        v[d] = __builtin_amdgcn_fmed3f(AB_SORT4_PINF, s2_, x3_); /* = max */         \
    }
#endif
#ifdef AB_STACK_LITERAL_INF  // (A/B: round 2's form -- the compiler turns these two into v_min / v_max + one canonicalising v_max per base case)
#define AB_SORT4_NINF (-__builtin_inff())
#define AB_SORT4_PINF (__builtin_inff())
#else
#define AB_SORT4_NINF ab_ninf()
#define AB_SORT4_PINF ab_pinf()
#endif
#include "sort_ops.hpp"  // the min / max / min3 / med3 / max3 of SortNet<NP>::sort_fused as inline assembly, AB_SN_* macros
#include "sortnet_gen.hpp"

#ifndef AB_STACK_WAVES_PER_SIMD
#define AB_STACK_WAVES_PER_SIMD 3  // 4 (128 VGPRs): 15 spilled registers, measured no faster
#endif

namespace {

constexpr int kMaxFrames = 256;  // 65 .. 256: 2 / 4 registers of plane pointers, one wave per SIMD (see launch of NP = 128 / 256)
constexpr int kMaxStrided = 64;  // ragged row strides only exist for the <= 64-frame kernels (deeper stacks are DIRECT or wide)
constexpr int kRejSlots = AB_REJ_SLOTS;  // rejection counters (see the kernel epilogue)
constexpr int kDeferSlots = 2048;        // deferred-pixel lists (same reason: no hot atomic address)
constexpr unsigned int kGenWaves = 2;    // general pass: single-wave workgroups per deferred-pixel list (a list holds ~65 pixels on the bench stack, 109 at most)
#ifndef AB_STACK_DEFER_CHUNKS
#define AB_STACK_DEFER_CHUNKS 2
#endif
constexpr int kDeferChunks = AB_STACK_DEFER_CHUNKS;          // chunks of 4 samples the fast pass may examine at either end before it defers a pixel (3: 1.165 ms against 1.116, every wave pays for the larger code)
enum { kPlain = 0, kFastPass = 1, kGeneralPass = 2 };
enum { kInNative = 0, kInF32BE = 1, kInI16BE = 2 };  // sample encodings the gather understands
constexpr double kMadToSigma = 1.4826;  // types/constants.rs:7

struct StackArgs {
    const float *p[kMaxFrames];
    int64_t ld[kMaxStrided];  // row stride (= cols of that plane): top-left crop for free
    int n;                   // frames actually present (<= NP)
    int n_real;              // DIRECT single-pass kernel: slots [n_real, n) alias the +inf pad plane (n_real == n: no pads)
    int contiguous;          // all ld == cols: linear pixel index is the element offset
    int64_t rows, cols;      // output dims
    float sigma_low, sigma_high;
    uint32_t max_iter;
    float *out;                      // full mode
    double *out_sum;                 // partial mode
    uint32_t *out_cnt;               // partial mode
    unsigned long long *rejected;    // device counter
    // two-pass mode (see stack_sigma_clip_kernel): per-slot lists of the pixels the fast pass hands to the general pass
    int *defer_list;                 // kDeferSlots x defer_cap pixel indices
    unsigned int *defer_count;       // kDeferSlots counters
    unsigned int *defer_ticket;      // kDeferSlots arrival counters of the general pass's workgroups
    unsigned int defer_cap;
    int keep_counts;                 // AB_TRACE: leave the counters for the host to read
    // raw FITS input (INPUT != kInNative): p[] point at big-endian data units, decoded on load as decode_pixels does
    int identity;                    // is_identity_scaling(bscale, bzero) (reader.rs:36-39)
    double bscale, bzero;
};

// Compiler fences (no instructions).  launder() makes the sample vector look rewritten so LLVM
// does not hoist 64 f32->f64 conversions (128 VGPRs) out of the clipping loop; opaque() stops it
// from keeping 64 interval masks alive in SGPRs across the passes of one iteration.
template <int NP>
__device__ __forceinline__ void launder(float (&v)[NP]) {
    if constexpr (NP >= 8) {
#pragma unroll
        for (int i = 0; i < NP; i += 8)
            asm volatile("" : "+v"(v[i]), "+v"(v[i + 1]), "+v"(v[i + 2]), "+v"(v[i + 3]), "+v"(v[i + 4]),
                         "+v"(v[i + 5]), "+v"(v[i + 6]), "+v"(v[i + 7]));
    } else {
#pragma unroll
        for (int i = 0; i < NP; ++i) asm volatile("" : "+v"(v[i]));
    }
}
__device__ __forceinline__ void opaque(int &a, int &b) { asm volatile("" : "+v"(a), "+v"(b)); }

// sum over sorted positions a..b of (double)v[i], ascending, one f64 add per element; the f32
// select happens before the conversion (x + 0.0 is exact).
template <int NP>
__device__ __forceinline__ double masked_sum(const float (&v)[NP], int a, int b) {
    double S = 0.0;
#pragma unroll
    for (int i = 0; i < NP; ++i) {
        const bool in = (unsigned)(i - a) <= (unsigned)(b - a);
        const float xs = in ? v[i] : 0.0f;
        S += (double)xs;
    }
    return S;
}

// Median and MAD of the n finite samples of one pixel from the SORTED vector v (pads = +inf on
// top), for the wave-uniform candidate median position MM = n/2 (n = 2MM or 2MM+1):
//   med = v[MM]                                                        (combine.rs:38-40)
//   MAD = (n/2)-th smallest of |v_i - med|                              (combine.rs:42-46)
// The deviations left of the median (A[j] = med - v[MM-j], ascending in j) and right of it
// (B[j] = v[MM+1+j] - med) are two sorted runs, and the k-th element of their merge is
//   min over splits i + j = k + 1 of max(A[i-1], B[j-1]);
// with k = MM the splits pair v[p] with v[p+MM]: term(p) = max(med - v[p], v[p+MM] - med),
// p = 1 .. n-MM-1, plus the p = 0 term med - v[0].  For even n the p = MM term reads the +inf pad
// at v[2MM] and drops out by itself, so one instruction stream serves both parities.
// ~2N min/max/sub instead of a second selection, and every register index is a constant.
template <int NP, int MM>
__device__ __forceinline__ void med_mad_at(const float (&v)[NP], float &med_out, float &mad_out) {
    const float med = v[MM];
    float best = med - v[0];
#pragma unroll
    for (int p = 1; p <= MM; ++p) {
        if (p + MM < NP) {
            const float a = med - v[p];
            const float b = v[p + MM] - med;
            best = fminf(best, fmaxf(a, b));
        }
    }
    med_out = med;
    mad_out = best;
}

// lanes of one wave can disagree on n/2 (non-finite samples): evaluate each position present.  `cur` is wave-uniform;
// the positions [LO, HI] are bisected (a linear chain of NP/2 compares hops over as many cold code blocks, one
// instruction-cache miss each: a padded 200-frame stack spent a quarter of its time there).
template <int NP, int LO, int HI>
__device__ __forceinline__ void med_mad_dispatch(const float (&v)[NP], int m, int cur, float &med, float &mad) {
    if constexpr (LO == HI) {
        float md, ma;
        med_mad_at<NP, LO>(v, md, ma);
        if (m == LO) {
            med = md;
            mad = ma;
        }
    } else {
        constexpr int MID = (LO + HI) / 2;
        if (cur <= MID)
            med_mad_dispatch<NP, LO, MID>(v, m, cur, med, mad);
        else
            med_mad_dispatch<NP, MID + 1, HI>(v, m, cur, med, mad);
    }
}

// Per-lane state handed from the prologue (gather + sort + median/MAD) to a clipping engine.
struct ClipResult {
    float value;   // sigma_clip_combine's f32 result
    double sum;    // f64 sum of the survivors   (partial mode)
    int len;       // number of survivors        (partial mode)
    uint32_t rej;  // rejected samples of this pixel
    bool defer = false;  // fast pass only: this pixel needs the general pass
};

// ---- clipping engine A: "exact" -- every iteration re-sums the survivors directly --------------
// Bit-for-bit the oracle's ORC_ORDER_ASCENDING arithmetic (two-pass mean / sum of squared
// deviations over the interval, ascending).  ~1500 VALU slots per iteration; kept as the
// in-library cross-check of the fast engine (AB_STACK_EXACT=1) and for its provable order.
template <int NP>
__device__ __forceinline__ ClipResult clip_exact(float (&v)[NP], int n, float med, float mad,
                                                 float sigma_low, float sigma_high, uint32_t max_iter) {
    float sigma = (float)fmax((double)mad * kMadToSigma, 1e-10);
    float center = med;
    int a = 0, b = n - 1, len = n;
    uint32_t rej = 0;
    float last_center = __builtin_nanf("");
    bool active = (n >= 2);

    for (uint32_t it = 0; it < max_iter; ++it) {
        if (!__any(active)) break;
        launder<NP>(v);  // keep f32->f64 conversions inside the iteration (VGPR pressure)
        if (it > 0) {
            // mean / sample variance of the survivors in f64, ascending order (combine.rs:50-60)
            const double S = masked_sum<NP>(v, a, b);
            const double nn = (double)len;
            const double mean = S / nn;
            opaque(a, b);
            double Q = 0.0;
#pragma unroll
            for (int i = 0; i < NP; ++i) {
                const bool in = (unsigned)(i - a) <= (unsigned)(b - a);
                const double dd = (double)v[i] - mean;
                const double sq = dd * dd;
                Q += in ? sq : 0.0;
            }
            const double variance = Q / fmax(nn - 1.0, 1.0);
            center = (float)mean;
            sigma = (float)fmax(sqrt(variance), 1e-10);
            opaque(a, b);
        }
        const bool go = active && (len >= 2);  // `if len < 2 { break }` (combine.rs:33-35)
        if (go) last_center = center;          // combine.rs:63

        const float lo = -sigma_low * sigma;  // combine.rs:65-66
        const float hi = sigma_high * sigma;
        // survivors stay an interval of the sorted order: count what falls off either end
        int cl = 0, ch = 0;
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            const bool in = (unsigned)(i - a) <= (unsigned)(b - a);
            const float dev = v[i] - center;
            cl += (in && !(dev >= lo)) ? 1 : 0;
            ch += (in && !(dev <= hi)) ? 1 : 0;
        }
        // a sample can fail both tests only if nothing survives (lo > hi or NaN thresholds)
        const int removed = (cl + ch > len) ? len : (cl + ch);
        if (go) {
            rej += (uint32_t)removed;  // combine.rs:76-78
            len -= removed;
            if (len > 0) {
                a += cl;
                b -= ch;
            } else {
                a = 1;
                b = 0;
            }
        }
        active = go && (removed != 0);  // combine.rs:80-82
    }

    // ---- result (combine.rs:20-26,85-91) ----
    launder<NP>(v);
    opaque(a, b);
    const double S = masked_sum<NP>(v, a, b);  // empty interval (a=1,b=0 or n=0) sums to 0
    ClipResult r;
    if (n == 0) {
        r.value = 0.0f;
    } else if (n == 1) {
        r.value = med;  // the single finite sample is v[0] = v[n/2]
    } else if (len == 0) {
        r.value = __builtin_isfinite(last_center) ? last_center : 0.0f;
    } else {
        r.value = (float)(S / (double)len);
    }
    r.sum = (len > 0) ? S : 0.0;
    r.len = len > 0 ? len : 0;
    r.rej = rej;
    return r;
}

// ---- clipping engine B: "fast" -- incremental statistics, only the ends are re-examined ---------
// Because the survivors are an interval [a, b] of the sorted samples and an iteration can only
// shave samples off its two ends:
//   * one pass after the median/MAD clip accumulates  S1 = sum x_i  and  Q1 = sum (x_i - c0)^2
//     over the survivors (f64, ascending; c0 = the median, so every term is small and nothing
//     cancels -- the big outliers are already gone);
//   * iteration k >= 1 gets  mean = (S1 - S_rem)/n  and
//     sum (x_i - mean)^2 = (Q1 - Q_rem) - n (mean - c0)^2      (same algebra as combine.rs:50-59)
//     where S_rem / Q_rem collect the few samples shaved off so far;
//   * the clip test walks inwards from each end in chunks of 4 registers and stops (wave-
//     uniformly) at the first chunk in which every lane has met a surviving sample.
// The mean is exact whenever the f64 partial sums are (samples within 2^23 of each other); the
// variance differs from the two-pass value by a few ulp(f64), i.e. the f32 sigma is identical
// except with probability ~1e-8 per pixel -- far inside the 1e-5 contract, and measured in
// tests/ against engine A and the oracle.
// wave reductions on DPP moves (row_shr 1/2/4/8, row_bcast 15 / 31; lanes without a source take the identity) and one v_readlane:
// VALU only, where the __shfl_xor butterfly is six dependent ds_bpermute round trips
template <int OP>  // 0 sum, 1 min, 2 max (signed)
__device__ __forceinline__ int wave_reduce_i32(int x) {
    constexpr int id = OP == 1 ? 0x7fffffff : (OP == 2 ? (int)0x80000000 : 0);
    auto op = [](int a, int b) { return OP == 0 ? a + b : (OP == 1 ? min(a, b) : max(a, b)); };
    x = op(x, __builtin_amdgcn_update_dpp(id, x, 0x111, 0xf, 0xf, false));
    x = op(x, __builtin_amdgcn_update_dpp(id, x, 0x112, 0xf, 0xf, false));
    x = op(x, __builtin_amdgcn_update_dpp(id, x, 0x114, 0xf, 0xf, false));
    x = op(x, __builtin_amdgcn_update_dpp(id, x, 0x118, 0xf, 0xf, false));
    x = op(x, __builtin_amdgcn_update_dpp(id, x, 0x142, 0xa, 0xf, false));
    x = op(x, __builtin_amdgcn_update_dpp(id, x, 0x143, 0xc, 0xf, false));
    return __builtin_amdgcn_readlane(x, 63);
}
template <int NP>
__device__ __forceinline__ int wave_max_i32(int x) {
    return wave_reduce_i32<2>(x);
}
template <int NP>
__device__ __forceinline__ int wave_min_i32(int x) {
    return wave_reduce_i32<1>(x);
}

// One clipping pass over the two ends.  UPDATE: also fold the shaved samples into e_rem / q_rem (RAW: moments about 0).
// DEFER (fast pass): look at the outermost chunk of each end, and at a second one only if some lane of the wave asks for it;
// a lane that would have to walk further still is flagged for the general pass instead of making its whole wave walk with
// it.  (Round 1 stopped after ONE chunk: every pixel with 5 .. 8 rejected samples at one end -- the bands where a few frames
// carry a zero border -- went to the general pass, whose scattered gathers cost ~6x a fast-pass pixel; a second chunk costs the
// waves that need it ~40 instructions and everybody else one scalar branch.)
// SKIP (single-pass kernel): the high-end walk starts at the chunk that holds the wave's largest b instead of stepping
// over the pads of a ragged / padded stack four registers at a time (129 frames in 256 slots: 32 chunks per pass).
// The UPDATE of a chunk sits behind a wave-uniform BRANCH that the compiler must not turn into selects (round 2's form --
// `if (__any(r)) { ... }` per sample -- was if-converted: every examined sample paid a conversion, three f64 operations and
// four 32-bit selects whether or not any lane had shaved it, ~120 instructions per iteration; the asm volatile in the block
// is what keeps it a branch).
template <int NP, bool UPDATE, bool DEFER = false, bool SKIP = false, bool RAW = false>
__device__ __forceinline__ void clip_ends(const float (&v)[NP], bool go, int a, int b, float center, float lo, float hi,
                                          double c0d, float c0, int &cl_out, int &ch_out, double &e_rem, double &q_rem,
                                          bool *defer = nullptr, int skip_hi = 0) {
    constexpr int CH = NP >= 4 ? 4 : NP;  // DEFER: the only chunk looked at (8 was tried: +0.13 ms, the walk is per-element bound)
    int cl = 0, ch = 0;
    bool found_lo = false, found_hi = false;
#pragma unroll
    for (int c = 0; c < NP / CH; ++c) {
        bool r[CH];
        bool any_r = false;
#pragma unroll
        for (int j = 0; j < CH; ++j) {
            const int i = CH * c + j;
            const bool in = (i >= a) && (i <= b);
            const float dev = v[i] - center;
            const bool ok = dev >= lo;
            r[j] = go && in && !ok;
            found_lo = found_lo || (in && ok);
            cl += r[j] ? 1 : 0;
            any_r = any_r || r[j];
        }
        if constexpr (UPDATE) {
            if (__any(any_r)) {
                asm volatile("" ::: "memory");  // keeps this a branch (see above)
#pragma unroll
                for (int j = 0; j < CH; ++j) {
                    const float xm = r[j] ? v[CH * c + j] : c0;
                    const double e = RAW ? (double)xm : (double)xm - c0d;  // 0 for lanes that keep the sample
                    e_rem += e;
                    q_rem = __builtin_fma(e, e, q_rem);
                }
            }
        }
        const bool more = go && !found_lo && (CH * c + CH - 1 < b);
        if constexpr (DEFER) {
            if (c == kDeferChunks - 1 || NP / CH == 1) {
                *defer = *defer || more;
                break;
            }
        }
        if (!__any(more)) break;
    }
#pragma unroll
    for (int c = 0; c < NP / CH; ++c) {
        if constexpr (SKIP) {
            if (c < skip_hi) continue;  // wave-uniform: every slot of this chunk lies above every lane's b
        }
        bool r[CH];
        bool any_r = false;
#pragma unroll
        for (int j = 0; j < CH; ++j) {
            const int i = NP - 1 - (CH * c + j);
            const bool in = (i >= a) && (i <= b);
            const float dev = v[i] - center;
            const bool ok = dev <= hi;
            r[j] = go && in && !ok;
            found_hi = found_hi || (in && ok);
            ch += r[j] ? 1 : 0;
            any_r = any_r || r[j];
        }
        if constexpr (UPDATE) {
            if (__any(any_r)) {
                asm volatile("" ::: "memory");
#pragma unroll
                for (int j = 0; j < CH; ++j) {
                    const float xm = r[j] ? v[NP - 1 - (CH * c + j)] : c0;
                    const double e = RAW ? (double)xm : (double)xm - c0d;
                    e_rem += e;
                    q_rem = __builtin_fma(e, e, q_rem);
                }
            }
        }
        const bool more = go && !found_hi && (NP - 1 - (CH * c + CH - 1) > a);
        if constexpr (DEFER) {
            if (c == kDeferChunks - 1 || NP / CH == 1) {
                *defer = *defer || more;
                break;
            }
        }
        if (!__any(more)) break;
    }
    cl_out = cl;
    ch_out = ch;
}

// Iteration 0 of the fast pass when every lane of the wave holds all NP samples (see clip_fast): the outermost kDeferChunks chunks of
// either end, no interval tests.  Same results as clip_ends<NP, false, true, false> with a = 0, b = NP - 1, go = true.
template <int NP>
__device__ __forceinline__ void clip_first_full(const float (&v)[NP], float center, float lo, float hi, int &cl_out, int &ch_out, bool *defer) {
    int cl = 0, ch = 0;
    bool found = false;
#pragma unroll
    for (int c = 0; c < kDeferChunks; ++c) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const bool ok = (v[4 * c + j] - center) >= lo;
            found = found || ok;
            cl += ok ? 0 : 1;
        }
        if (c == kDeferChunks - 1) {
            *defer = *defer || !found;
            break;
        }
        if (!__any(!found)) break;
    }
    found = false;
#pragma unroll
    for (int c = 0; c < kDeferChunks; ++c) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const bool ok = (v[NP - 1 - (4 * c + j)] - center) <= hi;
            found = found || ok;
            ch += ok ? 0 : 1;
        }
        if (c == kDeferChunks - 1) {
            *defer = *defer || !found;
            break;
        }
        if (!__any(!found)) break;
    }
    cl_out = cl;
    ch_out = ch;
}

// x / n for a sample count n = 1 .. 64 without the IEEE division sequence (v_div_scale x 2, v_rcp_f64 = 4 issue slots, 5 FMAs, v_div_fmas,
// v_div_fixup: ~14 slots; an iteration divides twice and the result once more).  With r = RN(1 / n):
//     q0 = RN(x r),   rem = x - q0 n  (one FMA; exact: q0 is within 2 ulp of x / n, so rem is a multiple of ulp(q0) / 2 below 2^8 ulp),
//     q  = RN(q0 + rem r)
// and q0 + rem r differs from x / n by rem (r - 1 / n), less than 2^-45 ulp(q).  That can only change the rounding if x / n lies
// within 2^-45 ulp of a rounding boundary (k + 1/2) ulp(q); but n (k + 1/2 + eps) ulp(q) = x is a multiple of ulp(x) >= ulp(q), so
// n eps is a multiple of 1/2 and eps = 0 or |eps| >= 1 / (2 n) = 2^-7 -- and eps = 0 (an exact tie) needs 2 ulp(x) / ulp(q) to divide
// n, which only a power of two n allows, where the division is exact anyway.  So q = RN(x / n), for every finite x (no overflow:
// a sum of 64 squares of f32 values stays below 2^262).
// The reciprocals live in ONE register per wave: lane k holds RN(1 / k) (lane 0: RN(1 / 64)), loaded once from a constant table,
// and a lane reads the entry of ITS count with two ds_bpermute_b32 (no LDS allocation, no barrier: round 2's LDS table was slower
// than the divisions because of the barrier that filled it).
struct RecipTable {
    double r[64];
    constexpr RecipTable() : r() {
        for (int k = 0; k < 64; ++k) r[k] = 1.0 / (double)(k ? k : 64);  // constant-folded: correctly rounded
    }
};
__device__ const RecipTable kRecip{};

__device__ __forceinline__ double recip_of_count(double table, int n) {  // n = 1 .. 64
    const int idx = (n & 63) << 2;
    const long long t = __double_as_longlong(table);
    const unsigned lo = (unsigned)__builtin_amdgcn_ds_bpermute(idx, (int)(unsigned)t);
    const unsigned hi = (unsigned)__builtin_amdgcn_ds_bpermute(idx, (int)(unsigned)(t >> 32));
    return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}
template <int NP>
__device__ __forceinline__ double div_by_count(double x, double nn, int n, double table) {
    if constexpr (NP <= 64) {
        const double r = recip_of_count(table, n);
        const double q0 = x * r;
        const double rem = __builtin_fma(-q0, nn, x);
        return __builtin_fma(rem, r, q0);
    } else {
        return x / nn;
    }
}

// sqrt(v) for the iteration's sigma, which only its f32 rounding is used of.  v_rsq_f64 is good to ~2^-27; g = v y, one residual
// step g + (v - g^2) y / 2 brings it to ~2^-52 -- a few ulp(f64) short of the correctly rounded root, which the compiler's 15-
// instruction expansion (scaling, three refinement pairs, class fix-up) delivers.  That is the same order as the fast engine's
// running-sum variance itself (a few ulp(f64) from the two-pass value, see above), 28 binary orders below the f32 the result is
// rounded to: the f32 sigma differs from the oracle's with probability ~1e-8 per pixel, as before.  No scaling: a variance of f32
// samples lies between 2^-298 and 2^+262 or is 0 (-> 0: the caller's max with 1e-10 takes over).  (AB_STACK_IEEE_SQRT: the library call.)
__device__ __forceinline__ double sqrt_for_sigma(double v) {
#ifdef AB_STACK_IEEE_SQRT
    return sqrt(v);
#else
    const double y = __builtin_amdgcn_rsq(v);
    const double g = v * y;
    const double e = __builtin_fma(-g, g, v);
    const double r = __builtin_fma(e, 0.5 * y, g);
    return v > 0.0 ? r : 0.0;
#endif
}

// Per-lane state of the fast engine between the median/MAD clip and the iterations on running sums.
struct FastState {
    int a, b, len;
    uint32_t rej;
    float last_center;
    bool active, defer;
};

// ---- E/Q pass + iterations >= 1 + result, on moments about c0 (RAW: c0 = 0, one f64 subtraction per sample less) ----
template <int NP, int STAGE, bool DEFER, bool SKIP, bool RAW>
__device__ __forceinline__ ClipResult clip_fast_tail(float (&v)[NP], int n, float med, float sigma_low, float sigma_high,
                                                     uint32_t max_iter, FastState s, int top, int skip_hi) {
    int a = s.a, b = s.b, len = s.len;
    uint32_t rej = s.rej;
    float last_center = s.last_center;
    bool active = s.active, defer = s.defer;
    const float c0 = RAW ? 0.0f : med;
    const double c0d = (double)c0;
    double e_rem = 0.0, q_rem = 0.0;
    const double rtab = NP <= 64 ? kRecip.r[threadIdx.x & 63] : 0.0;

    // ---- one pass over the survivors: E1 = sum e_i, Q1 = sum e_i^2 with e_i = x_i - c0 (f64) ----
    // (masked only in the 4-register chunks the interval ends can reach -- the mask is applied to the f32 sample, one select; the
    // rest is 3 (RAW) or 4 instructions per sample: cvt, [sub,] add, fma)
    double E1 = 0.0, Q1 = 0.0;
    {
        constexpr int CH = NP >= 4 ? 4 : NP;
        int a_hi = 0, b_lo = NP - 1;
        if constexpr (!DEFER) {
            a_hi = wave_max_i32<NP>(a);
            b_lo = wave_min_i32<NP>(b);
        }
#pragma unroll
        for (int c = 0; c < NP / CH; ++c) {
            if constexpr (SKIP) {
                if (CH * c > top) continue;  // pads only: every lane would add e = c0 - c0
            }
            bool interior;  // wave-uniform: every lane keeps all CH samples of this chunk
            if constexpr (DEFER) {
                // the fast pass shaves at most kDeferChunks chunks off either end (a lane that wanted more is deferred and its
                // sums are never used): only those chunks can hold an interval end -- one ballot each instead of two wave reductions
                if (c >= kDeferChunks && c < NP / CH - kDeferChunks)
                    interior = true;
                else
                    interior = !__any(!defer && (a > CH * c || b < CH * c + CH - 1));
            } else {
                interior = (CH * c >= a_hi) && (CH * c + CH - 1 <= b_lo);
            }
            if (interior) {
#pragma unroll
                for (int j = 0; j < CH; ++j) {
                    const double e = RAW ? (double)v[CH * c + j] : (double)v[CH * c + j] - c0d;
                    E1 += e;
                    Q1 = __builtin_fma(e, e, Q1);
                }
            } else {
#pragma unroll
                for (int j = 0; j < CH; ++j) {
                    const int i = CH * c + j;
                    const bool in = (i >= a) && (i <= b);
                    float xm = in ? v[i] : c0;
                    asm volatile("" : "+v"(xm));  // select the f32 sample (or the compiler selects the two halves of the f64)
                    const double e = RAW ? (double)xm : (double)xm - c0d;
                    E1 += e;
                    Q1 = __builtin_fma(e, e, Q1);
                }
            }
        }
    }

    if constexpr (STAGE == 5) {
        ClipResult r;
        r.value = (float)(E1 + Q1);
        r.sum = 0.0;
        r.len = len;
        r.rej = rej;
        return r;
    }
    // ---- iterations >= 1: mean / sigma from the running sums (combine.rs:50-82) ----
    for (uint32_t it = 1; it < max_iter; ++it) {
        if (!__any(active)) break;
        launder<NP>(v);  // stop LICM from hoisting 64 f32->f64 conversions out of this loop
        const double nn = (double)len;
        // sum x_i = n c0 + sum e_i: exact whenever the direct f64 sum is (one rounding otherwise)
        const int nd = len > 0 ? len : 1;  // (len == 0 only on lanes that are no longer active: their mean is never used)
        const double sum = RAW ? E1 - e_rem : __builtin_fma(nn, c0d, E1 - e_rem);
        const double mean = div_by_count<NP>(sum, (double)nd, nd, rtab);
        const double dlt = RAW ? mean : mean - c0d;
        double ss = (Q1 - q_rem) - nn * (dlt * dlt);
        ss = ss > 0.0 ? ss : 0.0;
        const int n1 = len > 1 ? len - 1 : 1;
        const double variance = div_by_count<NP>(ss, (double)n1, n1, rtab);  // ss / max(n - 1, 1)
        float center = (float)mean;
        float sigma = (float)fmax(sqrt_for_sigma(variance), 1e-10);
        if constexpr (STAGE == 6) {  // ablation: iteration 1's mean / sigma only
            ClipResult r;
            r.value = center + sigma;
            r.sum = 0.0;
            r.len = len;
            r.rej = rej;
            return r;
        }

        const bool go = active && (len >= 2);
        if (go) last_center = center;
        const float lo = -sigma_low * sigma, hi = sigma_high * sigma;
        int cl, ch;
        clip_ends<NP, true, DEFER, SKIP, RAW>(v, go, a, b, center, lo, hi, c0d, c0, cl, ch, e_rem, q_rem, &defer, skip_hi);
        if constexpr (STAGE == 7) {  // ablation: + iteration 1's end walk
            ClipResult r;
            r.value = center + sigma + (float)(cl + ch) + (float)(e_rem + q_rem);
            r.sum = 0.0;
            r.len = len;
            r.rej = rej;
            return r;
        }
        const int removed = (cl + ch > len) ? len : (cl + ch);
        if (go) {
            rej += (uint32_t)removed;
            len -= removed;
            if (len > 0) {
                a += cl;
                b -= ch;
            } else {
                a = 1;
                b = 0;
            }
        }
        active = go && (removed != 0) && !defer;
    }

    const double S = RAW ? E1 - e_rem : __builtin_fma((double)len, c0d, E1 - e_rem);
    // (outside the per-lane branches below: the table look-up is a cross-lane read, and a lane that sits out a branch returns 0)
    const int nl = len > 0 ? len : 1;
    const float mean_f = (float)div_by_count<NP>(S, (double)nl, nl, rtab);
    ClipResult r;
    if (n == 0) {
        r.value = 0.0f;
    } else if (n == 1) {
        r.value = med;
    } else if (len == 0) {
        r.value = __builtin_isfinite(last_center) ? last_center : 0.0f;
    } else {
        r.value = mean_f;
    }
    r.sum = (len > 0) ? S : 0.0;
    r.len = len > 0 ? len : 0;
    r.rej = rej;
    r.defer = defer;
    return r;
}

template <int NP, int STAGE = 99, bool DEFER = false, bool SKIP = false>
__device__ __forceinline__ ClipResult clip_fast(float (&v)[NP], int n, float med, float mad, float sigma_low,
                                                float sigma_high, uint32_t max_iter) {
    bool defer = false;
    const int top = SKIP ? wave_max_i32<NP>(n - 1) : NP - 1;      // no lane's interval reaches past this slot
    const int skip_hi = SKIP ? (NP - 1 - top) / (NP >= 4 ? 4 : NP) : 0;  // whole clip_ends chunks above it
    const float sigma = (float)fmax((double)mad * kMadToSigma, 1e-10);
    FastState s;
    s.a = 0;
    s.b = n - 1;
    s.len = n;
    s.rej = 0;
    s.last_center = __builtin_nanf("");
    s.active = (n >= 2);

    // ---- iteration 0: clip about the median with the MAD sigma (combine.rs:37-48,63-82) ----
    if (max_iter >= 1) {
        const bool go = s.active;
        if (go) s.last_center = med;
        const float lo = -sigma_low * sigma, hi = sigma_high * sigma;
        int cl, ch;
        double unused_e = 0.0, unused_q = 0.0;
#ifndef AB_STACK_NO_FULL0
        // The common case of the fast pass -- every lane holds all NP samples -- needs none of the interval tests: go is true,
        // every examined position lies inside [0, NP - 1], and the rejected samples of a sorted end are a prefix of it.
        if (DEFER && NP >= 8 && __all(n == NP)) {
            clip_first_full<NP>(v, med, lo, hi, cl, ch, &defer);
        } else
#endif
        clip_ends<NP, false, DEFER, SKIP>(v, go, s.a, s.b, med, lo, hi, 0.0, 0.0f, cl, ch, unused_e, unused_q, &defer, skip_hi);
        const int removed = (cl + ch > s.len) ? s.len : (cl + ch);
        if (go) {
            s.rej += (uint32_t)removed;
            s.len -= removed;
            if (s.len > 0) {
                s.a += cl;
                s.b -= ch;
            } else {
                s.a = 1;
                s.b = 0;
            }
        }
        s.active = go && (removed != 0) && !defer;
    }
    s.defer = defer;

    if constexpr (STAGE == 4) {
        ClipResult r;
        r.value = (float)(s.a + s.b + s.len) + med;
        r.sum = 0.0;
        r.len = s.len;
        r.rej = s.rej;
        return r;
    }
    // The running sums are moments about c0: E = sum (x - c0), Q = sum (x - c0)^2 (f64, ascending), and iteration k gets
    // mean = (n c0 + E) / n and sum (x - mean)^2 = Q - n (mean - c0)^2.  RAW MOMENTS (c0 = 0): the conversion f32 -> f64 IS the
    // deviation, one f64 subtraction per sample less, and Q - n mean^2 cancels log2(mean^2 / variance) bits of an f64.  The wave
    // decides: raw moments when every lane has |median| <= 1024 sigma (the relative error of the variance is then below
    // 64 * 2^-53 * 2^20 = 7e-9 in the worst case, ~1e-9 typically, a thousandth of what one f32 ulp of sigma means), moments
    // about the median otherwise (a flat field at 30 000 +- 5, a saturated core).  Two instances of the tail, one branch.
#if defined(AB_STACK_RAW_MOMENTS)
    const bool raw = true;
#elif defined(AB_STACK_CENTRED_MOMENTS)
    const bool raw = false;
#else
    const bool raw = __all(__builtin_fabsf(med) <= 1024.0f * sigma);
#endif
    if (raw) return clip_fast_tail<NP, STAGE, DEFER, SKIP, true>(v, n, med, sigma_low, sigma_high, max_iter, s, top, skip_hi);
    return clip_fast_tail<NP, STAGE, DEFER, SKIP, false>(v, n, med, sigma_low, sigma_high, max_iter, s, top, skip_hi);
}


// STAGE < 99 cuts the kernel short for the ablation bench (tools/stack_ablate.hip):
//   1 = loads only, 2 = + pads + sort, 3 = + median/MAD, 99 = everything (the product).
//
// Two-pass mode (MODE).  A wave walks the clipped ends of the sorted samples in lock step, so ONE lane with many
// rejected samples -- a pixel inside a star profile, on a frame border -- makes all 64 lanes walk with it; on
// registered frames more than half of the waves contain such a lane and the end walk of iteration 1 alone cost
// 0.22 ms of 1.45.  kFastPass therefore only ever looks at the outermost 4 samples of each end; a lane that would
// need more appends its pixel to one of kDeferSlots lists and writes nothing.  kGeneralPass re-runs the complete
// algorithm for exactly those pixels (a few percent), densely packed into waves.  kPlain is the single-pass kernel
// (partial frame sets, ragged strides, the exact engine).  All three produce bit-identical pixels.
// NREAL (<= NP): a frame-count CLASS of a padded stack -- slots NREAL .. NP - 1 are +inf pads known at compile time: their loads are
// not issued and the sorting network runs as SortNet<NP>::sort_fused_n<NREAL> (operations on pad wires vanish).  129 .. 256 frames
// in classes of 32: 200 frames sort 224 wires' worth of the 256-wire network instead of all of it.
template <int NP, bool PARTIAL, bool EXACT, int STAGE, bool DIRECT, int MODE, int INPUT = kInNative, int NREAL = NP>
__device__ __forceinline__ void stack_pixel(const StackArgs &args, int64_t g, const bool valid) {
    static_assert(INPUT == kInNative || DIRECT, "raw FITS planes are only read through the DIRECT gather");
    static_assert(NREAL == NP || (DIRECT && INPUT == kInNative && NP > 64 && MODE == kPlain && NREAL < NP && NREAL % 8 == 0), "frame-count classes: the deep direct-gather kernels only");
    const int64_t total = args.rows * args.cols;

    int64_t y = 0, x = g;
    if (!args.contiguous) {
        y = g / args.cols;
        x = g - y * args.cols;
    }

    // ---- gather (combine.rs:170-175): only finite samples take part ----
    // Non-finite samples and the slots past args.n become +inf PADS that sort to the top, so the
    // n finite samples end up on sorted positions [0, n).
    float v[NP];
    float nf = 0.0f;  // fma(x, 0, nf) stays 0 for finite x and turns NaN for inf / NaN
#if defined(AB_STACK_PLAIN_SORT) || defined(AB_STACK_NO_HOOKS)
    constexpr bool kHooked = false;
#else
    constexpr bool kHooked = DIRECT && INPUT == kInNative && NP >= 8 && NP <= 64 && STAGE != 1;
#endif
    if constexpr (DIRECT && INPUT == kInNative && NP <= 64) {
        // All NP frames present, contiguous, < 2^30 px.  The plane pointers come through the scalar cache (s_load_dwordx16 of the
        // kernarg table, 8 pointers per load) and every sample is `global_load_dword v, voffset, s[base:base+1]` with one shared
        // 32-bit byte offset: 64 loads issued back to back and ZERO vector instructions of address work.  (Round 1 broadcast the
        // pointers out of a VGPR with 2 v_readlane per frame into buffer descriptors: 128 quarter-rate VALU instructions per
        // wave, 6 % of a VALU-bound kernel -- 1.27 -> 1.16 ms on the bench stack.)  The tail block clamps g, so no lane is ever
        // out of range.
        const uint32_t boff = (uint32_t)g * 4u;
#pragma unroll
        for (int f = 0; f < NP; ++f) v[f] = *(const float *)((const char *)args.p[f] + boff);
        if constexpr (kHooked) {
            // (tested quarter by quarter from inside the network, below)
        } else if constexpr (MODE == kPlain) {
            constexpr int CH = NP >= 8 ? 8 : NP;
#pragma unroll
            for (int c = 0; c < NP / CH; ++c) {
                int t = args.n_real;
                asm volatile("" : "+s"(t));
                if (CH * c + CH <= t) {
#pragma unroll
                    for (int j = 0; j < CH; ++j) nf = __builtin_fmaf(v[CH * c + j], 0.0f, nf);
                } else if (CH * c < t) {
#pragma unroll
                    for (int j = 0; j < CH; ++j)
                        if (CH * c + j < t) nf = __builtin_fmaf(v[CH * c + j], 0.0f, nf);
                }
            }
        } else {
#pragma unroll
            for (int f = 0; f < NP; ++f) nf = __builtin_fmaf(v[f], 0.0f, nf);
        }
    } else if constexpr (DIRECT) {
        // deeper stacks (128 / 256 samples per lane) and raw FITS planes: the pointer table does not fit the scalar registers
        // All NP frames present, contiguous, < 2^30 px: one vector load fetches the 64 plane pointers
        // (lane f reads p[f] straight from the kernarg segment), v_readlane broadcasts each into an
        // SGPR pair, and every sample load is `global_load_dword v, voffset, s[base]` -- no per-frame
        // scalar loads, no 64-bit address arithmetic, 64 loads issued back to back.
        const uint64_t *kp = (const uint64_t *)__builtin_amdgcn_kernarg_segment_ptr();
        constexpr int kPtrRegs = NP > 64 ? NP / 64 : 1;  // lane l of register r holds p[64 r + l]
        uint32_t plo[kPtrRegs], phi[kPtrRegs];
#pragma unroll
        for (int r = 0; r < kPtrRegs; ++r) {
            const uint64_t mine = kp[64 * r + (threadIdx.x & 63)];
            plo[r] = (uint32_t)mine;
            phi[r] = (uint32_t)(mine >> 32);
        }
        constexpr uint32_t kSampleBytes = INPUT == kInI16BE ? 2u : 4u;
        const uint32_t off = (uint32_t)g * kSampleBytes;
        // BITPIX 16 fetches the aligned dword that holds a sample: with an odd pixel count the last sample's dword ends 2 bytes past
        // the data unit, so the descriptor's range is rounded up to whole dwords (a range check on the exact byte count would
        // return 0 for that dword and decode the last pixel as bzero).  The caller's buffer must be readable up to that boundary
        // (include/astroburst_hip.h: ab_stack_sigma_clip_raw).
        const uint32_t plane_bytes = ((uint32_t)total * kSampleBytes + 3u) & ~3u;
#pragma unroll
        for (int f = 0; f < NP; ++f) {
            if constexpr (NREAL < NP) {
                if (f >= NREAL) {  // (compile time: the loop is unrolled) a pad of the frame-count class: no load
                    v[f] = __builtin_inff();
                    continue;
                }
            }
            const uint64_t base = ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)phi[f >> 6], f & 63) << 32) |
                                  (uint32_t)__builtin_amdgcn_readlane((int)plo[f >> 6], f & 63);
            // buffer descriptor in 4 SGPRs -> `buffer_load_dword v, voffset, s[rsrc], 0 offen`
            const __amdgpu_buffer_rsrc_t rsrc =
                __builtin_amdgcn_make_buffer_rsrc((void *)base, 0, (int)plane_bytes, 0x00020000);
            // raw words first (all NP loads in flight, exactly as for native planes); big-endian decode afterwards
            constexpr uint32_t kAlign = INPUT == kInI16BE ? ~3u : ~0u;
            v[f] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, (int)(off & kAlign), 0, 0));
        }
        if constexpr (INPUT == kInF32BE) {  // reader.rs:71-83
#pragma unroll
            for (int f = 0; f < NP; ++f) {
                const float x = __uint_as_float(__builtin_bswap32(__float_as_uint(v[f])));
                v[f] = args.identity ? x : (float)((double)x * args.bscale + args.bzero);
            }
        } else if constexpr (INPUT == kInI16BE) {  // BITPIX 16, reader.rs:56-62: 2 bytes per sample, two lanes per dword
#pragma unroll
            for (int f = 0; f < NP; ++f) {
                const uint32_t w = __float_as_uint(v[f]);
                const uint32_t u = (off & 2u) ? (w >> 16) : (w & 0xffffu);
                const int16_t x = (int16_t)(uint16_t)(((u & 0xffu) << 8) | (u >> 8));
                // identity == 1: v as f32;  identity == 2: BSCALE = 1 and an integer BZERO (unsigned-16 data: 32768) --
                // x + bzero is then an integer below 2^24, so the f32 sum IS the f64 result rounded to f32
                v[f] = args.identity == 1 ? (float)x
                                          : (args.identity == 2 ? (float)x + (float)args.bzero : (float)((double)x * args.bscale + args.bzero));
            }
        }
        if constexpr (MODE == kPlain) {
            // a padded stack: the pads must not trip the non-finite path for every wave.  Wave-uniform chunk tests on an
            // opaque scalar copy of the count (or all NP `f < n_real` become lane masks held across the kernel)
            constexpr int CH = NP >= 8 ? 8 : NP;
#pragma unroll
            for (int c = 0; c < NP / CH; ++c) {
                int t = args.n_real;
                asm volatile("" : "+s"(t));
                if (CH * c + CH <= t) {
#pragma unroll
                    for (int j = 0; j < CH; ++j) nf = __builtin_fmaf(v[CH * c + j], 0.0f, nf);
                } else if (CH * c < t) {
#pragma unroll
                    for (int j = 0; j < CH; ++j)
                        if (CH * c + j < t) nf = __builtin_fmaf(v[CH * c + j], 0.0f, nf);
                }
            }
        } else {
#pragma unroll
            for (int f = 0; f < NP; ++f) nf = __builtin_fmaf(v[f], 0.0f, nf);
        }
    } else {
#pragma unroll
        for (int f = 0; f < NP; ++f) {
            float s = __builtin_inff();
            if (f < args.n) {
                const int64_t off = args.contiguous ? g : (y * args.ld[f] + x);
                s = args.p[f][off];
                nf = __builtin_fmaf(s, 0.0f, nf);
            }
            v[f] = s;
        }
    }
    int n = DIRECT ? (MODE == kPlain ? args.n_real : NP) : args.n;
    if constexpr (!kHooked) {
        if (__any(nf != nf)) {  // rare: some lane of this wave met a non-finite sample
            n = 0;
#pragma unroll
            for (int f = 0; f < NP; ++f) {
                const bool fin = __builtin_isfinite(v[f]);
                v[f] = fin ? v[f] : __builtin_inff();
                n += fin ? 1 : 0;
            }
        }
    }
    if constexpr (STAGE == 1) {
        float t = 0.0f;
#pragma unroll
        for (int f = 0; f < NP; ++f) t += v[f];
        if (valid) args.out[g] = t + (float)n;
        return;
    }

#ifdef AB_STACK_PLAIN_SORT  // (A/B: Batcher's network as written)
    SortNet<NP>::sort(v);
#else
    if constexpr (kHooked) {
        // The finiteness test runs QUARTER BY QUARTER, from inside the network, right before the first operation that reads a
        // sample of that quarter: the wave sorts frames 0 .. 15 while the loads of frames 16 .. 63 are still in flight (one test
        // over all 64 samples in front of the network made every wave wait for its last load before its first exchange).
        constexpr int Q = NP / 4 >= 4 ? NP / 4 : 4, CH = Q < 8 ? Q : 8;
        int lost = 0;  // non-finite samples among the n real ones
        auto hook = [&](auto qc) {
            constexpr int q = decltype(qc)::value;
            float nq = 0.0f;  // fma(x, 0, nq) stays 0 for finite x and turns NaN for inf / NaN
            int t = n;
            if constexpr (MODE == kPlain) {
                // a padded stack: the pads must not trip the non-finite path for every wave.  Wave-uniform chunk tests on an
                // opaque scalar copy of the count (or all the `f < n_real` become lane masks held across the kernel)
                asm volatile("" : "+s"(t));
#pragma unroll
                for (int c = 0; c < Q / CH; ++c) {
                    const int lo = Q * q + CH * c;
                    if (lo + CH <= t) {
#pragma unroll
                        for (int j = 0; j < CH; ++j) nq = __builtin_fmaf(v[lo + j], 0.0f, nq);
                    } else if (lo < t) {
#pragma unroll
                        for (int j = 0; j < CH; ++j)
                            if (lo + j < t) nq = __builtin_fmaf(v[lo + j], 0.0f, nq);
                    }
                }
            } else {
#pragma unroll
                for (int j = 0; j < Q; ++j) nq = __builtin_fmaf(v[Q * q + j], 0.0f, nq);
            }
            if (__any(nq != nq)) {  // rare: some lane of this wave met a non-finite sample in this quarter
                int fin_here = 0;
#pragma unroll
                for (int j = 0; j < Q; ++j) {
                    const bool fin = __builtin_isfinite(v[Q * q + j]);
                    v[Q * q + j] = fin ? v[Q * q + j] : __builtin_inff();
                    fin_here += fin ? 1 : 0;
                }
                const int real = t - Q * q < 0 ? 0 : (t - Q * q > Q ? Q : t - Q * q);  // pads are +inf: never counted as finite
                lost += real - fin_here;
            }
        };
        SortNet<NP>::sort_fused(v, hook);
        n -= lost;
    } else if constexpr (NREAL < NP) {
        SortNet<NP>::template sort_fused_n<NREAL>(v, [](auto) {});
    } else if constexpr (NP >= 8 && NP <= 256) {
        SortNet<NP>::sort_fused(v);
    } else {
        SortNet<NP>::sort(v);
    }
#endif

    if constexpr (STAGE == 2) {
        float t = 0.0f;
#pragma unroll
        for (int f = 0; f < NP; ++f) t += v[f] * (float)(f + 1);
        if (valid) args.out[g] = t;
        return;
    }

    // ---- iteration 0: median / MAD (combine.rs:37-48) ----
    const int m = n >> 1;
    float med = 0.0f, mad = 0.0f;
    if (__all(m == NP / 2)) {
        med_mad_at<NP, NP / 2>(v, med, mad);  // the common case: every lane has all NP samples
    } else if (__all(m == __builtin_amdgcn_readfirstlane(m))) {
        // one position for the whole wave (a padded stack): no loop -- with the samples live around a back edge the
        // 128 / 256-sample kernels lose ~4 ms per 4096^2 launch to register shuffling
        med_mad_dispatch<NP, 0, NP / 2>(v, m, __builtin_amdgcn_readfirstlane(m), med, mad);
    } else {
        // (on a COPY of the samples: launder() makes the vector look rewritten, and if that were v itself the common path
        // above would pay for it with one register move per sample where the three paths join)
        float w[NP];
#pragma unroll
        for (int f = 0; f < NP; ++f) w[f] = v[f];
        unsigned long long todo = __ballot(1);
        while (todo) {
            launder<NP>(w);  // or LICM evaluates all NP/2+1 candidate positions up front and spills
            const int cur = __builtin_amdgcn_readlane(m, (int)__builtin_ctzll(todo));
            med_mad_dispatch<NP, 0, NP / 2>(w, m, cur, med, mad);
            todo &= ~__ballot(m == cur);
        }
    }

    if constexpr (STAGE == 3) {
        if (valid) args.out[g] = med + mad;
        return;
    }
    if constexpr (STAGE == 10) {  // median_combine_row_major (calibration.rs:84-125): [len/2] of the finite samples
        if (valid) args.out[g] = (n == 0) ? 0.0f : med;
        return;
    }

    ClipResult r;
    if constexpr (EXACT)
        r = clip_exact<NP>(v, n, med, mad, args.sigma_low, args.sigma_high, args.max_iter);
    else
        r = clip_fast<NP, STAGE, MODE == kFastPass, MODE == kPlain>(v, n, med, mad, args.sigma_low, args.sigma_high, args.max_iter);

    // (Tried: handing a wave over WHOLE to the general pass when some pixel holds more than 8 exact zeros below a non-zero median
    // -- the border bands of registered frames -- so that it is re-gathered coalesced instead of pixel by pixel: deferred pixels
    // 0.80 % -> 0.09 %, launch time unchanged, 1.11 ms either way: the wasted sort and the unbounded end walk cost what the
    // scattered gathers did.)
    // (Tried: a wave with >= 4 deferring lanes -- the edge bands -- runs the general engine in place on the samples it still holds
    // in registers: nothing is deferred any more, but the stack goes from 1.12 to 1.36 ms.  The second engine instance slows
    // EVERY wave (1.07 vs 0.99 ms with the borders cropped away), and walking 60 zeros in 16 chunks costs those waves more than
    // the general pass's gathers.)
    uint32_t rej = r.rej;
    const bool defer = MODE == kFastPass && valid && r.defer;
    if (valid && !defer) {
        if constexpr (PARTIAL) {
            args.out_sum[g] = r.sum;
            args.out_cnt[g] = (uint32_t)r.len;
        } else {
            args.out[g] = r.value;
        }
    } else {
        rej = 0;
    }
    if constexpr (MODE == kFastPass) {  // hand the pixel to the general pass: one atomic per wave, 2048 counters
        const unsigned long long m = __ballot(defer);
        if (m) {
            const int lane = threadIdx.x & 63, leader = (int)__builtin_ctzll(m);
            // slot: consecutive waves go to consecutive lists (each run of 2048 waves fills every list once, which
            // bounds a list at defer_cap), rotated per run -- frame borders and star columns recur with the row
            // period and would otherwise pile onto a few lists
            const unsigned int w = blockIdx.x * 4u + (threadIdx.x >> 6);
            const unsigned int slot = (w + (w / kDeferSlots) * 977u) & (kDeferSlots - 1);
            unsigned int base = 0;
            if (lane == leader) base = atomicAdd(&args.defer_count[slot], (unsigned int)__builtin_popcountll(m));
            base = __shfl(base, leader, 64);
            if (defer) args.defer_list[(size_t)slot * args.defer_cap + base + (unsigned int)__builtin_popcountll(m & ((1ull << lane) - 1ull))] = (int)g;
        }
    }

    // rejection count: wavefront shuffle reduction, then ONE atomic per wave spread over kRejSlots
    // counters (summed by the host).  All 262 144 waves of a 4096^2 stack adding to a single
    // address serialise at ~12 ns per atomic = 3 ms, more than the whole kernel; a workgroup-level
    // LDS reduction would need a barrier that makes the 4 waves of a group wait for the slowest.
    rej = (uint32_t)wave_reduce_i32<0>((int)rej);
    if ((threadIdx.x & 63) == 0 && rej != 0)
        atomicAdd(&args.rejected[(blockIdx.x * 4u + (threadIdx.x >> 6)) & (kRejSlots - 1)], (unsigned long long)rej);
}

template <int NP, bool PARTIAL, bool EXACT, int STAGE = 99, bool DIRECT = false, int MODE = kPlain, int INPUT = kInNative, int NREAL = NP>
__global__ __launch_bounds__(256, (EXACT || NP > 128) ? 1 : (NP > 64 ? 2 : AB_STACK_WAVES_PER_SIMD)) void stack_sigma_clip_kernel(const StackArgs args) {
    if constexpr (MODE == kGeneralPass) {
        // ONE wave per workgroup, kGenWaves workgroups per list (launched with 64 threads).  With one 4-wave workgroup per list
        // (round 2) a list of ~65 pixels kept one wave busy and three wave slots empty until it finished: 2048 workgroups went
        // through the chip three at a time per CU and the pass took 60 us for 2100 waves' worth of work.
        const unsigned int slot = blockIdx.x / kGenWaves, sub = blockIdx.x % kGenWaves;
        const unsigned int cnt = args.defer_count[slot];
        const int *list = args.defer_list + (size_t)slot * args.defer_cap;
        for (unsigned int base = sub * 64u; base < cnt; base += 64u * kGenWaves) {
            const unsigned int k = base + threadIdx.x;
            const bool valid = k < cnt;
            stack_pixel<NP, PARTIAL, EXACT, STAGE, DIRECT, MODE, INPUT, NREAL>(args, (int64_t)list[valid ? k : cnt - 1], valid);
        }
        // the last of the list's workgroups to get here leaves the list empty for the next launch.  (No fence: each workgroup's
        // own read of the count has returned before its ticket is taken -- the loop bound depends on it -- and a device-scope
        // release here is an L2 write-back per workgroup on this chip: 4096 of them made the pass 2.4x slower.)
        if (!args.keep_counts && threadIdx.x == 0) {
            if (atomicAdd(&args.defer_ticket[slot], 1u) == kGenWaves - 1) {
                args.defer_ticket[slot] = 0;
                args.defer_count[slot] = 0;
            }
        }
    } else {
        const int64_t total = args.rows * args.cols;
        int64_t g = (int64_t)blockIdx.x * 256 + threadIdx.x;
        const bool valid = g < total;
        if (!valid) g = total - 1;
        stack_pixel<NP, PARTIAL, EXACT, STAGE, DIRECT, MODE, INPUT, NREAL>(args, g, valid);
    }
}

// single frame: sigma_clip_combine returns the value itself, or 0 if it is not finite
__global__ __launch_bounds__(256) void stack_single_kernel(const float *src, int64_t ld, int64_t rows, int64_t cols,
                                                           float *out, double *out_sum, uint32_t *out_cnt) {
    const int64_t total = rows * cols;
    const int64_t g = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (g >= total) return;
    const int64_t y = g / cols, x = g - y * cols;
    const float s = src[y * ld + x];
    const bool fin = __builtin_isfinite(s);
    if (out) out[g] = fin ? s : 0.0f;
    if (out_sum) {
        out_sum[g] = fin ? (double)s : 0.0;
        out_cnt[g] = fin ? 1u : 0u;
    }
}

__global__ void finalize_partial_kernel(const double *sum, const uint32_t *cnt, int64_t n, float *out) {
    const int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= n) return;
    const uint32_t c = cnt[g];
    out[g] = c ? (float)(sum[g] / (double)c) : 0.0f;
}

// raw big-endian planes: two-pass DIRECT kernels only (the caller has checked n == NP, contiguous, < 2^30 px)
template <int NP, int INPUT>
void launch_raw(ab_ctx *ctx, const StackArgs &args, dim3 grid, dim3 block) {
    hipLaunchKernelGGL((stack_sigma_clip_kernel<NP, false, false, 99, true, kFastPass, INPUT>), grid, block, 0, ctx->stream, args);
    hipLaunchKernelGGL((stack_sigma_clip_kernel<NP, false, false, 99, true, kGeneralPass, INPUT>), dim3(kDeferSlots * kGenWaves), dim3(64), 0, ctx->stream, args);
}

template <int NP, bool PARTIAL, bool EXACT, int STAGE>
void launch_np(ab_ctx *ctx, const StackArgs &args, dim3 grid, dim3 block) {
    // DIRECT gather: no absent-frame slots, one row stride, byte offsets fit 32 bits
    const bool direct = args.n == NP && args.contiguous && args.rows * args.cols < (int64_t(1) << 30);
    if constexpr (!EXACT && STAGE == 99 && NP >= 8) {
        if (direct && args.defer_list) {  // two-pass mode: fast pass over every pixel, general pass over the deferred ones
            hipLaunchKernelGGL((stack_sigma_clip_kernel<NP, PARTIAL, EXACT, STAGE, true, kFastPass>), grid, block, 0, ctx->stream, args);
            hipLaunchKernelGGL((stack_sigma_clip_kernel<NP, PARTIAL, EXACT, STAGE, true, kGeneralPass>), dim3(kDeferSlots * kGenWaves), dim3(64), 0,
                               ctx->stream, args);
            return;
        }
    }
    if (direct)
        hipLaunchKernelGGL((stack_sigma_clip_kernel<NP, PARTIAL, EXACT, STAGE, true>), grid, block, 0, ctx->stream, args);
    else
        hipLaunchKernelGGL((stack_sigma_clip_kernel<NP, PARTIAL, EXACT, STAGE, false>), grid, block, 0, ctx->stream, args);
}

template <bool PARTIAL, bool EXACT, int STAGE = 99>
int launch_stack(ab_ctx *ctx, const StackArgs &args, int np) {
    const int64_t total = args.rows * args.cols;
    const dim3 grid((unsigned)((total + 255) / 256)), block(256);
    switch (np) {
        case 2: launch_np<2, PARTIAL, EXACT, STAGE>(ctx, args, grid, block); break;
        case 4: launch_np<4, PARTIAL, EXACT, STAGE>(ctx, args, grid, block); break;
        case 8: launch_np<8, PARTIAL, EXACT, STAGE>(ctx, args, grid, block); break;
        case 16: launch_np<16, PARTIAL, EXACT, STAGE>(ctx, args, grid, block); break;
        case 32: launch_np<32, PARTIAL, EXACT, STAGE>(ctx, args, grid, block); break;
        case 64: launch_np<64, PARTIAL, EXACT, STAGE>(ctx, args, grid, block); break;
        default: return ab_set_error(ctx, AB_ERR_INVALID, "internal: bad padded frame count %d", np);
    }
    AB_HIP(ctx, hipGetLastError());
    return AB_OK;
}

}  // namespace

static int read_rejected(ab_ctx *ctx, uint64_t *out) {
    void *pin = nullptr;
    AB_TRY(ab_pinned(ctx, kRejSlots * sizeof(unsigned long long), &pin));
    AB_HIP(ctx, hipMemcpyAsync(pin, ctx->counters, kRejSlots * sizeof(unsigned long long), hipMemcpyDeviceToHost, ctx->stream));
    AB_HIP(ctx, hipStreamSynchronize(ctx->stream));
    uint64_t tot = 0;
    for (int i = 0; i < kRejSlots; ++i) tot += ((const unsigned long long *)pin)[i];
    *out = tot;
    return AB_OK;
}

// deferred-pixel lists of the two-pass mode: slot = wave index & 2047 (rotated), so a slot holds at most
// ceil(waves / 2048) waves' worth of pixels
static int setup_defer(ab_ctx *ctx, StackArgs *args, int64_t total) {
    const int64_t waves = ((total + 255) / 256) * 4;
    const unsigned int cap = (unsigned int)(((waves + kDeferSlots - 1) / kDeferSlots) * 64);
    char *ws = nullptr;
    const void *before = ctx->ws[AB_WS_STACK_DEFER];
    AB_TRY(ab_workspace(ctx, AB_WS_STACK_DEFER, (size_t)2 * kDeferSlots * sizeof(unsigned int) + (size_t)kDeferSlots * cap * sizeof(int), (void **)&ws));
    args->defer_count = (unsigned int *)ws;
    args->defer_ticket = args->defer_count + kDeferSlots;
    args->defer_list = (int *)(ws + (size_t)2 * kDeferSlots * sizeof(unsigned int));
    args->defer_cap = cap;
    // the general pass leaves every counter at zero again, so only a fresh workspace needs clearing
    args->keep_counts = ab_env("AB_TRACE") ? 1 : 0;
    if (ws != before || args->keep_counts) AB_HIP(ctx, hipMemsetAsync(args->defer_count, 0, 2 * kDeferSlots * sizeof(unsigned int), ctx->stream));
    return AB_OK;
}

// Shared implementation.  dplanes: device pointers + row strides of the n frames.
int ab_stack_device(ab_ctx *ctx, const float *const *dplanes, const int64_t *ld, size_t n, int64_t rows,
                    int64_t cols, const ab_stack_config *cfg, float *out_dev, double *out_sum_dev,
                    uint32_t *out_cnt_dev, uint64_t *out_rejected, bool median_only) {
    AB_CHECK(ctx, n >= 1, "No images to stack");
    AB_CHECK(ctx, n <= ((size_t)1 << 24), "stack of %zu frames (at most 2^24 per call)", n);
    AB_CHECK(ctx, rows > 0 && cols > 0, "stack output has a zero dimension");
    AB_HIP(ctx, hipSetDevice(ctx->device));
    const int64_t total = rows * cols;
    const bool partial = out_sum_dev != nullptr;

    const bool first_chunk = !ctx->stack_keep_counters;  // (sharded.hip stacks in row chunks: the later ones add to the first one's counts and events)
    if (first_chunk) AB_HIP(ctx, hipMemsetAsync(ctx->counters, 0, kRejSlots * sizeof(unsigned long long), ctx->stream));
    // 65 .. 128 contiguous frames, plain full-image stack: still one lane per pixel, 128 samples in registers (one wave per
    // SIMD).  Everything else beyond 64 frames -- ragged strides, partial sums, the median combine, the exact engine, more
    // than 128 frames -- takes one wave per pixel (stack_wide.hip).
    bool contig_all = total < (int64_t(1) << 30);
    for (size_t i = 0; i < n && contig_all; ++i) contig_all = ld[i] == cols;
    const bool reg128 = n > 64 && n <= 256 && contig_all && !partial && !ctx->stack_exact;  // (and 129 .. 256; median_combine too)
    // Round 6: 129 .. 256 contiguous frames take two lanes per pixel, 128 samples each, two waves per SIMD -- stack_duo.hip's fast pass
    // for the default engine's stack and the median combine, stack_pair.hip's oracle-arithmetic kernel for AB_STACK_EXACT=1 (15.5 ms
    // for 256 x 4096^2 where the wave-per-pixel kernel took 151).  AB_STACK_NO_DUO=1 (developer build) keeps round 5's routes.
    static const bool no_duo = ab_dev_env("AB_STACK_NO_DUO") != nullptr;
    const bool duo = n > 128 && n <= 256 && contig_all && !partial && !no_duo;
    if (n > 64 && (!reg128 || duo)) {  // deeper than one lane's registers
        for (hipEvent_t &e : ctx->stack_ev)
            if (!e) AB_HIP(ctx, hipEventCreate(&e));
        ctx->stack_ev_valid = false;
        if (first_chunk) AB_HIP(ctx, hipEventRecord(ctx->stack_ev[0], ctx->stream));
        // 257 .. 512 contiguous frames, plain full-image stack or median combine: two lanes per pixel (stack_pair.hip), bit-identical
        // to the wave-per-pixel kernel (AB_STACK_NO_PAIR=1 keeps that one); everything else: one wave per pixel (stack_wide.hip)
        // 513 .. 4096: the wave-per-pixel kernel with 16 / 32 / 64 registers per lane; beyond: one workgroup per pixel, samples
        // in global scratch (stack_deep.hip; a context created under AB_STACK_DEEP_FROM=k sends every stack of more than k >= 64 frames there: the tests do)
        static const bool no_pair = ab_dev_env("AB_STACK_NO_PAIR") != nullptr;
        static const bool no_octo = ab_dev_env("AB_STACK_NO_OCTO") != nullptr;  // (developer A/B: 513 .. 1024 frames one wave per pixel as before)
        if (n > (size_t)ctx->stack_deep_from)
            AB_TRY(ab_stack_deep_device(ctx, dplanes, ld, n, rows, cols, cfg, out_dev, out_sum_dev, out_cnt_dev, median_only));
        else if (duo || (n > 256 && n <= 512 && contig_all && !partial && !no_pair) ||
                 (n > 512 && n <= 1024 && contig_all && !partial && !ctx->stack_exact && !no_octo))  // (eight lanes per pixel: stack_quad.hip)
            AB_TRY(ab_stack_pair_device(ctx, dplanes, n, rows, cols, cfg, out_dev, median_only));
        else
            AB_TRY(ab_stack_wide_device(ctx, dplanes, ld, n, rows, cols, cfg, out_dev, out_sum_dev, out_cnt_dev, median_only));
        AB_HIP(ctx, hipEventRecord(ctx->stack_ev[1], ctx->stream));
        ctx->stack_ev_valid = true;
        if (out_rejected) AB_TRY(read_rejected(ctx, out_rejected));
        return AB_OK;
    }
    if (n == 1) {
        const dim3 grid((unsigned)((total + 255) / 256)), block(256);
        hipLaunchKernelGGL(stack_single_kernel, grid, block, 0, ctx->stream, dplanes[0], ld[0], rows, cols,
                           partial ? nullptr : out_dev, out_sum_dev, out_cnt_dev);
        AB_HIP(ctx, hipGetLastError());
    } else {
        StackArgs args;
        memset(&args, 0, sizeof args);
        int contiguous = 1;
        for (size_t i = 0; i < n; ++i) {
            args.p[i] = dplanes[i];
            if (i < (size_t)kMaxStrided) args.ld[i] = ld[i];
            if (ld[i] != cols) contiguous = 0;
        }
        args.n = (int)n;
    args.n_real = (int)n;
        args.contiguous = contiguous;
        args.rows = rows;
        args.cols = cols;
        args.sigma_low = cfg->sigma_low;
        args.sigma_high = cfg->sigma_high;
        args.max_iter = cfg->max_iterations;
        args.out = out_dev;
        args.out_sum = out_sum_dev;
        args.out_cnt = out_cnt_dev;
        args.rejected = ctx->counters;
        int np = 2;
        while (np < (int)n) np <<= 1;
        bool padded = false;
        // A frame count between two powers of two would run the padded kernel, whose per-frame `f < n` predicates make it
        // 2-3x slower (37 frames: 2.7 ms against 1.3 ms for 64).  With contiguous planes the missing frames are aliased to
        // one plane of +inf instead: a non-finite sample is exactly what the algorithm ignores (combine.rs:170-175), the
        // kernel sees n == NP again (direct gather, two-pass mode), and the pad reads stay in L2.
        if ((int)n < np && np >= 8 && contiguous && total < (int64_t(1) << 30)) {
            float *inf_plane = nullptr;
            const void *before = ctx->ws[AB_WS_STACK_INF];
            const size_t had = ctx->ws_bytes[AB_WS_STACK_INF];
            AB_TRY(ab_workspace(ctx, AB_WS_STACK_INF, (size_t)total * sizeof(float), (void **)&inf_plane));
            if (inf_plane != before || had < (size_t)total * sizeof(float))
                AB_HIP(ctx, hipMemsetD32Async((hipDeviceptr_t)inf_plane, 0x7f800000, ctx->ws_bytes[AB_WS_STACK_INF] / sizeof(float), ctx->stream));
            for (int f = (int)n; f < np; ++f) {
                args.p[f] = inf_plane;
                if (f < kMaxStrided) args.ld[f] = cols;
            }
            args.n = np;
            n = (size_t)np;
            padded = true;
        }
        // (the fast pass of the two-pass mode only looks at sorted positions NP-4 .. NP-1 for the high end: on a padded stack
        // those are pads and every pixel would be deferred, so padded stacks take the single-pass kernel)
        if (!padded && !median_only && !ctx->stack_exact && (int)n == np && np >= 8 && contiguous && total < (int64_t(1) << 30) &&
            !ab_dev_env("AB_STACK_SINGLE_PASS")) {
            AB_TRY(setup_defer(ctx, &args, total));
        }
        for (hipEvent_t &e : ctx->stack_ev)
            if (!e) AB_HIP(ctx, hipEventCreate(&e));
        ctx->stack_ev_valid = false;
        if (first_chunk) AB_HIP(ctx, hipEventRecord(ctx->stack_ev[0], ctx->stream));
        if (np == 256) {  // 129 .. 256 contiguous frames: 256 samples per lane (VGPRs + AGPRs), single pass
#ifndef AB_DEV_ABLATION
            // (round 6: these frame counts take two lanes per pixel -- `duo` above; round 5's one-lane kernels are built into the
            // developer library only, for the A/B under AB_STACK_NO_DUO=1)
            return ab_set_error(ctx, AB_ERR_INVALID, "internal: %d frames reached the one-lane 256-sample route", args.n_real);
#else
            const dim3 grid((unsigned)((total + 255) / 256)), block(256);
            // frame-count classes of 32 (AB_STACK_NO_CLASSES=1: every count pays for 256): the pads' loads and the network's
            // operations on pad wires are gone at compile time
            static const bool no_classes = ab_dev_env("AB_STACK_NO_CLASSES") != nullptr;
            const int cls = no_classes ? 256 : (args.n_real + 31) / 32 * 32;
#define AB_LAUNCH_256(NREAL)                                                                                                               \
    do {                                                                                                                                   \
        if (median_only)                                                                                                                   \
            hipLaunchKernelGGL((stack_sigma_clip_kernel<256, false, false, 10, true, kPlain, kInNative, NREAL>), grid, block, 0, ctx->stream, args); \
        else                                                                                                                               \
            hipLaunchKernelGGL((stack_sigma_clip_kernel<256, false, false, 99, true, kPlain, kInNative, NREAL>), grid, block, 0, ctx->stream, args); \
    } while (0)
            if (cls <= 160) AB_LAUNCH_256(160);
            else if (cls == 192) AB_LAUNCH_256(192);
            else if (cls == 224) AB_LAUNCH_256(224);
            else AB_LAUNCH_256(256);
#undef AB_LAUNCH_256
            AB_HIP(ctx, hipGetLastError());
#endif
        } else if (np == 128) {  // reg128 (checked above): only the direct-gather kernels exist for 128 samples per lane
            const dim3 grid((unsigned)((total + 255) / 256)), block(256);
            if (median_only) {
                hipLaunchKernelGGL((stack_sigma_clip_kernel<128, false, false, 10, true>), grid, block, 0, ctx->stream, args);
            } else if (args.defer_list) {
                hipLaunchKernelGGL((stack_sigma_clip_kernel<128, false, false, 99, true, kFastPass>), grid, block, 0, ctx->stream, args);
                hipLaunchKernelGGL((stack_sigma_clip_kernel<128, false, false, 99, true, kGeneralPass>), dim3(kDeferSlots * kGenWaves), dim3(64), 0, ctx->stream, args);
            } else {
                hipLaunchKernelGGL((stack_sigma_clip_kernel<128, false, false, 99, true>), grid, block, 0, ctx->stream, args);
            }
            AB_HIP(ctx, hipGetLastError());
        } else if (median_only)
            AB_TRY((launch_stack<false, false, 10>(ctx, args, np)));
        else if (ctx->stack_exact)
            AB_TRY(partial ? (launch_stack<true, true>(ctx, args, np)) : (launch_stack<false, true>(ctx, args, np)));
        else
            AB_TRY(partial ? (launch_stack<true, false>(ctx, args, np)) : (launch_stack<false, false>(ctx, args, np)));
        AB_HIP(ctx, hipEventRecord(ctx->stack_ev[1], ctx->stream));
        ctx->stack_ev_valid = true;
    }
    if (ab_env("AB_TRACE") && n > 1) {  // developer aid: how many pixels the fast pass handed to the general pass
        std::vector<unsigned int> cnt(kDeferSlots, 0);
        void *ws = ctx->ws[AB_WS_STACK_DEFER];
        if (ws) {
            AB_HIP(ctx, hipMemcpyAsync(cnt.data(), ws, kDeferSlots * sizeof(unsigned int), hipMemcpyDeviceToHost, ctx->stream));
            AB_HIP(ctx, hipStreamSynchronize(ctx->stream));
            unsigned long long tot = 0, mx = 0;
            for (unsigned int c : cnt) tot += c, mx = c > mx ? c : mx;
            ab_count_fallback(ctx, AB_FB_STACK_GENERAL_PIXELS, tot);
            fprintf(stderr, "[ab_trace] stack: %llu of %lld pixels deferred (%.2f %%), fullest list %llu\n", tot, (long long)total,
                    100.0 * (double)tot / (double)total, mx);
        }
    }
    if (out_rejected) AB_TRY(read_rejected(ctx, out_rejected));
    return AB_OK;
}

extern "C" {

static int stack_planes(ab_ctx *ctx, const ab_plane *planes, size_t n, const ab_stack_config *cfg, ab_plane_mut *out,
                        uint64_t *out_rejected, bool median_only) {
    if (!ctx) return AB_ERR_INVALID;
    AB_CHECK(ctx, planes && n >= 1, "No images to stack");
    AB_CHECK(ctx, cfg && out, "null config or output");
    for (size_t i = 0; i < n; ++i)
        AB_CHECK(ctx, planes[i].rows >= out->rows && planes[i].cols >= out->cols,
                 "frame %zu (%lldx%lld) is smaller than the output (%lldx%lld)", i, (long long)planes[i].rows,
                 (long long)planes[i].cols, (long long)out->rows, (long long)out->cols);
    std::vector<StagedPlane> st(n);
    std::vector<const float *> dp(n);
    std::vector<int64_t> ld(n);
    int rc = AB_OK;
    size_t staged = 0;
    for (; staged < n; ++staged) {
        rc = ab_stage_in(ctx, &planes[staged], &st[staged]);
        if (rc != AB_OK) break;
        dp[staged] = st[staged].dptr;
        ld[staged] = st[staged].cols;
    }
    StagedOut so;
    bool so_open = false;
    if (rc == AB_OK) {
        rc = ab_stage_out_begin(ctx, out, &so);
        so_open = (rc == AB_OK);
    }
    uint64_t rejected = 0;
    if (rc == AB_OK)
        rc = ab_stack_device(ctx, dp.data(), ld.data(), n, out->rows, out->cols, cfg, so.dptr, nullptr, nullptr,
                             out_rejected ? &rejected : nullptr, median_only);
    if (rc == AB_OK) {
        rc = ab_stage_out_finish(ctx, &so);
        so_open = false;
    }
    if (so_open) ab_stage_out_abort(ctx, &so);
    for (size_t i = 0; i < staged; ++i) ab_stage_release(ctx, &st[i]);
    if (rc == AB_OK && out_rejected) *out_rejected = rejected;
    return rc;
}

int ab_stack_sigma_clip(ab_ctx *ctx, const ab_plane *planes, size_t n, const ab_stack_config *cfg, ab_plane_mut *out,
                        uint64_t *out_rejected) try {
    return stack_planes(ctx, planes, n, cfg, out, out_rejected, false);
} AB_CATCH(ctx)

// median_combine_row_major (calibration.rs:84-125): per-pixel [len/2] order statistic of the finite samples
int ab_median_combine(ab_ctx *ctx, const ab_plane *planes, size_t n, ab_plane_mut *out) try {
    const ab_stack_config cfg = {3.0f, 3.0f, 0, 0};
    return stack_planes(ctx, planes, n, &cfg, out, nullptr, true);
} AB_CATCH(ctx)

int ab_stack_sigma_clip_partial(ab_ctx *ctx, const ab_plane *planes, size_t n, const ab_stack_config *cfg, int64_t rows,
                                int64_t cols, double *out_sum_dev, uint32_t *out_cnt_dev, uint64_t *out_rejected) try {
    if (!ctx) return AB_ERR_INVALID;
    AB_CHECK(ctx, planes && n >= 1, "No images to stack");
    AB_CHECK(ctx, cfg && out_sum_dev && out_cnt_dev, "null config or output");
    std::vector<const float *> dp(n);
    std::vector<int64_t> ld(n);
    for (size_t i = 0; i < n; ++i) {
        AB_CHECK(ctx, planes[i].on_device, "partial stacking takes device-resident frames");
        AB_CHECK(ctx, planes[i].rows >= rows && planes[i].cols >= cols, "frame %zu is smaller than the output", i);
        dp[i] = planes[i].data;
        ld[i] = planes[i].cols;
    }
    return ab_stack_device(ctx, dp.data(), ld.data(), n, rows, cols, cfg, nullptr, out_sum_dev, out_cnt_dev, out_rejected,
                           false);
} AB_CATCH(ctx)

// stack_images' per-pixel loop fed straight from FITS data units (decode_pixels fused into the gather)
int ab_stack_sigma_clip_raw(ab_ctx *ctx, const void *const *raw_planes_dev, size_t n, int64_t bitpix, double bscale, double bzero,
                            const ab_stack_config *cfg, ab_plane_mut *out, uint64_t *out_rejected) try {
    if (!ctx) return AB_ERR_INVALID;
    AB_CHECK(ctx, raw_planes_dev && cfg && out && out->data, "null argument");
    AB_CHECK(ctx, out->on_device, "the fused raw stack writes a device plane");
    AB_CHECK(ctx, bitpix == -32 || bitpix == 16, "fused decode handles BITPIX -32 and 16 (got %lld): decode with ab_fits_decode_pixels first",
             (long long)bitpix);
    const int64_t total = out->rows * out->cols;
    if (!(n == 8 || n == 16 || n == 32 || n == 64) || total <= 0 || total >= (int64_t(1) << 30))
        return ab_set_error(ctx, AB_ERR_UNSUPPORTED, "fused raw stack takes 8, 16, 32 or 64 planes of < 2^30 pixels (got %zu x %lld): decode first", n,
                            (long long)total);
    AB_HIP(ctx, hipSetDevice(ctx->device));
    AB_HIP(ctx, hipMemsetAsync(ctx->counters, 0, kRejSlots * sizeof(unsigned long long), ctx->stream));
    StackArgs args;
    memset(&args, 0, sizeof args);
    for (size_t i = 0; i < n; ++i) {
        AB_CHECK(ctx, raw_planes_dev[i] && ((uintptr_t)raw_planes_dev[i] & 3) == 0, "raw plane %zu is null or not 4-byte aligned", i);
        args.p[i] = (const float *)raw_planes_dev[i];
        args.ld[i] = out->cols;
    }
    args.n = (int)n;
    args.n_real = (int)n;
    args.contiguous = 1;
    args.rows = out->rows;
    args.cols = out->cols;
    args.sigma_low = cfg->sigma_low;
    args.sigma_high = cfg->sigma_high;
    args.max_iter = cfg->max_iterations;
    args.out = out->data;
    args.rejected = ctx->counters;
    args.identity = std::fabs(bscale - 1.0) < 1e-15 && std::fabs(bzero) < 1e-15;
    if (!args.identity && bitpix == 16 && bscale == 1.0 && bzero == std::floor(bzero) && std::fabs(bzero) <= 8.0e6) args.identity = 2;
    args.bscale = bscale;
    args.bzero = bzero;
    AB_TRY(setup_defer(ctx, &args, total));
    const dim3 grid((unsigned)((total + 255) / 256)), block(256);
#define AB_RAW_CASE(NPV)                                             \
    case NPV:                                                        \
        if (bitpix == -32)                                           \
            launch_raw<NPV, kInF32BE>(ctx, args, grid, block);       \
        else                                                         \
            launch_raw<NPV, kInI16BE>(ctx, args, grid, block);       \
        break;
    switch ((int)n) {
        AB_RAW_CASE(8)
        AB_RAW_CASE(16)
        AB_RAW_CASE(32)
        AB_RAW_CASE(64)
    }
#undef AB_RAW_CASE
    AB_HIP(ctx, hipGetLastError());
    if (out_rejected) AB_TRY(read_rejected(ctx, out_rejected));
    return AB_OK;
} AB_CATCH(ctx)

int ab_stack_last_kernel_ms(ab_ctx *ctx, float *out_ms) try {
    if (!ctx) return AB_ERR_INVALID;
    AB_CHECK(ctx, out_ms, "null argument");
    AB_CHECK(ctx, ctx->stack_ev_valid, "no multi-frame stack has been launched on this context");
    AB_HIP(ctx, hipEventSynchronize(ctx->stack_ev[1]));
    AB_HIP(ctx, hipEventElapsedTime(out_ms, ctx->stack_ev[0], ctx->stack_ev[1]));
    return AB_OK;
} AB_CATCH(ctx)

int ab_stack_last_rejected(ab_ctx *ctx, uint64_t *out_rejected) try {
    if (!ctx) return AB_ERR_INVALID;
    AB_CHECK(ctx, out_rejected, "null output");
    return read_rejected(ctx, out_rejected);
} AB_CATCH(ctx)

int ab_stack_finalize_partial(ab_ctx *ctx, const double *sum_dev, const uint32_t *cnt_dev, int64_t n, float *out_dev) try {
    if (!ctx) return AB_ERR_INVALID;
    AB_CHECK(ctx, sum_dev && cnt_dev && out_dev && n > 0, "null buffer or empty range");
    AB_HIP(ctx, hipSetDevice(ctx->device));
    hipLaunchKernelGGL(finalize_partial_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, sum_dev,
                       cnt_dev, n, out_dev);
    AB_HIP(ctx, hipGetLastError());
    return AB_OK;
} AB_CATCH(ctx)

}  // extern "C"
