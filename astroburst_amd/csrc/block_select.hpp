// Workgroup-level exact order statistics over a rectangular window of a plane (gfx950).
//
// Used wherever the reference takes a median / MAD of up to 256 x 256 pixels with
// select_nth_unstable: the detection background tiles (core/analysis/star_detection.rs:47-68) and the
// background-extraction grid cells (core/imaging/background.rs:151-190).  All candidate pixels are
// positive finite floats (or absolute deviations), whose IEEE bit patterns are monotone as u32, so
// rank k is found by an 11/11/10-bit radix select: LDS-histogram passes over the window (re-read
// from L2), one 1024-thread workgroup per window.  Exact for every rank -> medians of even counts
// average the two middle order statistics just as math/median.rs does.
//
// Cost model: a pass is one read of the window + one LDS atomic per candidate.  Sky pixels share
// their top 11 bits (sign, exponent, 3 mantissa bits), so a plain atomicAdd per lane would serialise
// ~all 65 536 updates of a tile on ONE LDS address (measured: 110 us per pass, 4.3 ms per frame).
// The top level therefore tallies each wave's dominant bin in a scalar (ModeTally: ballot + popcount, one
// atomic per wave and pass) and only the other lanes issue atomics.  The lower levels (keys spread over
// ~1000 bins) use plain per-lane atomics.
// The top-level histogram is computed once per key mode (`prepare`) and shared by the count and by
// every rank requested from it (the two middle ranks of an even count descend together while they
// stay in the same bin), so a median costs 3 passes instead of 7.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

#include "ab_common.hpp"

namespace absel {

constexpr int kBlock = 1024;

struct Window {
    const float *img;
    int64_t ld;
    int y0, y1, x0, x1;
    float min_valid;  // pixel is a candidate iff finite and > min_valid
    float lo, hi;     // and lo <= v <= hi (cumulative `retain` bounds; +-inf when unused)
    ab_pixel_xf xf;   // optional normalisation applied to every pixel as it is loaded
};

struct Keying {  // mode 0: key = bits(v); 1: bits((f32)|(f64)v - center64|); 2: bits(|v - center32|)
    int mode;
    double center64;
    float center32;
};

// finite && v > min_valid && lo <= v <= hi, folded into two compares: the bounds below are loop invariants
// (v > m  <=>  v >= next float above m, for m >= 0; |v| <= FLT_MAX  <=>  v is finite; NaN fails every compare)
__device__ __forceinline__ bool candidate(const Window &w, float v) {
    const float above_min = __uint_as_float(__float_as_uint(w.min_valid) + 1u);
    const float lo = fmaxf(fmaxf(w.lo, -3.402823466e+38f), above_min);
    const float hi = fminf(w.hi, 3.402823466e+38f);
    return v >= lo && v <= hi;
}

__device__ __forceinline__ uint32_t key_of(const Keying &k, float v) {
    float x = v;
    if (k.mode == 1) x = (float)fabs((double)v - k.center64);
    if (k.mode == 2) x = fabsf(v - k.center32);
    return __float_as_uint(x);
}

// Where a pass gets the window's pixels from: the window is re-read (L2) every pass.  A pass is latency-bound --
// one dependent load per 1024 pixels costs ~0.8 us -- so each thread keeps eight independent loads in flight
// (out-of-window slots read as NaN, which is never a candidate).  Keeping a whole 256 x 256 tile in registers
// instead was tried: at 1024 threads LLVM spills (128-VGPR budget), at 512 threads it still spills 60+ dwords.
struct StreamSource {
    // thread (tx, ty) of the 256 x 4 layout walks column x0 + tx (+ 256 per column block) downwards, four rows per
    // sweep: consecutive lanes read consecutive pixels and no per-pixel integer division is needed
    template <class F>
    __device__ __forceinline__ void for_each(const Window &w, F f) const {
        constexpr int U = 8;
        const int tx = threadIdx.x & 255, ty = threadIdx.x >> 8;
        for (int cb = w.x0; cb < w.x1; cb += 256) {
            const int c = cb + tx;
            const bool col_ok = c < w.x1;
            for (int rb = w.y0 + ty; rb < w.y1 + ty; rb += 4 * U) {  // uniform trip count over the block
                float v[U];
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const int r = rb + 4 * u;
                    v[u] = __builtin_nanf("");
                    if (col_ok && r < w.y1) v[u] = ab_px(w.xf, w.img[(int64_t)r * w.ld + c]);
                }
#pragma unroll
                for (int u = 0; u < U; ++u) f(v[u]);
            }
        }
    }
};

// A window of <= 256 x 256 pixels held entirely on chip: the first NL of a thread's 64 sweeps live in LDS
// (conflict-free, stride kBlock), the remaining NR in registers.  StreamSource re-reads a 256 KiB tile ~19 times;
// with one tile per CU the live set (32 tiles = 8 MiB per XCD) overflows the 4 MiB L2, and rocprofv3 counted
// 1.2 GB of fetches per 4096^2 frame for the tile kernel alone.  Same thread mapping as StreamSource.
template <int NL, int NR>
struct TileSource {
    static_assert(NL + NR == 64, "a 256 x 256 tile is 64 sweeps of the 256 x 4 thread layout");
    float *lds;  // NL * kBlock floats
    mutable float regs[NR];
    __device__ __forceinline__ void load(const Window &w) {
        const int tx = threadIdx.x & 255, ty = threadIdx.x >> 8;
        const int c = w.x0 + tx;
        const bool col_ok = c < w.x1;
#pragma unroll 8
        for (int k = 0; k < NL; ++k) {
            const int r = w.y0 + ty + 4 * k;
            float v = __builtin_nanf("");
            if (col_ok && r < w.y1) v = ab_px(w.xf, w.img[(int64_t)r * w.ld + c]);
            lds[k * kBlock + threadIdx.x] = v;
        }
#pragma unroll
        for (int k = 0; k < NR; ++k) {
            const int r = w.y0 + ty + 4 * (NL + k);
            float v = __builtin_nanf("");
            if (col_ok && r < w.y1) v = ab_px(w.xf, w.img[(int64_t)r * w.ld + c]);
            regs[k] = v;
        }
    }
    template <class F>
    __device__ __forceinline__ void for_each(const Window &, F f) const {
#pragma unroll 5
        for (int k = 0; k < NL; ++k) f(lds[k * kBlock + threadIdx.x]);
#pragma unroll
        for (int k = 0; k < NR; ++k) {
            asm volatile("" : "+v"(regs[k]));  // keeps LLVM from hoisting per-pixel keys across the passes (spills)
            f(regs[k]);
        }
    }
};

// Top-level histogram update with the wave's dominant bin counted in a scalar: the first candidate a wave meets
// names the "mode" bin; lanes hitting it are tallied with ballot + popcount (no atomic, no serialisation) and
// flushed once at the end of the pass, everybody else uses a per-lane atomic.
struct ModeTally {
    uint32_t bin = 0;
    unsigned int count = 0;
    bool have = false;
    __device__ __forceinline__ void add(unsigned int *hist, bool valid, uint32_t b) {
        if (!have) {
            const unsigned long long m = __ballot(valid);
            if (m) {
                // v_readlane, not __shfl: a ds_bpermute here puts an s_waitcnt lgkmcnt(0) on the common path of EVERY
                // pixel, which also waits for the previous pixel's histogram atomic
                bin = (uint32_t)__builtin_amdgcn_readlane((int)b, (int)__builtin_ctzll(m));
                have = true;
            }
        }
        const bool hit = valid && have && b == bin;
        count += (unsigned int)__builtin_popcountll(__ballot(hit));
        if (valid && !hit) atomicAdd(&hist[b], 1u);
    }
    __device__ __forceinline__ void flush(unsigned int *hist) {
        if (have && count && (threadIdx.x & 63) == 0) atomicAdd(&hist[bin], count);
    }
};

// histogram over bits [shift, shift+nbits) of the keys matching the prefix
template <class S>
__device__ inline void window_hist(const S &src, const Window &w, const Keying &k, uint32_t prefix_mask, uint32_t prefix_val, int shift,
                                   int nbits, unsigned int *hist /* LDS, 1 << nbits */) {
    const int nb = 1 << nbits;
    for (int i = threadIdx.x; i < nb; i += kBlock) hist[i] = 0;
    __syncthreads();
    auto visit = [&](auto add) {
        src.for_each(w, [&](float v) {
            bool ok = candidate(w, v);
            uint32_t bin = 0;
            if (ok) {
                const uint32_t key = key_of(k, v);
                ok = (key & prefix_mask) == prefix_val;
                bin = (key >> shift) & (uint32_t)(nb - 1);
            }
            add(ok, bin);
        });
    };
    if (prefix_mask == 0) {  // top level: sky pixels pile up in one or two bins
        ModeTally tally;
        visit([&](bool ok, uint32_t bin) { tally.add(hist, ok, bin); });
        tally.flush(hist);
    } else {  // lower levels: keys are spread over the bins
        visit([&](bool ok, uint32_t bin) {
            if (ok) atomicAdd(&hist[bin], 1u);
        });
    }
    __syncthreads();
}

// prepare() and the level-1 pass of the following select in ONE sweep: the top-level histogram as above, plus the
// level-1 histogram (bits 20..10) of the candidates whose top bits equal `guess` -- the bin the wanted rank is expected
// in (the previous iteration's median / MAD, or any sample of the window: all sky pixels share their top 11 bits).  A
// select that then finds its rank in that bin skips its own level-1 pass; otherwise nothing is lost but the atomics.
template <class S>
__device__ inline void window_hist_fused(const S &src, const Window &w, const Keying &k, uint32_t guess, unsigned int *hist0 /* LDS, 2048 */,
                                         unsigned int *hist1 /* LDS, 2048 */) {
    for (int i = threadIdx.x; i < 2048; i += kBlock) {
        hist0[i] = 0;
        hist1[i] = 0;
    }
    __syncthreads();
    ModeTally tally;
    src.for_each(w, [&](float v) {
        bool ok = candidate(w, v);
        uint32_t bin = 0, key = 0;
        if (ok) {
            key = key_of(k, v);
            bin = key >> 21;
        }
        tally.add(hist0, ok, bin);
        if (ok && bin == guess) atomicAdd(&hist1[(key >> 10) & 2047u], 1u);
    });
    tally.flush(hist0);
    __syncthreads();
}

// which top-level bin `hist` already holds the level-1 histogram of (none: valid == false).  lean: no top-level
// histogram was built at all, only the candidate counts below (before0) and inside (in0) that bin -- enough whenever
// the wanted ranks fall inside it (lean_holds), which the caller checks before it selects.
struct Spec {
    bool valid = false;
    uint32_t bin0 = 0;
    bool lean = false;
    unsigned int before0 = 0, in0 = 0;
};
__device__ __forceinline__ bool lean_holds(const Spec &s, unsigned int n) {  // ranks n/2 (and n/2 - 1 for even n)
    if (n == 0) return false;
    const unsigned int mid = n / 2, lowest = (n % 2 == 0) ? mid - 1 : mid;
    return lowest >= s.before0 && mid < s.before0 + s.in0;
}

// Block-wide exclusive scan of the histogram (every thread owns nb / kBlock consecutive bins: wave scan by
// shuffles, 16 wave totals through LDS) and location of up to two ranks in it.  Results are broadcast.
// For a rank >= total the answer is (nb - 1, 0), as a linear scan that never fires would give.
struct BinHit {
    unsigned int bin, before;
};
template <int NRANKS>
__device__ inline void locate(const unsigned int *hist, int nb, const unsigned int (&ranks)[NRANKS], BinHit (&hits)[NRANKS],
                              unsigned int *total_out) {
    __shared__ unsigned int wave_tot[kBlock / 64], s_bin[2], s_before[2], s_total;
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
    const int per = nb / kBlock;  // 2 (2048 bins) or 1 (1024 bins)
    __syncthreads();              // histogram complete; previous results consumed
    const unsigned int h0 = hist[t * per], h1 = per == 2 ? hist[t * per + 1] : 0u;
    const unsigned int s = h0 + h1;
    unsigned int inc = s;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const unsigned int u = __shfl_up(inc, off, 64);
        if (lane >= off) inc += u;
    }
    if (lane == 63) wave_tot[wv] = inc;
    if (t < NRANKS) {
        s_bin[t] = (unsigned int)nb - 1u;
        s_before[t] = 0u;
    }
    __syncthreads();
    unsigned int base = 0, total = 0;
#pragma unroll
    for (int i = 0; i < kBlock / 64; ++i) {
        const unsigned int w = wave_tot[i];
        base += i < wv ? w : 0u;
        total += w;
    }
    const unsigned int exc = base + inc - s;
#pragma unroll
    for (int q = 0; q < NRANKS; ++q) {
        const unsigned int rank = ranks[q];
        if (rank >= exc && rank < exc + s) {  // exactly one thread (s > 0) owns the rank
            const bool first = rank < exc + h0;
            s_bin[q] = (unsigned int)(t * per) + (first ? 0u : 1u);
            s_before[q] = first ? exc : exc + h0;
        }
    }
    if (t == 0) s_total = total;
    __syncthreads();
#pragma unroll
    for (int q = 0; q < NRANKS; ++q) {
        hits[q].bin = s_bin[q];
        hits[q].before = s_before[q];
    }
    *total_out = s_total;
}

__device__ inline void find_bin(const unsigned int *hist, int nb, unsigned int rank, unsigned int *bin_out, unsigned int *before_out,
                                unsigned int *total_out) {
    const unsigned int ranks[1] = {rank};
    BinHit hits[1];
    locate<1>(hist, nb, ranks, hits, total_out);
    *bin_out = hits[0].bin;
    *before_out = hits[0].before;
}

// find_bin for two ranks (rank_lo <= rank_hi) off one scan
__device__ inline void find_bin2(const unsigned int *hist, int nb, unsigned int rank_lo, unsigned int rank_hi, unsigned int *bin_lo,
                                 unsigned int *before_lo, unsigned int *bin_hi, unsigned int *before_hi) {
    const unsigned int ranks[2] = {rank_lo, rank_hi};
    BinHit hits[2];
    unsigned int total;
    locate<2>(hist, nb, ranks, hits, &total);
    *bin_lo = hits[0].bin;
    *before_lo = hits[0].before;
    *bin_hi = hits[1].bin;
    *before_hi = hits[1].before;
}

// top-level (bits 31..21) histogram of the window's keys into hist0; returns the candidate count
template <class S>
__device__ inline unsigned int prepare(const S &src, const Window &w, const Keying &k, unsigned int *hist0 /* LDS, 2048 */) {
    window_hist(src, w, k, 0, 0, 21, 11, hist0);
    unsigned int b, bf, n;
    find_bin(hist0, 2048, 0xffffffffu, &b, &bf, &n);
    return n;
}

// prepare() that also speculates the level-1 histogram of top-level bin `guess` into `hist` (see window_hist_fused)
template <class S>
__device__ inline unsigned int prepare_spec(const S &src, const Window &w, const Keying &k, unsigned int *hist0, unsigned int *hist, uint32_t guess,
                                            Spec *spec) {
    window_hist_fused(src, w, k, guess & 2047u, hist0, hist);
    unsigned int b, bf, n;
    find_bin(hist0, 2048, 0xffffffffu, &b, &bf, &n);
    spec->valid = true;
    spec->bin0 = guess & 2047u;
    return n;
}

// The lean form of prepare_spec: ONE sweep that histograms level 1 of the guessed top-level bin and merely counts the
// candidates in lower / higher top-level bins (two compares and two predicated adds per pixel instead of the mode tally).
template <class S>
__device__ inline unsigned int prepare_lean(const S &src, const Window &w, const Keying &k, unsigned int *hist /* LDS, 2048 */, uint32_t guess,
                                            Spec *spec, unsigned int *tally /* LDS, 2 */) {
    guess &= 2047u;
    for (int i = threadIdx.x; i < 2048; i += kBlock) hist[i] = 0;
    if (threadIdx.x < 2) tally[threadIdx.x] = 0;
    __syncthreads();
    unsigned int lt = 0, gt = 0;
    src.for_each(w, [&](float v) {
        if (candidate(w, v)) {
            const uint32_t key = key_of(k, v), b0 = key >> 21;
            lt += b0 < guess ? 1u : 0u;
            gt += b0 > guess ? 1u : 0u;
            if (b0 == guess) atomicAdd(&hist[(key >> 10) & 2047u], 1u);
        }
    });
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        lt += __shfl_xor(lt, off, 64);
        gt += __shfl_xor(gt, off, 64);
    }
    if ((threadIdx.x & 63) == 0) {
        if (lt) atomicAdd(&tally[0], lt);
        if (gt) atomicAdd(&tally[1], gt);
    }
    unsigned int b, bf, in;
    find_bin(hist, 2048, 0xffffffffu, &b, &bf, &in);  // (its leading barrier also publishes the tallies)
    spec->valid = true;
    spec->lean = true;
    spec->bin0 = guess;
    spec->before0 = tally[0];
    spec->in0 = in;
    return tally[0] + in + tally[1];
}

// levels 1 and 2 for a key whose top bits (val, mask) and in-bin rank are known
template <class S>
__device__ inline float descend(const S &src, const Window &w, const Keying &k, uint32_t mask, uint32_t val, unsigned int rank, int level,
                                unsigned int *hist /* LDS, 2048 */, bool first_level_ready = false) {
    const int shifts[3] = {21, 10, 0}, bits[3] = {11, 11, 10};
    for (int p = level; p < 3; ++p) {
        if (!(first_level_ready && p == level)) window_hist(src, w, k, mask, val, shifts[p], bits[p], hist);
        unsigned int bin, before, total;
        find_bin(hist, 1 << bits[p], rank, &bin, &before, &total);
        rank -= before;
        val |= bin << shifts[p];
        mask |= ((1u << bits[p]) - 1u) << shifts[p];
    }
    return __uint_as_float(val);
}

// the rank-th smallest key (0-based) given the prepared top-level histogram
template <class S>
__device__ inline float select_from(const S &src, const Window &w, const Keying &k, const unsigned int *hist0, unsigned int rank, unsigned int *hist,
                                    Spec spec = Spec()) {
    unsigned int bin = spec.bin0, before = spec.before0, total;
    if (!spec.lean) find_bin(hist0, 2048, rank, &bin, &before, &total);
    return descend(src, w, k, 0x7ffu << 21, bin << 21, rank - before, 1, hist, spec.valid && spec.bin0 == bin);
}

// keys of ranks r-1 and r (r >= 1): they share passes for as long as they sit in the same bin
template <class S>
__device__ inline void select_pair_from(const S &src, const Window &w, const Keying &k, const unsigned int *hist0, unsigned int r, unsigned int *hist,
                                        float *lower_out, float *upper_out, Spec spec = Spec()) {
    const int shifts[3] = {21, 10, 0}, bits[3] = {11, 11, 10};
    uint32_t mask = 0, val = 0;
    unsigned int rank_hi = r, rank_lo = r - 1;
    for (int p = 0; p < 3; ++p) {
        const unsigned int *h = hist0;
        if (p > 0) {
            // level 1 may already sit in hist (prepare_spec guessed the pair's top-level bin)
            if (!(p == 1 && spec.valid && (val >> 21) == spec.bin0)) window_hist(src, w, k, mask, val, shifts[p], bits[p], hist);
            h = hist;
        }
        unsigned int bin_hi, before_hi, bin_lo, before_lo;
        if (p == 0 && spec.lean) {  // both ranks are known to sit in the guessed bin (lean_holds)
            bin_hi = bin_lo = spec.bin0;
            before_hi = before_lo = spec.before0;
        } else {
            find_bin2(h, 1 << bits[p], rank_lo, rank_hi, &bin_lo, &before_lo, &bin_hi, &before_hi);
        }
        const uint32_t lvl_mask = ((1u << bits[p]) - 1u) << shifts[p];
        if (bin_hi != bin_lo) {  // the pair straddles a bin boundary: finish each on its own (a speculated level 1 is simply not used)
            *lower_out = descend(src, w, k, mask | lvl_mask, val | (bin_lo << shifts[p]), rank_lo - before_lo, p + 1, hist);
            *upper_out = descend(src, w, k, mask | lvl_mask, val | (bin_hi << shifts[p]), rank_hi - before_hi, p + 1, hist);
            return;
        }
        rank_hi -= before_hi;
        rank_lo -= before_lo;
        val |= bin_hi << shifts[p];
        mask |= lvl_mask;
    }
    *lower_out = __uint_as_float(val);  // identical keys
    *upper_out = __uint_as_float(val);
}

// median_f32_mut (math/median.rs:46-63) of the n > 0 prepared keys: f32 average of the two middle values
template <class S>
__device__ inline float median_f32_from(const S &src, const Window &w, const Keying &k, const unsigned int *hist0, unsigned int n, unsigned int *hist,
                                        Spec spec = Spec()) {
    const unsigned int mid = n / 2;
    if (n % 2 == 0) {
        float left, right;
        select_pair_from(src, w, k, hist0, mid, hist, &left, &right, spec);
        return (left + right) / 2.0f;
    }
    return select_from(src, w, k, hist0, mid, hist, spec);
}

// exact_median_mut (math/median.rs:27-44): f64 average of the two middle values
template <class S>
__device__ inline double exact_median_from(const S &src, const Window &w, const Keying &k, const unsigned int *hist0, unsigned int n, unsigned int *hist,
                                           Spec spec = Spec()) {
    const unsigned int mid = n / 2;
    if (n % 2 == 0) {
        float left, right;
        select_pair_from(src, w, k, hist0, mid, hist, &left, &right, spec);
        return ((double)left + (double)right) / 2.0;
    }
    return (double)select_from(src, w, k, hist0, mid, hist, spec);
}

}  // namespace absel
