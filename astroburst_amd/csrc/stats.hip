// Whole-image statistics on gfx950.
//
// Replaces core/imaging/stats.rs: compute_image_stats (:15-23), the exact path (:43-73,
// math/median.rs:27-73), the 65 536-bin histogram path for > 4 000 000 px (:75-210), the known-
// range variant (:25-41) and build_histogram (:378-421); plus cmd/common.rs:18-22
// (auto_stretch_preview = stats -> auto_stf -> apply_stf) as one asynchronous chain.
//
// Design (HBM-bound integer/histogram work -- no GEMM shapes here):
//   * every pass is one streaming read of the plane, float4 loads, two loads in flight per lane.
//   * 65 536-bin histograms are privatised per workgroup in LDS as PACKED 16-bit counters
//     (65 536 x u16 = 128 KiB of the CU's 160 KiB).  A workgroup consumes at most 65 535 pixels
//     between flushes, so a counter cannot overflow; a flush adds the non-zero counters to the
//     global u64 histogram.  Bin indices are computed in f64 exactly as the reference does
//     (`((v as f64 - min) * inv) as usize`, saturating), so every bin count is bit-exact.
//   * the exact path (<= 4M px) needs order statistics, not a sort: valid pixels are positive
//     finite floats, whose bit patterns are monotone as u32, so rank k is found by an 11/11/10-bit
//     radix select; the two middle ranks of an even count descend together.
//   * the scalar bookkeeping between passes (percentile bin, in-bin interpolation: stats.rs:302-353)
//     runs ON THE DEVICE in one-workgroup kernels that write the next pass's parameters into a small
//     state block in HBM: the host enqueues the whole chain and synchronises once, for the 56-byte
//     result (round 1 downloaded four 512 KiB histograms and synchronised after each: 0.3 ms of host
//     round trips in a 0.65 ms call).
//   * row-band sharding: with a communicator every rank runs the same chain over ITS
//     rows and the integer partials -- min/max, counts, the three 65 536-bin histograms -- are
//     all-reduced in-stream between a pass and its bookkeeping kernel (comm.hip); histograms and
//     counts are integers, so the sharded statistics equal the single-GPU ones bin for bin (the f64
//     sum of the mean is reduced in a different order: 1e-16 relative).
//   * f64 sums: per-thread sequential over a strided slice, then a fixed-shape tree; the
//     reference's own order is rayon's unspecified reduce tree (stats.rs:252-257).
#include "ab_common.hpp"

#include <fcntl.h>
#include <sys/file.h>
#include <unistd.h>
#include "stf_device.hpp"

#include <algorithm>
#include <cfloat>
#include <cmath>
#include <mutex>

namespace {

constexpr double kMadToSigma = 1.4826;      // types/constants.rs:7
constexpr int kHistBins = 65536;            // stats.rs:8
constexpr int kExactLimit = 4000000;        // stats.rs:18

// Rust `f64 as usize` then `.min(last)`: saturating, NaN -> 0
__device__ __forceinline__ uint32_t bin_index(double t, uint32_t last) {
    if (!(t > 0.0)) return 0;
    if (t >= (double)last) return last;
    return (uint32_t)t;
}

struct ScanPartial {
    double mn, mx, sum;
    unsigned long long cnt;
};

constexpr int kScanBlock = 256;
constexpr int kMaxPartials = 4096;
constexpr int kBookBlock = 1024;

// Parameters and results that travel from pass to pass in HBM (one per context, AB_WS_STATS).
struct StatsDev {
    double negmin_max[2];       // {-min, max} of the valid pixels: one all-reduce(MAX) serves both
    double sum;                 // f64 sum of the valid pixels (all-reduce SUM)
    int empty;                  // no valid pixel anywhere: result is all zeros (stats.rs:79-81,95-97)
    int two;                    // exact path: even count, two middle ranks
    // histogram path (stats.rs:85-210)
    double gmin, inv, bin_width, range;
    unsigned long long total_valid, half_count, count_before_median;
    double median_bin_lo, median_bin_hi, refine_inv, refine_range;
    double dev_inv, dev_bw;
    double median;
    double mad_region_lo, mad_refine_inv, mad_refine_range;
    float coarse_med_f32, exact_med_f32, mad_lo_f32, mad_hi_f32;
    // exact path (radix select; rank[0] = mid, rank[1] = mid - 1)
    unsigned long long rank[2];
    unsigned int prefix[2];
    unsigned int prefix_mask;
    int use_dev;
    float center;
    ab_image_stats result;
    ab_stf_params stf;
    unsigned long long done;    // the resident kernel's completion marker (fetched with result and stf: one copy)
    StfTx tx;
};

// the resident kernel's part of the workspace (stats_resident.hpp)
constexpr int kResMaxGrid = 512;   // workgroups (one per CU)
constexpr int kResLevels = 6;      // grid-wide histogram reductions per call
constexpr int kSlabRow = 512;      // u32 per workgroup and level: histogram A [0, 256), histogram B / a count [256, 512)
struct ResWs {
    unsigned int *slab;       // kResLevels x kResMaxGrid x kSlabRow: every workgroup's 256-bin counts, written whole each call
    ScanPartial *p1, *p2, *p3;  // per-workgroup partials (range / sum and count / count below the MAD region), 128 bytes apart
    unsigned int *bar;        // kResMaxGrid arrival flags, the abort flag, timing stamps (4 KiB, cleared per call)
};

// workspace carved from AB_WS_STATS
struct Ws {
    StatsDev *st;
    unsigned long long *H;   // [0] = a count riding along with a histogram all-reduce; [1, 65537) = h0; [65537, 131073) = h1
    ScanPartial *partials;   // kMaxPartials
    unsigned int *sel;       // 2 x 2048
    unsigned int *gsum;      // 2 x 1024 group totals (64 bins each) of h0 / h1
    ResWs res;
};

__device__ __forceinline__ double shfl_xor_f64(double v, int off) { return __shfl_xor(v, off, 64); }

__device__ __forceinline__ void block_reduce_scan(double mn, double mx, double sum, unsigned long long cnt,
                                                  ScanPartial *out) {
    __shared__ double s_mn[16], s_mx[16], s_sum[16];
    __shared__ unsigned long long s_cnt[16];
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        mn = fmin(mn, shfl_xor_f64(mn, off));
        mx = fmax(mx, shfl_xor_f64(mx, off));
        sum += shfl_xor_f64(sum, off);
        cnt += __shfl_xor(cnt, off, 64);
    }
    const int w = threadIdx.x >> 6, nw = blockDim.x >> 6;
    if ((threadIdx.x & 63) == 0) {
        s_mn[w] = mn;
        s_mx[w] = mx;
        s_sum[w] = sum;
        s_cnt[w] = cnt;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int i = 1; i < nw; ++i) {
            mn = fmin(mn, s_mn[i]);
            mx = fmax(mx, s_mx[i]);
            sum += s_sum[i];
            cnt += s_cnt[i];
        }
        out->mn = mn;
        out->mx = mx;
        out->sum = sum;
        out->cnt = cnt;
    }
}

// Streams `n` floats through f(v): float4 loads, two in flight per lane, grid-stride over blocks of 8 floats per lane.
// The plane base must be 16-byte aligned (checked by the callers; otherwise the scalar tail loop takes everything).
template <typename F>
__device__ __forceinline__ void stream_pixels(const float *__restrict__ data, int64_t begin, int64_t end, F &&f) {
    const bool aligned = (((uintptr_t)(data + begin)) & 15) == 0;
    int64_t i = begin;
    if (aligned) {
        const int64_t n8 = (end - begin) >> 3;
        const float4 *d4 = reinterpret_cast<const float4 *>(data + begin);
        const int64_t stride = (int64_t)blockDim.x;
        for (int64_t k = threadIdx.x; k < n8; k += stride) {
            const float4 a = d4[2 * k], b = d4[2 * k + 1];
            f(a.x); f(a.y); f(a.z); f(a.w);
            f(b.x); f(b.y); f(b.z); f(b.w);
        }
        i = begin + (n8 << 3);
    }
    for (int64_t k = i + threadIdx.x; k < end; k += blockDim.x) f(data[k]);
}

// block b of g owns the pixel range [lo, hi): equal contiguous slices, 8-float aligned
__device__ __forceinline__ void block_slice(int64_t n, int64_t *lo, int64_t *hi) {
    const int64_t per = (((n + gridDim.x - 1) / gridDim.x) + 7) & ~(int64_t)7;
    const int64_t a = per * blockIdx.x;
    *lo = a < n ? a : n;
    *hi = (a + per) < n ? (a + per) : n;
}

// stats.rs:212-258: min / max / sum / count over valid pixels; one partial per workgroup
__global__ __launch_bounds__(kScanBlock) void scan_kernel(const float *__restrict__ data, int64_t n,
                                                          ScanPartial *__restrict__ partials) {
    double mn = DBL_MAX, mx = -DBL_MAX, sum = 0.0;
    unsigned long long cnt = 0;
    int64_t lo, hi;
    block_slice(n, &lo, &hi);
    stream_pixels(data, lo, hi, [&](float v) {
        if (is_valid_pixel(v)) {
            const double vf = (double)v;
            mn = fmin(mn, vf);
            mx = fmax(mx, vf);
            sum += vf;
            cnt += 1;
        }
    });
    block_reduce_scan(mn, mx, sum, cnt, &partials[blockIdx.x]);
}

// one workgroup: fixed-shape reduction of the per-workgroup partials (thread t folds partials t, t + 1024, ...)
// (stride: in units of ScanPartial; the resident kernel keeps every workgroup's partial on a cache line of its own)
// AGENT: the partials were written by other workgroups of THIS launch (the resident kernel): they are read with device-scope
// (sc1) loads, the pair of the producers' sc1 stores -- a plain load may be served a line this XCD's L2 kept from an earlier launch
template <bool AGENT = false>
__device__ __forceinline__ void reduce_partials(const ScanPartial *p, int np, double *mn, double *mx, double *sum,
                                                unsigned long long *cnt, int stride = 1) {
    double a = DBL_MAX, b = -DBL_MAX, s = 0.0;
    unsigned long long c = 0;
    for (int i = threadIdx.x; i < np; i += blockDim.x) {
        ScanPartial q;
        if constexpr (AGENT) {
            unsigned long long *src = reinterpret_cast<unsigned long long *>(const_cast<ScanPartial *>(p + (size_t)i * stride));
            static_assert(sizeof(ScanPartial) == 32, "four 8-byte words");
            q.mn = __longlong_as_double((long long)__hip_atomic_load(&src[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
            q.mx = __longlong_as_double((long long)__hip_atomic_load(&src[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
            q.sum = __longlong_as_double((long long)__hip_atomic_load(&src[2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
            q.cnt = __hip_atomic_load(&src[3], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            q = p[(size_t)i * stride];
        }
        a = fmin(a, q.mn);
        b = fmax(b, q.mx);
        s += q.sum;
        c += q.cnt;
    }
    __shared__ ScanPartial tot;
    block_reduce_scan(a, b, s, c, &tot);
    __syncthreads();
    *mn = tot.mn;
    *mx = tot.mx;
    *sum = tot.sum;
    *cnt = tot.cnt;
    __syncthreads();
}

// after scan_kernel: this rank's {-min, max}, sum and count into the state / the count slot
__global__ __launch_bounds__(kBookBlock) void finish_scan_kernel(const ScanPartial *p, int np, StatsDev *st, unsigned long long *H) {
    double mn, mx, sum;
    unsigned long long cnt;
    reduce_partials(p, np, &mn, &mx, &sum, &cnt);
    if (threadIdx.x == 0) {
        st->negmin_max[0] = -mn;
        st->negmin_max[1] = mx;
        st->sum = sum;
        H[0] = cnt;
    }
}

__global__ void set_range_kernel(StatsDev *st, double kmin, double kmax) {
    st->negmin_max[0] = -kmin;
    st->negmin_max[1] = kmax;
}

// ---- dense 65 536-bin histograms, LDS-privatised with packed u16 counters -------------------
constexpr int kHistBlock = 1024;
constexpr int kHistChunkMax = 61440;  // pixels per workgroup between flushes (< 65 536)

enum HistKind { HIST_VALUE = 0, HIST_DEV = 1, HIST_MAD = 2 };

__device__ __forceinline__ void lds_hist_add(unsigned int *lds, uint32_t idx) {
    atomicAdd(&lds[idx >> 1], (idx & 1) ? 0x10000u : 1u);
}

__device__ __forceinline__ void lds_hist_flush(unsigned int *lds, unsigned long long *hist) {
    __syncthreads();
    for (int w = threadIdx.x; w < kHistBins / 2; w += kHistBlock) {
        const unsigned int packed = lds[w];
        if (packed) {
            const unsigned int lo = packed & 0xffffu, hi = packed >> 16;
            if (lo) atomicAdd(&hist[2 * w], (unsigned long long)lo);
            if (hi) atomicAdd(&hist[2 * w + 1], (unsigned long long)hi);
            lds[w] = 0;
        }
    }
    __syncthreads();
}

// KIND = HIST_VALUE (stats.rs:260-300):   h0[bin(v)] += 1, per-workgroup sum / count of the valid pixels
// KIND = HIST_DEV   (stats.rs:119-146):   h0[bin(|v - coarse_median|)] += 1 (dense, LDS) and the 65 536 sub-bins of the
//                                         median bin in h1 (sparse: one coarse bin's worth of pixels, global atomics)
// KIND = HIST_MAD   (stats.rs:166-191):   count of deviations below the MAD region (per-workgroup) and the region's
//                                         65 536 sub-bins in h0 (three coarse bins' worth of pixels spread over all the sub-bins;
//                                         LDS-privatised like the others: straight global atomics were tried and took 115 us
//                                         against 75 -- device-scope atomics are served at the memory side, not in an XCD's L2)
template <int KIND>
__global__ __launch_bounds__(kHistBlock) void dense_hist_kernel(const float *__restrict__ data, int64_t n, int64_t chunk,
                                                                const StatsDev *__restrict__ st, unsigned long long *__restrict__ h0,
                                                                unsigned long long *__restrict__ h1, ScanPartial *__restrict__ partials) {
    extern __shared__ __attribute__((aligned(16))) unsigned int lds[];  // 32 768 words = 65 536 x u16
    for (int i = threadIdx.x; i < kHistBins / 2; i += kHistBlock) lds[i] = 0;
    __syncthreads();

    const uint32_t last = kHistBins - 1;
    double sum = 0.0;
    unsigned long long cnt = 0;
    // pass parameters (uniform; a handful of scalar loads)
    const double origin = st->gmin, inv = st->inv;
    const double refine_lo = st->median_bin_lo, refine_hi = st->median_bin_hi, refine_inv = st->refine_inv;
    const double dev_inv = st->dev_inv;
    const float center = KIND == HIST_DEV ? st->coarse_med_f32 : st->exact_med_f32;
    const float mad_lo = st->mad_lo_f32, mad_hi = st->mad_hi_f32;
    const double mad_region_lo = st->mad_region_lo, mad_inv = st->mad_refine_inv;
    const int skip = st->empty;

    const int64_t nchunks = skip ? 0 : (n + chunk - 1) / chunk;
    for (int64_t c = blockIdx.x; c < nchunks; c += gridDim.x) {
        const int64_t base = c * chunk;
        const int64_t end = (base + chunk < n) ? base + chunk : n;
        stream_pixels(data, base, end, [&](float v) {
            if (is_valid_pixel(v)) {
                if (KIND == HIST_VALUE) {
                    const double vf = (double)v;
                    sum += vf;
                    cnt += 1;
                    lds_hist_add(lds, bin_index((vf - origin) * inv, last));  // stats.rs:282-283
                } else if (KIND == HIST_DEV) {
                    const double vf = (double)v;
                    if (vf >= refine_lo && vf < refine_hi)  // stats.rs:127-130
                        atomicAdd(&h1[bin_index((vf - refine_lo) * refine_inv, last)], 1ull);
                    const float d = fabsf(v - center);  // stats.rs:131-133
                    lds_hist_add(lds, bin_index((double)d * dev_inv, last));
                } else {
                    const float dev = fabsf(v - center);  // stats.rs:172-181
                    if (dev < mad_lo)
                        cnt += 1;
                    else if (dev < mad_hi)
                        lds_hist_add(lds, bin_index(((double)dev - mad_region_lo) * mad_inv, last));
                }
            }
        });
        lds_hist_flush(lds, h0);
    }
    if (KIND != HIST_DEV) block_reduce_scan(0.0, 0.0, sum, cnt, &partials[blockIdx.x]);
}

// after the VALUE / MAD pass: per-rank sum and count
__global__ __launch_bounds__(kBookBlock) void finish_hist_kernel(const ScanPartial *p, int np, StatsDev *st, unsigned long long *H, int want_sum) {
    double mn, mx, sum;
    unsigned long long cnt;
    reduce_partials(p, np, &mn, &mx, &sum, &cnt);
    if (threadIdx.x == 0) {
        if (want_sum) st->sum = sum;
        H[0] = cnt;
    }
}

// ---- one-workgroup bookkeeping between the passes (stats.rs:302-353 on the device) -----------------------------
__device__ __forceinline__ unsigned long long f64_to_u64_sat(double v) {  // Rust `as u64`
    if (!(v > 0.0)) return 0;
    if (v >= 18446744073709551615.0) return 0xffffffffffffffffull;
    return (unsigned long long)v;
}

// first bin i with cum(i) >= target over 65 536 bins (cum inclusive).  Every thread returns the same answer:
// found, the bin, its count and the inclusive cumulative count there; `before` = cum - count.
struct RankHit {
    int found;
    uint32_t bin;
    unsigned long long count, cum;
};
// totals of the 1024 groups of 64 consecutive bins of `nhist` consecutive 65 536-bin histograms: 64 workgroups per histogram read
// it coalesced (one workgroup reading 512 KiB on its own took 20 us of the bookkeeping kernels' 19 - 35)
__global__ __launch_bounds__(256) void group_totals_kernel(const unsigned long long *__restrict__ hist, unsigned int *__restrict__ gsum) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const size_t g = (size_t)blockIdx.x * 16 + (size_t)w * 4 + i;  // group index over all histograms
        // (a plane has fewer than 2^31 pixels, so do all the bins of a histogram together: group totals fit 32 bits)
        unsigned int c = (unsigned int)hist[g * 64 + lane];
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) c += __shfl_xor(c, off, 64);
        if (lane == 0) gsum[g] = c;
    }
}

// first bin i with cum(i) >= target over 65 536 bins (cum inclusive), given the 1024 group totals
__device__ __forceinline__ RankHit block_find_rank(const unsigned long long *hist, const unsigned int *gsum, unsigned long long target) {
    __shared__ unsigned long long s_part[kBookBlock];
    __shared__ RankHit s_hit;
    __shared__ int s_owner;
    constexpr int PER = kHistBins / kBookBlock;  // 64 consecutive bins per thread
    const int t = threadIdx.x;
    const unsigned long long local = gsum[t];
    s_part[t] = local;
    if (t == 0) {
        s_hit.found = 0;
        s_owner = -1;
    }
    __syncthreads();
    // inclusive scan of 1024 partials (Hillis-Steele in LDS)
    for (int off = 1; off < kBookBlock; off <<= 1) {
        const unsigned long long add = t >= off ? s_part[t - off] : 0ull;
        __syncthreads();
        s_part[t] += add;
        __syncthreads();
    }
    const unsigned long long incl = s_part[t], excl = incl - local;
    if (excl < target && incl >= target) s_owner = t;  // the crossing lies in this thread's 64 bins (exactly one thread, target >= 1)
    __syncthreads();
    const int owner = s_owner;
    if (owner >= 0 && t < 64) {  // wave 0 resolves the owner's 64 bins: one bin per lane, wave prefix sum, first lane at or past the target
        const unsigned long long c = hist[(size_t)owner * PER + t];
        unsigned long long cum = c;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const unsigned long long up = __shfl_up(cum, off, 64);
            if (t >= off) cum += up;
        }
        cum += s_part[owner] - __shfl(cum, 63, 64);  // + the exclusive prefix of the owner's range
        const unsigned long long m = __ballot(cum >= target);
        if (m && t == (int)__builtin_ctzll(m)) {
            s_hit.found = 1;
            s_hit.bin = (uint32_t)(owner * PER + t);
            s_hit.count = c;
            s_hit.cum = cum;
        }
    }
    __syncthreads();
    const RankHit r = s_hit;
    __syncthreads();
    return r;
}

// resolve_rank_in_hist (stats.rs:333-353)
__device__ __forceinline__ double resolve_from_hit(const RankHit &h, unsigned long long rank, double region_lo, double sub_bw) {
    if (!h.found) return region_lo + (double)kHistBins * sub_bw;
    const unsigned long long overshoot = h.cum - rank;
    const double frac = h.count > 0 ? 1.0 - ((double)overshoot / (double)h.count) : 0.5;
    return region_lo + ((double)h.bin + frac) * sub_bw;
}
__device__ __forceinline__ double resolve_rank(const unsigned long long *hist, const unsigned int *gsum, unsigned long long rank, double region_lo,
                                               double sub_bw) {
    if (rank == 0) return region_lo;  // uniform
    return resolve_from_hit(block_find_rank(hist, gsum, rank), rank, region_lo, sub_bw);
}

__device__ __forceinline__ void write_zero_result(StatsDev *st) {
    st->empty = 1;
    st->result.min = st->result.max = st->result.median = st->result.mad = st->result.sigma = st->result.mean = 0.0;
    st->result.valid_count = 0;
}

// The bookkeeping bodies below are shared by the one-workgroup kernels of the chain and by the resident kernel
// (stats_resident.hpp), where every workgroup runs them on its own LDS copy of the state: `st` is a generic pointer.
// before the VALUE pass: the histogram's origin and scale from the (all-reduced) range; one thread
__device__ void book_range_body(StatsDev *st) {
    const double gmin = -st->negmin_max[0], gmax = st->negmin_max[1];
    st->empty = 0;
    st->two = 0;
    if (gmin == DBL_MAX) {  // stats.rs:79-81: no valid pixel
        write_zero_result(st);
        return;
    }
    const double range = fmax(gmax - gmin, 1e-30);  // stats.rs:91-92
    st->gmin = gmin;
    st->range = range;
    st->bin_width = range / (double)kHistBins;
    st->inv = (double)kHistBins / range;  // stats.rs:269
    st->result.min = gmin;
    st->result.max = gmax;
}
__global__ void book_range_kernel(StatsDev *st) { book_range_body(st); }

// after the VALUE pass (stats.rs:94-117): mean, the median's coarse bin, parameters of the DEV pass
// one thread: `h` = where the value histogram's cumulative count reaches `half`; last_count = the count of its last bin
__device__ void book_value_apply(StatsDev *st, unsigned long long total, unsigned long long half, const RankHit &h, unsigned long long last_count) {
    const double gmin = st->gmin, bw = st->bin_width, range = st->range;
    const uint32_t median_bin = h.found ? h.bin : (uint32_t)(kHistBins - 1);  // find_percentile_bin (:302-311)
    // count_before_median = sum of the bins below the median bin (:103)
    const unsigned long long before = h.found ? h.cum - h.count : total - last_count;
    double coarse;  // interpolate_percentile (:313-331)
    if (h.found) {
        const unsigned long long overshoot = h.cum - half;
        const double frac = h.count > 0 ? 1.0 - ((double)overshoot / (double)h.count) : 0.5;
        coarse = gmin + ((double)h.bin + frac) * bw;
    } else {
        coarse = gmin + (double)kHistBins * bw;
    }
    st->total_valid = total;
    st->half_count = half;
    st->count_before_median = before;
    st->median_bin_lo = gmin + (double)median_bin * bw;  // :104-105
    st->median_bin_hi = st->median_bin_lo + bw;
    st->refine_range = fmax(st->median_bin_hi - st->median_bin_lo, 1e-30);  // :116
    st->refine_inv = (double)kHistBins / st->refine_range;
    st->dev_bw = range / (double)kHistBins;  // :111-113
    st->dev_inv = (double)kHistBins / range;
    st->coarse_med_f32 = (float)coarse;  // :114
    st->result.mean = st->sum / (double)total;  // :98
    st->result.valid_count = total;
}
__device__ __forceinline__ unsigned long long half_of(unsigned long long total) {
    return f64_to_u64_sat(ceil((double)total * 0.5));  // :100 (== find_percentile_bin's target)
}
// (a workgroup of kBookBlock threads; `total` = the valid count, `hist` = the value histogram, G its group totals)
__device__ void book_value_body(StatsDev *st, unsigned long long total, const unsigned long long *hist, const unsigned int *G) {
    if (st->empty) return;
    if (total == 0) {  // stats.rs:95-97
        __syncthreads();
        if (threadIdx.x == 0) write_zero_result(st);
        return;
    }
    const unsigned long long half = half_of(total);
    const RankHit h = block_find_rank(hist, G, half);
    if (threadIdx.x == 0) book_value_apply(st, total, half, h, hist[kHistBins - 1]);
}
__global__ __launch_bounds__(kBookBlock) void book_value_kernel(StatsDev *st, const unsigned long long *H, const unsigned int *G) {
    book_value_body(st, H[0], H + 1, G);
}

// after the DEV pass (stats.rs:148-164): the exact median from the refined bin, the MAD's coarse region
// one thread: the refined median and `h` = where the deviation histogram's cumulative count reaches half
__device__ void book_dev_apply(StatsDev *st, double median, const RankHit &h) {
    const uint32_t mad_bin = h.found ? h.bin : (uint32_t)(kHistBins - 1);
    const uint32_t expand_lo = mad_bin > 0 ? mad_bin - 1 : 0;                                         // :155
    const uint32_t expand_hi = (mad_bin + 2 < (uint32_t)kHistBins) ? mad_bin + 2 : (uint32_t)kHistBins;  // :156
    const double lo = (double)expand_lo * st->dev_bw, hi = (double)expand_hi * st->dev_bw;
    st->median = median;
    st->exact_med_f32 = (float)median;  // :160
    st->mad_region_lo = lo;
    st->mad_refine_range = fmax(hi - lo, 1e-30);
    st->mad_refine_inv = (double)kHistBins / st->mad_refine_range;
    st->mad_lo_f32 = (float)lo;  // :163-164
    st->mad_hi_f32 = (float)hi;
    st->result.median = median;
}
// (dev = the deviation histogram, refine = the 65 536 sub-bins of the median's bin; Gd / Gr their group totals)
__device__ void book_dev_body(StatsDev *st, const unsigned long long *dev, const unsigned long long *refine, const unsigned int *Gd, const unsigned int *Gr) {
    if (st->empty) return;
    const unsigned long long half = st->half_count, before = st->count_before_median;
    const unsigned long long rank_in_bin = half > before ? half - before : 0;  // saturating_sub (:148)
    const double median = resolve_rank(refine, Gr, rank_in_bin, st->median_bin_lo, st->refine_range / (double)kHistBins);
    const RankHit h = block_find_rank(dev, Gd, half);  // find_percentile_bin(dev_hist, total, 0.5) (:154)
    if (threadIdx.x == 0) book_dev_apply(st, median, h);
}
__global__ __launch_bounds__(kBookBlock) void book_dev_kernel(StatsDev *st, const unsigned long long *H, const unsigned int *G) {
    book_dev_body(st, H + 1, H + 1 + kHistBins, G, G + 1024);
}

__device__ __forceinline__ void finish_result(StatsDev *st, double mad, const ab_auto_stf_config cfg) {
    if (!st->empty) {  // (an empty image keeps the all-zero ImageStats, stats.rs:79-81)
        st->result.mad = mad;
        st->result.sigma = fmax(mad * kMadToSigma, 1e-30);
    }
    ab_auto_stf_hd(&st->result, &cfg, &st->stf);
    st->tx = make_tx(&st->stf, &st->result);
}

// after the MAD pass (stats.rs:193-209)
__global__ __launch_bounds__(kBookBlock) void book_mad_kernel(StatsDev *st, const unsigned long long *H, const unsigned int *G, ab_auto_stf_config cfg) {
    if (st->empty) {
        if (threadIdx.x == 0) finish_result(st, 0.0, cfg);
        return;
    }
    const unsigned long long below = H[0], half = st->half_count;
    const unsigned long long rank = half > below ? half - below : 0;
    const double mad = resolve_rank(H + 1, G, rank, st->mad_region_lo, st->mad_refine_range / (double)kHistBins);
    if (threadIdx.x == 0) finish_result(st, mad, cfg);
}

// stats.rs:393-410 build_histogram: arbitrary bin count, u32 bins; sparse enough for global atomics
// when bins is small is NOT true (512 display bins are hot), so privatise in LDS when it fits.
__global__ __launch_bounds__(kScanBlock) void small_hist_kernel(const float *__restrict__ data, int64_t n, uint32_t bins,
                                                                double dmin, double inv, unsigned int *__restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) unsigned int lds[];
    const bool use_lds = bins <= 16384;
    if (use_lds) {
        for (uint32_t i = threadIdx.x; i < bins; i += kScanBlock) lds[i] = 0;
        __syncthreads();
    }
    const uint32_t last = bins - 1;
    const int64_t stride = (int64_t)gridDim.x * kScanBlock;
    for (int64_t i = (int64_t)blockIdx.x * kScanBlock + threadIdx.x; i < n; i += stride) {
        const float v = data[i];
        if (is_valid_pixel(v)) {
            const uint32_t idx = bin_index(((double)v - dmin) * inv, last);
            if (use_lds)
                atomicAdd(&lds[idx], 1u);
            else
                atomicAdd(&out[idx], 1u);
        }
    }
    if (use_lds) {
        __syncthreads();
        for (uint32_t i = threadIdx.x; i < bins; i += kScanBlock)
            if (lds[i]) atomicAdd(&out[i], lds[i]);
    }
}

// ---- radix select for the exact path --------------------------------------------------------
// keys: u32 bit pattern of v (valid pixels, positive) or of |v - center| (>= +0); both monotone.  Two ranks (the two
// middle order statistics of an even count) descend together: sel[0 .. 2048) counts the keys under prefix[0],
// sel[2048 .. 4096) those under prefix[1].
__global__ __launch_bounds__(kScanBlock) void select_hist_kernel(const float *__restrict__ data, int64_t n, const StatsDev *__restrict__ st,
                                                                 int shift, int nbits, unsigned int *__restrict__ sel) {
    __shared__ unsigned int lds[4096];
    for (uint32_t i = threadIdx.x; i < 4096; i += kScanBlock) lds[i] = 0;
    __syncthreads();
    if (st->empty) return;
    const uint32_t nb = 1u << nbits;
    const uint32_t mask = st->prefix_mask, p0 = st->prefix[0], p1 = st->prefix[1];
    const int two = st->two, use_dev = st->use_dev;
    const float center = st->center;
    int64_t lo, hi;
    block_slice(n, &lo, &hi);
    stream_pixels(data, lo, hi, [&](float v) {
        if (is_valid_pixel(v)) {
            const float k = use_dev ? fabsf(v - center) : v;
            const uint32_t key = __float_as_uint(k);
            const uint32_t bin = (key >> shift) & (nb - 1);
            if ((key & mask) == p0) atomicAdd(&lds[bin], 1u);
            if (two && (key & mask) == p1) atomicAdd(&lds[2048 + bin], 1u);
        }
    });
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < 4096; i += kScanBlock)
        if (lds[i]) atomicAdd(&sel[i], lds[i]);
}

// each of the two ranks steps into the bin that holds it: one wave per rank, a lane adds up its 32 (16) consecutive bins, the wave
// scans the lane totals, and the lane whose range holds the rank walks it.  (The first version walked all 2048 bins from one lane,
// every step a dependent load: 40 .. 170 us per call, six calls per statistics call -- a third of the GPU time of BASELINE
// configs[0].)
__global__ __launch_bounds__(128) void select_pick_kernel(StatsDev *st, const unsigned int *__restrict__ sel, int shift, int nbits) {
    if (st->empty) return;
    const uint32_t nb = 1u << nbits, per = nb / 64u;  // nb = 2048 or 1024
    const int r = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const bool active = r == 0 || st->two;
    __shared__ uint32_t s_bin[2];
    __shared__ unsigned long long s_rank[2];
    if (active) {
        const unsigned int *h = sel + 2048 * r + (size_t)lane * per;
        unsigned int c[32];
        unsigned long long mine = 0;
#pragma unroll
        for (uint32_t i = 0; i < 32; ++i) {
            c[i] = i < per ? h[i] : 0u;
            mine += c[i];
        }
        unsigned long long incl = mine;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const unsigned long long up = __shfl_up(incl, off, 64);
            if (lane >= off) incl += up;
        }
        const unsigned long long rank = st->rank[r], excl = incl - mine, total = __shfl(incl, 63, 64);
        // the first bin i with cum(i) > rank; none (rank >= total): the serial walk fell through to bin nb - 1 with everything counted
        if (rank >= excl && rank < incl) {
            unsigned long long cum = excl;
            int bin = -1;
#pragma unroll
            for (uint32_t i = 0; i < 32; ++i)
                if (i < per && bin < 0) {
                    if (cum + c[i] > rank)
                        bin = (int)i;
                    else
                        cum += c[i];
                }
            s_bin[r] = lane * per + (uint32_t)bin;
            s_rank[r] = rank - cum;
        }
        if (rank >= total && lane == 63) {
            s_bin[r] = nb - 1;
            s_rank[r] = rank - total;
        }
    }
    __syncthreads();
    if (active && lane == 0) {
        st->rank[r] = s_rank[r];
        st->prefix[r] |= s_bin[r] << shift;
    }
    __syncthreads();
    if (threadIdx.x == 0) st->prefix_mask |= (nb - 1) << shift;
}

// exact path bookkeeping (stats.rs:43-73, math/median.rs:27-73)
__global__ void exact_begin_kernel(StatsDev *st, const unsigned long long *H) {
    const unsigned long long cnt = H[0];
    st->empty = 0;
    if (cnt == 0) {
        write_zero_result(st);
        return;
    }
    st->result.min = -st->negmin_max[0];
    st->result.max = st->negmin_max[1];
    st->result.mean = st->sum / (double)cnt;
    st->result.valid_count = cnt;
    st->two = (cnt % 2 == 0) ? 1 : 0;
    st->rank[0] = cnt / 2;
    st->rank[1] = cnt / 2 - (st->two ? 1 : 0);
    st->prefix[0] = st->prefix[1] = 0;
    st->prefix_mask = 0;
    st->use_dev = 0;
    st->center = 0.0f;
}
__global__ void exact_mid_kernel(StatsDev *st) {  // after the value select: the median, then the same select on |v - median|
    if (st->empty) return;
    const float right = __uint_as_float(st->prefix[0]), left = __uint_as_float(st->prefix[1]);
    const double median = st->two ? ((double)left + (double)right) / 2.0 : (double)right;  // exact_median_mut (median.rs:27-44)
    st->result.median = median;
    st->center = (float)median;  // exact_mad_mut(valid, median as f32) (stats.rs:60)
    st->use_dev = 1;
    st->rank[0] = st->result.valid_count / 2;
    st->rank[1] = st->result.valid_count / 2 - (st->two ? 1 : 0);
    st->prefix[0] = st->prefix[1] = 0;
    st->prefix_mask = 0;
}
__global__ void exact_end_kernel(StatsDev *st, ab_auto_stf_config cfg) {
    if (st->empty) {
        finish_result(st, 0.0, cfg);
        return;
    }
    const float dr = __uint_as_float(st->prefix[0]), dl = __uint_as_float(st->prefix[1]);
    const float mad_f32 = st->two ? (dl + dr) / 2.0f : dr;  // median_f32_mut (median.rs:46-63): f32 average
    finish_result(st, (double)mad_f32, cfg);
}

#include "stats_resident.hpp"

// ---------------------------------------------------------------------------------------------
int grid_for(ab_ctx *ctx, int64_t n, int block, int per_cu) {
    int64_t want = (n + block - 1) / block;
    int64_t cap = (int64_t)(ctx->cu_count > 0 ? ctx->cu_count : 256) * per_cu;
    return (int)std::max<int64_t>(1, std::min(want, cap));
}

int carve(ab_ctx *ctx, Ws *w) {
    static_assert(sizeof(StatsDev) <= 1024, "StatsDev outgrew its slot");
    constexpr size_t kChain = 1024 + (size_t)(1 + 2 * kHistBins) * sizeof(unsigned long long) + kMaxPartials * sizeof(ScanPartial) + 4096 * 4 + 2048 * 4;
    constexpr size_t kRes = 256 + 4096 + 3 * kResMaxGrid * 128 + (size_t)kResLevels * kResMaxGrid * kSlabRow * 4;
    char *c = nullptr;
    AB_TRY(ab_workspace(ctx, AB_WS_STATS, kChain + kRes, (void **)&c));
    w->st = (StatsDev *)c;
    c += 1024;
    w->H = (unsigned long long *)c;
    c += (size_t)(1 + 2 * kHistBins) * sizeof(unsigned long long);
    w->partials = (ScanPartial *)c;
    c += kMaxPartials * sizeof(ScanPartial);
    w->sel = (unsigned int *)c;
    c += 4096 * sizeof(unsigned int);
    w->gsum = (unsigned int *)c;
    c += 2048 * sizeof(unsigned int);
    c = (char *)(((uintptr_t)c + 255) & ~(uintptr_t)255);  // (rows and flags on cache lines of their own)
    w->res.bar = (unsigned int *)c;
    c += 4096;
    static_assert(sizeof(ScanPartial) == 32, "the partials are spaced 4 apart = one 128-byte line each");
    w->res.p1 = (ScanPartial *)c;
    w->res.p2 = w->res.p1 + 4 * kResMaxGrid;
    w->res.p3 = w->res.p2 + 4 * kResMaxGrid;
    c += 3 * kResMaxGrid * 128;
    w->res.slab = (unsigned int *)c;
    return AB_OK;
}

// pixels per workgroup between flushes: the chunks divide evenly among the workgroups (273 chunks of 61 440 px on 256
// CUs made 17 workgroups do double duty: the pass took twice one chunk's time)
void hist_launch_shape(ab_ctx *ctx, int64_t n, int *grid, int64_t *chunk) {
    const int cus = ctx->cu_count > 0 ? ctx->cu_count : 256;
    int64_t nchunks = std::max<int64_t>(1, (n + kHistChunkMax - 1) / kHistChunkMax);  // (n == 0: an empty row band of a sharded image)
    const int g = (int)std::max<int64_t>(1, std::min<int64_t>(nchunks, cus));
    nchunks = ((nchunks + g - 1) / g) * g;
    int64_t c = (n + nchunks - 1) / nchunks;
    c = (c + 7) & ~(int64_t)7;  // keep every chunk 32-byte aligned relative to the plane
    *grid = g;
    *chunk = std::min<int64_t>(std::max<int64_t>(c, 8), kHistChunkMax);
}

template <int KIND>
int launch_dense(ab_ctx *ctx, const float *data, int64_t n, const Ws &w, int *grid_out) {
    int grid;
    int64_t chunk;
    hist_launch_shape(ctx, n, &grid, &chunk);
    const size_t lds_bytes = kHistBins * sizeof(unsigned short);
    AB_HIP(ctx, hipFuncSetAttribute((const void *)dense_hist_kernel<KIND>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
    hipLaunchKernelGGL(dense_hist_kernel<KIND>, dim3(grid), dim3(kHistBlock), lds_bytes, ctx->stream, data, n, chunk, w.st, w.H + 1,
                       w.H + 1 + kHistBins, w.partials);
    AB_HIP(ctx, hipGetLastError());
    *grid_out = grid;
    return AB_OK;
}

const ab_auto_stf_config kDefaultStf = {0.25, -2.8};  // AutoStfConfig::default (types/image.rs:52-65)

// min / max / sum / count of this rank's pixels -> state, all-reduced
int enqueue_scan(ab_ctx *ctx, ab_comm *comm, const float *data, int64_t n, const Ws &w, bool need_sum) {
    const int grid = std::min(grid_for(ctx, n, kScanBlock * 16, 8), kMaxPartials);
    hipLaunchKernelGGL(scan_kernel, dim3(grid), dim3(kScanBlock), 0, ctx->stream, data, n, w.partials);
    hipLaunchKernelGGL(finish_scan_kernel, dim3(1), dim3(kBookBlock), 0, ctx->stream, w.partials, grid, w.st, w.H);
    AB_HIP(ctx, hipGetLastError());
    if (comm) {
        AB_TRY(ab_comm_allreduce(ctx, comm, w.st->negmin_max, 2, AB_DT_F64, AB_RED_MAX));
        if (need_sum) {  // the exact path takes its mean from this pass; the histogram path re-sums in its VALUE pass
            AB_TRY(ab_comm_allreduce(ctx, comm, &w.st->sum, 1, AB_DT_F64, AB_RED_SUM));
            AB_TRY(ab_comm_allreduce(ctx, comm, w.H, 1, AB_DT_U64, AB_RED_SUM));
        }
    }
    return AB_OK;
}

// stats.rs:85-210, everything on the stream; w.st->result / stf / tx are final when the stream reaches this point
int enqueue_hist_path(ab_ctx *ctx, ab_comm *comm, const float *data, int64_t n, const Ws &w, const ab_auto_stf_config &cfg) {
    int grid = 0;
    hipLaunchKernelGGL(book_range_kernel, dim3(1), dim3(1), 0, ctx->stream, w.st);
    AB_HIP(ctx, hipMemsetAsync(w.H, 0, (size_t)(1 + 2 * kHistBins) * sizeof(unsigned long long), ctx->stream));
    AB_TRY(launch_dense<HIST_VALUE>(ctx, data, n, w, &grid));
    hipLaunchKernelGGL(finish_hist_kernel, dim3(1), dim3(kBookBlock), 0, ctx->stream, w.partials, grid, w.st, w.H, 1);
    if (comm) {
        AB_TRY(ab_comm_allreduce(ctx, comm, w.H, 1 + kHistBins, AB_DT_U64, AB_RED_SUM));
        AB_TRY(ab_comm_allreduce(ctx, comm, &w.st->sum, 1, AB_DT_F64, AB_RED_SUM));
    }
    hipLaunchKernelGGL(group_totals_kernel, dim3(64), dim3(256), 0, ctx->stream, w.H + 1, w.gsum);
    hipLaunchKernelGGL(book_value_kernel, dim3(1), dim3(kBookBlock), 0, ctx->stream, w.st, w.H, w.gsum);
    AB_HIP(ctx, hipMemsetAsync(w.H + 1, 0, (size_t)(2 * kHistBins) * sizeof(unsigned long long), ctx->stream));
    AB_TRY(launch_dense<HIST_DEV>(ctx, data, n, w, &grid));
    if (comm) AB_TRY(ab_comm_allreduce(ctx, comm, w.H + 1, 2 * kHistBins, AB_DT_U64, AB_RED_SUM));
    hipLaunchKernelGGL(group_totals_kernel, dim3(128), dim3(256), 0, ctx->stream, w.H + 1, w.gsum);
    hipLaunchKernelGGL(book_dev_kernel, dim3(1), dim3(kBookBlock), 0, ctx->stream, w.st, w.H, w.gsum);
    AB_HIP(ctx, hipMemsetAsync(w.H, 0, (size_t)(1 + kHistBins) * sizeof(unsigned long long), ctx->stream));
    AB_TRY(launch_dense<HIST_MAD>(ctx, data, n, w, &grid));
    hipLaunchKernelGGL(finish_hist_kernel, dim3(1), dim3(kBookBlock), 0, ctx->stream, w.partials, grid, w.st, w.H, 0);
    if (comm) AB_TRY(ab_comm_allreduce(ctx, comm, w.H, 1 + kHistBins, AB_DT_U64, AB_RED_SUM));
    hipLaunchKernelGGL(group_totals_kernel, dim3(64), dim3(256), 0, ctx->stream, w.H + 1, w.gsum);
    hipLaunchKernelGGL(book_mad_kernel, dim3(1), dim3(kBookBlock), 0, ctx->stream, w.st, w.H, w.gsum, cfg);
    AB_HIP(ctx, hipGetLastError());
    return AB_OK;
}

// stats.rs:43-73
int enqueue_exact_path(ab_ctx *ctx, ab_comm *comm, const float *data, int64_t n, const Ws &w, const ab_auto_stf_config &cfg) {
    AB_TRY(enqueue_scan(ctx, comm, data, n, w, true));
    hipLaunchKernelGGL(exact_begin_kernel, dim3(1), dim3(1), 0, ctx->stream, w.st, w.H);
    const int shifts[3] = {21, 10, 0};
    const int bits[3] = {11, 11, 10};
    const int grid = grid_for(ctx, n, kScanBlock * 8, 8);
    for (int sel = 0; sel < 2; ++sel) {  // values, then deviations from the median
        for (int pass = 0; pass < 3; ++pass) {
            AB_HIP(ctx, hipMemsetAsync(w.sel, 0, 4096 * sizeof(unsigned int), ctx->stream));
            hipLaunchKernelGGL(select_hist_kernel, dim3(grid), dim3(kScanBlock), 0, ctx->stream, data, n, w.st, shifts[pass], bits[pass], w.sel);
            if (comm) AB_TRY(ab_comm_allreduce(ctx, comm, w.sel, 4096, AB_DT_U32, AB_RED_SUM));
            hipLaunchKernelGGL(select_pick_kernel, dim3(1), dim3(128), 0, ctx->stream, w.st, w.sel, shifts[pass], bits[pass]);
        }
        if (sel == 0)
            hipLaunchKernelGGL(exact_mid_kernel, dim3(1), dim3(1), 0, ctx->stream, w.st);
        else
            hipLaunchKernelGGL(exact_end_kernel, dim3(1), dim3(1), 0, ctx->stream, w.st, cfg);
    }
    AB_HIP(ctx, hipGetLastError());
    return AB_OK;
}

// The resident kernel (stats_resident.hpp) takes the histogram path of one unsharded, 16-byte aligned plane that one
// workgroup per CU can hold; AB_STATS_CHAIN=1 keeps the chain (the GPU tests run both).
bool resident_takes(ab_ctx *ctx, const float *data, int64_t n, const uint8_t *u8) {
    const char *e = ab_env("AB_STATS_CHAIN");
    if (e && *e && *e != '0') return false;
    if (ctx->stats_aborts >= 3) return false;  // (fewer CUs than it reports, e.g. a CU mask: every launch would wait out its barrier first)
    const int64_t cus = ctx->cu_count;
    return cus > 0 && n > 0 && (((uintptr_t)data) & 15) == 0 && (((uintptr_t)u8) & 3) == 0 &&
           (n + kResTile - 1) / kResTile <= std::min<int64_t>(cus, kResMaxGrid);
}

int enqueue_resident(ab_ctx *ctx, const float *data, int64_t n, const Ws &w, int known, double kmin, double kmax, const ab_auto_stf_config &cfg,
                     uint8_t *u8) {
    if (ctx->stats_bar != (const void *)w.res.bar || ctx->stats_epoch > 0xffff0000u) {  // new workspace / after an abort / epoch about to wrap
        AB_HIP(ctx, hipMemsetAsync(w.res.bar, 0, 4096, ctx->stream));
        AB_HIP(ctx, hipMemsetAsync(&w.st->done, 0, sizeof w.st->done, ctx->stream));
        ctx->stats_bar = w.res.bar;
        ctx->stats_epoch = 0;
    }
    const unsigned int epoch_base = ctx->stats_epoch;
    ctx->stats_epoch += 8;  // (seven barriers per launch)
    ctx->stats_expect = (unsigned long long)epoch_base + 8u;
    // test hooks.  AB_STATS_FORCE_ABORT=1: the kernel finds the abort flag raised at its first barrier.  =b<k> (k = 1 .. 7): the
    // LAST workgroup behaves at its k-th barrier as if it had timed out AFTER publishing its arrival -- its peers pass that barrier
    // and run to the end without it (what a real time-out at the final barrier does)
    int abort_at = 0;
    if (const char *e = ab_dev_env("AB_STATS_FORCE_ABORT"); e && *e) {
        if (*e == '1') AB_HIP(ctx, hipMemsetAsync(w.res.bar + kBarAbort, 1, 1, ctx->stream));
        if (*e == 'b') abort_at = atoi(e + 1);
    }
    const unsigned grid = (unsigned)((n + kResTile - 1) / kResTile);
    hipLaunchKernelGGL(stats_resident_kernel, dim3(grid), dim3(kResBlock), 0, ctx->stream, data, n, w.st, w.res, epoch_base, known, kmin, kmax, cfg, u8, abort_at);
    AB_HIP(ctx, hipGetLastError());
    return AB_OK;
}

// One resident kernel at a time per device: a workgroup needs a whole CU (16 waves x 128 registers), so two of them launched
// together would each hold part of the chip and wait for the rest until the barrier times out.  A caller that finds the lock
// taken uses the chain.  The lock has two halves: a mutex for the threads of this process, and an advisory file lock
// (flock on /dev/shm/astroburst_resident_<uid>_<PCI bus id>.lock, non-blocking) for OTHER processes on the same GPU -- several ranks
// may share one device (tests/multirank_worker.py does).  Where the file cannot be opened the process half alone decides; two
// processes colliding then costs each a timed-out barrier (0.5 s), the abort flag, the chain's (exact) result, and after three in
// a row the context stays on the chain (resident_takes): slow, never wrong.
struct ResidentLock {
    std::mutex m;
    int fd = -2;  // -2: not opened yet, -1: unavailable
    bool try_lock(int device) {
        if (!m.try_lock()) return false;
        if (fd == -2) {
            // keyed on the GPU's PCI address, not on this process's HIP ordinal: under HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES two
            // processes number the same GPU differently (no guard at all) and different GPUs alike (false serialisation) -- ADVICE r4.
            // O_NOFOLLOW: /dev/shm is world-writable, the name is predictable; a planted symlink is refused, not followed.
            char path[128], bus[32] = "";
            if (hipDeviceGetPCIBusId(bus, (int)sizeof bus, device) != hipSuccess) {
                (void)hipGetLastError();
                snprintf(bus, sizeof bus, "ordinal%d", device);
            }
            for (char *c = bus; *c; ++c)
                if (*c == ':' || *c == '.' || *c == '/') *c = '_';
            snprintf(path, sizeof path, "/dev/shm/astroburst_resident_%u_%s.lock", (unsigned)getuid(), bus);
            fd = open(path, O_CREAT | O_RDWR | O_CLOEXEC | O_NOFOLLOW, 0600);
        }
        if (fd >= 0 && flock(fd, LOCK_EX | LOCK_NB) != 0) {
            m.unlock();
            return false;
        }
        return true;
    }
    void unlock() {
        if (fd >= 0) (void)flock(fd, LOCK_UN);
        m.unlock();
    }
};
ResidentLock &resident_lock(int device) {
    static ResidentLock l[64];
    return l[device & 63];
}

// `resident` (nullable): the caller will look at the abort flag after its synchronisation and can re-run the chain, so the
// resident kernel may be used; it then also writes the stretched plane to `u8` (nullable).  *resident = whether it was.
int stats_enqueue(ab_ctx *ctx, ab_comm *comm, const float *data, int64_t n, int64_t n_total, int use_known, double known_min, double known_max,
                  const ab_auto_stf_config *stf_cfg, const ab_image_stats **result_dev, const void **tx_dev, const ab_stf_params **stf_dev,
                  uint8_t *u8, bool *resident) {
    AB_HIP(ctx, hipSetDevice(ctx->device));
    Ws w;
    AB_TRY(carve(ctx, &w));
    const ab_auto_stf_config cfg = stf_cfg ? *stf_cfg : kDefaultStf;
    if (resident) *resident = false;
    if (n_total <= kExactLimit) {  // stats.rs:18-22,32-34
        AB_TRY(enqueue_exact_path(ctx, comm, data, n, w, cfg));
    } else {
        const bool known = use_known && std::isfinite(known_min) && std::isfinite(known_max) && known_min < known_max;  // stats.rs:36-38
        if (resident && !comm && resident_takes(ctx, data, n, u8) && resident_lock(ctx->device).try_lock(ctx->device)) {
            *resident = true;  // (the caller unlocks after its synchronisation: ResidentGuard)
            AB_TRY(enqueue_resident(ctx, data, n, w, known ? 1 : 0, known_min, known_max, cfg, u8));
        } else {
            if (known) {
                hipLaunchKernelGGL(set_range_kernel, dim3(1), dim3(1), 0, ctx->stream, w.st, known_min, known_max);
            } else {
                AB_TRY(enqueue_scan(ctx, comm, data, n, w, false));
            }
            AB_TRY(enqueue_hist_path(ctx, comm, data, n, w, cfg));
        }
    }
    if (result_dev) *result_dev = &w.st->result;
    if (tx_dev) *tx_dev = &w.st->tx;
    if (stf_dev) *stf_dev = &w.st->stf;
    return AB_OK;
}

struct ResidentGuard {  // releases the device's resident-kernel lock when the call that may have taken it returns
    ab_ctx *ctx;
    bool *held;
    ~ResidentGuard() {
        if (*held) resident_lock(ctx->device).unlock();
    }
};

}  // namespace

// The asynchronous form: enqueues compute_image_stats (+ auto_stf with `stf_cfg`, default config if null) of `data` on the
// context's stream and returns the device addresses of the result / the STF transform; nothing is synchronised.
// n_total = the pixel count that selects the exact (<= 4 000 000) or the histogram path: the plane's own n, or the
// whole image's when `data` is one row band of it and `comm` joins the bands (stats.rs:18).
int ab_stats_enqueue(ab_ctx *ctx, ab_comm *comm, const float *data, int64_t n, int64_t n_total, int use_known, double known_min,
                     double known_max, const ab_auto_stf_config *stf_cfg, const ab_image_stats **result_dev, const void **tx_dev,
                     const ab_stf_params **stf_dev) {
    return stats_enqueue(ctx, comm, data, n, n_total, use_known, known_min, known_max, stf_cfg, result_dev, tx_dev, stf_dev, nullptr, nullptr);
}

// *aborted (nullable; after a resident launch): whether the completion marker beside the result is NOT this launch's -- a grid
// barrier timed out and the workgroups left.  The barrier flags are then cleared before the next launch.
static int fetch_result(ab_ctx *ctx, const ab_image_stats *result_dev, ab_image_stats *out, ab_stf_params *stf_out, ab_comm *comm = nullptr,
                        bool *aborted = nullptr) {
    void *pin = nullptr;
    constexpr size_t kDoneAt = sizeof(ab_image_stats) + sizeof(ab_stf_params);
    static_assert(offsetof(StatsDev, done) == offsetof(StatsDev, result) + kDoneAt, "result, stf and the marker are adjacent");
    AB_TRY(ab_pinned(ctx, kDoneAt + 8, &pin));
    static_assert(offsetof(StatsDev, stf) == offsetof(StatsDev, result) + sizeof(ab_image_stats), "result and stf are adjacent");
    AB_HIP(ctx, hipMemcpyAsync(pin, result_dev, kDoneAt + 8, hipMemcpyDeviceToHost, ctx->stream));
    AB_TRY(ab_comm_stream_wait(ctx, comm));  // (a plain hipStreamSynchronize without a communicator)
    if (aborted) {
        unsigned long long done;
        memcpy(&done, (const char *)pin + kDoneAt, sizeof done);
        *aborted = done != ctx->stats_expect;
        if (*aborted) ctx->stats_bar = nullptr;
        ctx->stats_aborts = *aborted ? ctx->stats_aborts + 1 : 0;
        if (*aborted) ab_count_fallback(ctx, AB_FB_STATS_CHAIN);  // (the caller repeats the launch as the chain)
    }
    if (aborted && ab_dev_env("AB_STATS_TIMING")) {  // workgroup 0's phase stamps (s_memtime: shader clock cycles)
        Ws w;
        AB_TRY(carve(ctx, &w));
        unsigned long long st[24];
        AB_HIP(ctx, hipMemcpy(st, w.res.bar + kBarStamps, sizeof st, hipMemcpyDeviceToHost));
        fprintf(stderr, "stats_resident phases (100 cycles):");
        for (int i = 1; i < 23; ++i) fprintf(stderr, " %d:%.1f", i, st[i] > st[0] ? (double)(st[i] - st[0]) / 100.0 : -1.0);
        fprintf(stderr, "\n");
    }
    if (out) memcpy(out, pin, sizeof *out);
    if (stf_out) memcpy(stf_out, (char *)pin + sizeof(ab_image_stats), sizeof *stf_out);
    return AB_OK;
}

int ab_stats_device(ab_ctx *ctx, const float *data, int64_t n, int use_known, double known_min, double known_max,
                    ab_image_stats *out) {
    const ab_image_stats *res = nullptr;
    bool resident = false, aborted = false;
    ResidentGuard guard{ctx, &resident};
    AB_TRY(stats_enqueue(ctx, nullptr, data, n, n, use_known, known_min, known_max, nullptr, &res, nullptr, nullptr, nullptr, &resident));
    AB_TRY(fetch_result(ctx, res, out, nullptr, nullptr, resident ? &aborted : nullptr));
    if (!aborted) return AB_OK;
    // a grid barrier of the resident kernel timed out (its workgroups were not all resident): the chain takes the plane
    AB_TRY(ab_stats_enqueue(ctx, nullptr, data, n, n, use_known, known_min, known_max, nullptr, &res, nullptr, nullptr));
    return fetch_result(ctx, res, out, nullptr);
}

extern "C" {

int ab_compute_image_stats(ab_ctx *ctx, const ab_plane *img, ab_image_stats *out) try {
    if (!ctx) return AB_ERR_INVALID;
    AB_CHECK(ctx, img && out, "null plane or output");
    StagedPlane in;
    AB_TRY(ab_stage_in(ctx, img, &in));
    const int rc = ab_stats_device(ctx, in.dptr, in.rows * in.cols, 0, 0.0, 0.0, out);
    ab_stage_release(ctx, &in);
    return rc;
} AB_CATCH(ctx)

int ab_compute_image_stats_with_known_range(ab_ctx *ctx, const ab_plane *img, double known_min, double known_max,
                                            ab_image_stats *out) try {
    if (!ctx) return AB_ERR_INVALID;
    AB_CHECK(ctx, img && out, "null plane or output");
    StagedPlane in;
    AB_TRY(ab_stage_in(ctx, img, &in));
    const int rc = ab_stats_device(ctx, in.dptr, in.rows * in.cols, 1, known_min, known_max, out);
    ab_stage_release(ctx, &in);
    return rc;
} AB_CATCH(ctx)

// compute_image_stats of an image whose rows are spread over the ranks of `comm`: `band` is this rank's
// rows, total_rows the whole image's.  Every rank receives the statistics of the WHOLE image.
int ab_compute_image_stats_sharded(ab_ctx *ctx, ab_comm *comm, const ab_plane *band, int64_t total_rows, ab_image_stats *out) try {
    if (!ctx) return AB_ERR_INVALID;
    // arguments + workspace first, agreed on across the ranks: the chain below interleaves kernels and all-reduces, a rank that
    // bailed out before it would leave its peers inside the first of them
    auto local = [&]() -> int {
        AB_CHECK(ctx, band && out && band->on_device, "sharded statistics take a device-resident row band");
        AB_CHECK(ctx, band->rows >= 0 && (band->rows == 0 || band->data), "null band data");
        AB_CHECK(ctx, total_rows >= band->rows && band->cols > 0, "total_rows (%lld) is smaller than the band (%lld rows)", (long long)total_rows,
                 (long long)band->rows);
        AB_HIP(ctx, hipSetDevice(ctx->device));
        Ws w;
        return carve(ctx, &w);
    };
    AB_TRY(ab_comm_agree(ctx, comm, local()));
    const ab_image_stats *res = nullptr;
    AB_TRY(ab_stats_enqueue(ctx, comm, band->data, band->rows * band->cols, total_rows * band->cols, 0, 0.0, 0.0, nullptr, &res, nullptr, nullptr));
    return fetch_result(ctx, res, out, nullptr, comm);
} AB_CATCH(ctx)

// auto_stretch_preview (cmd/common.rs:18-22): compute_image_stats -> auto_stf(default config) -> apply_stf, as one
// asynchronous chain: the STF kernel reads its transform from the state block the statistics chain leaves in HBM.
// out_u8_dev: rows x cols bytes on the device.  out_stats / out_stf (nullable): fetched at the end (one synchronisation;
// pass both NULL to keep the call fully asynchronous).  With a communicator `img` is this rank's row band of an image of
// total_rows rows and the statistics are those of the whole image (each rank stretches its own band).
int ab_auto_stretch_preview(ab_ctx *ctx, ab_comm *comm, const ab_plane *img, int64_t total_rows, const ab_auto_stf_config *cfg,
                            uint8_t *out_u8_dev, ab_image_stats *out_stats, ab_stf_params *out_stf) try {
    if (!ctx) return AB_ERR_INVALID;
    auto local = [&]() -> int {
        AB_CHECK(ctx, img && img->on_device && img->rows >= 0 && img->cols > 0, "auto_stretch_preview takes a device plane");
        AB_CHECK(ctx, comm ? (img->rows == 0 || (img->data && out_u8_dev)) : (img->rows > 0 && img->data && out_u8_dev),
                 "auto_stretch_preview takes a non-empty device plane and a device output (an empty row band only with a communicator)");
        if (total_rows <= 0) total_rows = img->rows;
        AB_CHECK(ctx, total_rows >= img->rows && total_rows > 0, "total_rows is smaller than the band");
        AB_HIP(ctx, hipSetDevice(ctx->device));
        Ws w;
        return carve(ctx, &w);
    };
    AB_TRY(ab_comm_agree(ctx, comm, local()));  // (no communicator: returns the local status)
    const ab_image_stats *res = nullptr;
    const void *tx = nullptr;
    // (a caller that does not fetch gets the chain: nobody would see the resident kernel's abort flag)
    const bool fetch = out_stats || out_stf;
    bool resident = false, aborted = false;
    ResidentGuard guard{ctx, &resident};
    AB_TRY(stats_enqueue(ctx, comm, img->data, img->rows * img->cols, total_rows * img->cols, 0, 0.0, 0.0, cfg, &res, &tx, nullptr, out_u8_dev,
                         fetch ? &resident : nullptr));
    if (!resident && img->rows > 0) AB_TRY(ab_stf_u8_device_tx(ctx, img->data, img->rows * img->cols, tx, out_u8_dev));
    if (!fetch) return AB_OK;
    AB_TRY(fetch_result(ctx, res, out_stats, out_stf, comm, resident ? &aborted : nullptr));
    if (!aborted) return AB_OK;
    AB_TRY(ab_stats_enqueue(ctx, comm, img->data, img->rows * img->cols, total_rows * img->cols, 0, 0.0, 0.0, cfg, &res, &tx, nullptr));
    AB_TRY(ab_stf_u8_device_tx(ctx, img->data, img->rows * img->cols, tx, out_u8_dev));
    return fetch_result(ctx, res, out_stats, out_stf, comm);
} AB_CATCH(ctx)

int ab_stats_value_hist(ab_ctx *ctx, const ab_plane *img, double gmin, double gmax, uint64_t *hist65536_host,
                        double *out_sum, uint64_t *out_cnt) try {
    if (!ctx) return AB_ERR_INVALID;
    AB_CHECK(ctx, img && hist65536_host, "null plane or output");
    AB_HIP(ctx, hipSetDevice(ctx->device));
    StagedPlane in;
    AB_TRY(ab_stage_in(ctx, img, &in));
    Ws w;
    int rc = carve(ctx, &w);
    if (rc == AB_OK) {
        const int64_t n = in.rows * in.cols;
        int grid = 0;
        hipLaunchKernelGGL(set_range_kernel, dim3(1), dim3(1), 0, ctx->stream, w.st, gmin, gmax);
        hipLaunchKernelGGL(book_range_kernel, dim3(1), dim3(1), 0, ctx->stream, w.st);
        hipError_t e = hipMemsetAsync(w.H, 0, (size_t)(1 + kHistBins) * sizeof(unsigned long long), ctx->stream);
        if (e == hipSuccess) rc = launch_dense<HIST_VALUE>(ctx, in.dptr, n, w, &grid);
        if (e == hipSuccess && rc == AB_OK) {
            hipLaunchKernelGGL(finish_hist_kernel, dim3(1), dim3(kBookBlock), 0, ctx->stream, w.partials, grid, w.st, w.H, 1);
            e = hipMemcpyAsync(hist65536_host, w.H + 1, kHistBins * sizeof(uint64_t), hipMemcpyDeviceToHost, ctx->stream);
            unsigned long long cnt = 0;
            double sum = 0.0;
            if (e == hipSuccess) e = hipMemcpyAsync(&cnt, w.H, sizeof cnt, hipMemcpyDeviceToHost, ctx->stream);
            if (e == hipSuccess) e = hipMemcpyAsync(&sum, &w.st->sum, sizeof sum, hipMemcpyDeviceToHost, ctx->stream);
            if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
            if (out_sum) *out_sum = sum;
            if (out_cnt) *out_cnt = cnt;
        }
        if (e != hipSuccess) rc = ab_set_error(ctx, AB_ERR_HIP, "stats_value_hist: %s", hipGetErrorString(e));
    }
    ab_stage_release(ctx, &in);
    return rc;
} AB_CATCH(ctx)

int ab_build_histogram(ab_ctx *ctx, const ab_plane *img, size_t bins, double dmin, double dmax, uint32_t *out_bins_host) try {
    if (!ctx) return AB_ERR_INVALID;
    AB_CHECK(ctx, img && out_bins_host && bins >= 1, "null plane/output or zero bins");
    memset(out_bins_host, 0, bins * sizeof(uint32_t));
    const double range = dmax - dmin;
    if (range < 1e-10) return AB_OK;  // stats.rs:380-387
    AB_HIP(ctx, hipSetDevice(ctx->device));
    StagedPlane in;
    AB_TRY(ab_stage_in(ctx, img, &in));
    void *dev = nullptr;
    int rc = ab_scratch(ctx, bins * sizeof(unsigned int), &dev);
    if (rc == AB_OK) {
        const int64_t n = in.rows * in.cols;
        hipError_t e = hipMemsetAsync(dev, 0, bins * sizeof(unsigned int), ctx->stream);
        if (e == hipSuccess) {
            const int grid = grid_for(ctx, n, kScanBlock, 8);
            const size_t lds_bytes = bins <= 16384 ? bins * sizeof(unsigned int) : 16;
            hipLaunchKernelGGL(small_hist_kernel, dim3(grid), dim3(kScanBlock), lds_bytes, ctx->stream, in.dptr, n,
                               (uint32_t)bins, dmin, (double)bins / range, (unsigned int *)dev);
            e = hipGetLastError();
        }
        if (e == hipSuccess) e = hipMemcpyAsync(out_bins_host, dev, bins * sizeof(unsigned int), hipMemcpyDeviceToHost, ctx->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
        if (e != hipSuccess) rc = ab_set_error(ctx, AB_ERR_HIP, "build_histogram: %s", hipGetErrorString(e));
    }
    ab_stage_release(ctx, &in);
    return rc;
} AB_CATCH(ctx)

}  // extern "C"
