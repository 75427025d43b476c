// Whole-image statistics on gfx950.
//
// Replaces core/imaging/stats.rs: compute_image_stats (:15-23), the exact path (:43-73,
// math/median.rs:27-73), the 65 536-bin histogram path for > 4 000 000 px (:75-210), the known-
// range variant (:25-41) and build_histogram (:378-421).
//
// Design (HBM-bound integer/histogram work -- no GEMM shapes here):
//   * every pass is one grid-stride streaming read of the plane with float4 loads.
//   * 65 536-bin histograms are privatised per workgroup in LDS as PACKED 16-bit counters
//     (65 536 x u16 = 128 KiB of the CU's 160 KiB).  A workgroup consumes at most 65 535 pixels
//     between flushes, so a counter cannot overflow; a flush adds the non-zero counters to the
//     global u64 histogram.  Bin indices are computed in f64 exactly as the reference does
//     (`((v as f64 - min) * inv) as usize`, saturating), so every bin count is bit-exact.
//   * sparse histograms (the refinement passes touch one coarse bin's worth of pixels) go
//     straight to global 64-bit atomics.
//   * the exact path (<= 4M px) needs order statistics, not a sort: valid pixels are positive
//     finite floats, whose bit patterns are monotone as u32, so rank k is found by an 11/11/10-bit
//     radix select (three small histogram passes), the same for |v - median|.
//   * the scalar bookkeeping between passes (percentile bin, in-bin interpolation) runs on the
//     host on the downloaded histograms, transcribing stats.rs:302-353.
//   * f64 sums: per-thread sequential over a strided slice, then a fixed-shape tree; the
//     reference's own order is rayon's unspecified reduce tree (stats.rs:252-257).
#include "ab_common.hpp"

#include <algorithm>
#include <cfloat>
#include <cmath>

namespace {

constexpr float kPaddingThreshold = 1e-7f;  // types/constants.rs:6
constexpr double kMadToSigma = 1.4826;      // types/constants.rs:7
constexpr int kHistBins = 65536;            // stats.rs:8
constexpr int kExactLimit = 4000000;        // stats.rs:18

__device__ __forceinline__ bool is_valid_pixel(float v) {  // stats.rs:10-13
    return __builtin_isfinite(v) && v > kPaddingThreshold;
}

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

__device__ __forceinline__ void block_reduce_scan(double mn, double mx, double sum, unsigned long long cnt,
                                                  ScanPartial *out) {
    __shared__ double s_mn[kScanBlock / 64], s_mx[kScanBlock / 64], s_sum[kScanBlock / 64];
    __shared__ unsigned long long s_cnt[kScanBlock / 64];
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        mn = fmin(mn, __shfl_xor(mn, off, 64));
        mx = fmax(mx, __shfl_xor(mx, off, 64));
        sum += __shfl_xor(sum, off, 64);
        cnt += __shfl_xor(cnt, off, 64);
    }
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) {
        s_mn[w] = mn;
        s_mx[w] = mx;
        s_sum[w] = sum;
        s_cnt[w] = cnt;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int i = 1; i < kScanBlock / 64; ++i) {
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

// stats.rs:212-258: min / max / sum / count over valid pixels; one partial per workgroup
__global__ __launch_bounds__(kScanBlock) void scan_kernel(const float *__restrict__ data, int64_t n,
                                                          ScanPartial *__restrict__ partials) {
    double mn = DBL_MAX, mx = -DBL_MAX, sum = 0.0;
    unsigned long long cnt = 0;
    const int64_t stride = (int64_t)gridDim.x * kScanBlock;
    for (int64_t i = (int64_t)blockIdx.x * kScanBlock + threadIdx.x; i < n; i += stride) {
        const float v = data[i];
        if (is_valid_pixel(v)) {
            const double vf = (double)v;
            mn = fmin(mn, vf);
            mx = fmax(mx, vf);
            sum += vf;
            cnt += 1;
        }
    }
    block_reduce_scan(mn, mx, sum, cnt, &partials[blockIdx.x]);
}

// ---- dense 65 536-bin histogram, LDS-privatised with packed u16 counters -------------------
constexpr int kHistBlock = 1024;
constexpr int kHistChunk = 61440;  // pixels per workgroup between flushes (< 65 536)

enum HistKind { HIST_VALUE = 0, HIST_DEV = 1 };

struct HistArgs {
    const float *data;
    int64_t n;
    double origin;   // value hist: data_min          dev hist: unused
    double inv;      // bins / range
    float center;    // dev hist: coarse median as f32 (stats.rs:114,131)
    unsigned long long *hist;   // 65 536 x u64, zeroed by the caller
    ScanPartial *partials;      // value hist only: per-workgroup sum/count (stats.rs:279-281)
    // sparse side histogram taken in the same pass (stats.rs:127-130): refine of the median bin
    int want_refine;
    double refine_lo, refine_hi, refine_inv;
    unsigned long long *refine;  // 65 536 x u64
};

template <int KIND>
__global__ __launch_bounds__(kHistBlock) void dense_hist_kernel(const HistArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned int lds[];  // 32 768 words = 65 536 x u16
    for (int i = threadIdx.x; i < kHistBins / 2; i += kHistBlock) lds[i] = 0;
    __syncthreads();

    double sum = 0.0;
    unsigned long long cnt = 0;
    const uint32_t last = kHistBins - 1;
    const int64_t nchunks = (a.n + kHistChunk - 1) / kHistChunk;
    for (int64_t chunk = blockIdx.x; chunk < nchunks; chunk += gridDim.x) {
        const int64_t base = chunk * kHistChunk;
        const int64_t end = (base + kHistChunk < a.n) ? base + kHistChunk : a.n;
        for (int64_t i = base + threadIdx.x; i < end; i += kHistBlock) {
            const float v = a.data[i];
            if (is_valid_pixel(v)) {
                uint32_t idx;
                if (KIND == HIST_VALUE) {
                    const double vf = (double)v;
                    sum += vf;
                    cnt += 1;
                    idx = bin_index((vf - a.origin) * a.inv, last);  // stats.rs:282-283
                    if (a.want_refine && vf >= a.refine_lo && vf < a.refine_hi) {  // stats.rs:127-130
                        const uint32_t r = bin_index((vf - a.refine_lo) * a.refine_inv, last);
                        atomicAdd(&a.refine[r], 1ull);
                    }
                } else {
                    const double vf = (double)v;
                    if (a.want_refine && vf >= a.refine_lo && vf < a.refine_hi) {
                        const uint32_t r = bin_index((vf - a.refine_lo) * a.refine_inv, last);
                        atomicAdd(&a.refine[r], 1ull);
                    }
                    const float d = fabsf(v - a.center);              // stats.rs:131-133
                    idx = bin_index((double)d * a.inv, last);
                }
                atomicAdd(&lds[idx >> 1], (idx & 1) ? 0x10000u : 1u);
            }
        }
        __syncthreads();
        // flush: non-zero packed counters -> global u64 bins, then clear
        for (int w = threadIdx.x; w < kHistBins / 2; w += kHistBlock) {
            const unsigned int packed = lds[w];
            if (packed) {
                const unsigned int lo = packed & 0xffffu, hi = packed >> 16;
                if (lo) atomicAdd(&a.hist[2 * w], (unsigned long long)lo);
                if (hi) atomicAdd(&a.hist[2 * w + 1], (unsigned long long)hi);
                lds[w] = 0;
            }
        }
        __syncthreads();
    }
    if (KIND == HIST_VALUE && a.partials) {
        // block reduction of sum / count (fixed tree)
        __shared__ double s_sum[kHistBlock / 64];
        __shared__ unsigned long long s_cnt[kHistBlock / 64];
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) {
            sum += __shfl_xor(sum, off, 64);
            cnt += __shfl_xor(cnt, off, 64);
        }
        if ((threadIdx.x & 63) == 0) {
            s_sum[threadIdx.x >> 6] = sum;
            s_cnt[threadIdx.x >> 6] = cnt;
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            for (int i = 1; i < kHistBlock / 64; ++i) {
                sum += s_sum[i];
                cnt += s_cnt[i];
            }
            a.partials[blockIdx.x].sum = sum;
            a.partials[blockIdx.x].cnt = cnt;
            a.partials[blockIdx.x].mn = 0.0;
            a.partials[blockIdx.x].mx = 0.0;
        }
    }
}

// stats.rs:166-191: count of deviations below the MAD region + sparse refine of the region
__global__ __launch_bounds__(kScanBlock) void mad_refine_kernel(const float *__restrict__ data, int64_t n, float med_f32,
                                                                float lo_f32, float hi_f32, double region_lo, double inv,
                                                                unsigned long long *__restrict__ refine,
                                                                ScanPartial *__restrict__ partials) {
    unsigned long long below = 0;
    const uint32_t last = kHistBins - 1;
    const int64_t stride = (int64_t)gridDim.x * kScanBlock;
    for (int64_t i = (int64_t)blockIdx.x * kScanBlock + threadIdx.x; i < n; i += stride) {
        const float v = data[i];
        if (is_valid_pixel(v)) {
            const float dev = fabsf(v - med_f32);
            if (dev < lo_f32) {
                below += 1;
            } else if (dev < hi_f32) {
                const uint32_t r = bin_index(((double)dev - region_lo) * inv, last);
                atomicAdd(&refine[r], 1ull);
            }
        }
    }
    block_reduce_scan(0.0, 0.0, 0.0, below, &partials[blockIdx.x]);
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
// keys: u32 bit pattern of v (valid pixels, positive) or of |v - center| (>= +0); both monotone.
struct SelectArgs {
    const float *data;
    int64_t n;
    int use_dev;     // 0: key = bits(v)   1: key = bits(|v - center|)
    float center;
    uint32_t prefix_mask, prefix_val;  // only keys with (key & mask) == val are counted
    int shift, nbits;
    unsigned int *hist;  // 2^nbits bins, zeroed
};

__global__ __launch_bounds__(kScanBlock) void select_hist_kernel(const SelectArgs a) {
    __shared__ unsigned int lds[2048];
    const uint32_t nb = 1u << a.nbits;
    for (uint32_t i = threadIdx.x; i < nb; i += kScanBlock) lds[i] = 0;
    __syncthreads();
    const int64_t stride = (int64_t)gridDim.x * kScanBlock;
    for (int64_t i = (int64_t)blockIdx.x * kScanBlock + threadIdx.x; i < a.n; i += stride) {
        const float v = a.data[i];
        if (is_valid_pixel(v)) {
            const float k = a.use_dev ? fabsf(v - a.center) : v;
            const uint32_t key = __float_as_uint(k);
            if ((key & a.prefix_mask) == a.prefix_val) atomicAdd(&lds[(key >> a.shift) & (nb - 1)], 1u);
        }
    }
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < nb; i += kScanBlock)
        if (lds[i]) atomicAdd(&a.hist[i], lds[i]);
}

// ---------------------------------------------------------------------------------------------
int grid_for(ab_ctx *ctx, int64_t n, int block, int per_cu) {
    int64_t want = (n + block - 1) / block;
    int64_t cap = (int64_t)(ctx->cu_count > 0 ? ctx->cu_count : 256) * per_cu;
    return (int)std::max<int64_t>(1, std::min(want, cap));
}

// host transcriptions of stats.rs:302-353
inline uint64_t f64_to_u64_sat(double v) {
    if (!(v > 0.0)) return 0;
    if (v >= 18446744073709551615.0) return UINT64_MAX;
    return (uint64_t)v;
}
size_t find_percentile_bin(const unsigned long long *hist, size_t nb, uint64_t total, double pct) {
    const uint64_t target = f64_to_u64_sat(std::ceil((double)total * pct));
    uint64_t cum = 0;
    for (size_t i = 0; i < nb; ++i) {
        cum += hist[i];
        if (cum >= target) return i;
    }
    return nb - 1;
}
double interpolate_percentile(const unsigned long long *hist, size_t nb, uint64_t total, double pct, double data_min,
                              double bin_width) {
    const uint64_t target = f64_to_u64_sat(std::ceil((double)total * pct));
    uint64_t cum = 0;
    for (size_t i = 0; i < nb; ++i) {
        const uint64_t count = hist[i];
        cum += count;
        if (cum >= target) {
            const uint64_t overshoot = cum - target;
            const double frac = count > 0 ? 1.0 - ((double)overshoot / (double)count) : 0.5;
            return data_min + ((double)i + frac) * bin_width;
        }
    }
    return data_min + (double)nb * bin_width;
}
double resolve_rank_in_hist(const unsigned long long *hist, size_t nb, uint64_t rank, double region_lo,
                            double sub_bin_width) {
    if (rank == 0) return region_lo;
    uint64_t cum = 0;
    for (size_t i = 0; i < nb; ++i) {
        const uint64_t count = hist[i];
        cum += count;
        if (cum >= rank) {
            const uint64_t overshoot = cum - rank;
            const double frac = count > 0 ? 1.0 - ((double)overshoot / (double)count) : 0.5;
            return region_lo + ((double)i + frac) * sub_bin_width;
        }
    }
    return region_lo + (double)nb * sub_bin_width;
}

struct DeviceHists {  // carved from the context scratch arena
    unsigned long long *h0, *h1;  // 2 x 65 536 u64
    ScanPartial *partials;        // up to kMaxPartials
    unsigned int *sel;            // 2048 u32
};
constexpr int kMaxPartials = 4096;

int carve(ab_ctx *ctx, DeviceHists *d) {
    const size_t bytes = 2 * kHistBins * sizeof(unsigned long long) + kMaxPartials * sizeof(ScanPartial) + 2048 * 4;
    void *p = nullptr;
    AB_TRY(ab_scratch(ctx, bytes, &p));
    char *c = (char *)p;
    d->h0 = (unsigned long long *)c;
    c += kHistBins * sizeof(unsigned long long);
    d->h1 = (unsigned long long *)c;
    c += kHistBins * sizeof(unsigned long long);
    d->partials = (ScanPartial *)c;
    c += kMaxPartials * sizeof(ScanPartial);
    d->sel = (unsigned int *)c;
    return AB_OK;
}

// D2H through the context's pinned buffer: a pageable destination makes the runtime bounce the copy through its own
// staging pages (~100 us per 512 KiB histogram, four of them per compute_image_stats)
int download(ab_ctx *ctx, void *dst, const void *src, size_t bytes) {
    void *pin = nullptr;
    AB_TRY(ab_pinned(ctx, bytes, &pin));
    AB_HIP(ctx, hipMemcpyAsync(pin, src, bytes, hipMemcpyDeviceToHost, ctx->stream));
    AB_HIP(ctx, hipStreamSynchronize(ctx->stream));
    memcpy(dst, pin, bytes);
    return AB_OK;
}

int scan(ab_ctx *ctx, const float *data, int64_t n, const DeviceHists &d, double *mn, double *mx, double *sum,
         uint64_t *cnt) {
    const int grid = std::min(grid_for(ctx, n, kScanBlock, 8), kMaxPartials);
    hipLaunchKernelGGL(scan_kernel, dim3(grid), dim3(kScanBlock), 0, ctx->stream, data, n, d.partials);
    AB_HIP(ctx, hipGetLastError());
    std::vector<ScanPartial> host(grid);
    AB_TRY(download(ctx, host.data(), d.partials, grid * sizeof(ScanPartial)));
    double a = DBL_MAX, b = -DBL_MAX, s = 0.0;
    uint64_t c = 0;
    for (int i = 0; i < grid; ++i) {
        a = std::fmin(a, host[i].mn);
        b = std::fmax(b, host[i].mx);
        s += host[i].sum;
        c += host[i].cnt;
    }
    *mn = a;
    *mx = b;
    *sum = s;
    *cnt = c;
    return AB_OK;
}

// rank-k order statistic (0-based) of the keys, by 11/11/10-bit radix select
int radix_select(ab_ctx *ctx, const float *data, int64_t n, int use_dev, float center, uint64_t rank,
                 const DeviceHists &d, float *out) {
    const int shifts[3] = {21, 10, 0};
    const int bits[3] = {11, 11, 10};
    uint32_t prefix_mask = 0, prefix_val = 0;
    std::vector<unsigned int> host(2048);
    const int grid = grid_for(ctx, n, kScanBlock, 8);
    for (int pass = 0; pass < 3; ++pass) {
        const uint32_t nb = 1u << bits[pass];
        AB_HIP(ctx, hipMemsetAsync(d.sel, 0, nb * sizeof(unsigned int), ctx->stream));
        SelectArgs a;
        a.data = data;
        a.n = n;
        a.use_dev = use_dev;
        a.center = center;
        a.prefix_mask = prefix_mask;
        a.prefix_val = prefix_val;
        a.shift = shifts[pass];
        a.nbits = bits[pass];
        a.hist = d.sel;
        hipLaunchKernelGGL(select_hist_kernel, dim3(grid), dim3(kScanBlock), 0, ctx->stream, a);
        AB_HIP(ctx, hipGetLastError());
        AB_TRY(download(ctx, host.data(), d.sel, nb * sizeof(unsigned int)));
        uint64_t cum = 0;
        uint32_t bin = nb - 1;
        for (uint32_t i = 0; i < nb; ++i) {
            if (cum + host[i] > rank) {
                bin = i;
                break;
            }
            cum += host[i];
        }
        rank -= cum;
        prefix_val |= bin << shifts[pass];
        prefix_mask |= (nb - 1) << shifts[pass];
    }
    float f;
    memcpy(&f, &prefix_val, sizeof f);
    *out = f;
    return AB_OK;
}

// stats.rs:43-73 with math/median.rs:27-73
int stats_exact(ab_ctx *ctx, const float *data, int64_t n, const DeviceHists &d, ab_image_stats *out) {
    double mn, mx, sum;
    uint64_t cnt;
    AB_TRY(scan(ctx, data, n, d, &mn, &mx, &sum, &cnt));
    if (cnt == 0) {
        memset(out, 0, sizeof *out);
        return AB_OK;
    }
    const double mean = sum / (double)cnt;
    const uint64_t mid = cnt / 2;
    // exact_median_mut (median.rs:27-44): even n -> mean of the two middle order statistics in f64
    float right, left = 0.0f;
    AB_TRY(radix_select(ctx, data, n, 0, 0.0f, mid, d, &right));
    double median;
    if (cnt % 2 == 0) {
        AB_TRY(radix_select(ctx, data, n, 0, 0.0f, mid - 1, d, &left));
        median = ((double)left + (double)right) / 2.0;
    } else {
        median = (double)right;
    }
    // exact_mad_mut(valid, median as f32) -> median_f32_mut of |v - med| (median.rs:46-73), f32 average
    const float med_f32 = (float)median;
    float dr, dl = 0.0f;
    AB_TRY(radix_select(ctx, data, n, 1, med_f32, mid, d, &dr));
    float mad_f32;
    if (cnt % 2 == 0) {
        AB_TRY(radix_select(ctx, data, n, 1, med_f32, mid - 1, d, &dl));
        mad_f32 = (dl + dr) / 2.0f;
    } else {
        mad_f32 = dr;
    }
    const double mad = (double)mad_f32;
    out->min = mn;
    out->max = mx;
    out->mean = mean;
    out->median = median;
    out->mad = mad;
    out->sigma = std::fmax(mad * kMadToSigma, 1e-30);
    out->valid_count = cnt;
    return AB_OK;
}

int launch_dense_hist(ab_ctx *ctx, int kind, const HistArgs &a, int *grid_out) {
    const int64_t nchunks = (a.n + kHistChunk - 1) / kHistChunk;
    const int grid = (int)std::max<int64_t>(1, std::min<int64_t>(nchunks, std::min(ctx->cu_count > 0 ? ctx->cu_count : 256, kMaxPartials)));
    const size_t lds_bytes = kHistBins * sizeof(unsigned short);
    if (kind == HIST_VALUE) {
        AB_HIP(ctx, hipFuncSetAttribute((const void *)dense_hist_kernel<HIST_VALUE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
        hipLaunchKernelGGL(dense_hist_kernel<HIST_VALUE>, dim3(grid), dim3(kHistBlock), lds_bytes, ctx->stream, a);
    } else {
        AB_HIP(ctx, hipFuncSetAttribute((const void *)dense_hist_kernel<HIST_DEV>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
        hipLaunchKernelGGL(dense_hist_kernel<HIST_DEV>, dim3(grid), dim3(kHistBlock), lds_bytes, ctx->stream, a);
    }
    AB_HIP(ctx, hipGetLastError());
    if (grid_out) *grid_out = grid;
    return AB_OK;
}

// stats.rs:260-300
int value_hist(ab_ctx *ctx, const float *data, int64_t n, double gmin, double gmax, const DeviceHists &d,
               std::vector<unsigned long long> &hist, double *sum, uint64_t *cnt) {
    const double range = std::fmax(gmax - gmin, 1e-30);
    AB_HIP(ctx, hipMemsetAsync(d.h0, 0, kHistBins * sizeof(unsigned long long), ctx->stream));
    HistArgs a;
    memset(&a, 0, sizeof a);
    a.data = data;
    a.n = n;
    a.origin = gmin;
    a.inv = (double)kHistBins / range;
    a.hist = d.h0;
    a.partials = d.partials;
    int grid = 0;
    AB_TRY(launch_dense_hist(ctx, HIST_VALUE, a, &grid));
    hist.resize(kHistBins);
    std::vector<ScanPartial> parts(grid);
    AB_HIP(ctx, hipMemcpyAsync(hist.data(), d.h0, kHistBins * sizeof(unsigned long long), hipMemcpyDeviceToHost, ctx->stream));
    AB_TRY(download(ctx, parts.data(), d.partials, grid * sizeof(ScanPartial)));
    double s = 0.0;
    uint64_t c = 0;
    for (int i = 0; i < grid; ++i) {
        s += parts[i].sum;
        c += parts[i].cnt;
    }
    *sum = s;
    *cnt = c;
    return AB_OK;
}

// stats.rs:85-210
int stats_hist_core(ab_ctx *ctx, const float *data, int64_t n, double global_min, double global_max,
                    const DeviceHists &d, ab_image_stats *out) {
    const double range = std::fmax(global_max - global_min, 1e-30);
    const double bin_width = range / (double)kHistBins;

    std::vector<unsigned long long> value_hist_h, refine_h(kHistBins), dev_h(kHistBins), mad_refine_h(kHistBins);
    double global_sum;
    uint64_t total_valid;
    AB_TRY(value_hist(ctx, data, n, global_min, global_max, d, value_hist_h, &global_sum, &total_valid));
    if (total_valid == 0) {
        memset(out, 0, sizeof *out);
        return AB_OK;
    }
    const double mean = global_sum / (double)total_valid;
    const uint64_t half_count = f64_to_u64_sat(std::ceil((double)total_valid * 0.5));  // :100

    const size_t median_bin = find_percentile_bin(value_hist_h.data(), kHistBins, total_valid, 0.5);
    uint64_t count_before_median = 0;
    for (size_t i = 0; i < median_bin; ++i) count_before_median += value_hist_h[i];
    const double median_bin_lo = global_min + (double)median_bin * bin_width;
    const double median_bin_hi = median_bin_lo + bin_width;
    const double coarse_median =
        interpolate_percentile(value_hist_h.data(), kHistBins, total_valid, 0.5, global_min, bin_width);

    const double dev_range = range;  // :111-117
    const double dev_bw = dev_range / (double)kHistBins;
    const double dev_inv = (double)kHistBins / dev_range;
    const float coarse_med_f32 = (float)coarse_median;
    const double refine_range = std::fmax(median_bin_hi - median_bin_lo, 1e-30);
    const double refine_inv = (double)kHistBins / refine_range;

    // pass 3 (:119-146): deviation histogram (dense, LDS) + median-bin refinement (sparse, global)
    AB_HIP(ctx, hipMemsetAsync(d.h0, 0, 2 * kHistBins * sizeof(unsigned long long), ctx->stream));
    {
        HistArgs a;
        memset(&a, 0, sizeof a);
        a.data = data;
        a.n = n;
        a.inv = dev_inv;
        a.center = coarse_med_f32;
        a.hist = d.h0;
        a.want_refine = 1;
        a.refine_lo = median_bin_lo;
        a.refine_hi = median_bin_hi;
        a.refine_inv = refine_inv;
        a.refine = d.h1;
        AB_TRY(launch_dense_hist(ctx, HIST_DEV, a, nullptr));
    }
    AB_HIP(ctx, hipMemcpyAsync(dev_h.data(), d.h0, kHistBins * sizeof(unsigned long long), hipMemcpyDeviceToHost, ctx->stream));
    AB_TRY(download(ctx, refine_h.data(), d.h1, kHistBins * sizeof(unsigned long long)));

    const uint64_t median_rank_in_bin = half_count > count_before_median ? half_count - count_before_median : 0;
    const double median_refine_bw = refine_range / (double)kHistBins;
    const double median =
        resolve_rank_in_hist(refine_h.data(), kHistBins, median_rank_in_bin, median_bin_lo, median_refine_bw);

    const size_t mad_bin = find_percentile_bin(dev_h.data(), kHistBins, total_valid, 0.5);  // :154-164
    const size_t expand_lo = mad_bin > 0 ? mad_bin - 1 : 0;
    const size_t expand_hi = std::min<size_t>(mad_bin + 2, kHistBins);
    const double mad_region_lo = (double)expand_lo * dev_bw;
    const double mad_region_hi = (double)expand_hi * dev_bw;
    const float exact_med_f32 = (float)median;
    const double mad_refine_range = std::fmax(mad_region_hi - mad_region_lo, 1e-30);
    const double mad_refine_inv = (double)kHistBins / mad_refine_range;
    const float mad_lo_f32 = (float)mad_region_lo;
    const float mad_hi_f32 = (float)mad_region_hi;

    // pass 4 (:166-191)
    AB_HIP(ctx, hipMemsetAsync(d.h0, 0, kHistBins * sizeof(unsigned long long), ctx->stream));
    const int grid = std::min(grid_for(ctx, n, kScanBlock, 8), kMaxPartials);
    hipLaunchKernelGGL(mad_refine_kernel, dim3(grid), dim3(kScanBlock), 0, ctx->stream, data, n, exact_med_f32, mad_lo_f32,
                       mad_hi_f32, mad_region_lo, mad_refine_inv, d.h0, d.partials);
    AB_HIP(ctx, hipGetLastError());
    std::vector<ScanPartial> parts(grid);
    AB_HIP(ctx, hipMemcpyAsync(mad_refine_h.data(), d.h0, kHistBins * sizeof(unsigned long long), hipMemcpyDeviceToHost, ctx->stream));
    AB_TRY(download(ctx, parts.data(), d.partials, grid * sizeof(ScanPartial)));
    uint64_t count_below = 0;
    for (int i = 0; i < grid; ++i) count_below += parts[i].cnt;

    const uint64_t mad_rank_in_region = half_count > count_below ? half_count - count_below : 0;
    const double mad_refine_bw = mad_refine_range / (double)kHistBins;
    const double mad =
        resolve_rank_in_hist(mad_refine_h.data(), kHistBins, mad_rank_in_region, mad_region_lo, mad_refine_bw);

    out->min = global_min;
    out->max = global_max;
    out->mean = mean;
    out->median = median;
    out->mad = mad;
    out->sigma = std::fmax(mad * kMadToSigma, 1e-30);
    out->valid_count = total_valid;
    return AB_OK;
}

// stats.rs:75-83
int stats_hist(ab_ctx *ctx, const float *data, int64_t n, const DeviceHists &d, ab_image_stats *out) {
    double mn, mx, sum;
    uint64_t cnt;
    AB_TRY(scan(ctx, data, n, d, &mn, &mx, &sum, &cnt));
    if (mn == DBL_MAX) {
        memset(out, 0, sizeof *out);
        return AB_OK;
    }
    return stats_hist_core(ctx, data, n, mn, mx, d, out);
}

}  // namespace

int ab_stats_device(ab_ctx *ctx, const float *data, int64_t n, int use_known, double known_min, double known_max,
                    ab_image_stats *out) {
    AB_HIP(ctx, hipSetDevice(ctx->device));
    DeviceHists d;
    AB_TRY(carve(ctx, &d));
    if (n <= kExactLimit) return stats_exact(ctx, data, n, d, out);  // stats.rs:18-22,32-34
    if (use_known) {
        if (!std::isfinite(known_min) || !std::isfinite(known_max) || known_min >= known_max)
            return stats_hist(ctx, data, n, d, out);  // stats.rs:36-38
        return stats_hist_core(ctx, data, n, known_min, known_max, d, out);
    }
    return stats_hist(ctx, data, n, d, out);
}

extern "C" {

int ab_compute_image_stats(ab_ctx *ctx, const ab_plane *img, ab_image_stats *out) {
    if (!ctx) return AB_ERR_INVALID;
    AB_CHECK(ctx, img && out, "null plane or output");
    StagedPlane in;
    AB_TRY(ab_stage_in(ctx, img, &in));
    const int rc = ab_stats_device(ctx, in.dptr, in.rows * in.cols, 0, 0.0, 0.0, out);
    ab_stage_release(ctx, &in);
    return rc;
}

int ab_compute_image_stats_with_known_range(ab_ctx *ctx, const ab_plane *img, double known_min, double known_max,
                                            ab_image_stats *out) {
    if (!ctx) return AB_ERR_INVALID;
    AB_CHECK(ctx, img && out, "null plane or output");
    StagedPlane in;
    AB_TRY(ab_stage_in(ctx, img, &in));
    const int rc = ab_stats_device(ctx, in.dptr, in.rows * in.cols, 1, known_min, known_max, out);
    ab_stage_release(ctx, &in);
    return rc;
}

int ab_stats_value_hist(ab_ctx *ctx, const ab_plane *img, double gmin, double gmax, uint64_t *hist65536_host,
                        double *out_sum, uint64_t *out_cnt) {
    if (!ctx) return AB_ERR_INVALID;
    AB_CHECK(ctx, img && hist65536_host, "null plane or output");
    AB_HIP(ctx, hipSetDevice(ctx->device));
    StagedPlane in;
    AB_TRY(ab_stage_in(ctx, img, &in));
    DeviceHists d;
    int rc = carve(ctx, &d);
    std::vector<unsigned long long> h;
    double s = 0.0;
    uint64_t c = 0;
    if (rc == AB_OK) rc = value_hist(ctx, in.dptr, in.rows * in.cols, gmin, gmax, d, h, &s, &c);
    if (rc == AB_OK) {
        memcpy(hist65536_host, h.data(), kHistBins * sizeof(uint64_t));
        if (out_sum) *out_sum = s;
        if (out_cnt) *out_cnt = c;
    }
    ab_stage_release(ctx, &in);
    return rc;
}

int ab_build_histogram(ab_ctx *ctx, const ab_plane *img, size_t bins, double dmin, double dmax, uint32_t *out_bins_host) {
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
}

}  // extern "C"
