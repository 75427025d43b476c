// Per-pixel kappa-sigma stacking of MORE than 4096 frames on gfx950: one WORKGROUP per pixel, samples in global scratch.
//
// sigma_clip_combine (core/stacking/combine.rs:14-92) and median_combine_row_major (calibration.rs:84-125) take however many frames
// the caller has: stack_images gathers a Vec per pixel (combine.rs:160-182).  Up to 64 frames a pixel lives in one lane's
// registers (stack_sigma_clip.hip), up to 512 in two lanes' (stack_pair.hip), up to 4096 in one wave's (stack_wide.hip).  Beyond
// that -- frames of at most a few megapixels, or they would not fit the HBM -- this kernel keeps the same DEFINITION with no
// limit but the scratch: the finite samples are sorted in a per-workgroup segment (block_sort.hpp), survivors are rank
// intervals of the sorted run, and every f64 sum walks its interval in ascending order one addition at a time, which is what the
// oracle's ORC_ORDER_ASCENDING does and what the narrower kernels reproduce: bit-identical to them.  A fallback, not a roofline
// kernel: ~log2(N)^2 / 2 barriers per sort and serial sums.
#include "ab_common.hpp"
#include "block_sort.hpp"

#include <algorithm>
#include <cmath>

namespace {

constexpr double kMadToSigma = 1.4826;  // types/constants.rs:7
constexpr int kRejSlots = AB_REJ_SLOTS;

struct DeepArgs {
    const float *const *p;  // n plane pointers (device array)
    const int64_t *ld;      // n row strides
    int n, np2, contiguous;
    int64_t rows, cols;
    float sigma_low, sigma_high;
    uint32_t max_iter;
    float *out;       // full mode
    double *out_sum;  // partial mode
    uint32_t *out_cnt;
    unsigned long long *rejected;
    int median_only;
    float *scratch;  // gridDim.x segments of 2 * np2 floats: the sorted samples, the sorted deviations
};

__global__ __launch_bounds__(256) void stack_deep_kernel(const DeepArgs a) {
    __shared__ int sh_cnt[2];
    __shared__ double sh_d[2];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    float *X = a.scratch + (size_t)blockIdx.x * 2 * a.np2, *D = X + a.np2;
    const int64_t total = a.rows * a.cols;
    unsigned long long rej_total = 0;  // (thread 0's is the one that counts)
    for (int64_t g = blockIdx.x; g < total; g += gridDim.x) {
        int64_t y = 0, xcol = g;
        if (!a.contiguous) {
            y = g / a.cols;
            xcol = g - y * a.cols;
        }
        if (tid < 2) sh_cnt[tid] = 0;
        __syncthreads();
        // ---- gather (combine.rs:170-175): only finite samples take part; the rest are +inf pads on top of the order ----
        int fin = 0;
        for (int f = tid; f < a.np2; f += 256) {
            float s = __builtin_inff();
            if (f < a.n) {
                const float v = a.p[f][a.contiguous ? g : (y * a.ld[f] + xcol)];
                if (__builtin_isfinite(v)) {
                    s = v;
                    ++fin;
                }
            }
            X[f] = s;
        }
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) fin += __shfl_xor(fin, off, 64);
        if (lane == 0 && fin) atomicAdd(&sh_cnt[0], fin);
        __syncthreads();
        const int n = sh_cnt[0];  // finite samples: sorted ranks [0, n)
        block_bitonic_sort(X, a.np2);

        float value = 0.0f;
        double S = 0.0;
        int len = n;
        uint32_t rej = 0;
        if (n == 1) {
            value = X[0];  // combine.rs:24-26
            S = (double)value;
        } else if (n >= 2) {
            const float med = X[n >> 1];  // combine.rs:38-40
            if (a.median_only) {
                value = med;
            } else {
                // MAD (combine.rs:42-46): the n/2-th smallest |v - med|
                for (int e = tid; e < a.np2; e += 256) D[e] = e < n ? fabsf(X[e] - med) : __builtin_inff();
                block_bitonic_sort(D, a.np2);
                const float mad = D[n >> 1];
                float sigma = (float)fmax((double)mad * kMadToSigma, 1e-10);
                float center = med, last_center = __builtin_nanf("");
                int lo_r = 0, hi_r = n - 1;  // survivors = sorted ranks lo_r..hi_r (`v - center` is monotone in v)
                for (uint32_t it = 0; it < a.max_iter; ++it) {
                    if (len < 2) break;  // combine.rs:33-35
                    if (it > 0) {        // mean / sample variance of the survivors, f64, ascending (combine.rs:50-60)
                        __syncthreads();
                        if (wave == 0) {
                            const double nn = (double)len;
                            const double mean = wave_serial_sum_f64<0>(X, lo_r, hi_r, 0.0, lane) / nn;
                            const double q = wave_serial_sum_f64<1>(X, lo_r, hi_r, mean, lane);
                            if (lane == 0) {
                                sh_d[0] = mean;
                                sh_d[1] = q / fmax(nn - 1.0, 1.0);
                            }
                        }
                        __syncthreads();
                        center = (float)sh_d[0];
                        sigma = (float)fmax(sqrt(sh_d[1]), 1e-10);
                    }
                    last_center = center;                   // combine.rs:63
                    const float lo = -a.sigma_low * sigma;  // combine.rs:65-66
                    const float hi = a.sigma_high * sigma;
                    __syncthreads();
                    if (tid < 2) sh_cnt[tid] = 0;
                    __syncthreads();
                    int cl = 0, ch = 0;
                    for (int e = lo_r + tid; e <= hi_r; e += 256) {
                        const float dev = X[e] - center;
                        cl += !(dev >= lo) ? 1 : 0;
                        ch += !(dev <= hi) ? 1 : 0;
                    }
#pragma unroll
                    for (int off = 32; off >= 1; off >>= 1) {
                        cl += __shfl_xor(cl, off, 64);
                        ch += __shfl_xor(ch, off, 64);
                    }
                    if (lane == 0) {
                        if (cl) atomicAdd(&sh_cnt[0], cl);
                        if (ch) atomicAdd(&sh_cnt[1], ch);
                    }
                    __syncthreads();
                    cl = sh_cnt[0];
                    ch = sh_cnt[1];
                    // a sample can fail both tests only if nothing survives (lo > hi or NaN thresholds)
                    const int removed = (cl + ch > len) ? len : (cl + ch);
                    rej += (uint32_t)removed;  // combine.rs:76-78
                    len -= removed;
                    if (len > 0) {
                        lo_r += cl;
                        hi_r -= ch;
                    } else {
                        lo_r = 1;
                        hi_r = 0;
                    }
                    if (removed == 0) break;  // combine.rs:80-82
                }
                if (len == 0) {  // combine.rs:85-88
                    value = __builtin_isfinite(last_center) ? last_center : 0.0f;
                } else {
                    __syncthreads();
                    if (wave == 0) {
                        const double s = wave_serial_sum_f64<0>(X, lo_r, hi_r, 0.0, lane);  // combine.rs:90-91
                        if (lane == 0) sh_d[0] = s;
                    }
                    __syncthreads();
                    S = sh_d[0];
                    value = (float)(S / (double)len);
                }
            }
        }
        if (tid == 0) {
            if (a.out_sum) {
                a.out_sum[g] = len > 0 ? S : 0.0;
                a.out_cnt[g] = (uint32_t)(len > 0 ? len : 0);
            } else {
                a.out[g] = value;
            }
            rej_total += rej;
        }
        __syncthreads();  // the next pixel reuses the segment and the shared words
    }
    if (tid == 0 && rej_total) atomicAdd(&a.rejected[blockIdx.x & (kRejSlots - 1)], rej_total);
}

}  // namespace

// dplanes / ld are HOST arrays of n entries (any n >= 2; the dispatcher sends n > 4096 here, AB_STACK_DEEP_FROM=k everything above k
// frames -- the tests hold this kernel to the oracle at sizes the narrower ones cover too); counters already cleared by the caller
int ab_stack_deep_device(ab_ctx *ctx, const float *const *dplanes, const int64_t *ld, size_t n, int64_t rows, int64_t cols,
                         const ab_stack_config *cfg, float *out_dev, double *out_sum_dev, uint32_t *out_cnt_dev, bool median_only) {
    AB_CHECK(ctx, n >= 2 && n <= ((size_t)1 << 24), "the workgroup-per-pixel stack takes 2 .. 2^24 frames (got %zu)", n);
    int np2 = 2;
    while ((size_t)np2 < n) np2 <<= 1;
    const int64_t total = rows * cols;
    // one segment of 2 np2 floats per workgroup; as many workgroups as fit 1 GiB of scratch, at most four per compute unit
    const int64_t seg_bytes = (int64_t)2 * np2 * (int64_t)sizeof(float);
    int64_t grid = std::min<int64_t>(total, (int64_t)(ctx->cu_count > 0 ? ctx->cu_count : 256) * 4);
    grid = std::max<int64_t>(1, std::min<int64_t>(grid, ((int64_t)1 << 30) / seg_bytes));
    const size_t tab_bytes = ((n * (sizeof(float *) + sizeof(int64_t)) + 255) / 256) * 256;
    char *ws = nullptr;
    AB_TRY(ab_workspace(ctx, AB_WS_STACK_DEEP, tab_bytes + (size_t)grid * (size_t)seg_bytes, (void **)&ws));
    AB_HIP(ctx, hipStreamSynchronize(ctx->stream));
    AB_HIP(ctx, hipMemcpy(ws, dplanes, n * sizeof(float *), hipMemcpyHostToDevice));
    AB_HIP(ctx, hipMemcpy(ws + n * sizeof(float *), ld, n * sizeof(int64_t), hipMemcpyHostToDevice));
    DeepArgs a;
    a.p = (const float *const *)ws;
    a.ld = (const int64_t *)(ws + n * sizeof(float *));
    a.n = (int)n;
    a.np2 = np2;
    a.contiguous = 1;
    for (size_t i = 0; i < n; ++i)
        if (ld[i] != cols) a.contiguous = 0;
    a.rows = rows;
    a.cols = cols;
    a.sigma_low = cfg->sigma_low;
    a.sigma_high = cfg->sigma_high;
    a.max_iter = cfg->max_iterations;
    a.out = out_dev;
    a.out_sum = out_sum_dev;
    a.out_cnt = out_cnt_dev;
    a.rejected = ctx->counters;
    a.median_only = median_only ? 1 : 0;
    a.scratch = (float *)(ws + tab_bytes);
    hipLaunchKernelGGL(stack_deep_kernel, dim3((unsigned)grid), dim3(256), 0, ctx->stream, a);
    AB_HIP(ctx, hipGetLastError());
    return AB_OK;
}
