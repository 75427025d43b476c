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

#define AB_CE(a, b)                         \
    {                                       \
        T lo_ = fminf(v[a], v[b]);          \
        T hi_ = fmaxf(v[a], v[b]);          \
        v[a] = lo_;                         \
        v[b] = hi_;                         \
    }
#include "sortnet_gen.hpp"

namespace {

constexpr int kMaxFrames = 64;
constexpr double kMadToSigma = 1.4826;  // types/constants.rs:7

struct StackArgs {
    const float *p[kMaxFrames];
    int64_t ld[kMaxFrames];  // row stride (= cols of that plane): top-left crop for free
    int n;                   // frames actually present (<= NP)
    int contiguous;          // all ld == cols: linear pixel index is the element offset
    int64_t rows, cols;      // output dims
    float sigma_low, sigma_high;
    uint32_t max_iter;
    float *out;                      // full mode
    double *out_sum;                 // partial mode
    uint32_t *out_cnt;               // partial mode
    unsigned long long *rejected;    // device counter
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

// k-th (k = NP/2) smallest of |v_i - med| for a fully populated sorted vector (n == NP),
// med = v[NP/2].  See the header comment; pairs are (v[p], v[p+m]) with m = NP/2.
template <int NP>
__device__ __forceinline__ float mad_full(const float (&v)[NP], float med) {
    constexpr int m = NP / 2;
    float best = med - v[0];
#pragma unroll
    for (int p = 1; p <= NP - m - 1; ++p) {
        float a = med - v[p];
        float b = v[p + m] - med;
        best = fminf(best, fmaxf(a, b));
    }
    return best;
}

template <int NP, bool PARTIAL>
__global__ __launch_bounds__(256) void stack_sigma_clip_kernel(const StackArgs args) {
    const int64_t total = args.rows * args.cols;
    int64_t g = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const bool valid = g < total;
    if (!valid) g = total - 1;

    int64_t y = 0, x = g;
    if (!args.contiguous) {
        y = g / args.cols;
        x = g - y * args.cols;
    }

    // ---- gather (combine.rs:170-175): only finite samples take part ----
    // Non-finite samples and the slots past args.n become +-inf PADS.  They are split so that the
    // (upper) median of the n finite samples always lands on sorted position NP/2:
    //   c_lo = NP/2 - n/2 pads of -inf below, the rest +inf above.
    // That keeps every register index a compile-time constant (a per-lane `v[n/2]` would turn
    // the sample vector into a runtime-indexed private array, i.e. scratch memory).
    float v[NP];
    int n = 0;
#pragma unroll
    for (int f = 0; f < NP; ++f) {
        float s = __builtin_inff();
        if (f < args.n) {
            const int64_t off = args.contiguous ? g : (y * args.ld[f] + x);
            s = args.p[f][off];
        }
        v[f] = s;
        n += __builtin_isfinite(s) ? 1 : 0;
    }
    const bool full_wave = __all(n == NP);
    const int c_lo = NP / 2 - (n >> 1);
    if (!full_wave) {
        int k = 0;  // running index among this pixel's pads
#pragma unroll
        for (int f = 0; f < NP; ++f) {
            const bool fin = __builtin_isfinite(v[f]);
            const float pad = (k < c_lo) ? -__builtin_inff() : __builtin_inff();
            v[f] = fin ? v[f] : pad;
            k += fin ? 0 : 1;
        }
    }

    SortNet<NP>::sort(v);  // finite samples now occupy sorted positions [c_lo, c_lo + n)

    // ---- iteration 0: median / MAD (combine.rs:37-48) ----
    const float med = v[NP / 2];
    float mad;
    if (full_wave) {
        mad = mad_full<NP>(v, med);
    } else {
        // |v_i - med| of the finite samples, plus c_lo pads of -1 below and +inf above, so that
        // the n/2-th smallest deviation also lands on position NP/2 of the sorted deviations.
        float d[NP];
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            const float dev = fabsf(v[i] - med);
            d[i] = (i < c_lo) ? -1.0f : ((i < c_lo + n) ? dev : __builtin_inff());
        }
        SortNet<NP>::sort(d);
        mad = d[NP / 2];
    }
    float sigma = (float)fmax((double)mad * kMadToSigma, 1e-10);
    float center = med;

    int a = c_lo, b = c_lo + n - 1, len = n;
    uint32_t rej = 0;
    float last_center = __builtin_nanf("");
    bool active = (n >= 2);

    for (uint32_t it = 0; it < args.max_iter; ++it) {
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

        const float lo = -args.sigma_low * sigma;  // combine.rs:65-66
        const float hi = args.sigma_high * sigma;
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
    float result;
    if (n == 0) {
        result = 0.0f;
    } else if (n == 1) {
        result = med;  // the single finite sample sits on position NP/2
    } else if (len == 0) {
        result = __builtin_isfinite(last_center) ? last_center : 0.0f;
    } else {
        result = (float)(S / (double)len);
    }

    if (valid) {
        if constexpr (PARTIAL) {
            args.out_sum[g] = (len > 0) ? S : 0.0;
            args.out_cnt[g] = (uint32_t)(len > 0 ? len : 0);
        } else {
            args.out[g] = result;
        }
    } else {
        rej = 0;
    }

    // wavefront reduction of the rejection count, one atomic per wave
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) rej += __shfl_xor(rej, off, 64);
    if ((threadIdx.x & 63) == 0 && rej != 0) atomicAdd(args.rejected, (unsigned long long)rej);
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

template <bool PARTIAL>
int launch_stack(ab_ctx *ctx, const StackArgs &args, int np) {
    const int64_t total = args.rows * args.cols;
    const dim3 grid((unsigned)((total + 255) / 256)), block(256);
    switch (np) {
        case 2: hipLaunchKernelGGL((stack_sigma_clip_kernel<2, PARTIAL>), grid, block, 0, ctx->stream, args); break;
        case 4: hipLaunchKernelGGL((stack_sigma_clip_kernel<4, PARTIAL>), grid, block, 0, ctx->stream, args); break;
        case 8: hipLaunchKernelGGL((stack_sigma_clip_kernel<8, PARTIAL>), grid, block, 0, ctx->stream, args); break;
        case 16: hipLaunchKernelGGL((stack_sigma_clip_kernel<16, PARTIAL>), grid, block, 0, ctx->stream, args); break;
        case 32: hipLaunchKernelGGL((stack_sigma_clip_kernel<32, PARTIAL>), grid, block, 0, ctx->stream, args); break;
        case 64: hipLaunchKernelGGL((stack_sigma_clip_kernel<64, PARTIAL>), grid, block, 0, ctx->stream, args); break;
        default: return ab_set_error(ctx, AB_ERR_INVALID, "internal: bad padded frame count %d", np);
    }
    AB_HIP(ctx, hipGetLastError());
    return AB_OK;
}

}  // namespace

// Shared implementation.  dplanes: device pointers + row strides of the n frames.
int ab_stack_device(ab_ctx *ctx, const float *const *dplanes, const int64_t *ld, size_t n, int64_t rows,
                    int64_t cols, const ab_stack_config *cfg, float *out_dev, double *out_sum_dev,
                    uint32_t *out_cnt_dev, uint64_t *out_rejected) {
    AB_CHECK(ctx, n >= 1, "No images to stack");
    if (n > (size_t)kMaxFrames)
        return ab_set_error(ctx, AB_ERR_UNSUPPORTED, "stack of %zu frames: this build keeps <= %d frames per pixel in registers",
                            n, kMaxFrames);
    AB_CHECK(ctx, rows > 0 && cols > 0, "stack output has a zero dimension");
    AB_HIP(ctx, hipSetDevice(ctx->device));
    const int64_t total = rows * cols;
    const bool partial = out_sum_dev != nullptr;

    AB_HIP(ctx, hipMemsetAsync(ctx->counters, 0, sizeof(unsigned long long), ctx->stream));
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
            args.ld[i] = ld[i];
            if (ld[i] != cols) contiguous = 0;
        }
        args.n = (int)n;
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
        AB_TRY(partial ? launch_stack<true>(ctx, args, np) : launch_stack<false>(ctx, args, np));
    }
    if (out_rejected) {
        void *pin = nullptr;
        AB_TRY(ab_pinned(ctx, sizeof(unsigned long long), &pin));
        AB_HIP(ctx, hipMemcpyAsync(pin, ctx->counters, sizeof(unsigned long long), hipMemcpyDeviceToHost, ctx->stream));
        AB_HIP(ctx, hipStreamSynchronize(ctx->stream));
        *out_rejected = *(unsigned long long *)pin;
    }
    return AB_OK;
}

extern "C" {

int ab_stack_sigma_clip(ab_ctx *ctx, const ab_plane *planes, size_t n, const ab_stack_config *cfg, ab_plane_mut *out,
                        uint64_t *out_rejected) {
    if (!ctx) return AB_ERR_INVALID;
    AB_CHECK(ctx, planes && n >= 1, "No images to stack");
    AB_CHECK(ctx, cfg && out, "null config or output");
    if (n > (size_t)kMaxFrames)
        return ab_set_error(ctx, AB_ERR_UNSUPPORTED, "stack of %zu frames: this build keeps <= %d frames per pixel in registers",
                            n, kMaxFrames);
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
                             out_rejected ? &rejected : nullptr);
    if (rc == AB_OK) {
        rc = ab_stage_out_finish(ctx, &so);
        so_open = false;
    }
    if (so_open) ab_stage_out_abort(ctx, &so);
    for (size_t i = 0; i < staged; ++i) ab_stage_release(ctx, &st[i]);
    if (rc == AB_OK && out_rejected) *out_rejected = rejected;
    return rc;
}

int ab_stack_sigma_clip_partial(ab_ctx *ctx, const ab_plane *planes, size_t n, const ab_stack_config *cfg, int64_t rows,
                                int64_t cols, double *out_sum_dev, uint32_t *out_cnt_dev, uint64_t *out_rejected) {
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
    return ab_stack_device(ctx, dp.data(), ld.data(), n, rows, cols, cfg, nullptr, out_sum_dev, out_cnt_dev, out_rejected);
}

int ab_stack_last_rejected(ab_ctx *ctx, uint64_t *out_rejected) {
    if (!ctx) return AB_ERR_INVALID;
    AB_CHECK(ctx, out_rejected, "null output");
    void *pin = nullptr;
    AB_TRY(ab_pinned(ctx, sizeof(unsigned long long), &pin));
    AB_HIP(ctx, hipMemcpyAsync(pin, ctx->counters, sizeof(unsigned long long), hipMemcpyDeviceToHost, ctx->stream));
    AB_HIP(ctx, hipStreamSynchronize(ctx->stream));
    *out_rejected = *(unsigned long long *)pin;
    return AB_OK;
}

int ab_stack_finalize_partial(ab_ctx *ctx, const double *sum_dev, const uint32_t *cnt_dev, int64_t n, float *out_dev) {
    if (!ctx) return AB_ERR_INVALID;
    AB_CHECK(ctx, sum_dev && cnt_dev && out_dev && n > 0, "null buffer or empty range");
    AB_HIP(ctx, hipSetDevice(ctx->device));
    hipLaunchKernelGGL(finalize_partial_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, sum_dev,
                       cnt_dev, n, out_dev);
    AB_HIP(ctx, hipGetLastError());
    return AB_OK;
}

}  // extern "C"
