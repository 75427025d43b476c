// Star mask + masked (star-protecting) iterative stretch on gfx950.
//
// Replaces core/imaging/star_mask.rs (generate_star_mask :38-44, generate_star_mask_from_detection
// :46-138) and core/imaging/masked_stretch.rs (masked_stretch :44-58, masked_stretch_with_mask
// :60-118, masked_stretch_rgb_shared :155-193, normalize_to_01 :195-212, compute_masked_median
// :214-230, mtf_balance :232-238, apply_mtf :240-255, clamp_inplace :257-259).
//
// Mask: one workgroup paints each star's disc + smoothstep skirt with atomicMax on the f32 bit
// patterns (values >= 0 order like their bits, so the reference's serial "keep the larger" merge is
// reproduced independent of order), then one streaming pass applies the luminance protection and
// counts the coverage.  Stretch: every iteration is two HBM-bound steps -- the exact [len/2]
// element of the unmasked positive pixels (plane_select.hip; the reference rebuilds and selects a
// P-sized Vec serially, masked_stretch.rs:213-229) and one fused kernel doing apply_mtf + the
// mask-weighted blend in place (the reference materialises the stretched copy first).
#include "ab_common.hpp"

#include <algorithm>
#include <cmath>

namespace {

constexpr int kBlock = 256;

struct StarDisc {
    double x, y, fwhm;
};

__device__ __forceinline__ long long sat_index(double v) {  // Rust `f64 as usize` (saturating, NaN -> 0), capped for 64-bit maths
    if (!(v > 0.0)) return 0;
    if (v >= 4.0e18) return 4000000000000000000LL;
    return (long long)v;
}

__global__ __launch_bounds__(kBlock) void star_paint_kernel(const StarDisc *__restrict__ stars, int h, int w, double growth, double softness,
                                                            unsigned int *__restrict__ mask_bits) {
    const StarDisc s = stars[blockIdx.x];
    const double radius = s.fwhm * growth, soft_radius = radius + softness;  // star_mask.rs:65-66
    const long long y_min = sat_index(fmax(floor(s.y - soft_radius), 0.0));
    const long long y_max = min(sat_index(ceil(s.y + soft_radius)), (long long)(h - 1));
    const long long x_min = sat_index(fmax(floor(s.x - soft_radius), 0.0));
    const long long x_max = min(sat_index(ceil(s.x + soft_radius)), (long long)(w - 1));
    if (y_min > y_max || x_min > x_max) return;
    const double r2_inner = radius * radius, r2_outer = soft_radius * soft_radius;
    const double fade_range = fmax(r2_outer - r2_inner, 1e-10);
    const long long bw = x_max - x_min + 1, total = bw * (y_max - y_min + 1);
    for (long long i = threadIdx.x; i < total; i += kBlock) {
        const long long py = y_min + i / bw, px = x_min + i % bw;
        const double dx = (double)px - s.x, dy = (double)py - s.y;
        const double d2 = dx * dx + dy * dy;
        float val;
        if (d2 <= r2_inner) {
            val = 1.0f;
        } else if (d2 <= r2_outer) {
            const float t = (float)((d2 - r2_inner) / fade_range);
            const float smooth = t * t * (3.0f - 2.0f * t);
            val = 1.0f - smooth;
        } else {
            continue;
        }
        if (val > 0.0f) atomicMax(&mask_bits[py * w + px], __float_as_uint(val));  // :106-113
    }
}

__global__ __launch_bounds__(kBlock) void mask_protect_count_kernel(const float *__restrict__ img, float *__restrict__ mask, int64_t n,
                                                                    int protect, float ceiling, float inv_range,
                                                                    unsigned long long *__restrict__ covered) {
    __shared__ unsigned int s_cnt;
    if (threadIdx.x == 0) s_cnt = 0;
    __syncthreads();
    unsigned int local = 0;
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) {
        float m = mask[i];
        if (protect) {  // :115-132
            const float pixel = img[i];
            if (pixel > ceiling && m < 1.0f) {
                float excess = (pixel - ceiling) * inv_range;
                excess = excess < 0.0f ? 0.0f : (excess > 1.0f ? 1.0f : excess);
                const float smooth = excess * excess * (3.0f - 2.0f * excess);
                if (smooth > m) {
                    m = smooth;
                    mask[i] = m;
                }
            }
        }
        local += m > 0.01f ? 1u : 0u;
    }
    if (local) atomicAdd(&s_cnt, local);
    __syncthreads();
    if (threadIdx.x == 0 && s_cnt) atomicAdd(covered, (unsigned long long)s_cnt);
}

__global__ __launch_bounds__(kBlock) void normalize01_kernel(const float *__restrict__ in, int64_t n, int zero_all, float dmin, float inv,
                                                             float *__restrict__ out) {
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) {
        const float v = in[i];
        float r = 0.0f;
        if (!zero_all && __builtin_isfinite(v) && !(v <= 0.0f)) {  // masked_stretch.rs:204-210
            const float t = (v - dmin) * inv;
            r = t < 0.0f ? 0.0f : (t > 1.0f ? 1.0f : t);
        }
        out[i] = r;
    }
}

__global__ __launch_bounds__(kBlock) void mtf_blend_kernel(float *__restrict__ work, const float *__restrict__ mask, int64_t n, float m,
                                                           float protection) {
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) {
        const float x = work[i];
        float stretched;  // apply_mtf :240-255
        if (x <= 0.0f) {
            stretched = 0.0f;
        } else if (x >= 1.0f) {
            stretched = 1.0f;
        } else {
            const float denom = (2.0f * m - 1.0f) * x - m;
            if (fabsf(denom) < 1e-10f) {
                stretched = x;
            } else {
                const float v = (m - 1.0f) * x / denom;
                stretched = v < 0.0f ? 0.0f : (v > 1.0f ? 1.0f : v);
            }
        }
        const float blend = mask[i] * protection;  // :96-99
        work[i] = x * blend + stretched * (1.0f - blend);
    }
}

int stream_grid(ab_ctx *ctx, int64_t n) {
    return (int)std::max<int64_t>(1, std::min<int64_t>((n + kBlock - 1) / kBlock, (int64_t)(ctx->cu_count > 0 ? ctx->cu_count : 256) * 8));
}

// frees / aborts everything registered with it when the entry point returns
struct Scope {
    ab_ctx *ctx;
    std::vector<StagedPlane *> ins;
    std::vector<StagedOut *> outs;
    std::vector<void *> dev;
    int next_slot = 0;
    explicit Scope(ab_ctx *c) : ctx(c) {}
    ~Scope() {
        for (StagedOut *o : outs) ab_stage_out_abort(ctx, o);
        if (!dev.empty()) (void)hipStreamSynchronize(ctx->stream);
        for (void *p : dev) (void)hipFree(p);
        for (StagedPlane *p : ins) ab_stage_release(ctx, p);
    }
    // the call's device planes come out of the context's scope slots (kept between calls); a ninth one would be allocated and freed
    int alloc(void **p, size_t bytes) {
        if (next_slot < 8) return ab_workspace(ctx, AB_WS_SCOPE0 + next_slot++, std::max<size_t>(bytes, 16), p);
        AB_HIP(ctx, hipMalloc(p, bytes));
        dev.push_back(*p);
        return AB_OK;
    }
};

// generate_star_mask_from_detection on device planes (mask: rows * cols floats)
int star_mask_device(ab_ctx *ctx, const float *img, int64_t rows, int64_t cols, const std::vector<StarDisc> &all, const ab_star_mask_config &cfg,
                     float *mask, ab_star_mask_info *info, Scope &sc) {
    const int64_t n = rows * cols;
    std::vector<StarDisc> valid;
    for (const StarDisc &s : all)
        if (s.fwhm >= cfg.min_fwhm && s.fwhm <= cfg.max_fwhm) valid.push_back(s);  // star_mask.rs:54-58
    info->stars_masked = valid.size();
    info->coverage_fraction = 0.0;
    if (n == 0) {
        info->coverage_fraction = std::nan("");  // 0 / 0 (:134-135)
        return AB_OK;
    }
    AB_HIP(ctx, hipMemsetAsync(mask, 0, (size_t)n * sizeof(float), ctx->stream));
    if (!valid.empty()) {
        StarDisc *dstars = nullptr;
        AB_TRY(sc.alloc((void **)&dstars, valid.size() * sizeof(StarDisc)));
        AB_HIP(ctx, hipMemcpyAsync(dstars, valid.data(), valid.size() * sizeof(StarDisc), hipMemcpyHostToDevice, ctx->stream));
        hipLaunchKernelGGL(star_paint_kernel, dim3((unsigned)valid.size()), dim3(kBlock), 0, ctx->stream, dstars, (int)rows, (int)cols,
                           cfg.growth_factor, cfg.softness, (unsigned int *)mask);
        AB_HIP(ctx, hipGetLastError());
        AB_HIP(ctx, hipStreamSynchronize(ctx->stream));  // `valid` is pageable host memory
    }
    unsigned long long *dcov = nullptr;
    AB_TRY(sc.alloc((void **)&dcov, sizeof(unsigned long long)));
    AB_HIP(ctx, hipMemsetAsync(dcov, 0, sizeof(unsigned long long), ctx->stream));
    const float ceiling = (float)cfg.luminance_ceiling;
    const float inv_range = ceiling < 1.0f ? 1.0f / (1.0f - ceiling) : 1.0f;
    hipLaunchKernelGGL(mask_protect_count_kernel, dim3(stream_grid(ctx, n)), dim3(kBlock), 0, ctx->stream, img, mask, n, cfg.luminance_protect ? 1 : 0,
                       ceiling, inv_range, dcov);
    AB_HIP(ctx, hipGetLastError());
    unsigned long long covered = 0;
    AB_HIP(ctx, hipMemcpyAsync(&covered, dcov, sizeof covered, hipMemcpyDeviceToHost, ctx->stream));
    AB_HIP(ctx, hipStreamSynchronize(ctx->stream));
    info->coverage_fraction = (double)covered / (double)n;
    return AB_OK;
}

int detect_discs(ab_ctx *ctx, const float *img, int64_t rows, int64_t cols, double sigma, std::vector<StarDisc> *out) {
    std::vector<ab_detected_star> stars;
    double m, s;
    AB_TRY(ab_detect_stars_device(ctx, img, rows, cols, cols, sigma, &stars, &m, &s));
    out->clear();
    for (const ab_detected_star &st : stars) out->push_back({st.x, st.y, st.fwhm});
    return AB_OK;
}

__host__ __device__ inline double mtf_balance(double median, double target) {  // masked_stretch.rs:232-238
    const double denom = 2.0 * target * median - target - median;
    if (fabs(denom) < 1e-15) return 0.5;
    const double v = median * (target - 1.0) / denom;
    return v < 0.0001 ? 0.0001 : (v > 0.9999 ? 0.9999 : v);
}

// ---- masked_stretch_with_mask (:60-118) as ONE device-resident chain -----------------------------------------------------------
// Round 3 joined the host after every pass of every masked median (three passes per median, four medians per channel) and ran
// the three channels of masked_stretch_rgb_shared one after another: 9.85 ms for 3 x 8192^2.  Here the loop's scalar state lives
// in device memory (MsState): a median is three histogram passes (11 / 11 / 10 bits of the f32 pattern of the unmasked positive
// pixels) each followed by a one-workgroup kernel that picks the rank's bin and narrows the prefix; the last of them also DECIDES
// the iteration (at_target / stagnated / mtf_balance, the reference's f64 arithmetic); the blend reads the midtone from the state
// and histograms the first level of the NEXT median on its way out (one pass over the plane less per iteration); every kernel
// of an iteration that the decision cancelled returns at once.  The host enqueues the whole chain for the configured number of
// iterations, joins ONCE, and the three channels run on three streams.
struct MsState {
    float dmin, inv;  // normalize_to_01's transform (:195-212)
    int zero_all;
    uint32_t kmin, kmax;  // ordered keys of the valid pixels' minimum / maximum (range pass)
    uint32_t prefix_mask, prefix_val;
    unsigned long long rank, count;
    double bg, prev_bg;
    float m;
    int done, converged;
    unsigned long long iterations_run;
    // the NEXT median's first level predicted (see ms_blend_hist0_kernel): its level-0 bin, and whether the blend's second histogram
    // (level 1 under that bin) is the one the median needs
    uint32_t pred_b0;
    int need_clamp;  // some pixel left the last blend outside [0, 1] (clamp_inplace, :105, is the identity otherwise and its pass is skipped)
    int l1_ready, pred_valid;  // pred_valid: a blend with a prediction has filled hist + 2048 since the last level-1 pick
};
constexpr int kPickBlock = 1024;

__device__ __forceinline__ uint32_t ord_key(float v) {  // floats of either sign as monotone unsigned integers
    const uint32_t b = __float_as_uint(v);
    return b ^ ((b >> 31) ? 0xffffffffu : 0x80000000u);
}
__device__ __forceinline__ float from_ord_key(uint32_t u) { return __uint_as_float(u ^ ((u >> 31) ? 0x80000000u : 0xffffffffu)); }
__device__ __forceinline__ bool ms_candidate(float v, float mk) { return mk < 0.5f && __builtin_isfinite(v) && v > 0.0f; }  // :217-221

__global__ void ms_init_kernel(MsState *st, unsigned int *hist) {
    hist[blockIdx.x * 1024 + threadIdx.x] = 0u;
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        MsState z = {};
        z.kmin = 0xffffffffu;
        *st = z;
    }
}

// masked_stretch_rgb_shared reads the three channels for its luminance anyway: their ranges (ms_range_kernel's minimum / maximum of
// the valid pixels) are taken in the same pass, into the three chains' states (initialised before it)
struct RangeWave {
    uint32_t lo = 0xffffffffu, hi = 0u;
    __device__ __forceinline__ void add(float v) {
        if (__builtin_isfinite(v) && v > 1e-7f) {
            const uint32_t b = __float_as_uint(v);
            const uint32_t k = b ^ ((b >> 31) ? 0xffffffffu : 0x80000000u);  // ord_key
            lo = min(lo, k);
            hi = max(hi, k);
        }
    }
    __device__ __forceinline__ void flush(MsState *st, uint32_t *s_lo, uint32_t *s_hi) {  // s_lo / s_hi: kBlock / 64 words each
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) {
            lo = min(lo, (uint32_t)__shfl_xor((int)lo, off, 64));
            hi = max(hi, (uint32_t)__shfl_xor((int)hi, off, 64));
        }
        __syncthreads();
        if ((threadIdx.x & 63) == 0) {
            s_lo[threadIdx.x >> 6] = lo;
            s_hi[threadIdx.x >> 6] = hi;
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            for (int w = 1; w < kBlock / 64; ++w) {
                lo = min(lo, s_lo[w]);
                hi = max(hi, s_hi[w]);
            }
            if (lo <= hi) {
                atomicMin(&st->kmin, lo);
                atomicMax(&st->kmax, hi);
            }
        }
    }
};
__global__ __launch_bounds__(kBlock) void luminance_range_kernel(const float *__restrict__ r, const float *__restrict__ g, const float *__restrict__ b,
                                                                 int64_t n, float *__restrict__ out, MsState *sr, MsState *sg, MsState *sb) {
    __shared__ uint32_t s_lo[kBlock / 64], s_hi[kBlock / 64];
    RangeWave wr, wg, wb;
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) {  // masked_stretch.rs:143-154
        const float rv = r[i], gv = g[i], bv = b[i];
        const float rn = __builtin_isfinite(rv) ? rv : 0.0f, gn = __builtin_isfinite(gv) ? gv : 0.0f, bn = __builtin_isfinite(bv) ? bv : 0.0f;
        out[i] = 0.2126f * rn + 0.7152f * gn + 0.0722f * bn;
        wr.add(rv);
        wg.add(gv);
        wb.add(bv);
    }
    wr.flush(sr, s_lo, s_hi);
    wg.flush(sg, s_lo, s_hi);
    wb.flush(sb, s_lo, s_hi);
}

// minimum / maximum of the valid pixels (compute_image_stats' range: finite and above the padding threshold, stats.rs:10-13)
__global__ __launch_bounds__(kBlock) void ms_range_kernel(const float *__restrict__ in, int64_t n, MsState *st) {
    uint32_t lo = 0xffffffffu, hi = 0u;
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) {
        const float v = in[i];
        if (__builtin_isfinite(v) && v > 1e-7f) {
            const uint32_t k = ord_key(v);
            lo = min(lo, k);
            hi = max(hi, k);
        }
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        lo = min(lo, (uint32_t)__shfl_xor((int)lo, off, 64));
        hi = max(hi, (uint32_t)__shfl_xor((int)hi, off, 64));
    }
    __shared__ uint32_t s_lo[kBlock / 64], s_hi[kBlock / 64];
    if ((threadIdx.x & 63) == 0) {
        s_lo[threadIdx.x >> 6] = lo;
        s_hi[threadIdx.x >> 6] = hi;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < kBlock / 64; ++w) {
            lo = min(lo, s_lo[w]);
            hi = max(hi, s_hi[w]);
        }
        if (lo <= hi) {
            atomicMin(&st->kmin, lo);
            atomicMax(&st->kmax, hi);
        }
    }
}
__global__ void ms_range_finish_kernel(MsState *st) {
    // stats.min / stats.max as f64 of the f32 extremes (0 / 0 without a valid pixel, stats.rs:95-97), then :196-201 in f32
    const bool any = st->kmin <= st->kmax;
    const double mn = any ? (double)from_ord_key(st->kmin) : 0.0, mx = any ? (double)from_ord_key(st->kmax) : 0.0;
    const float range = (float)(mx - mn);
    st->zero_all = range < 1e-10f;
    st->dmin = (float)mn;
    st->inv = st->zero_all ? 0.0f : 1.0f / range;
}

// a block's share of one level of the select: LDS histogram of the candidates under the current prefix, flushed to `hist`
struct LevelHist {
    unsigned int *lds;
    uint32_t nb, pmask, pval;
    int shift;
    __device__ __forceinline__ void begin(unsigned int *l, int nbits, int sh, uint32_t m, uint32_t v) {
        lds = l;
        nb = 1u << nbits;
        shift = sh;
        pmask = m;
        pval = v;
        for (uint32_t i = threadIdx.x; i < nb; i += kBlock) lds[i] = 0;
        __syncthreads();
    }
    __device__ __forceinline__ void add(float v) {
        const uint32_t key = __float_as_uint(v);
        if ((key & pmask) == pval) atomicAdd(&lds[(key >> shift) & (nb - 1)], 1u);
    }
    __device__ __forceinline__ void end(unsigned int *hist) {
        __syncthreads();
        for (uint32_t i = threadIdx.x; i < nb; i += kBlock)
            if (lds[i]) atomicAdd(&hist[i], lds[i]);
    }
};

// normalize_to_01 (:195-212) + level 0 of the first median
__global__ __launch_bounds__(kBlock) void ms_normalize_hist0_kernel(const float *__restrict__ in, const float *__restrict__ mask, int64_t n,
                                                                    const MsState *__restrict__ st, float *__restrict__ out, unsigned int *hist) {
    __shared__ unsigned int lds[2048];
    LevelHist H;
    H.begin(lds, 11, 21, 0u, 0u);
    const int zero_all = st->zero_all;
    const float dmin = st->dmin, inv = st->inv;
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) {
        const float v = in[i];
        float r = 0.0f;
        if (!zero_all && __builtin_isfinite(v) && !(v <= 0.0f)) {  // masked_stretch.rs:204-210
            const float t = (v - dmin) * inv;
            r = t < 0.0f ? 0.0f : (t > 1.0f ? 1.0f : t);
        }
        out[i] = r;
        if (ms_candidate(r, mask[i])) H.add(r);
    }
    H.end(hist);
}

// levels 1 and 2 of a median
__global__ __launch_bounds__(kBlock) void ms_hist_kernel(const float *__restrict__ work, const float *__restrict__ mask, int64_t n,
                                                         const MsState *__restrict__ st, unsigned int *hist, int level) {
    if (st->done || st->count == 0 || (level == 1 && st->l1_ready)) return;
    __shared__ unsigned int lds[2048];
    LevelHist H;
    H.begin(lds, level == 1 ? 11 : 10, level == 1 ? 10 : 0, st->prefix_mask, st->prefix_val);
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) {
        const float v = work[i];
        if (ms_candidate(v, mask[i])) H.add(v);
    }
    H.end(hist);
}

// apply_mtf's pixel function (:240-255)
__device__ __forceinline__ float mtf_pixel(float x, float m) {
    if (x <= 0.0f) return 0.0f;
    if (x >= 1.0f) return 1.0f;
    const float denom = (2.0f * m - 1.0f) * x - m;
    if (fabsf(denom) < 1e-10f) return x;
    const float v = (m - 1.0f) * x / denom;
    return v < 0.0f ? 0.0f : (v > 1.0f ? 1.0f : v);
}

// apply_mtf (:240-255) + the mask-weighted blend (:93-100) in place, + level 0 of the next median -- and, round 4, its LEVEL 1 too
// when a prediction holds: an unmasked pixel (mask 0, all of the sky) becomes mtf(x) exactly, a monotone function, so the next
// median would be mtf(this median) if every candidate were unmasked; the few per cent with a soft mask value move it by a few
// level-1 bins at most, never out of its level-0 bin (a quarter of a binade).  The pass therefore also histograms level 1 of the
// candidates inside the PREDICTED level-0 bin; the pick after it checks the prediction against the real level-0 histogram and,
// when it holds, the level-1 pass over the plane returns at once (one of three passes per iteration: 28 -> 20 bytes per pixel).
__global__ __launch_bounds__(kBlock) void ms_blend_hist0_kernel(float *__restrict__ work, const float *__restrict__ mask, int64_t n,
                                                                MsState *st, float protection, unsigned int *hist) {
    if (st->done) return;
    bool outside = false;  // a blend of two values of [0, 1] leaves [0, 1] only by a rounding of 1 - blend (x = stretched = 1)
    __shared__ unsigned int lds[2048], lds1[2048];
    LevelHist H;
    for (uint32_t i = threadIdx.x; i < 2048u; i += kBlock) lds1[i] = 0;
    H.begin(lds, 11, 21, 0u, 0u);
    const float m = st->m;
    const uint32_t pred = st->pred_b0;
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) {
        const float x = work[i], mk = mask[i];
        const float stretched = mtf_pixel(x, m);
        const float blend = mk * protection;
        const float r = x * blend + stretched * (1.0f - blend);
        work[i] = r;
        outside |= r < 0.0f || r > 1.0f;
        if (ms_candidate(r, mk)) {
            H.add(r);
            const uint32_t key = __float_as_uint(r);
            if ((key >> 21) == pred) atomicAdd(&lds1[(key >> 10) & 0x7ffu], 1u);
        }
    }
    H.end(hist);
    for (uint32_t i = threadIdx.x; i < 2048u; i += kBlock)
        if (lds1[i]) atomicAdd(&hist[2048 + i], lds1[i]);
    if (__any(outside) && (threadIdx.x & 63) == 0) atomicOr(&st->need_clamp, 1);
}

// clamp_inplace (:105, :256-258): the identity on [0, 1], -0.0 and NaN -- the pass runs only when the last blend said it must
__global__ __launch_bounds__(kBlock) void ms_clamp_kernel(float *__restrict__ work, int64_t n, const MsState *__restrict__ st) {
    if (!st->need_clamp) return;
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) {
        const float v = work[i];
        work[i] = v < 0.0f ? 0.0f : (v > 1.0f ? 1.0f : v);
    }
}

// One workgroup after every histogram pass: the bin of the wanted rank, the narrowed prefix, the histogram cleared for the next
// pass.  Level 0 also takes the candidate count (rank = count / 2: the [len / 2] element, :226-229).  Level 2 completes a median
// and runs the loop's head for iteration `it` (:78-92): it = -1: the median before the loop (:71); it = iterations: the loop has run out.
__global__ __launch_bounds__(kPickBlock) void ms_pick_kernel(unsigned int *hist, MsState *st, int level, int it, int iterations, double target, double threshold) {
    if (st->done) return;
    __shared__ unsigned long long s_wave[kPickBlock / 64];
    __shared__ uint32_t s_bin;
    __shared__ unsigned long long s_before;
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
    const uint32_t nb = level == 2 ? 1024u : 2048u;
    // level 1 reads the blend's predicted histogram (hist + 2048) when the level-0 pick found the prediction right; it clears both
    const unsigned int *src = (level == 1 && st->l1_ready) ? hist + 2048 : hist;
    // thread t owns bins 2t, 2t + 1 (levels 0, 1) or bin t (level 2)
    const unsigned int c0 = level == 2 ? (t < 1024 ? src[t] : 0u) : src[2 * t], c1 = level == 2 ? 0u : src[2 * t + 1];
    if (level == 2) {
        if (t < 1024) hist[t] = 0;
    } else {
        hist[2 * t] = 0;
        hist[2 * t + 1] = 0;
        if (level == 1) {
            hist[2048 + 2 * t] = 0;
            hist[2048 + 2 * t + 1] = 0;
        }
    }
    unsigned long long own = (unsigned long long)c0 + c1, incl = own;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const unsigned long long up = __shfl_up(incl, off, 64);
        if (lane >= off) incl += up;
    }
    if (lane == 63) s_wave[wv] = incl;
    if (t == 0) s_bin = 0xffffffffu;
    __syncthreads();
    unsigned long long before = incl - own, total = 0;
    for (int w = 0; w < kPickBlock / 64; ++w) {
        if (w < wv) before += s_wave[w];
        total += s_wave[w];
    }
    if (level == 0 && t == 0) {
        st->count = total;
        st->rank = total / 2;
    }
    __syncthreads();
    if (level == 0 && total == 0) {  // no candidate: the median is 0.0 (:222-224)
        if (t == 0) {
            st->prefix_mask = st->prefix_val = 0u;
            st->rank = 0;
            st->l1_ready = 0;
        }
    } else {
        const unsigned long long rank = level == 0 ? total / 2 : st->rank;
        if (own && rank >= before && rank < before + own) {  // exactly one thread
            const bool second = level != 2 && rank >= before + c0;
            s_bin = (level == 2 ? (uint32_t)t : 2u * t) + (second ? 1u : 0u);
            s_before = before + (second ? c0 : 0u);
        }
        __syncthreads();
        if (t == 0) {
            const uint32_t bin = s_bin == 0xffffffffu ? nb - 1 : s_bin;
            const int shift = level == 0 ? 21 : (level == 1 ? 10 : 0);
            const uint32_t bits = level == 2 ? 0x3ffu : 0x7ffu;
            const uint32_t pm = level == 0 ? 0u : st->prefix_mask, pv = level == 0 ? 0u : st->prefix_val;
            st->prefix_mask = pm | (bits << shift);
            st->prefix_val = pv | (bin << shift);
            st->rank = s_bin == 0xffffffffu ? 0ull : rank - s_before;
            if (level == 0) st->l1_ready = (s_bin != 0xffffffffu && bin == st->pred_b0 && st->pred_valid) ? 1 : 0;
        }
    }
    if (level == 1) {
        __syncthreads();
        if (t == 0) st->pred_valid = 0;  // hist + 2048 has been cleared
    }
    if (level != 2) return;
    __syncthreads();
    if (t == 0) {
        const double bg = st->count == 0 ? 0.0 : (double)__uint_as_float(st->prefix_val);
        if (it < 0) {  // :71
            st->prev_bg = bg;
            st->bg = bg;
            it = 0;
        } else {
            st->prev_bg = st->bg;  // (:103 after the blend of the iteration that just ended)
            st->bg = bg;
        }
        if (it >= iterations) {  // the loop has run its course (:77)
            st->done = 1;
            return;
        }
        // the head of iteration `it` (:78-92)
        st->iterations_run = (unsigned long long)it + 1ull;
        const bool at_target = fabs(bg - target) < threshold;
        const bool stagnated = it > 0 && fabs(bg - st->prev_bg) < threshold * 0.1;
        if (at_target) {
            st->converged = 1;
            st->done = 1;
        } else if (stagnated) {
            st->done = 1;
        } else {
            const float mt = (float)mtf_balance(bg, target);
            st->m = mt;
            st->need_clamp = 0;  // the blend that follows overwrites every pixel and raises it again if it has to
            // the blend that follows predicts the next median's level-0 bin from this median (see ms_blend_hist0_kernel)
            st->pred_b0 = __float_as_uint(mtf_pixel(st->count == 0 ? 0.0f : __uint_as_float(st->prefix_val), mt)) >> 21;
            st->pred_valid = 1;
        }
    }
}

// the whole chain of one channel on `stream`; nothing is synchronised.  scratch: sizeof(MsState) + 2048 words, device memory.
int masked_stretch_enqueue(ab_ctx *ctx, hipStream_t stream, const float *img, const float *mask, int64_t n, const ab_masked_stretch_config &cfg, float *work,
                           void *scratch, bool have_range = false /* the state is initialised and holds the plane's range already */) {
    MsState *st = (MsState *)scratch;
    unsigned int *hist = (unsigned int *)((char *)scratch + ((sizeof(MsState) + 63) & ~(size_t)63));
    if (!have_range) hipLaunchKernelGGL(ms_init_kernel, dim3(4), dim3(1024), 0, stream, st, hist);  // (no fill, no copy: the chain's state starts on the device)
    if (n <= 0) return AB_OK;
    const int grid = stream_grid(ctx, n);
    const int iterations = (int)std::min<size_t>(cfg.iterations, 1000000);
    const float protection = (float)cfg.protection_amount;
    if (!have_range) hipLaunchKernelGGL(ms_range_kernel, dim3(grid), dim3(kBlock), 0, stream, img, n, st);
    hipLaunchKernelGGL(ms_range_finish_kernel, dim3(1), dim3(1), 0, stream, st);
    hipLaunchKernelGGL(ms_normalize_hist0_kernel, dim3(grid), dim3(kBlock), 0, stream, img, mask, n, (const MsState *)st, work, hist);
    auto finish_median = [&](int it) {
        hipLaunchKernelGGL(ms_pick_kernel, dim3(1), dim3(kPickBlock), 0, stream, hist, st, 0, it, iterations, cfg.target_background, cfg.convergence_threshold);
        hipLaunchKernelGGL(ms_hist_kernel, dim3(grid), dim3(kBlock), 0, stream, (const float *)work, mask, n, (const MsState *)st, hist, 1);
        hipLaunchKernelGGL(ms_pick_kernel, dim3(1), dim3(kPickBlock), 0, stream, hist, st, 1, it, iterations, cfg.target_background, cfg.convergence_threshold);
        hipLaunchKernelGGL(ms_hist_kernel, dim3(grid), dim3(kBlock), 0, stream, (const float *)work, mask, n, (const MsState *)st, hist, 2);
        hipLaunchKernelGGL(ms_pick_kernel, dim3(1), dim3(kPickBlock), 0, stream, hist, st, 2, it, iterations, cfg.target_background, cfg.convergence_threshold);
    };
    finish_median(-1);
    // The reference leaves its loop at convergence (masked_stretch.rs:82-91); here an iteration that the decision has switched off
    // still costs six launches that read the state and return.  The usual ten iterations are enqueued in one go (one join for the
    // whole call); a configuration that asks for many more -- a convergence-bounded loop with a large ceiling is a natural use --
    // is enqueued kIterChunk at a time with a look at `done` in between, so its cost is bounded by what the loop really runs
    // (6 x iterations no-op launches, up to 6 000 000).  Beyond the first chunk the channels of the shared-mask form
    // run one after the other.
    constexpr int kIterChunk = 32;
    for (int it = 0; it < iterations; ++it) {
        if (it > 0 && it % kIterChunk == 0) {
            void *pin = nullptr;
            AB_TRY(ab_pinned(ctx, sizeof(MsState), &pin));
            AB_HIP(ctx, hipMemcpyAsync(pin, st, sizeof(MsState), hipMemcpyDeviceToHost, stream));
            AB_HIP(ctx, hipStreamSynchronize(stream));
            if (((const MsState *)pin)->done) break;
        }
        hipLaunchKernelGGL(ms_blend_hist0_kernel, dim3(grid), dim3(kBlock), 0, stream, work, mask, n, st, protection, hist);
        finish_median(it + 1);
    }
    hipLaunchKernelGGL(ms_clamp_kernel, dim3(grid), dim3(kBlock), 0, stream, work, n, (const MsState *)st);
    AB_HIP(ctx, hipGetLastError());
    return AB_OK;
}
// after the stream has been joined: the loop's outcome (host copy of the state)
void masked_stretch_result_of(const MsState &st, ab_masked_stretch_result *res) {
    res->iterations_run = (size_t)st.iterations_run;
    res->final_background = st.bg;
    res->converged = st.converged;
}
constexpr size_t kMsScratch = ((sizeof(MsState) + 63) & ~(size_t)63) + 4096 * sizeof(unsigned int);  // the level histogram + the predicted level 1

// masked_stretch_with_mask (:60-118) on device planes; `work` receives the result
int masked_stretch_device(ab_ctx *ctx, const float *img, const float *mask, int64_t n, const ab_masked_stretch_config &cfg, float *work,
                          ab_masked_stretch_result *res, Scope &sc) {
    void *scratch = nullptr;
    AB_TRY(sc.alloc(&scratch, kMsScratch));
    AB_TRY(masked_stretch_enqueue(ctx, ctx->stream, img, mask, n, cfg, work, scratch));
    void *pin = nullptr;
    AB_TRY(ab_pinned(ctx, sizeof(MsState), &pin));
    AB_HIP(ctx, hipMemcpyAsync(pin, scratch, sizeof(MsState), hipMemcpyDeviceToHost, ctx->stream));
    AB_HIP(ctx, hipStreamSynchronize(ctx->stream));
    masked_stretch_result_of(*(const MsState *)pin, res);
    return AB_OK;
}

ab_star_mask_config mask_config_of(const ab_masked_stretch_config &c) {  // masked_stretch.rs:48-54 (..StarMaskConfig::default())
    ab_star_mask_config m;
    m.growth_factor = c.mask_growth;
    m.softness = c.mask_softness;
    m.detection_sigma = 5.0;
    m.min_fwhm = 1.5;
    m.max_fwhm = 30.0;
    m.luminance_protect = c.luminance_protect;
    m.luminance_ceiling = c.luminance_ceiling;
    return m;
}

bool same_dims(const ab_plane *a, int64_t rows, int64_t cols) { return a->rows == rows && a->cols == cols; }

}  // namespace

extern "C" {

int ab_generate_star_mask_from_stars(ab_ctx *ctx, const ab_plane *img, const ab_detected_star *stars, size_t n_stars,
                                     const ab_star_mask_config *cfg, ab_plane_mut *out_mask, ab_star_mask_info *info) try {
    if (!ctx) return AB_ERR_INVALID;
    AB_CHECK(ctx, img && cfg && out_mask && info && (stars || n_stars == 0), "null argument");
    AB_CHECK(ctx, out_mask->rows == img->rows && out_mask->cols == img->cols, "mask must have the image's dims");
    AB_CHECK(ctx, img->rows * img->cols < (int64_t(1) << 31), "image too large for this build");
    AB_HIP(ctx, hipSetDevice(ctx->device));
    StagedPlane in;
    StagedOut so;
    Scope sc(ctx);
    AB_TRY(ab_stage_in(ctx, img, &in));
    sc.ins.push_back(&in);
    AB_TRY(ab_stage_out_begin(ctx, out_mask, &so));
    sc.outs.push_back(&so);
    std::vector<StarDisc> discs;
    for (size_t i = 0; i < n_stars; ++i) discs.push_back({stars[i].x, stars[i].y, stars[i].fwhm});
    AB_TRY(star_mask_device(ctx, in.dptr, in.rows, in.cols, discs, *cfg, so.dptr, info, sc));
    return ab_stage_out_finish(ctx, &so);
} AB_CATCH(ctx)

int ab_generate_star_mask(ab_ctx *ctx, const ab_plane *img, const ab_star_mask_config *cfg, ab_plane_mut *out_mask, ab_star_mask_info *info) try {
    if (!ctx) return AB_ERR_INVALID;
    AB_CHECK(ctx, img && cfg && out_mask && info, "null argument");
    AB_CHECK(ctx, out_mask->rows == img->rows && out_mask->cols == img->cols, "mask must have the image's dims");
    AB_CHECK(ctx, img->rows * img->cols < (int64_t(1) << 31), "image too large for this build");
    AB_HIP(ctx, hipSetDevice(ctx->device));
    StagedPlane in;
    StagedOut so;
    Scope sc(ctx);
    AB_TRY(ab_stage_in(ctx, img, &in));
    sc.ins.push_back(&in);
    AB_TRY(ab_stage_out_begin(ctx, out_mask, &so));
    sc.outs.push_back(&so);
    std::vector<StarDisc> discs;
    AB_TRY(detect_discs(ctx, in.dptr, in.rows, in.cols, cfg->detection_sigma, &discs));
    AB_TRY(star_mask_device(ctx, in.dptr, in.rows, in.cols, discs, *cfg, so.dptr, info, sc));
    return ab_stage_out_finish(ctx, &so);
} AB_CATCH(ctx)

int ab_masked_stretch_with_mask(ab_ctx *ctx, const ab_plane *img, const ab_plane *mask, const ab_star_mask_info *mask_info,
                                const ab_masked_stretch_config *cfg, ab_plane_mut *out, ab_masked_stretch_result *res) try {
    if (!ctx) return AB_ERR_INVALID;
    AB_CHECK(ctx, img && mask && cfg && out && res, "null argument");
    AB_CHECK(ctx, same_dims(mask, img->rows, img->cols) && out->rows == img->rows && out->cols == img->cols,
             "mask and output must have the image's dims");
    AB_HIP(ctx, hipSetDevice(ctx->device));
    StagedPlane in, mk;
    StagedOut so;
    Scope sc(ctx);
    AB_TRY(ab_stage_in(ctx, img, &in));
    sc.ins.push_back(&in);
    AB_TRY(ab_stage_in(ctx, mask, &mk));
    sc.ins.push_back(&mk);
    AB_TRY(ab_stage_out_begin(ctx, out, &so));
    sc.outs.push_back(&so);
    memset(res, 0, sizeof *res);
    AB_TRY(masked_stretch_device(ctx, in.dptr, mk.dptr, in.rows * in.cols, *cfg, so.dptr, res, sc));
    if (mask_info) {
        res->stars_masked = mask_info->stars_masked;
        res->mask_coverage = mask_info->coverage_fraction;
    }
    return ab_stage_out_finish(ctx, &so);
} AB_CATCH(ctx)

int ab_masked_stretch(ab_ctx *ctx, const ab_plane *img, const ab_masked_stretch_config *cfg, ab_plane_mut *out, ab_masked_stretch_result *res) try {
    if (!ctx) return AB_ERR_INVALID;
    AB_CHECK(ctx, img && cfg && out && res, "null argument");
    AB_CHECK(ctx, out->rows == img->rows && out->cols == img->cols, "output must have the image's dims");
    AB_CHECK(ctx, img->rows * img->cols < (int64_t(1) << 31), "image too large for this build");
    AB_HIP(ctx, hipSetDevice(ctx->device));
    StagedPlane in;
    StagedOut so;
    Scope sc(ctx);
    AB_TRY(ab_stage_in(ctx, img, &in));
    sc.ins.push_back(&in);
    AB_TRY(ab_stage_out_begin(ctx, out, &so));
    sc.outs.push_back(&so);
    const int64_t n = in.rows * in.cols;
    float *mask = nullptr;
    AB_TRY(sc.alloc((void **)&mask, std::max<size_t>((size_t)n, 1) * sizeof(float)));
    const ab_star_mask_config mc = mask_config_of(*cfg);
    ab_star_mask_info mi;
    std::vector<StarDisc> discs;
    AB_TRY(detect_discs(ctx, in.dptr, in.rows, in.cols, mc.detection_sigma, &discs));
    AB_TRY(star_mask_device(ctx, in.dptr, in.rows, in.cols, discs, mc, mask, &mi, sc));
    memset(res, 0, sizeof *res);
    AB_TRY(masked_stretch_device(ctx, in.dptr, mask, n, *cfg, so.dptr, res, sc));
    res->stars_masked = mi.stars_masked;
    res->mask_coverage = mi.coverage_fraction;
    return ab_stage_out_finish(ctx, &so);
} AB_CATCH(ctx)

int ab_masked_stretch_rgb_shared(ab_ctx *ctx, const ab_plane *r, const ab_plane *g, const ab_plane *b, const ab_masked_stretch_config *cfg,
                                 ab_plane_mut *out_r, ab_plane_mut *out_g, ab_plane_mut *out_b, ab_masked_stretch_result *res3,
                                 ab_star_mask_info *shared) try {
    if (!ctx) return AB_ERR_INVALID;
    AB_CHECK(ctx, r && g && b && cfg && out_r && out_g && out_b && res3, "null argument");
    if (!same_dims(g, r->rows, r->cols) || !same_dims(b, r->rows, r->cols))  // masked_stretch.rs:126-134
        return ab_set_error(ctx, AB_ERR_INVALID, "Channel dimension mismatch: R=(%lld, %lld) G=(%lld, %lld) B=(%lld, %lld)", (long long)r->rows,
                            (long long)r->cols, (long long)g->rows, (long long)g->cols, (long long)b->rows, (long long)b->cols);
    const ab_plane_mut *outs[3] = {out_r, out_g, out_b};
    for (const ab_plane_mut *o : outs) AB_CHECK(ctx, o->rows == r->rows && o->cols == r->cols, "outputs must have the channels' dims");
    AB_CHECK(ctx, r->rows * r->cols < (int64_t(1) << 31), "image too large for this build");
    AB_HIP(ctx, hipSetDevice(ctx->device));
    StagedPlane in[3];
    StagedOut so[3];
    Scope sc(ctx);
    const ab_plane *ins[3] = {r, g, b};
    for (int c = 0; c < 3; ++c) {
        AB_TRY(ab_stage_in(ctx, ins[c], &in[c]));
        sc.ins.push_back(&in[c]);
    }
    const int64_t rows = in[0].rows, cols = in[0].cols, n = rows * cols;
    float *lum = nullptr, *mask = nullptr;
    AB_TRY(sc.alloc((void **)&lum, std::max<size_t>((size_t)n, 1) * sizeof(float)));
    AB_TRY(sc.alloc((void **)&mask, std::max<size_t>((size_t)n, 1) * sizeof(float)));
    char *scratch = nullptr;
    AB_TRY(sc.alloc((void **)&scratch, 3 * kMsScratch));
    MsState *states[3];
    for (int c = 0; c < 3; ++c) {
        states[c] = (MsState *)(scratch + (size_t)c * kMsScratch);
        hipLaunchKernelGGL(ms_init_kernel, dim3(4), dim3(1024), 0, ctx->stream, states[c],
                           (unsigned int *)(scratch + (size_t)c * kMsScratch + ((sizeof(MsState) + 63) & ~(size_t)63)));
    }
    if (n > 0) {
        hipLaunchKernelGGL(luminance_range_kernel, dim3(stream_grid(ctx, n)), dim3(kBlock), 0, ctx->stream, in[0].dptr, in[1].dptr, in[2].dptr, n, lum,
                           states[0], states[1], states[2]);
        AB_HIP(ctx, hipGetLastError());
    }
    const ab_star_mask_config mc = mask_config_of(*cfg);
    ab_star_mask_info mi;
    std::vector<StarDisc> discs;
    AB_TRY(detect_discs(ctx, lum, rows, cols, mc.detection_sigma, &discs));
    AB_TRY(star_mask_device(ctx, lum, rows, cols, discs, mc, mask, &mi, sc));
    if (shared) *shared = mi;
    // the three channels as the reference's three-way join (:175-181): one chain each, on the context's stream and its two
    // auxiliary streams, ordered after the mask by an event and joined by two more
    if (!ctx->aux_stream) AB_HIP(ctx, hipStreamCreateWithFlags(&ctx->aux_stream, hipStreamNonBlocking));
    if (!ctx->warp_stream) AB_HIP(ctx, hipStreamCreateWithFlags(&ctx->warp_stream, hipStreamNonBlocking));
    hipStream_t streams[3] = {ctx->stream, ctx->aux_stream, ctx->warp_stream};
    if (!ctx->switch_ev) AB_HIP(ctx, hipEventCreateWithFlags(&ctx->switch_ev, hipEventDisableTiming));
    for (int c = 0; c < 3; ++c) {
        AB_TRY(ab_stage_out_begin(ctx, outs[c], &so[c]));
        sc.outs.push_back(&so[c]);
        memset(&res3[c], 0, sizeof res3[c]);
    }
    AB_HIP(ctx, hipEventRecord(ctx->switch_ev, ctx->stream));  // mask, staged inputs and outputs are in place
    int rc = AB_OK;
    for (int c = 0; c < 3 && rc == AB_OK; ++c) {
        if (c > 0 && hipStreamWaitEvent(streams[c], ctx->switch_ev, 0) != hipSuccess) rc = ab_set_error(ctx, AB_ERR_HIP, "hipStreamWaitEvent failed");
        if (rc == AB_OK) rc = masked_stretch_enqueue(ctx, streams[c], in[c].dptr, mask, n, *cfg, so[c].dptr, scratch + (size_t)c * kMsScratch, /*have_range=*/true);
    }
    // (whatever was enqueued is drained before anything is released, also on an error path)
    for (int c = 1; c < 3; ++c)
        if (hipStreamSynchronize(streams[c]) != hipSuccess && rc == AB_OK) rc = ab_set_error(ctx, AB_ERR_HIP, "hipStreamSynchronize failed");
    void *pin = nullptr;
    if (rc == AB_OK) rc = ab_pinned(ctx, 3 * sizeof(MsState), &pin);
    if (rc == AB_OK)
        for (int c = 0; c < 3; ++c)
            if (hipMemcpyAsync((char *)pin + (size_t)c * sizeof(MsState), scratch + (size_t)c * kMsScratch, sizeof(MsState), hipMemcpyDeviceToHost, ctx->stream) != hipSuccess)
                rc = ab_set_error(ctx, AB_ERR_HIP, "hipMemcpyAsync failed");
    if (hipStreamSynchronize(ctx->stream) != hipSuccess && rc == AB_OK) rc = ab_set_error(ctx, AB_ERR_HIP, "hipStreamSynchronize failed");
    if (rc != AB_OK) return rc;
    for (int c = 0; c < 3; ++c) {
        masked_stretch_result_of(((const MsState *)pin)[c], &res3[c]);
        res3[c].stars_masked = mi.stars_masked;
        res3[c].mask_coverage = mi.coverage_fraction;
    }
    for (int c = 0; c < 3; ++c) AB_TRY(ab_stage_out_finish(ctx, &so[c]));
    return AB_OK;
} AB_CATCH(ctx)

}  // extern "C"
