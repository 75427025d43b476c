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

__global__ __launch_bounds__(kBlock) void clamp01_kernel(float *__restrict__ work, int64_t n) {
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) {
        const float v = work[i];
        work[i] = v < 0.0f ? 0.0f : (v > 1.0f ? 1.0f : v);
    }
}

__global__ __launch_bounds__(kBlock) void luminance_kernel(const float *__restrict__ r, const float *__restrict__ g, const float *__restrict__ b,
                                                           int64_t n, float *__restrict__ out) {
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) {  // masked_stretch.rs:143-154
        const float rv = r[i], gv = g[i], bv = b[i];
        const float rn = __builtin_isfinite(rv) ? rv : 0.0f, gn = __builtin_isfinite(gv) ? gv : 0.0f, bn = __builtin_isfinite(bv) ? bv : 0.0f;
        out[i] = 0.2126f * rn + 0.7152f * gn + 0.0722f * bn;
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
    explicit Scope(ab_ctx *c) : ctx(c) {}
    ~Scope() {
        for (StagedOut *o : outs) ab_stage_out_abort(ctx, o);
        if (!dev.empty()) (void)hipStreamSynchronize(ctx->stream);
        for (void *p : dev) (void)hipFree(p);
        for (StagedPlane *p : ins) ab_stage_release(ctx, p);
    }
    int alloc(void **p, size_t bytes) {
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

double mtf_balance(double median, double target) {  // masked_stretch.rs:232-238
    const double denom = 2.0 * target * median - target - median;
    if (std::fabs(denom) < 1e-15) return 0.5;
    const double v = median * (target - 1.0) / denom;
    return v < 0.0001 ? 0.0001 : (v > 0.9999 ? 0.9999 : v);
}

// masked_stretch_with_mask (:60-118) on device planes; `work` receives the result
int masked_stretch_device(ab_ctx *ctx, const float *img, const float *mask, int64_t n, const ab_masked_stretch_config &cfg, float *work,
                          ab_masked_stretch_result *res) {
    const int grid = stream_grid(ctx, n);
    ab_image_stats st;
    memset(&st, 0, sizeof st);
    if (n > 0) AB_TRY(ab_stats_device(ctx, img, n, 0, 0.0, 0.0, &st));
    const float range = (float)(st.max - st.min);
    const int zero_all = range < 1e-10f;
    if (n > 0) {
        hipLaunchKernelGGL(normalize01_kernel, dim3(grid), dim3(kBlock), 0, ctx->stream, img, n, zero_all, (float)st.min,
                           zero_all ? 0.0f : 1.0f / range, work);
        AB_HIP(ctx, hipGetLastError());
    }
    ab_plane_sel sel;
    sel.data = work;
    sel.mask = mask;
    sel.n = n;
    auto masked_median = [&](double *out) -> int {  // :214-230 ([len/2] element, 0.0 when empty)
        uint64_t cnt;
        float mid;
        AB_TRY(ab_plane_order_stats(ctx, sel, 0, &cnt, &mid, nullptr));
        *out = cnt == 0 ? 0.0 : (double)mid;
        return AB_OK;
    };
    const float protection = (float)cfg.protection_amount;
    double current_bg;  // median of `work` as it stands (the reference recomputes it at :71, :79 and :107)
    AB_TRY(masked_median(&current_bg));
    double prev_bg = current_bg;
    size_t iterations_run = 0;
    int converged = 0;
    for (size_t it = 0; it < cfg.iterations; ++it) {
        iterations_run = it + 1;
        const double bg = current_bg;
        const bool at_target = std::fabs(bg - cfg.target_background) < cfg.convergence_threshold;
        const bool stagnated = it > 0 && std::fabs(bg - prev_bg) < cfg.convergence_threshold * 0.1;
        if (at_target) {
            converged = 1;
            break;
        }
        if (stagnated) break;
        const float m = (float)mtf_balance(bg, cfg.target_background);
        if (n > 0) {
            hipLaunchKernelGGL(mtf_blend_kernel, dim3(grid), dim3(kBlock), 0, ctx->stream, work, mask, n, m, protection);
            AB_HIP(ctx, hipGetLastError());
        }
        prev_bg = bg;
        AB_TRY(masked_median(&current_bg));
    }
    res->iterations_run = iterations_run;
    res->final_background = current_bg;
    res->converged = converged;
    if (n > 0) {
        hipLaunchKernelGGL(clamp01_kernel, dim3(grid), dim3(kBlock), 0, ctx->stream, work, n);
        AB_HIP(ctx, hipGetLastError());
    }
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
    AB_TRY(masked_stretch_device(ctx, in.dptr, mk.dptr, in.rows * in.cols, *cfg, so.dptr, res));
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
    AB_TRY(masked_stretch_device(ctx, in.dptr, mask, n, *cfg, so.dptr, res));
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
    if (n > 0) {
        hipLaunchKernelGGL(luminance_kernel, dim3(stream_grid(ctx, n)), dim3(kBlock), 0, ctx->stream, in[0].dptr, in[1].dptr, in[2].dptr, n, lum);
        AB_HIP(ctx, hipGetLastError());
    }
    const ab_star_mask_config mc = mask_config_of(*cfg);
    ab_star_mask_info mi;
    std::vector<StarDisc> discs;
    AB_TRY(detect_discs(ctx, lum, rows, cols, mc.detection_sigma, &discs));
    AB_TRY(star_mask_device(ctx, lum, rows, cols, discs, mc, mask, &mi, sc));
    if (shared) *shared = mi;
    for (int c = 0; c < 3; ++c) {
        AB_TRY(ab_stage_out_begin(ctx, outs[c], &so[c]));
        sc.outs.push_back(&so[c]);
        memset(&res3[c], 0, sizeof res3[c]);
        AB_TRY(masked_stretch_device(ctx, in[c].dptr, mask, n, *cfg, so[c].dptr, &res3[c]));
        res3[c].stars_masked = mi.stars_masked;
        res3[c].mask_coverage = mi.coverage_fraction;
    }
    for (int c = 0; c < 3; ++c) AB_TRY(ab_stage_out_finish(ctx, &so[c]));
    return AB_OK;
} AB_CATCH(ctx)

}  // extern "C"
