// Spectrophotometric colour calibration on gfx950.
//
// Replaces core/astrometry/spcc.rs: spcc_calibrate_rgb (:73-183), synthesize_luminance (:185-196),
// bp_rp_to_teff (:198-213), planck_rgb / planck_intensity (:215-243), white_reference_rgb (:245-255),
// estimate_bp_rp_from_flux (:275-279), cross_match_stars (:285-339), aperture_flux_f32 (:341-383),
// compute_correction_factors (:385-435).
//
// Device work: the luminance plane (12 B read + 4 B written per pixel), star detection and the
// luminance statistics (their own kernels), and the aperture photometry -- one lane per
// (star, channel) walks its aperture + annulus in the reference's raster order, so the f64 sums are
// bit-identical to the CPU restatement; at <= 200 stars x 3 channels this is latency, not bandwidth.
// Host work: the <= 200-star filter / sort and the Planck-law colour maths (scalar f64).
//
// The WCS enters the built-in catalogue path only through its pixel scale: the catalogue is
// synthesised from the detections' own sky positions (:257-273), so the cross-match is the identity
// whenever 0 < (pixel_scale * 3 / 3600)^2.  Header parsing (wcs.rs) stays with the caller.
#include "ab_common.hpp"

#include <algorithm>
#include <cmath>

namespace {

constexpr int kBlock = 256;

// synthesize_luminance (:185-196) and, in the same pass, the one number spcc_calibrate_rgb takes from compute_image_stats(luminance)
// (:88-89): its maximum over the valid pixels (finite and above the padding threshold, stats.rs:10-13; 0 without any, :95-97).  A
// valid pixel is positive, so its bit pattern orders like its value: one atomicMax per workgroup on the pattern, 0 = "none".
// (Round 4: the full statistics chain ran here for that one field, 0.3 ms of a 1.7 ms call.)
__global__ __launch_bounds__(kBlock) void spcc_luminance_kernel(const float *__restrict__ r, const float *__restrict__ g,
                                                                const float *__restrict__ b, int64_t n, float *__restrict__ out, unsigned int *max_bits) {
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    unsigned int mx = 0u;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) {
        const float l = 0.2126f * r[i] + 0.7152f * g[i] + 0.0722f * b[i];  // :193 (no finite guard here)
        out[i] = l;
        if (__builtin_isfinite(l) && l > 1e-7f) mx = max(mx, __float_as_uint(l));
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) mx = max(mx, (unsigned int)__shfl_xor((int)mx, off, 64));
    if ((threadIdx.x & 63) == 0 && mx) atomicMax(max_bits, mx);
}

struct Aperture {
    double x, y, radius;
};

__device__ __forceinline__ long long sat_index(double v) {  // `f64 as usize`, capped for 64-bit maths
    if (!(v > 0.0)) return 0;
    if (v >= 4.0e18) return 4000000000000000000LL;
    return (long long)v;
}

// aperture_flux_f32 (:341-383): lane = (star, channel), raster-order f64 sums
__global__ __launch_bounds__(64) void aperture_flux_kernel(const float *__restrict__ r, const float *__restrict__ g, const float *__restrict__ b,
                                                           int h, int w, const Aperture *__restrict__ aps, int n_stars,
                                                           double *__restrict__ out /* n_stars x 3 */) {
    const int t = blockIdx.x * 64 + threadIdx.x;
    if (t >= n_stars * 3) return;
    const int s = t / 3, c = t % 3;
    const float *img = c == 0 ? r : (c == 1 ? g : b);
    const Aperture a = aps[s];
    const double r2 = a.radius * a.radius, inner = a.radius * 1.2, outer = a.radius * 1.8;
    const double inner_r2 = inner * inner, outer_r2 = outer * outer;
    const long long y_min = sat_index(fmax(floor(a.y - outer), 0.0)), y_max = min(sat_index(ceil(a.y + outer)), (long long)(h - 1));
    const long long x_min = sat_index(fmax(floor(a.x - outer), 0.0)), x_max = min(sat_index(ceil(a.x + outer)), (long long)(w - 1));
    double flux = 0.0, bg_sum = 0.0;
    unsigned int bg_count = 0;
    for (long long py = y_min; py <= y_max; ++py)
        for (long long px = x_min; px <= x_max; ++px) {
            const double dx = (double)px - a.x, dy = (double)py - a.y;
            const double d2 = dx * dx + dy * dy;
            const double v = (double)img[py * w + px];
            if (d2 <= r2) {
                flux += v;
            } else if (d2 >= inner_r2 && d2 <= outer_r2) {
                bg_sum += v;
                ++bg_count;
            }
        }
    if (bg_count > 0) {
        const double bg_per_pixel = bg_sum / (double)bg_count;
        flux -= bg_per_pixel * (3.14159265358979323846264338327950288 * r2);
    }
    out[t] = flux > 0.0 ? flux : 0.0;  // f64::max(0.0): NaN -> 0.0
}

double clampd(double v, double lo, double hi) { return v < lo ? lo : (v > hi ? hi : v); }

double bp_rp_to_teff(double bp_rp) {  // :198-213
    const double x = clampd(bp_rp, -0.5, 5.0);
    if (x < 0.0) return 10000.0 + (-x) * 20000.0;
    if (x < 0.5) return 7500.0 + (0.5 - x) * 5000.0;
    if (x < 1.0) return 5800.0 + (1.0 - x) * 3400.0;
    if (x < 1.5) return 4500.0 + (1.5 - x) * 2600.0;
    if (x < 2.5) return 3500.0 + (2.5 - x) * 1000.0;
    return 2800.0 + (5.0 - x) * 280.0;
}

double planck_intensity(double teff, double wavelength_nm) {  // :228-243
    const double lambda = wavelength_nm * 1e-9, h = 6.626e-34, c = 2.998e8, k = 1.381e-23;
    const double exponent = h * c / (lambda * k * teff);
    if (exponent > 500.0) return 0.0;
    const double l2 = lambda * lambda;
    const double l5 = lambda * (l2 * l2);  // powi(5)
    const double numerator = 2.0 * h * c * c / l5;
    return numerator / (std::exp(exponent) - 1.0);
}

void planck_rgb(double teff, double out[3]) {  // :215-226
    const double r = planck_intensity(teff, 640.0), g = planck_intensity(teff, 530.0), b = planck_intensity(teff, 460.0);
    const double max_val = std::fmax(std::fmax(r, g), b);
    if (max_val < 1e-30) {
        out[0] = out[1] = out[2] = 1.0;
        return;
    }
    out[0] = r / max_val, out[1] = g / max_val, out[2] = b / max_val;
}

void white_reference_rgb(int kind, const double custom[3], double out[3]) {  // :245-255
    if (kind == 1) {
        planck_rgb(5778.0, out);
    } else if (kind == 0) {
        planck_rgb(5500.0, out);
        out[0] *= 0.98, out[1] *= 1.0, out[2] *= 1.02;
    } else if (kind == 2) {
        out[0] = out[1] = out[2] = 1.0;
    } else {
        memcpy(out, custom, 3 * sizeof(double));
    }
}

double estimate_bp_rp_from_flux(const ab_detected_star &s) {  // :275-279
    const double norm_flux = clampd(s.flux / std::fmax(s.peak, 1e-10), 0.1, 100.0);
    const double fwhm_factor = clampd(s.fwhm - 3.0, -2.0, 5.0) * 0.1;
    return clampd(1.0 / std::sqrt(norm_flux) + fwhm_factor, -0.3, 4.0);
}

struct Matched {
    double bp_rp, r, g, b;
};

void compute_correction_factors(const std::vector<Matched> &m, const double wr[3], ab_spcc_result *res) {  // :385-435
    double sr = 0.0, sg = 0.0, sb = 0.0, sw = 0.0, sci = 0.0;
    for (const Matched &s : m) {
        double e[3];
        planck_rgb(bp_rp_to_teff(s.bp_rp), e);
        const double total_measured = s.r + s.g + s.b, total_expected = e[0] + e[1] + e[2];
        if (total_measured < 1e-10 || total_expected < 1e-10) continue;
        const double weight = std::sqrt(total_measured);
        const double mr = s.r / total_measured, mg = s.g / total_measured, mb = s.b / total_measured;
        const double er = e[0] / total_expected, eg = e[1] / total_expected, eb = e[2] / total_expected;
        if (mr > 1e-6) sr += (er / mr) * weight;
        if (mg > 1e-6) sg += (eg / mg) * weight;
        if (mb > 1e-6) sb += (eb / mb) * weight;
        sw += weight;
        sci += s.bp_rp;
    }
    if (sw < 1e-10 || m.empty()) {
        res->r_factor = res->g_factor = res->b_factor = 1.0;
        res->avg_color_index = 0.0;
        return;
    }
    double rf = sr / sw, gf = sg / sw, bf = sb / sw;
    rf *= wr[0], gf *= wr[1], bf *= wr[2];
    const double norm = gf;
    if (norm > 1e-10) {
        rf /= norm;
        gf = 1.0;
        bf /= norm;
    }
    res->r_factor = rf, res->g_factor = gf, res->b_factor = bf;
    res->avg_color_index = sci / (double)m.size();
}

// spcc.rs:90-183 on device planes and a given detection
int spcc_from_detection(ab_ctx *ctx, const float *r, const float *g, const float *b, int64_t h, int64_t w,
                        const std::vector<ab_detected_star> &stars, double lum_max, double pixel_scale, const ab_spcc_config &cfg,
                        ab_spcc_result *res) {
    memset(res, 0, sizeof *res);
    const float sat_limit = (float)(lum_max * cfg.saturation_limit);  // :89
    const double x_hi = (double)(uint64_t)(w - 10), y_hi = (double)(uint64_t)(h - 10);  // usize wrap below 10, as in release builds
    std::vector<const ab_detected_star *> good;
    for (const ab_detected_star &s : stars)
        if (s.snr >= cfg.min_snr && s.peak < (double)sat_limit && s.x >= 10.0 && s.y >= 10.0 && s.x < x_hi && s.y < y_hi) good.push_back(&s);
    std::stable_sort(good.begin(), good.end(), [](const ab_detected_star *a, const ab_detected_star *b) { return a->snr > b->snr; });  // :105
    if (good.size() > cfg.max_stars) good.resize((size_t)cfg.max_stars);
    res->stars_total = good.size();
    if (good.size() < 5)
        return ab_set_error(ctx, AB_ERR_INVALID, "Only %zu stars passed quality filters (need 5+). Try lowering min_snr.", good.size());
    const double match_radius = (pixel_scale * 3.0) / 3600.0, match_r2 = match_radius * match_radius;
    std::vector<Matched> matched;
    if (0.0 < match_r2) {  // identity cross-match (see file header)
        std::vector<Aperture> aps;
        for (const ab_detected_star *s : good) aps.push_back({s->x, s->y, std::fmax(s->fwhm * 1.5, 3.0)});
        Aperture *daps = nullptr;
        double *dflux = nullptr;
        std::vector<double> flux(aps.size() * 3);
        AB_TRY(ab_workspace(ctx, AB_WS_SCOPE1, aps.size() * sizeof(Aperture), (void **)&daps));  // (kept between calls, see ab_common.hpp)
        AB_TRY(ab_workspace(ctx, AB_WS_SCOPE2, flux.size() * sizeof(double), (void **)&dflux));
        hipError_t e = hipSuccess;
        if (e == hipSuccess) e = hipMemcpyAsync(daps, aps.data(), aps.size() * sizeof(Aperture), hipMemcpyHostToDevice, ctx->stream);
        if (e == hipSuccess) {
            const int lanes = (int)aps.size() * 3;
            hipLaunchKernelGGL(aperture_flux_kernel, dim3((lanes + 63) / 64), dim3(64), 0, ctx->stream, r, g, b, (int)h, (int)w, daps,
                               (int)aps.size(), dflux);
            e = hipGetLastError();
        }
        if (e == hipSuccess) e = hipMemcpyAsync(flux.data(), dflux, flux.size() * sizeof(double), hipMemcpyDeviceToHost, ctx->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
        if (e != hipSuccess) return ab_set_error(ctx, AB_ERR_HIP, "aperture photometry failed: %s", hipGetErrorString(e));
        for (size_t i = 0; i < good.size(); ++i) {
            const double rf = flux[3 * i], gf = flux[3 * i + 1], bf = flux[3 * i + 2];
            if (rf > 0.0 && gf > 0.0 && bf > 0.0) matched.push_back({estimate_bp_rp_from_flux(*good[i]), rf, gf, bf});  // :326-333
        }
    }
    res->stars_matched = matched.size();
    if (matched.size() < 3)
        return ab_set_error(ctx, AB_ERR_INVALID, "Only %zu stars cross-matched (need 3+). Check WCS solution quality.", matched.size());
    double wr[3];
    white_reference_rgb(cfg.white_reference, cfg.custom, wr);
    compute_correction_factors(matched, wr, res);
    return AB_OK;
}

struct Staged3 {
    StagedPlane p[3];
    ab_ctx *ctx;
    int n = 0;
    explicit Staged3(ab_ctx *c) : ctx(c) {}
    ~Staged3() {
        for (int i = 0; i < n; ++i) ab_stage_release(ctx, &p[i]);
    }
    int stage(const ab_plane *r, const ab_plane *g, const ab_plane *b) {
        const ab_plane *in[3] = {r, g, b};
        for (int i = 0; i < 3; ++i) {
            AB_TRY(ab_stage_in(ctx, in[i], &p[i]));
            n = i + 1;
        }
        return AB_OK;
    }
};

int check_planes(ab_ctx *ctx, const ab_plane *r, const ab_plane *g, const ab_plane *b) {
    AB_CHECK(ctx, r && g && b, "null plane");
    AB_CHECK(ctx, g->rows == r->rows && g->cols == r->cols && b->rows == r->rows && b->cols == r->cols, "SPCC channels must share dims");
    AB_CHECK(ctx, r->rows * r->cols < (int64_t(1) << 31), "image too large for this build");
    return AB_OK;
}

}  // namespace

extern "C" {

int ab_spcc_white_reference_rgb(int32_t kind, const double custom[3], double out[3]) try {
    if (!out || (kind == 3 && !custom) || kind < 0 || kind > 3) return AB_ERR_INVALID;
    white_reference_rgb(kind, custom, out);
    return AB_OK;
} AB_CATCH_NOCTX

int ab_spcc_from_detection(ab_ctx *ctx, const ab_plane *r, const ab_plane *g, const ab_plane *b, const ab_detected_star *stars,
                           size_t n_stars, double lum_max, double pixel_scale_arcsec, const ab_spcc_config *cfg, ab_spcc_result *res) try {
    if (!ctx) return AB_ERR_INVALID;
    AB_CHECK(ctx, cfg && res && (stars || n_stars == 0), "null argument");
    AB_TRY(check_planes(ctx, r, g, b));
    AB_HIP(ctx, hipSetDevice(ctx->device));
    Staged3 st(ctx);
    AB_TRY(st.stage(r, g, b));
    std::vector<ab_detected_star> v(stars, stars + n_stars);
    return spcc_from_detection(ctx, st.p[0].dptr, st.p[1].dptr, st.p[2].dptr, r->rows, r->cols, v, lum_max, pixel_scale_arcsec, *cfg, res);
} AB_CATCH(ctx)

int ab_spcc_calibrate_rgb(ab_ctx *ctx, const ab_plane *r, const ab_plane *g, const ab_plane *b, double pixel_scale_arcsec,
                          const ab_spcc_config *cfg, ab_spcc_result *res) try {
    if (!ctx) return AB_ERR_INVALID;
    AB_CHECK(ctx, cfg && res, "null argument");
    AB_TRY(check_planes(ctx, r, g, b));
    AB_HIP(ctx, hipSetDevice(ctx->device));
    Staged3 st(ctx);
    AB_TRY(st.stage(r, g, b));
    const int64_t h = r->rows, w = r->cols, n = h * w;
    float *lum = nullptr;
    AB_TRY(ab_workspace(ctx, AB_WS_SCOPE0, std::max<size_t>((size_t)n, 1) * sizeof(float), (void **)&lum));
    int rc = AB_OK;
    std::vector<ab_detected_star> stars;
    unsigned int *dmax = nullptr;
    AB_TRY(ab_workspace(ctx, AB_WS_SCOPE3, 64, (void **)&dmax));
    unsigned int bits = 0u;  // the bit pattern of the luminance's largest valid pixel (0: none)
    if (n > 0) {
        AB_HIP(ctx, hipMemsetAsync(dmax, 0, sizeof(unsigned int), ctx->stream));
        const int grid = (int)std::min<int64_t>((n + kBlock - 1) / kBlock, (int64_t)(ctx->cu_count > 0 ? ctx->cu_count : 256) * 8);
        hipLaunchKernelGGL(spcc_luminance_kernel, dim3(grid), dim3(kBlock), 0, ctx->stream, st.p[0].dptr, st.p[1].dptr, st.p[2].dptr, n, lum, dmax);
        if (hipGetLastError() != hipSuccess) rc = ab_set_error(ctx, AB_ERR_HIP, "luminance launch failed");
        if (rc == AB_OK && hipMemcpyAsync(&bits, dmax, sizeof bits, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess)
            rc = ab_set_error(ctx, AB_ERR_HIP, "hipMemcpyAsync failed");
        if (rc == AB_OK && hipStreamSynchronize(ctx->stream) != hipSuccess) rc = ab_set_error(ctx, AB_ERR_HIP, "hipStreamSynchronize failed");
    }
    float lum_max_f;
    memcpy(&lum_max_f, &bits, sizeof lum_max_f);
    const double lum_max = bits ? (double)lum_max_f : 0.0;  // compute_image_stats(lum).max (:88)
    double bm, bs;
    if (rc == AB_OK) rc = ab_detect_stars_device(ctx, lum, h, w, w, 5.0, &stars, &bm, &bs);  // :86
    (void)hipStreamSynchronize(ctx->stream);
    if (rc != AB_OK) return rc;
    return spcc_from_detection(ctx, st.p[0].dptr, st.p[1].dptr, st.p[2].dptr, h, w, stars, lum_max, pixel_scale_arcsec, *cfg, res);
} AB_CATCH(ctx)

}  // extern "C"
