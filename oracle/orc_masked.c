/* ORACLE (test infrastructure).  Restates core/imaging/star_mask.rs (generate_star_mask :38-44,
 * generate_star_mask_from_detection :46-138) and core/imaging/masked_stretch.rs
 * (masked_stretch :44-58, masked_stretch_with_mask :60-118, normalize_to_01 :195-212,
 * compute_masked_median :214-230, mtf_balance :232-238, apply_mtf :240-255, clamp_inplace :257-259).
 * See ab_oracle.h for the rules. */
#include "ab_oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

static size_t sat_usize(double v) {                    /* Rust `f64 as usize`: saturating, NaN -> 0 */
    if (!(v > 0.0)) return 0;
    if (v >= 18446744073709551615.0) return (size_t)-1;
    return (size_t)v;
}

/* star_mask.rs:46-138 on the (x, y, fwhm) of the detection's stars.  Returns stars_masked. */
size_t orc_star_mask_from_stars(const float *image, size_t h, size_t w, const double *xs, const double *ys,
                                const double *fwhms, size_t n_stars, double growth_factor, double softness,
                                double min_fwhm, double max_fwhm, int luminance_protect, double luminance_ceiling,
                                float *mask, double *coverage_out) {
    size_t npix = h * w, star_count = 0;
    memset(mask, 0, npix * sizeof(float));
    for (size_t s = 0; s < n_stars; s++) {
        if (!(fwhms[s] >= min_fwhm && fwhms[s] <= max_fwhm)) continue;           /* :54-58 */
        star_count++;
        double radius = fwhms[s] * growth_factor, soft_radius = radius + softness;
        size_t y_min = sat_usize(fmax(floor(ys[s] - soft_radius), 0.0));
        size_t y_max = sat_usize(ceil(ys[s] + soft_radius)), x_max = sat_usize(ceil(xs[s] + soft_radius));
        size_t hm = h ? h - 1 : 0, wm = w ? w - 1 : 0;
        if (y_max > hm) y_max = hm;
        if (x_max > wm) x_max = wm;
        size_t x_min = sat_usize(fmax(floor(xs[s] - soft_radius), 0.0));
        double r2_inner = radius * radius, r2_outer = soft_radius * soft_radius;
        double fade_range = fmax(r2_outer - r2_inner, 1e-10);
        if (npix == 0) continue;
        for (size_t py = y_min; py <= y_max; py++)
            for (size_t px = x_min; px <= x_max; px++) {
                double dx = (double)px - xs[s], dy = (double)py - ys[s];
                double d2 = dx * dx + dy * dy;
                float val;
                if (d2 <= r2_inner) val = 1.0f;
                else if (d2 <= r2_outer) {
                    float t = (float)((d2 - r2_inner) / fade_range);
                    float smooth = t * t * (3.0f - 2.0f * t);
                    val = 1.0f - smooth;
                } else continue;
                if (val > mask[py * w + px]) mask[py * w + px] = val;             /* :106-113 */
            }
    }
    if (luminance_protect) {                                                      /* :115-132 */
        float ceiling = (float)luminance_ceiling;
        float inv_range = ceiling < 1.0f ? 1.0f / (1.0f - ceiling) : 1.0f;
        for (size_t i = 0; i < npix; i++) {
            float pixel = image[i];
            if (pixel > ceiling && mask[i] < 1.0f) {
                float excess = (pixel - ceiling) * inv_range;
                excess = excess < 0.0f ? 0.0f : (excess > 1.0f ? 1.0f : excess);
                float smooth = excess * excess * (3.0f - 2.0f * excess);
                if (smooth > mask[i]) mask[i] = smooth;
            }
        }
    }
    size_t covered = 0;
    for (size_t i = 0; i < npix; i++) if (mask[i] > 0.01f) covered++;
    if (coverage_out) *coverage_out = (double)covered / (double)npix;             /* :134-135 (0/0 = NaN when empty) */
    return star_count;
}

/* generate_star_mask (:38-44): detect_stars(image, detection_sigma) then the above */
size_t orc_generate_star_mask(const float *image, size_t h, size_t w, double growth_factor, double softness,
                              double detection_sigma, double min_fwhm, double max_fwhm, int luminance_protect,
                              double luminance_ceiling, float *mask, double *coverage_out) {
    size_t cap = 1u << 16, total = 0;
    double bm, bs;
    orc_star *st = (orc_star *)malloc(cap * sizeof(orc_star));
    size_t n = orc_detect_stars(image, h, w, detection_sigma, st, cap, &total, &bm, &bs);
    if (total > cap) {
        cap = total;
        st = (orc_star *)realloc(st, cap * sizeof(orc_star));
        n = orc_detect_stars(image, h, w, detection_sigma, st, cap, &total, &bm, &bs);
    }
    double *xs = (double *)malloc((n ? n : 1) * 3 * sizeof(double)), *ys = xs + n, *fw = ys + n;
    for (size_t i = 0; i < n; i++) { xs[i] = st[i].x; ys[i] = st[i].y; fw[i] = st[i].fwhm; }
    size_t c = orc_star_mask_from_stars(image, h, w, xs, ys, fw, n, growth_factor, softness, min_fwhm, max_fwhm,
                                        luminance_protect, luminance_ceiling, mask, coverage_out);
    free(xs);
    free(st);
    return c;
}

static void normalize_to_01(const float *image, size_t n, float *out) {            /* :195-212 */
    orc_image_stats st;
    orc_compute_image_stats(image, n, &st);
    float range = (float)(st.max - st.min);
    if (range < 1e-10f) { memset(out, 0, n * sizeof(float)); return; }
    float dmin = (float)st.min, inv = 1.0f / range;
    for (size_t i = 0; i < n; i++) {
        float v = image[i];
        if (!isfinite(v) || v <= 0.0f) out[i] = 0.0f;
        else {
            float t = (v - dmin) * inv;
            out[i] = t < 0.0f ? 0.0f : (t > 1.0f ? 1.0f : t);
        }
    }
}

static double masked_median(const float *img, const float *mask, size_t n, float *tmp) {   /* :214-230 */
    size_t k = 0;
    for (size_t i = 0; i < n; i++) if (mask[i] < 0.5f && isfinite(img[i]) && img[i] > 0.0f) tmp[k++] = img[i];
    if (k == 0) return 0.0;
    size_t mid = k / 2;
    orc_select_nth_f32(tmp, k, mid);
    return (double)tmp[mid];
}

static double mtf_balance(double median, double target) {                          /* :232-238 */
    double denom = 2.0 * target * median - target - median;
    if (fabs(denom) < 1e-15) return 0.5;
    double v = median * (target - 1.0) / denom;
    return v < 0.0001 ? 0.0001 : (v > 0.9999 ? 0.9999 : v);
}

static float mtf_f32(float x, float m) {                                           /* :240-255 */
    if (x <= 0.0f) return 0.0f;
    if (x >= 1.0f) return 1.0f;
    float denom = (2.0f * m - 1.0f) * x - m;
    if (fabsf(denom) < 1e-10f) return x;
    float v = (m - 1.0f) * x / denom;
    return v < 0.0f ? 0.0f : (v > 1.0f ? 1.0f : v);                                /* f32::clamp (NaN stays NaN) */
}

/* masked_stretch_with_mask (:60-118).  out: n floats.  info: iterations_run, converged; final_bg. */
void orc_masked_stretch_with_mask(const float *image, const float *mask, size_t n, size_t iterations, double target_bg,
                                  double protection_amount, double convergence_threshold, float *out,
                                  size_t *iterations_run_out, double *final_bg_out, int *converged_out) {
    float *tmp = (float *)malloc((n ? n : 1) * sizeof(float));
    normalize_to_01(image, n, out);
    float protection = (float)protection_amount;
    double prev_bg = masked_median(out, mask, n, tmp);
    size_t iterations_run = 0;
    int converged = 0;
    for (size_t it = 0; it < iterations; it++) {
        iterations_run = it + 1;
        double bg = masked_median(out, mask, n, tmp);
        int at_target = fabs(bg - target_bg) < convergence_threshold;
        int stagnated = it > 0 && fabs(bg - prev_bg) < convergence_threshold * 0.1;
        if (at_target) { converged = 1; break; }
        if (stagnated) break;
        float m = (float)mtf_balance(bg, target_bg);
        for (size_t i = 0; i < n; i++) {
            float stretched = mtf_f32(out[i], m);
            float blend = mask[i] * protection;
            out[i] = out[i] * blend + stretched * (1.0f - blend);
        }
        prev_bg = bg;
    }
    double final_bg = masked_median(out, mask, n, tmp);
    for (size_t i = 0; i < n; i++) {                                               /* clamp_inplace :257-259 */
        float v = out[i];
        out[i] = v < 0.0f ? 0.0f : (v > 1.0f ? 1.0f : v);
    }
    free(tmp);
    if (iterations_run_out) *iterations_run_out = iterations_run;
    if (final_bg_out) *final_bg_out = final_bg;
    if (converged_out) *converged_out = converged;
}
