/* ORACLE (test infrastructure).  Restates the elementwise colour / tone / calibration functions:
 * core/imaging/scnr.rs, core/compose/channel_blend.rs, core/imaging/curves.rs,
 * core/imaging/stretch.rs:10-45, core/imaging/masked_stretch.rs:143-154 (luminance),
 * cmd/compose/color.rs:21-49 (white-balance scale), core/stacking/calibration.rs:47-125.
 * See ab_oracle.h for the rules.  channel_blend.rs / masked_stretch.rs have no reference tests:
 * parity unpinned beyond this restatement. */
#include "ab_oracle.h"
#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

static inline float clampf(float x, float lo, float hi) { return x < lo ? lo : (x > hi ? hi : x); }
static inline double clampd(double x, double lo, double hi) { return x < lo ? lo : (x > hi ? hi : x); }

/* scnr.rs:5-53; method 0 = AverageNeutral, 1 = MaximumNeutral */
void orc_apply_scnr_inplace(float *r, float *g, float *b, size_t n, int method, float amount_in, int preserve) {
    const float LUM_R = 0.2126f, LUM_G = 0.7152f, LUM_B = 0.0722f;
    const float INV_RB_WEIGHT = 1.0f / (LUM_R + LUM_B);
    float amount = clampf(amount_in, 0.0f, 1.0f);
    if (amount < 1e-7f) return;
    for (size_t i = 0; i < n; i++) {
        float rv = r[i], gv = g[i], bv = b[i];
        float limit = method == 0 ? (rv + bv) * 0.5f : fmaxf(rv, bv);   /* f32::max ignores NaN */
        float g_corrected = fminf(gv, limit);                            /* f32::min ignores NaN */
        float g_new = gv + amount * (g_corrected - gv);
        float delta_g = gv - g_new;
        if (preserve && delta_g > 1e-10f && rv <= 1.0f && bv <= 1.0f) {
            float lum_lost = LUM_G * delta_g;
            float boost = lum_lost * INV_RB_WEIGHT;
            float rv_new = rv + boost, bv_new = bv + boost;
            r[i] = rv_new > 1.0f ? 1.0f : rv_new;
            b[i] = bv_new > 1.0f ? 1.0f : bv_new;
        }
        g[i] = g_new;
    }
}

/* channel_blend.rs:13-70.  weights: n_weights rows of {channel_idx, r, g, b} as f64. */
void orc_blend_channels(const float *const *channels, size_t n_channels, const double *weights, size_t n_weights,
                        size_t npix, float *r_out, float *g_out, float *b_out) {
    for (size_t i = 0; i < npix; i++) {
        float rv = 0.0f, gv = 0.0f, bv = 0.0f;
        for (size_t w = 0; w < n_weights; w++) {
            size_t ch = (size_t)weights[4 * w];
            if (ch >= n_channels) continue;                              /* :20-23 filter */
            float rw = (float)weights[4 * w + 1], gw = (float)weights[4 * w + 2], bw = (float)weights[4 * w + 3];
            float v = channels[ch][i];
            rv += v * rw;
            gv += v * gw;
            bv += v * bw;
        }
        r_out[i] = rv; g_out[i] = gv; b_out[i] = bv;
    }
}

/* curves.rs:112-160 */
static double signum(double x) { return isnan(x) ? x : (signbit(x) ? -1.0 : 1.0); }
static void fritsch_carlson_tangents(const double *px, const double *py, size_t n, double *m) {
    if (n < 2) { for (size_t i = 0; i < n; i++) m[i] = 0.0; return; }
    if (n == 2) {
        double slope = (py[1] - py[0]) / fmax(px[1] - px[0], 1e-15);
        m[0] = m[1] = slope;
        return;
    }
    double *slopes = (double *)malloc((n - 1) * sizeof(double));
    for (size_t i = 0; i + 1 < n; i++) {
        double dx = fmax(px[i + 1] - px[i], 1e-15);
        slopes[i] = (py[i + 1] - py[i]) / dx;
    }
    m[0] = slopes[0];
    m[n - 1] = slopes[n - 2];
    for (size_t i = 1; i + 1 < n; i++) {
        if (signum(slopes[i - 1]) != signum(slopes[i])) m[i] = 0.0;
        else m[i] = (slopes[i - 1] + slopes[i]) * 0.5;
    }
    for (size_t i = 0; i + 1 < n; i++) {
        if (fabs(slopes[i]) < 1e-15) { m[i] = 0.0; m[i + 1] = 0.0; continue; }
        double alpha = m[i] / slopes[i], beta = m[i + 1] / slopes[i];
        double tau = alpha * alpha + beta * beta;
        if (tau > 9.0) {
            double s = 3.0 / sqrt(tau);
            m[i] = s * alpha * slopes[i];
            m[i + 1] = s * beta * slopes[i];
        }
    }
    free(slopes);
}

/* curves.rs:162-184 */
static double hermite_eval(const double *px, const double *py, const double *tan, size_t n, double x) {
    if (x <= px[0]) return py[0];
    if (x >= px[n - 1]) return py[n - 1];
    size_t seg = 0;
    for (size_t i = 1; i < n; i++) if (x < px[i]) { seg = i - 1; break; }
    double dx = fmax(px[seg + 1] - px[seg], 1e-15);
    double t = (x - px[seg]) / dx;
    double t2 = t * t, t3 = t2 * t;
    double h00 = 2.0 * t3 - 3.0 * t2 + 1.0;
    double h10 = t3 - 2.0 * t2 + t;
    double h01 = -2.0 * t3 + 3.0 * t2;
    double h11 = t3 - t2;
    return h00 * py[seg] + h10 * dx * tan[seg] + h01 * py[seg + 1] + h11 * dx * tan[seg + 1];
}

static int cmp_pt(const void *a, const void *b) {
    double xa = ((const double *)a)[0], xb = ((const double *)b)[0];
    return xa < xb ? -1 : (xa > xb ? 1 : 0);   /* partial_cmp, NaN -> Equal */
}

/* SplineLut::from_points, curves.rs:69-95.  points: n rows of (x, y). */
void orc_spline_lut_from_points(const double *points, size_t n_in, float *lut4096) {
    size_t cap = n_in + 2;
    double *pts = (double *)malloc(cap * 2 * sizeof(double));
    memcpy(pts, points, n_in * 2 * sizeof(double));
    /* stable sort by x (Rust sort_by is stable): insertion sort keeps it simple and stable */
    for (size_t i = 1; i < n_in; i++) {
        double x = pts[2 * i], y = pts[2 * i + 1];
        size_t j = i;
        while (j > 0 && cmp_pt(&pts[2 * (j - 1)], &x) > 0) { pts[2 * j] = pts[2 * (j - 1)]; pts[2 * j + 1] = pts[2 * (j - 1) + 1]; j--; }
        pts[2 * j] = x; pts[2 * j + 1] = y;
    }
    /* dedup_by(|a, b| |a.0 - b.0| < 1e-9): drop a when it matches the previous KEPT element b */
    size_t n = 0;
    for (size_t i = 0; i < n_in; i++) {
        if (n > 0 && fabs(pts[2 * i] - pts[2 * (n - 1)]) < 1e-9) continue;
        pts[2 * n] = pts[2 * i]; pts[2 * n + 1] = pts[2 * i + 1]; n++;
    }
    if (n == 0 || pts[0] > 1e-6) {                                       /* insert(0, (0,0)) */
        memmove(pts + 2, pts, n * 2 * sizeof(double));
        pts[0] = 0.0; pts[1] = 0.0; n++;
    }
    if (n == 0 || pts[2 * (n - 1)] < 1.0 - 1e-6) { pts[2 * n] = 1.0; pts[2 * n + 1] = 1.0; n++; }
    double *px = (double *)malloc(n * sizeof(double)), *py = (double *)malloc(n * sizeof(double)),
           *tan = (double *)malloc(n * sizeof(double));
    for (size_t i = 0; i < n; i++) { px[i] = pts[2 * i]; py[i] = pts[2 * i + 1]; }
    fritsch_carlson_tangents(px, py, n, tan);
    for (int i = 0; i < 4096; i++) {
        double t = (double)i / 4095.0;
        lut4096[i] = (float)clampd(hermite_eval(px, py, tan, n, t), 0.0, 1.0);
    }
    free(pts); free(px); free(py); free(tan);
}

/* curves.rs:104-108,186-197 */
void orc_apply_curve(const float *data, size_t n, const float *lut4096, float *out) {
    for (size_t i = 0; i < n; i++) {
        float v = data[i];
        if (!isfinite(v) || v < 0.0f) { out[i] = 0.0f; continue; }
        float t = clampf(v, 0.0f, 1.0f) * 4095.0f;
        size_t idx = t > 0.0f ? (size_t)t : 0;                            /* `as usize` */
        out[i] = lut4096[idx < 4095 ? idx : 4095];
    }
}

/* curves.rs:17-52 */
void orc_apply_levels(const float *data, size_t n, double black, double gamma, double white, float *out) {
    if (fabs(black) < 1e-7 && fabs(gamma - 1.0) < 1e-7 && fabs(white - 1.0) < 1e-7) {   /* is_identity */
        memcpy(out, data, n * sizeof(float));
        return;
    }
    double range = fmax(white - black, 1e-15);
    double inv_range = 1.0 / range;
    double inv_gamma = 1.0 / clampd(gamma, 0.01, 10.0);
    for (size_t i = 0; i < n; i++) {
        float v = data[i];
        if (!isfinite(v) || v < 0.0f) { out[i] = 0.0f; continue; }
        double norm = clampd(((double)v - black) * inv_range, 0.0, 1.0);
        out[i] = (float)pow(norm, inv_gamma);
    }
}

/* stretch.rs:10-45 */
void orc_arcsinh_stretch_with_stats(const float *data, size_t n, float dmin, float dmax, float factor, float gamma,
                                    float *out) {
    if (fabsf(factor) < 1e-10f) { memcpy(out, data, n * sizeof(float)); return; }
    float range = dmax - dmin;
    if (range < 1e-10f) { memset(out, 0, n * sizeof(float)); return; }
    float inv_range = 1.0f / range;
    float inv_denom = 1.0f / asinhf(factor);
    int apply_gamma = fabsf(gamma - 1.0f) > 1e-6f;
    for (size_t i = 0; i < n; i++) {
        float val = data[i];
        if (!isfinite(val)) { out[i] = 0.0f; continue; }
        float norm = clampf((val - dmin) * inv_range, 0.0f, 1.0f);
        float stretched = asinhf(norm * factor) * inv_denom;
        out[i] = apply_gamma ? powf(stretched, gamma) : stretched;
    }
}

/* masked_stretch.rs:143-154 */
void orc_luminance(const float *r, const float *g, const float *b, size_t n, float *out) {
    for (size_t i = 0; i < n; i++) {
        float rn = isfinite(r[i]) ? r[i] : 0.0f, gn = isfinite(g[i]) ? g[i] : 0.0f, bn = isfinite(b[i]) ? b[i] : 0.0f;
        out[i] = 0.2126f * rn + 0.7152f * gn + 0.0722f * bn;
    }
}

/* cmd/compose/color.rs:28,35-40 */
void orc_scale(const float *data, size_t n, float factor, float *out) {
    for (size_t i = 0; i < n; i++) out[i] = data[i] * factor;
}

/* calibration.rs:47-82; bias / dark / flat may be NULL */
void orc_calibrate_image(const float *raw, const float *bias, const float *dark, const float *flat, float dark_ratio,
                         size_t n, float *out) {
    for (size_t i = 0; i < n; i++) {
        float v = raw[i];
        if (bias) v -= bias[i];
        if (dark) v -= dark[i] * dark_ratio;
        if (flat) {
            float fv = flat[i];
            if (isfinite(fv) && fabsf(fv) > 1e-4f) v /= fv;
        }
        out[i] = v < 0.0f ? 0.0f : v;
    }
}

/* calibration.rs:84-125 median_combine_row_major */
void orc_median_combine(const float *const *planes, size_t n_frames, size_t npix, float *out) {
    float *vals = (float *)malloc((n_frames ? n_frames : 1) * sizeof(float));
    for (size_t i = 0; i < npix; i++) {
        size_t cnt = 0;
        for (size_t s = 0; s < n_frames; s++) {
            float v = planes[s][i];
            if (isfinite(v)) vals[cnt++] = v;
        }
        if (cnt == 0) { out[i] = 0.0f; continue; }
        size_t mid = cnt / 2;
        orc_select_nth_f32(vals, cnt, mid);
        out[i] = vals[mid];
    }
    free(vals);
}
