/* ORACLE (test infrastructure).  Restates the small plane-level helpers around the compose / calibration
 * callers: core/compose/lrgb.rs (apply_lrgb :4-45, synthesize_luminance :47-64), cmd/helpers.rs:175-202
 * (compute_linked_stf_with_stats), cmd/compose/color.rs:21-49 (calibrate_channel) and the plane-level bodies
 * of core/stacking/calibration.rs create_master_bias / _dark / _flat (:127-255; FITS loading excluded).
 * See ab_oracle.h for the rules. */
#include "ab_oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

static float clamp01(float v) { return v < 0.0f ? 0.0f : (v > 1.0f ? 1.0f : v); }

/* lrgb.rs:21-43 (dims are checked by the caller; mismatch = Err, :14-19) */
void orc_apply_lrgb(const float *l, float *r, float *g, float *b, size_t n, float lightness_weight, float chrominance_weight) {
    for (size_t i = 0; i < n; i++) {
        float rv = r[i], gv = g[i], bv = b[i], lum_new = l[i];
        float lum_old = rv * 0.2126f + gv * 0.7152f + bv * 0.0722f;
        if (lum_old < 1e-10f) {
            float blended = lum_new * lightness_weight;
            r[i] = g[i] = b[i] = blended;
            continue;
        }
        float ratio = (lum_new * lightness_weight + lum_old * (1.0f - lightness_weight)) / lum_old;
        float cb = chrominance_weight;
        r[i] = clamp01(rv * ratio * cb + lum_new * (1.0f - cb));
        g[i] = clamp01(gv * ratio * cb + lum_new * (1.0f - cb));
        b[i] = clamp01(bv * ratio * cb + lum_new * (1.0f - cb));
    }
}

/* lrgb.rs:47-64 (no finite guard, unlike masked_stretch.rs:143-154) */
void orc_synthesize_luminance(const float *r, const float *g, const float *b, size_t n, float *out) {
    for (size_t i = 0; i < n; i++) out[i] = r[i] * 0.2126f + g[i] * 0.7152f + b[i] * 0.0722f;
}

/* cmd/helpers.rs:185-202 */
void orc_compute_linked_stf(const orc_image_stats *sr, const orc_image_stats *sg, const orc_image_stats *sb, double target_bg,
                            double shadow_k, orc_stf_params *stf, orc_image_stats *combined) {
    orc_image_stats c;
    c.min = fmin(fmin(sr->min, sg->min), sb->min);
    c.max = fmax(fmax(sr->max, sg->max), sb->max);
    c.mean = (sr->mean + sg->mean + sb->mean) / 3.0;
    c.median = (sr->median + sg->median + sb->median) / 3.0;
    c.sigma = sqrt((sr->sigma * sr->sigma + sg->sigma * sg->sigma + sb->sigma * sb->sigma) / 3.0);
    c.mad = (sr->mad + sg->mad + sb->mad) / 3.0;
    c.valid_count = sr->valid_count;
    orc_auto_stf(&c, target_bg, shadow_k, stf);
    if (combined) *combined = c;
}

/* cmd/compose/color.rs:21-49 */
void orc_calibrate_channel(const float *orig, size_t n, float factor, const orc_image_stats *orig_stats, float *out,
                           orc_image_stats *stats) {
    for (size_t i = 0; i < n; i++) out[i] = orig[i] * factor;
    if (n <= 4000000u) { orc_compute_image_stats(out, n, stats); return; }
    double known_min, known_max;
    if (factor >= 0.0f) { known_min = orig_stats->min * (double)factor; known_max = orig_stats->max * (double)factor; }
    else { known_min = orig_stats->max * (double)factor; known_max = orig_stats->min * (double)factor; }
    orc_compute_image_stats_with_known_range(out, n, known_min, known_max, stats);
}

/* calibration.rs:127-255 on in-memory frames.  kind 0 bias (median), 1 dark (frames - bias), 2 flat
 * (frames - bias - dark * 1.0, median, normalised to mean 1 over the finite positive pixels, others -> 1.0). */
void orc_create_master(int kind, const float *const *frames, size_t n_frames, size_t npix, const float *master_bias,
                       const float *master_dark, float *out) {
    float **pre = (float **)malloc(n_frames * sizeof(float *));
    for (size_t f = 0; f < n_frames; f++) {
        pre[f] = (float *)malloc((npix ? npix : 1) * sizeof(float));
        for (size_t i = 0; i < npix; i++) {
            float v = frames[f][i];
            if (kind >= 1 && master_bias) v = v - master_bias[i];                 /* subtract_bias :15-17 */
            if (kind == 2 && master_dark) v = v - master_dark[i] * 1.0f;          /* subtract_dark(.., 1.0) :19-25 */
            pre[f][i] = v;
        }
    }
    orc_median_combine((const float *const *)pre, n_frames, npix, out);
    for (size_t f = 0; f < n_frames; f++) free(pre[f]);
    free(pre);
    if (kind != 2) return;
    double sum = 0.0;                                                              /* :228-247 */
    size_t count = 0;
    for (size_t i = 0; i < npix; i++) if (isfinite(out[i]) && out[i] > 0.0f) { sum += (double)out[i]; count++; }
    if (count > 0) {
        double mean = sum / (double)count;
        float inv_mean = fabs(mean) > 1e-10 ? 1.0f / (float)mean : 1.0f;
        for (size_t i = 0; i < npix; i++) out[i] = (isfinite(out[i]) && out[i] > 0.0f) ? out[i] * inv_mean : 1.0f;
    }
}
