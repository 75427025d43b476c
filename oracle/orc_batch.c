/* ORACLE (test infrastructure).  Restates core/imaging/calibration_pipeline.rs: calibrate_light (:74-118),
 * run_batch_pipeline's per-channel body (:157-190), compose_rgb_from_masters (:201-267), apply_luminance (:269-289),
 * normalize_channel (:291-307), normalize_frames (:309-319), sigma_clipped_mean_stack (:321-378).
 * The reference has NO tests for this file: parity unpinned beyond tests/test_oracle_batch_cases.py's independent
 * numpy restatement.  See ab_oracle.h for the rules. */
#include "ab_oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

void orc_calibrate_light(const float *light, size_t npix, const float *bias, size_t bias_len, const float *dark, size_t dark_len,
                         const float *flat, size_t flat_len, float *out) {                       /* :74-118 */
    int bias_ok = !bias || bias_len == npix, dark_ok = !dark || dark_len == npix, flat_ok = !flat || flat_len == npix;
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < (int64_t)npix; i++) {
        float v = light[i];
        if (bias_ok && bias) v -= bias[i];
        if (dark_ok && dark) v -= dark[i];
        if (flat_ok && flat) {
            float fv = flat[i];
            if (isfinite(fv) && fabsf(fv) > 1e-4f) v /= fv;
        }
        out[i] = v < 0.0f ? 0.0f : v;
    }
}

void orc_normalize_frame(const float *frame, size_t npix, float *out) {                         /* :309-319, one frame */
    double sum = 0.0;
    for (size_t i = 0; i < npix; i++) sum += (double)frame[i];                                  /* sequential, frame.iter() order */
    double mean = sum / (double)npix;
    if (mean > 0.0) {
        float inv_mean = 1.0f / (float)mean;
        for (size_t i = 0; i < npix; i++) out[i] = frame[i] * inv_mean;
    } else if (out != frame) {
        memcpy(out, frame, npix * sizeof(float));
    }
}

/* one pixel of sigma_clipped_mean_stack (:340-370); rejected[frame] is incremented per rejected sample */
static float scms_pixel(const float *const *frames, size_t n, size_t idx, float sigma_low, float sigma_high, size_t max_iter,
                        float *vals, size_t *owner, float *scratch, uint64_t *rejected) {
    size_t len = n;
    for (size_t i = 0; i < n; i++) { vals[i] = frames[i][idx]; owner[i] = i; }
    for (size_t it = 0; it < max_iter; it++) {
        if (len < 3) break;
        memcpy(scratch, vals, len * sizeof(float));
        size_t mid = len / 2;
        orc_select_nth_f32(scratch, len, mid);
        float median = scratch[mid];
        for (size_t i = 0; i < len; i++) scratch[i] = fabsf(scratch[i] - median);
        orc_select_nth_f32(scratch, len, mid);
        float sigma = (float)((double)scratch[mid] * 1.4826);
        if (sigma < 1e-10f) break;
        size_t kept = 0;
        for (size_t i = 0; i < len; i++) {
            float z = (vals[i] - median) / sigma;
            int keep = z > -sigma_low && z < sigma_high;
            if (!keep) { rejected[owner[i]]++; continue; }
            vals[kept] = vals[i]; owner[kept] = owner[i]; kept++;
        }
        if (kept == len) break;
        len = kept;
    }
    if (len == 0) return 0.0f;
    float s = 0.0f;
    for (size_t i = 0; i < len; i++) s += vals[i];
    return s / (float)len;
}

void orc_sigma_clipped_mean_stack(const float *const *frames, size_t n, size_t npix, float sigma_low, float sigma_high, size_t max_iter,
                                  float *out, uint64_t *rejection_counts) {                     /* :321-378 */
    memset(rejection_counts, 0, n * sizeof(uint64_t));
#pragma omp parallel
    {
        float *vals = (float *)malloc(n * sizeof(float)), *scratch = (float *)malloc(n * sizeof(float));
        size_t *owner = (size_t *)malloc(n * sizeof(size_t));
        uint64_t *local = (uint64_t *)calloc(n, sizeof(uint64_t));
#pragma omp for schedule(static)
        for (int64_t i = 0; i < (int64_t)npix; i++)
            out[i] = scms_pixel(frames, n, (size_t)i, sigma_low, sigma_high, max_iter, vals, owner, scratch, local);
#pragma omp critical
        for (size_t f = 0; f < n; f++) rejection_counts[f] += local[f];
        free(vals); free(scratch); free(owner); free(local);
    }
}

/* one channel of run_batch_pipeline (:157-190): calibrate every light, normalise, stack, mean / stddev of the master */
void orc_run_batch_channel(const float *const *lights, size_t n, size_t npix, const float *bias, size_t bias_len, const float *dark,
                           size_t dark_len, const float *flat, size_t flat_len, float sigma_low, float sigma_high, size_t max_iter,
                           int normalize, float *out, uint64_t *rejection_counts, double *mean_out, double *stddev_out) {
    float **cal = (float **)malloc(n * sizeof(float *));
    for (size_t f = 0; f < n; f++) {
        cal[f] = (float *)malloc(npix * sizeof(float));
        orc_calibrate_light(lights[f], npix, bias, bias_len, dark, dark_len, flat, flat_len, cal[f]);
    }
    if (normalize) {
#pragma omp parallel for schedule(dynamic)
        for (int64_t f = 0; f < (int64_t)n; f++) orc_normalize_frame(cal[f], npix, cal[f]);
    }
    orc_sigma_clipped_mean_stack((const float *const *)cal, n, npix, sigma_low, sigma_high, max_iter, out, rejection_counts);
    double s = 0.0;
    for (size_t i = 0; i < npix; i++) s += (double)out[i];
    double mean = s / (double)npix, var = 0.0;
    for (size_t i = 0; i < npix; i++) { double d = (double)out[i] - mean; var += d * d; }
    *mean_out = mean;
    *stddev_out = sqrt(var / (double)npix);
    for (size_t f = 0; f < n; f++) free(cal[f]);
    free(cal);
}

void orc_normalize_channel(const float *ch, size_t rows, size_t cols, size_t ld, float *out) {  /* :291-307 on a top-left crop */
    float mn = INFINITY, mx = -INFINITY;
    for (size_t y = 0; y < rows; y++)
        for (size_t x = 0; x < cols; x++) {
            float v = ch[y * ld + x];
            if (v < mn) mn = v;
            if (v > mx) mx = v;
        }
    float range = mx - mn;
    if (range < 1e-10f) { memset(out, 0, rows * cols * sizeof(float)); return; }
    float inv_range = 1.0f / range;
    for (size_t y = 0; y < rows; y++)
        for (size_t x = 0; x < cols; x++) {
            float t = (ch[y * ld + x] - mn) * inv_range;
            out[y * cols + x] = t < 0.0f ? 0.0f : (t > 1.0f ? 1.0f : t);                         /* f32::clamp: NaN stays NaN */
        }
}

/* compose_rgb_from_masters (:201-267) for found R, G, B (and optional L); out = h x w x 3 interleaved, h / w returned */
void orc_compose_rgb_from_masters(const float *r, size_t r_rows, size_t r_cols, const float *g, size_t g_rows, size_t g_cols,
                                  const float *b, size_t b_rows, size_t b_cols, const float *l, size_t l_rows, size_t l_cols,
                                  float *out, size_t *out_rows, size_t *out_cols) {
    size_t h = r_rows, w = r_cols;
    int same = g_rows == h && g_cols == w && b_rows == h && b_cols == w;
    if (!same) {
        h = h < g_rows ? h : g_rows; h = h < b_rows ? h : b_rows;
        w = w < g_cols ? w : g_cols; w = w < b_cols ? w : b_cols;
    }
    size_t n = h * w;
    float *rn = (float *)malloc(n * 4), *gn = (float *)malloc(n * 4), *bn = (float *)malloc(n * 4);
    orc_normalize_channel(r, h, w, r_cols, rn);
    orc_normalize_channel(g, h, w, g_cols, gn);
    orc_normalize_channel(b, h, w, b_cols, bn);
    if (same && l && l_rows == h && l_cols == w) {                                              /* :237-249 + apply_luminance */
        float *ln = (float *)malloc(n * 4);
        orc_normalize_channel(l, h, w, l_cols, ln);
        for (size_t i = 0; i < n; i++) {
            float rgb_lum = 0.2126f * rn[i] + 0.7152f * gn[i] + 0.0722f * bn[i];
            float scale = rgb_lum > 1e-10f ? ln[i] / rgb_lum : 1.0f;
            float ch[3] = {rn[i], gn[i], bn[i]};
            for (int c = 0; c < 3; c++) {
                float t = ch[c] * scale;
                out[i * 3 + c] = t < 0.0f ? 0.0f : (t > 1.0f ? 1.0f : t);
            }
        }
        free(ln);
    } else {
        for (size_t i = 0; i < n; i++) { out[i * 3] = rn[i]; out[i * 3 + 1] = gn[i]; out[i * 3 + 2] = bn[i]; }
    }
    free(rn); free(gn); free(bn);
    *out_rows = h; *out_cols = w;
}
