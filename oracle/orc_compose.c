/* ORACLE (test infrastructure).  Restates core/compose/rgb.rs (harmonize_dimensions :42-125,
 * apply_multiplier_inplace :127-130, channel_or_synth :132-151, merge_for_stf :153-163,
 * align_channels :165-189, apply_stf_inplace :191-207, process_rgb :209-323),
 * core/compose/white_balance.rs:3-20, core/imaging/resample.rs:25-61 and
 * core/alignment/pair.rs:41-77 (align_pair).  See ab_oracle.h for the rules. */
#include "ab_oracle.h"
#include <float.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

/* resample.rs:25-61.  Returns 1 on "Target dimensions must be > 0". */
int orc_resample_image(const float *src, size_t src_rows, size_t src_cols, size_t target_rows, size_t target_cols,
                       float *out) {
    if (target_rows == 0 || target_cols == 0) return 1;
    if (target_rows == src_rows && target_cols == src_cols) {
        memcpy(out, src, src_rows * src_cols * sizeof(float));
        return 0;
    }
    double scale_y = (double)src_rows / (double)target_rows, scale_x = (double)src_cols / (double)target_cols;
    double half_shift_y = (scale_y - 1.0) * 0.5, half_shift_x = (scale_x - 1.0) * 0.5;
#pragma omp parallel for schedule(static)
    for (size_t ty = 0; ty < target_rows; ty++) {
        double sy = (double)ty * scale_y + half_shift_y;
        for (size_t tx = 0; tx < target_cols; tx++) {
            double sx = (double)tx * scale_x + half_shift_x;
            out[ty * target_cols + tx] = orc_bicubic_sample(src, src_rows, src_cols, sy, sx);
        }
    }
    return 0;
}

/* white_balance.rs:3-20 */
void orc_select_wb_reference(const orc_image_stats *sr, const orc_image_stats *sg, const orc_image_stats *sb,
                             double out[3]) {
    double stab_r = sr->median > 1e-10 ? sr->mad / sr->median : DBL_MAX;
    double stab_g = sg->median > 1e-10 ? sg->mad / sg->median : DBL_MAX;
    double stab_b = sb->median > 1e-10 ? sb->mad / sb->median : DBL_MAX;
    double mr = fmax(sr->median, 1e-10), mg = fmax(sg->median, 1e-10), mb = fmax(sb->median, 1e-10);
    if (stab_r <= stab_g && stab_r <= stab_b) { out[0] = 1.0; out[1] = mr / mg; out[2] = mr / mb; }
    else if (stab_b <= stab_g) { out[0] = mb / mr; out[1] = mb / mg; out[2] = 1.0; }
    else { out[0] = mg / mr; out[1] = 1.0; out[2] = mg / mb; }
}

/* rgb.rs:191-207 (NOT stf.rs's StfTransform: multiplies by 1/range but divides by clip_range) */
void orc_compose_apply_stf_inplace(float *data, size_t n, const orc_stf_params *p, const orc_image_stats *st) {
    double range = fmax(st->max - st->min, 1e-30), inv_range = 1.0 / range, dmin = st->min;
    double shadow = p->shadow, clip_range = fmax(p->highlight - shadow, 1e-15), m = p->midtone;
#pragma omp parallel for schedule(static)
    for (size_t i = 0; i < n; i++) {
        float v = data[i];
        if (!isfinite(v) || v <= 1e-7f) { data[i] = 0.0f; continue; }
        double norm = ((double)v - dmin) * inv_range;
        double clipped = (norm - shadow) / clip_range;
        clipped = clipped < 0.0 ? 0.0 : (clipped > 1.0 ? 1.0 : clipped);
        if (clipped <= 0.0) { data[i] = 0.0f; continue; }
        if (clipped >= 1.0) { data[i] = 1.0f; continue; }
        data[i] = (float)((m - 1.0) * clipped / ((2.0 * m - 1.0) * clipped - m));
    }
}

static float *clone_plane(const float *src, size_t n) {
    float *p = (float *)malloc((n ? n : 1) * sizeof(float));
    memcpy(p, src, n * sizeof(float));
    return p;
}

/* rgb.rs:132-151 */
static float *channel_or_synth(const float *primary, const float *alt1, const float *alt2, size_t n) {
    if (primary) return clone_plane(primary, n);
    if (alt1 && alt2) {
        float *o = (float *)malloc((n ? n : 1) * sizeof(float));
        for (size_t i = 0; i < n; i++) o[i] = (alt1[i] + alt2[i]) * 0.5f;
        return o;
    }
    if (alt1) return clone_plane(alt1, n);
    if (alt2) return clone_plane(alt2, n);
    return (float *)calloc(n ? n : 1, sizeof(float));
}

/* pair.rs:41-77 (align_pair): returns the aligned plane (malloc), offset = (dy, dx) / (ty, tx) */
static float *align_pair(const float *reference, const float *target, size_t rows, size_t cols, int method, int threads,
                         double offset[2]) {
    float *out = (float *)malloc((rows * cols ? rows * cols : 1) * sizeof(float));
    if (method == 0) {
        double dx, dy, conf;
        orc_phase_correlate(reference, rows, cols, target, rows, cols, &dx, &dy, &conf);
        orc_shift_image_subpixel(target, rows, cols, dy, dx, 0, out);
        offset[0] = dy;
        offset[1] = dx;
    } else {
        orc_affine_result res;
        orc_align_channel_affine(reference, target, rows, cols, threads, &res);
        orc_warp_image(target, rows, cols, res.t, rows, cols, 0, out);
        offset[0] = res.t[5];
        offset[1] = res.t[2];
    }
    return out;
}

/* process_rgb (rgb.rs:209-323).  Channels may be NULL (absent).  out_* / pre_*: max_rows * max_cols floats
 * (pre_* nullable).  Returns 0 ok, 1 "Need at least 2 channels", 2 dimension ratio exceeded (message in err). */
int orc_process_rgb(const float *r, size_t r_rows, size_t r_cols, const float *g, size_t g_rows, size_t g_cols,
                    const float *b, size_t b_rows, size_t b_cols, const orc_rgb_config *cfg, float *out_r, float *out_g,
                    float *out_b, float *pre_r, float *pre_g, float *pre_b, orc_rgb_result *res, char *err,
                    size_t err_cap) {
    const float *ch[3] = {r, g, b};
    size_t rws[3] = {r_rows, g_rows, b_rows}, cls[3] = {r_cols, g_cols, b_cols};
    int count = (r != NULL) + (g != NULL) + (b != NULL);
    memset(res, 0, sizeof *res);
    if (count < 2) {
        if (err) snprintf(err, err_cap, "Need at least 2 channels for RGB compose (got %d)", count);
        return 1;
    }
    /* harmonize_dimensions :42-125 */
    size_t min_rows = (size_t)-1, min_cols = (size_t)-1, max_rows = 0, max_cols = 0;
    for (int c = 0; c < 3; c++) if (ch[c]) {
        if (rws[c] < min_rows) min_rows = rws[c];
        if (cls[c] < min_cols) min_cols = cls[c];
        if (rws[c] > max_rows) max_rows = rws[c];
        if (cls[c] > max_cols) max_cols = cls[c];
    }
    float *harm[3] = {NULL, NULL, NULL};
    if (!(max_rows == min_rows && max_cols == min_cols)) {
        double ratio_rows = (double)max_rows / (double)(min_rows > 1 ? min_rows : 1);
        double ratio_cols = (double)max_cols / (double)(min_cols > 1 ? min_cols : 1);
        double ratio = fmax(ratio_rows, ratio_cols);
        if (ratio > 8.0) {                                                   /* MAX_DIMENSION_RATIO, constants.rs:165 */
            if (err) {
                static const char *names = "RGB";
                int k = snprintf(err, err_cap, "Channel dimension ratio %.1fx exceeds %.0fx limit.", ratio, 8.0);
                for (int c = 0; c < 3; c++)
                    if (ch[c] && k > 0 && (size_t)k < err_cap)
                        k += snprintf(err + k, err_cap - k, " %c=%zux%zu", names[c], cls[c], rws[c]);
                if (k > 0 && (size_t)k < err_cap) snprintf(err + k, err_cap - k, ". Check channel assignments.");
            }
            return 2;
        }
        res->resampled = 1;
        for (int c = 0; c < 3; c++) if (ch[c]) {
            harm[c] = (float *)malloc(max_rows * max_cols * sizeof(float));
            orc_resample_image(ch[c], rws[c], cls[c], max_rows, max_cols, harm[c]);
        }
    }
    size_t rows = max_rows, cols = max_cols, n = rows * cols;
    res->rows = rows;
    res->cols = cols;
    const float *eff[3];
    for (int c = 0; c < 3; c++) eff[c] = harm[c] ? harm[c] : ch[c];

    float *img[3];
    img[0] = channel_or_synth(eff[0], eff[1], eff[2], n);
    img[1] = channel_or_synth(eff[1], eff[0], eff[2], n);
    img[2] = channel_or_synth(eff[2], eff[0], eff[1], n);
    if (cfg->align) {                                                        /* align_channels :165-189 */
        const float *ref_ch = eff[0] ? eff[0] : (eff[1] ? eff[1] : eff[2]);
        for (int c = 1; c < 3; c++) if (ch[c]) {
            double *off = c == 1 ? res->offset_g : res->offset_b;
            float *al = align_pair(ref_ch, img[c], rows, cols, cfg->align_method, cfg->num_threads, off);
            free(img[c]);
            img[c] = al;
        }
    }
    for (int c = 0; c < 3; c++) free(harm[c]);

    orc_image_stats full[3];
    for (int c = 0; c < 3; c++) {
        orc_compute_image_stats(img[c], n, &full[c]);
        res->chan_stats[c][0] = full[c].min;
        res->chan_stats[c][1] = full[c].max;
        res->chan_stats[c][2] = full[c].median;
        res->chan_stats[c][3] = full[c].mean;
    }
    double wb[3] = {1.0, 1.0, 1.0};
    if (cfg->white_balance == 0) orc_select_wb_reference(&full[0], &full[1], &full[2], wb);
    else if (cfg->white_balance == 1) memcpy(wb, cfg->wb_manual, sizeof wb);
    for (int c = 0; c < 3; c++) {                                            /* apply_multiplier_inplace :127-130 */
        float mult = (float)wb[c];
        if (fabsf(mult - 1.0f) < 1e-7f) continue;
        for (size_t i = 0; i < n; i++) img[c][i] = img[c][i] * mult;
    }
    orc_image_stats wbst[3];
    if (cfg->auto_stretch && cfg->linked_stf) {
        float *comb = (float *)malloc((n ? n : 1) * sizeof(float));
        for (size_t i = 0; i < n; i++) comb[i] = (img[0][i] + img[1][i] + img[2][i]) * (1.0f / 3.0f);   /* :153-163 */
        orc_image_stats st;
        orc_compute_image_stats(comb, n, &st);
        free(comb);
        orc_stf_params p;
        orc_auto_stf(&st, 0.25, -2.8, &p);
        for (int c = 0; c < 3; c++) { res->stf[c] = p; orc_compute_image_stats(img[c], n, &wbst[c]); }
    } else if (cfg->auto_stretch) {
        for (int c = 0; c < 3; c++) { orc_compute_image_stats(img[c], n, &wbst[c]); orc_auto_stf(&wbst[c], 0.25, -2.8, &res->stf[c]); }
    } else {
        for (int c = 0; c < 3; c++) {
            orc_compute_image_stats(img[c], n, &wbst[c]);
            if (cfg->has_stf[c]) res->stf[c] = cfg->stf[c];
            else { res->stf[c].shadow = 0.0; res->stf[c].midtone = 0.5; res->stf[c].highlight = 1.0; }
        }
    }
    float *pre[3] = {pre_r, pre_g, pre_b}, *out[3] = {out_r, out_g, out_b};
    for (int c = 0; c < 3; c++) {
        res->stats_wb[c] = wbst[c];
        if (pre[c]) memcpy(pre[c], img[c], n * sizeof(float));
        orc_compose_apply_stf_inplace(img[c], n, &res->stf[c], &wbst[c]);
    }
    if (cfg->has_scnr) {                                                     /* :301-306 (dims always equal here) */
        orc_apply_scnr_inplace(img[0], img[1], img[2], n, cfg->scnr_method, cfg->scnr_amount, cfg->scnr_preserve);
        res->scnr_applied = 1;
    }
    for (int c = 0; c < 3; c++) { memcpy(out[c], img[c], n * sizeof(float)); free(img[c]); }
    return 0;
}
