/* ORACLE (test infrastructure).  Restates src-tauri/src/core/imaging/stats.rs.
 * The reference has no unit tests for this file: PARITY UNPINNED -- the
 * restatement (reviewable line by line against the citations) is the pin.
 *
 * Reduction order: the reference folds 65 536-pixel chunks with rayon and
 * combines the per-chunk f64 sums in an unspecified tree order
 * (stats.rs:233-258,272-297).  The oracle pins it: sequential inside a chunk,
 * chunks combined left to right.  Integer results (counts, histograms) do
 * not depend on the order. */
#include "ab_oracle.h"
#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

#define PADDING_THRESHOLD 1e-7f /* types/constants.rs:6 */
#define MAD_TO_SIGMA 1.4826     /* types/constants.rs:7 */
#define CHUNK_SIZE 65536u       /* stats.rs:7 */
#define HIST_BINS 65536u        /* stats.rs:8 */

static inline int is_valid_pixel(float v) { return isfinite(v) && v > PADDING_THRESHOLD; } /* :10-13 */

/* Rust `f64 as usize`: saturating, NaN -> 0 */
static inline size_t f64_to_usize_sat(double v) {
    if (!(v > 0.0)) return 0;
    if (v >= 18446744073709551615.0) return SIZE_MAX;
    return (size_t)v;
}

static void stats_default(orc_image_stats *o) { memset(o, 0, sizeof(*o)); }

/* stats.rs:233-258 */
static void scan_stats(const float *s, size_t n, double *mn_o, double *mx_o, double *sum_o, size_t *cnt_o) {
    double mn = DBL_MAX, mx = -DBL_MAX, sum = 0.0;
    size_t cnt = 0;
    for (size_t c0 = 0; c0 < n; c0 += CHUNK_SIZE) {
        size_t c1 = c0 + CHUNK_SIZE < n ? c0 + CHUNK_SIZE : n;
        double cmn = DBL_MAX, cmx = -DBL_MAX, cs = 0.0;
        size_t cc = 0;
        for (size_t i = c0; i < c1; i++) {
            float v = s[i];
            if (is_valid_pixel(v)) {
                double vf = (double)v;
                if (vf < cmn) cmn = vf;
                if (vf > cmx) cmx = vf;
                cs += vf;
                cc++;
            }
        }
        mn = fmin(mn, cmn);
        mx = fmax(mx, cmx);
        sum = sum + cs;
        cnt += cc;
    }
    *mn_o = mn; *mx_o = mx; *sum_o = sum; *cnt_o = cnt;
}

/* stats.rs:43-73 */
void orc_compute_image_stats_exact(const float *slice, size_t n, orc_image_stats *out) {
    double gmin, gmax, gsum;
    size_t total_valid;
    scan_stats(slice, n, &gmin, &gmax, &gsum, &total_valid);
    if (total_valid == 0) { stats_default(out); return; }
    float *valid = (float *)malloc(total_valid * sizeof(float));
    size_t w = 0;
    for (size_t i = 0; i < n; i++) if (is_valid_pixel(slice[i])) valid[w++] = slice[i];
    uint64_t nn = (uint64_t)w;
    double mean = gsum / (double)nn;
    double median = orc_exact_median_mut(valid, w);
    float mad_f32 = orc_exact_mad_mut(valid, w, (float)median);
    double mad = (double)mad_f32;
    double sigma = fmax(mad * MAD_TO_SIGMA, 1e-30);
    out->min = gmin; out->max = gmax; out->mean = mean; out->sigma = sigma;
    out->median = median; out->mad = mad; out->valid_count = nn;
    free(valid);
}

/* stats.rs:212-231 */
static void scan_minmax(const float *s, size_t n, double *mn_o, double *mx_o) {
    double mn = DBL_MAX, mx = -DBL_MAX;
    for (size_t i = 0; i < n; i++) {
        float v = s[i];
        if (is_valid_pixel(v)) {
            double vf = (double)v;
            if (vf < mn) mn = vf;
            if (vf > mx) mx = vf;
        }
    }
    *mn_o = mn; *mx_o = mx;
}

/* stats.rs:260-300 */
void orc_stats_value_hist(const float *s, size_t n, double data_min, double data_max, uint64_t *hist,
                          double *out_sum, uint64_t *out_cnt) {
    double range = fmax(data_max - data_min, 1e-30);
    double inv_bin = (double)HIST_BINS / range;
    size_t last_bin = HIST_BINS - 1;
    memset(hist, 0, HIST_BINS * sizeof(uint64_t));
    double sum = 0.0;
    uint64_t cnt = 0;
    for (size_t c0 = 0; c0 < n; c0 += CHUNK_SIZE) {
        size_t c1 = c0 + CHUNK_SIZE < n ? c0 + CHUNK_SIZE : n;
        double cs = 0.0;
        for (size_t i = c0; i < c1; i++) {
            float v = s[i];
            if (is_valid_pixel(v)) {
                double vf = (double)v;
                cs += vf;
                cnt++;
                size_t idx = f64_to_usize_sat((vf - data_min) * inv_bin);
                hist[idx < last_bin ? idx : last_bin] += 1;
            }
        }
        sum = sum + cs;
    }
    *out_sum = sum;
    *out_cnt = cnt;
}

/* stats.rs:302-312 */
static size_t find_percentile_bin(const uint64_t *hist, size_t nb, size_t total, double pct) {
    uint64_t target = (uint64_t)f64_to_usize_sat(ceil((double)total * pct));
    uint64_t cum = 0;
    for (size_t i = 0; i < nb; i++) {
        cum += hist[i];
        if (cum >= target) return i;
    }
    return nb - 1;
}

/* stats.rs:314-332 */
static double interpolate_percentile(const uint64_t *hist, size_t nb, size_t total, double pct,
                                     double data_min, double bin_width) {
    uint64_t target = (uint64_t)f64_to_usize_sat(ceil((double)total * pct));
    uint64_t cum = 0;
    for (size_t i = 0; i < nb; i++) {
        uint64_t count = hist[i];
        cum += count;
        if (cum >= target) {
            uint64_t overshoot = cum - target;
            double frac = count > 0 ? 1.0 - ((double)overshoot / (double)count) : 0.5;
            return data_min + ((double)i + frac) * bin_width;
        }
    }
    return data_min + (double)nb * bin_width;
}

/* stats.rs:334-353 */
static double resolve_rank_in_hist(const uint64_t *hist, size_t nb, uint64_t rank, double region_lo,
                                   double sub_bin_width) {
    if (rank == 0) return region_lo;
    uint64_t cum = 0;
    for (size_t i = 0; i < nb; i++) {
        uint64_t count = hist[i];
        cum += count;
        if (cum >= rank) {
            uint64_t overshoot = cum - rank;
            double frac = count > 0 ? 1.0 - ((double)overshoot / (double)count) : 0.5;
            return region_lo + ((double)i + frac) * sub_bin_width;
        }
    }
    return region_lo + (double)nb * sub_bin_width;
}

/* stats.rs:85-210 */
static void compute_stats_hist_core(const float *slice, size_t n, double global_min, double global_max,
                                    orc_image_stats *out) {
    double range = fmax(global_max - global_min, 1e-30);
    double bin_width = range / (double)HIST_BINS;
    size_t last_bin = HIST_BINS - 1;

    uint64_t *value_hist = (uint64_t *)malloc(HIST_BINS * sizeof(uint64_t));
    uint64_t *refine = (uint64_t *)calloc(HIST_BINS, sizeof(uint64_t));
    uint64_t *dev = (uint64_t *)calloc(HIST_BINS, sizeof(uint64_t));
    uint64_t *mad_refine = (uint64_t *)calloc(HIST_BINS, sizeof(uint64_t));
    double global_sum;
    uint64_t total_valid_u;
    orc_stats_value_hist(slice, n, global_min, global_max, value_hist, &global_sum, &total_valid_u);
    size_t total_valid = (size_t)total_valid_u;
    if (total_valid == 0) { stats_default(out); goto done; }

    {
        uint64_t nn = (uint64_t)total_valid;
        double mean = global_sum / (double)nn;
        uint64_t half_count = (uint64_t)f64_to_usize_sat(ceil((double)total_valid * 0.5));  /* :100 */

        size_t median_bin = find_percentile_bin(value_hist, HIST_BINS, total_valid, 0.5);
        uint64_t count_before_median = 0;
        for (size_t i = 0; i < median_bin; i++) count_before_median += value_hist[i];
        double median_bin_lo = global_min + (double)median_bin * bin_width;
        double median_bin_hi = median_bin_lo + bin_width;

        double coarse_median =
            interpolate_percentile(value_hist, HIST_BINS, total_valid, 0.5, global_min, bin_width);

        double dev_range = range;                                              /* :111-114 */
        double dev_bw = dev_range / (double)HIST_BINS;
        double dev_inv = (double)HIST_BINS / dev_range;
        float coarse_med_f32 = (float)coarse_median;

        double refine_range = fmax(median_bin_hi - median_bin_lo, 1e-30);      /* :116-117 */
        double refine_inv = (double)HIST_BINS / refine_range;

        for (size_t i = 0; i < n; i++) {                                        /* :119-146 */
            float v = slice[i];
            if (is_valid_pixel(v)) {
                double vf = (double)v;
                if (vf >= median_bin_lo && vf < median_bin_hi) {
                    size_t idx = f64_to_usize_sat((vf - median_bin_lo) * refine_inv);
                    refine[idx < last_bin ? idx : last_bin] += 1;
                }
                float d = fabsf(v - coarse_med_f32);
                size_t didx = f64_to_usize_sat((double)d * dev_inv);
                dev[didx < last_bin ? didx : last_bin] += 1;
            }
        }

        uint64_t median_rank_in_bin = half_count > count_before_median ? half_count - count_before_median : 0;
        double median_refine_bw = refine_range / (double)HIST_BINS;
        double median = resolve_rank_in_hist(refine, HIST_BINS, median_rank_in_bin, median_bin_lo,
                                             median_refine_bw);

        size_t mad_bin = find_percentile_bin(dev, HIST_BINS, total_valid, 0.5);   /* :154-158 */
        size_t expand_lo = mad_bin > 0 ? mad_bin - 1 : 0;
        size_t expand_hi = mad_bin + 2 < HIST_BINS ? mad_bin + 2 : HIST_BINS;
        double mad_region_lo = (double)expand_lo * dev_bw;
        double mad_region_hi = (double)expand_hi * dev_bw;

        float exact_med_f32 = (float)median;                                      /* :160-164 */
        double mad_refine_range = fmax(mad_region_hi - mad_region_lo, 1e-30);
        double mad_refine_inv = (double)HIST_BINS / mad_refine_range;
        float mad_lo_f32 = (float)mad_region_lo;
        float mad_hi_f32 = (float)mad_region_hi;

        uint64_t count_below = 0;
        for (size_t i = 0; i < n; i++) {                                          /* :166-191 */
            float v = slice[i];
            if (is_valid_pixel(v)) {
                float dv = fabsf(v - exact_med_f32);
                if (dv < mad_lo_f32) {
                    count_below++;
                } else if (dv < mad_hi_f32) {
                    size_t idx = f64_to_usize_sat(((double)dv - mad_region_lo) * mad_refine_inv);
                    mad_refine[idx < last_bin ? idx : last_bin] += 1;
                }
            }
        }

        uint64_t mad_rank_in_region = half_count > count_below ? half_count - count_below : 0;
        double mad_refine_bw = mad_refine_range / (double)HIST_BINS;
        double mad = resolve_rank_in_hist(mad_refine, HIST_BINS, mad_rank_in_region, mad_region_lo,
                                          mad_refine_bw);
        double sigma = fmax(mad * MAD_TO_SIGMA, 1e-30);

        out->min = global_min; out->max = global_max; out->mean = mean; out->sigma = sigma;
        out->median = median; out->mad = mad; out->valid_count = nn;
    }
done:
    free(value_hist); free(refine); free(dev); free(mad_refine);
}

/* stats.rs:75-83 */
void orc_compute_image_stats_hist(const float *slice, size_t n, orc_image_stats *out) {
    double gmin, gmax;
    scan_minmax(slice, n, &gmin, &gmax);
    if (gmin == DBL_MAX) { stats_default(out); return; }
    compute_stats_hist_core(slice, n, gmin, gmax, out);
}

/* stats.rs:15-23 */
void orc_compute_image_stats(const float *slice, size_t n, orc_image_stats *out) {
    if (n > 4000000u) { orc_compute_image_stats_hist(slice, n, out); return; }
    orc_compute_image_stats_exact(slice, n, out);
}

/* stats.rs:25-41 */
void orc_compute_image_stats_with_known_range(const float *slice, size_t n, double known_min,
                                              double known_max, orc_image_stats *out) {
    if (n <= 4000000u) { orc_compute_image_stats_exact(slice, n, out); return; }
    if (!isfinite(known_min) || !isfinite(known_max) || known_min >= known_max) {
        orc_compute_image_stats_hist(slice, n, out);
        return;
    }
    compute_stats_hist_core(slice, n, known_min, known_max, out);
}

/* stats.rs:378-421 */
int orc_build_histogram(const float *slice, size_t n, size_t bins, double dmin, double dmax,
                        uint32_t *out_bins) {
    memset(out_bins, 0, bins * sizeof(uint32_t));
    double range = dmax - dmin;
    if (range < 1e-10) return 1;
    double inv_bin_width = (double)bins / range;
    size_t last = bins - 1;
    for (size_t i = 0; i < n; i++) {
        float v = slice[i];
        if (is_valid_pixel(v)) {
            size_t idx = f64_to_usize_sat(((double)v - dmin) * inv_bin_width);
            out_bins[idx < last ? idx : last] += 1;
        }
    }
    return 0;
}
