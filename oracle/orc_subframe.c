/* ORACLE (test infrastructure).  Restates core/analysis/subframe.rs: analyze_subframe (:51-121), compute_weight
 * (:123-146), normalize_weights (:148-159), median_of (:161-173).  See ab_oracle.h for the rules. */
#include "ab_oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

static int cmp_f64(const void *a, const void *b) {
    double x = *(const double *)a, y = *(const double *)b;
    return x < y ? -1 : (x > y ? 1 : 0);
}

static double median_of(const orc_star *stars, size_t n, int field) {           /* :161-173 */
    double *vals = (double *)malloc((n ? n : 1) * sizeof(double));
    size_t k = 0;
    for (size_t i = 0; i < n; i++) {
        double v = field == 0 ? stars[i].fwhm : (field == 1 ? stars[i].eccentricity : stars[i].snr);
        if (isfinite(v)) vals[k++] = v;
    }
    double r = 0.0;
    if (k) {
        qsort(vals, k, sizeof(double), cmp_f64);
        size_t mid = k / 2;
        r = (k % 2 == 0) ? (vals[mid - 1] + vals[mid]) / 2.0 : vals[mid];
    }
    free(vals);
    return r;
}

double orc_subframe_compute_weight(double fwhm, double ecc, double snr, double noise, const orc_subframe_config *c) {   /* :123-146 */
    double fwhm_score = fwhm > 0.5 ? 1.0 / fwhm : 0.0;
    double ecc_score = 1.0 - ecc;
    double snr_score = fmax(log(snr), 0.0);
    double noise_score = 1.0 / (1.0 + noise * 10.0);
    double total_weight = c->fwhm_weight + c->eccentricity_weight + c->snr_weight + c->noise_weight;
    if (total_weight < 1e-15) return 0.0;
    double raw = c->fwhm_weight * fwhm_score + c->eccentricity_weight * ecc_score + c->snr_weight * snr_score + c->noise_weight * noise_score;
    return fmax(raw / total_weight, 0.0);
}

/* the metrics of analyze_subframe from a detect_stars() result (:62-120) */
void orc_subframe_from_detection(const orc_star *stars, size_t n, double bg_median, double bg_sigma, const orc_subframe_config *c,
                                 orc_subframe_metrics *out) {
    memset(out, 0, sizeof *out);
    out->star_count = n;
    out->background_median = bg_median;
    out->background_sigma = bg_sigma;
    size_t need = c->min_stars < 5 ? c->min_stars : 5;                             /* MIN_STARS_FOR_METRICS.min(config.min_stars) */
    if (n < need) return;
    out->median_fwhm = median_of(stars, n, 0);
    out->median_eccentricity = median_of(stars, n, 1);
    out->median_snr = median_of(stars, n, 2);
    out->noise_ratio = bg_median > 1e-15 ? bg_sigma / bg_median : 0.0;
    out->weight = orc_subframe_compute_weight(out->median_fwhm, out->median_eccentricity, out->median_snr, out->noise_ratio, c);
    out->accepted = n >= c->min_stars && out->median_fwhm <= c->max_fwhm && out->median_eccentricity <= c->max_eccentricity &&
                    out->median_snr >= c->min_snr;
}

void orc_analyze_subframe(const float *image, size_t rows, size_t cols, const orc_subframe_config *c, orc_subframe_metrics *out) {   /* :51-121 */
    size_t cap = 1u << 16, total = 0;
    double bm, bs;
    orc_star *st = (orc_star *)malloc(cap * sizeof(orc_star));
    size_t n = orc_detect_stars(image, rows, cols, 4.0, st, cap, &total, &bm, &bs);   /* DETECTION_SIGMA, :6 */
    if (total > cap) {
        cap = total;
        st = (orc_star *)realloc(st, cap * sizeof(orc_star));
        n = orc_detect_stars(image, rows, cols, 4.0, st, cap, &total, &bm, &bs);
    }
    orc_subframe_from_detection(st, n, bm, bs, c, out);
    free(st);
}

void orc_subframe_normalize_weights(orc_subframe_metrics *m, size_t n) {          /* :148-159 */
    double max_w = 0.0;
    for (size_t i = 0; i < n; i++) max_w = fmax(max_w, m[i].weight);
    if (max_w > 1e-15)
        for (size_t i = 0; i < n; i++) m[i].weight /= max_w;
}
