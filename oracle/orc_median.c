/* ORACLE (test infrastructure).  Restates src-tauri/src/math/median.rs and
 * src-tauri/src/math/sigma_clip.rs.  See ab_oracle.h for the rules. */
#include "ab_oracle.h"
#include <math.h>
#include <float.h>
#include <stdlib.h>
#include <string.h>

/* median.rs:4-13 -- partial_cmp, NaN sorts last, NaN == NaN */
int orc_f32_cmp(float a, float b) {
    if (a < b) return -1;
    if (a > b) return 1;
    if (a == b) return 0;
    int an = isnan(a), bn = isnan(b);
    if (an && bn) return 0;
    if (an) return 1;
    return -1;
}

static inline int lt(float a, float b) { return orc_f32_cmp(a, b) < 0; }

/* select_nth_unstable_by(k, f32_cmp): a[k] becomes the k-th order statistic
 * under the total order above; smaller-or-equal before, greater-or-equal
 * after.  Median-of-three quickselect (Hoare partition), insertion sort for
 * short ranges.  The value at k is algorithm-independent; the permutation of
 * the rest is not (nor is it in Rust). */
void orc_select_nth_f32(float *a, size_t n, size_t k) {
    if (n == 0 || k >= n) return;
    size_t lo = 0, hi = n - 1;
    while (hi > lo) {
        if (hi - lo < 12) {
            for (size_t i = lo + 1; i <= hi; i++) {
                float v = a[i];
                size_t j = i;
                while (j > lo && lt(v, a[j - 1])) { a[j] = a[j - 1]; j--; }
                a[j] = v;
            }
            return;
        }
        size_t mid = lo + (hi - lo) / 2;
        if (lt(a[mid], a[lo])) { float t = a[mid]; a[mid] = a[lo]; a[lo] = t; }
        if (lt(a[hi], a[lo]))  { float t = a[hi];  a[hi] = a[lo];  a[lo] = t; }
        if (lt(a[hi], a[mid])) { float t = a[hi];  a[hi] = a[mid]; a[mid] = t; }
        float pivot = a[mid];
        size_t i = lo, j = hi;
        for (;;) {
            while (lt(a[i], pivot)) i++;
            while (lt(pivot, a[j])) j--;
            if (i >= j) break;
            float t = a[i]; a[i] = a[j]; a[j] = t;
            i++; j--;
        }
        /* now a[lo..j] <= pivot <= a[j+1..hi] */
        if (k <= j) hi = j; else lo = j + 1;
    }
}

/* median.rs:27-44 */
double orc_exact_median_mut(float *data, size_t n) {
    if (n == 0) return 0.0;
    size_t mid = n / 2;
    orc_select_nth_f32(data, n, mid);
    if (n % 2 == 0) {
        double right = (double)data[mid];
        float left = -FLT_MAX;                      /* f32::MIN */
        for (size_t i = 0; i < mid; i++) if (data[i] > left) left = data[i];
        return ((double)left + right) / 2.0;
    }
    return (double)data[mid];
}

/* median.rs:46-63 */
float orc_median_f32_mut(float *data, size_t n) {
    if (n == 0) return 0.0f;
    size_t mid = n / 2;
    orc_select_nth_f32(data, n, mid);
    if (n % 2 == 0) {
        float right = data[mid];
        float left = -FLT_MAX;
        for (size_t i = 0; i < mid; i++) if (data[i] > left) left = data[i];
        return (left + right) / 2.0f;
    }
    return data[mid];
}

/* median.rs:65-73 */
float orc_exact_mad_mut(float *data, size_t n, float median) {
    if (n == 0) return 0.0f;
    for (size_t i = 0; i < n; i++) data[i] = fabsf(data[i] - median);
    return orc_median_f32_mut(data, n);
}

/* sigma_clip.rs:4-34 */
void orc_sigma_clipped_stats(float *values, size_t *n_io, float kappa, size_t iterations,
                             double *out_median, double *out_sigma) {
    size_t n = *n_io;
    float *devs = (float *)malloc((n ? n : 1) * sizeof(float));
    const double MAD_TO_SIGMA = 1.4826;              /* types/constants.rs:7 */
    for (size_t it = 0; it < iterations; it++) {
        if (n < 3) break;
        double median = orc_exact_median_mut(values, n);
        for (size_t i = 0; i < n; i++) devs[i] = (float)fabs((double)values[i] - median);
        double mad = (double)orc_median_f32_mut(devs, n);
        double sig = fmax(mad * MAD_TO_SIGMA, 1e-30);
        float lo = (float)(median - (double)kappa * sig);
        float hi = (float)(median + (double)kappa * sig);
        size_t w = 0;
        for (size_t i = 0; i < n; i++) {
            float v = values[i];
            if (v >= lo && v <= hi) values[w++] = v;   /* Vec::retain keeps order */
        }
        n = w;
    }
    *n_io = n;
    if (n == 0) { *out_median = 0.0; *out_sigma = 1.0; free(devs); return; }
    double median = orc_exact_median_mut(values, n);
    for (size_t i = 0; i < n; i++) devs[i] = (float)fabs((double)values[i] - median);
    double sigma = fmax((double)orc_median_f32_mut(devs, n) * MAD_TO_SIGMA, 1e-30);
    *out_median = median;
    *out_sigma = sigma;
    free(devs);
}
