/* ORACLE (test infrastructure).  Restates src-tauri/src/core/stacking/combine.rs.
 * See ab_oracle.h for the rules. */
#include "ab_oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define MAD_TO_SIGMA 1.4826 /* types/constants.rs:7 */

int orc_max_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

static int cmp_asc(const void *pa, const void *pb) {
    float a = *(const float *)pa, b = *(const float *)pb;
    return orc_f32_cmp(a, b);
}

/* combine.rs:14-92.  `devs` must hold n floats of scratch (the reference
 * heap-allocates it per call, combine.rs:42-43). */
static float sigma_clip_combine_impl(float *values, float *devs, size_t n_orig, float sigma_low,
                                     float sigma_high, size_t max_iter, int order_mode,
                                     uint32_t *out_rejected, double *out_sum, uint32_t *out_len) {
    if (out_sum) *out_sum = 0.0;
    if (out_len) *out_len = 0;
    if (n_orig == 0) { *out_rejected = 0; return 0.0f; }                 /* :21-23 */
    if (n_orig == 1) {                                                    /* :24-26 */
        *out_rejected = 0;
        if (out_sum) *out_sum = (double)values[0];
        if (out_len) *out_len = 1;
        return values[0];
    }

    size_t len = n_orig;
    uint32_t rejected = 0;
    float last_center = NAN;

    for (size_t iteration = 0; iteration < max_iter; iteration++) {       /* :32 */
        if (len < 2) break;                                               /* :33-35 */
        float center, sigma;
        if (iteration == 0) {                                             /* :37-48 */
            size_t mid = len / 2;
            orc_select_nth_f32(values, len, mid);
            float med = values[mid];
            for (size_t i = 0; i < len; i++) devs[i] = fabsf(values[i] - med);
            size_t dmid = len / 2;
            orc_select_nth_f32(devs, len, dmid);
            float mad = devs[dmid];
            float sig = (float)fmax((double)mad * MAD_TO_SIGMA, 1e-10);
            center = med;
            sigma = sig;
            if (order_mode == ORC_ORDER_ASCENDING)
                qsort(values, len, sizeof(float), cmp_asc);   /* canonical order for later sums */
        } else {                                                          /* :49-61 */
            double n = (double)len;
            double s = 0.0;
            for (size_t i = 0; i < len; i++) s += (double)values[i];
            double mean = s / n;
            double q = 0.0;
            for (size_t i = 0; i < len; i++) {
                double d = (double)values[i] - mean;
                q += d * d;
            }
            double variance = q / fmax(n - 1.0, 1.0);
            center = (float)mean;
            sigma = (float)fmax(sqrt(variance), 1e-10);
        }
        last_center = center;                                             /* :63 */

        float lo = -sigma_low * sigma;                                    /* :65-66 */
        float hi = sigma_high * sigma;
        size_t write = 0;
        for (size_t read = 0; read < len; read++) {                       /* :68-74 */
            float dev = values[read] - center;
            if (dev >= lo && dev <= hi) values[write++] = values[read];
        }
        size_t removed = len - write;                                     /* :76-82 */
        rejected += (uint32_t)removed;
        len = write;
        if (removed == 0) break;
    }

    *out_rejected = rejected;
    if (len == 0) {                                                       /* :85-88 */
        return isfinite(last_center) ? last_center : 0.0f;
    }
    if (order_mode == ORC_ORDER_ASCENDING && max_iter == 0)
        qsort(values, len, sizeof(float), cmp_asc);
    double s = 0.0;                                                       /* :90-91 */
    for (size_t i = 0; i < len; i++) s += (double)values[i];
    if (out_sum) *out_sum = s;
    if (out_len) *out_len = (uint32_t)len;
    return (float)(s / (double)len);
}

float orc_sigma_clip_combine(float *values, size_t n, float sigma_low, float sigma_high,
                             size_t max_iter, int order_mode, uint32_t *out_rejected) {
    float *devs = (float *)malloc((n ? n : 1) * sizeof(float));
    uint32_t rej = 0;
    float r = sigma_clip_combine_impl(values, devs, n, sigma_low, sigma_high, max_iter, order_mode,
                                      &rej, NULL, NULL);
    if (out_rejected) *out_rejected = rej;
    free(devs);
    return r;
}

/* combine.rs:94-193 with align == false */
int orc_stack_images_noalign(const float *const *planes, const int64_t *rows, const int64_t *cols,
                             size_t n_images, float sigma_low, float sigma_high, size_t max_iter,
                             int order_mode, int threads, float *out, uint64_t *out_rejected,
                             int64_t *out_rows, int64_t *out_cols) {
    if (n_images == 0) return -1;                                         /* :98-100 */
    int64_t min_rows = rows[0], min_cols = cols[0];                       /* :104-105 */
    for (size_t i = 1; i < n_images; i++) {
        if (rows[i] < min_rows) min_rows = rows[i];
        if (cols[i] < min_cols) min_cols = cols[i];
    }
    if (out_rows) *out_rows = min_rows;
    if (out_cols) *out_cols = min_cols;
    uint64_t total_rejected = 0;
#ifdef _OPENMP
    if (threads <= 0) threads = omp_get_max_threads();
#else
    threads = 1;
#endif
    /* rows in parallel like par_chunks_mut(cols) (:160-182); crop = top-left (:107-113) */
#pragma omp parallel num_threads(threads) reduction(+ : total_rejected)
    {
        float *vals = (float *)malloc((n_images ? n_images : 1) * sizeof(float));
        float *devs = (float *)malloc((n_images ? n_images : 1) * sizeof(float));
#pragma omp for schedule(dynamic, 8)
        for (int64_t y = 0; y < min_rows; y++) {
            uint64_t local_rejected = 0;
            for (int64_t x = 0; x < min_cols; x++) {
                size_t cnt = 0;
                for (size_t s = 0; s < n_images; s++) {                   /* :170-175 */
                    float v = planes[s][y * cols[s] + x];
                    if (isfinite(v)) vals[cnt++] = v;
                }
                uint32_t rej = 0;
                out[y * min_cols + x] = sigma_clip_combine_impl(vals, devs, cnt, sigma_low, sigma_high,
                                                                max_iter, order_mode, &rej, NULL, NULL);
                local_rejected += rej;
            }
            total_rejected += local_rejected;
        }
        free(vals);
        free(devs);
    }
    if (out_rejected) *out_rejected = total_rejected;
    return 0;
}

/* SURVEY 8(e) frame-sharded mode: one shard's (sum of kept, count of kept). */
void orc_stack_partial_noalign(const float *const *planes, size_t n_images, int64_t npix,
                               float sigma_low, float sigma_high, size_t max_iter, int threads,
                               double *out_sum, uint32_t *out_cnt, uint64_t *out_rejected) {
    uint64_t total_rejected = 0;
#ifdef _OPENMP
    if (threads <= 0) threads = omp_get_max_threads();
#else
    threads = 1;
#endif
#pragma omp parallel num_threads(threads) reduction(+ : total_rejected)
    {
        float *vals = (float *)malloc((n_images ? n_images : 1) * sizeof(float));
        float *devs = (float *)malloc((n_images ? n_images : 1) * sizeof(float));
#pragma omp for schedule(static)
        for (int64_t p = 0; p < npix; p++) {
            size_t cnt = 0;
            for (size_t s = 0; s < n_images; s++) {
                float v = planes[s][p];
                if (isfinite(v)) vals[cnt++] = v;
            }
            uint32_t rej = 0, len = 0;
            double sum = 0.0;
            (void)sigma_clip_combine_impl(vals, devs, cnt, sigma_low, sigma_high, max_iter,
                                          ORC_ORDER_ASCENDING, &rej, &sum, &len);
            out_sum[p] = sum;
            out_cnt[p] = len;
            total_rejected += rej;
        }
        free(vals);
        free(devs);
    }
    if (out_rejected) *out_rejected = total_rejected;
}
