/* ORACLE (test infrastructure).  Restates src-tauri/src/core/imaging/sampling.rs,
 * core/imaging/boundary.rs:9-20, core/stacking/align.rs:36-57 and
 * core/alignment/affine.rs:55-80,663-690.  See ab_oracle.h for the rules. */
#include "ab_oracle.h"
#include <math.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* Rust `f64 as i64` saturates and maps NaN to 0 */
static inline int64_t f64_to_i64_sat(double v) {
    if (isnan(v)) return 0;
    if (v >= 9223372036854775807.0) return INT64_MAX;
    if (v <= -9223372036854775808.0) return INT64_MIN;
    return (int64_t)v;
}

/* sampling.rs:4-14 */
double orc_catmull_rom(double t) {
    double abs_t = fabs(t);
    if (abs_t <= 1.0) return abs_t * abs_t * (1.5 * abs_t - 2.5) + 1.0;
    if (abs_t <= 2.0) return abs_t * (abs_t * (2.5 - 0.5 * abs_t) - 4.0) + 2.0;
    return 0.0;
}

/* boundary.rs:9-20 */
size_t orc_clamp_index(int64_t idx, size_t len) {
    if (len == 0) return 0;
    if (idx < 0) return 0;
    if (idx >= (int64_t)len) return len - 1;
    return (size_t)idx;
}

/* sampling.rs:16-24 */
float orc_nearest_sample(const float *s, size_t rows, size_t cols, double y, double x) {
    if (rows == 0 || cols == 0 || s == NULL) return 0.0f;
    size_t iy = orc_clamp_index(f64_to_i64_sat(round(y)), rows);
    size_t ix = orc_clamp_index(f64_to_i64_sat(round(x)), cols);
    return s[iy * cols + ix];
}

/* sampling.rs:26-49 */
float orc_bilinear_sample(const float *s, size_t rows, size_t cols, double y, double x) {
    if (rows == 0 || cols == 0 || s == NULL) return 0.0f;
    int64_t ix0 = f64_to_i64_sat(floor(x));
    int64_t iy0 = f64_to_i64_sat(floor(y));
    double fx = x - (double)ix0;
    double fy = y - (double)iy0;
    size_t r0 = orc_clamp_index(iy0, rows), r1 = orc_clamp_index(iy0 + 1, rows);
    size_t c0 = orc_clamp_index(ix0, cols), c1 = orc_clamp_index(ix0 + 1, cols);
    double v00 = (double)s[r0 * cols + c0], v01 = (double)s[r0 * cols + c1];
    double v10 = (double)s[r1 * cols + c0], v11 = (double)s[r1 * cols + c1];
    double top = v00 + (v01 - v00) * fx;
    double bot = v10 + (v11 - v10) * fx;
    return (float)(top + (bot - top) * fy);
}

/* sampling.rs:51-80 */
float orc_bicubic_sample(const float *s, size_t rows, size_t cols, double y, double x) {
    if (rows == 0 || cols == 0 || s == NULL) return 0.0f;
    int64_t ix = f64_to_i64_sat(floor(x));
    int64_t iy = f64_to_i64_sat(floor(y));
    double fx = x - (double)ix;
    double fy = y - (double)iy;
    double wx[4] = {orc_catmull_rom(fx + 1.0), orc_catmull_rom(fx), orc_catmull_rom(fx - 1.0),
                    orc_catmull_rom(fx - 2.0)};
    double val = 0.0;
    for (int64_t j = 0; j < 4; j++) {
        size_t r = orc_clamp_index(iy + j - 1, rows);
        size_t row_off = r * cols;
        double row_val = 0.0;
        for (int64_t i = 0; i < 4; i++) {
            size_t c = orc_clamp_index(ix + i - 1, cols);
            row_val += (double)s[row_off + c] * wx[i];
        }
        val += row_val * orc_catmull_rom(fy - (double)(j - 1));
    }
    return (float)val;
}

/* align.rs:36-57 */
void orc_shift_image_subpixel(const float *src, size_t rows, size_t cols, double dy, double dx,
                              int threads, float *out) {
    if (fabs(dy) < 1e-12 && fabs(dx) < 1e-12) {                 /* :37-39 identity clone */
        memcpy(out, src, rows * cols * sizeof(float));
        return;
    }
#ifdef _OPENMP
    if (threads <= 0) threads = omp_get_max_threads();
#else
    threads = 1;
#endif
    double rows_f = (double)rows, cols_f = (double)cols;
#pragma omp parallel for num_threads(threads) schedule(static)
    for (int64_t y = 0; y < (int64_t)rows; y++) {
        float *row = out + (size_t)y * cols;
        for (size_t x = 0; x < cols; x++) {
            double sy = (double)y + dy;
            double sx = (double)x + dx;
            if (sy < -0.5 || sy > rows_f - 0.5 || sx < -0.5 || sx > cols_f - 0.5) {
                row[x] = 0.0f;                                   /* vec![0.0; ..], `continue` */
                continue;
            }
            row[x] = orc_bicubic_sample(src, rows, cols, sy, sx);
        }
    }
}

/* affine.rs:663-690; map() is affine.rs:74-80 */
void orc_warp_image(const float *src, size_t src_rows, size_t src_cols, const double t[6],
                    size_t out_rows, size_t out_cols, int threads, float *out) {
#ifdef _OPENMP
    if (threads <= 0) threads = omp_get_max_threads();
#else
    threads = 1;
#endif
    const double a = t[0], b = t[1], tx = t[2], c = t[3], d = t[4], ty = t[5];
    /* (src_cols - 1) as f64 with usize arithmetic: src dims >= 1 assumed (0 would
     * underflow-panic in the reference's debug build) */
    double lim_x = (double)(src_cols - 1), lim_y = (double)(src_rows - 1);
#pragma omp parallel for num_threads(threads) schedule(static)
    for (int64_t y = 0; y < (int64_t)out_rows; y++) {
        float *row = out + (size_t)y * out_cols;
        for (size_t x = 0; x < out_cols; x++) {
            double xf = (double)x, yf = (double)y;
            double sx = a * xf + b * yf + tx;
            double sy = c * xf + d * yf + ty;
            if (sx >= 0.0 && sy >= 0.0 && sx < lim_x && sy < lim_y)
                row[x] = orc_bicubic_sample(src, src_rows, src_cols, sy, sx);
            else
                row[x] = 0.0f;
        }
    }
}
