/* ORACLE (test infrastructure).  Restates core/imaging/background.rs: extract_background (:55-116),
 * auto_sample_grid (:118-210), min_samples_for_degree (:212-215), poly_basis_into (:217-228),
 * eval_poly_inline (:230-249), fit_polynomial_surface (:251-290), evaluate_polynomial_surface
 * (:307-341), apply_correction (:343-383), compute_rms_residual (:385-415), solve_linear_system
 * (:417-459).  See ab_oracle.h for the rules. */
#include "ab_oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

#define MAX_POLY_TERMS 21
#define MAD_TO_SIGMA_F32 ((float)1.4826)   /* `MAD_TO_SIGMA as f32` */

/* f64::powi(i32) = compiler-rt __powidf2: square-and-multiply, LSB first */
static double powi(double a, int b) {
    double r = 1.0;
    for (;;) {
        if (b & 1) r *= a;
        b /= 2;
        if (b == 0) break;
        a *= a;
    }
    return r;
}

static size_t poly_basis_into(double y, double x, size_t degree, double *out) {   /* :217-228 */
    size_t idx = 0;
    for (size_t total = 0; total <= degree; total++)
        for (size_t yp = total + 1; yp-- > 0;) {
            size_t xp = total - yp;
            out[idx++] = powi(y, (int)yp) * powi(x, (int)xp);
        }
    return idx;
}

static double eval_poly(size_t degree, const double *coeffs, const double *y_pows, const double *x_pows) {   /* :230-249 */
    double val = 0.0;
    size_t idx = 0;
    for (size_t total = 0; total <= degree; total++)
        for (size_t yp = total + 1; yp-- > 0;) {
            size_t xp = total - yp;
            val += coeffs[idx] * y_pows[yp] * x_pows[xp];
            idx++;
        }
    return val;
}

static void fill_pows(double n, size_t degree, double *pows) {
    memset(pows, 0, 7 * sizeof(double));
    pows[0] = 1.0;
    size_t lim = degree < 6 ? degree : 6;
    for (size_t i = 1; i <= lim; i++) pows[i] = pows[i - 1] * n;
}

/* :417-459; returns 0 on "Singular matrix in polynomial fit" */
static int solve_linear_system(double *a, double *b, size_t n) {
    for (size_t col = 0; col < n; col++) {
        size_t max_row = col;
        double max_val = fabs(a[col * n + col]);
        for (size_t row = col + 1; row < n; row++) {
            double v = fabs(a[row * n + col]);
            if (v > max_val) { max_val = v; max_row = row; }
        }
        if (max_val < 1e-14) return 0;
        if (max_row != col) {
            for (size_t k = 0; k < n; k++) { double t = a[col * n + k]; a[col * n + k] = a[max_row * n + k]; a[max_row * n + k] = t; }
            double t = b[col]; b[col] = b[max_row]; b[max_row] = t;
        }
        double pivot = a[col * n + col];
        for (size_t row = col + 1; row < n; row++) {
            double factor = a[row * n + col] / pivot;
            for (size_t k = col; k < n; k++) a[row * n + k] -= factor * a[col * n + k];
            b[row] -= factor * b[col];
        }
    }
    for (size_t col = n; col-- > 0;) {
        double sum = b[col];
        for (size_t k = col + 1; k < n; k++) sum -= a[col * n + k] * b[k];
        b[col] = sum / a[col * n + col];
    }
    return 1;
}

typedef struct { float y, x, value; } sample_t;

/* returns: 0 ok; 1 "Image too small for grid_size"; 2 "Not enough background samples"; 3 singular fit.
 * mode 0 subtract, 1 divide.  model / corrected: rows*cols (may be NULL to skip).  coeffs_out: 21 doubles. */
int orc_extract_background(const float *image, size_t rows, size_t cols, size_t grid, size_t degree, float sigma_clip,
                           size_t iterations, int mode, float *model, float *corrected, size_t *sample_count_out,
                           double *rms_out, double *coeffs_out) {
    size_t cell_h = rows / grid, cell_w = cols / grid;
    if (cell_h < 4 || cell_w < 4) return 1;                                  /* :127-129 */
    size_t margin_h = cell_h / 4, margin_w = cell_w / 4;
    size_t inner_h = cell_h - 2 * margin_h, inner_w = cell_w - 2 * margin_w;
    size_t npix = rows * cols;

    float *all = (float *)malloc((npix ? npix : 1) * sizeof(float));
    size_t na = 0;
    for (size_t i = 0; i < npix; i++) if (isfinite(image[i]) && image[i] > 0.0f) all[na++] = image[i];   /* :135-141 */
    float global_median = orc_median_f32_mut(all, na);
    for (size_t i = 0; i < na; i++) all[i] = fabsf(all[i] - global_median);  /* :144 (devs of the permuted copy: same multiset) */
    float global_mad = orc_median_f32_mut(all, na);
    float sigma = global_mad * MAD_TO_SIGMA_F32;
    free(all);

    sample_t *samples = (sample_t *)malloc(grid * grid * sizeof(sample_t));
    size_t ns = 0;
    float *cell = (float *)malloc((inner_h * inner_w ? inner_h * inner_w : 1) * sizeof(float));
    for (size_t gy = 0; gy < grid; gy++)
        for (size_t gx = 0; gx < grid; gx++) {
            size_t y0 = gy * cell_h + margin_h, x0 = gx * cell_w + margin_w, nc = 0, zero_count = 0;
            size_t total_cell = inner_h * inner_w;
            for (size_t y = y0; y < y0 + inner_h; y++)
                for (size_t x = x0; x < x0 + inner_w; x++)
                    if (y < rows && x < cols) {
                        float v = image[y * cols + x];
                        if (isfinite(v) && v > 1e-7f) cell[nc++] = v; else zero_count++;
                    }
            if (nc == 0 || (double)zero_count / (double)total_cell > 0.3) continue;
            float cell_median = orc_median_f32_mut(cell, nc);
            float lo = global_median - sigma_clip * sigma, hi = global_median + sigma_clip * sigma;
            if (cell_median >= lo && cell_median <= hi) {
                samples[ns].y = (float)(y0 + inner_h / 2);
                samples[ns].x = (float)(x0 + inner_w / 2);
                samples[ns].value = cell_median;
                ns++;
            }
        }
    free(cell);
    size_t n_terms = (degree + 1) * (degree + 2) / 2, min_samples = n_terms + 2;
    float *vals = (float *)malloc((ns ? ns : 1) * sizeof(float));
    for (size_t it = 1; it < iterations; it++) {                             /* :192-207 */
        if (ns < min_samples) break;
        for (size_t i = 0; i < ns; i++) vals[i] = samples[i].value;
        float med = orc_median_f32_mut(vals, ns);
        for (size_t i = 0; i < ns; i++) vals[i] = fabsf(vals[i] - med);
        float mad = orc_median_f32_mut(vals, ns);
        float sig = mad * MAD_TO_SIGMA_F32;
        float lo = med - sigma_clip * sig, hi = med + sigma_clip * sig;
        size_t w = 0;
        for (size_t i = 0; i < ns; i++) if (samples[i].value >= lo && samples[i].value <= hi) samples[w++] = samples[i];
        ns = w;
    }
    free(vals);
    if (sample_count_out) *sample_count_out = ns;
    if (ns < min_samples) { free(samples); return 2; }                      /* :71-77 */

    /* fit_polynomial_surface :251-290 */
    double row_scale = (double)rows, col_scale = (double)cols;
    double *ata = (double *)calloc(n_terms * n_terms, sizeof(double)), *atb = (double *)calloc(n_terms, sizeof(double));
    double basis[MAX_POLY_TERMS];
    for (size_t s = 0; s < ns; s++) {
        double ny = (double)samples[s].y / row_scale - 0.5, nx = (double)samples[s].x / col_scale - 0.5;
        double val = (double)samples[s].value;
        size_t cnt = poly_basis_into(ny, nx, degree, basis);
        for (size_t i = 0; i < cnt; i++) {
            atb[i] += basis[i] * val;
            for (size_t j = 0; j < cnt; j++) ata[i * n_terms + j] += basis[i] * basis[j];
        }
    }
    for (size_t i = 0; i < n_terms; i++) ata[i * n_terms + i] += 1e-8;
    if (!solve_linear_system(ata, atb, n_terms)) { free(ata); free(atb); free(samples); return 3; }
    if (coeffs_out) memcpy(coeffs_out, atb, n_terms * sizeof(double));

    /* evaluate_polynomial_surface :307-341 */
    float *mdl = model ? model : (float *)malloc(npix * sizeof(float));
    for (size_t y = 0; y < rows; y++) {
        double ny = (double)y / row_scale - 0.5, yp[7], xp[7];
        fill_pows(ny, degree, yp);
        for (size_t x = 0; x < cols; x++) {
            double nx = (double)x / col_scale - 0.5;
            fill_pows(nx, degree, xp);
            mdl[y * cols + x] = (float)eval_poly(degree, atb, yp, xp);
        }
    }
    /* apply_correction :343-383 */
    if (corrected) {
        float *fv = (float *)malloc((npix ? npix : 1) * sizeof(float));
        size_t nf = 0;
        for (size_t i = 0; i < npix; i++) if (isfinite(mdl[i]) && mdl[i] > 0.0f) fv[nf++] = mdl[i];
        float model_median = nf == 0 ? 0.0f : orc_median_f32_mut(fv, nf);
        free(fv);
        for (size_t i = 0; i < npix; i++) {
            float img = image[i], bg = mdl[i];
            if (mode == 0) corrected[i] = img - bg + model_median;
            else corrected[i] = fabsf(bg) > 1e-10f ? (img / bg) * model_median : img;
        }
    }
    /* compute_rms_residual :385-415 */
    double sum_sq = 0.0;
    for (size_t s = 0; s < ns; s++) {
        double ny = (double)samples[s].y / row_scale - 0.5, nx = (double)samples[s].x / col_scale - 0.5, yp[7], xp[7];
        fill_pows(ny, degree, yp);
        fill_pows(nx, degree, xp);
        double diff = (double)samples[s].value - eval_poly(degree, atb, yp, xp);
        sum_sq += diff * diff;
    }
    if (rms_out) *rms_out = sqrt(sum_sq / (double)ns);
    if (!model) free(mdl);
    free(ata); free(atb); free(samples);
    return 0;
}
