/* ORACLE (test infrastructure).  Restates core/alignment/phase_correlation.rs,
 * core/alignment/downsample.rs:6-46 and the pieces of math/ it uses: window.rs:3-18 (periodic
 * Hann), fft.rs:202-226 (prepare_windowed_buffer), :136-167 (forward_2d / inverse_2d),
 * :271-282 (find_peak), complex.rs:6-44 (cross power spectrum), subpixel.rs:27-100,
 * normalization.rs:128-170 (mean / sigma / SNR).  See ab_oracle.h for the rules.
 *
 * THIRD-PARTY ARITHMETIC: the reference's FFT is rustfft 6.4.1 (Cargo.lock:4254), which is not in
 * /root/reference and cannot be built here.  Its butterfly order (mixed radix, SIMD-dependent) is
 * not reproducible, so the bit pattern of the correlation surface is NOT pinned; the reference's
 * own tests pin the estimated shift to +-0.5 .. 1.5 px (phase_correlation.rs:197-240,
 * align.rs:216-223).  This restatement uses a textbook iterative radix-2 decimation-in-time FFT
 * (power-of-two sizes only, which is all phase_correlate ever asks for: fft.rs next_power_of_two)
 * with a libm twiddle table; the HIP kernel performs the same butterflies in the same order, so
 * GPU and oracle agree bit for bit, and both agree with any exact DFT to ~1e-13.
 * find_peak's tie-break (rayon reduce_with on `>`, schedule dependent in the reference) is pinned
 * to the LOWEST index. */
#include "ab_oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

#define COARSE_MAX_DIM 512   /* phase_correlation.rs:10 */
#define REFINE_CROP_SIZE 512 /* :11 */
#define PC_EPSILON 1e-15     /* :13 */

static size_t next_pow2(size_t n) { size_t p = 1; while (p < n) p <<= 1; return p; }

/* window.rs:3-18 */
void orc_hann_periodic(size_t n, double *w) {
    if (n == 0) return;
    if (n == 1) { w[0] = 1.0; return; }
    const double two_pi = 2.0 * 3.14159265358979323846;
    const double nf = (double)n;
    for (size_t i = 0; i < n; i++) {
        double phase = two_pi * (double)i / nf;
        w[i] = 0.5 * (1.0 - cos(phase));
    }
}

/* forward twiddles tw[k] = exp(-2 pi i k / n), k < n/2 (interleaved re, im) */
void orc_fft_twiddles(size_t n, double *tw) {
    for (size_t k = 0; k < n / 2; k++) {
        double ang = -2.0 * 3.14159265358979323846 * (double)k / (double)n;
        tw[2 * k] = cos(ang);
        tw[2 * k + 1] = sin(ang);
    }
}

/* in-place radix-2 DIT FFT of one line (n power of two; data interleaved re, im with element stride) */
static void fft_line(double *data, size_t n, size_t stride, const double *tw, int inverse, double *tmp) {
    unsigned bits = 0;
    while (((size_t)1 << bits) < n) bits++;
    for (size_t i = 0; i < n; i++) {                                  /* bit reversal into tmp */
        size_t r = 0;
        for (unsigned b = 0; b < bits; b++) if (i & ((size_t)1 << b)) r |= (size_t)1 << (bits - 1 - b);
        tmp[2 * r] = data[2 * i * stride];
        tmp[2 * r + 1] = data[2 * i * stride + 1];
    }
    for (size_t m = 2; m <= n; m <<= 1) {
        size_t half = m >> 1, step = n / m;
        for (size_t k = 0; k < n; k += m) {
            for (size_t j = 0; j < half; j++) {
                double wr = tw[2 * j * step], wi = tw[2 * j * step + 1];
                if (inverse) wi = -wi;
                double xr = tmp[2 * (k + j + half)], xi = tmp[2 * (k + j + half) + 1];
                double tr = wr * xr - wi * xi;
                double ti = wr * xi + wi * xr;
                double ur = tmp[2 * (k + j)], ui = tmp[2 * (k + j) + 1];
                tmp[2 * (k + j)] = ur + tr;
                tmp[2 * (k + j) + 1] = ui + ti;
                tmp[2 * (k + j + half)] = ur - tr;
                tmp[2 * (k + j + half) + 1] = ui - ti;
            }
        }
    }
    for (size_t i = 0; i < n; i++) { data[2 * i * stride] = tmp[2 * i]; data[2 * i * stride + 1] = tmp[2 * i + 1]; }
}

/* fft.rs:136-167: rows then columns; inverse scales by 1/(rows*cols) */
void orc_fft2d(double *buf, size_t rows, size_t cols, int inverse) {
    double *twc = (double *)malloc((cols > 1 ? cols : 2) * sizeof(double));
    double *twr = (double *)malloc((rows > 1 ? rows : 2) * sizeof(double));
    double *tmp = (double *)malloc(2 * (rows > cols ? rows : cols) * sizeof(double));
    orc_fft_twiddles(cols, twc);
    orc_fft_twiddles(rows, twr);
    for (size_t y = 0; y < rows; y++) fft_line(buf + 2 * y * cols, cols, 1, twc, inverse, tmp);
    for (size_t x = 0; x < cols; x++) fft_line(buf + 2 * x, rows, cols, twr, inverse, tmp);
    if (inverse) {
        double norm = 1.0 / (double)(rows * cols);
        for (size_t i = 0; i < rows * cols; i++) { buf[2 * i] = buf[2 * i] * norm; buf[2 * i + 1] = buf[2 * i + 1] * norm; }
    }
    free(twc); free(twr); free(tmp);
}

/* downsample.rs:6-46 */
void orc_area_downsample(const float *src, size_t in_rows, size_t in_cols, size_t out_rows, size_t out_cols, float *out) {
    if (in_rows == out_rows && in_cols == out_cols) { memcpy(out, src, in_rows * in_cols * sizeof(float)); return; }
    double scale_y = (double)in_rows / (double)out_rows, scale_x = (double)in_cols / (double)out_cols;
    for (size_t oy = 0; oy < out_rows; oy++) {
        size_t y0 = orc_clamp_index((int64_t)floor((double)oy * scale_y), in_rows);
        int64_t y1_raw = (int64_t)ceil((double)(oy + 1) * scale_y);
        size_t y1 = y1_raw <= 0 ? 0 : ((size_t)y1_raw < in_rows ? (size_t)y1_raw : in_rows);
        for (size_t ox = 0; ox < out_cols; ox++) {
            size_t x0 = orc_clamp_index((int64_t)floor((double)ox * scale_x), in_cols);
            int64_t x1_raw = (int64_t)ceil((double)(ox + 1) * scale_x);
            size_t x1 = x1_raw <= 0 ? 0 : ((size_t)x1_raw < in_cols ? (size_t)x1_raw : in_cols);
            double sum = 0.0;
            uint32_t count = 0;
            for (size_t y = y0; y < y1; y++)
                for (size_t x = x0; x < x1; x++) {
                    float v = src[y * in_cols + x];
                    if (isfinite(v)) { sum += (double)v; count++; }
                }
            out[oy * out_cols + ox] = count > 0 ? (float)(sum / (double)count) : 0.0f;
        }
    }
}

/* phase_correlation.rs:143-160 */
static int is_constant_or_zero(const float *img, size_t rows, size_t cols, size_t ld) {
    float mn = INFINITY, mx = -INFINITY;
    uint64_t finite = 0;
    for (size_t y = 0; y < rows; y++)
        for (size_t x = 0; x < cols; x++) {
            float v = img[y * ld + x];
            if (isfinite(v)) { if (v < mn) mn = v; if (v > mx) mx = v; finite++; }
        }
    return finite < 16 || fabsf(mx - mn) < 1e-10f;
}

/* subpixel.rs:27-100 (f64 instantiation) */
static double refine_1d(const double *s, size_t rows, size_t cols, size_t py, size_t px, int axis_y) {
    double center = s[py * cols + px], prev, next;
    if (axis_y) {
        size_t p = py == 0 ? rows - 1 : py - 1, n = py == rows - 1 ? 0 : py + 1;
        prev = s[p * cols + px]; next = s[n * cols + px];
    } else {
        size_t p = px == 0 ? cols - 1 : px - 1, n = px == cols - 1 ? 0 : px + 1;
        prev = s[py * cols + p]; next = s[py * cols + n];
    }
    double denom = 2.0 * (2.0 * center - prev - next);
    if (fabs(denom) < 1e-15) return 0.0;                               /* FftFloat::epsilon_val() is 1e-15 for f64 (fft.rs) */
    double r = (prev - next) / denom;
    return fmin(fmax(r, -0.5), 0.5);
}

/* phase_correlation.rs:105-141; a, b contiguous rows x cols */
static void correlate_single(const float *a, const float *b, size_t rows, size_t cols, double *dx, double *dy, double *conf,
                             double *surface_out) {
    size_t fr = next_pow2(rows), fc = next_pow2(cols);
    double *hy = (double *)malloc((rows ? rows : 1) * sizeof(double)), *hx = (double *)malloc((cols ? cols : 1) * sizeof(double));
    orc_hann_periodic(rows, hy);
    orc_hann_periodic(cols, hx);
    double *fa = (double *)calloc(2 * fr * fc, sizeof(double)), *fb = (double *)calloc(2 * fr * fc, sizeof(double));
    for (size_t y = 0; y < rows; y++)                                  /* fft.rs:202-226 */
        for (size_t x = 0; x < cols; x++) {
            double va = (double)a[y * cols + x], vb = (double)b[y * cols + x];
            fa[2 * (y * fc + x)] = isfinite(va) ? va * hy[y] * hx[x] : 0.0;
            fb[2 * (y * fc + x)] = isfinite(vb) ? vb * hy[y] * hx[x] : 0.0;
        }
    orc_fft2d(fa, fr, fc, 0);
    orc_fft2d(fb, fr, fc, 0);
    for (size_t i = 0; i < fr * fc; i++) {                              /* complex.rs:27-44 */
        double ar = fa[2 * i], ai = fa[2 * i + 1], br = fb[2 * i], bi = fb[2 * i + 1];
        double pr = ar * br + ai * bi, pi = ai * br - ar * bi;
        double mag = sqrt(pr * pr + pi * pi);
        if (mag > PC_EPSILON) { fa[2 * i] = pr / mag; fa[2 * i + 1] = pi / mag; }
        else { fa[2 * i] = 0.0; fa[2 * i + 1] = 0.0; }
    }
    orc_fft2d(fa, fr, fc, 1);
    double *corr = (double *)malloc(fr * fc * sizeof(double));
    for (size_t i = 0; i < fr * fc; i++) corr[i] = fa[2 * i];           /* extract_real */
    size_t best = 0;                                                    /* find_peak, lowest index on ties */
    for (size_t i = 1; i < fr * fc; i++) if (corr[i] > corr[best]) best = i;
    size_t py = best / fc, px = best % fc;
    double peak = corr[best];
    double sum = 0.0, count = 0.0;                                      /* normalization.rs:128-161 */
    for (size_t i = 0; i < fr * fc; i++) if (isfinite(corr[i])) { sum = sum + corr[i]; count = count + 1.0; }
    double mean = 0.0, sigma = 0.0;
    if (count >= 1.0) {
        mean = sum / count;
        double var_sum = 0.0;
        for (size_t i = 0; i < fr * fc; i++) if (isfinite(corr[i])) { double d = corr[i] - mean; var_sum = var_sum + d * d; }
        double nm1 = count > 1.0 ? count - 1.0 : 1.0;
        sigma = sqrt(var_sum / nm1);
    }
    *conf = fabs(sigma) < 1e-15 ? 0.0 : (peak - mean) / sigma;                   /* :163-168 */
    double raw_dy = py > fr / 2 ? (double)py - (double)fr : (double)py;          /* subpixel.rs:77-83 */
    double raw_dx = px > fc / 2 ? (double)px - (double)fc : (double)px;
    *dy = raw_dy + refine_1d(corr, fr, fc, py, px, 1);
    *dx = raw_dx + refine_1d(corr, fr, fc, py, px, 0);
    if (surface_out) memcpy(surface_out, corr, fr * fc * sizeof(double));
    free(hy); free(hx); free(fa); free(fb); free(corr);
}

static float *crop_copy(const float *img, size_t ld, size_t y0, size_t y1, size_t x0, size_t x1) {
    size_t r = y1 - y0, c = x1 - x0;
    float *out = (float *)malloc((r * c ? r * c : 1) * sizeof(float));
    for (size_t y = 0; y < r; y++) memcpy(out + y * c, img + (y0 + y) * ld + x0, c * sizeof(float));
    return out;
}

/* Rust `f64 as isize` after round(): saturating, NaN -> 0 */
static int64_t f64_to_i64_sat(double v) {
    if (isnan(v)) return 0;
    if (v >= 9223372036854775807.0) return INT64_MAX;
    if (v <= -9223372036854775808.0) return INT64_MIN;
    return (int64_t)v;
}

/* phase_correlation.rs:22-89 */
void orc_phase_correlate(const float *reference, size_t ref_rows, size_t ref_cols, const float *target, size_t tgt_rows,
                         size_t tgt_cols, double *dx, double *dy, double *confidence) {
    size_t rows = ref_rows < tgt_rows ? ref_rows : tgt_rows, cols = ref_cols < tgt_cols ? ref_cols : tgt_cols;
    *dx = 0.0; *dy = 0.0; *confidence = 0.0;
    if (rows == 0 || cols == 0) return;
    if (is_constant_or_zero(reference, rows, cols, ref_cols) || is_constant_or_zero(target, rows, cols, tgt_cols)) return;
    float *rc = crop_copy(reference, ref_cols, 0, rows, 0, cols), *tc = crop_copy(target, tgt_cols, 0, rows, 0, cols);
    if (rows <= COARSE_MAX_DIM && cols <= COARSE_MAX_DIM) {
        correlate_single(rc, tc, rows, cols, dx, dy, confidence, NULL);
        free(rc); free(tc);
        return;
    }
    double scale_y = (double)rows / (double)COARSE_MAX_DIM, scale_x = (double)cols / (double)COARSE_MAX_DIM;
    size_t ds_rows = COARSE_MAX_DIM < rows ? COARSE_MAX_DIM : rows, ds_cols = COARSE_MAX_DIM < cols ? COARSE_MAX_DIM : cols;
    float *rds = (float *)malloc(ds_rows * ds_cols * sizeof(float)), *tds = (float *)malloc(ds_rows * ds_cols * sizeof(float));
    orc_area_downsample(rc, rows, cols, ds_rows, ds_cols, rds);
    orc_area_downsample(tc, rows, cols, ds_rows, ds_cols, tds);
    double cdx, cdy, cconf;
    correlate_single(rds, tds, ds_rows, ds_cols, &cdx, &cdy, &cconf, NULL);
    double coarse_dx = cdx * scale_x, coarse_dy = cdy * scale_y;
    size_t half = REFINE_CROP_SIZE / 2, ref_cy = rows / 2, ref_cx = cols / 2;
    int64_t ty = f64_to_i64_sat(round((double)ref_cy + coarse_dy)), tx = f64_to_i64_sat(round((double)ref_cx + coarse_dx));
    size_t tgt_cy = (size_t)(ty < 0 ? 0 : (ty > (int64_t)rows - 1 ? (int64_t)rows - 1 : ty));
    size_t tgt_cx = (size_t)(tx < 0 ? 0 : (tx > (int64_t)cols - 1 ? (int64_t)cols - 1 : tx));
#define CROP(cy, cx, Y0, Y1, X0, X1)                                      \
    size_t Y0 = cy > half ? cy - half : 0, Y1 = cy + half < rows ? cy + half : rows, \
           X0 = cx > half ? cx - half : 0, X1 = cx + half < cols ? cx + half : cols;
    CROP(ref_cy, ref_cx, ry0, ry1, rx0, rx1)
    CROP(tgt_cy, tgt_cx, ty0, ty1, tx0, tx1)
    if (ry1 - ry0 != ty1 - ty0 || rx1 - rx0 != tx1 - tx0) {              /* :74-80 */
        *dx = coarse_dx; *dy = coarse_dy; *confidence = cconf;
    } else {
        float *rcrop = crop_copy(rc, cols, ry0, ry1, rx0, rx1), *tcrop = crop_copy(tc, cols, ty0, ty1, tx0, tx1);
        double rdx, rdy, rconf;
        correlate_single(rcrop, tcrop, ry1 - ry0, rx1 - rx0, &rdx, &rdy, &rconf, NULL);
        *dx = coarse_dx + rdx; *dy = coarse_dy + rdy; *confidence = rconf;
        free(rcrop); free(tcrop);
    }
    free(rc); free(tc); free(rds); free(tds);
}

/* exposed for bit-level parity of the correlation surface (rows, cols <= 512) */
void orc_correlate_single(const float *a, const float *b, size_t rows, size_t cols, double *dx, double *dy, double *conf,
                          double *surface) {
    correlate_single(a, b, rows, cols, dx, dy, conf, surface);
}

/* combine.rs:94-193 with config.align == true: crop to min dims, phase-correlate every frame
 * against frame 0 (AlignMethod::PhaseCorrelation is hard-coded, :130), shift it sub-pixel
 * (align.rs:96-98), report rounded offsets (:135-137), then combine. */
int orc_stack_images_align(const float *const *planes, const int64_t *rows, const int64_t *cols, size_t n_images,
                           float sigma_low, float sigma_high, size_t max_iter, int order_mode, int threads, float *out,
                           uint64_t *out_rejected, int32_t *offsets_dy_dx) {
    if (n_images == 0) return -1;
    int64_t min_rows = rows[0], min_cols = cols[0];
    for (size_t i = 1; i < n_images; i++) {
        if (rows[i] < min_rows) min_rows = rows[i];
        if (cols[i] < min_cols) min_cols = cols[i];
    }
    float **aligned = (float **)malloc(n_images * sizeof(float *));
    int64_t *r = (int64_t *)malloc(n_images * sizeof(int64_t)), *c = (int64_t *)malloc(n_images * sizeof(int64_t));
    for (size_t i = 0; i < n_images; i++) { r[i] = min_rows; c[i] = min_cols; }
    aligned[0] = crop_copy(planes[0], (size_t)cols[0], 0, (size_t)min_rows, 0, (size_t)min_cols);
    if (offsets_dy_dx) { offsets_dy_dx[0] = 0; offsets_dy_dx[1] = 0; }
    for (size_t i = 1; i < n_images; i++) {
        float *cropped = crop_copy(planes[i], (size_t)cols[i], 0, (size_t)min_rows, 0, (size_t)min_cols);
        double dx, dy, conf;
        orc_phase_correlate(aligned[0], (size_t)min_rows, (size_t)min_cols, cropped, (size_t)min_rows, (size_t)min_cols, &dx, &dy,
                            &conf);
        aligned[i] = (float *)malloc((size_t)(min_rows * min_cols) * sizeof(float));
        orc_shift_image_subpixel(cropped, (size_t)min_rows, (size_t)min_cols, dy, dx, threads, aligned[i]);
        if (offsets_dy_dx) {
            offsets_dy_dx[2 * i] = (int32_t)f64_to_i64_sat(round(dy));
            offsets_dy_dx[2 * i + 1] = (int32_t)f64_to_i64_sat(round(dx));
        }
        free(cropped);
    }
    int rc = orc_stack_images_noalign((const float *const *)aligned, r, c, n_images, sigma_low, sigma_high, max_iter, order_mode,
                                      threads, out, out_rejected, NULL, NULL);
    for (size_t i = 0; i < n_images; i++) free(aligned[i]);
    free(aligned); free(r); free(c);
    return rc;
}
