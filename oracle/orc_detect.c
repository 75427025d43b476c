/* ORACLE (test infrastructure).  Restates core/analysis/star_detection.rs (estimate_background
 * :32-84, detect_stars :86-258), core/analysis/confidence.rs:3-8 and
 * core/alignment/affine.rs:24-53 (normalize_for_detection).  See ab_oracle.h for the rules.
 *
 * The labelling here IS the reference's sequential raster-scan + 8-connected BFS, including its
 * quirks: seeds are interior pixels only (1..rows-1, 1..cols-1) while growth may enter the border;
 * components outside 3..5000 pixels are dropped but stay visited; moments are summed in BFS order. */
#include "ab_oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

static int cmp_f64(const void *pa, const void *pb) {   /* f64_cmp, NaN last (math/median.rs:15-25) */
    double a = *(const double *)pa, b = *(const double *)pb;
    if (a < b) return -1;
    if (a > b) return 1;
    if (a == b) return 0;
    int an = isnan(a), bn = isnan(b);
    return an && bn ? 0 : (an ? 1 : -1);
}

/* star_detection.rs:32-84 */
void orc_estimate_background(const float *image, size_t rows, size_t cols, size_t tile_size, double *out_median,
                             double *out_sigma) {
    size_t step = tile_size > 16 ? tile_size : 16;
    size_t nty = (rows + step - 1) / step, ntx = (cols + step - 1) / step;
    double *medians = (double *)malloc((nty * ntx ? nty * ntx : 1) * sizeof(double));
    double *sigmas = (double *)malloc((nty * ntx ? nty * ntx : 1) * sizeof(double));
    float *vals = (float *)malloc(step * step * sizeof(float));
    size_t nres = 0;
    for (size_t ty = 0; ty < rows; ty += step)
        for (size_t tx = 0; tx < cols; tx += step) {
            size_t ye = ty + step < rows ? ty + step : rows, xe = tx + step < cols ? tx + step : cols;
            size_t n = 0;
            for (size_t r = ty; r < ye; r++)
                for (size_t c = tx; c < xe; c++) {
                    float v = image[r * cols + c];
                    if (isfinite(v) && v > 1e-7f) vals[n++] = v;
                }
            if (n >= 8) {
                double med, sig;
                orc_sigma_clipped_stats(vals, &n, 3.0f, 2, &med, &sig);
                medians[nres] = med;
                sigmas[nres] = sig;
                nres++;
            }
        }
    if (nres == 0) { *out_median = 0.0; *out_sigma = 1.0; }
    else {
        qsort(medians, nres, sizeof(double), cmp_f64);
        qsort(sigmas, nres, sizeof(double), cmp_f64);
        *out_median = medians[nres / 2];
        *out_sigma = fmax(sigmas[nres / 2], 1e-10);
    }
    free(medians); free(sigmas); free(vals);
}

static int cmp_star_flux_desc(const void *pa, const void *pb) {  /* stable: ties keep discovery order */
    const orc_star *a = (const orc_star *)pa, *b = (const orc_star *)pb;
    if (b->flux < a->flux) return -1;
    if (b->flux > a->flux) return 1;
    return a->order < b->order ? -1 : (a->order > b->order ? 1 : 0);
}

/* star_detection.rs:86-258.  Returns the number of stars written (<= cap); *total = all found. */
size_t orc_detect_stars(const float *image, size_t rows, size_t cols, double sigma_threshold, orc_star *out, size_t cap,
                        size_t *total, double *bg_median_out, double *bg_sigma_out) {
    *total = 0;
    if (rows < 3 || cols < 3) { *bg_median_out = 0.0; *bg_sigma_out = 1.0; return 0; }
    size_t m = rows < cols ? rows : cols;
    size_t tile_size = m / 8;
    if (tile_size < 32) tile_size = 32;
    if (tile_size > 256) tile_size = 256;
    double bg_median, bg_sigma;
    orc_estimate_background(image, rows, cols, tile_size, &bg_median, &bg_sigma);
    *bg_median_out = bg_median; *bg_sigma_out = bg_sigma;
    double threshold = bg_median + sigma_threshold * bg_sigma;

    unsigned char *visited = (unsigned char *)calloc(rows * cols, 1);
    size_t *queue = (size_t *)malloc(rows * cols * sizeof(size_t));
    size_t nstars = 0, cap_s = 1024;
    orc_star *stars = (orc_star *)malloc(cap_s * sizeof(orc_star));
    static const int DR[8] = {-1, 1, 0, 0, -1, -1, 1, 1}, DC[8] = {0, 0, -1, 1, -1, 1, -1, 1};   /* :120 */

    for (size_t r = 1; r + 1 < rows; r++)
        for (size_t c = 1; c + 1 < cols; c++) {
            double v = (double)image[r * cols + c];
            if (v <= threshold || visited[r * cols + c] || !isfinite(v)) continue;
            size_t head = 0, tail = 0;
            queue[tail++] = r * cols + c;
            visited[r * cols + c] = 1;
            while (head < tail) {                                       /* pop_front; component = queue[0..tail) */
                size_t cur = queue[head++];
                size_t cr = cur / cols, cc = cur % cols;
                for (int k = 0; k < 8; k++) {
                    long nr = (long)cr + DR[k], nc = (long)cc + DC[k];
                    if (nr < 0 || nc < 0 || nr >= (long)rows || nc >= (long)cols) continue;
                    size_t ni = (size_t)nr * cols + (size_t)nc;
                    if (visited[ni]) continue;
                    double nv = (double)image[ni];
                    if (nv > threshold && isfinite(nv)) { visited[ni] = 1; queue[tail++] = ni; }
                }
            }
            size_t npix = tail;
            if (npix < 3 || npix > 5000) continue;
            double sum_flux = 0.0, sum_x = 0.0, sum_y = 0.0, peak_val = 0.0;
            for (size_t i = 0; i < npix; i++) {
                size_t pr = queue[i] / cols, pc = queue[i] % cols;
                double w = fmax((double)image[queue[i]] - bg_median, 0.0);
                sum_flux += w; sum_x += (double)pc * w; sum_y += (double)pr * w;
                peak_val = fmax(peak_val, w);
            }
            if (sum_flux <= 0.0) continue;
            double cx = sum_x / sum_flux, cy = sum_y / sum_flux;
            double sum_r2 = 0.0, sum_xx = 0.0, sum_yy = 0.0, sum_xy = 0.0;
            for (size_t i = 0; i < npix; i++) {
                size_t pr = queue[i] / cols, pc = queue[i] % cols;
                double w = fmax((double)image[queue[i]] - bg_median, 0.0);
                double dx = (double)pc - cx, dy = (double)pr - cy;
                sum_r2 += (dx * dx + dy * dy) * w;
                sum_xx += dx * dx * w; sum_yy += dy * dy * w; sum_xy += dx * dy * w;
            }
            double sigma_star = sqrt(sum_r2 / (2.0 * sum_flux));
            double fwhm = sigma_star * 2.3548200450309493;
            if (fwhm < 0.5 || fwhm > 30.0) continue;
            double ixx = sum_xx / sum_flux, iyy = sum_yy / sum_flux, ixy = sum_xy / sum_flux;
            double trace = ixx + iyy;
            double det = fmax(ixx * iyy - ixy * ixy, 0.0);
            double disc = sqrt(fmax((trace * trace / 4.0) - det, 0.0));
            double lambda1 = trace / 2.0 + disc, lambda2 = fmax(trace / 2.0 - disc, 0.0);
            double ecc = 0.0;
            if (lambda1 > 1e-15) { ecc = sqrt(1.0 - lambda2 / lambda1); ecc = ecc < 0.0 ? 0.0 : (ecc > 1.0 ? 1.0 : ecc); }
            double snr = bg_sigma <= 2.220446049250313e-16 ? 0.0 : peak_val / bg_sigma;    /* confidence.rs:3-8 */
            if (nstars == cap_s) { cap_s *= 2; stars = (orc_star *)realloc(stars, cap_s * sizeof(orc_star)); }
            orc_star *s = &stars[nstars];
            s->x = cx; s->y = cy; s->flux = sum_flux; s->fwhm = fwhm; s->eccentricity = ecc; s->peak = peak_val;
            s->npix = npix; s->snr = snr; s->order = nstars;
            nstars++;
        }

    qsort(stars, nstars, sizeof(orc_star), cmp_star_flux_desc);          /* :215 (stable) */

    /* dedup radius 3 px (:217-248); a linear scan over the kept stars is equivalent to the 3 px grid */
    size_t kept = 0;
    orc_star *dedup = (orc_star *)malloc((nstars ? nstars : 1) * sizeof(orc_star));
    for (size_t i = 0; i < nstars; i++) {
        int too_close = 0;
        size_t gx = (size_t)(stars[i].x / 3.0), gy = (size_t)(stars[i].y / 3.0);
        for (size_t j = 0; j < kept && !too_close; j++) {
            size_t hx = (size_t)(dedup[j].x / 3.0), hy = (size_t)(dedup[j].y / 3.0);
            /* only stars in the 3x3 neighbourhood of grid cells are compared in the reference */
            size_t ddx = hx > gx ? hx - gx : gx - hx, ddy = hy > gy ? hy - gy : gy - hy;
            if (ddx > 1 || ddy > 1) continue;
            double dx = stars[i].x - dedup[j].x, dy = stars[i].y - dedup[j].y;
            if (dx * dx + dy * dy < 9.0) too_close = 1;
        }
        if (!too_close) dedup[kept++] = stars[i];
    }
    *total = kept;
    size_t nout = kept < cap ? kept : cap;
    if (out) memcpy(out, dedup, nout * sizeof(orc_star));
    free(visited); free(queue); free(stars); free(dedup);
    return nout;
}

static int cmp_f32_partial(const void *pa, const void *pb) {   /* partial_cmp(...).unwrap_or(Equal) */
    float a = *(const float *)pa, b = *(const float *)pb;
    return a < b ? -1 : (a > b ? 1 : 0);
}

/* affine.rs:24-53.  Returns 1 when the image is returned unchanged (clone paths). */
int orc_normalize_for_detection(const float *image, size_t len, float *out) {
    if (len == 0) return 1;
    size_t step = len / 100000 > 1 ? len / 100000 : 1;
    size_t cap = len / step + 1, ns = 0;
    float *samples = (float *)malloc(cap * sizeof(float));
    for (size_t i = 0; i < len; i += step) if (isfinite(image[i])) samples[ns++] = image[i];
    if (ns < 100) { memcpy(out, image, len * sizeof(float)); free(samples); return 1; }
    qsort(samples, ns, sizeof(float), cmp_f32_partial);
    double lo = (double)samples[ns / 100], hi = (double)samples[ns * 999 / 1000];
    free(samples);
    double range = hi - lo;
    if (range < 1e-15) { memcpy(out, image, len * sizeof(float)); return 1; }
    double inv_range = 1.0 / range;
    for (size_t i = 0; i < len; i++) {
        double t = ((double)image[i] - lo) * inv_range;
        t = t < 0.0 ? 0.0 : (t > 1.0 ? 1.0 : t);                       /* f64::clamp: NaN stays NaN */
        out[i] = (float)t;
    }
    return 0;
}
