/* ORACLE (test infrastructure).  Restates core/astrometry/spcc.rs: spcc_calibrate_rgb (:73-183),
 * synthesize_luminance (:185-196), bp_rp_to_teff (:198-213), planck_rgb / planck_intensity (:215-243),
 * white_reference_rgb (:245-255), estimate_bp_rp_from_flux (:275-279), cross_match_stars (:285-339),
 * aperture_flux_f32 (:341-383), compute_correction_factors (:385-435).
 *
 * The WCS enters the built-in (synthetic) catalogue path only through cross_match_stars: the catalogue is
 * generated FROM the detections' own world coordinates (:257-273), so star i's nearest catalogue entry is
 * entry i at distance 0, accepted iff 0 < match_r2 = (pixel_scale * 3 / 3600)^2.  The caller therefore
 * passes the WCS pixel scale (arcsec / px); header parsing (wcs.rs) is outside the hot path.
 * See ab_oracle.h for the rules. */
#include "ab_oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

static double clampd(double v, double lo, double hi) { return v < lo ? lo : (v > hi ? hi : v); }

static double bp_rp_to_teff(double bp_rp) {                                        /* :198-213 */
    double x = clampd(bp_rp, -0.5, 5.0);
    if (x < 0.0) return 10000.0 + (-x) * 20000.0;
    if (x < 0.5) return 7500.0 + (0.5 - x) * 5000.0;
    if (x < 1.0) return 5800.0 + (1.0 - x) * 3400.0;
    if (x < 1.5) return 4500.0 + (1.5 - x) * 2600.0;
    if (x < 2.5) return 3500.0 + (2.5 - x) * 1000.0;
    return 2800.0 + (5.0 - x) * 280.0;
}

static double planck_intensity(double teff, double wavelength_nm) {               /* :228-243 */
    double lambda = wavelength_nm * 1e-9, h = 6.626e-34, c = 2.998e8, k = 1.381e-23;
    double exponent = h * c / (lambda * k * teff);
    if (exponent > 500.0) return 0.0;
    double l2 = lambda * lambda;                                                   /* powi(5): l * (l^2)^2 (compiler-rt) */
    double l5 = lambda * (l2 * l2);
    double numerator = 2.0 * h * c * c / l5;
    return numerator / (exp(exponent) - 1.0);
}

static void planck_rgb(double teff, double out[3]) {                              /* :215-226 */
    double r = planck_intensity(teff, 640.0), g = planck_intensity(teff, 530.0), b = planck_intensity(teff, 460.0);
    double max_val = fmax(fmax(r, g), b);
    if (max_val < 1e-30) { out[0] = out[1] = out[2] = 1.0; return; }
    out[0] = r / max_val; out[1] = g / max_val; out[2] = b / max_val;
}

void orc_spcc_white_reference_rgb(int kind, const double custom[3], double out[3]) {   /* :245-255 */
    if (kind == 1) planck_rgb(5778.0, out);                                        /* G2V */
    else if (kind == 0) { planck_rgb(5500.0, out); out[0] *= 0.98; out[1] *= 1.0; out[2] *= 1.02; }   /* AverageSpiral */
    else if (kind == 2) out[0] = out[1] = out[2] = 1.0;                            /* Photopic */
    else memcpy(out, custom, 3 * sizeof(double));                                  /* Custom */
}

static double estimate_bp_rp_from_flux(const orc_star *s) {                        /* :275-279 */
    double norm_flux = clampd(s->flux / fmax(s->peak, 1e-10), 0.1, 100.0);
    double fwhm_factor = clampd(s->fwhm - 3.0, -2.0, 5.0) * 0.1;
    return clampd(1.0 / sqrt(norm_flux) + fwhm_factor, -0.3, 4.0);
}

static size_t sat_usize(double v) { return !(v > 0.0) ? 0 : (v >= 1.8e19 ? (size_t)-1 : (size_t)v); }

double orc_aperture_flux_f32(const float *image, size_t h, size_t w, double x, double y, double radius) {   /* :341-383 */
    double r2 = radius * radius, inner = radius * 1.2, outer = radius * 1.8;
    double inner_r2 = inner * inner, outer_r2 = outer * outer, flux = 0.0, bg_sum = 0.0;
    uint32_t bg_count = 0;
    size_t y_min = sat_usize(fmax(floor(y - outer), 0.0)), y_max = sat_usize(ceil(y + outer));
    size_t x_min = sat_usize(fmax(floor(x - outer), 0.0)), x_max = sat_usize(ceil(x + outer));
    size_t hm = h ? h - 1 : 0, wm = w ? w - 1 : 0;
    if (y_max > hm) y_max = hm;
    if (x_max > wm) x_max = wm;
    for (size_t py = y_min; py <= y_max; py++)
        for (size_t px = x_min; px <= x_max; px++) {
            double dx = (double)px - x, dy = (double)py - y, d2 = dx * dx + dy * dy;
            double v = (double)image[py * w + px];
            if (d2 <= r2) flux += v;
            else if (d2 >= inner_r2 && d2 <= outer_r2) { bg_sum += v; bg_count++; }
        }
    if (bg_count > 0) {
        double bg_per_pixel = bg_sum / (double)bg_count;
        flux -= bg_per_pixel * (3.14159265358979323846264338327950288 * r2);
    }
    return flux > 0.0 ? flux : 0.0;                                                /* f64::max(0.0): NaN -> 0.0 */
}

typedef struct { double bp_rp, r, g, b; } matched_t;

static void compute_correction_factors(const matched_t *m, size_t n, const double wr[3], double out[4]) {   /* :385-435 */
    double sr = 0.0, sg = 0.0, sb = 0.0, sw = 0.0, sci = 0.0;
    for (size_t i = 0; i < n; i++) {
        double e[3];
        planck_rgb(bp_rp_to_teff(m[i].bp_rp), e);
        double total_measured = m[i].r + m[i].g + m[i].b, total_expected = e[0] + e[1] + e[2];
        if (total_measured < 1e-10 || total_expected < 1e-10) continue;
        double weight = sqrt(total_measured);
        double mr = m[i].r / total_measured, mg = m[i].g / total_measured, mb = m[i].b / total_measured;
        double er = e[0] / total_expected, eg = e[1] / total_expected, eb = e[2] / total_expected;
        if (mr > 1e-6) sr += (er / mr) * weight;
        if (mg > 1e-6) sg += (eg / mg) * weight;
        if (mb > 1e-6) sb += (eb / mb) * weight;
        sw += weight;
        sci += m[i].bp_rp;
    }
    if (sw < 1e-10 || n == 0) { out[0] = out[1] = out[2] = 1.0; out[3] = 0.0; return; }
    double rf = sr / sw, gf = sg / sw, bf = sb / sw;
    rf *= wr[0]; gf *= wr[1]; bf *= wr[2];
    double norm = gf;
    if (norm > 1e-10) { rf /= norm; gf = 1.0; bf /= norm; }
    out[0] = rf; out[1] = gf; out[2] = bf; out[3] = sci / (double)n;
}

static int by_snr_desc(const void *a, const void *b) {   /* stable: ties keep detection order (slice::sort_by) */
    const orc_star *x = *(const orc_star *const *)a, *y = *(const orc_star *const *)b;
    if (x->snr > y->snr) return -1;
    if (x->snr < y->snr) return 1;
    return x < y ? -1 : (x > y ? 1 : 0);
}

/* the part of spcc_calibrate_rgb after detection (:90-183) on a given detection (stars in detect_stars order) and
 * luminance maximum.  Returns 0 ok; 1 "Only N stars passed quality filters" (N in res->stars_total);
 * 2 "Only N stars cross-matched" (N in res->stars_matched). */
int orc_spcc_from_detection(const float *r, const float *g, const float *b, size_t h, size_t w, const orc_star *stars,
                            size_t n_stars, double lum_max, double pixel_scale_arcsec, const orc_spcc_config *cfg,
                            orc_spcc_result *res) {
    memset(res, 0, sizeof *res);
    float sat_limit = (float)(lum_max * cfg->saturation_limit);                    /* :89 */
    double x_hi = (double)(size_t)(w - 10), y_hi = (double)(size_t)(h - 10);       /* usize wrap when < 10, as release builds do */
    const orc_star **good = (const orc_star **)malloc((n_stars ? n_stars : 1) * sizeof(*good));
    size_t ng = 0;
    for (size_t i = 0; i < n_stars; i++) {
        const orc_star *s = &stars[i];
        if (s->snr >= cfg->min_snr && s->peak < (double)sat_limit && s->x >= 10.0 && s->y >= 10.0 && s->x < x_hi && s->y < y_hi)
            good[ng++] = s;
    }
    qsort(good, ng, sizeof(*good), by_snr_desc);
    if (ng > cfg->max_stars) ng = (size_t)cfg->max_stars;
    res->stars_total = ng;
    if (ng < 5) { free(good); return 1; }
    double match_radius = (pixel_scale_arcsec * 3.0) / 3600.0, match_r2 = match_radius * match_radius;
    matched_t *m = (matched_t *)malloc(ng * sizeof(matched_t));
    size_t nm = 0;
    for (size_t i = 0; i < ng; i++) {
        if (!(0.0 < match_r2)) continue;                                           /* d2 = 0 < match_r2 (see header note) */
        double radius = fmax(good[i]->fwhm * 1.5, 3.0);
        double rf = orc_aperture_flux_f32(r, h, w, good[i]->x, good[i]->y, radius);
        double gf = orc_aperture_flux_f32(g, h, w, good[i]->x, good[i]->y, radius);
        double bf = orc_aperture_flux_f32(b, h, w, good[i]->x, good[i]->y, radius);
        if (rf > 0.0 && gf > 0.0 && bf > 0.0) { m[nm].bp_rp = estimate_bp_rp_from_flux(good[i]); m[nm].r = rf; m[nm].g = gf; m[nm].b = bf; nm++; }
    }
    res->stars_matched = nm;
    if (nm < 3) { free(m); free(good); return 2; }
    double wr[3], out[4];
    orc_spcc_white_reference_rgb(cfg->white_reference, cfg->custom, wr);
    compute_correction_factors(m, nm, wr, out);
    res->r_factor = out[0]; res->g_factor = out[1]; res->b_factor = out[2]; res->avg_color_index = out[3];
    free(m);
    free(good);
    return 0;
}

/* spcc_calibrate_rgb (:73-183) with the WCS reduced to its pixel scale */
int orc_spcc_calibrate_rgb(const float *r, const float *g, const float *b, size_t h, size_t w, double pixel_scale_arcsec,
                           const orc_spcc_config *cfg, orc_spcc_result *res) {
    size_t n = h * w;
    float *lum = (float *)malloc((n ? n : 1) * sizeof(float));
    for (size_t i = 0; i < n; i++) lum[i] = 0.2126f * r[i] + 0.7152f * g[i] + 0.0722f * b[i];       /* :185-196 */
    size_t cap = 1u << 16, total = 0;
    double bm, bs;
    orc_star *st = (orc_star *)malloc(cap * sizeof(orc_star));
    size_t ns = orc_detect_stars(lum, h, w, 5.0, st, cap, &total, &bm, &bs);
    if (total > cap) {
        cap = total;
        st = (orc_star *)realloc(st, cap * sizeof(orc_star));
        ns = orc_detect_stars(lum, h, w, 5.0, st, cap, &total, &bm, &bs);
    }
    orc_image_stats stats;
    orc_compute_image_stats(lum, n, &stats);
    int rc = orc_spcc_from_detection(r, g, b, h, w, st, ns, stats.max, pixel_scale_arcsec, cfg, res);
    free(st);
    free(lum);
    return rc;
}
