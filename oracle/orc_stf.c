/* ORACLE (test infrastructure).  Restates src-tauri/src/core/imaging/stf.rs.
 * See ab_oracle.h for the rules. */
#include "ab_oracle.h"
#include <math.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define PADDING_THRESHOLD 1e-7f
static inline int is_valid_pixel(float v) { return isfinite(v) && v > PADDING_THRESHOLD; } /* stats.rs:10-13 */

/* f64::clamp: NaN stays NaN */
static inline double clampd(double x, double lo, double hi) { return x < lo ? lo : (x > hi ? hi : x); }

/* stf.rs:41-47 */
double orc_mtf_balance(double m, double t) {
    double denom = 2.0 * t * m - t - m;
    if (fabs(denom) < 1e-15) return 0.5;
    return clampd(m * (t - 1.0) / denom, 0.0001, 0.9999);
}

/* stf.rs:50-58 */
double orc_mtf(double x, double m) {
    if (x <= 0.0) return 0.0;
    if (x >= 1.0) return 1.0;
    return (m - 1.0) * x / ((2.0 * m - 1.0) * x - m);
}

/* stf.rs:13-39 */
void orc_auto_stf(const orc_image_stats *st, double target_bg, double shadow_k, orc_stf_params *out) {
    if (st->valid_count == 0) { out->shadow = 0.0; out->midtone = 0.5; out->highlight = 1.0; return; }
    double range = fmax(st->max - st->min, 1e-30);
    double median_norm = (st->median - st->min) / range;
    double sigma_norm = st->sigma / range;
    double shadow_norm = clampd(median_norm + shadow_k * sigma_norm, 0.0, 0.98);
    double highlight_norm = 1.0;
    double clip_range = fmax(highlight_norm - shadow_norm, 1e-15);
    double m_clipped = clampd((median_norm - shadow_norm) / clip_range, 0.0, 1.0);
    double midtone = (m_clipped <= 0.0 || m_clipped >= 1.0) ? 0.5 : orc_mtf_balance(m_clipped, target_bg);
    out->shadow = shadow_norm; out->midtone = midtone; out->highlight = highlight_norm;
}

/* stf.rs:60-87 StfTransform */
typedef struct { double inv_range, dmin, shadow, inv_clip, midtone; } stf_tx;
static stf_tx tx_new(const orc_stf_params *p, const orc_image_stats *st) {
    stf_tx t;
    double range = fmax(st->max - st->min, 1e-30);
    double clip_range = fmax(p->highlight - p->shadow, 1e-15);
    t.inv_range = 1.0 / range; t.dmin = st->min; t.shadow = p->shadow;
    t.inv_clip = 1.0 / clip_range; t.midtone = p->midtone;
    return t;
}
static inline double tx_apply(const stf_tx *t, double v) {
    double norm = (v - t->dmin) * t->inv_range;
    double clipped = clampd((norm - t->shadow) * t->inv_clip, 0.0, 1.0);
    return orc_mtf(clipped, t->midtone);
}

/* Rust `f64 as u8`: saturating, NaN -> 0 */
static inline uint8_t f64_to_u8_sat(double v) {
    if (!(v > 0.0)) return 0;
    if (v >= 255.0) return 255;
    return (uint8_t)v;
}

/* stf.rs:89-102 */
void orc_apply_stf_u8(const float *data, size_t n, const orc_stf_params *p, const orc_image_stats *st,
                      int threads, uint8_t *out) {
    stf_tx t = tx_new(p, st);
#ifdef _OPENMP
    if (threads <= 0) threads = omp_get_max_threads();
#else
    threads = 1;
#endif
#pragma omp parallel for num_threads(threads) schedule(static)
    for (int64_t i = 0; i < (int64_t)n; i++) {
        float v = data[i];
        if (!is_valid_pixel(v)) { out[i] = 0; continue; }
        out[i] = f64_to_u8_sat(clampd(round(tx_apply(&t, (double)v) * 255.0), 0.0, 255.0));
    }
}

/* stf.rs:104-120 (and the in-place twin :147-155) */
void orc_apply_stf_f32(const float *data, size_t n, const orc_stf_params *p, const orc_image_stats *st,
                       int threads, float *out) {
    stf_tx t = tx_new(p, st);
#ifdef _OPENMP
    if (threads <= 0) threads = omp_get_max_threads();
#else
    threads = 1;
#endif
#pragma omp parallel for num_threads(threads) schedule(static)
    for (int64_t i = 0; i < (int64_t)n; i++) {
        float v = data[i];
        out[i] = is_valid_pixel(v) ? (float)tx_apply(&t, (double)v) : 0.0f;
    }
}
