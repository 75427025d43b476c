// The per-pixel screen transfer function shared by stf.hip (whole-plane maps) and render.hip (previews and tiles):
// StfTransform (core/imaging/stf.rs:60-87), apply_stf's u8 rounding (:96-100) == make_stf_u8_fn (:122-145).
#pragma once
#include "ab_common.hpp"

#include <cmath>

namespace {

constexpr float kPaddingThreshold = 1e-7f;  // types/constants.rs:6

struct StfTx {  // stf.rs:60-78
    double inv_range, dmin, shadow, inv_clip, midtone;
};

__device__ __forceinline__ bool is_valid_pixel(float v) { return __builtin_isfinite(v) && v > kPaddingThreshold; }

__device__ __forceinline__ double mtf(double x, double m) {  // stf.rs:50-58
    if (x <= 0.0) return 0.0;
    if (x >= 1.0) return 1.0;
    return (m - 1.0) * x / ((2.0 * m - 1.0) * x - m);
}

__device__ __forceinline__ double tx_apply(const StfTx &t, double v) {  // stf.rs:80-86
    const double norm = (v - t.dmin) * t.inv_range;
    double clipped = (norm - t.shadow) * t.inv_clip;
    clipped = clipped < 0.0 ? 0.0 : (clipped > 1.0 ? 1.0 : clipped);  // f64::clamp (NaN stays NaN)
    return mtf(clipped, t.midtone);
}

__device__ __forceinline__ unsigned char to_u8(float v, const StfTx &t) {  // stf.rs:96-100
    if (!is_valid_pixel(v)) return 0;
    double r = round(tx_apply(t, (double)v) * 255.0);
    r = r < 0.0 ? 0.0 : (r > 255.0 ? 255.0 : r);
    return (r > 0.0) ? (unsigned char)r : (unsigned char)0;  // `as u8`: NaN -> 0
}

__device__ __forceinline__ float to_f32(float v, const StfTx &t) {  // stf.rs:112-116
    return is_valid_pixel(v) ? (float)tx_apply(t, (double)v) : 0.0f;
}

__host__ __device__ inline StfTx make_tx(const ab_stf_params *p, const ab_image_stats *st) {  // stf.rs:69-78
    StfTx t;
    const double range = fmax(st->max - st->min, 1e-30);
    const double clip_range = fmax(p->highlight - p->shadow, 1e-15);
    t.inv_range = 1.0 / range;
    t.dmin = st->min;
    t.shadow = p->shadow;
    t.inv_clip = 1.0 / clip_range;
    t.midtone = p->midtone;
    return t;
}

__host__ __device__ inline double clampd_hd(double x, double lo, double hi) { return x < lo ? lo : (x > hi ? hi : x); }

// auto_stf (stf.rs:13-39) with mtf_balance (:41-47): scalar f64 maths, the same instruction sequence on the host (ab_auto_stf)
// and at the end of the device-side statistics chain (stats.hip), IEEE basic operations only
__host__ __device__ inline void ab_auto_stf_hd(const ab_image_stats *stats, const ab_auto_stf_config *cfg, ab_stf_params *out) {
    if (stats->valid_count == 0) {
        out->shadow = 0.0;
        out->midtone = 0.5;
        out->highlight = 1.0;
        return;
    }
    const double range = fmax(stats->max - stats->min, 1e-30);
    const double median_norm = (stats->median - stats->min) / range;
    const double sigma_norm = stats->sigma / range;
    const double shadow_norm = clampd_hd(median_norm + cfg->shadow_k * sigma_norm, 0.0, 0.98);
    const double highlight_norm = 1.0;
    const double clip_range = fmax(highlight_norm - shadow_norm, 1e-15);
    const double m_clipped = clampd_hd((median_norm - shadow_norm) / clip_range, 0.0, 1.0);
    double midtone = 0.5;
    if (!(m_clipped <= 0.0 || m_clipped >= 1.0)) {
        const double t = cfg->target_bg, m = m_clipped;  // mtf_balance, stf.rs:41-47
        const double denom = 2.0 * t * m - t - m;
        midtone = fabs(denom) < 1e-15 ? 0.5 : clampd_hd(m * (t - 1.0) / denom, 0.0001, 0.9999);
    }
    out->shadow = shadow_norm;
    out->midtone = midtone;
    out->highlight = highlight_norm;
}

}  // namespace
