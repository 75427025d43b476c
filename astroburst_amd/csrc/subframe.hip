// Subframe scoring: core/analysis/subframe.rs.  The pixel work is detect_stars(image, 4.0) -- the
// detect.hip kernels, one frame per worker stream -- and the rest is a handful of f64 scalars per frame on the host.
#include "ab_common.hpp"

#include <algorithm>
#include <cmath>

namespace {

constexpr double kDetectionSigma = 4.0;        // subframe.rs:6
constexpr uint64_t kMinStarsForMetrics = 5;    // subframe.rs:7

template <class F>
double median_of(const std::vector<ab_detected_star> &stars, F field) {  // subframe.rs:161-173
    std::vector<double> v;
    v.reserve(stars.size());
    for (const ab_detected_star &s : stars) {
        const double x = field(s);
        if (std::isfinite(x)) v.push_back(x);
    }
    if (v.empty()) return 0.0;
    std::sort(v.begin(), v.end());
    const size_t mid = v.size() / 2;
    return v.size() % 2 == 0 ? (v[mid - 1] + v[mid]) / 2.0 : v[mid];
}

double compute_weight(double fwhm, double ecc, double snr, double noise, const ab_subframe_weight_config &c) {  // :123-146
    const double fwhm_score = fwhm > 0.5 ? 1.0 / fwhm : 0.0;
    const double ecc_score = 1.0 - ecc;
    const double snr_score = std::fmax(std::log(snr), 0.0);
    const double noise_score = 1.0 / (1.0 + noise * 10.0);
    const double total = c.fwhm_weight + c.eccentricity_weight + c.snr_weight + c.noise_weight;
    if (total < 1e-15) return 0.0;
    const double raw = c.fwhm_weight * fwhm_score + c.eccentricity_weight * ecc_score + c.snr_weight * snr_score + c.noise_weight * noise_score;
    return std::fmax(raw / total, 0.0);
}

int analyze_one(ab_ctx *ctx, const ab_plane *img, const ab_subframe_weight_config &c, ab_subframe_metrics *out, const double *bg = nullptr) {  // :51-121
    StagedPlane in;
    AB_TRY(ab_stage_in(ctx, img, &in));
    std::vector<ab_detected_star> stars;
    double bm = 0.0, bs = 1.0;
    const int rc = ab_detect_stars_device(ctx, in.dptr, in.rows, in.cols, in.cols, kDetectionSigma, &stars, &bm, &bs, ab_pixel_xf(), (size_t)-1, false, bg);
    ab_stage_release(ctx, &in);
    if (rc != AB_OK) return rc;
    *out = ab_subframe_metrics{};
    out->star_count = stars.size();
    out->background_median = bm;
    out->background_sigma = bs;
    if (out->star_count < std::min<uint64_t>(kMinStarsForMetrics, c.min_stars)) return AB_OK;  // weight 0, rejected (:66-79)
    out->median_fwhm = median_of(stars, [](const ab_detected_star &s) { return s.fwhm; });
    out->median_eccentricity = median_of(stars, [](const ab_detected_star &s) { return s.eccentricity; });
    out->median_snr = median_of(stars, [](const ab_detected_star &s) { return s.snr; });
    out->noise_ratio = bm > 1e-15 ? bs / bm : 0.0;
    out->weight = compute_weight(out->median_fwhm, out->median_eccentricity, out->median_snr, out->noise_ratio, c);
    out->accepted = out->star_count >= c.min_stars && out->median_fwhm <= c.max_fwhm && out->median_eccentricity <= c.max_eccentricity &&
                    out->median_snr >= c.min_snr;
    return AB_OK;
}

}  // namespace

extern "C" {

void ab_subframe_weight_config_default(ab_subframe_weight_config *c) {  // subframe.rs:36-49
    if (!c) return;
    *c = ab_subframe_weight_config{1.0, 0.5, 1.0, 0.3, 8.0, 0.7, 5.0, 5};
}

int ab_analyze_subframes(ab_ctx *ctx, const ab_plane *images, size_t n, const ab_subframe_weight_config *config, ab_subframe_metrics *out) try {
    if (!ctx) return AB_ERR_INVALID;
    AB_CHECK(ctx, (images && out) || n == 0, "null argument");
    ab_subframe_weight_config c;
    ab_subframe_weight_config_default(&c);
    if (config) c = *config;
    for (size_t i = 0; i < n; ++i)
        AB_CHECK(ctx, images[i].data && images[i].rows > 0 && images[i].cols > 0, "subframe %zu is null or has a zero dimension", i);
    AB_HIP(ctx, hipSetDevice(ctx->device));
    // device-resident subframes of one size: their background tiles as a pipeline on the auxiliary stream, as in a registration
    // batch (detect.hip: ab_bg_pipeline_*), instead of one tile kernel inside every frame's chain
    ab_bg_pipeline pipe;
    // whatever path leaves this function, tile launches still in flight on the auxiliary stream (they read the caller's frames)
    // are drained first
    struct AuxDrain {
        ab_ctx *c;
        ab_bg_pipeline *p;
        ~AuxDrain() {
            if (p->on && c->aux_stream) (void)hipStreamSynchronize(c->aux_stream);
        }
    } aux_drain{ctx, &pipe};
    bool uniform = n >= 4;
    for (size_t i = 0; i < n && uniform; ++i)
        uniform = images[i].on_device && images[i].rows == images[0].rows && images[i].cols == images[0].cols;
    if (uniform) {
        std::vector<const float *> planes(n);
        for (size_t i = 0; i < n; ++i) planes[i] = (const float *)images[i].data;
        const std::vector<ab_pixel_xf> xf(n);  // no load transform: subframes are measured as they are
        AB_HIP(ctx, hipStreamSynchronize(ctx->stream));  // the caller's frames are complete before another stream reads them
        AB_TRY(ab_bg_pipeline_begin(ctx, planes.data(), n, images[0].rows, images[0].cols, xf.data(), 8, &pipe));
    }
    const int rc = ab_parallel_frames(ctx, n, "subframe", [&](ab_ctx *wc, size_t f) {
        double bg[2];
        if (pipe.on) AB_TRY(ab_bg_pipeline_get(wc, &pipe, f, bg));
        return analyze_one(wc, &images[f], c, &out[f], pipe.on ? bg : nullptr);
    });
    if (pipe.on) (void)hipStreamSynchronize(ctx->aux_stream);  // (an error or a cancel may leave tile launches in flight: they read the caller's frames)
    return rc;
} AB_CATCH(ctx)

int ab_analyze_subframe(ab_ctx *ctx, const ab_plane *image, const ab_subframe_weight_config *config, ab_subframe_metrics *out) try {
    return ab_analyze_subframes(ctx, image, 1, config, out);
} AB_CATCH(ctx)

void ab_normalize_subframe_weights(ab_subframe_metrics *metrics, size_t n) {  // subframe.rs:148-159
    if (!metrics) return;
    double max_w = 0.0;
    for (size_t i = 0; i < n; ++i) max_w = std::fmax(max_w, metrics[i].weight);
    if (max_w > 1e-15)
        for (size_t i = 0; i < n; ++i) metrics[i].weight /= max_w;
}

}  // extern "C"
