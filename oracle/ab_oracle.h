/*
 * ab_oracle.h -- CPU ORACLE (test infrastructure, NOT product code).
 *
 * A plain-C restatement of the AstroBurst reference's pixel-compute hot path
 * (Rust, src-tauri/src/core + src-tauri/src/math).  Every function cites the
 * reference file:line it follows.  Only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg may load this library; the product library
 * (astroburst_amd/csrc -> libastroburst_hip.so) never links or calls it.
 *
 * PARITY PINNING: the reference is Rust and cannot be built in this image
 * (no rustc/cargo, un-vendored crates), and it ships no golden vectors -- only
 * behavioural unit tests with loose tolerances.  The oracle is pinned against
 * every one of those unit tests for the path (tests/test_oracle_reference_cases.py
 * transcribes inputs + asserted tolerances).  For functions the reference does
 * not test at all (stats.rs, channel_blend.rs, masked_stretch.rs, star_mask.rs)
 * the oracle says "parity unpinned" next to the function.
 *
 * Arithmetic types follow the reference exactly (f32 vs f64 as cited).  Build
 * with -ffp-contract=off: Rust never fuses a*b+c.
 */
#ifndef AB_ORACLE_H
#define AB_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- math/median.rs ---------------------------------------------------- */
/* f32_cmp total order, NaN last (median.rs:4-13): returns <0, 0, >0 */
int orc_f32_cmp(float a, float b);
/* slice::select_nth_unstable_by(k, f32_cmp): after the call a[k] is the k-th
 * order statistic, everything before is <=, everything after is >=.  The
 * permutation of the rest is unspecified in Rust too. */
void orc_select_nth_f32(float *a, size_t n, size_t k);
double orc_exact_median_mut(float *data, size_t n);            /* median.rs:27-44 */
float orc_median_f32_mut(float *data, size_t n);               /* median.rs:46-63 */
float orc_exact_mad_mut(float *data, size_t n, float median);  /* median.rs:65-73 */
/* sigma_clip.rs:4-34; values is compacted in place, *n updated */
void orc_sigma_clipped_stats(float *values, size_t *n, float kappa, size_t iterations,
                             double *out_median, double *out_sigma);

/* ---- core/stacking/combine.rs ------------------------------------------ */
/* Summation-order modes for iterations >= 1 (combine.rs:50-60,90).  The
 * reference sums the survivors in whatever order select_nth_unstable left
 * them (unspecified).  ORC_ORDER_SELECT sums in the order our quickselect
 * leaves; ORC_ORDER_ASCENDING sorts the survivors ascending first -- a
 * function of the multiset only, which is what the HIP kernel reproduces
 * bit-for-bit.  Both agree to <= 1 ulp(f64) before the f32 cast. */
enum { ORC_ORDER_SELECT = 0, ORC_ORDER_ASCENDING = 1 };

/* combine.rs:14-92.  values is scratch (permuted/compacted). */
float orc_sigma_clip_combine(float *values, size_t n, float sigma_low, float sigma_high,
                             size_t max_iter, int order_mode, uint32_t *out_rejected);

/* combine.rs:94-193 with config.align == false (crop to min dims, gather
 * finite values per pixel in frame order, sigma_clip_combine).  planes[i] is
 * rows[i] x cols[i] row-major.  out is min_rows x min_cols.  Returns 0, or
 * -1 for "No images to stack" (combine.rs:98-100).  threads<=0 -> all cores. */
int orc_stack_images_noalign(const float *const *planes, const int64_t *rows, const int64_t *cols,
                             size_t n_images, float sigma_low, float sigma_high, size_t max_iter,
                             int order_mode, int threads, float *out, uint64_t *out_rejected,
                             int64_t *out_rows, int64_t *out_cols);

/* Two-level estimator used by the frame-sharded multi-GPU mode:
 * per-pixel kept-sum (f64) and kept-count of one shard, same clipping as
 * sigma_clip_combine.  Not a reference function; restated here so the N>1
 * path has a CPU checker. */
void orc_stack_partial_noalign(const float *const *planes, size_t n_images, int64_t npix,
                               float sigma_low, float sigma_high, size_t max_iter, int threads,
                               double *out_sum, uint32_t *out_cnt, uint64_t *out_rejected);

/* ---- core/imaging/sampling.rs, boundary.rs ------------------------------ */
double orc_catmull_rom(double t);                                        /* sampling.rs:4-14 */
size_t orc_clamp_index(int64_t idx, size_t len);                         /* boundary.rs:9-20 */
float orc_nearest_sample(const float *s, size_t rows, size_t cols, double y, double x);   /* :16-24 */
float orc_bilinear_sample(const float *s, size_t rows, size_t cols, double y, double x);  /* :26-49 */
float orc_bicubic_sample(const float *s, size_t rows, size_t cols, double y, double x);   /* :51-80 */

/* ---- core/stacking/align.rs, core/alignment/affine.rs ------------------- */
/* align.rs:36-57 */
void orc_shift_image_subpixel(const float *src, size_t rows, size_t cols, double dy, double dx,
                              int threads, float *out);
/* affine.rs:663-690; t = {a,b,tx,c,d,ty} maps OUTPUT (x,y) -> SOURCE (sx,sy) */
void orc_warp_image(const float *src, size_t src_rows, size_t src_cols, const double t[6],
                    size_t out_rows, size_t out_cols, int threads, float *out);

/* ---- core/imaging/stats.rs  (reference has NO tests here: parity unpinned,
 *      the restatement itself is the pin) --------------------------------- */
typedef struct {
    double min, max, median, mad, sigma, mean;
    uint64_t valid_count;
} orc_image_stats;

void orc_compute_image_stats(const float *data, size_t n, orc_image_stats *out);  /* stats.rs:15-23 */
void orc_compute_image_stats_exact(const float *data, size_t n, orc_image_stats *out); /* :43-73 */
void orc_compute_image_stats_hist(const float *data, size_t n, orc_image_stats *out);  /* :75-210 */
void orc_compute_image_stats_with_known_range(const float *data, size_t n, double known_min,
                                              double known_max, orc_image_stats *out); /* :25-41 */
/* stats.rs:378-421 build_histogram; bins out u32[bins]; returns 1 if range<1e-10 (all-zero bins) */
int orc_build_histogram(const float *data, size_t n, size_t bins, double dmin, double dmax,
                        uint32_t *out_bins);
/* intermediate histograms of the >4M path, exposed so the GPU histograms can
 * be checked bin-for-bin: pass 2 (value hist) for a given min/max */
void orc_stats_value_hist(const float *data, size_t n, double gmin, double gmax, uint64_t *hist65536,
                          double *out_sum, uint64_t *out_cnt);

/* ---- core/imaging/stf.rs ------------------------------------------------ */
typedef struct { double shadow, midtone, highlight; } orc_stf_params;
void orc_auto_stf(const orc_image_stats *st, double target_bg, double shadow_k, orc_stf_params *out); /* stf.rs:13-39 */
double orc_mtf(double x, double m);                                     /* stf.rs:50-58 */
double orc_mtf_balance(double m, double t);                             /* stf.rs:41-47 */
void orc_apply_stf_u8(const float *data, size_t n, const orc_stf_params *p, const orc_image_stats *st,
                      int threads, uint8_t *out);                        /* stf.rs:89-102 */
void orc_apply_stf_f32(const float *data, size_t n, const orc_stf_params *p, const orc_image_stats *st,
                       int threads, float *out);                         /* stf.rs:104-120,147-155 */

/* ---- colour / tone / calibration maps (orc_color.c) --------------------------------- */
/* scnr.rs:18-53; method 0 AverageNeutral, 1 MaximumNeutral */
void orc_apply_scnr_inplace(float *r, float *g, float *b, size_t n, int method, float amount, int preserve);
/* channel_blend.rs:13-70; weights = n_weights x {channel_idx, r, g, b} (f64).  parity unpinned */
void orc_blend_channels(const float *const *channels, size_t n_channels, const double *weights, size_t n_weights,
                        size_t npix, float *r_out, float *g_out, float *b_out);
void orc_spline_lut_from_points(const double *points_xy, size_t n, float *lut4096);  /* curves.rs:69-95 */
void orc_apply_curve(const float *data, size_t n, const float *lut4096, float *out); /* curves.rs:186-197 */
void orc_apply_levels(const float *data, size_t n, double black, double gamma, double white, float *out); /* :31-52 */
void orc_arcsinh_stretch_with_stats(const float *data, size_t n, float dmin, float dmax, float factor, float gamma,
                                    float *out);                                       /* stretch.rs:10-45 */
void orc_luminance(const float *r, const float *g, const float *b, size_t n, float *out); /* masked_stretch.rs:143-154 */
void orc_scale(const float *data, size_t n, float factor, float *out);                 /* cmd/compose/color.rs:28-40 */
void orc_calibrate_image(const float *raw, const float *bias, const float *dark, const float *flat, float dark_ratio,
                         size_t n, float *out);                                        /* calibration.rs:47-82 */
void orc_median_combine(const float *const *planes, size_t n_frames, size_t npix, float *out); /* :84-125 */

/* ---- core/alignment/phase_correlation.rs, downsample.rs (orc_phasecorr.c) ------------------ */
void orc_hann_periodic(size_t n, double *w);                                           /* window.rs:3-18 */
void orc_fft_twiddles(size_t n, double *tw_interleaved);            /* exp(-2 pi i k/n), k < n/2 */
void orc_fft2d(double *buf_interleaved, size_t rows, size_t cols, int inverse);        /* fft.rs:136-167 */
void orc_area_downsample(const float *src, size_t in_rows, size_t in_cols, size_t out_rows, size_t out_cols,
                         float *out);                                                  /* downsample.rs:6-46 */
void orc_phase_correlate(const float *reference, size_t ref_rows, size_t ref_cols, const float *target,
                         size_t tgt_rows, size_t tgt_cols, double *dx, double *dy, double *confidence); /* :22-89 */
void orc_correlate_single(const float *a, const float *b, size_t rows, size_t cols, double *dx, double *dy,
                          double *conf, double *surface);                              /* :105-141 */
/* combine.rs:94-193 with align == true */
int orc_stack_images_align(const float *const *planes, const int64_t *rows, const int64_t *cols, size_t n_images,
                           float sigma_low, float sigma_high, size_t max_iter, int order_mode, int threads, float *out,
                           uint64_t *out_rejected, int32_t *offsets_dy_dx);

/* ---- core/analysis/star_detection.rs, core/alignment/affine.rs (orc_detect.c, orc_affine.c) ---- */
typedef struct { /* DetectedStar, star_detection.rs:10-20 (+ discovery order for stable sorting) */
    double x, y, flux, fwhm, eccentricity, peak, snr;
    uint64_t npix, order;
} orc_star;
void orc_estimate_background(const float *image, size_t rows, size_t cols, size_t tile_size, double *out_median,
                             double *out_sigma);                                       /* :32-84 */
size_t orc_detect_stars(const float *image, size_t rows, size_t cols, double sigma_threshold, orc_star *out,
                        size_t cap, size_t *total, double *bg_median, double *bg_sigma); /* :86-258 */
int orc_normalize_for_detection(const float *image, size_t len, float *out);           /* affine.rs:24-53 */

typedef struct { /* AffineAlignResult, affine.rs:82-89; method 0 affine, 1 rigid, 2 phase_correlation, 3 identity */
    double t[6]; /* a, b, tx, c, d, ty */
    uint64_t matched_stars, inliers;
    double residual_px;
    int32_t method;
} orc_affine_result;
/* align_channel_affine (affine.rs:129-212).  num_threads pins rayon::current_num_threads()
 * (RANSAC chunking and seeds depend on it, :410-416); vote ties are broken by (ref idx, tgt idx)
 * ascending (the reference iterates a HashMap: unspecified). */
void orc_align_channel_affine(const float *reference, const float *target, size_t rows, size_t cols, int num_threads,
                              orc_affine_result *out);
/* the star-list half of it (triangles -> votes -> RANSAC -> sanity), on given centroids (x, y pairs) */
int orc_affine_from_stars(const double *ref_xy, size_t n_ref, const double *tgt_xy, size_t n_tgt, size_t rows, size_t cols,
                          int num_threads, orc_affine_result *out);   /* returns 0 if it fell through to phase correlation */
int orc_fit_rigid(const double *matches, size_t n, double t[6]);                       /* affine.rs:597-642 */
int orc_fit_affine(const double *matches, size_t n, double t[6]);                      /* affine.rs:519-536 */

/* ---- core/imaging/background.rs (orc_background.c) ------------------------------------------ */
/* extract_background (:55-116).  Returns 0 ok; 1 "Image too small for grid_size"; 2 "Not enough background
 * samples"; 3 singular fit.  mode 0 subtract, 1 divide; model / corrected may be NULL; coeffs_out: 21 doubles. */
int orc_extract_background(const float *image, size_t rows, size_t cols, size_t grid, size_t degree, float sigma_clip,
                           size_t iterations, int mode, float *model, float *corrected, size_t *sample_count_out,
                           double *rms_out, double *coeffs_out);

/* ---- core/imaging/star_mask.rs, masked_stretch.rs (orc_masked.c) -------------------------------- */
/* generate_star_mask_from_detection (star_mask.rs:46-138) on the stars' (x, y, fwhm); returns stars_masked */
size_t orc_star_mask_from_stars(const float *image, size_t h, size_t w, const double *xs, const double *ys,
                                const double *fwhms, size_t n_stars, double growth_factor, double softness,
                                double min_fwhm, double max_fwhm, int luminance_protect, double luminance_ceiling,
                                float *mask, double *coverage_out);
size_t orc_generate_star_mask(const float *image, size_t h, size_t w, double growth_factor, double softness,
                              double detection_sigma, double min_fwhm, double max_fwhm, int luminance_protect,
                              double luminance_ceiling, float *mask, double *coverage_out);      /* :38-44 */
void orc_masked_stretch_with_mask(const float *image, const float *mask, size_t n, size_t iterations, double target_bg,
                                  double protection_amount, double convergence_threshold, float *out,
                                  size_t *iterations_run_out, double *final_bg_out, int *converged_out); /* :60-118 */

/* ---- core/compose/rgb.rs, white_balance.rs, core/imaging/resample.rs (orc_compose.c) ----------- */
int orc_resample_image(const float *src, size_t src_rows, size_t src_cols, size_t target_rows, size_t target_cols,
                       float *out);                                                  /* resample.rs:25-61 */
void orc_select_wb_reference(const orc_image_stats *sr, const orc_image_stats *sg, const orc_image_stats *sb,
                             double out[3]);                                         /* white_balance.rs:3-20 */
void orc_compose_apply_stf_inplace(float *data, size_t n, const orc_stf_params *p, const orc_image_stats *st); /* rgb.rs:191-207 */
typedef struct { /* RgbComposeConfig, types/compose.rs:47-75 */
    int32_t white_balance;          /* 0 Auto, 1 Manual(wb_manual), 2 None */
    double wb_manual[3];
    int32_t auto_stretch, linked_stf;
    int32_t has_stf[3];
    orc_stf_params stf[3];
    int32_t align, align_method;    /* AlignMethod: 0 PhaseCorrelation, 1 Affine */
    int32_t has_scnr, scnr_method;
    float scnr_amount;
    int32_t scnr_preserve;
    int32_t num_threads;            /* pins rayon's worker count for the affine RANSAC */
} orc_rgb_config;
typedef struct { /* scalars of ProcessedRgb, rgb.rs:18-40 */
    uint64_t rows, cols;
    orc_stf_params stf[3];
    double chan_stats[3][4];        /* ChannelStats {min, max, median, mean} before white balance */
    double offset_g[2], offset_b[2];
    int32_t scnr_applied, resampled;
    orc_image_stats stats_wb[3];
} orc_rgb_result;
int orc_process_rgb(const float *r, size_t r_rows, size_t r_cols, const float *g, size_t g_rows, size_t g_cols,
                    const float *b, size_t b_rows, size_t b_cols, const orc_rgb_config *cfg, float *out_r, float *out_g,
                    float *out_b, float *pre_r, float *pre_g, float *pre_b, orc_rgb_result *res, char *err,
                    size_t err_cap);                                                 /* rgb.rs:209-323 */

/* ---- core/astrometry/spcc.rs (orc_spcc.c; the reference has no tests here: parity unpinned) ------- */
typedef struct { /* SpccConfig, spcc.rs:9-28 (defaults 20.0, 200, 0.90, AverageSpiral) */
    double min_snr;
    uint64_t max_stars;
    double saturation_limit;
    int32_t white_reference;   /* 0 AverageSpiral, 1 G2V, 2 Photopic, 3 Custom(custom) */
    double custom[3];
} orc_spcc_config;
typedef struct { /* SpccResult numbers, spcc.rs:45-56 */
    double r_factor, g_factor, b_factor;
    uint64_t stars_matched, stars_total;
    double avg_color_index;
} orc_spcc_result;
void orc_spcc_white_reference_rgb(int kind, const double custom[3], double out[3]);                     /* :245-255 */
double orc_aperture_flux_f32(const float *image, size_t h, size_t w, double x, double y, double radius); /* :341-383 */
int orc_spcc_from_detection(const float *r, const float *g, const float *b, size_t h, size_t w, const orc_star *stars,
                            size_t n_stars, double lum_max, double pixel_scale_arcsec, const orc_spcc_config *cfg,
                            orc_spcc_result *res);                                                       /* :90-183 */
int orc_spcc_calibrate_rgb(const float *r, const float *g, const float *b, size_t h, size_t w, double pixel_scale_arcsec,
                           const orc_spcc_config *cfg, orc_spcc_result *res);                            /* :73-183 */

/* ---- small caller-side helpers (orc_extras.c) ------------------------------------------------------ */
void orc_apply_lrgb(const float *l, float *r, float *g, float *b, size_t n, float lightness_weight,
                    float chrominance_weight);                                         /* lrgb.rs:4-45 */
void orc_synthesize_luminance(const float *r, const float *g, const float *b, size_t n, float *out); /* lrgb.rs:47-64 */
void orc_compute_linked_stf(const orc_image_stats *sr, const orc_image_stats *sg, const orc_image_stats *sb, double target_bg,
                            double shadow_k, orc_stf_params *stf, orc_image_stats *combined); /* cmd/helpers.rs:185-202 */
void orc_calibrate_channel(const float *orig, size_t n, float factor, const orc_image_stats *orig_stats, float *out,
                           orc_image_stats *stats);                                    /* cmd/compose/color.rs:21-49 */
void orc_create_master(int kind, const float *const *frames, size_t n_frames, size_t npix, const float *master_bias,
                       const float *master_dark, float *out);                          /* calibration.rs:127-255 */

/* ---- infra/fits pixel codecs (orc_fits.c), SURVEY 8(f) row 1 ------------------------------------ */
size_t orc_fits_decode_pixels(const uint8_t *data, size_t nbytes, int64_t bitpix, double bscale, double bzero,
                              float *out);                                             /* reader.rs:42-101 */
void orc_fits_compute_bzero_bscale(const float *data, size_t n, double *bzero, double *bscale); /* writer.rs:143-159 */
size_t orc_fits_encode_pixels(const float *data, size_t n, int32_t bitpix, double bzero, double bscale,
                              uint8_t *out);                                           /* writer.rs:82-135 */

/* ---- core/analysis/subframe.rs (orc_subframe.c), SURVEY 8(f) row 3 -------------------------------- */
typedef struct { /* SubframeWeightConfig, subframe.rs:24-49 (defaults 1.0, 0.5, 1.0, 0.3, 8.0, 0.7, 5.0, 5) */
    double fwhm_weight, eccentricity_weight, snr_weight, noise_weight, max_fwhm, max_eccentricity, min_snr;
    uint64_t min_stars;
} orc_subframe_config;
typedef struct { /* numbers of SubframeMetrics, subframe.rs:9-22 */
    uint64_t star_count;
    double median_fwhm, median_eccentricity, median_snr, background_median, background_sigma, noise_ratio, weight;
    int32_t accepted;
} orc_subframe_metrics;
double orc_subframe_compute_weight(double fwhm, double ecc, double snr, double noise, const orc_subframe_config *c); /* :123-146 */
void orc_subframe_from_detection(const orc_star *stars, size_t n, double bg_median, double bg_sigma, const orc_subframe_config *c,
                                 orc_subframe_metrics *out);                           /* :62-120 */
void orc_analyze_subframe(const float *image, size_t rows, size_t cols, const orc_subframe_config *c,
                          orc_subframe_metrics *out);                                  /* :51-121 */
void orc_subframe_normalize_weights(orc_subframe_metrics *m, size_t n);              /* :148-159 */

/* ---- preview / tile renderers up to the PNG encoder (orc_render.c), SURVEY 8(f) row 4 ------------------ */
typedef struct { /* TileLevel, tiles.rs:21-29, + the byte offset of the level's first tile in the packed buffer */
    uint64_t level, width, height, cols, rows;
    double scale_factor;
    uint64_t offset;
} orc_tile_level;
void orc_preview_dims(size_t rows, size_t cols, size_t max_dim, size_t *ph, size_t *pw);       /* helpers.rs:283-290 */
void orc_render_rgb_preview(const float *r, const float *g, const float *b, size_t rows, size_t cols, size_t max_dim,
                            const orc_stf_params *stf, const orc_image_stats *stats, uint8_t *out); /* helpers.rs:204-322 */
size_t orc_ipc_encode_with_header(const float *arr, size_t rows, size_t cols, size_t max_dim, uint8_t *out); /* ipc.rs:36-148 */
size_t orc_tile_compute_num_levels(size_t width, size_t height, size_t tile_size);             /* tiles.rs:137-147 */
void orc_tile_downsample_2x(const float *src, size_t rows, size_t cols, float *out);           /* tiles.rs:41-70 */
void orc_tile_percentile_bounds(const float *slice, size_t n, double low_pct, double high_pct, float *lo, float *hi); /* :149-178 */
void orc_render_tile(const float *src, size_t rows, size_t cols, size_t tx, size_t ty, size_t ts, float gmin, float gmax,
                     uint8_t *buf);                                                            /* :72-113 */
void orc_render_tile_rgb(const float *r, const float *g, const float *b, size_t rows, size_t cols, size_t tx, size_t ty, size_t ts,
                         const orc_stf_params *stf, const orc_image_stats *stats, uint8_t *buf); /* :257-341 */
size_t orc_tile_pyramid_layout(size_t rows, size_t cols, size_t ts, size_t channels, orc_tile_level *levels, size_t *num_levels);
void orc_generate_tile_pyramid(const float *normalized, size_t rows, size_t cols, size_t ts, uint8_t *tiles, orc_tile_level *levels,
                               size_t *num_levels, float *gmin_out, float *gmax_out);          /* :180-255 */
void orc_generate_tile_pyramid_rgb(const float *r, const float *g, const float *b, size_t rows, size_t cols, size_t ts,
                                   const orc_stf_params *stf, const orc_image_stats *stats, uint8_t *tiles, orc_tile_level *levels,
                                   size_t *num_levels);                                        /* :383-481 */

/* ---- core/imaging/calibration_pipeline.rs (orc_batch.c); the reference has no tests here ---- */
void orc_calibrate_light(const float *light, size_t npix, const float *bias, size_t bias_len, const float *dark, size_t dark_len,
                         const float *flat, size_t flat_len, float *out);                      /* :74-118 */
void orc_normalize_frame(const float *frame, size_t npix, float *out);                         /* :309-319 */
void orc_sigma_clipped_mean_stack(const float *const *frames, size_t n, size_t npix, float sigma_low, float sigma_high, size_t max_iter,
                                  float *out, uint64_t *rejection_counts);                     /* :321-378 */
void orc_run_batch_channel(const float *const *lights, size_t n, size_t npix, const float *bias, size_t bias_len, const float *dark,
                           size_t dark_len, const float *flat, size_t flat_len, float sigma_low, float sigma_high, size_t max_iter,
                           int normalize, float *out, uint64_t *rejection_counts, double *mean_out, double *stddev_out); /* :157-190 */
void orc_normalize_channel(const float *ch, size_t rows, size_t cols, size_t ld, float *out);  /* :291-307 */
void orc_compose_rgb_from_masters(const float *r, size_t r_rows, size_t r_cols, const float *g, size_t g_rows, size_t g_cols,
                                  const float *b, size_t b_rows, size_t b_cols, const float *l, size_t l_rows, size_t l_cols,
                                  float *out, size_t *out_rows, size_t *out_cols);             /* :201-289 */

/* utility */
int orc_max_threads(void);

#ifdef __cplusplus
}
#endif
#endif
