/*
 * astroburst_hip.h -- C ABI of libastroburst_hip.so, the MI355X (gfx950) implementation of
 * AstroBurst's pixel-compute hot path.
 *
 * The reference (Rust, src-tauri/) has no FFI seam of its own; the replaceable edge is the
 * internal call `cmd::* -> core::*`.  Every entry point below replaces one
 * `core::*` function and cites it (paths relative to src-tauri/src/).  INTEGRATION.md shows the
 * Rust `extern "C"` block and the safe wrappers a maintainer would add.
 *
 * Conventions
 *  - plain C types only; all functions return an ab_status (0 = ok) and never abort/throw
 *    (the app is built panic="abort", Cargo.toml:64).  ab_last_error(ctx) returns the message,
 *    worded like the reference's anyhow strings where one exists (e.g. "No images to stack").
 *  - planes are contiguous row-major f32, dims (rows, cols), exactly ndarray::Array2<f32>'s
 *    standard layout (`.as_slice().expect("contiguous")`, combine.rs:152-155).
 *  - an ab_plane may point at HOST memory (on_device = 0: staged through HBM by the library)
 *    or at DEVICE memory (on_device = 1: used in place, nothing is copied).  Inputs are
 *    borrowed and never modified; outputs are caller-allocated.
 *  - a context owns one HIP stream (or borrows the caller's, ab_ctx_set_stream) and a scratch
 *    arena; it is not shared between threads -- create one per calling thread (the Tauri
 *    commands run concurrently on tokio blocking threads, cmd/common.rs:345-352).
 *  - device-plane calls are asynchronous on the context's stream unless they return scalars.
 *
 * Environment
 *  The library reads these eight variables and no others (each when a context / communicator is created, AB_TRACE per call).
 *  None changes a result: every engine they select is held to the same oracle by the tests.
 *    AB_TRACE=1              stage stamps, deferred-pixel and redone-frame notes on stderr (also fills AB_FB_STACK_GENERAL_PIXELS)
 *    AB_STACK_EXACT=1        ab_stack_*: the direct two-pass clipping engine instead of the running-sum one (bit-exact cross-check
 *                            of the default engine's 1e-5 contract; ~1.3x slower)
 *    AB_STATS_CHAIN=1        ab_compute_image_stats*: the histogram chain instead of the register-resident kernel (identical results;
 *                            what a resident launch falls back to by itself when its grid barrier times out)
 *    AB_REGISTER_WORKERS=n   host threads (= frame groups in flight) of ab_register_frames / ab_align_pairs_affine (default 12)
 *    AB_STACK_DEEP_FROM=n    stacks of more than n frames (64 <= n <= 4096, default 4096) take the workgroup-per-pixel kernel
 *    AB_BATCH_DEEP_FROM=n    the same for ab_sigma_clipped_mean_stack (64 <= n <= 2048, default 2048)
 *    AB_COMM_TIMEOUT_MS=n    how long a rank waits on a collective (either transport) before AB_ERR_COMM (default 300000)
 *    AB_COMM_HOST_SLOT_MB=n  host-staged communicator: size of a rank's shared-memory slot, 1 .. 1024 (default 4)
 *  Superseded kernel forms, sweep knobs, stage cuts and fault injection exist only in the developer build
 *  (`make -C astroburst_amd/csrc dev` -> libastroburst_hip_dev.so, ab_version() ends in "+dev"); the release library does not
 *  contain their names (tests/test_abi_cpu.py checks both lists against the sources and the built library).
 */
#ifndef ASTROBURST_HIP_H
#define ASTROBURST_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define AB_API __attribute__((visibility("default")))

typedef enum {
    AB_OK = 0,
    AB_ERR_INVALID = 1,     /* bad argument (message says which) */
    AB_ERR_HIP = 2,         /* HIP runtime error (message carries hipGetErrorString) */
    AB_ERR_NO_DEVICE = 3,   /* no gfx950 device / wrong architecture */
    AB_ERR_UNSUPPORTED = 4, /* valid request this build cannot serve yet */
    AB_ERR_NOMEM = 5,
    AB_ERR_COMM = 6,      /* RCCL unavailable or a collective failed (message carries ncclGetErrorString) */
    AB_ERR_CANCELLED = 7  /* the host set the context's cancel flag (AppError::Cancelled, core/imaging/background.rs:80-82) */
} ab_status;

typedef struct ab_ctx ab_ctx;
typedef struct ab_comm ab_comm; /* one RCCL rank bound to a context's GPU: section (e) at the end of this header */

typedef struct {
    const float *data;
    int64_t rows, cols;
    int32_t on_device; /* 0 host, 1 device (HBM) */
} ab_plane;

typedef struct {
    float *data;
    int64_t rows, cols;
    int32_t on_device;
} ab_plane_mut;

/* ---- context / memory ------------------------------------------------------------------- */
AB_API int ab_ctx_create(int device_id, ab_ctx **out);
AB_API void ab_ctx_destroy(ab_ctx *ctx);
AB_API const char *ab_last_error(const ab_ctx *ctx);
AB_API const char *ab_version(void);
/* borrow an existing hipStream_t (e.g. PyTorch's current stream).  A NULL handle means HIP's
 * default stream, which is what PyTorch runs on unless told otherwise. */
AB_API int ab_ctx_set_stream(ab_ctx *ctx, void *hip_stream);
/* go back to the context's own (non-blocking) stream */
AB_API int ab_ctx_reset_stream(ab_ctx *ctx);
AB_API void *ab_ctx_get_stream(ab_ctx *ctx);
AB_API int ab_ctx_synchronize(ab_ctx *ctx);
AB_API int ab_device_alloc(ab_ctx *ctx, size_t bytes, void **out_dptr);
AB_API int ab_device_free(ab_ctx *ctx, void *dptr);
AB_API int ab_upload(ab_ctx *ctx, void *dst_device, const void *src_host, size_t bytes);
AB_API int ab_download(ab_ctx *ctx, void *dst_host, const void *src_device, size_t bytes);
/* Progress / cancel (infra/progress.rs:39-74; taken as Option<&ProgressHandle> by core/imaging/background.rs:55-59).
 * cb(stage, current, total, user) is called on the calling thread at the reference's own stage boundaries with its stage
 * strings ("sampling background", "fitting polynomial surface", "generating model", "applying correction") and once per
 * frame by the frame loops (stage "registration" for ab_register_frames / ab_align_pairs_affine, "subframe" for
 * ab_analyze_subframes); NULL removes it.  ab_ctx_request_cancel may be called from any
 * thread: the next stage boundary returns AB_ERR_CANCELLED ("Operation cancelled") and leaves the flag set until
 * ab_ctx_clear_cancel. */
typedef void (*ab_progress_cb)(const char *stage, uint64_t current, uint64_t total, void *user);
AB_API int ab_ctx_set_progress_cb(ab_ctx *ctx, ab_progress_cb cb, void *user);
AB_API int ab_ctx_request_cancel(ab_ctx *ctx);
AB_API int ab_ctx_clear_cancel(ab_ctx *ctx);
/* device properties the bench prints: name, CU count, HBM bytes */
AB_API int ab_device_info(ab_ctx *ctx, char *name, size_t name_cap, int *cu_count, uint64_t *hbm_bytes);
/* Release what the context has grown for its calls: the scratch arena, every workspace, the HBM staging area of host-resident
 * frames (as large as the largest frame set it was given), the pinned read-back buffers and the declined-tile lists -- its frame
 * workers' too.  The context stays usable: the next call allocates what it needs again.  Blocks until the context's streams are
 * idle.  Every buffer is released whatever an earlier release returned; the first failure is the return code, its text in
 * ab_last_error(ctx).  A long-lived host (one ab_ctx per command thread, infra/cache.rs keeps the planes) calls it after a large
 * batch.  (ab_stack_images(align) gives its registered copies back by itself when they exceed 1 GiB.) */
AB_API int ab_ctx_trim(ab_ctx *ctx);
/* Every fast path of the library has an exact fallback that gives the SAME result (a frame redone through the full component list,
 * a background tile settled by the resident kernel, a statistics launch repeated as the histogram chain).  They are silent in the
 * results and not in the run time, so the context counts them (its frame workers' too): out[k] = events of kind k since the
 * context was created (or since the last call with reset != 0), for k < min(cap, AB_FB_COUNT).  Thread-safe. */
typedef enum {
    AB_FB_FRAMES_REDONE = 0,  /* detect_stars of a registration frame redone through the full path (any of the four reasons below) */
    AB_FB_TILE_SLOTS,         /* ... a 32 x 128 labelling tile held more components than its record slots */
    AB_FB_COMPONENT_TABLE,    /* ... more components than the chained table holds */
    AB_FB_SELECTION_SHORT,    /* ... the brightest 480 components did not yield the matcher's 120 stars and fainter ones exist */
    AB_FB_SELECTION_CUT,      /* ... the faintest kept star lay within two key steps of the selection's cut */
    AB_FB_TILES_DECLINED,     /* background tiles the streaming kernel declined (settled by the resident tile kernel) */
    AB_FB_STATS_CHAIN,        /* compute_image_stats launches whose resident kernel aborted and were repeated as the chain */
    AB_FB_STACK_GENERAL_PIXELS, /* pixels the stack's fast pass handed to its general pass (counted only under AB_TRACE=1) */
    AB_FB_LABEL_TILES_DENSE,  /* 256 x 256 labelling tiles of registration frames read from the frame: no usable candidate list from the tile pass
                                 (a partial tile, a tile brighter than the frame's threshold, an overflowed list, a tile the streaming kernel declined) */
    AB_FB_COUNT
} ab_fallback_kind;
AB_API int ab_ctx_fallback_counts(ab_ctx *ctx, uint64_t *out, size_t cap, int reset);

/* ---- a1/a2  core/stacking/combine.rs ------------------------------------------------------ */
/* StackConfig, types/stacking.rs:3-20 (defaults 3.0, 3.0, 5, align=true) */
typedef struct {
    float sigma_low;
    float sigma_high;
    uint32_t max_iterations;
    int32_t align; /* bool */
} ab_stack_config;

/* The per-pixel kernel of stack_images (combine.rs:160-182 + sigma_clip_combine :14-92):
 * out[y][x] = sigma_clip_combine({planes[k][y][x] finite}, sigma_low, sigma_high, max_iter).
 * All planes must be at least out->rows x out->cols; each is read with its own row stride
 * (= its cols), i.e. the reference's top-left crop to the minimum dims (combine.rs:104-113)
 * costs nothing.  *out_rejected receives StackResult.rejected_pixels (sum of per-pixel
 * rejection counts, combine.rs:158,181).  Any frame count the reference takes (it gathers a Vec per pixel: no limit; here up to
 * 2^24 per call): up to 256 contiguous frames a pixel's samples live in one lane's registers (64: the HBM-bound kernel; 65 .. 128:
 * ~4 ms and 129 .. 256: ~10-16 ms for 4096^2), 257 .. 512 in two lanes', up to 4096 -- or with ragged strides -- in one wave's
 * (stack_wide.hip), beyond that a workgroup sorts a pixel's samples in a scratch segment (stack_deep.hip): same results, slower.
 * Floating-point contract.  The reference's result depends on the order select_nth_unstable leaves the survivors in (SURVEY.md 7
 * hard part 2); the definition here: the f64 sums of iterations >= 1 run over the survivors in ascending value order.
 *   - DEFAULT ENGINE (2 .. 64 frames): within 1e-5 relative of that definition on every pixel -- north_star's tolerance.  Its
 *     iterations >= 1 take mean and variance from running sums about the median (E = sum(x - c0), Q = sum(x - c0)^2 in f64: mean =
 *     (n c0 + E) / n, sum(x - mean)^2 = Q - n (mean - c0)^2), a few ulp(f64) away from the two-pass sums; an ulp can flip the
 *     clipping decision of a sample that sits exactly on a bound.  MEASURED bit-identical: 0 of 16.7 M pixels of the bench stack
 *     differ from the oracle; the GPU tests allow at most 1e-4 of the pixels to differ at all (tests/test_gpu_stack.py:28-39).
 *   - AB_STACK_EXACT=1 when the context is created: the direct two-pass engine, bit-identical BY CONSTRUCTION (the tests run both).
 *   - more than 64 frames, median combine, partial sums: always bit-identical by construction. */
AB_API int ab_stack_sigma_clip(ab_ctx *ctx, const ab_plane *planes, size_t n, const ab_stack_config *cfg,
                               ab_plane_mut *out, uint64_t *out_rejected);

/* Passing out_rejected = NULL to ab_stack_sigma_clip keeps the call fully asynchronous on the
 * context's stream; the count of that last stack can be fetched later (synchronises). */
AB_API int ab_stack_last_rejected(ab_ctx *ctx, uint64_t *out_rejected);
/* milliseconds between the HIP events the library records on the context's stream immediately before the first and
 * after the last stack kernel of the most recent multi-frame ab_stack_* call (blocks until that call has finished):
 * the kernels' own duration, free of the caller's launch overhead (bench.py's roofline) */
AB_API int ab_stack_last_kernel_ms(ab_ctx *ctx, float *out_ms);

/* stack_images (combine.rs:94-193): crops to the minimum dims, optionally registers frames
 * 1..n-1 on frame 0 (PhaseCorrelation, combine.rs:126-138), then sigma-clip combines.
 * out must be min_rows x min_cols.  offsets (nullable) receives n (dy, dx) pairs rounded to
 * i32 as the reference reports them (combine.rs:135-137).  Errors: n == 0 -> AB_ERR_INVALID
 * "No images to stack" (combine.rs:98-100). */
AB_API int ab_stack_images(ab_ctx *ctx, const ab_plane *planes, size_t n, const ab_stack_config *cfg,
                           ab_plane_mut *out, int32_t *offsets_dy_dx, uint64_t *out_rejected);

/* Frame-sharded partial: per pixel the f64 sum and the u32 count of
 * the survivors of THIS shard's frames; ranks all-reduce (sum, count) and divide.  This is a
 * two-level estimator, not the reference's single-level one; its CPU checker is
 * orc_stack_partial_noalign.  out_sum/out_cnt are device pointers of rows*cols elements. */
AB_API int ab_stack_sigma_clip_partial(ab_ctx *ctx, const ab_plane *planes, size_t n,
                                       const ab_stack_config *cfg, int64_t rows, int64_t cols,
                                       double *out_sum_dev, uint32_t *out_cnt_dev, uint64_t *out_rejected);
/* out[i] = cnt[i] ? (float)(sum[i] / cnt[i]) : 0  -- the divide after the all-reduce */
AB_API int ab_stack_finalize_partial(ab_ctx *ctx, const double *sum_dev, const uint32_t *cnt_dev, int64_t n,
                                     float *out_dev);

/* ---- a3/a5  core/stacking/align.rs:36-57, core/alignment/affine.rs:663-690 ---------------- */
/* shift_image_subpixel(image, dy, dx): out(y,x) = bicubic(src, y+dy, x+dx), 0.0 where the
 * sample centre leaves [-0.5, dim-0.5]; |dy|,|dx| < 1e-12 is a copy. */
AB_API int ab_shift_image_subpixel(ab_ctx *ctx, const ab_plane *src, double dy, double dx, ab_plane_mut *out);
/* warp_image(image, &AffineTransform{a,b,tx,c,d,ty}, out_rows, out_cols): the transform maps
 * OUTPUT (x,y) to SOURCE (sx,sy) (affine.rs:74-80); 0.0 outside 0<=sx<cols-1, 0<=sy<rows-1. */
AB_API int ab_warp_image(ab_ctx *ctx, const ab_plane *src, const double transform[6], ab_plane_mut *out);

/* rows [row0, row0 + out_band->rows) of warp_image(src, transform, out_rows, out_band->cols): the band of a row-sharded
 * registration; coordinates are those of the whole output, so the band equals the same rows of ab_warp_image. */
AB_API int ab_warp_image_rows(ab_ctx *ctx, const ab_plane *src, const double transform[6], int64_t out_rows, int64_t row0,
                              ab_plane_mut *out_band);
/* The same with the SOURCE given as a band (a rank of the row-band scheme ingests its rows of every frame + a
 * halo, not the frame set): src_band = rows [src_row0, src_row0 + src_band->rows) of a frame of src_rows rows.  The band must
 * cover ab_warp_source_rows of the request (AB_ERR_INVALID names the rows otherwise); the result equals the same rows of
 * ab_warp_image on the whole frame bit for bit.  Device planes. */
AB_API int ab_warp_image_rows_from_band(ab_ctx *ctx, const ab_plane *src_band, int64_t src_row0, int64_t src_rows, const double transform[6],
                                        int64_t out_rows, int64_t row0, ab_plane_mut *out_band);
/* source rows [*src_row0, *src_row0 + *src_nrows) that rows [row0, row0 + nrows) of warp_image(src, transform, .., out_cols) read
 * from a src_rows x src_cols frame (affine.rs:663-690: sy = c x + d y + ty at the band's corners; sampling.rs:48-80: rows
 * floor(sy) - 1 .. floor(sy) + 2, clamped): exact, not a bound with a margin; 0 rows when the band maps outside the frame */
AB_API int ab_warp_source_rows(const double transform[6], int64_t src_rows, int64_t src_cols, int64_t out_cols, int64_t row0, int64_t nrows,
                               int64_t *src_row0, int64_t *src_nrows);

/* ---- a8  core/alignment/phase_correlation.rs ------------------------------------------------- */
typedef struct { double dx, dy, confidence; } ab_phase_correlation_result; /* PhaseCorrelationResult, :15-20 */
/* phase_correlate(reference, target) (phase_correlation.rs:22-89): crops both to the common dims,
 * returns (0, 0, 0) for constant / nearly empty images, correlates directly up to 512 x 512 and
 * coarse (area-averaged 512 x 512) + fine (centred 512 x 512 crop) above.  The FFT arithmetic is
 * ours (the reference uses rustfft); the reference's own tests pin the shift to +-1 px. */
AB_API int ab_phase_correlate(ab_ctx *ctx, const ab_plane *reference, const ab_plane *target,
                              ab_phase_correlation_result *out);
/* correlate_single (phase_correlation.rs:105-141) for two equal planes of at most 512 x 512;
 * surface_host (nullable) receives the fft_rows x fft_cols correlation surface (parity tests) */
AB_API int ab_correlate_single(ab_ctx *ctx, const ab_plane *a, const ab_plane *b, ab_phase_correlation_result *out,
                               double *surface_host);

/* ---- a7  core/analysis/star_detection.rs ------------------------------------------------------ */
typedef struct { /* DetectedStar, star_detection.rs:10-20 */
    double x, y, flux, fwhm, eccentricity, peak, snr;
    uint64_t npix;
} ab_detected_star;
/* estimate_background (star_detection.rs:32-84): per-tile sigma-clipped median / sigma, then the
 * [len/2] element of the sorted tile medians and sigmas; (0, 1) if no tile has 8 valid pixels */
AB_API int ab_estimate_background(ab_ctx *ctx, const ab_plane *img, int64_t tile_size, double *out_median,
                                  double *out_sigma);
/* the tile map behind estimate_background (star_detection.rs:36-68): tiles in row-major order, step = max(tile_size, 16);
 * per tile sigma_clipped_stats(valid pixels, 3.0, 2) = (median, sigma) and valid = the tile had >= 8 valid pixels.
 * Writes at most cap tiles; *out_tiles = tile count, *out_tiles_x = tiles per row. */
AB_API int ab_background_tile_stats(ab_ctx *ctx, const ab_plane *img, int64_t tile_size, double *out_median, double *out_sigma,
                                    int32_t *out_valid, size_t cap, size_t *out_tiles, size_t *out_tiles_x);
/* detect_stars(image, sigma) (star_detection.rs:86-258): threshold at median + sigma * bg_sigma,
 * 8-connected components seeded from interior pixels, 3 <= npix <= 5000, flux-weighted moments,
 * FWHM in [0.5, 30], sorted by flux (descending), deduplicated within 3 px.  Writes at most cap
 * stars; *out_total (nullable) is the number found.  Segmentation (component set, npix, order)
 * is exact; the f64 moments are summed in raster instead of BFS order (~1e-15 relative). */
AB_API int ab_detect_stars(ab_ctx *ctx, const ab_plane *img, double sigma_threshold, ab_detected_star *out, size_t cap,
                           size_t *out_count, size_t *out_total, double *bg_median, double *bg_sigma);

/* ---- a6  core/alignment/affine.rs -------------------------------------------------------------- */
typedef struct { /* AffineAlignResult, affine.rs:82-89 */
    double transform[6]; /* AffineTransform a, b, tx, c, d, ty: output (x, y) -> source (affine.rs:55-80) */
    uint64_t matched_stars, inliers;
    double residual_px;
    int32_t method; /* 0 affine, 1 rigid, 2 phase_correlation, 3 identity (AffineAlignMethod, :91-97) */
} ab_affine_align_result;
/* normalize_for_detection (affine.rs:24-53): clamp((v - p1) / (p99.9 - p1), 0, 1) on a <= 100 k subsample's percentiles */
AB_API int ab_normalize_for_detection(ab_ctx *ctx, const ab_plane *img, ab_plane_mut *out);
/* align_channel_affine(reference, target) (affine.rs:129-212): normalise, detect at 3.5 sigma, top 120
 * stars, triangle voting, RANSAC affine then rigid (2000 draws, 3 px inliers), sanity limits, else
 * phase correlation (confidence >= 1.5), else identity.  num_threads pins what the reference takes
 * from rayon::current_num_threads() (the draws are partitioned and seeded per worker, :410-416);
 * vote ties are broken by (ref, tgt) index (the reference iterates a HashMap). */
AB_API int ab_align_channel_affine(ab_ctx *ctx, const ab_plane *reference, const ab_plane *target, int num_threads,
                                   ab_affine_align_result *out);
/* align_channel_affine(reference, targets[i]) for i < n in one call: the loop a stacking / compose caller runs
 * (cmd/compose/blend.rs:226-232, rgb.rs:165-189).  The reference frame's normalisation, detection and triangle
 * table are computed once; each out[i] equals what ab_align_channel_affine(reference, targets[i]) returns. */
AB_API int ab_register_frames(ab_ctx *ctx, const ab_plane *reference, const ab_plane *targets, size_t n, int num_threads,
                              ab_affine_align_result *out);
/* align_pair(reference, targets[i], AlignMethod::Affine) for i < n (core/alignment/pair.rs:41-77): the estimate above
 * followed by warp_image(target, transform) into aligned[i] (device planes); each worker warps its frame right after
 * estimating it, so the warps overlap the other frames' detection.  The reference and any target may be HOST planes
 * (on_device = 0), as the application holds its frames (core/stacking/calibration.rs:306-315, infra/cache.rs:306-310):
 * the library copies them into HBM on its own stream, one event per frame, and registers each frame as it lands -- the
 * call costs the upload plus one group's registration, not upload + registration.  Pinned host memory keeps the copies
 * asynchronous; results do not depend on where a frame started. */
AB_API int ab_align_pairs_affine(ab_ctx *ctx, const ab_plane *reference, const ab_plane *targets, size_t n, int num_threads,
                                 ab_affine_align_result *out, ab_plane_mut *aligned);
/* the star-list half (triangles -> votes -> RANSAC -> sanity) on given centroids; host only */
AB_API int ab_affine_from_stars(const double *ref_xy, size_t n_ref, const double *tgt_xy, size_t n_tgt, int64_t rows,
                                int64_t cols, int num_threads, ab_affine_align_result *out, int *found);

/* ---- a9  core/imaging/stats.rs ------------------------------------------------------------- */
typedef struct { /* ImageStats, types/image.rs:2-10 */
    double min, max, median, mad, sigma, mean;
    uint64_t valid_count;
} ab_image_stats;

/* compute_image_stats (stats.rs:15-23): exact select for <= 4 000 000 px, 65 536-bin
 * histogram refinement above.  Synchronous (returns scalars). */
AB_API int ab_compute_image_stats(ab_ctx *ctx, const ab_plane *img, ab_image_stats *out);
/* compute_image_stats_with_known_range (stats.rs:25-41) */
AB_API int ab_compute_image_stats_with_known_range(ab_ctx *ctx, const ab_plane *img, double known_min,
                                                   double known_max, ab_image_stats *out);
/* build_histogram (stats.rs:378-421): bins u32[bins] on the HOST; range < 1e-10 -> zeros */
AB_API int ab_build_histogram(ab_ctx *ctx, const ab_plane *img, size_t bins, double dmin, double dmax,
                              uint32_t *out_bins_host);
/* pass-2 histogram of the >4M-px path (stats.rs:260-300), exposed for bin-for-bin parity
 * tests and for the multi-GPU all-reduce of row-band partial histograms. */
AB_API int ab_stats_value_hist(ab_ctx *ctx, const ab_plane *img, double gmin, double gmax,
                               uint64_t *hist65536_host, double *out_sum, uint64_t *out_cnt);

/* ---- a10/a11  core/imaging/stf.rs ----------------------------------------------------------- */
typedef struct { double shadow, midtone, highlight; } ab_stf_params;        /* types/image.rs:36-40 */
typedef struct { double target_bg, shadow_k; } ab_auto_stf_config;          /* types/image.rs:52-65 */
/* auto_stf (stf.rs:13-39) -- host scalar maths, kept in the library so callers get one ABI */
AB_API int ab_auto_stf(const ab_image_stats *stats, const ab_auto_stf_config *cfg, ab_stf_params *out);
/* apply_stf -> Vec<u8> (stf.rs:89-102) */
AB_API int ab_apply_stf_u8(ab_ctx *ctx, const ab_plane *img, const ab_stf_params *p, const ab_image_stats *st,
                           uint8_t *out, int32_t out_on_device);
/* apply_stf_f32 (stf.rs:104-120); out may alias img.data (apply_stf_inplace, stf.rs:147-155) */
AB_API int ab_apply_stf_f32(ab_ctx *ctx, const ab_plane *img, const ab_stf_params *p, const ab_image_stats *st,
                            ab_plane_mut *out);

/* auto_stretch_preview (cmd/common.rs:18-22): compute_image_stats -> auto_stf -> apply_stf as ONE asynchronous chain on the
 * device (the percentile bookkeeping between the histogram passes runs in one-workgroup kernels; the STF kernel reads its
 * transform from HBM).  img, out_u8_dev: device.  cfg NULL = AutoStfConfig::default().  out_stats / out_stf (nullable) are
 * fetched with the call's only synchronisation; both NULL keeps it fully asynchronous.  With a communicator, img is this
 * rank's row band of an image of total_rows rows (0 = img->rows) and the statistics are the whole image's. */
AB_API int ab_auto_stretch_preview(ab_ctx *ctx, ab_comm *comm, const ab_plane *img, int64_t total_rows, const ab_auto_stf_config *cfg,
                                   uint8_t *out_u8_dev, ab_image_stats *out_stats, ab_stf_params *out_stf);

/* ---- a14  core/imaging/scnr.rs ----------------------------------------------------------------- */
typedef struct { /* ScnrConfig, types/image.rs:82-100 */
    int32_t method; /* 0 AverageNeutral, 1 MaximumNeutral */
    float amount;
    int32_t preserve_luminance;
} ab_scnr_config;
/* apply_scnr_inplace (scnr.rs:18-53): mutates the three caller-owned planes; mismatched dims or
 * amount < 1e-7 is a silent no-op exactly as in the reference. */
AB_API int ab_apply_scnr_inplace(ab_ctx *ctx, ab_plane_mut *r, ab_plane_mut *g, ab_plane_mut *b,
                                 const ab_scnr_config *cfg);

/* ---- a16  core/compose/channel_blend.rs ------------------------------------------------------------ */
typedef struct { /* BlendWeight, channel_blend.rs:5-11 */
    uint64_t channel_idx;
    double r_weight, g_weight, b_weight;
} ab_blend_weight;
/* blend_channels (channel_blend.rs:13-70): out_c[i] = sum over weights (list order) of ch[idx][i] * w_c,
 * f32 multiply then add; weights whose channel_idx >= n_channels are skipped. */
AB_API int ab_blend_channels(ab_ctx *ctx, const ab_plane *channels, size_t n_channels, const ab_blend_weight *weights,
                             size_t n_weights, ab_plane_mut *r, ab_plane_mut *g, ab_plane_mut *b);

/* ---- a15  core/imaging/curves.rs ----------------------------------------------------------------------- */
typedef struct { double black, gamma, white; } ab_levels_params; /* LevelsParams, curves.rs:4-9 */
/* SplineLut::from_points (curves.rs:69-95): Fritsch-Carlson monotone cubic -> 4096-entry LUT (host maths) */
AB_API int ab_spline_lut_from_points(const double *points_xy, size_t n_points, float *lut4096);
/* apply_curve (curves.rs:186-197): truncating LUT lookup, non-finite or negative -> 0 */
AB_API int ab_apply_curve(ab_ctx *ctx, const ab_plane *img, const float *lut4096_host, ab_plane_mut *out);
/* apply_levels (curves.rs:31-52): clamp((v-black)/(white-black))^(1/gamma) in f64 */
AB_API int ab_apply_levels(ab_ctx *ctx, const ab_plane *img, const ab_levels_params *p, ab_plane_mut *out);

/* ---- a19  core/imaging/stretch.rs:10-45 ------------------------------------------------------------------ */
AB_API int ab_arcsinh_stretch_with_stats(ab_ctx *ctx, const ab_plane *img, float dmin, float dmax, float factor,
                                         float gamma, ab_plane_mut *out);

/* ---- helpers of a13/a17/a18: luminance (masked_stretch.rs:143-154), WB scale (cmd/compose/color.rs:28-40) */
AB_API int ab_luminance(ab_ctx *ctx, const ab_plane *r, const ab_plane *g, const ab_plane *b, ab_plane_mut *out);
AB_API int ab_scale(ab_ctx *ctx, const ab_plane *img, float factor, ab_plane_mut *out);

/* ---- a20  core/stacking/calibration.rs ---------------------------------------------------------------------- */
/* calibrate_image (calibration.rs:47-82): (raw - bias - dark*ratio) / flat (flat finite, |f| > 1e-4), clamped
 * at 0; any of bias/dark/flat may be NULL. */
AB_API int ab_calibrate_image(ab_ctx *ctx, const ab_plane *raw, const ab_plane *bias, const ab_plane *dark,
                              const ab_plane *flat, float dark_exposure_ratio, ab_plane_mut *out);
/* median_combine_row_major (calibration.rs:84-125), the per-pixel combine of create_master_{bias,dark,flat}:
 * element [len/2] of the finite samples (upper median), 0 if none.  Any n >= 1 (as ab_stack_sigma_clip). */
AB_API int ab_median_combine(ab_ctx *ctx, const ab_plane *planes, size_t n, ab_plane_mut *out);

/* ---- a12  core/imaging/background.rs ---------------------------------------------------------------------------- */
typedef struct { /* BackgroundConfig, background.rs:14-33 (defaults 8, 3, 2.5, 3, Subtract) */
    size_t grid_size, poly_degree;
    float sigma_clip;
    size_t iterations;
    int mode; /* CorrectionMode: 0 Subtract, 1 Divide */
} ab_background_config;
typedef struct { /* BackgroundResult's scalars (:35-42) + the fitted coefficients */
    size_t sample_count;
    double rms_residual;
    double coeffs[21];
} ab_background_info;
/* extract_background(image, config) (background.rs:55-116): grid of cell medians clipped against the global
 * median / MAD -> polynomial surface (degree <= 5) by ridge least squares -> model and corrected image.
 * out_model may be NULL.  Error strings follow the reference ("Image too small for grid_size=..",
 * "Not enough background samples (..) for polynomial degree ..", "Failed to solve polynomial fit: ..."). */
AB_API int ab_extract_background(ab_ctx *ctx, const ab_plane *img, const ab_background_config *cfg, ab_plane_mut *out_model,
                                 ab_plane_mut *out_corrected, ab_background_info *info);

/* ---- a13  core/imaging/star_mask.rs, masked_stretch.rs --------------------------------------------------------- */
typedef struct { /* StarMaskConfig, star_mask.rs:6-30 (defaults 2.5, 4.0, 5.0, 1.5, 30.0, false, 0.85) */
    double growth_factor, softness, detection_sigma, min_fwhm, max_fwhm;
    int luminance_protect;
    double luminance_ceiling;
} ab_star_mask_config;
typedef struct { size_t stars_masked; double coverage_fraction; } ab_star_mask_info; /* StarMaskResult scalars, :32-37 */
/* generate_star_mask (star_mask.rs:38-44) = detect_stars(image, detection_sigma) + the painter below */
AB_API int ab_generate_star_mask(ab_ctx *ctx, const ab_plane *img, const ab_star_mask_config *cfg, ab_plane_mut *out_mask,
                                 ab_star_mask_info *info);
/* generate_star_mask_from_detection (star_mask.rs:46-138): stars with min_fwhm <= fwhm <= max_fwhm paint a disc of
 * radius fwhm * growth_factor plus a smoothstep skirt `softness` wide (max-combined); optional luminance protection
 * above luminance_ceiling; coverage = fraction of mask > 0.01.  Only x, y, fwhm of each star are read. */
AB_API int ab_generate_star_mask_from_stars(ab_ctx *ctx, const ab_plane *img, const ab_detected_star *stars, size_t n_stars,
                                            const ab_star_mask_config *cfg, ab_plane_mut *out_mask, ab_star_mask_info *info);
typedef struct { /* MaskedStretchConfig, masked_stretch.rs:7-32 (defaults 10, 0.25, 2.5, 4.0, true, 0.85, 0.85, 1e-5) */
    size_t iterations;
    double target_background, mask_growth, mask_softness;
    int luminance_protect;
    double luminance_ceiling, protection_amount, convergence_threshold;
} ab_masked_stretch_config;
typedef struct { /* MaskedStretchResult scalars, masked_stretch.rs:34-42 */
    size_t iterations_run;
    double final_background;
    size_t stars_masked;
    double mask_coverage;
    int converged;
} ab_masked_stretch_result;
/* masked_stretch (masked_stretch.rs:44-58): star mask from the image itself, then the loop below */
AB_API int ab_masked_stretch(ab_ctx *ctx, const ab_plane *img, const ab_masked_stretch_config *cfg, ab_plane_mut *out,
                             ab_masked_stretch_result *res);
/* masked_stretch_with_mask (:60-118): normalise to [0,1]; up to `iterations` times: bg = [len/2] element of the
 * unmasked (mask < 0.5) positive pixels, stop at target / stagnation, else blend the MTF-stretched image in with
 * weight 1 - mask * protection; clamp.  mask_info (nullable) is echoed into res (stars_masked, mask_coverage). */
AB_API int ab_masked_stretch_with_mask(ab_ctx *ctx, const ab_plane *img, const ab_plane *mask,
                                       const ab_star_mask_info *mask_info, const ab_masked_stretch_config *cfg,
                                       ab_plane_mut *out, ab_masked_stretch_result *res);
/* masked_stretch_rgb_shared (:155-193): one mask from the luminance (:120-153), applied to r, g, b; res3[0..3] */
AB_API int ab_masked_stretch_rgb_shared(ab_ctx *ctx, const ab_plane *r, const ab_plane *g, const ab_plane *b,
                                        const ab_masked_stretch_config *cfg, ab_plane_mut *out_r, ab_plane_mut *out_g,
                                        ab_plane_mut *out_b, ab_masked_stretch_result *res3, ab_star_mask_info *shared);

/* ---- a17  core/compose/rgb.rs, white_balance.rs, core/imaging/resample.rs ----------------------------------------- */
/* resample_image(image, target_rows, target_cols) (resample.rs:25-61): bicubic resize to out's dims */
AB_API int ab_resample_image(ab_ctx *ctx, const ab_plane *src, ab_plane_mut *out);
/* select_wb_reference (white_balance.rs:3-20): multipliers (r, g, b) against the most stable channel (host maths) */
AB_API int ab_select_wb_reference(const ab_image_stats *sr, const ab_image_stats *sg, const ab_image_stats *sb,
                                  double out_rgb[3]);
typedef struct { /* RgbComposeConfig, types/compose.rs:47-75 */
    int32_t white_balance;      /* WhiteBalance: 0 Auto, 1 Manual(wb_manual), 2 None */
    double wb_manual[3];
    int32_t auto_stretch, linked_stf;
    int32_t has_stf[3];         /* Option<StfParams> stf_r / stf_g / stf_b (used when !auto_stretch) */
    ab_stf_params stf[3];
    int32_t align, align_method; /* AlignMethod: 0 PhaseCorrelation, 1 Affine */
    int32_t has_scnr;           /* Option<ScnrConfig> */
    ab_scnr_config scnr;
    int32_t num_threads;        /* rayon::current_num_threads() of the host being replaced (affine RANSAC seeds) */
} ab_rgb_compose_config;
typedef struct { /* scalars of ProcessedRgb, rgb.rs:18-40 */
    uint64_t rows, cols;
    ab_stf_params stf[3];
    double chan_stats[3][4];    /* ChannelStats {min, max, median, mean} per channel, before white balance */
    double offset_g[2], offset_b[2]; /* (dy, dx) resp. (ty, tx) */
    int32_t scnr_applied, resampled;
    ab_image_stats stats_wb[3]; /* stats after white balance (the ones the STF is applied with) */
} ab_processed_rgb_info;
/* process_rgb(r?, g?, b?, &config) (rgb.rs:209-323): harmonise dims (bicubic up to the largest), synthesise a missing
 * channel, align G and B to the first present channel, white balance, (linked) auto-STF, compose-local STF
 * (rgb.rs:191-207), SCNR.  Absent channels are NULL; at least two must be present.  out_* (and the nullable
 * pre_* = white-balanced, pre-stretch copies) must have the largest channel's dims. */
AB_API int ab_process_rgb(ab_ctx *ctx, const ab_plane *r, const ab_plane *g, const ab_plane *b,
                          const ab_rgb_compose_config *cfg, ab_plane_mut *out_r, ab_plane_mut *out_g, ab_plane_mut *out_b,
                          ab_plane_mut *pre_r, ab_plane_mut *pre_g, ab_plane_mut *pre_b, ab_processed_rgb_info *info);

/* ---- a18  core/astrometry/spcc.rs -------------------------------------------------------------------------------- */
typedef struct { /* SpccConfig, spcc.rs:9-28 (defaults 20.0, 200, 0.90, AverageSpiral; catalog = BuiltinBpRp) */
    double min_snr;
    uint64_t max_stars;
    double saturation_limit;
    int32_t white_reference; /* WhiteReference: 0 AverageSpiral, 1 G2V, 2 Photopic, 3 Custom(custom) */
    double custom[3];
} ab_spcc_config;
typedef struct { /* the numbers of SpccResult, spcc.rs:45-56 */
    double r_factor, g_factor, b_factor;
    uint64_t stars_matched, stars_total;
    double avg_color_index;
} ab_spcc_result;
/* spcc_calibrate_rgb(r, g, b, header, config) (spcc.rs:73-183) with the built-in Bp-Rp catalogue.  The header's
 * WCS enters that path only through its pixel scale (the catalogue is synthesised from the detections' own
 * sky positions, :257-273, so the cross-match is the identity whenever pixel_scale > 0): the caller passes
 * WcsTransform::pixel_scale_arcsec().  Err strings as the reference ("Only N stars passed quality filters
 * (need 5+). Try lowering min_snr." / "Only N stars cross-matched (need 3+). Check WCS solution quality."). */
AB_API int ab_spcc_calibrate_rgb(ab_ctx *ctx, const ab_plane *r, const ab_plane *g, const ab_plane *b,
                                 double pixel_scale_arcsec, const ab_spcc_config *cfg, ab_spcc_result *res);
/* the part after detection (spcc.rs:90-183) on a given detect_stars() result and luminance maximum */
AB_API int ab_spcc_from_detection(ab_ctx *ctx, const ab_plane *r, const ab_plane *g, const ab_plane *b,
                                  const ab_detected_star *stars, size_t n_stars, double lum_max, double pixel_scale_arcsec,
                                  const ab_spcc_config *cfg, ab_spcc_result *res);
/* white_reference_rgb (spcc.rs:245-255); host maths */
AB_API int ab_spcc_white_reference_rgb(int32_t kind, const double custom[3], double out_rgb[3]);

/* ---- caller-side helpers of a17 / a20 ---------------------------------------------------------------------------- */
/* apply_lrgb(l, &mut r, &mut g, &mut b, lightness_weight, chrominance_weight) (core/compose/lrgb.rs:4-45); mutates
 * r, g, b; Err "L dimensions .. do not match RGB (..)" on mismatched dims */
AB_API int ab_apply_lrgb(ab_ctx *ctx, const ab_plane *l, ab_plane_mut *r, ab_plane_mut *g, ab_plane_mut *b,
                         float lightness_weight, float chrominance_weight);
/* lrgb.rs:47-64 synthesize_luminance (= spcc.rs:185-196): r*0.2126 + g*0.7152 + b*0.0722, NO finite guard
 * (ab_luminance is the guarded masked_stretch.rs variant) */
AB_API int ab_synthesize_luminance(ab_ctx *ctx, const ab_plane *r, const ab_plane *g, const ab_plane *b, ab_plane_mut *out);
/* compute_linked_stf_with_stats (cmd/helpers.rs:185-202): auto_stf of the channel-averaged statistics (host maths);
 * out_combined nullable */
AB_API int ab_compute_linked_stf(const ab_image_stats *sr, const ab_image_stats *sg, const ab_image_stats *sb,
                                 const ab_auto_stf_config *cfg, ab_stf_params *out_stf, ab_image_stats *out_combined);
/* calibrate_channel (cmd/compose/color.rs:21-49): out = orig * factor, statistics of the result (known-range fast
 * path above 4 000 000 px) */
AB_API int ab_calibrate_channel(ab_ctx *ctx, const ab_plane *orig, float factor, const ab_image_stats *orig_stats,
                                ab_plane_mut *out, ab_image_stats *out_stats);
/* create_master_bias / _dark / _flat on in-memory frames (calibration.rs:127-255): kind 0 bias = median combine;
 * 1 dark = median of (frame - bias?); 2 flat = median of (frame - bias? - dark?), normalised to mean 1 over its
 * finite positive pixels (others -> 1.0).  Err strings as the reference ("No bias frames provided", "Dimension
 * mismatch: expected (..), got (..)").  Any n_frames >= 1. */
AB_API int ab_create_master(ab_ctx *ctx, int32_t kind, const ab_plane *frames, size_t n_frames, const ab_plane *master_bias,
                            const ab_plane *master_dark, ab_plane_mut *out);

/* ---- FITS pixel codecs, infra/fits/reader.rs + writer.rs ------------------------------------------- */
/* decode_pixels(data, bitpix, bscale, bzero) (reader.rs:42-101): big-endian BITPIX 8 / 16 / 32 / -32 / -64 -> f32,
 * `v as f64 * bscale + bzero` unless is_identity_scaling (:36-39).  `data` = the HDU's data unit as it sits in the
 * file (host or device); out must hold nbytes / bytes-per-pixel pixels. */
AB_API int ab_fits_decode_pixels(ab_ctx *ctx, const void *data, size_t nbytes, int32_t data_on_device, int64_t bitpix, double bscale,
                                 double bzero, ab_plane_mut *out);
/* compute_bzero_bscale (writer.rs:143-159): i16 scaling from the finite min / max */
AB_API int ab_fits_compute_bzero_bscale(ab_ctx *ctx, const ab_plane *img, double *bzero, double *bscale);
/* write_f32 / i16 / f64_slice_as_be (writer.rs:82-135): the data unit for BITPIX -32, 16 (with bzero / bscale) or -64 */
AB_API int ab_fits_encode_pixels(ab_ctx *ctx, const ab_plane *img, int32_t bitpix, double bzero, double bscale, void *out,
                                 int32_t out_on_device);
/* stack_images' per-pixel loop (combine.rs:160-182) fed straight from n device-resident data units (BITPIX -32 or 16,
 * all rows x cols of `out`): decode_pixels is fused into the kernel's gather, so BITPIX 16 stacks read 2 bytes per
 * sample from HBM and no decoded copy exists.  Result = ab_fits_decode_pixels + ab_stack_sigma_clip, bit for bit.
 * n must be 8, 16, 32 or 64.  Every plane must hold rows x cols samples and be readable up to the next 4-byte boundary
 * (BITPIX 16 with an odd pixel count: the last sample is fetched inside an aligned dword; a FITS data unit is padded to
 * 2880 bytes, so a whole data unit always qualifies). */
AB_API int ab_stack_sigma_clip_raw(ab_ctx *ctx, const void *const *raw_planes_dev, size_t n, int64_t bitpix, double bscale,
                                   double bzero, const ab_stack_config *cfg, ab_plane_mut *out, uint64_t *out_rejected);

/* ---- batch calibration pipeline, core/imaging/calibration_pipeline.rs ------------------------------- */
typedef struct { /* BatchStackConfig (:20-37); defaults 2.5, 3.0, 5, true */
    float sigma_low, sigma_high;
    uint64_t max_iterations;
    int32_t normalize_before_stack; /* bool */
} ab_batch_stack_config;
typedef struct { /* CalibrationMasters (:6-11); NULL = None.  A master whose rows * cols differs from the light's is skipped (:87-89) */
    const ab_plane *bias, *dark, *flat;
} ab_calibration_masters;
typedef struct { /* BatchChannelStats (:65-72) without the label; lights_after_rejection is the rejection_counts array */
    uint64_t lights_input;
    double mean, stddev;
} ab_batch_channel_stats;
typedef struct { /* ChannelInput (:13-17) + where the channel's per-frame rejection counts go (n_lights entries, nullable) */
    const char *label;
    const ab_plane *lights;
    size_t n_lights;
    uint64_t *rejection_counts;
} ab_batch_channel_input;
AB_API void ab_batch_stack_config_default(ab_batch_stack_config *cfg);
/* calibrate_light (:74-118): ((light - bias) - dark) / flat where flat is finite and |flat| > 1e-4, negatives -> 0 */
AB_API int ab_calibrate_light(ab_ctx *ctx, const ab_plane *light, const ab_calibration_masters *masters, ab_plane_mut *out);
/* normalize_frames (:309-319): frame * (1 / (mean as f32)) where the f64 mean is > 0, else a copy.  outs[i] may alias frames[i].
 * The f64 mean is a fixed-shape tree sum (the reference's is sequential): equal to ~1e-13 relative. */
AB_API int ab_normalize_frames(ab_ctx *ctx, const ab_plane *frames, size_t n, ab_plane_mut *outs);
/* sigma_clipped_mean_stack (:321-378): per pixel, up to max_iterations passes of { median, MAD -> sigma = 1.4826 MAD (f32);
 * stop if sigma < 1e-10; keep -sigma_low < (v - median) / sigma < sigma_high }, NaN samples included and rejected by the
 * first pass; result = f32 sum of the survivors in frame order / count (0 if none).  rejection_counts[f] (nullable) =
 * samples of frame f rejected over the whole image.  Any n >= 1 frames of identical dims (65 .. 2048: one wave per
 * pixel instead of one lane, ~40x slower per sample; beyond: one workgroup per pixel, samples in a scratch segment).  Bit-exact. */
AB_API int ab_sigma_clipped_mean_stack(ab_ctx *ctx, const ab_plane *frames, size_t n, const ab_batch_stack_config *config, ab_plane_mut *out,
                                       uint64_t *rejection_counts);
/* one channel of run_batch_pipeline (:157-190): calibrate_light on every light, normalize_frames (if configured),
 * sigma_clipped_mean_stack, mean / stddev of the master.  Fused: the lights are read twice (frame means, stack) and no
 * calibrated or normalised frame is ever written; the stacked samples are bit-identical to the reference's. */
AB_API int ab_run_batch_channel(ab_ctx *ctx, const ab_plane *lights, size_t n, const ab_calibration_masters *masters,
                                const ab_batch_stack_config *config, ab_plane_mut *out_master, uint64_t *rejection_counts,
                                ab_batch_channel_stats *stats);
/* compose_rgb_from_masters (:201-267) for the R, G, B (and optional L) masters: normalize_channel (:291-307) on the common
 * top-left crop, apply_luminance (:269-289) when all four share their dims; out_rgb = rows x cols x 3 interleaved f32
 * (Array3), host or device.  out_rgb == NULL only reports the dims. */
AB_API int ab_compose_rgb_from_masters(ab_ctx *ctx, const ab_plane *r, const ab_plane *g, const ab_plane *b, const ab_plane *l,
                                       float *out_rgb, int32_t out_on_device, int64_t *out_rows, int64_t *out_cols);
/* run_batch_pipeline (:120-199): validation with the reference's messages, every channel through ab_run_batch_channel into
 * out_masters[c] (dims of the channel's lights), stats[c] (nullable), then compose_rgb_from_masters when channels labelled
 * R, G and B exist (ASCII case-insensitive): out_rgb must then hold min-rows x min-cols x 3 floats; *rgb_rows = 0 = None. */
AB_API int ab_run_batch_pipeline(ab_ctx *ctx, const ab_batch_channel_input *channels, size_t n_channels,
                                 const ab_calibration_masters *masters, const ab_batch_stack_config *config, ab_plane_mut *out_masters,
                                 ab_batch_channel_stats *stats, float *out_rgb, int32_t rgb_on_device, int64_t *rgb_rows,
                                 int64_t *rgb_cols);

/* ---- subframe scoring, core/analysis/subframe.rs -------------------------------------------------- */
typedef struct { /* SubframeWeightConfig (subframe.rs:24-49) */
    double fwhm_weight, eccentricity_weight, snr_weight, noise_weight, max_fwhm, max_eccentricity, min_snr;
    uint64_t min_stars;
} ab_subframe_weight_config;
typedef struct { /* the numbers of SubframeMetrics (subframe.rs:9-22); file_path stays with the caller */
    uint64_t star_count;
    double median_fwhm, median_eccentricity, median_snr, background_median, background_sigma, noise_ratio, weight;
    int32_t accepted;
} ab_subframe_metrics;
AB_API void ab_subframe_weight_config_default(ab_subframe_weight_config *cfg);      /* Default, :36-49 */
/* analyze_subframe(image, _, config) (subframe.rs:51-121): detect_stars(image, 4.0), medians of the finite fwhm /
 * eccentricity / snr, noise_ratio, compute_weight (:123-146) and the accept flags.  config == NULL -> defaults. */
AB_API int ab_analyze_subframe(ab_ctx *ctx, const ab_plane *image, const ab_subframe_weight_config *config, ab_subframe_metrics *out);
/* the caller's loop over a night's subframes: n frames scored frame-parallel on the context's worker streams */
AB_API int ab_analyze_subframes(ab_ctx *ctx, const ab_plane *images, size_t n, const ab_subframe_weight_config *config,
                                ab_subframe_metrics *out);
AB_API void ab_normalize_subframe_weights(ab_subframe_metrics *metrics, size_t n);  /* normalize_weights, :148-159 */

/* ---- preview / tile renderers up to the PNG encoder ---------------------------------------------- */
/* preview size for max_dim (cmd/helpers.rs:283-290, infra/ipc.rs:100-103): the plane's own dims when both fit */
AB_API int ab_preview_dims(int64_t rows, int64_t cols, int64_t max_dim, int64_t *out_rows, int64_t *out_cols);
/* render_rgb_preview (helpers.rs:204-262; == render_rgb, infra/render/rgb.rs:7-34, when the planes fit max_dim) with
 * stf == NULL: nearest-neighbour pick, (v.clamp(0, 1) * 255.0) as u8.  render_rgb_preview_with_stf (:264-322) with
 * stf / stats = 3 entries (R, G, B): make_stf_u8_fn(stf[c], stats[c]) (core/imaging/stf.rs:122-145) per channel.
 * out_rgb = preview_rows x preview_cols x 3 interleaved bytes (host or device), i.e. the encoder's input. */
AB_API int ab_render_rgb_preview(ab_ctx *ctx, const ab_plane *r, const ab_plane *g, const ab_plane *b, int64_t max_dim,
                                 const ab_stf_params *stf, const ab_image_stats *stats, uint8_t *out_rgb, int32_t out_on_device);
/* encode_with_header (infra/ipc.rs:93-103; max_dim == 0) / encode_with_header_downsampled (:105-148): 16-byte header
 * (u32 width, u32 height, f32 data_min, f32 data_max, little-endian) + cleaned f32 pixels (non-finite -> 0).
 * out holds 16 + 4 * preview_rows * preview_cols bytes; *out_len (nullable) = the length written. */
AB_API int ab_ipc_encode_with_header(ab_ctx *ctx, const ab_plane *img, int64_t max_dim, void *out, int32_t out_on_device, size_t *out_len);
#define AB_MAX_TILE_LEVELS 32
typedef struct { /* TileLevel (infra/render/tiles.rs:21-29) + where the level's tiles start in the packed buffer */
    uint64_t level, width, height, cols, rows;
    double scale_factor;
    uint64_t offset;
} ab_tile_level;
AB_API int ab_tile_compute_num_levels(int64_t width, int64_t height, int64_t tile_size);   /* tiles.rs:137-147; 0 = bad args */
/* the pyramid's levels (level 0 = coarsest, tiles.rs:203-246) and the packed size: per level, tiles in (tile_y, tile_x)
 * order, each tile_size x tile_size x channels bytes.  levels holds AB_MAX_TILE_LEVELS entries. */
AB_API int ab_tile_pyramid_layout(int64_t rows, int64_t cols, int64_t tile_size, int32_t channels, ab_tile_level *levels,
                                  int32_t *num_levels, size_t *total_bytes);
AB_API int ab_tile_downsample_2x(ab_ctx *ctx, const ab_plane *img, ab_plane_mut *out);      /* downsample_2x, tiles.rs:41-70 */
/* percentile_bounds (tiles.rs:149-178): order statistics of the finite pixels > 1e-7; finite min / max if there are none */
AB_API int ab_tile_percentile_bounds(ab_ctx *ctx, const ab_plane *img, double low_pct, double high_pct, float *lo, float *hi);
/* generate_tile_pyramid (tiles.rs:180-255) up to the encoder: percentile_bounds(0.001, 0.999), the 2x chain, and every
 * render_tile buffer (:72-113) into `tiles` (layout above, channels = 1; zero outside the image). */
AB_API int ab_generate_tile_pyramid(ab_ctx *ctx, const ab_plane *normalized, int64_t tile_size, uint8_t *tiles, int32_t tiles_on_device,
                                    ab_tile_level *levels, int32_t *num_levels, float *global_min, float *global_max);
/* generate_tile_pyramid_rgb / _rgb_stf (tiles.rs:363-481): stf == NULL -> render_tile_rgb (:257-298, rounded), else
 * render_tile_rgb_stf (:300-341) with make_stf_u8_fn(stf[c], stats[c]); channels = 3 */
AB_API int ab_generate_tile_pyramid_rgb(ab_ctx *ctx, const ab_plane *r, const ab_plane *g, const ab_plane *b, int64_t tile_size,
                                        const ab_stf_params *stf, const ab_image_stats *stats, uint8_t *tiles, int32_t tiles_on_device,
                                        ab_tile_level *levels, int32_t *num_levels);

/* ---- bench support: a plain float4 device copy, the measured HBM ceiling ---- */
AB_API int ab_bench_copy(ab_ctx *ctx, const float *src_dev, float *dst_dev, size_t n_floats);

/* ---- (e) multi-GPU: SURVEY.md 8e -------------------------------------------------------------- */
/* The reference is one process on one machine (rayon); what shards is its per-pixel loop (combine.rs:160-182, rows are
 * independent) and its frame list.  One ab_ctx + one ab_comm rank per GPU; collectives are RCCL (xGMI inside a node),
 * enqueued on the context's stream.  librccl is dlopen'ed on first use: the rest of the library works without it. */
#define AB_COMM_ID_BYTES 128
typedef enum { AB_DT_I32 = 0, AB_DT_U32 = 1, AB_DT_I64 = 2, AB_DT_U64 = 3, AB_DT_F32 = 4, AB_DT_F64 = 5 } ab_dtype;
typedef enum { AB_RED_SUM = 0, AB_RED_MAX = 1, AB_RED_MIN = 2 } ab_redop;
/* multi-process: rank 0 makes the id, the host carries it to the other processes, every rank joins */
AB_API int ab_comm_get_unique_id(uint8_t id[AB_COMM_ID_BYTES]);
AB_API int ab_comm_init_rank(ab_ctx *ctx, const uint8_t id[AB_COMM_ID_BYTES], int nranks, int rank, ab_comm **out);
/* single process, n contexts on n distinct devices: out_comms[i] is rank i, bound to ctxs[i]'s device.  Drive each
 * context from its own host thread, or bracket the per-device calls with ab_comm_group_start / _end. */
AB_API int ab_comm_init_all(ab_ctx *const *ctxs, int n, ab_comm **out_comms);
/* HOST-STAGED transport (no RCCL, ranks may share a device): collectives go through the POSIX shared-memory segment
 * /abcomm_<name> -- every rank of the job passes the same name; blocks until all nranks joined.  The same entry points
 * work on it; a collective then returns when its result is in device memory.  What a one-GPU box runs the N > 1 paths
 * on (tests/test_gpu_multirank.py) and a host without librccl falls back to.  AB_COMM_HOST_SLOT_MB (default 4) = the
 * per-rank staging window. */
AB_API int ab_comm_init_rank_host(ab_ctx *ctx, const char *name, int nranks, int rank, ab_comm **out);
AB_API int ab_comm_is_host(const ab_comm *comm);
/* Failure handling.  ab_comm_agree: every rank passes the status of its local work; AB_OK comes back only if every rank
 * passed AB_OK -- a failed rank gets its own status, the others AB_ERR_CANCELLED (a peer was cancelled) or AB_ERR_COMM.
 * Every sharded entry point below calls it BEFORE its data collectives, so one failing rank fails the call everywhere and
 * the communicator stays usable.  Waits on collectives are bounded by ab_comm_set_timeout_ms (default: AB_COMM_TIMEOUT_MS
 * or 300 000): a peer that died turns into AB_ERR_COMM and a dead communicator, not a hang.  ab_comm_abort gives up at
 * once (host transport: peers blocked in a collective return AB_ERR_COMM immediately; RCCL: ncclCommAbort); afterwards
 * every collective on the handle fails with AB_ERR_COMM and only ab_comm_destroy is useful. */
AB_API int ab_comm_agree(ab_ctx *ctx, ab_comm *comm, int local_status);
AB_API int ab_comm_abort(ab_comm *comm);
AB_API int ab_comm_set_timeout_ms(ab_comm *comm, int64_t ms);
AB_API void ab_comm_destroy(ab_comm *comm);
AB_API int ab_comm_rank(const ab_comm *comm);  /* a NULL communicator is a world of one: rank 0 */
AB_API int ab_comm_size(const ab_comm *comm);  /* ... of size 1 */
AB_API uint64_t ab_comm_collectives_issued(const ab_comm *comm);
AB_API int ab_comm_group_start(void);
AB_API int ab_comm_group_end(void);
/* in-place all-reduce of `count` elements on the context's stream (RCCL: asynchronous; host-staged: done on return) */
AB_API int ab_comm_allreduce(ab_ctx *ctx, ab_comm *comm, void *buf_dev, size_t count, int dtype /* ab_dtype */, int op /* ab_redop */);
/* recv_dev = size x bytes_per_rank bytes, rank r's block at r * bytes_per_rank */
AB_API int ab_comm_allgather(ab_ctx *ctx, ab_comm *comm, const void *send_dev, void *recv_dev, size_t bytes_per_rank);
AB_API int ab_comm_broadcast(ab_ctx *ctx, ab_comm *comm, void *buf_dev, size_t bytes, int root);

/* partitions: rows [row0, row0 + nrows) (ceil(rows / nranks) per rank, trailing bands shorter or empty) and frames
 * [f0, f0 + nf) (contiguous, balanced) of rank `rank` */
AB_API int ab_shard_rows(int64_t rows, int nranks, int rank, int64_t *row0, int64_t *nrows);
AB_API int ab_shard_frames(size_t n_frames, int nranks, int rank, size_t *f0, size_t *nf);
/* what rank `rank` must hold of every TARGET frame to warp its band (ab_shard_rows over out_rows) with the n transforms
 * (n x 6 doubles, e.g. ab_register_frames_sharded's estimates): the hull of ab_warp_source_rows = its rows + the halo the
 * transform set needs (about max |ty| + |c| cols + 2 rows either side) */
AB_API int ab_shard_source_rows(const double *transforms, size_t n, int64_t src_rows, int64_t src_cols, int64_t out_rows, int64_t out_cols,
                                int nranks, int rank, int64_t *src_row0, int64_t *src_nrows);

/* ROW-BAND (exact) sharding of stack_images' per-pixel loop (combine.rs:160-182): rows [row0, row0 + out_band->rows) of
 * the stack of ALL n frames -- the reference's single-level estimator restricted to a band, bit for bit.  Device planes. */
AB_API int ab_stack_sigma_clip_rows(ab_ctx *ctx, const ab_plane *planes, size_t n, const ab_stack_config *cfg, int64_t row0,
                                    ab_plane_mut *out_band, uint64_t *out_rejected);
/* the same with the band = this rank's share (ab_shard_rows over the minimum frame dims, combine.rs:104-113) and
 * StackResult.rejected_pixels summed over the ranks (one u64 all-reduce) */
AB_API int ab_stack_sigma_clip_rowband(ab_ctx *ctx, ab_comm *comm, const ab_plane *planes, size_t n, const ab_stack_config *cfg,
                                       ab_plane_mut *out_band, uint64_t *out_rejected_total);
/* FRAME-SHARDED two-level stack (BASELINE configs[3]): ab_stack_sigma_clip_partial over this rank's frames ->
 * all-reduce(sum f64, count u32) -> ab_stack_finalize_partial.  `out` (device) is the full image on every rank.  NOT the
 * reference's estimator (a median over all frames is not decomposable); checker: orc_stack_partial_noalign. */
AB_API int ab_stack_sigma_clip_sharded(ab_ctx *ctx, ab_comm *comm, const ab_plane *local_planes, size_t n_local,
                                       const ab_stack_config *cfg, ab_plane_mut *out, uint64_t *out_rejected_total);
/* the last ab_stack_sigma_clip_sharded of this context: the span of its partial stacks on the context's stream, and the time inside
 * its all-reduces + divisions (they run chunk by chunk on a second stream while the next chunk is stacked: with overlap the two add
 * up to more than the call took).  Blocks until that call's work is done.  bench.py: stage_ms.comm_* */
AB_API int ab_stack_sharded_last_ms(ab_ctx *ctx, float *stack_ms, float *comm_ms);
/* every rank's band -> the full image on every rank (one broadcast per rank inside one RCCL group) */
AB_API int ab_allgather_rows(ab_ctx *ctx, ab_comm *comm, const ab_plane *band, ab_plane_mut *full);
/* align_channel_affine(reference, targets[i]) for i < n, target i estimated on rank i mod size, all n results on every
 * rank (exchanged bit for bit as integer words); targets a rank does not own are not read there */
AB_API int ab_register_frames_sharded(ab_ctx *ctx, ab_comm *comm, const ab_plane *reference, const ab_plane *targets, size_t n,
                                      int num_threads, ab_affine_align_result *out);
/* The row-band scheme's registration as ONE call (pair.rs:41-77 per target): this rank's rows [row0, row0 + out_bands[i].rows)
 * of warp_image(target i, its transform) for every i.  targets[i] holds rows [target_row0[i], + targets[i].rows) of target i
 * (target_row0 NULL = whole frames): whole for the targets this rank estimates (i mod size == rank), of the others at least the rows
 * its band reads (ab_shard_source_rows).  Own frames are warped as they are fitted, overlapped with the remaining estimates; then the
 * estimates are exchanged and the other frames warped from their bands.  out[i] / the pixels equal ab_register_frames_sharded +
 * ab_warp_image_rows(_from_band) bit for bit. */
AB_API int ab_align_pairs_affine_rowband(ab_ctx *ctx, ab_comm *comm, const ab_plane *reference, const ab_plane *targets, const int64_t *target_row0,
                                         size_t n, int num_threads, int64_t row0, ab_affine_align_result *out, ab_plane_mut *out_bands);
/* compute_image_stats (stats.rs:15-210) of an image whose rows are spread over the ranks: `band` = this rank's rows,
 * total_rows = the whole image's.  min / max, counts and the 65 536-bin histograms are all-reduced between the passes
 * (integers: every bin equals the single-GPU bin); every rank receives the whole image's statistics. */
AB_API int ab_compute_image_stats_sharded(ab_ctx *ctx, ab_comm *comm, const ab_plane *band, int64_t total_rows, ab_image_stats *out);

#ifdef __cplusplus
}
#endif
#endif /* ASTROBURST_HIP_H */
