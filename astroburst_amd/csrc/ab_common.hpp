// Internal shared definitions of libastroburst_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <atomic>
#include <chrono>
#include <functional>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/astroburst_hip.h"

#define AB_REJ_SLOTS 2048

enum {
    AB_WS_DETECT_PARENT = 0,  // int[P] union-find forest (defined at labelled pixels only)
    AB_WS_DETECT_CID,         // int[P] root -> component id (written at roots only)
    AB_WS_DETECT_ROOTS,       // int[P/4 + 1] component roots + counters
    AB_WS_DETECT_COMPS,       // per-component statistics / moment records
    AB_WS_DETECT_LIST,        // int[P] indices of the above-threshold pixels
    AB_WS_DETECT_MASK,        // u32[P / 32] one bit per pixel: above threshold
    AB_WS_REGISTER,           // triangle tables / votes of the star matcher
    AB_WS_STACK_DEFER,        // per-slot pixel lists of the stacking kernel's two-pass mode
    AB_WS_RENDER,             // the 2x-reduced levels of a tile pyramid
    AB_WS_BATCH_REJ,          // per-block per-frame rejection counters of the batch stack
    AB_WS_STACK_WIDE,         // plane pointer / stride tables of a > 64-frame stack
    AB_WS_BATCH_WIDE,         // tables of a > 64-frame batch stack
    AB_WS_STACK_INF,          // one plane of +inf: stands in for the frames a stack is short of a power of two
    AB_WS_STACK_PAIR_LISTS,   // the pixels a two-lane fast pass hands to the oracle-arithmetic kernel (stack_pair.hip, stack_duo.hip)
    AB_WS_BATCH_PAD,          // one plane of FLT_MAX: the same for the batch stack (where +inf is a sample like any other)
    AB_WS_STATS,              // state block, 65 536-bin histograms and partials of the statistics chain (stats.hip)
    AB_WS_SHARD,              // (sum f64, count u32) partial planes of the frame-sharded stack (sharded.hip)
    AB_WS_SUBSAMPLE,          // the <= ~100 000-pixel subsample normalize_for_detection takes its percentiles from
    AB_WS_DETECT_DEV,         // FrameDev + the tile statistics of the chained detection (detect.hip)
    AB_WS_REGISTER_GROUP,     // the target-side matcher workspaces of a group of frames (affine.hip)
    AB_WS_PHASE_TABLES,       // two sets of Hann windows + FFT twiddles of the phase correlation, kept between calls (phase_corr.hip)
    AB_WS_PIPE_TABLES,        // plane pointers + transforms of the fed background pipeline (detect.hip: ab_bg_pipeline_begin_fed)
    AB_WS_PIPE_SUBSAMPLE,     // its subsamples (one per plane: the chunks' percentile launches overlap)
    AB_WS_SCOPE0,             // eight slots for the planes an entry point needs for the length of one call (masked stretch: luminance,
    AB_WS_SCOPE1,             // mask, disc table, coverage counter, chain state; SPCC: luminance, apertures, fluxes): grow-only like every
    AB_WS_SCOPE2,             // workspace -- a hipMalloc + hipFree pair per call cost 0.2-0.5 ms each for a 268 MB plane
    AB_WS_SCOPE3,
    AB_WS_SCOPE4,
    AB_WS_SCOPE5,
    AB_WS_SCOPE6,
    AB_WS_SCOPE7,
    AB_WS_STACK_DEEP,         // plane tables + per-workgroup sample segments of a > 4096-frame stack (stack_deep.hip)
    AB_WS_BATCH_DEEP,         // the same for the batch stack (batch_pipeline.hip)
    AB_WS_STACK_SHIFTED,      // the registered copies of stack_images(align = true)'s frames 1 .. n - 1 (stack_images.hip)
    AB_WS_DETECT_CAND,        // the candidate lists the tile pass of a registration batch leaves for the labelling pass (detect.hip: TileCand)
    AB_WS_SLOTS
};

// normalize_for_detection (affine.rs:24-53) as a per-pixel transform applied on load: consumers of a normalised frame
// (tile statistics, threshold, moments) read the raw plane and evaluate clamp((v - lo) * inv, 0, 1) themselves, so
// the normalised copy is never written (-128 MB of traffic and one kernel per frame).  on = 0: identity.
struct ab_pixel_xf {
    double lo = 0.0, inv = 1.0;
    int on = 0;
};
// Wave priority (s_setprio) of the latency-bound kernels of the registration chain (labelling, component statistics, triangle
// matcher), which inside a batch share SIMDs with the f64-saturated warp kernel.  MEASURED AND LEFT OFF (round 4, same box,
// interleaved, tools/time_register.py): priority 3 for these kernels 12.0-12.4 ms min / 13.2-13.8 median against 11.2-11.6 /
// 12.4-12.7 without; priority 3 for the warp instead (-DAB_WARP_WAVE_PRIO=3) 12.7-15.8 median against 11.9-12.8.  Equal
// priorities are the best this stage gets; the switches stay for the next person who suspects the arbiter.
#ifdef AB_LATENCY_PRIO
#define AB_LATENCY_KERNEL_PRIO() __builtin_amdgcn_s_setprio(3)
#else
#define AB_LATENCY_KERNEL_PRIO() ((void)0)
#endif

__device__ __forceinline__ float ab_px(const ab_pixel_xf &x, float v) {
    if (!x.on) return v;
    double t = ((double)v - x.lo) * x.inv;
    t = t < 0.0 ? 0.0 : (t > 1.0 ? 1.0 : t);  // f64::clamp: NaN stays NaN
    return (float)t;
}

struct ab_ctx {
    int device = 0;
    hipStream_t own_stream = nullptr;
    hipStream_t stream = nullptr;  // own_stream or a borrowed one
    std::string err;
    // scratch arena in HBM (grown on demand, reused across calls)
    void *scratch = nullptr;
    size_t scratch_bytes = 0;
    // small pinned host buffer for scalar/histogram read-back
    void *pinned = nullptr;
    size_t pinned_bytes = 0;
    // device-side u64 counters (rejected pixels etc.)
    unsigned long long *counters = nullptr;  // AB_REJ_SLOTS x u64
    int cu_count = 0;
    unsigned int *sel_hist = nullptr;  // 2048-bin device histogram of plane_select.hip
    // named persistent device workspaces (grown on demand, kept until the context dies): the
    // whole-image passes of detection / registration reuse them instead of hipMalloc per call
    void *ws[AB_WS_SLOTS] = {};
    size_t ws_bytes[AB_WS_SLOTS] = {};
    // frame-parallel registration (affine.hip): child contexts (own stream + workspaces), one per host worker;
    // AB_REGISTER_WORKERS overrides the default.  The stage is bound by the four in-order hardware queues the worker streams
    // share; end of round 2, whole stage, three runs each: 6 -> 17.7 ms, 8 -> 16.9, 10 -> 17.0, 12 -> 16.8,
    // 14 -> 17.2, 16 -> 17.3, 20 -> 17.9, 24 -> 18.2
    std::vector<ab_ctx *> workers;
    struct ab_worker_pool *pool = nullptr;  // the persistent host threads that drive `workers` (ab_parallel_frames)
    int register_workers = 12;
    // AB_STACK_EXACT=1: use the direct re-summing clipping engine (cross-check of the fast one)
    bool stack_exact = false;
    // stacks of more than this many frames take the workgroup-per-pixel kernels (stack_deep.hip / scms_deep_kernel); read from
    // AB_STACK_DEEP_FROM / AB_BATCH_DEEP_FROM when the context is created: the tests lower them to hold those kernels to the oracle
    int stack_deep_from = 4096, batch_deep_from = 2048;
    // HIP events recorded on ctx->stream right around the stack kernels of the last ab_stack_* call (ab_stack_last_kernel_ms)
    hipEvent_t stack_ev[2] = {nullptr, nullptr};
    bool stack_ev_valid = false;
    hipEvent_t switch_ev = nullptr;  // orders the stream being left before the one switched to (ab_ctx_set_stream)
    // the frame-sharded stack in row chunks (sharded.hip): chunk k's all-reduces + division run on comm_stream while chunk k + 1 is
    // stacked on the context's stream; shard_ev[2 k] = chunk k stacked, [2 k + 1] = chunk k reduced; shard_tm = timing events
    // {first partial starts, last partial done} on the context's stream and {begin, end} per chunk on comm_stream
    hipStream_t comm_stream = nullptr;
    std::vector<hipEvent_t> shard_ev, shard_tm;
    int shard_chunks_timed = 0;
    bool stack_keep_counters = false;  // ab_stack_device: do not clear the rejection counters / restart the stack events (chunks 2 .. K of one stack)
    // the background-tile pipeline of a registration batch (detect.hip: ab_bg_pipeline_*): its own stream, one event per chunk
    // of frames, its own pinned result buffer (the context's general one may be reallocated by the reference's detection)
    hipStream_t aux_stream = nullptr;
    hipStream_t warp_stream = nullptr;  // the warps of a registration batch (affine.hip): off the workers' own streams
    std::vector<hipEvent_t> aux_events;
    hipStream_t pct_stream = nullptr;  // the fed pipeline's percentile launches (ab_bg_pipeline_begin_fed)
    std::vector<hipEvent_t> pct_events;
    // frames that arrive from the host inside a registration call (ab_align_pairs_affine with on_device = 0 targets): the copy
    // stream, one event per frame, the HBM staging area
    hipStream_t upload_stream = nullptr;
    std::vector<hipEvent_t> upload_events;
    void *upload_buf = nullptr;
    size_t upload_bytes = 0;
    void *aux_pinned = nullptr;
    size_t aux_pinned_bytes = 0;
    // the tiles the streaming tile kernel declined (detect.hip): {count, finished blocks, tile ids ...} per stream it is launched
    // on ([0] the context's stream, [1] the auxiliary one); zeroed once, the fallback kernel leaves it zeroed
    // detect.hip's round-4 forms, kept as cross-checks (read from AB_LABEL_LEGACY / AB_DETECT_FULL_RECORDS when the context is created,
    // inherited by its workers): two-pass labelling instead of the tile-local union-find; every component's record instead of the
    // device-side selection of the brightest
    bool label_legacy = false, label_pixelwise = false, detect_no_recs = false, detect_full_records = false, detect_midjoin = false;
    std::atomic<uint64_t> fallbacks[AB_FB_COUNT] = {};  // ab_ctx_fallback_counts: kept in the ROOT context (a worker's events are added to its parent's)
    unsigned int *tile_fail[2] = {nullptr, nullptr};
    size_t tile_fail_cap[2] = {0, 0};
    // progress / cancel (infra/progress.rs:39-74): the callback is serialised by progress_mu (frame workers tick it too);
    // worker contexts forward to their parent
    ab_progress_cb progress_cb = nullptr;
    void *progress_user = nullptr;
    std::atomic<int> cancel{0};
    std::mutex progress_mu;
    std::mutex err_mu;  // ab_set_error may be reached from the reference-preparation thread and the caller at once (affine.hip)
    ab_ctx *parent = nullptr;
    // the register-resident statistics kernel (stats_resident.hpp): its barrier flags carry epochs that grow from call to call, so
    // they are cleared only when the workspace is new, after an abort, or before the 32-bit epoch wraps
    const void *stats_bar = nullptr;
    unsigned int stats_epoch = 0;
    unsigned long long stats_expect = 0;  // the completion marker the last resident launch writes beside its result
    int stats_aborts = 0;                 // consecutive aborted resident launches: three in a row switch this context to the chain for good
    // phase_corr.hip: the (rows, cols) the two table sets in AB_WS_PHASE_TABLES were built for, and the workspace they live in
    int pc_tab_dims[2][2] = {{0, 0}, {0, 0}};
    const void *pc_tab_ws = nullptr;
};

// stage boundary: AB_ERR_CANCELLED ("Operation cancelled") if the host asked to stop, else ticks the callback (if any)
int ab_progress(ab_ctx *ctx, const char *stage, uint64_t current, uint64_t total);

int ab_set_error(ab_ctx *ctx, int code, const char *fmt, ...);

// ---- environment -------------------------------------------------------------------------------------------------------------------
// The RELEASE library reads eight environment variables -- AB_TRACE, AB_STACK_EXACT, AB_STATS_CHAIN, AB_REGISTER_WORKERS,
// AB_STACK_DEEP_FROM, AB_BATCH_DEEP_FROM, AB_COMM_TIMEOUT_MS, AB_COMM_HOST_SLOT_MB: the list in include/astroburst_hip.h
// ("Environment") and INTEGRATION.md -- through ab_env().  Every other switch (superseded forms kept as cross-checks, sweep knobs,
// fault injection, stage cuts) goes through ab_dev_env(), which answers "unset" unless the library was built with -DAB_DEV_ABLATION
// (`make dev` -> libastroburst_hip_dev.so): a drop-in for a desktop application must not change its numerics path with variables
// nobody documented.  tests/test_abi_cpu.py scans the release library's strings for any other AB_* name.
static inline const char *ab_env(const char *name) { return getenv(name); }
#ifdef AB_DEV_ABLATION
static inline const char *ab_dev_env(const char *name) { return getenv(name); }
#define AB_DEV_NAME(s) s
#else
static inline const char *ab_dev_env(const char *) { return nullptr; }
#define AB_DEV_NAME(s) nullptr  // (a developer variable's NAME handed to a helper: not even the string reaches the release library)
#endif

// one more event of a fallback kind (include/astroburst_hip.h: ab_fallback_kind), counted in the root context
static inline void ab_count_fallback(ab_ctx *ctx, int kind, uint64_t n = 1) {
    if (!ctx || kind < 0 || kind >= AB_FB_COUNT || n == 0) return;
    while (ctx->parent) ctx = ctx->parent;
    ctx->fallbacks[kind].fetch_add(n, std::memory_order_relaxed);
}

// AB_UPLOAD_TRACE=1 (developer knob): a timeline of a host-fed registration call on stderr, milliseconds since the call began
inline std::chrono::steady_clock::time_point &ab_trace_t0() {
    static std::chrono::steady_clock::time_point t = std::chrono::steady_clock::now();
    return t;
}
inline bool ab_upload_trace_on() {
    static const bool on = ab_dev_env("AB_UPLOAD_TRACE") != nullptr;
    return on;
}
inline void ab_upload_trace(const char *what, long a, long b = -1) {
    if (!ab_upload_trace_on()) return;
    const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - ab_trace_t0()).count();
    fprintf(stderr, "[ab timeline] %8.3f ms  %s %ld %ld\n", ms, what, a, b);
}

// A non-blocking stream, optionally confined to a subset of the compute units: `env` names an environment variable holding a 32-bit
// hex pattern that is repeated over the chip's CU mask (0x55555555 = every other CU).  Unset / 0 / ffffffff: an ordinary stream.
// prio_env / prio_default: the stream's priority (-1 high, 0 normal, 1 low; clamped to the device's range).  The runtime multiplexes
// all streams of one priority onto four in-order hardware queues; a stream of another priority gets a queue of another pool, so
// a stream that holds long-running or long-waiting packets (the tile kernels, the upload's copy barriers) no longer stands in
// front of a quarter of the worker streams' kernels.
inline hipError_t ab_stream_create_masked(ab_ctx *ctx, hipStream_t *out, const char *env, const char *prio_env = nullptr, int prio_default = 0) {
    const char *v = env ? ab_dev_env(env) : nullptr;
    const uint32_t pat = v ? (uint32_t)strtoul(v, nullptr, 16) : 0u;
    if (pat == 0u || pat == 0xffffffffu) {
        const char *pv = prio_env ? ab_dev_env(prio_env) : nullptr;
        int prio = pv ? atoi(pv) : prio_default;
        if (prio == 0) return hipStreamCreateWithFlags(out, hipStreamNonBlocking);
        int least = 0, greatest = 0;  // (numerically: greatest priority <= least priority)
        if (hipDeviceGetStreamPriorityRange(&least, &greatest) != hipSuccess) return hipStreamCreateWithFlags(out, hipStreamNonBlocking);
        prio = prio < greatest ? greatest : (prio > least ? least : prio);
        return hipStreamCreateWithPriority(out, hipStreamNonBlocking, prio);
    }
    const int words = (ctx->cu_count > 0 ? ctx->cu_count + 31 : 256) / 32;
    uint32_t mask[16];
    for (int i = 0; i < 16; ++i) mask[i] = pat;
    return hipExtStreamCreateWithCUMask(out, (uint32_t)(words > 16 ? 16 : words), mask);
}

#define AB_HIP(ctx, call)                                                                        \
    do {                                                                                         \
        hipError_t e_ = (call);                                                                  \
        if (e_ != hipSuccess)                                                                    \
            return ab_set_error((ctx), AB_ERR_HIP, "%s failed: %s (%s:%d)", #call,               \
                                hipGetErrorString(e_), __FILE__, __LINE__);                      \
    } while (0)

#define AB_CHECK(ctx, cond, ...)                                         \
    do {                                                                 \
        if (!(cond)) return ab_set_error((ctx), AB_ERR_INVALID, __VA_ARGS__); \
    } while (0)

#define AB_TRY(expr)              \
    do {                          \
        int rc_ = (expr);         \
        if (rc_ != AB_OK) return rc_; \
    } while (0)

// Exception barrier of the C ABI (tools/add_exception_barrier.py wraps every entry point in a function-try-block): nothing may
// unwind through `extern "C"` into a host built with panic = "abort" (Cargo.toml:64).
int ab_catch(ab_ctx *ctx, const char *fn);
#define AB_CATCH(ctx) catch (...) { return ab_catch((ctx), __func__); }
#define AB_CATCH_NOCTX catch (...) { return ab_catch(nullptr, __func__); }

// Scratch arena: returns a device pointer valid until the next ab_scratch() call with a larger size.
extern thread_local std::string *ab_tls_error_sink;  // see ab_set_error
int ab_scratch(ab_ctx *ctx, size_t bytes, void **out);
int ab_pinned(ab_ctx *ctx, size_t bytes, void **out);
// persistent workspace `slot` of at least `bytes` (contents are undefined after a growth)
int ab_workspace(ab_ctx *ctx, int slot, size_t bytes, void **out);

// Frame-parallel fan-out: items 0..n-1 are pulled by up to ctx->register_workers host threads, each driving a child context
// (own non-blocking stream + workspaces, cached in ctx->workers).  ctx->stream is drained first (unless the caller did and keeps
// using it concurrently); each worker's stream is drained
// before return.  With one worker fn runs inline on ctx.
// prologue (nullable): run once, on one more pool thread, concurrently with the frame loop -- work on the caller's context that
// overlaps the workers' frames (the reference frame's detection in a registration batch)
int ab_parallel_frames(ab_ctx *ctx, size_t n, const char *what, const std::function<int(ab_ctx *, size_t)> &fn, bool drain_caller_stream = true,
                       const std::function<void()> *prologue = nullptr);

// RAII staging of an input plane: host planes are uploaded to a temporary device buffer.
struct StagedPlane {
    const float *dptr = nullptr;
    int64_t rows = 0, cols = 0;
    void *owned = nullptr;
};
int ab_stage_in(ab_ctx *ctx, const ab_plane *p, StagedPlane *out);
void ab_stage_release(ab_ctx *ctx, StagedPlane *p);

// Output staging: device planes are written in place; host planes get a temp that is
// downloaded by ab_stage_out_finish.
struct StagedOut {
    float *dptr = nullptr;
    void *owned = nullptr;
    float *host = nullptr;
    size_t bytes = 0;
};
int ab_stage_out_begin(ab_ctx *ctx, const ab_plane_mut *p, StagedOut *out);
int ab_stage_out_finish(ab_ctx *ctx, StagedOut *o);  // downloads (sync) + frees when host
void ab_stage_out_abort(ab_ctx *ctx, StagedOut *o);

// plane_select.hip: exact order statistics of {v : finite, v > min_valid, (mask == nullptr || mask < 0.5)};
// keys are v or, with use_dev, |v - center| (f32).
struct ab_plane_sel {
    const float *data = nullptr;
    const float *mask = nullptr;
    int64_t n = 0;
    float min_valid = 0.0f;
    int use_dev = 0;
    float center = 0.0f;
};
// count, the [count/2] element and (want_lower, even count) the [count/2 - 1] element
int ab_plane_order_stats(ab_ctx *ctx, const ab_plane_sel &s, int want_lower, uint64_t *count_out, float *mid_out, float *lower_out);
// the general form: ranks_of(count, ranks) fills up to max_ranks 0-based ranks (clamped to count - 1) and returns how many;
// vals[i] = the ranks[i]-th smallest candidate.  Nothing is written to vals when there are no candidates.
int ab_plane_select_ranks(ab_ctx *ctx, const ab_plane_sel &s, int max_ranks, const std::function<int(uint64_t, uint64_t *)> &ranks_of,
                          uint64_t *count_out, float *vals);
// median_f32_mut (math/median.rs:46-63) of the candidates; 0 when there are none
int ab_plane_median_f32(ab_ctx *ctx, const ab_plane_sel &s, float *out, uint64_t *count_out);

// device-level entry points shared between translation units
int ab_stats_device(ab_ctx *ctx, const float *data, int64_t n, int use_known, double known_min, double known_max,
                    ab_image_stats *out);
// the asynchronous statistics chain (stats.hip): nothing is synchronised; *result_dev / *tx_dev / *stf_dev point into the context's
// state block and are valid once the stream reaches this point.  comm joins row bands (n_total = the whole image's pixel count).
int ab_stats_enqueue(ab_ctx *ctx, ab_comm *comm, const float *data, int64_t n, int64_t n_total, int use_known, double known_min,
                     double known_max, const ab_auto_stf_config *stf_cfg, const ab_image_stats **result_dev, const void **tx_dev,
                     const ab_stf_params **stf_dev);
// apply_stf -> u8 with the transform (StfTx) read from device memory
int ab_stf_u8_device_tx(ab_ctx *ctx, const float *in, int64_t n, const void *tx_dev, uint8_t *out);
// detect_stars of G frames of one size in lockstep (one launch per step for all of them); bg[f] = {median, sigma} of frame f's background
struct ab_frame_cand;
int ab_detect_stars_group_device(ab_ctx *ctx, const float *const *imgs, int G, int64_t rows, int64_t cols, double sigma_threshold, const ab_pixel_xf *xf,
                                 const double (*bg)[2], size_t max_keep, std::vector<ab_detected_star> *stars /* [G] */,
                                 const struct ab_frame_cand *cand = nullptr /* [G], nullable: the frames' candidate lists (ab_bg_pipeline_cand) */);
int ab_detect_stars_device(ab_ctx *ctx, const float *img, int64_t rows, int64_t cols, int64_t ld, double sigma_threshold,
                           std::vector<ab_detected_star> *stars, double *bg_median_out, double *bg_sigma_out,
                           ab_pixel_xf xf = ab_pixel_xf(), size_t max_keep = (size_t)-1 /* only the brightest max_keep stars are wanted */,
                           bool normalize_first = false /* normalize_for_detection's transform is derived and applied on the device */,
                           const double *bg_known = nullptr /* {bg_median, bg_sigma} of estimate_background, if the caller has them */);
// estimate_background of a batch of equally sized planes as a pipeline: the tile kernel of `chunk` planes at a time on the
// context's auxiliary stream, an event after each chunk; ab_bg_pipeline_get blocks until plane i's chunk has run and reduces
// its tiles (any thread)
struct ab_bg_pipeline {
    bool on = false;
    const void *tiles = nullptr;  // pinned TileOut[n][ntiles]
    const hipEvent_t *events = nullptr;
    int ntiles = 0, chunk = 1;
    int first = 0;  // planes of the FIRST launch when it is smaller than the others (0: `chunk` like every launch): chunk of plane i = i < first ? 0 : 1 + (i - first) / chunk
    size_t n = 0;
    const ab_pixel_xf *xf_host = nullptr;  // fed pipeline: plane i's transform, valid once ab_bg_pipeline_get(i) has returned
    struct ab_bg_feed_impl *feed = nullptr;  // the feeder thread of a pipeline whose planes are still landing (detect.hip)
    // the candidate lists of plane i's tiles (DEVICE; valid once ab_bg_pipeline_get(i) has returned): entries at cand_ent + (i * ntiles
    // + tile) * cand_cap (uint2 {position in the tile, raw bits}), counts / cuts at [i * ntiles + tile].  nullptr: the pipeline left none
    // (tiles smaller than 256 px).  detect.hip: label_bgtile_body
    void *cand_ent = nullptr;
    unsigned int *cand_cnt = nullptr;
    float *cand_cut = nullptr;
    int cand_step = 0;  // the background tile's edge (256)
};
// plane i's lists, as ab_detect_stars_group_device takes them
struct ab_frame_cand {
    const void *ent = nullptr;
    const unsigned int *cnt = nullptr;
    const float *cut = nullptr;
};
static inline ab_frame_cand ab_bg_pipeline_cand(const ab_bg_pipeline *p, size_t i) {
    ab_frame_cand c;
    if (p && p->on && p->cand_ent) {
        c.ent = (const char *)p->cand_ent + i * (size_t)p->ntiles * (size_t)2048 * 8u;  // (kCandCap entries of 8 bytes: detect.hip)
        c.cnt = p->cand_cnt + i * (size_t)p->ntiles;
        c.cut = p->cand_cut + i * (size_t)p->ntiles;
    }
    return c;
}
void ab_bg_pipeline_end(ab_bg_pipeline *p);  // joins the feeder, if any (before the streams are drained / the planes released)
// the pipeline fed chunk by chunk: the percentiles run per chunk on the device (no host join before the first tile launch); with
// `landed` events (nullable; a null entry = the plane is complete already) a feeder thread enqueues a chunk when its planes have landed
// want_cand: whole 256-px tiles also leave their candidate lists (ab_bg_pipeline_cand) for the labelling pass of a registration batch
int ab_bg_pipeline_begin_fed(ab_ctx *ctx, const float *const *planes, size_t n, int64_t rows, int64_t cols, int chunk, const hipEvent_t *landed,
                             ab_bg_pipeline *p, bool want_cand = false);
int ab_bg_pipeline_begin(ab_ctx *ctx, const float *const *planes, size_t n, int64_t rows, int64_t cols, const ab_pixel_xf *xf, int chunk,
                         ab_bg_pipeline *p, bool want_cand = false);
int ab_bg_pipeline_get(ab_ctx *ctx, const ab_bg_pipeline *p, size_t i, double *bg /* [2] */);
// the percentile normalisation's parameters (xf->on = 0 where the reference returns image.clone())
int ab_normalize_params_device(ab_ctx *ctx, const float *img, int64_t len, ab_pixel_xf *xf);
// the same for n equally sized planes in two launches and one synchronisation
int ab_normalize_params_many_device(ab_ctx *ctx, const float *const *planes, size_t n, int64_t len, ab_pixel_xf *xf);

int ab_phase_correlate_device(ab_ctx *ctx, const float *ref, int64_t ref_rows, int64_t ref_cols, int64_t ref_ld, const float *tgt,
                              int64_t tgt_rows, int64_t tgt_cols, int64_t tgt_ld, double *dx, double *dy, double *confidence);
int ab_align_channel_affine_device(ab_ctx *ctx, const float *ref, const float *tgt, int64_t rows, int64_t cols, int num_threads,
                                   ab_affine_align_result *out);
int ab_shift_device(ab_ctx *ctx, const float *src, int64_t rows, int64_t cols, int64_t src_ld, double dy, double dx, float *out);
// affine.hip: align_channel_affine of n device-resident targets against one reference, optionally with the registered copies (whole
// frames, or rows [band_row0, band_row0 + band_rows) of them when band_rows >= 0) written as each frame is fitted
int ab_register_frames_device(ab_ctx *ctx, const float *ref, const float *const *targets, size_t n, int64_t rows, int64_t cols, int num_threads,
                              ab_affine_align_result *out, float *const *aligned, const hipEvent_t *landed, int64_t band_row0, int64_t band_rows);
int ab_warp_device(ab_ctx *ctx, const float *src, int64_t src_rows, int64_t src_cols, const double t[6], int64_t out_rows,
                   int64_t out_cols, float *out);
int ab_warp_rows_device(ab_ctx *ctx, const float *src, int64_t src_rows, int64_t src_cols, const double t[6], int64_t out_rows,
                        int64_t out_cols, int64_t row0, int64_t nrows, float *out);
int ab_resample_device(ab_ctx *ctx, const float *src, int64_t src_rows, int64_t src_cols, int64_t out_rows, int64_t out_cols,
                       float *out);

// comm.hip: wait for ctx->stream while a collective of `comm` may be in flight -- bounded (AB_COMM_TIMEOUT_MS) and watching
// RCCL's asynchronous error state; on failure the communicator is aborted and AB_ERR_COMM returned.  NULL / host-staged
// communicators: a plain hipStreamSynchronize.
int ab_comm_stream_wait(ab_ctx *ctx, ab_comm *comm);

// stack_wide.hip: 65 .. 512 frames, one wave per pixel (host tables of n plane pointers / strides; counters pre-cleared)
int ab_stack_wide_device(ab_ctx *ctx, const float *const *dplanes, const int64_t *ld, size_t n, int64_t rows, int64_t cols,
                         const ab_stack_config *cfg, float *out_dev, double *out_sum_dev, uint32_t *out_cnt_dev, bool median_only);
// 257 .. 512 contiguous frames, two lanes per pixel (stack_pair.hip); dplanes is a HOST array of n device pointers
// more than 4096 frames (any count): one workgroup per pixel, samples in global scratch (stack_deep.hip)
int ab_stack_deep_device(ab_ctx *ctx, const float *const *dplanes, const int64_t *ld, size_t n, int64_t rows, int64_t cols,
                         const ab_stack_config *cfg, float *out_dev, double *out_sum_dev, uint32_t *out_cnt_dev, bool median_only);
int ab_stack_pair_device(ab_ctx *ctx, const float *const *dplanes, size_t n, int64_t rows, int64_t cols, const ab_stack_config *cfg,
                         float *out_dev, bool median_only);
// stack_wide.hip: the pixels a 513 .. 1024-frame fast pass (stack_quad.hip, eight lanes per pixel) handed over -- 2048 lists of `cap`
// pixel indices each -- one wave per pixel, the oracle's arithmetic; leaves the lists empty.  table_dev: DEVICE array of >= n plane pointers
int ab_stack_wide_list_device(ab_ctx *ctx, const float *const *table_dev, size_t n, int64_t rows, int64_t cols, const ab_stack_config *cfg, float *out_dev,
                              bool median_only, unsigned int *list_count, const int *list, unsigned int cap);

static inline int ab_div_up(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }

// AB_TRACE=1: wall-clock stamps of the host-visible stages on stderr (developer aid)
struct ab_trace {
    bool on;
    std::chrono::steady_clock::time_point t0;
    explicit ab_trace(const char *what) : on(ab_env("AB_TRACE") != nullptr), t0(std::chrono::steady_clock::now()) {
        if (on) fprintf(stderr, "[ab_trace] %s:", what);
    }
    void mark(const char *stage) {
        if (!on) return;
        const auto t1 = std::chrono::steady_clock::now();
        fprintf(stderr, " %s %.3f ms;", stage, std::chrono::duration<double, std::milli>(t1 - t0).count());
        t0 = t1;
    }
    ~ab_trace() {
        if (on) fprintf(stderr, "\n");
    }
};
