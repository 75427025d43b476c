// Context, error reporting, HBM scratch and host<->device staging.
#include "ab_common.hpp"

#include <condition_variable>

#include <algorithm>
#include <atomic>
#include <exception>
#include <new>
#include <thread>

// A helper thread that works on a context another thread is calling into (detect.hip's feeder) must not write that context's
// error slot: it points this at a string of its own, and whoever waits for the helper reports the message.
thread_local std::string *ab_tls_error_sink = nullptr;

int ab_set_error(ab_ctx *ctx, int code, const char *fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (ab_tls_error_sink) {
        *ab_tls_error_sink = buf;
        return code;
    }
    if (ctx) {
        std::lock_guard<std::mutex> lk(ctx->err_mu);
        ctx->err = buf;
    }
    return code;
}

// the handler behind AB_CATCH: rethrows the exception in flight to classify it
int ab_catch(ab_ctx *ctx, const char *fn) {
    int code = AB_ERR_INVALID;
    char msg[512];
    try {
        throw;
    } catch (const std::bad_alloc &) {
        code = AB_ERR_NOMEM;
        snprintf(msg, sizeof msg, "%s: out of host memory (std::bad_alloc)", fn);
    } catch (const std::exception &e) {
        snprintf(msg, sizeof msg, "%s: internal error: %s", fn, e.what());
    } catch (...) {
        snprintf(msg, sizeof msg, "%s: internal error (unknown exception)", fn);
    }
    if (ctx) {
        try {
            std::lock_guard<std::mutex> lk(ctx->err_mu);
            ctx->err.assign(msg);
        } catch (...) {  // not even the message fits: the code still says what happened
        }
    }
    return code;
}

int ab_progress(ab_ctx *ctx, const char *stage, uint64_t current, uint64_t total) {
    ab_ctx *root = ctx;
    while (root->parent) root = root->parent;
    if (root->cancel.load(std::memory_order_relaxed)) return ab_set_error(ctx, AB_ERR_CANCELLED, "Operation cancelled");
    {
        std::lock_guard<std::mutex> lk(root->progress_mu);  // the callback and its user pointer are read under the lock that sets them
        if (root->progress_cb) root->progress_cb(stage, current, total, root->progress_user);
    }
    // a host that learns of a cancel while it is being ticked (bindings/mod.rs forwards ProgressHandle::is_cancelled() from the tick)
    // stops at THIS stage boundary, as the reference's `if p.is_cancelled()` after each tick does (background.rs:80-91)
    if (root->cancel.load(std::memory_order_relaxed)) return ab_set_error(ctx, AB_ERR_CANCELLED, "Operation cancelled");
    return AB_OK;
}

void ab_worker_pool_destroy(ab_ctx *ctx);

extern "C" {

int ab_ctx_set_progress_cb(ab_ctx *ctx, ab_progress_cb cb, void *user) try {
    if (!ctx) return AB_ERR_INVALID;
    std::lock_guard<std::mutex> lk(ctx->progress_mu);
    ctx->progress_cb = cb;
    ctx->progress_user = user;
    return AB_OK;
} AB_CATCH(ctx)

int ab_ctx_request_cancel(ab_ctx *ctx) try {
    if (!ctx) return AB_ERR_INVALID;
    ctx->cancel.store(1);
    return AB_OK;
} AB_CATCH(ctx)

int ab_ctx_clear_cancel(ab_ctx *ctx) try {
    if (!ctx) return AB_ERR_INVALID;
    ctx->cancel.store(0);
    return AB_OK;
} AB_CATCH(ctx)

// "+dev": built with -DAB_DEV_ABLATION (`make dev`): the developer switches of ab_dev_env() are live
#ifdef AB_DEV_ABLATION
const char *ab_version(void) { return "astroburst_hip 0.3.0 (gfx950) +dev"; }
#else
const char *ab_version(void) { return "astroburst_hip 0.3.0 (gfx950)"; }
#endif

int ab_ctx_create(int device_id, ab_ctx **out) try {
    if (!out) return AB_ERR_INVALID;
    *out = nullptr;
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count == 0) return AB_ERR_NO_DEVICE;
    if (device_id < 0 || device_id >= count) return AB_ERR_INVALID;
    ab_ctx *ctx = new (std::nothrow) ab_ctx();
    if (!ctx) return AB_ERR_NOMEM;
    ctx->device = device_id;
    hipDeviceProp_t prop;
    if (hipSetDevice(device_id) != hipSuccess || hipGetDeviceProperties(&prop, device_id) != hipSuccess) {
        delete ctx;
        return AB_ERR_HIP;
    }
    ctx->cu_count = prop.multiProcessorCount;
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        // kernels are built for gfx950 only; refuse loudly instead of failing at first launch
        delete ctx;
        return AB_ERR_NO_DEVICE;
    }
    if (hipStreamCreateWithFlags(&ctx->own_stream, hipStreamNonBlocking) != hipSuccess ||
        hipMalloc((void **)&ctx->counters, (AB_REJ_SLOTS + 8) * sizeof(unsigned long long)) != hipSuccess) {  // + spare slots (sharded.hip)
        if (ctx->own_stream) (void)hipStreamDestroy(ctx->own_stream);
        delete ctx;
        return AB_ERR_HIP;
    }
    ctx->stream = ctx->own_stream;
    const char *ex = ab_env("AB_STACK_EXACT");
    ctx->stack_exact = ex && ex[0] == '1';
    ctx->label_legacy = ab_dev_env("AB_LABEL_LEGACY") != nullptr;
    ctx->label_pixelwise = ab_dev_env("AB_LABEL_PIXELWISE") != nullptr;
    ctx->detect_no_recs = ab_dev_env("AB_DETECT_NO_RECS") != nullptr;
    ctx->detect_full_records = ab_dev_env("AB_DETECT_FULL_RECORDS") != nullptr;
    ctx->detect_midjoin = ab_dev_env("AB_DETECT_MIDJOIN") != nullptr;
    if (const char *e = ab_env("AB_STACK_DEEP_FROM")) ctx->stack_deep_from = std::min(4096, std::max(64, atoi(e)));
    if (const char *e = ab_env("AB_BATCH_DEEP_FROM")) ctx->batch_deep_from = std::min(2048, std::max(64, atoi(e)));
    if (const char *rw = ab_env("AB_REGISTER_WORKERS")) ctx->register_workers = std::max(1, atoi(rw));
    *out = ctx;
    return AB_OK;
} AB_CATCH_NOCTX

void ab_ctx_destroy(ab_ctx *ctx) {
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    ab_worker_pool_destroy(ctx);  // (the threads index ctx->workers)
    for (ab_ctx *w : ctx->workers) ab_ctx_destroy(w);
    ctx->workers.clear();
    if (ctx->scratch) (void)hipFree(ctx->scratch);
    if (ctx->pinned) (void)hipHostFree(ctx->pinned);
    if (ctx->counters) (void)hipFree(ctx->counters);
    if (ctx->sel_hist) (void)hipFree(ctx->sel_hist);
    for (int i = 0; i < AB_WS_SLOTS; ++i)
        if (ctx->ws[i]) (void)hipFree(ctx->ws[i]);
    for (hipEvent_t e : ctx->stack_ev)
        if (e) (void)hipEventDestroy(e);
    if (ctx->switch_ev) (void)hipEventDestroy(ctx->switch_ev);
    for (hipEvent_t e : ctx->shard_ev) (void)hipEventDestroy(e);
    for (hipEvent_t e : ctx->shard_tm) (void)hipEventDestroy(e);
    if (ctx->comm_stream) (void)hipStreamDestroy(ctx->comm_stream);
    for (hipEvent_t e : ctx->aux_events) (void)hipEventDestroy(e);
    for (hipEvent_t e : ctx->pct_events) (void)hipEventDestroy(e);
    for (hipEvent_t e : ctx->upload_events) (void)hipEventDestroy(e);
    if (ctx->pct_stream) (void)hipStreamDestroy(ctx->pct_stream);
    if (ctx->upload_stream) (void)hipStreamDestroy(ctx->upload_stream);
    if (ctx->upload_buf) (void)hipFree(ctx->upload_buf);
    if (ctx->aux_pinned) (void)hipHostFree(ctx->aux_pinned);
    for (int i = 0; i < 2; ++i)
        if (ctx->tile_fail[i]) (void)hipFree(ctx->tile_fail[i]);
    if (ctx->aux_stream) (void)hipStreamDestroy(ctx->aux_stream);
    if (ctx->warp_stream) (void)hipStreamDestroy(ctx->warp_stream);
    if (ctx->own_stream) (void)hipStreamDestroy(ctx->own_stream);
    delete ctx;
}

// Give back what the context has grown: the scratch arena, every workspace, the host-frame staging area (63 frames of 8192^2 are
// 17 GB that used to stay pinned to the context until ab_ctx_destroy -- ADVICE r4), of this context and of its frame workers.  The
// context stays usable; the next call that needs a buffer allocates it again (and re-initialises what it keeps in it: every user
// compares the pointer it gets with the one it initialised).  Blocks until the context's streams are idle.
int ab_ctx_trim(ab_ctx *ctx) try {
    if (!ctx) return AB_ERR_INVALID;
    AB_HIP(ctx, hipSetDevice(ctx->device));
    for (hipStream_t st : {ctx->stream, ctx->aux_stream, ctx->pct_stream, ctx->warp_stream, ctx->upload_stream, ctx->comm_stream})
        if (st || st == ctx->stream) AB_HIP(ctx, hipStreamSynchronize(st));
    // Every buffer is forgotten BEFORE its release is checked and every slot is visited whatever an earlier release returned
    // (an early return left dangling pointers and untrimmed slots); the first failure is what the call reports, with
    // its message in THIS context (a worker's own error text lives in the worker).
    int first_rc = AB_OK;
    std::string first_msg;
    auto note = [&](hipError_t e, const char *what) {
        if (e == hipSuccess || first_rc != AB_OK) return;
        first_rc = AB_ERR_HIP;
        first_msg = std::string(what) + " failed: " + hipGetErrorString(e);
    };
    auto drop = [&](void *&p, size_t &bytes, bool host, const char *what) {
        void *q = p;
        p = nullptr;
        bytes = 0;
        if (q) note(host ? hipHostFree(q) : hipFree(q), what);
    };
    for (ab_ctx *w : ctx->workers) {
        const int rc = ab_ctx_trim(w);
        if (rc != AB_OK && first_rc == AB_OK) {
            first_rc = rc;
            first_msg = std::string("worker context: ") + ab_last_error(w);
        }
    }
    drop(ctx->scratch, ctx->scratch_bytes, false, "hipFree(scratch)");
    for (int i = 0; i < AB_WS_SLOTS; ++i) drop(ctx->ws[i], ctx->ws_bytes[i], false, "hipFree(workspace)");
    ctx->pc_tab_ws = nullptr;  // (phase_corr.hip rebuilds its tables when the workspace pointer changes)
    ctx->stats_bar = nullptr;  // (stats.hip clears the resident kernel's barrier flags when its workspace is new)
    drop(ctx->upload_buf, ctx->upload_bytes, false, "hipFree(upload_buf)");
    // the pinned read-back buffers (they grow with the frame-group size: G * kSelCap selection records) and the declined-tile lists
    drop(ctx->pinned, ctx->pinned_bytes, true, "hipHostFree(pinned)");
    drop(ctx->aux_pinned, ctx->aux_pinned_bytes, true, "hipHostFree(aux_pinned)");
    for (int i = 0; i < 2; ++i) {
        void *p = ctx->tile_fail[i];
        size_t cap = 0;
        ctx->tile_fail[i] = nullptr;
        ctx->tile_fail_cap[i] = 0;
        drop(p, cap, false, "hipFree(tile_fail)");
    }
    if (first_rc != AB_OK) return ab_set_error(ctx, first_rc, "ab_ctx_trim: %s", first_msg.c_str());
    return AB_OK;
} AB_CATCH(ctx)

int ab_ctx_fallback_counts(ab_ctx *ctx, uint64_t *out, size_t cap, int reset) try {
    if (!ctx || (!out && cap)) return AB_ERR_INVALID;
    while (ctx->parent) ctx = ctx->parent;
    for (size_t k = 0; k < (size_t)AB_FB_COUNT; ++k) {
        const uint64_t v = reset ? ctx->fallbacks[k].exchange(0, std::memory_order_relaxed) : ctx->fallbacks[k].load(std::memory_order_relaxed);
        if (k < cap) out[k] = v;
    }
    return AB_OK;
} AB_CATCH(ctx)

// (a copy per calling thread, taken under the lock ab_set_error writes under: a frame worker may be recording an error while the
// caller reads the previous one; the pointer stays valid until this thread's next ab_last_error)
const char *ab_last_error(const ab_ctx *ctx) {
    if (!ctx) return "null context";
    thread_local std::string copy;
    {
        std::lock_guard<std::mutex> lk(const_cast<ab_ctx *>(ctx)->err_mu);
        copy = ctx->err;
    }
    return copy.c_str();
}

// State the context owns (rejection counters, scratch arena, defer lists, statistics block) may still be in use by asynchronous
// work queued on the stream being left: the new stream waits for an event recorded there, so calls stay ordered across a switch.
static int switch_stream(ab_ctx *ctx, hipStream_t next) {
    if (next == ctx->stream) return AB_OK;
    AB_HIP(ctx, hipSetDevice(ctx->device));
    if (!ctx->switch_ev) AB_HIP(ctx, hipEventCreateWithFlags(&ctx->switch_ev, hipEventDisableTiming));
    AB_HIP(ctx, hipEventRecord(ctx->switch_ev, ctx->stream));
    AB_HIP(ctx, hipStreamWaitEvent(next, ctx->switch_ev, 0));
    ctx->stream = next;
    return AB_OK;
}

int ab_ctx_set_stream(ab_ctx *ctx, void *hip_stream) try {
    if (!ctx) return AB_ERR_INVALID;
    // a NULL handle IS a stream: HIP's legacy default stream (what PyTorch uses by default)
    return switch_stream(ctx, (hipStream_t)hip_stream);
} AB_CATCH(ctx)

int ab_ctx_reset_stream(ab_ctx *ctx) try {
    if (!ctx) return AB_ERR_INVALID;
    return switch_stream(ctx, ctx->own_stream);
} AB_CATCH(ctx)

void *ab_ctx_get_stream(ab_ctx *ctx) { return ctx ? (void *)ctx->stream : nullptr; }

int ab_ctx_synchronize(ab_ctx *ctx) try {
    if (!ctx) return AB_ERR_INVALID;
    AB_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return AB_OK;
} AB_CATCH(ctx)

int ab_device_alloc(ab_ctx *ctx, size_t bytes, void **out_dptr) try {
    if (!ctx || !out_dptr) return AB_ERR_INVALID;
    AB_HIP(ctx, hipSetDevice(ctx->device));
    AB_HIP(ctx, hipMalloc(out_dptr, bytes ? bytes : 1));
    return AB_OK;
} AB_CATCH(ctx)

int ab_device_free(ab_ctx *ctx, void *dptr) try {
    if (!ctx) return AB_ERR_INVALID;
    if (dptr) AB_HIP(ctx, hipFree(dptr));
    return AB_OK;
} AB_CATCH(ctx)

int ab_upload(ab_ctx *ctx, void *dst_device, const void *src_host, size_t bytes) try {
    if (!ctx) return AB_ERR_INVALID;
    AB_HIP(ctx, hipMemcpyAsync(dst_device, src_host, bytes, hipMemcpyHostToDevice, ctx->stream));
    AB_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return AB_OK;
} AB_CATCH(ctx)

int ab_download(ab_ctx *ctx, void *dst_host, const void *src_device, size_t bytes) try {
    if (!ctx) return AB_ERR_INVALID;
    AB_HIP(ctx, hipMemcpyAsync(dst_host, src_device, bytes, hipMemcpyDeviceToHost, ctx->stream));
    AB_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return AB_OK;
} AB_CATCH(ctx)

int ab_device_info(ab_ctx *ctx, char *name, size_t name_cap, int *cu_count, uint64_t *hbm_bytes) try {
    if (!ctx) return AB_ERR_INVALID;
    hipDeviceProp_t prop;
    AB_HIP(ctx, hipGetDeviceProperties(&prop, ctx->device));
    if (name && name_cap) snprintf(name, name_cap, "%s (%s)", prop.name, prop.gcnArchName);
    if (cu_count) *cu_count = prop.multiProcessorCount;
    if (hbm_bytes) *hbm_bytes = (uint64_t)prop.totalGlobalMem;
    return AB_OK;
} AB_CATCH(ctx)

}  // extern "C"

int ab_scratch(ab_ctx *ctx, size_t bytes, void **out) {
    if (bytes > ctx->scratch_bytes) {
        if (ctx->scratch) {
            AB_HIP(ctx, hipStreamSynchronize(ctx->stream));
            AB_HIP(ctx, hipFree(ctx->scratch));
            ctx->scratch = nullptr;
            ctx->scratch_bytes = 0;
        }
        size_t want = bytes + (bytes >> 2);
        AB_HIP(ctx, hipMalloc(&ctx->scratch, want));
        ctx->scratch_bytes = want;
    }
    *out = ctx->scratch;
    return AB_OK;
}

int ab_workspace(ab_ctx *ctx, int slot, size_t bytes, void **out) {
    if (slot < 0 || slot >= AB_WS_SLOTS) return ab_set_error(ctx, AB_ERR_INVALID, "bad workspace slot %d", slot);
    if (bytes > ctx->ws_bytes[slot]) {
        if (ctx->ws[slot]) {
            AB_HIP(ctx, hipStreamSynchronize(ctx->stream));
            AB_HIP(ctx, hipFree(ctx->ws[slot]));
            ctx->ws[slot] = nullptr;
            ctx->ws_bytes[slot] = 0;
        }
        AB_HIP(ctx, hipMalloc(&ctx->ws[slot], bytes));
        ctx->ws_bytes[slot] = bytes;
    }
    *out = ctx->ws[slot];
    return AB_OK;
}

int ab_pinned(ab_ctx *ctx, size_t bytes, void **out) {
    if (bytes > ctx->pinned_bytes) {
        if (ctx->pinned) {
            AB_HIP(ctx, hipStreamSynchronize(ctx->stream));
            AB_HIP(ctx, hipHostFree(ctx->pinned));
            ctx->pinned = nullptr;
            ctx->pinned_bytes = 0;
        }
        AB_HIP(ctx, hipHostMalloc(&ctx->pinned, bytes, hipHostMallocDefault));
        ctx->pinned_bytes = bytes;
    }
    *out = ctx->pinned;
    return AB_OK;
}

int ab_stage_in(ab_ctx *ctx, const ab_plane *p, StagedPlane *out) {
    AB_CHECK(ctx, p && p->data && p->rows > 0 && p->cols > 0, "plane is null or has a zero dimension");
    out->rows = p->rows;
    out->cols = p->cols;
    if (p->on_device) {
        out->dptr = p->data;
        out->owned = nullptr;
        return AB_OK;
    }
    size_t bytes = (size_t)p->rows * (size_t)p->cols * sizeof(float);
    void *d = nullptr;
    AB_HIP(ctx, hipMalloc(&d, bytes));
    hipError_t e = hipMemcpyAsync(d, p->data, bytes, hipMemcpyHostToDevice, ctx->stream);
    if (e != hipSuccess) {
        (void)hipFree(d);
        return ab_set_error(ctx, AB_ERR_HIP, "H2D copy failed: %s", hipGetErrorString(e));
    }
    out->dptr = (const float *)d;
    out->owned = d;
    return AB_OK;
}

void ab_stage_release(ab_ctx *ctx, StagedPlane *p) {
    if (p->owned) {
        (void)hipStreamSynchronize(ctx->stream);
        (void)hipFree(p->owned);
        p->owned = nullptr;
    }
}

int ab_stage_out_begin(ab_ctx *ctx, const ab_plane_mut *p, StagedOut *out) {
    AB_CHECK(ctx, p && p->data && p->rows > 0 && p->cols > 0, "output plane is null or has a zero dimension");
    out->bytes = (size_t)p->rows * (size_t)p->cols * sizeof(float);
    if (p->on_device) {
        out->dptr = p->data;
        out->owned = nullptr;
        out->host = nullptr;
        return AB_OK;
    }
    void *d = nullptr;
    AB_HIP(ctx, hipMalloc(&d, out->bytes));
    out->dptr = (float *)d;
    out->owned = d;
    out->host = p->data;
    return AB_OK;
}

int ab_stage_out_finish(ab_ctx *ctx, StagedOut *o) {
    if (o->owned) {
        hipError_t e = hipMemcpyAsync(o->host, o->dptr, o->bytes, hipMemcpyDeviceToHost, ctx->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
        (void)hipFree(o->owned);
        o->owned = nullptr;
        if (e != hipSuccess) return ab_set_error(ctx, AB_ERR_HIP, "D2H copy failed: %s", hipGetErrorString(e));
    }
    return AB_OK;
}

void ab_stage_out_abort(ab_ctx *ctx, StagedOut *o) {
    if (o->owned) {
        (void)hipStreamSynchronize(ctx->stream);
        (void)hipFree(o->owned);
        o->owned = nullptr;
    }
}

// The host threads behind ab_parallel_frames live as long as their context: a registration call used to create and join 12 + 1
// std::threads (~0.5 ms of a 17 ms stage); now a call publishes a job and the parked threads pick it up.
struct ab_worker_pool {
    std::mutex mu;
    std::condition_variable cv_start, cv_done;
    std::vector<std::thread> threads;
    uint64_t generation = 0;
    bool stop = false;
    // the job in flight (valid while remaining > 0)
    const std::function<int(ab_ctx *, size_t)> *fn = nullptr;
    const std::function<void()> *prologue = nullptr;
    const char *what = "";
    size_t n = 0, active = 0, remaining = 0;
    std::atomic<size_t> next{0};
    std::vector<int> rcs;
};

static void pool_thread(ab_ctx *ctx, ab_worker_pool *p, size_t t) {
    uint64_t seen = 0;
    bool device_ok = hipSetDevice(ctx->device) == hipSuccess;
    for (;;) {
        {
            std::unique_lock<std::mutex> lk(p->mu);
            p->cv_start.wait(lk, [&] { return p->stop || p->generation != seen; });
            if (p->stop) return;
            seen = p->generation;
            if (t >= p->active) continue;  // this job uses fewer workers
        }
        int rc = device_ok ? AB_OK : AB_ERR_HIP;
        if (p->prologue && t == p->active - 1) {  // the extra thread of a job with a prologue runs that and nothing else
            // ALWAYS run it, a thread without a device included: the prologue is what publishes the reference table the job's
            // workers block on (affine.hip: rt.publish), and its own first HIP call reports the device failure -- skipping it left
            // them waiting for ever
            (*p->prologue)();
            std::lock_guard<std::mutex> lk(p->mu);
            p->rcs[t] = rc;
            if (--p->remaining == 0) p->cv_done.notify_all();
            continue;
        }
        ab_ctx *wc = ctx->workers[t];
        if (rc == AB_OK) {
            for (size_t f = p->next.fetch_add(1); f < p->n; f = p->next.fetch_add(1)) {
                rc = ab_progress(wc, p->what, f + 1, p->n);  // per-frame tick; a cancel request stops the fan-out here
                if (rc == AB_OK) rc = (*p->fn)(wc, f);
                if (rc != AB_OK) break;  // (the stream is still drained below: nothing of this worker may be in flight when the caller cleans up)
            }
            if (hipStreamSynchronize(wc->stream) != hipSuccess && rc == AB_OK) rc = AB_ERR_HIP;
        }
        std::lock_guard<std::mutex> lk(p->mu);
        p->rcs[t] = rc;
        if (--p->remaining == 0) p->cv_done.notify_all();
    }
}

void ab_worker_pool_destroy(ab_ctx *ctx) {
    ab_worker_pool *p = ctx->pool;
    if (!p) return;
    {
        std::lock_guard<std::mutex> lk(p->mu);
        p->stop = true;
    }
    p->cv_start.notify_all();
    for (std::thread &th : p->threads) th.join();
    delete p;
    ctx->pool = nullptr;
}

int ab_parallel_frames(ab_ctx *ctx, size_t n, const char *what, const std::function<int(ab_ctx *, size_t)> &fn, bool drain_caller_stream,
                       const std::function<void()> *prologue) {
    const size_t workers = std::min<size_t>(n, (size_t)std::max(ctx->register_workers, 1));
    if (workers <= 1) {
        if (prologue) (*prologue)();
        for (size_t f = 0; f < n; ++f) {
            AB_TRY(ab_progress(ctx, what, f + 1, n));
            AB_TRY(fn(ctx, f));
        }
        return AB_OK;
    }
    while (ctx->workers.size() < workers) {
        ab_ctx *wc = nullptr;
        if (ab_ctx_create(ctx->device, &wc) != AB_OK) return ab_set_error(ctx, AB_ERR_HIP, "cannot create %s worker context", what);
        wc->register_workers = 1;
        wc->parent = ctx;
        // (Round 6, measured and not kept: HIGH-priority worker streams, so that no worker shares an in-order hardware queue with the tile
        // pipeline's stream -- the group whose worker does sits behind all tile launches, profiles/r06_register_timeline.txt -- made the
        // step 0.9 ms SLOWER, 10.5 against 9.6 ms: profiles/r06_priority_ab.txt.)
        wc->label_legacy = ctx->label_legacy;
        wc->label_pixelwise = ctx->label_pixelwise;
        wc->detect_no_recs = ctx->detect_no_recs;  // (the parent's choices, not the environment's at the time the pool grows)
        wc->detect_full_records = ctx->detect_full_records;
        wc->detect_midjoin = ctx->detect_midjoin;
        ctx->workers.push_back(wc);
    }
    if (drain_caller_stream) AB_HIP(ctx, hipStreamSynchronize(ctx->stream));  // whatever the caller queued on ctx (its frames, shared tables) is complete
    if (!ctx->pool) ctx->pool = new ab_worker_pool();
    ab_worker_pool *p = ctx->pool;
    const size_t active = workers + (prologue ? 1 : 0);
    while (p->threads.size() < active) {  // (only ever grows; the worker contexts it indexes exist above)
        const size_t t = p->threads.size();
        p->threads.emplace_back(pool_thread, ctx, p, t);
    }
    {
        std::unique_lock<std::mutex> lk(p->mu);
        p->fn = &fn;
        p->prologue = prologue;
        p->what = what;
        p->n = n;
        p->active = active;
        p->remaining = active;
        p->next.store(0);
        p->rcs.assign(p->threads.size(), AB_OK);
        ++p->generation;
        p->cv_start.notify_all();
        p->cv_done.wait(lk, [&] { return p->remaining == 0; });
    }
    for (size_t t = 0; t < workers; ++t)
        if (p->rcs[t] != AB_OK) return ab_set_error(ctx, p->rcs[t], "%s worker %zu: %s", what, t, ctx->workers[t]->err.c_str());
    if (prologue && p->rcs[workers] != AB_OK) return ab_set_error(ctx, p->rcs[workers], "%s: no device for the prologue thread", what);
    return AB_OK;
}
