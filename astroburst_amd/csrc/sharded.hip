// The hot path across the GPUs of one node: one context + one communicator rank per GPU.
//
// What the reference shards naturally:
//   * the per-pixel kappa-sigma loop (core/stacking/combine.rs:160-182: `par_chunks_mut(cols)`, rows are independent)
//     -> ROW BANDS.  Rank r stacks rows [row0_r, row0_r + nrows_r) of ALL n frames with the single-GPU kernel; the result is
//     the reference's single-level estimator bit for bit.  The only exchanged data are integers: the rejected-sample count
//     (one u64), and for the statistics of the stacked image min / max and three 65 536-bin histograms (stats.hip).
//   * per-frame whole-image work (align_channel_affine, core/alignment/affine.rs:129-212) -> BY FRAME (frame k on rank
//     k mod G); the estimates are exchanged as N x 80 bytes.
//   * BASELINE.json configs[3] names a third scheme: the FRAME SET is sharded (512 frames, 64 per GPU), every rank clips its
//     own frames to per-pixel (sum of survivors f64, count u32), the partials are all-reduced over xGMI and divided.  That
//     is a two-level estimator -- not the reference's -- checked against its own oracle (orc_stack_partial_noalign).
#include "ab_common.hpp"

#include <algorithm>

int ab_stack_device(ab_ctx *ctx, const float *const *dplanes, const int64_t *ld, size_t n, int64_t rows, int64_t cols,
                    const ab_stack_config *cfg, float *out_dev, double *out_sum_dev, uint32_t *out_cnt_dev, uint64_t *out_rejected,
                    bool median_only);

namespace {

__global__ void sum_counters_kernel(const unsigned long long *__restrict__ counters, int n, unsigned long long *__restrict__ out) {
    unsigned long long s = 0;
    for (int i = threadIdx.x; i < n; i += 64) s += counters[i];
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) s += __shfl_xor(s, off, 64);
    if (threadIdx.x == 0) *out = s;
}

// rejected samples of the stack just enqueued on ctx, summed over the ranks of comm
int total_rejected(ab_ctx *ctx, ab_comm *comm, uint64_t *out) {
    unsigned long long *slot = ctx->counters + AB_REJ_SLOTS;  // one spare u64 behind the per-wave counters
    hipLaunchKernelGGL(sum_counters_kernel, dim3(1), dim3(64), 0, ctx->stream, ctx->counters, AB_REJ_SLOTS, slot);
    AB_HIP(ctx, hipGetLastError());
    AB_TRY(ab_comm_allreduce(ctx, comm, slot, 1, AB_DT_U64, AB_RED_SUM));
    void *pin = nullptr;
    AB_TRY(ab_pinned(ctx, sizeof(unsigned long long), &pin));
    AB_HIP(ctx, hipMemcpyAsync(pin, slot, sizeof(unsigned long long), hipMemcpyDeviceToHost, ctx->stream));
    AB_TRY(ab_comm_stream_wait(ctx, comm));
    *out = *(const unsigned long long *)pin;
    return AB_OK;
}

}  // namespace

extern "C" {

// rank r of nranks owns rows [row0, row0 + nrows): ceil(rows / nranks) rows each, the last bands shorter (possibly empty)
int ab_shard_rows(int64_t rows, int nranks, int rank, int64_t *row0, int64_t *nrows) try {
    if (rows < 0 || nranks < 1 || rank < 0 || rank >= nranks || !row0 || !nrows) return AB_ERR_INVALID;
    const int64_t per = (rows + nranks - 1) / nranks;
    const int64_t lo = std::min<int64_t>(rows, per * rank), hi = std::min<int64_t>(rows, per * (rank + 1));
    *row0 = lo;
    *nrows = hi - lo;
    return AB_OK;
} AB_CATCH_NOCTX

// contiguous, balanced frame ranges: rank r gets frames [f0, f0 + nf)
int ab_shard_frames(size_t n_frames, int nranks, int rank, size_t *f0, size_t *nf) try {
    if (nranks < 1 || rank < 0 || rank >= nranks || !f0 || !nf) return AB_ERR_INVALID;
    const size_t base = n_frames / (size_t)nranks, extra = n_frames % (size_t)nranks;
    *f0 = (size_t)rank * base + std::min<size_t>((size_t)rank, extra);
    *nf = base + ((size_t)rank < extra ? 1 : 0);
    return AB_OK;
} AB_CATCH_NOCTX

// Rows [row0, row0 + out_band->rows) of stack_images' per-pixel loop over all n frames (combine.rs:160-182): the exact
// single-level estimator, restricted to a band.  Frames are device-resident and at least (row0 + band rows) x band cols.
int ab_stack_sigma_clip_rows(ab_ctx *ctx, const ab_plane *planes, size_t n, const ab_stack_config *cfg, int64_t row0,
                             ab_plane_mut *out_band, uint64_t *out_rejected) try {
    if (!ctx) return AB_ERR_INVALID;
    AB_CHECK(ctx, planes && n >= 1, "No images to stack");
    AB_CHECK(ctx, cfg && out_band, "null config or output");
    AB_CHECK(ctx, row0 >= 0 && out_band->rows >= 0 && out_band->cols > 0, "bad row band");
    if (out_band->rows == 0) {  // an empty band (more ranks than rows; its data pointer may be NULL): nothing to stack, nothing rejected
        if (out_rejected) *out_rejected = 0;
        AB_HIP(ctx, hipSetDevice(ctx->device));
        AB_HIP(ctx, hipMemsetAsync(ctx->counters, 0, AB_REJ_SLOTS * sizeof(unsigned long long), ctx->stream));
        return AB_OK;
    }
    AB_CHECK(ctx, out_band->data && out_band->on_device, "device band expected");
    std::vector<const float *> dp(n);
    std::vector<int64_t> ld(n);
    for (size_t i = 0; i < n; ++i) {
        AB_CHECK(ctx, planes[i].on_device && planes[i].data, "row-band stacking takes device-resident frames");
        AB_CHECK(ctx, planes[i].rows >= row0 + out_band->rows && planes[i].cols >= out_band->cols, "frame %zu is smaller than the band's extent", i);
        dp[i] = planes[i].data + row0 * planes[i].cols;
        ld[i] = planes[i].cols;
    }
    return ab_stack_device(ctx, dp.data(), ld.data(), n, out_band->rows, out_band->cols, cfg, out_band->data, nullptr, nullptr, out_rejected,
                           false);
} AB_CATCH(ctx)

// The same with the band taken from the communicator (ab_shard_rows over the minimum frame dims, combine.rs:104-113) and
// StackResult.rejected_pixels summed over the ranks.  out_band must hold this rank's rows x min cols.
int ab_stack_sigma_clip_rowband(ab_ctx *ctx, ab_comm *comm, const ab_plane *planes, size_t n, const ab_stack_config *cfg,
                                ab_plane_mut *out_band, uint64_t *out_rejected_total) try {
    if (!ctx) return AB_ERR_INVALID;
    // the local part: argument checks + this rank's band.  Its status is agreed on before the count is exchanged, so a rank
    // that fails here (bad band, out of memory, cancelled) fails the call on every rank instead of leaving them in RCCL
    auto local = [&]() -> int {
        AB_CHECK(ctx, planes && n >= 1, "No images to stack");
        AB_CHECK(ctx, cfg && out_band, "null config or output");
        int64_t min_rows = planes[0].rows, min_cols = planes[0].cols;
        for (size_t i = 1; i < n; ++i) {
            min_rows = std::min(min_rows, planes[i].rows);
            min_cols = std::min(min_cols, planes[i].cols);
        }
        int64_t row0 = 0, nrows = 0;
        AB_CHECK(ctx, ab_shard_rows(min_rows, ab_comm_size(comm), ab_comm_rank(comm), &row0, &nrows) == AB_OK, "bad communicator");
        AB_CHECK(ctx, out_band->rows == nrows && out_band->cols == min_cols, "this rank's band is %lld x %lld (got %lld x %lld)", (long long)nrows,
                 (long long)min_cols, (long long)out_band->rows, (long long)out_band->cols);
        return ab_stack_sigma_clip_rows(ctx, planes, n, cfg, row0, out_band, nullptr);
    };
    const int rc = local();
    if (!out_rejected_total) return rc;  // no collective in this call: nothing to agree on
    AB_TRY(ab_comm_agree(ctx, comm, rc));
    return total_rejected(ctx, comm, out_rejected_total);
} AB_CATCH(ctx)

// Frame-sharded two-level stack (BASELINE configs[3]): per-GPU partial over THIS rank's frames -> all-reduce of the
// per-pixel (sum f64, count u32) over xGMI -> divide.  `out` (device, rows x cols) is the full image on every rank.
// The partial planes live in the context (12 bytes per pixel).  Collective payload: 12 x rows x cols bytes, independent
// of the frame count.
//
// Round 6 (VERDICT r5 item 3 ii): the image is stacked in kShardChunks ROW CHUNKS, and chunk k's two all-reduces + its division run
// on the context's comm stream while chunk k + 1 is being stacked on its own stream -- at eight ranks the 201 MB exchange (a ring
// over xGMI, of the order of the local 64-frame stack itself) hides behind the stacking instead of following it.  Per pixel nothing
// changes (the same partial kernel on a band of rows, the same sums, the same division): bit-identical to the unchunked call.  The
// ranks agree on the status of the LOCAL CHECKS before anything is enqueued (the agreement is a host join: it no longer waits for
// the stack).  A host-staged communicator blocks the calling thread inside its all-reduce, so the next chunk's stack is enqueued
// first and the overlap is the same.  ab_stack_sharded_last_ms reports the two spans.
constexpr int kShardChunks = 4;
int ab_stack_sigma_clip_sharded(ab_ctx *ctx, ab_comm *comm, const ab_plane *local_planes, size_t n_local, const ab_stack_config *cfg,
                                ab_plane_mut *out, uint64_t *out_rejected_total) try {
    if (!ctx) return AB_ERR_INVALID;
    double *psum = nullptr;
    uint32_t *pcnt = nullptr;
    int64_t total = 0, rows = 0, cols = 0;
    std::vector<const float *> dp;
    std::vector<int64_t> ld;
    // local part first, then ONE agreement: a rank with an empty shard, a failed allocation or a cancel fails the call on
    // every rank before anybody enters the 12-bytes-per-pixel all-reduces
    auto local = [&]() -> int {
        AB_CHECK(ctx, local_planes && n_local >= 1, "every rank needs at least one frame (No images to stack)");
        AB_CHECK(ctx, cfg && out && out->data && out->on_device, "null config or output (device plane expected)");
        rows = out->rows;
        cols = out->cols;
        total = rows * cols;
        AB_CHECK(ctx, total > 0, "stack output has a zero dimension");
        dp.resize(n_local);
        ld.resize(n_local);
        for (size_t i = 0; i < n_local; ++i) {
            AB_CHECK(ctx, local_planes[i].on_device && local_planes[i].data, "partial stacking takes device-resident frames");
            AB_CHECK(ctx, local_planes[i].rows >= rows && local_planes[i].cols >= cols, "frame %zu is smaller than the output", i);
            dp[i] = local_planes[i].data;
            ld[i] = local_planes[i].cols;
        }
        AB_HIP(ctx, hipSetDevice(ctx->device));
        char *ws = nullptr;
        const size_t sum_bytes = (size_t)total * sizeof(double);
        AB_TRY(ab_workspace(ctx, AB_WS_SHARD, sum_bytes + (size_t)total * sizeof(uint32_t), (void **)&ws));
        psum = (double *)ws;
        pcnt = (uint32_t *)(ws + sum_bytes);
        if (!ctx->comm_stream) AB_HIP(ctx, hipStreamCreateWithFlags(&ctx->comm_stream, hipStreamNonBlocking));
        while (ctx->shard_ev.size() < (size_t)(2 * kShardChunks)) {
            hipEvent_t e;
            AB_HIP(ctx, hipEventCreateWithFlags(&e, hipEventDisableTiming));
            ctx->shard_ev.push_back(e);
        }
        while (ctx->shard_tm.size() < (size_t)(2 + 2 * kShardChunks)) {
            hipEvent_t e;
            AB_HIP(ctx, hipEventCreate(&e));
            ctx->shard_tm.push_back(e);
        }
        ab_ctx *root = ctx;
        while (root->parent) root = root->parent;
        if (root->cancel.load(std::memory_order_relaxed)) return ab_set_error(ctx, AB_ERR_CANCELLED, "Operation cancelled");
        return AB_OK;
    };
    int rc = local();
    // row chunks: whole rows, at least 16 of them each; a small image is one chunk
    const int K = (int)std::max<int64_t>(1, std::min<int64_t>(kShardChunks, rows / 16));
    auto chunk_rows = [&](int k, int64_t *r0, int64_t *nr) {
        const int64_t per = (rows + K - 1) / K;
        *r0 = std::min<int64_t>(rows, per * k);
        *nr = std::min<int64_t>(rows, per * (k + 1)) - *r0;
    };
    hipStream_t s0 = ctx->stream, s1 = ctx->comm_stream;
    struct Restore {  // whatever path leaves: the context's own stream and counters mode are back, and its stream is ordered behind the comm stream
        ab_ctx *c;
        hipStream_t s0;
        ~Restore() {
            c->stream = s0;
            c->stack_keep_counters = false;
        }
    } restore{ctx, s0};
    ctx->shard_chunks_timed = 0;
    auto enqueue_stack = [&](int k) -> int {
        int64_t r0, nr;
        chunk_rows(k, &r0, &nr);
        if (nr <= 0) return AB_OK;
        std::vector<const float *> cp(n_local);
        for (size_t i = 0; i < n_local; ++i) cp[i] = dp[i] + r0 * ld[i];
        ctx->stack_keep_counters = k > 0;
        const int rc = ab_stack_device(ctx, cp.data(), ld.data(), n_local, nr, cols, cfg, nullptr, psum + r0 * cols, pcnt + r0 * cols, nullptr, false);
        ctx->stack_keep_counters = false;
        AB_TRY(rc);
        AB_HIP(ctx, hipEventRecord(ctx->shard_ev[2 * k], s0));
        if (k == K - 1) AB_HIP(ctx, hipEventRecord(ctx->shard_tm[1], s0));
        return AB_OK;
    };
    // the FIRST chunk's stack is enqueued before the ranks agree: whatever can fail on the way to a launch (a workspace that cannot
    // grow, an unsupported frame set) fails the call on every rank, and the agreement's host join waits for a quarter of the stack
    if (rc == AB_OK) {
        const hipError_t e = hipEventRecord(ctx->shard_tm[0], s0);
        rc = e == hipSuccess ? enqueue_stack(0) : ab_set_error(ctx, AB_ERR_HIP, "hipEventRecord failed: %s", hipGetErrorString(e));
    }
    AB_TRY(ab_comm_agree(ctx, comm, rc));
    for (int k = 0; k < K && rc == AB_OK; ++k) {
        if (k + 1 < K) rc = enqueue_stack(k + 1);  // (in flight before this thread may block in a host-staged all-reduce of chunk k)
        if (rc != AB_OK) break;
        int64_t r0, nr;
        chunk_rows(k, &r0, &nr);
        if (nr <= 0) continue;
        const size_t off = (size_t)(r0 * cols), cnt = (size_t)(nr * cols);
        auto reduce_chunk = [&]() -> int {
            AB_HIP(ctx, hipStreamWaitEvent(s1, ctx->shard_ev[2 * k], 0));
            AB_HIP(ctx, hipEventRecord(ctx->shard_tm[2 + 2 * k], s1));
            AB_TRY(ab_comm_allreduce(ctx, comm, psum + off, cnt, AB_DT_F64, AB_RED_SUM));
            AB_TRY(ab_comm_allreduce(ctx, comm, pcnt + off, cnt, AB_DT_U32, AB_RED_SUM));
            AB_TRY(ab_stack_finalize_partial(ctx, psum + off, pcnt + off, (int64_t)cnt, out->data + off));
            AB_HIP(ctx, hipEventRecord(ctx->shard_tm[3 + 2 * k], s1));
            AB_HIP(ctx, hipEventRecord(ctx->shard_ev[2 * k + 1], s1));
            return AB_OK;
        };
        ctx->stream = s1;
        rc = reduce_chunk();
        ctx->stream = s0;
        if (rc == AB_OK) {
            AB_HIP(ctx, hipStreamWaitEvent(s0, ctx->shard_ev[2 * k + 1], 0));  // whatever follows on the context's stream sees the finished rows
            ctx->shard_chunks_timed = k + 1;
        }
    }
    if (rc != AB_OK) {  // (collectives may be in flight on the comm stream: drain both before the error goes up)
        (void)hipStreamSynchronize(s1);
        (void)hipStreamSynchronize(s0);
        return rc;
    }
    if (out_rejected_total) AB_TRY(total_rejected(ctx, comm, out_rejected_total));
    return AB_OK;
} AB_CATCH(ctx)

// the last ab_stack_sigma_clip_sharded of this context: the span of its partial stacks on the context's stream and the time inside
// its all-reduces + divisions on the comm stream (summed over the chunks; with overlap the two add up to more than the call took).
// Blocks until that call's work is done.
int ab_stack_sharded_last_ms(ab_ctx *ctx, float *stack_ms, float *comm_ms) try {
    if (!ctx) return AB_ERR_INVALID;
    AB_CHECK(ctx, stack_ms && comm_ms, "null argument");
    AB_CHECK(ctx, ctx->shard_chunks_timed > 0 && ctx->shard_tm.size() >= (size_t)(2 + 2 * ctx->shard_chunks_timed), "no sharded stack has run on this context");
    const int K = ctx->shard_chunks_timed;
    AB_HIP(ctx, hipEventSynchronize(ctx->shard_tm[1]));
    AB_HIP(ctx, hipEventSynchronize(ctx->shard_tm[3 + 2 * (K - 1)]));
    AB_HIP(ctx, hipEventElapsedTime(stack_ms, ctx->shard_tm[0], ctx->shard_tm[1]));
    float sum = 0.0f;
    for (int k = 0; k < K; ++k) {
        float ms = 0.0f;
        AB_HIP(ctx, hipEventElapsedTime(&ms, ctx->shard_tm[2 + 2 * k], ctx->shard_tm[3 + 2 * k]));
        sum += ms;
    }
    *comm_ms = sum;
    return AB_OK;
} AB_CATCH(ctx)

// Assemble row bands into the full image on every rank: `full` (device, total rows x cols) receives rank r's band at
// rows [row0_r, row0_r + nrows_r) (ab_shard_rows).  One broadcast per rank inside one RCCL group; bands may be unequal.
int ab_allgather_rows(ab_ctx *ctx, ab_comm *comm, const ab_plane *band, ab_plane_mut *full) try {
    if (!ctx) return AB_ERR_INVALID;
    const int size = ab_comm_size(comm), rank = ab_comm_rank(comm);
    auto local = [&]() -> int {
        AB_CHECK(ctx, band && full && full->data && full->on_device && (band->rows == 0 || (band->data && band->on_device)), "device planes expected");
        int64_t row0 = 0, nrows = 0;
        AB_CHECK(ctx, ab_shard_rows(full->rows, size, rank, &row0, &nrows) == AB_OK && nrows == band->rows && (nrows == 0 || band->cols == full->cols),
                 "band does not match this rank's share of the image");
        float *mine = full->data + row0 * full->cols;
        AB_HIP(ctx, hipSetDevice(ctx->device));
        if (nrows > 0 && band->data != mine)
            AB_HIP(ctx, hipMemcpyAsync(mine, band->data, (size_t)nrows * full->cols * sizeof(float), hipMemcpyDeviceToDevice, ctx->stream));
        return AB_OK;
    };
    AB_TRY(ab_comm_agree(ctx, comm, local()));
    if (!comm || size == 1) return AB_OK;
    AB_TRY(ab_comm_group_start());
    int rc = AB_OK;
    for (int r = 0; r < size && rc == AB_OK; ++r) {
        int64_t r0 = 0, nr = 0;
        ab_shard_rows(full->rows, size, r, &r0, &nr);
        if (nr > 0) rc = ab_comm_broadcast(ctx, comm, full->data + r0 * full->cols, (size_t)nr * full->cols * sizeof(float), r);
    }
    const int rc2 = ab_comm_group_end();
    if (rc != AB_OK) return rc;
    if (rc2 != AB_OK) return ab_set_error(ctx, AB_ERR_COMM, "ncclGroupEnd failed");
    return AB_OK;
} AB_CATCH(ctx)

// align_channel_affine(reference, targets[i]) for all i < n, the estimates computed by frame across the ranks (target i on
// rank i mod size) and exchanged: every rank receives all n results.  Targets this rank does not own are not read.
// The exchange is an all-reduce(SUM) of the results' bit patterns as u64 words over zero-initialised slots: adding zeros
// is the identity on integers, so every f64 arrives bit for bit.
int ab_register_frames_sharded(ab_ctx *ctx, ab_comm *comm, const ab_plane *reference, const ab_plane *targets, size_t n, int num_threads,
                               ab_affine_align_result *out) try {
    if (!ctx) return AB_ERR_INVALID;
    AB_CHECK(ctx, reference && targets && out, "null argument");
    const int size = ab_comm_size(comm), rank = ab_comm_rank(comm);
    std::vector<ab_plane> mine;
    std::vector<size_t> idx;
    for (size_t i = (size_t)rank; i < n; i += (size_t)size) {
        mine.push_back(targets[i]);
        idx.push_back(i);
    }
    std::vector<ab_affine_align_result> res(mine.size());
    void *dev = nullptr;
    const size_t bytes = n * sizeof *out;
    // this rank's estimates (a frame that cannot be read, a cancel, an allocation may fail here) and the exchange buffer; the
    // status is agreed on before the exchange, so a rank that failed or was cancelled fails the call on EVERY rank
    auto local = [&]() -> int {
        if (!mine.empty()) AB_TRY(ab_register_frames(ctx, reference, mine.data(), mine.size(), num_threads, res.data()));
        if (comm && size > 1 && n > 0) AB_TRY(ab_scratch(ctx, bytes, &dev));
        return AB_OK;
    };
    AB_TRY(ab_comm_agree(ctx, comm, local()));
    static_assert(sizeof(ab_affine_align_result) % 8 == 0, "results travel as u64 words");
    memset(out, 0, n * sizeof *out);
    for (size_t k = 0; k < idx.size(); ++k) out[idx[k]] = res[k];
    if (!comm || size == 1 || n == 0) return AB_OK;
    AB_HIP(ctx, hipMemcpyAsync(dev, out, bytes, hipMemcpyHostToDevice, ctx->stream));
    AB_TRY(ab_comm_allreduce(ctx, comm, dev, bytes / 8, AB_DT_U64, AB_RED_SUM));
    AB_HIP(ctx, hipMemcpyAsync(out, dev, bytes, hipMemcpyDeviceToHost, ctx->stream));
    AB_TRY(ab_comm_stream_wait(ctx, comm));
    return AB_OK;
} AB_CATCH(ctx)

// align_pair(reference, targets[i], Affine) for all i in the ROW-BAND scheme, as one call (round 6; VERDICT r5 item 3 i): this rank's
// rows [row0, row0 + out_bands[i].rows) of every registered frame.  targets[i] holds rows [target_row0[i], target_row0[i] + targets[i].rows)
// of target i (target_row0 nullable = whole frames): WHOLE for the targets this rank estimates (i mod size == rank), of the others at
// least the rows its band of the output reads (ab_shard_source_rows / ab_warp_source_rows; a band that is short is refused with the
// rows named).  The rank's own frames are warped -- its rows only -- the moment they are fitted, overlapped with the remaining estimates
// exactly like the full warps of ab_align_pairs_affine (one rank: the single-GPU call, no estimate -> exchange -> warp sequence any
// more); the estimates are then exchanged (80 bytes each) and the other ranks' frames are warped from their bands.
// Results and pixels equal ab_register_frames_sharded followed by ab_warp_image_rows(_from_band) bit for bit.
int ab_align_pairs_affine_rowband(ab_ctx *ctx, ab_comm *comm, const ab_plane *reference, const ab_plane *targets, const int64_t *target_row0, size_t n,
                                  int num_threads, int64_t row0, ab_affine_align_result *out, ab_plane_mut *out_bands) try {
    if (!ctx) return AB_ERR_INVALID;
    AB_CHECK(ctx, reference && out && ((targets && out_bands) || n == 0), "null argument");
    const int size = ab_comm_size(comm), rank = ab_comm_rank(comm);
    const int64_t rows = reference->rows, cols = reference->cols;
    std::vector<size_t> idx;
    std::vector<const float *> mine;
    std::vector<float *> bands;
    std::vector<ab_affine_align_result> res;
    void *dev = nullptr;
    const size_t bytes = n * sizeof *out;
    int64_t nrows = n ? out_bands[0].rows : 0;
    auto local = [&]() -> int {
        AB_CHECK(ctx, reference->data && reference->on_device, "the row-band registration takes device-resident planes");
        AB_CHECK(ctx, row0 >= 0 && nrows >= 0 && row0 + nrows <= rows, "row band [%lld, %lld) leaves the frame's %lld rows", (long long)row0,
                 (long long)(row0 + nrows), (long long)rows);
        for (size_t i = 0; i < n; ++i) {
            AB_CHECK(ctx, out_bands[i].rows == nrows && out_bands[i].cols == cols && (nrows == 0 || (out_bands[i].data && out_bands[i].on_device)),
                     "band %zu: every output band is %lld x %lld on the device", i, (long long)nrows, (long long)cols);
            AB_CHECK(ctx, targets[i].cols == cols && (targets[i].rows == 0 || (targets[i].data && targets[i].on_device)), "target %zu: device rows of %lld columns expected", i,
                     (long long)cols);
            if ((int)(i % (size_t)size) == rank) {
                AB_CHECK(ctx, targets[i].rows == rows && (!target_row0 || target_row0[i] == 0), "target %zu is estimated on this rank: the whole frame is needed here", i);
                idx.push_back(i);
                mine.push_back(targets[i].data);
                bands.push_back(out_bands[i].data);
            }
        }
        res.resize(idx.size());
        if (!mine.empty())
            AB_TRY(ab_register_frames_device(ctx, reference->data, mine.data(), mine.size(), rows, cols, num_threads, res.data(), bands.data(), nullptr, row0, nrows));
        if (comm && size > 1 && n > 0) AB_TRY(ab_scratch(ctx, bytes, &dev));
        return AB_OK;
    };
    AB_TRY(ab_comm_agree(ctx, comm, local()));
    memset(out, 0, n * sizeof *out);
    for (size_t k = 0; k < idx.size(); ++k) out[idx[k]] = res[k];
    if (!comm || size == 1 || n == 0) return AB_OK;
    static_assert(sizeof(ab_affine_align_result) % 8 == 0, "results travel as u64 words");
    AB_HIP(ctx, hipMemcpyAsync(dev, out, bytes, hipMemcpyHostToDevice, ctx->stream));
    AB_TRY(ab_comm_allreduce(ctx, comm, dev, bytes / 8, AB_DT_U64, AB_RED_SUM));
    AB_HIP(ctx, hipMemcpyAsync(out, dev, bytes, hipMemcpyDeviceToHost, ctx->stream));
    AB_TRY(ab_comm_stream_wait(ctx, comm));
    // the other ranks' frames: this rank's rows from what it holds of them (a local failure here -- a short band -- is this rank's alone:
    // no collective follows inside this call)
    for (size_t i = 0; i < n; ++i) {
        if ((int)(i % (size_t)size) == rank) continue;
        AB_TRY(ab_warp_image_rows_from_band(ctx, &targets[i], target_row0 ? target_row0[i] : 0, rows, out[i].transform, rows, row0, &out_bands[i]));
    }
    return AB_OK;
} AB_CATCH(ctx)

}  // extern "C"
