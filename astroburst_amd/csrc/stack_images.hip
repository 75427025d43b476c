// stack_images driver (core/stacking/combine.rs:94-193): crop to the minimum dims, register
// frames 1..n-1 on frame 0 by phase correlation + bicubic sub-pixel shift, then per-pixel
// kappa-sigma combine.  Everything stays in HBM between the stages; the top-left crop
// (combine.rs:107-113) is only a row stride.
#include "ab_common.hpp"

#include <cmath>

int ab_stack_device(ab_ctx *ctx, const float *const *dplanes, const int64_t *ld, size_t n, int64_t rows, int64_t cols,
                    const ab_stack_config *cfg, float *out_dev, double *out_sum_dev, uint32_t *out_cnt_dev,
                    uint64_t *out_rejected, bool median_only);
int ab_phase_correlate_device(ab_ctx *ctx, const float *ref, int64_t ref_rows, int64_t ref_cols, int64_t ref_ld, const float *tgt,
                              int64_t tgt_rows, int64_t tgt_cols, int64_t tgt_ld, double *dx, double *dy, double *confidence);
int ab_phase_correlate_many_device(ab_ctx *ctx, const float *ref, int64_t ref_ld, const float *const *tgts, const int64_t *tgt_ld, size_t n, int64_t rows,
                                   int64_t cols, double *dx, double *dy, double *confidence);
int ab_shift_device(ab_ctx *ctx, const float *src, int64_t rows, int64_t cols, int64_t src_ld, double dy, double dx, float *out);

namespace {
int32_t round_to_i32(double v) {  // `result.offset.0.round() as i32` (combine.rs:135-136): saturating, NaN -> 0
    const double r = std::round(v);
    if (std::isnan(r)) return 0;
    if (r >= 2147483647.0) return INT32_MAX;
    if (r <= -2147483648.0) return INT32_MIN;
    return (int32_t)r;
}
}  // namespace

extern "C" int ab_stack_images(ab_ctx *ctx, const ab_plane *planes, size_t n, const ab_stack_config *cfg,
                               ab_plane_mut *out, int32_t *offsets_dy_dx, uint64_t *out_rejected) try {
    if (!ctx) return AB_ERR_INVALID;
    AB_CHECK(ctx, planes && n >= 1, "No images to stack");  // combine.rs:98-100
    AB_CHECK(ctx, cfg && out, "null config or output");
    int64_t min_rows = planes[0].rows, min_cols = planes[0].cols;  // combine.rs:104-105
    for (size_t i = 1; i < n; ++i) {
        min_rows = planes[i].rows < min_rows ? planes[i].rows : min_rows;
        min_cols = planes[i].cols < min_cols ? planes[i].cols : min_cols;
    }
    AB_CHECK(ctx, out->rows == min_rows && out->cols == min_cols, "output must be %lldx%lld (minimum frame dims)",
             (long long)min_rows, (long long)min_cols);
    if (offsets_dy_dx)
        for (size_t i = 0; i < 2 * n; ++i) offsets_dy_dx[i] = 0;  // combine.rs:121,140
    if (!cfg->align || n == 1) return ab_stack_sigma_clip(ctx, planes, n, cfg, out, out_rejected);

    // ---- align == true: PhaseCorrelation against frame 0 (combine.rs:123-138, align.rs:92-106) ----
    AB_HIP(ctx, hipSetDevice(ctx->device));
    std::vector<StagedPlane> st(n);
    std::vector<const float *> dp(n);
    std::vector<int64_t> ld(n);
    char *shifted_pool = nullptr;
    int rc = AB_OK;
    size_t staged = 0;
    for (; staged < n; ++staged) {
        rc = ab_stage_in(ctx, &planes[staged], &st[staged]);
        if (rc != AB_OK) break;
        dp[staged] = st[staged].dptr;
        ld[staged] = st[staged].cols;
    }
    const size_t plane_bytes = (size_t)min_rows * (size_t)min_cols * sizeof(float);
    // every frame against frame 0 as ONE batch per stage (AB_STACK_PAIRWISE=1: a phase_correlate call per pair, round 4's form)
    std::vector<double> sdx(n, 0.0), sdy(n, 0.0), scf(n, 0.0);
    static const bool pairwise = ab_dev_env("AB_STACK_PAIRWISE") != nullptr;
    if (rc == AB_OK && !pairwise)
        rc = ab_phase_correlate_many_device(ctx, dp[0], ld[0], dp.data() + 1, ld.data() + 1, n - 1, min_rows, min_cols, sdx.data() + 1, sdy.data() + 1, scf.data() + 1);
    for (size_t i = 1; rc == AB_OK && i < n; ++i) {
        double dx = sdx[i], dy = sdy[i], conf = scf[i];
        if (pairwise) rc = ab_phase_correlate_device(ctx, dp[0], min_rows, min_cols, ld[0], dp[i], min_rows, min_cols, ld[i], &dx, &dy, &conf);
        if (rc != AB_OK) break;
        if (offsets_dy_dx) {
            offsets_dy_dx[2 * i] = round_to_i32(dy);
            offsets_dy_dx[2 * i + 1] = round_to_i32(dx);
        }
        if (std::fabs(dy) < 1e-12 && std::fabs(dx) < 1e-12) continue;  // align.rs:37-39: clone -> read the frame in place
        if (!shifted_pool) {  // one grow-only workspace for all the registered copies (a hipMalloc + hipFree pair per frame cost 0.2 - 0.5 ms each)
            rc = ab_workspace(ctx, AB_WS_STACK_SHIFTED, (n - 1) * plane_bytes, (void **)&shifted_pool);
            if (rc != AB_OK) break;
        }
        float *dst = (float *)(shifted_pool + (i - 1) * plane_bytes);
        rc = ab_shift_device(ctx, dp[i], min_rows, min_cols, ld[i], dy, dx, dst);
        dp[i] = dst;
        ld[i] = min_cols;
    }
    StagedOut so;
    bool so_open = false;
    if (rc == AB_OK) {
        rc = ab_stage_out_begin(ctx, out, &so);
        so_open = (rc == AB_OK);
    }
    uint64_t rejected = 0;
    if (rc == AB_OK)
        rc = ab_stack_device(ctx, dp.data(), ld.data(), n, min_rows, min_cols, cfg, so.dptr, nullptr, nullptr, &rejected, false);
    if (rc == AB_OK) {
        rc = ab_stage_out_finish(ctx, &so);
        so_open = false;
    }
    if (so_open) ab_stage_out_abort(ctx, &so);
    for (size_t i = 0; i < staged; ++i) ab_stage_release(ctx, &st[i]);
    // The registered copies are (n - 1) planes -- 17 GB for 63 frames of 8192^2 -- in a grow-only workspace: a pool that large goes
    // back at the end of the call (a long-lived context would otherwise hold it beside every later call's buffers: ADVICE r5);
    // up to 1 GiB (C1: 31 MB) stays for the next call, which is what the workspace is for.
    if (shifted_pool && ctx->ws_bytes[AB_WS_STACK_SHIFTED] > ((size_t)1 << 30)) {
        (void)hipStreamSynchronize(ctx->stream);
        void *p = ctx->ws[AB_WS_STACK_SHIFTED];
        ctx->ws[AB_WS_STACK_SHIFTED] = nullptr;
        ctx->ws_bytes[AB_WS_STACK_SHIFTED] = 0;
        const hipError_t e = hipFree(p);
        if (e != hipSuccess && rc == AB_OK) rc = ab_set_error(ctx, AB_ERR_HIP, "hipFree(registered copies) failed: %s", hipGetErrorString(e));
    }
    if (rc == AB_OK && out_rejected) *out_rejected = rejected;
    return rc;
} AB_CATCH(ctx)
