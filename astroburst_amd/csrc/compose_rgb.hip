// RGB composition pipeline on gfx950.
//
// Replaces core/compose/rgb.rs: harmonize_dimensions (:42-125), apply_multiplier_inplace (:127-130),
// channel_or_synth (:132-151), merge_for_stf (:153-163), align_channels (:165-189), the compose-local
// apply_stf_inplace (:191-207) and process_rgb (:209-323); core/compose/white_balance.rs:3-20 and
// core/alignment/pair.rs:41-77 (align_pair).
//
// The three channels stay resident in HBM from upload to the final planes: resampling, registration
// (phase correlation + bicubic shift, or star-based affine + warp), statistics, white balance, STF
// and SCNR are launched back to back on the context's stream; only the per-channel statistics and
// the registration scalars visit the host.  Every per-pixel map is a streaming f32 kernel at
// 4 B read + 4 B written per pixel.
#include "ab_common.hpp"

#include <algorithm>
#include <cfloat>
#include <cmath>

namespace {

constexpr int kBlock = 256;

int stream_grid(ab_ctx *ctx, int64_t n) {
    return (int)std::max<int64_t>(1, std::min<int64_t>((n + kBlock - 1) / kBlock, (int64_t)(ctx->cu_count > 0 ? ctx->cu_count : 256) * 8));
}

#define AB_GRID_LOOP(i, n) \
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x, stride_ = (int64_t)gridDim.x * kBlock; i < (n); i += stride_)

__global__ __launch_bounds__(kBlock) void avg2_kernel(const float *__restrict__ a, const float *__restrict__ b, int64_t n, float *__restrict__ out) {
    AB_GRID_LOOP(i, n) out[i] = (a[i] + b[i]) * 0.5f;  // rgb.rs:142-146
}

__global__ __launch_bounds__(kBlock) void merge3_kernel(const float *__restrict__ r, const float *__restrict__ g, const float *__restrict__ b,
                                                        int64_t n, float *__restrict__ out) {
    AB_GRID_LOOP(i, n) out[i] = (r[i] + g[i] + b[i]) * (1.0f / 3.0f);  // rgb.rs:157-160
}

__global__ __launch_bounds__(kBlock) void scale_inplace_kernel(float *__restrict__ data, int64_t n, float mult) {
    AB_GRID_LOOP(i, n) data[i] = data[i] * mult;  // rgb.rs:129
}

struct ComposeStf {
    double dmin, inv_range, shadow, clip_range, m;
};

__global__ __launch_bounds__(kBlock) void compose_stf_kernel(float *__restrict__ data, int64_t n, const ComposeStf t) {
    AB_GRID_LOOP(i, n) {  // rgb.rs:199-206
        const float v = data[i];
        float r;
        if (!__builtin_isfinite(v) || v <= 1e-7f) {
            r = 0.0f;
        } else {
            const double norm = ((double)v - t.dmin) * t.inv_range;
            double clipped = (norm - t.shadow) / t.clip_range;
            clipped = clipped < 0.0 ? 0.0 : (clipped > 1.0 ? 1.0 : clipped);
            if (clipped <= 0.0)
                r = 0.0f;
            else if (clipped >= 1.0)
                r = 1.0f;
            else
                r = (float)((t.m - 1.0) * clipped / ((2.0 * t.m - 1.0) * clipped - t.m));
        }
        data[i] = r;
    }
}

struct Scope {  // frees / aborts everything registered with it when the entry point returns
    ab_ctx *ctx;
    std::vector<StagedPlane *> ins;
    std::vector<StagedOut *> outs;
    std::vector<void *> dev;
    int next_slot = 0;
    explicit Scope(ab_ctx *c) : ctx(c) {}
    ~Scope() {
        for (StagedOut *o : outs) ab_stage_out_abort(ctx, o);
        if (!dev.empty()) (void)hipStreamSynchronize(ctx->stream);
        for (void *p : dev) (void)hipFree(p);
        for (StagedPlane *p : ins) ab_stage_release(ctx, p);
    }
    int alloc(float **p, int64_t n) {  // out of the context's scope slots (kept between calls); a ninth plane is allocated and freed
        if (next_slot < 8) return ab_workspace(ctx, AB_WS_SCOPE0 + next_slot++, std::max<size_t>((size_t)n, 1) * sizeof(float), (void **)p);
        AB_HIP(ctx, hipMalloc((void **)p, std::max<size_t>((size_t)n, 1) * sizeof(float)));
        dev.push_back(*p);
        return AB_OK;
    }
};

int copy_plane(ab_ctx *ctx, float *dst, const float *src, int64_t n) {
    if (n > 0 && dst != src) AB_HIP(ctx, hipMemcpyAsync(dst, src, (size_t)n * sizeof(float), hipMemcpyDeviceToDevice, ctx->stream));
    return AB_OK;
}

// channel_or_synth (rgb.rs:132-151) into dst
int channel_or_synth(ab_ctx *ctx, const float *primary, const float *alt1, const float *alt2, int64_t n, float *dst) {
    if (primary) return copy_plane(ctx, dst, primary, n);
    if (alt1 && alt2) {
        if (n > 0) {
            hipLaunchKernelGGL(avg2_kernel, dim3(stream_grid(ctx, n)), dim3(kBlock), 0, ctx->stream, alt1, alt2, n, dst);
            AB_HIP(ctx, hipGetLastError());
        }
        return AB_OK;
    }
    if (alt1) return copy_plane(ctx, dst, alt1, n);
    if (alt2) return copy_plane(ctx, dst, alt2, n);
    if (n > 0) AB_HIP(ctx, hipMemsetAsync(dst, 0, (size_t)n * sizeof(float), ctx->stream));
    return AB_OK;
}

}  // namespace

extern "C" {

int ab_select_wb_reference(const ab_image_stats *sr, const ab_image_stats *sg, const ab_image_stats *sb, double out[3]) try {
    if (!sr || !sg || !sb || !out) return AB_ERR_INVALID;
    auto stability = [](const ab_image_stats *s) { return s->median > 1e-10 ? s->mad / s->median : DBL_MAX; };
    const double stab_r = stability(sr), stab_g = stability(sg), stab_b = stability(sb);
    const double mr = std::fmax(sr->median, 1e-10), mg = std::fmax(sg->median, 1e-10), mb = std::fmax(sb->median, 1e-10);
    if (stab_r <= stab_g && stab_r <= stab_b) {
        out[0] = 1.0, out[1] = mr / mg, out[2] = mr / mb;
    } else if (stab_b <= stab_g) {
        out[0] = mb / mr, out[1] = mb / mg, out[2] = 1.0;
    } else {
        out[0] = mg / mr, out[1] = 1.0, out[2] = mg / mb;
    }
    return AB_OK;
} AB_CATCH_NOCTX

int ab_process_rgb(ab_ctx *ctx, const ab_plane *r, const ab_plane *g, const ab_plane *b, const ab_rgb_compose_config *cfg,
                   ab_plane_mut *out_r, ab_plane_mut *out_g, ab_plane_mut *out_b, ab_plane_mut *pre_r, ab_plane_mut *pre_g,
                   ab_plane_mut *pre_b, ab_processed_rgb_info *info) try {
    if (!ctx) return AB_ERR_INVALID;
    AB_CHECK(ctx, cfg && out_r && out_g && out_b && info, "null argument");
    const ab_plane *ch[3] = {r, g, b};
    ab_plane_mut *outs[3] = {out_r, out_g, out_b}, *pres[3] = {pre_r, pre_g, pre_b};
    const int count = (r != nullptr) + (g != nullptr) + (b != nullptr);
    memset(info, 0, sizeof *info);
    if (count < 2) return ab_set_error(ctx, AB_ERR_INVALID, "Need at least 2 channels for RGB compose (got %d)", count);  // :217
    // harmonize_dimensions (:42-125)
    int64_t min_rows = INT64_MAX, min_cols = INT64_MAX, max_rows = 0, max_cols = 0;
    for (int c = 0; c < 3; ++c)
        if (ch[c]) {
            min_rows = std::min(min_rows, ch[c]->rows), min_cols = std::min(min_cols, ch[c]->cols);
            max_rows = std::max(max_rows, ch[c]->rows), max_cols = std::max(max_cols, ch[c]->cols);
        }
    const bool resample = !(max_rows == min_rows && max_cols == min_cols);
    if (resample) {
        const double ratio = std::fmax((double)max_rows / (double)std::max<int64_t>(min_rows, 1),
                                       (double)max_cols / (double)std::max<int64_t>(min_cols, 1));
        if (ratio > 8.0) {  // MAX_DIMENSION_RATIO (types/constants.rs:165)
            char msg[256];
            int k = snprintf(msg, sizeof msg, "Channel dimension ratio %.1fx exceeds %.0fx limit.", ratio, 8.0);
            for (int c = 0; c < 3; ++c)
                if (ch[c] && k > 0 && k < (int)sizeof msg)
                    k += snprintf(msg + k, sizeof msg - k, " %c=%lldx%lld", "RGB"[c], (long long)ch[c]->cols, (long long)ch[c]->rows);
            if (k > 0 && k < (int)sizeof msg) snprintf(msg + k, sizeof msg - k, ". Check channel assignments.");
            return ab_set_error(ctx, AB_ERR_INVALID, "%s", msg);
        }
    }
    const int64_t rows = max_rows, cols = max_cols, n = rows * cols;
    for (int c = 0; c < 3; ++c) {
        AB_CHECK(ctx, outs[c]->rows == rows && outs[c]->cols == cols, "output planes must have the largest channel's dims (%lld x %lld)",
                 (long long)rows, (long long)cols);
        AB_CHECK(ctx, !pres[c] || (pres[c]->rows == rows && pres[c]->cols == cols), "pre-stretch planes must have the largest channel's dims");
    }
    AB_CHECK(ctx, n < (int64_t(1) << 31), "image too large for this build");
    AB_HIP(ctx, hipSetDevice(ctx->device));
    info->rows = (uint64_t)rows;
    info->cols = (uint64_t)cols;
    info->resampled = resample ? 1 : 0;

    StagedPlane in[3];
    StagedOut so[3], sp[3];
    Scope sc(ctx);
    const float *eff[3] = {nullptr, nullptr, nullptr};
    for (int c = 0; c < 3; ++c) {
        if (!ch[c]) continue;
        AB_TRY(ab_stage_in(ctx, ch[c], &in[c]));
        sc.ins.push_back(&in[c]);
        eff[c] = in[c].dptr;
        if (resample && !(ch[c]->rows == rows && ch[c]->cols == cols)) {  // :106-119
            float *h = nullptr;
            AB_TRY(sc.alloc(&h, n));
            AB_TRY(ab_resample_device(ctx, in[c].dptr, ch[c]->rows, ch[c]->cols, rows, cols, h));
            eff[c] = h;
        }
    }
    float *img[3];
    for (int c = 0; c < 3; ++c) {
        AB_TRY(ab_stage_out_begin(ctx, outs[c], &so[c]));
        sc.outs.push_back(&so[c]);
        img[c] = so[c].dptr;
    }
    const int alt[3][2] = {{1, 2}, {0, 2}, {0, 1}};
    const bool do_align = cfg->align != 0;  // && count >= 2 (always true here)
    for (int c = 0; c < 3; ++c) {
        if (do_align && c > 0 && ch[c]) continue;  // produced by the registration below
        AB_TRY(channel_or_synth(ctx, eff[c], eff[alt[c][0]], eff[alt[c][1]], n, img[c]));
    }
    if (do_align) {  // align_channels (:165-189) + align_pair (pair.rs:41-77)
        const float *ref_ch = eff[0] ? eff[0] : (eff[1] ? eff[1] : eff[2]);
        for (int c = 1; c < 3; ++c) {
            if (!ch[c]) continue;
            double *off = c == 1 ? info->offset_g : info->offset_b;
            if (cfg->align_method == 0) {
                double dx, dy, conf;
                AB_TRY(ab_phase_correlate_device(ctx, ref_ch, rows, cols, cols, eff[c], rows, cols, cols, &dx, &dy, &conf));
                AB_TRY(ab_shift_device(ctx, eff[c], rows, cols, cols, dy, dx, img[c]));
                off[0] = dy, off[1] = dx;
            } else {
                ab_affine_align_result ar;
                AB_TRY(ab_align_channel_affine_device(ctx, ref_ch, eff[c], rows, cols, cfg->num_threads > 0 ? cfg->num_threads : 1, &ar));
                AB_TRY(ab_warp_device(ctx, eff[c], rows, cols, ar.transform, rows, cols, img[c]));
                off[0] = ar.transform[5], off[1] = ar.transform[2];
            }
        }
    }
    ab_image_stats full[3];
    for (int c = 0; c < 3; ++c) {  // :243-249
        memset(&full[c], 0, sizeof full[c]);
        if (n > 0) AB_TRY(ab_stats_device(ctx, img[c], n, 0, 0.0, 0.0, &full[c]));
        info->chan_stats[c][0] = full[c].min, info->chan_stats[c][1] = full[c].max;
        info->chan_stats[c][2] = full[c].median, info->chan_stats[c][3] = full[c].mean;
    }
    double wb[3] = {1.0, 1.0, 1.0};  // :251-255
    if (cfg->white_balance == 0)
        ab_select_wb_reference(&full[0], &full[1], &full[2], wb);
    else if (cfg->white_balance == 1)
        memcpy(wb, cfg->wb_manual, sizeof wb);
    bool scaled[3] = {false, false, false};
    for (int c = 0; c < 3; ++c) {  // apply_multiplier_inplace (:127-130)
        const float mult = (float)wb[c];
        if (std::fabs(mult - 1.0f) < 1e-7f || n == 0) continue;
        hipLaunchKernelGGL(scale_inplace_kernel, dim3(stream_grid(ctx, n)), dim3(kBlock), 0, ctx->stream, img[c], n, mult);
        AB_HIP(ctx, hipGetLastError());
        scaled[c] = true;
    }
    ab_image_stats wbst[3];
    const ab_auto_stf_config stf_cfg = {0.25, -2.8};  // AutoStfConfig::default() (types/image.rs:52-65)
    auto stats_of = [&](int c) -> int {  // an unscaled channel's statistics are the ones already taken
        if (!scaled[c]) {
            wbst[c] = full[c];
            return AB_OK;
        }
        memset(&wbst[c], 0, sizeof wbst[c]);
        return ab_stats_device(ctx, img[c], n, 0, 0.0, 0.0, &wbst[c]);
    };
    if (cfg->auto_stretch && cfg->linked_stf) {  // :265-273
        float *comb = nullptr;
        AB_TRY(sc.alloc(&comb, n));
        ab_image_stats st;
        memset(&st, 0, sizeof st);
        if (n > 0) {
            hipLaunchKernelGGL(merge3_kernel, dim3(stream_grid(ctx, n)), dim3(kBlock), 0, ctx->stream, img[0], img[1], img[2], n, comb);
            AB_HIP(ctx, hipGetLastError());
            AB_TRY(ab_stats_device(ctx, comb, n, 0, 0.0, 0.0, &st));
        }
        ab_stf_params p;
        ab_auto_stf(&st, &stf_cfg, &p);
        for (int c = 0; c < 3; ++c) {
            info->stf[c] = p;
            AB_TRY(stats_of(c));
        }
    } else if (cfg->auto_stretch) {  // :274-282
        for (int c = 0; c < 3; ++c) {
            AB_TRY(stats_of(c));
            ab_auto_stf(&wbst[c], &stf_cfg, &info->stf[c]);
        }
    } else {  // :283-294
        for (int c = 0; c < 3; ++c) {
            AB_TRY(stats_of(c));
            if (cfg->has_stf[c])
                info->stf[c] = cfg->stf[c];
            else
                info->stf[c] = ab_stf_params{0.0, 0.5, 1.0};
        }
    }
    for (int c = 0; c < 3; ++c) {
        info->stats_wb[c] = wbst[c];
        if (pres[c]) {  // pre_stretch_* (:296-298)
            AB_TRY(ab_stage_out_begin(ctx, pres[c], &sp[c]));
            sc.outs.push_back(&sp[c]);
            AB_TRY(copy_plane(ctx, sp[c].dptr, img[c], n));
        }
        if (n > 0) {  // apply_stf_inplace (:191-207)
            ComposeStf t;
            t.dmin = wbst[c].min;
            t.inv_range = 1.0 / std::fmax(wbst[c].max - wbst[c].min, 1e-30);
            t.shadow = info->stf[c].shadow;
            t.clip_range = std::fmax(info->stf[c].highlight - info->stf[c].shadow, 1e-15);
            t.m = info->stf[c].midtone;
            hipLaunchKernelGGL(compose_stf_kernel, dim3(stream_grid(ctx, n)), dim3(kBlock), 0, ctx->stream, img[c], n, t);
            AB_HIP(ctx, hipGetLastError());
        }
    }
    if (cfg->has_scnr) {  // :306-312 (the three planes always share dims here)
        ab_plane_mut pm[3];
        for (int c = 0; c < 3; ++c) pm[c] = ab_plane_mut{img[c], rows, cols, 1};
        AB_TRY(ab_apply_scnr_inplace(ctx, &pm[0], &pm[1], &pm[2], &cfg->scnr));
        info->scnr_applied = 1;
    }
    for (int c = 0; c < 3; ++c) {
        AB_TRY(ab_stage_out_finish(ctx, &so[c]));
        if (pres[c]) AB_TRY(ab_stage_out_finish(ctx, &sp[c]));
    }
    return AB_OK;
} AB_CATCH(ctx)

}  // extern "C"
