// Small plane-level helpers around the compose / calibration callers (gfx950).
//
// Replaces core/compose/lrgb.rs (apply_lrgb :4-45, synthesize_luminance :47-64), cmd/helpers.rs:175-202
// (compute_linked_stf_with_stats), cmd/compose/color.rs:21-49 (calibrate_channel) and the plane-level
// bodies of core/stacking/calibration.rs create_master_bias / _dark / _flat (:127-255).
// All maps are streaming f32 kernels in the reference's operation order (bit-exact); the one reduction
// (the master flat's mean, calibration.rs:228-236) is a two-level f64 sum, so the flat's f32 scale factor
// can differ from the reference's sequential sum in the last ulp on rare inputs (tests bound it at 1 ulp).
#include "ab_common.hpp"

#include <algorithm>
#include <cmath>

namespace {

constexpr int kBlock = 256;

int stream_grid(ab_ctx *ctx, int64_t n) {
    return (int)std::max<int64_t>(1, std::min<int64_t>((n + kBlock - 1) / kBlock, (int64_t)(ctx->cu_count > 0 ? ctx->cu_count : 256) * 8));
}

#define AB_GRID_LOOP(i, n) \
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x, stride_ = (int64_t)gridDim.x * kBlock; i < (n); i += stride_)

__device__ __forceinline__ float clamp01(float v) { return v < 0.0f ? 0.0f : (v > 1.0f ? 1.0f : v); }

__global__ __launch_bounds__(kBlock) void lrgb_kernel(const float *__restrict__ l, float *__restrict__ r, float *__restrict__ g,
                                                      float *__restrict__ b, int64_t n, float lw, float cw) {
    AB_GRID_LOOP(i, n) {  // lrgb.rs:25-42
        const float rv = r[i], gv = g[i], bv = b[i], lum_new = l[i];
        const float lum_old = rv * 0.2126f + gv * 0.7152f + bv * 0.0722f;
        if (lum_old < 1e-10f) {
            const float blended = lum_new * lw;
            r[i] = blended, g[i] = blended, b[i] = blended;
            continue;
        }
        const float ratio = (lum_new * lw + lum_old * (1.0f - lw)) / lum_old;
        r[i] = clamp01(rv * ratio * cw + lum_new * (1.0f - cw));
        g[i] = clamp01(gv * ratio * cw + lum_new * (1.0f - cw));
        b[i] = clamp01(bv * ratio * cw + lum_new * (1.0f - cw));
    }
}

__global__ __launch_bounds__(kBlock) void lum_plain_kernel(const float *__restrict__ r, const float *__restrict__ g, const float *__restrict__ b,
                                                           int64_t n, float *__restrict__ out) {
    AB_GRID_LOOP(i, n) out[i] = r[i] * 0.2126f + g[i] * 0.7152f + b[i] * 0.0722f;  // lrgb.rs:59-61
}

__global__ __launch_bounds__(kBlock) void scale_kernel(const float *__restrict__ in, int64_t n, float factor, float *__restrict__ out) {
    AB_GRID_LOOP(i, n) out[i] = in[i] * factor;  // color.rs:29,36-39
}

__global__ __launch_bounds__(kBlock) void preprocess_kernel(const float *__restrict__ frame, const float *__restrict__ bias,
                                                            const float *__restrict__ dark, int64_t n, float *__restrict__ out) {
    AB_GRID_LOOP(i, n) {  // calibration.rs:15-25
        float v = frame[i];
        if (bias) v = v - bias[i];
        if (dark) v = v - dark[i] * 1.0f;
        out[i] = v;
    }
}

__global__ __launch_bounds__(kBlock) void flat_sum_kernel(const float *__restrict__ data, int64_t n, double *__restrict__ part_sum,
                                                          unsigned long long *__restrict__ part_cnt) {
    __shared__ double s_sum[kBlock];
    __shared__ unsigned long long s_cnt[kBlock];
    double sum = 0.0;
    unsigned long long cnt = 0;
    AB_GRID_LOOP(i, n) {
        const float v = data[i];
        if (__builtin_isfinite(v) && v > 0.0f) {
            sum += (double)v;
            ++cnt;
        }
    }
    s_sum[threadIdx.x] = sum;
    s_cnt[threadIdx.x] = cnt;
    __syncthreads();
    for (int s = kBlock / 2; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) {
            s_sum[threadIdx.x] += s_sum[threadIdx.x + s];
            s_cnt[threadIdx.x] += s_cnt[threadIdx.x + s];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        part_sum[blockIdx.x] = s_sum[0];
        part_cnt[blockIdx.x] = s_cnt[0];
    }
}

__global__ __launch_bounds__(kBlock) void flat_normalise_kernel(float *__restrict__ data, int64_t n, float inv_mean) {
    AB_GRID_LOOP(i, n) {  // calibration.rs:241-247
        const float v = data[i];
        data[i] = (__builtin_isfinite(v) && v > 0.0f) ? v * inv_mean : 1.0f;
    }
}

}  // namespace

extern "C" {

int ab_apply_lrgb(ab_ctx *ctx, const ab_plane *l, ab_plane_mut *r, ab_plane_mut *g, ab_plane_mut *b, float lightness_weight,
                  float chrominance_weight) try {
    if (!ctx) return AB_ERR_INVALID;
    AB_CHECK(ctx, l && r && g && b, "null plane");
    const ab_plane_mut *ch[3] = {r, g, b};
    for (const ab_plane_mut *p : ch)
        if (p->rows != l->rows || p->cols != l->cols)  // lrgb.rs:14-19
            return ab_set_error(ctx, AB_ERR_INVALID, "L dimensions (%lld, %lld) do not match RGB (R: (%lld, %lld), G: (%lld, %lld), B: (%lld, %lld))",
                                (long long)l->rows, (long long)l->cols, (long long)r->rows, (long long)r->cols, (long long)g->rows,
                                (long long)g->cols, (long long)b->rows, (long long)b->cols);
    AB_CHECK(ctx, r->on_device == g->on_device && g->on_device == b->on_device, "LRGB planes must live on the same side");
    AB_HIP(ctx, hipSetDevice(ctx->device));
    const int64_t n = l->rows * l->cols;
    if (n == 0) return AB_OK;
    StagedPlane sl;
    AB_TRY(ab_stage_in(ctx, l, &sl));
    float *d[3] = {r->data, g->data, b->data};
    void *tmp = nullptr;
    hipError_t e = hipSuccess;
    if (!r->on_device) {
        e = hipMalloc(&tmp, 3 * (size_t)n * sizeof(float));
        for (int c = 0; c < 3 && e == hipSuccess; ++c) {
            d[c] = (float *)tmp + (size_t)c * n;
            e = hipMemcpyAsync(d[c], ch[c]->data, (size_t)n * sizeof(float), hipMemcpyHostToDevice, ctx->stream);
        }
    }
    if (e == hipSuccess) {
        hipLaunchKernelGGL(lrgb_kernel, dim3(stream_grid(ctx, n)), dim3(kBlock), 0, ctx->stream, sl.dptr, d[0], d[1], d[2], n, lightness_weight,
                           chrominance_weight);
        e = hipGetLastError();
    }
    if (tmp) {
        for (int c = 0; c < 3 && e == hipSuccess; ++c)
            e = hipMemcpyAsync(ch[c]->data, d[c], (size_t)n * sizeof(float), hipMemcpyDeviceToHost, ctx->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
        (void)hipFree(tmp);
    }
    ab_stage_release(ctx, &sl);
    if (e != hipSuccess) return ab_set_error(ctx, AB_ERR_HIP, "apply_lrgb: %s", hipGetErrorString(e));
    return AB_OK;
} AB_CATCH(ctx)

int ab_synthesize_luminance(ab_ctx *ctx, const ab_plane *r, const ab_plane *g, const ab_plane *b, ab_plane_mut *out) try {
    if (!ctx) return AB_ERR_INVALID;
    AB_CHECK(ctx, r && g && b && out, "null plane");
    AB_CHECK(ctx, g->rows == r->rows && g->cols == r->cols && b->rows == r->rows && b->cols == r->cols && out->rows == r->rows &&
                      out->cols == r->cols,
             "luminance planes must share dims");
    AB_HIP(ctx, hipSetDevice(ctx->device));
    StagedPlane in[3];
    const ab_plane *src[3] = {r, g, b};
    int staged = 0, rc = AB_OK;
    for (; staged < 3 && rc == AB_OK; ++staged) rc = ab_stage_in(ctx, src[staged], &in[staged]);
    if (rc != AB_OK) --staged;
    StagedOut so;
    if (rc == AB_OK) rc = ab_stage_out_begin(ctx, out, &so);
    if (rc == AB_OK) {
        const int64_t n = r->rows * r->cols;
        if (n > 0) {
            hipLaunchKernelGGL(lum_plain_kernel, dim3(stream_grid(ctx, n)), dim3(kBlock), 0, ctx->stream, in[0].dptr, in[1].dptr, in[2].dptr, n,
                               so.dptr);
            if (hipGetLastError() != hipSuccess) rc = ab_set_error(ctx, AB_ERR_HIP, "luminance launch failed");
        }
        if (rc == AB_OK)
            rc = ab_stage_out_finish(ctx, &so);
        else
            ab_stage_out_abort(ctx, &so);
    }
    for (int i = 0; i < staged; ++i) ab_stage_release(ctx, &in[i]);
    return rc;
} AB_CATCH(ctx)

int ab_compute_linked_stf(const ab_image_stats *sr, const ab_image_stats *sg, const ab_image_stats *sb, const ab_auto_stf_config *cfg,
                          ab_stf_params *out_stf, ab_image_stats *out_combined) try {
    if (!sr || !sg || !sb || !cfg || !out_stf) return AB_ERR_INVALID;
    ab_image_stats c;  // cmd/helpers.rs:191-199
    c.min = std::fmin(std::fmin(sr->min, sg->min), sb->min);
    c.max = std::fmax(std::fmax(sr->max, sg->max), sb->max);
    c.mean = (sr->mean + sg->mean + sb->mean) / 3.0;
    c.median = (sr->median + sg->median + sb->median) / 3.0;
    c.sigma = std::sqrt((sr->sigma * sr->sigma + sg->sigma * sg->sigma + sb->sigma * sb->sigma) / 3.0);
    c.mad = (sr->mad + sg->mad + sb->mad) / 3.0;
    c.valid_count = sr->valid_count;
    if (out_combined) *out_combined = c;
    return ab_auto_stf(&c, cfg, out_stf);
} AB_CATCH_NOCTX

int ab_calibrate_channel(ab_ctx *ctx, const ab_plane *orig, float factor, const ab_image_stats *orig_stats, ab_plane_mut *out,
                         ab_image_stats *out_stats) try {
    if (!ctx) return AB_ERR_INVALID;
    AB_CHECK(ctx, orig && orig_stats && out && out_stats, "null argument");
    AB_CHECK(ctx, out->rows == orig->rows && out->cols == orig->cols, "output must have the channel's dims");
    AB_HIP(ctx, hipSetDevice(ctx->device));
    StagedPlane in;
    AB_TRY(ab_stage_in(ctx, orig, &in));
    StagedOut so;
    int rc = ab_stage_out_begin(ctx, out, &so);
    if (rc == AB_OK) {
        const int64_t n = in.rows * in.cols;
        memset(out_stats, 0, sizeof *out_stats);
        if (n > 0) {
            hipLaunchKernelGGL(scale_kernel, dim3(stream_grid(ctx, n)), dim3(kBlock), 0, ctx->stream, in.dptr, n, factor, so.dptr);
            if (hipGetLastError() != hipSuccess) rc = ab_set_error(ctx, AB_ERR_HIP, "scale launch failed");
            if (rc == AB_OK) {
                if (n <= 4000000) {  // PAR_THRESHOLD, color.rs:19,28-32
                    rc = ab_stats_device(ctx, so.dptr, n, 0, 0.0, 0.0, out_stats);
                } else {  // :41-47
                    const double f = (double)factor;
                    const double kmin = factor >= 0.0f ? orig_stats->min * f : orig_stats->max * f;
                    const double kmax = factor >= 0.0f ? orig_stats->max * f : orig_stats->min * f;
                    rc = ab_stats_device(ctx, so.dptr, n, 1, kmin, kmax, out_stats);
                }
            }
        }
        if (rc == AB_OK)
            rc = ab_stage_out_finish(ctx, &so);
        else
            ab_stage_out_abort(ctx, &so);
    }
    ab_stage_release(ctx, &in);
    return rc;
} AB_CATCH(ctx)

int ab_create_master(ab_ctx *ctx, int32_t kind, const ab_plane *frames, size_t n_frames, const ab_plane *master_bias,
                     const ab_plane *master_dark, ab_plane_mut *out) try {
    if (!ctx) return AB_ERR_INVALID;
    AB_CHECK(ctx, kind >= 0 && kind <= 2 && out, "kind must be 0 (bias), 1 (dark) or 2 (flat)");
    if (!frames || n_frames == 0)  // calibration.rs:128-130,158-160,199-201
        return ab_set_error(ctx, AB_ERR_INVALID, "No %s frames provided", kind == 0 ? "bias" : (kind == 1 ? "dark" : "flat"));
    const int64_t rows = frames[0].rows, cols = frames[0].cols, n = rows * cols;
    for (size_t i = 1; i < n_frames; ++i)
        if (frames[i].rows != rows || frames[i].cols != cols)  // :139-144
            return ab_set_error(ctx, AB_ERR_INVALID, "Dimension mismatch: expected (%lld, %lld), got (%lld, %lld)", (long long)rows,
                                (long long)cols, (long long)frames[i].rows, (long long)frames[i].cols);
    const ab_plane *bias = kind >= 1 ? master_bias : nullptr, *dark = kind == 2 ? master_dark : nullptr;
    AB_CHECK(ctx, (!bias || (bias->rows == rows && bias->cols == cols)) && (!dark || (dark->rows == rows && dark->cols == cols)) &&
                      out->rows == rows && out->cols == cols,
             "master frames and output must have the frames' dims");
    AB_HIP(ctx, hipSetDevice(ctx->device));
    if (!bias && !dark) {
        AB_TRY(ab_median_combine(ctx, frames, n_frames, out));
    } else {
        StagedPlane sb, sd;
        if (bias) AB_TRY(ab_stage_in(ctx, bias, &sb));
        int rc = dark ? ab_stage_in(ctx, dark, &sd) : AB_OK;
        float *pre = nullptr;
        if (rc == AB_OK && hipMalloc((void **)&pre, std::max<size_t>((size_t)n, 1) * n_frames * sizeof(float)) != hipSuccess)
            rc = ab_set_error(ctx, AB_ERR_HIP, "out of device memory for %zu preprocessed frames", n_frames);
        std::vector<ab_plane> planes(n_frames);
        for (size_t f = 0; f < n_frames && rc == AB_OK; ++f) {
            StagedPlane sf;
            rc = ab_stage_in(ctx, &frames[f], &sf);
            if (rc != AB_OK) break;
            float *dst = pre + f * (size_t)n;
            if (n > 0) {
                hipLaunchKernelGGL(preprocess_kernel, dim3(stream_grid(ctx, n)), dim3(kBlock), 0, ctx->stream, sf.dptr, bias ? sb.dptr : nullptr,
                                   dark ? sd.dptr : nullptr, n, dst);
                if (hipGetLastError() != hipSuccess) rc = ab_set_error(ctx, AB_ERR_HIP, "preprocess launch failed");
            }
            if (sf.owned) (void)hipStreamSynchronize(ctx->stream);
            ab_stage_release(ctx, &sf);
            planes[f] = ab_plane{dst, rows, cols, 1};
        }
        if (rc == AB_OK) rc = ab_median_combine(ctx, planes.data(), n_frames, out);
        (void)hipStreamSynchronize(ctx->stream);
        if (pre) (void)hipFree(pre);
        if (bias) ab_stage_release(ctx, &sb);
        if (dark) ab_stage_release(ctx, &sd);
        if (rc != AB_OK) return rc;
    }
    if (kind != 2 || n == 0) return AB_OK;
    // master flat normalisation (:228-247) on the combined plane
    float *d = out->data;
    void *tmp = nullptr;
    if (!out->on_device) {
        AB_HIP(ctx, hipMalloc(&tmp, (size_t)n * sizeof(float)));
        d = (float *)tmp;
    }
    const int grid = stream_grid(ctx, n);
    double *psum = nullptr;
    unsigned long long *pcnt = nullptr;
    hipError_t e = hipMalloc((void **)&psum, grid * sizeof(double));
    if (e == hipSuccess) e = hipMalloc((void **)&pcnt, grid * sizeof(unsigned long long));
    if (e == hipSuccess && tmp) e = hipMemcpyAsync(d, out->data, (size_t)n * sizeof(float), hipMemcpyHostToDevice, ctx->stream);
    std::vector<double> hs(grid);
    std::vector<unsigned long long> hc(grid);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(flat_sum_kernel, dim3(grid), dim3(kBlock), 0, ctx->stream, d, n, psum, pcnt);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpyAsync(hs.data(), psum, grid * sizeof(double), hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(hc.data(), pcnt, grid * sizeof(unsigned long long), hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    if (e == hipSuccess) {
        double sum = 0.0;
        unsigned long long count = 0;
        for (int i = 0; i < grid; ++i) sum += hs[i], count += hc[i];
        if (count > 0) {
            const double mean = sum / (double)count;
            const float inv_mean = std::fabs(mean) > 1e-10 ? 1.0f / (float)mean : 1.0f;
            hipLaunchKernelGGL(flat_normalise_kernel, dim3(grid), dim3(kBlock), 0, ctx->stream, d, n, inv_mean);
            e = hipGetLastError();
        }
    }
    if (e == hipSuccess && tmp) {
        e = hipMemcpyAsync(out->data, d, (size_t)n * sizeof(float), hipMemcpyDeviceToHost, ctx->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    }
    if (psum) (void)hipFree(psum);
    if (pcnt) (void)hipFree(pcnt);
    if (tmp) (void)hipFree(tmp);
    if (e != hipSuccess) return ab_set_error(ctx, AB_ERR_HIP, "master flat normalisation: %s", hipGetErrorString(e));
    return AB_OK;
} AB_CATCH(ctx)

}  // extern "C"
