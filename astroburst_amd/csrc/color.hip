// Elementwise colour / tone / calibration maps on gfx950.
//
// Replaces core/imaging/scnr.rs:18-53 (apply_scnr_inplace), core/compose/channel_blend.rs:13-70
// (blend_channels), core/imaging/curves.rs (SplineLut::from_points :69-95, apply_curve :186-197,
// apply_levels :31-52), core/imaging/stretch.rs:10-45 (arcsinh_stretch_with_stats),
// core/imaging/masked_stretch.rs:143-154 (luminance), cmd/compose/color.rs:28-40 (white-balance
// scale) and core/stacking/calibration.rs:47-82 (calibrate_image).
//
// All of these are HBM-bound streaming maps: every plane is read once and written once with
// float4 accesses from a grid-stride loop sized to the chip (256 CUs x 8 workgroups).  The blend
// "matrix" is 3 x K with K <= 7 in the product, i.e. 3K multiply-adds per 4(K+3) bytes -- far
// below the HBM ridge, so it is a VALU kernel, not an MFMA one; products and sums are kept as
// separate f32 operations in weight order (no FMA contraction) so the result is bit-identical to
// the reference's `rv += v * rw`.  The curve LUT (4096 f32 = 16 KiB) is staged in LDS.
#include "ab_common.hpp"

#include <algorithm>
#include <cmath>

namespace {

constexpr int kBlock = 256;
constexpr int kMaxBlendChannels = 16;
constexpr int kMaxBlendWeights = 32;

int stream_grid(ab_ctx *ctx, int64_t n4) {
    const int64_t want = (n4 + kBlock - 1) / kBlock;
    const int64_t cap = (int64_t)(ctx->cu_count > 0 ? ctx->cu_count : 256) * 8;
    return (int)std::max<int64_t>(1, std::min(want, cap));
}

// ---- SCNR (scnr.rs:36-52) ---------------------------------------------------------------------
__device__ __forceinline__ void scnr_px(float &rv, float &gv, float &bv, int method, float amount, int preserve) {
    const float LUM_R = 0.2126f, LUM_G = 0.7152f, LUM_B = 0.0722f;
    const float INV_RB_WEIGHT = 1.0f / (LUM_R + LUM_B);
    const float limit = method == 0 ? (rv + bv) * 0.5f : fmaxf(rv, bv);
    const float g_corrected = fminf(gv, limit);
    const float g_new = gv + amount * (g_corrected - gv);
    const float delta_g = gv - g_new;
    if (preserve && delta_g > 1e-10f && rv <= 1.0f && bv <= 1.0f) {
        const float lum_lost = LUM_G * delta_g;
        const float boost = lum_lost * INV_RB_WEIGHT;
        const float rn = rv + boost, bn = bv + boost;
        rv = rn > 1.0f ? 1.0f : rn;
        bv = bn > 1.0f ? 1.0f : bn;
    }
    gv = g_new;
}

__global__ __launch_bounds__(kBlock) void scnr_kernel(float *r, float *g, float *b, int64_t n, int method, float amount,
                                                      int preserve, int vec) {
    const int64_t n4 = vec ? (n >> 2) : 0, stride = (int64_t)gridDim.x * kBlock;
    float4 *r4 = (float4 *)r, *g4 = (float4 *)g, *b4 = (float4 *)b;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n4; i += stride) {
        float4 rv = r4[i], gv = g4[i], bv = b4[i];
        scnr_px(rv.x, gv.x, bv.x, method, amount, preserve);
        scnr_px(rv.y, gv.y, bv.y, method, amount, preserve);
        scnr_px(rv.z, gv.z, bv.z, method, amount, preserve);
        scnr_px(rv.w, gv.w, bv.w, method, amount, preserve);
        r4[i] = rv;
        g4[i] = gv;
        b4[i] = bv;
    }
    // scalar remainder (everything when the planes are not 16-byte aligned)
    for (int64_t i = (n4 << 2) + (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) {
        float rv = r[i], gv = g[i], bv = b[i];
        scnr_px(rv, gv, bv, method, amount, preserve);
        r[i] = rv;
        g[i] = gv;
        b[i] = bv;
    }
}

// ---- blend matrix (channel_blend.rs:36-63) -------------------------------------------------------
struct BlendArgs {
    const float *ch[kMaxBlendChannels];
    int idx[kMaxBlendWeights];
    float rw[kMaxBlendWeights], gw[kMaxBlendWeights], bw[kMaxBlendWeights];
    int n_weights;
    int64_t n;
    float *r, *g, *b;
};

__global__ __launch_bounds__(kBlock) void blend_kernel(const BlendArgs a) {
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < a.n; i += stride) {
        float rv = 0.0f, gv = 0.0f, bv = 0.0f;
        for (int w = 0; w < a.n_weights; ++w) {
            const float v = a.ch[a.idx[w]][i];
            rv += v * a.rw[w];
            gv += v * a.gw[w];
            bv += v * a.bw[w];
        }
        a.r[i] = rv;
        a.g[i] = gv;
        a.b[i] = bv;
    }
}

// ---- curve LUT / levels / arcsinh / luminance / scale / calibrate -----------------------------------
__global__ __launch_bounds__(kBlock) void curve_kernel(const float *in, int64_t n, const float *__restrict__ lut, float *out) {
    __shared__ float s_lut[4096];
    for (int i = threadIdx.x; i < 4096; i += kBlock) s_lut[i] = lut[i];
    __syncthreads();
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) {
        const float v = in[i];
        float r = 0.0f;
        if (__builtin_isfinite(v) && !(v < 0.0f)) {              // curves.rs:192
            const float t = fminf(fmaxf(v, 0.0f), 1.0f) * 4095.0f;  // curves.rs:105
            int idx = (int)t;                                     // `as usize` of a value in [0, 4095]
            idx = idx > 4095 ? 4095 : idx;
            r = s_lut[idx];
        }
        out[i] = r;
    }
}

__global__ __launch_bounds__(kBlock) void levels_kernel(const float *in, int64_t n, double black, double inv_range,
                                                        double inv_gamma, float *out) {
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) {
        const float v = in[i];
        float r = 0.0f;
        if (__builtin_isfinite(v) && !(v < 0.0f)) {              // curves.rs:46
            double norm = ((double)v - black) * inv_range;        // curves.rs:27-29
            norm = norm < 0.0 ? 0.0 : (norm > 1.0 ? 1.0 : norm);
            r = (float)pow(norm, inv_gamma);
        }
        out[i] = r;
    }
}

__global__ __launch_bounds__(kBlock) void arcsinh_kernel(const float *in, int64_t n, float dmin, float inv_range, float factor,
                                                         float inv_denom, int apply_gamma, float gamma, float *out) {
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) {
        const float val = in[i];
        float r = 0.0f;
        if (__builtin_isfinite(val)) {                             // stretch.rs:36-43
            float norm = (val - dmin) * inv_range;
            norm = norm < 0.0f ? 0.0f : (norm > 1.0f ? 1.0f : norm);
            const float stretched = asinhf(norm * factor) * inv_denom;
            r = apply_gamma ? powf(stretched, gamma) : stretched;
        }
        out[i] = r;
    }
}

__global__ __launch_bounds__(kBlock) void luminance_kernel(const float *r, const float *g, const float *b, int64_t n,
                                                           float *out) {
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) {
        const float rv = r[i], gv = g[i], bv = b[i];
        const float rn = __builtin_isfinite(rv) ? rv : 0.0f, gn = __builtin_isfinite(gv) ? gv : 0.0f,
                    bn = __builtin_isfinite(bv) ? bv : 0.0f;
        out[i] = 0.2126f * rn + 0.7152f * gn + 0.0722f * bn;    // masked_stretch.rs:152
    }
}

__global__ __launch_bounds__(kBlock) void scale_kernel(const float *in, int64_t n, float factor, float *out, int vec) {
    const int64_t n4 = vec ? (n >> 2) : 0, stride = (int64_t)gridDim.x * kBlock;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n4; i += stride) {
        float4 v = ((const float4 *)in)[i];
        v.x *= factor;
        v.y *= factor;
        v.z *= factor;
        v.w *= factor;
        ((float4 *)out)[i] = v;
    }
    for (int64_t i = (n4 << 2) + (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) out[i] = in[i] * factor;
}

__global__ __launch_bounds__(kBlock) void calibrate_kernel(const float *raw, const float *bias, const float *dark,
                                                           const float *flat, float dark_ratio, int64_t n, float *out) {
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) {
        float v = raw[i];                                          // calibration.rs:60-79
        if (bias) v -= bias[i];
        if (dark) v -= dark[i] * dark_ratio;
        if (flat) {
            const float fv = flat[i];
            if (__builtin_isfinite(fv) && fabsf(fv) > 1e-4f) v /= fv;
        }
        out[i] = v < 0.0f ? 0.0f : v;
    }
}

// ---- host: SplineLut::from_points (curves.rs:69-184), scalar f64 ----------------------------------
inline double clampd(double x, double lo, double hi) { return x < lo ? lo : (x > hi ? hi : x); }
inline double signum(double x) { return std::isnan(x) ? x : (std::signbit(x) ? -1.0 : 1.0); }

void fritsch_carlson(const std::vector<double> &px, const std::vector<double> &py, std::vector<double> &m) {
    const size_t n = px.size();
    m.assign(n, 0.0);
    if (n < 2) return;
    if (n == 2) {
        const double slope = (py[1] - py[0]) / std::fmax(px[1] - px[0], 1e-15);
        m[0] = m[1] = slope;
        return;
    }
    std::vector<double> slopes(n - 1);
    for (size_t i = 0; i + 1 < n; ++i) slopes[i] = (py[i + 1] - py[i]) / std::fmax(px[i + 1] - px[i], 1e-15);
    m[0] = slopes[0];
    m[n - 1] = slopes[n - 2];
    for (size_t i = 1; i + 1 < n; ++i)
        m[i] = (signum(slopes[i - 1]) != signum(slopes[i])) ? 0.0 : (slopes[i - 1] + slopes[i]) * 0.5;
    for (size_t i = 0; i + 1 < n; ++i) {
        if (std::fabs(slopes[i]) < 1e-15) {
            m[i] = 0.0;
            m[i + 1] = 0.0;
            continue;
        }
        const double alpha = m[i] / slopes[i], beta = m[i + 1] / slopes[i];
        const double tau = alpha * alpha + beta * beta;
        if (tau > 9.0) {
            const double s = 3.0 / std::sqrt(tau);
            m[i] = s * alpha * slopes[i];
            m[i + 1] = s * beta * slopes[i];
        }
    }
}

double hermite_eval(const std::vector<double> &px, const std::vector<double> &py, const std::vector<double> &tan, double x) {
    const size_t n = px.size();
    if (x <= px[0]) return py[0];
    if (x >= px[n - 1]) return py[n - 1];
    size_t seg = 0;
    for (size_t i = 1; i < n; ++i)
        if (x < px[i]) {
            seg = i - 1;
            break;
        }
    const double dx = std::fmax(px[seg + 1] - px[seg], 1e-15);
    const double t = (x - px[seg]) / dx;
    const double t2 = t * t, t3 = t2 * t;
    const double h00 = 2.0 * t3 - 3.0 * t2 + 1.0, h10 = t3 - 2.0 * t2 + t, h01 = -2.0 * t3 + 3.0 * t2, h11 = t3 - t2;
    return h00 * py[seg] + h10 * dx * tan[seg] + h01 * py[seg + 1] + h11 * dx * tan[seg + 1];
}

// plane plumbing shared by the unary maps: stage in, run, stage out
template <typename F>
int unary_map(ab_ctx *ctx, const ab_plane *img, ab_plane_mut *out, F launch) {
    AB_CHECK(ctx, img && out, "null plane");
    AB_CHECK(ctx, img->rows == out->rows && img->cols == out->cols, "output dims must equal input dims");
    AB_HIP(ctx, hipSetDevice(ctx->device));
    StagedPlane in;
    AB_TRY(ab_stage_in(ctx, img, &in));
    StagedOut so;
    int rc = ab_stage_out_begin(ctx, out, &so);
    if (rc == AB_OK) {
        launch(in.dptr, in.rows * in.cols, so.dptr);
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) rc = ab_set_error(ctx, AB_ERR_HIP, "kernel launch: %s", hipGetErrorString(e));
        if (rc == AB_OK)
            rc = ab_stage_out_finish(ctx, &so);
        else
            ab_stage_out_abort(ctx, &so);
    }
    ab_stage_release(ctx, &in);
    return rc;
}

}  // namespace

extern "C" {

int ab_apply_scnr_inplace(ab_ctx *ctx, ab_plane_mut *r, ab_plane_mut *g, ab_plane_mut *b, const ab_scnr_config *cfg) try {
    if (!ctx) return AB_ERR_INVALID;
    AB_CHECK(ctx, r && g && b && cfg, "null argument");
    if (r->rows != g->rows || r->cols != g->cols || g->rows != b->rows || g->cols != b->cols) return AB_OK;  // scnr.rs:24-26
    float amount = cfg->amount < 0.0f ? 0.0f : (cfg->amount > 1.0f ? 1.0f : cfg->amount);                     // :28
    if (amount < 1e-7f) return AB_OK;                                                                           // :29-31
    AB_CHECK(ctx, r->on_device == g->on_device && g->on_device == b->on_device, "SCNR planes must live on the same side");
    AB_HIP(ctx, hipSetDevice(ctx->device));
    const int64_t n = r->rows * r->cols;
    float *dr = r->data, *dg = g->data, *db = b->data;
    void *tmp = nullptr;
    if (!r->on_device) {
        const int64_t np4 = (n + 3) & ~int64_t(3);  // keep the three staged planes 16-byte aligned
        AB_HIP(ctx, hipMalloc(&tmp, 3 * np4 * sizeof(float)));
        dr = (float *)tmp;
        dg = dr + np4;
        db = dg + np4;
        hipError_t e = hipMemcpyAsync(dr, r->data, n * 4, hipMemcpyHostToDevice, ctx->stream);
        if (e == hipSuccess) e = hipMemcpyAsync(dg, g->data, n * 4, hipMemcpyHostToDevice, ctx->stream);
        if (e == hipSuccess) e = hipMemcpyAsync(db, b->data, n * 4, hipMemcpyHostToDevice, ctx->stream);
        if (e != hipSuccess) {
            (void)hipFree(tmp);
            return ab_set_error(ctx, AB_ERR_HIP, "H2D: %s", hipGetErrorString(e));
        }
    }
    const int vec = (((uintptr_t)dr | (uintptr_t)dg | (uintptr_t)db) & 15) == 0;
    hipLaunchKernelGGL(scnr_kernel, dim3(stream_grid(ctx, vec ? (n >> 2) : n)), dim3(kBlock), 0, ctx->stream, dr, dg, db, n,
                       (int)cfg->method, amount, (int)cfg->preserve_luminance, vec);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess && tmp) {
        e = hipMemcpyAsync(r->data, dr, n * 4, hipMemcpyDeviceToHost, ctx->stream);
        if (e == hipSuccess) e = hipMemcpyAsync(g->data, dg, n * 4, hipMemcpyDeviceToHost, ctx->stream);
        if (e == hipSuccess) e = hipMemcpyAsync(b->data, db, n * 4, hipMemcpyDeviceToHost, ctx->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    }
    if (tmp) (void)hipFree(tmp);
    if (e != hipSuccess) return ab_set_error(ctx, AB_ERR_HIP, "SCNR: %s", hipGetErrorString(e));
    return AB_OK;
} AB_CATCH(ctx)

int ab_blend_channels(ab_ctx *ctx, const ab_plane *channels, size_t n_channels, const ab_blend_weight *weights,
                      size_t n_weights, ab_plane_mut *r, ab_plane_mut *g, ab_plane_mut *b) try {
    if (!ctx) return AB_ERR_INVALID;
    AB_CHECK(ctx, channels && weights && r && g && b, "null argument");
    AB_CHECK(ctx, n_channels >= 1 && n_channels <= (size_t)kMaxBlendChannels, "1..%d channels", kMaxBlendChannels);
    const int64_t rows = r->rows, cols = r->cols, n = rows * cols;
    AB_CHECK(ctx, g->rows == rows && g->cols == cols && b->rows == rows && b->cols == cols, "R/G/B outputs must share dims");
    for (size_t i = 0; i < n_channels; ++i)
        AB_CHECK(ctx, channels[i].rows * channels[i].cols >= n, "channel %zu is smaller than the output", i);
    AB_HIP(ctx, hipSetDevice(ctx->device));
    BlendArgs a;
    memset(&a, 0, sizeof a);
    int nw = 0;
    for (size_t w = 0; w < n_weights; ++w) {                       // channel_blend.rs:20-23
        if (weights[w].channel_idx >= n_channels) continue;
        AB_CHECK(ctx, nw < kMaxBlendWeights, "at most %d blend weights", kMaxBlendWeights);
        a.idx[nw] = (int)weights[w].channel_idx;
        a.rw[nw] = (float)weights[w].r_weight;
        a.gw[nw] = (float)weights[w].g_weight;
        a.bw[nw] = (float)weights[w].b_weight;
        ++nw;
    }
    a.n_weights = nw;
    a.n = n;
    std::vector<StagedPlane> st(n_channels);
    int rc = AB_OK;
    size_t staged = 0;
    for (; staged < n_channels; ++staged) {
        rc = ab_stage_in(ctx, &channels[staged], &st[staged]);
        if (rc != AB_OK) break;
        a.ch[staged] = st[staged].dptr;
    }
    StagedOut so[3];
    ab_plane_mut *outs[3] = {r, g, b};
    int opened = 0;
    for (; rc == AB_OK && opened < 3; ++opened) rc = ab_stage_out_begin(ctx, outs[opened], &so[opened]);
    if (rc != AB_OK && opened > 0) --opened;
    if (rc == AB_OK) {
        a.r = so[0].dptr;
        a.g = so[1].dptr;
        a.b = so[2].dptr;
        hipLaunchKernelGGL(blend_kernel, dim3(stream_grid(ctx, n)), dim3(kBlock), 0, ctx->stream, a);
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) rc = ab_set_error(ctx, AB_ERR_HIP, "blend: %s", hipGetErrorString(e));
    }
    for (int i = 0; i < opened; ++i) {
        if (rc == AB_OK)
            rc = ab_stage_out_finish(ctx, &so[i]);
        else
            ab_stage_out_abort(ctx, &so[i]);
    }
    for (size_t i = 0; i < staged; ++i) ab_stage_release(ctx, &st[i]);
    return rc;
} AB_CATCH(ctx)

int ab_spline_lut_from_points(const double *points_xy, size_t n_in, float *lut4096) try {
    if ((!points_xy && n_in) || !lut4096) return AB_ERR_INVALID;
    std::vector<std::pair<double, double>> pts(n_in);
    for (size_t i = 0; i < n_in; ++i) pts[i] = {points_xy[2 * i], points_xy[2 * i + 1]};
    std::stable_sort(pts.begin(), pts.end(), [](const auto &a, const auto &b) { return a.first < b.first; });  // :71
    std::vector<std::pair<double, double>> s;
    for (const auto &p : pts)                                                                                   // :72
        if (s.empty() || !(std::fabs(p.first - s.back().first) < 1e-9)) s.push_back(p);
    if (s.empty() || s[0].first > 1e-6) s.insert(s.begin(), {0.0, 0.0});                                        // :74-76
    if (s.empty() || s.back().first < 1.0 - 1e-6) s.push_back({1.0, 1.0});                                      // :77-79
    std::vector<double> px(s.size()), py(s.size()), tan;
    for (size_t i = 0; i < s.size(); ++i) {
        px[i] = s[i].first;
        py[i] = s[i].second;
    }
    fritsch_carlson(px, py, tan);
    for (int i = 0; i < 4096; ++i) lut4096[i] = (float)clampd(hermite_eval(px, py, tan, (double)i / 4095.0), 0.0, 1.0);
    return AB_OK;
} AB_CATCH_NOCTX

int ab_apply_curve(ab_ctx *ctx, const ab_plane *img, const float *lut4096_host, ab_plane_mut *out) try {
    if (!ctx) return AB_ERR_INVALID;
    AB_CHECK(ctx, lut4096_host, "null LUT");
    void *dlut = nullptr;
    AB_HIP(ctx, hipSetDevice(ctx->device));
    AB_TRY(ab_scratch(ctx, 4096 * sizeof(float), &dlut));
    AB_HIP(ctx, hipMemcpyAsync(dlut, lut4096_host, 4096 * sizeof(float), hipMemcpyHostToDevice, ctx->stream));
    AB_HIP(ctx, hipStreamSynchronize(ctx->stream));  // the host LUT may be a temporary
    return unary_map(ctx, img, out, [&](const float *in, int64_t n, float *o) {
        hipLaunchKernelGGL(curve_kernel, dim3(stream_grid(ctx, n)), dim3(kBlock), 0, ctx->stream, in, n, (const float *)dlut, o);
    });
} AB_CATCH(ctx)

int ab_apply_levels(ab_ctx *ctx, const ab_plane *img, const ab_levels_params *p, ab_plane_mut *out) try {
    if (!ctx) return AB_ERR_INVALID;
    AB_CHECK(ctx, p, "null params");
    const bool identity = std::fabs(p->black) < 1e-7 && std::fabs(p->gamma - 1.0) < 1e-7 && std::fabs(p->white - 1.0) < 1e-7;
    if (identity)  // curves.rs:32-34: data.clone()
        return unary_map(ctx, img, out, [&](const float *in, int64_t n, float *o) {
            if (in != o) (void)hipMemcpyAsync(o, in, n * sizeof(float), hipMemcpyDeviceToDevice, ctx->stream);
        });
    const double range = std::fmax(p->white - p->black, 1e-15);
    const double inv_range = 1.0 / range, inv_gamma = 1.0 / clampd(p->gamma, 0.01, 10.0);
    return unary_map(ctx, img, out, [&](const float *in, int64_t n, float *o) {
        hipLaunchKernelGGL(levels_kernel, dim3(stream_grid(ctx, n)), dim3(kBlock), 0, ctx->stream, in, n, p->black, inv_range,
                           inv_gamma, o);
    });
} AB_CATCH(ctx)

int ab_arcsinh_stretch_with_stats(ab_ctx *ctx, const ab_plane *img, float dmin, float dmax, float factor, float gamma,
                                  ab_plane_mut *out) try {
    if (!ctx) return AB_ERR_INVALID;
    if (std::fabs(factor) < 1e-10f)  // stretch.rs:17-19
        return unary_map(ctx, img, out, [&](const float *in, int64_t n, float *o) {
            if (in != o) (void)hipMemcpyAsync(o, in, n * sizeof(float), hipMemcpyDeviceToDevice, ctx->stream);
        });
    const float range = dmax - dmin;
    if (range < 1e-10f)              // stretch.rs:22-24
        return unary_map(ctx, img, out, [&](const float *, int64_t n, float *o) {
            (void)hipMemsetAsync(o, 0, n * sizeof(float), ctx->stream);
        });
    const float inv_range = 1.0f / range, inv_denom = 1.0f / asinhf(factor);
    const int apply_gamma = std::fabs(gamma - 1.0f) > 1e-6f;
    return unary_map(ctx, img, out, [&](const float *in, int64_t n, float *o) {
        hipLaunchKernelGGL(arcsinh_kernel, dim3(stream_grid(ctx, n)), dim3(kBlock), 0, ctx->stream, in, n, dmin, inv_range, factor,
                           inv_denom, apply_gamma, gamma, o);
    });
} AB_CATCH(ctx)

int ab_scale(ab_ctx *ctx, const ab_plane *img, float factor, ab_plane_mut *out) try {
    if (!ctx) return AB_ERR_INVALID;
    return unary_map(ctx, img, out, [&](const float *in, int64_t n, float *o) {
        const int vec = (((uintptr_t)in | (uintptr_t)o) & 15) == 0;
        hipLaunchKernelGGL(scale_kernel, dim3(stream_grid(ctx, vec ? (n >> 2) : n)), dim3(kBlock), 0, ctx->stream, in, n, factor, o,
                           vec);
    });
} AB_CATCH(ctx)

int ab_luminance(ab_ctx *ctx, const ab_plane *r, const ab_plane *g, const ab_plane *b, ab_plane_mut *out) try {
    if (!ctx) return AB_ERR_INVALID;
    AB_CHECK(ctx, r && g && b && out, "null plane");
    AB_CHECK(ctx, r->rows == g->rows && r->cols == g->cols && g->rows == b->rows && g->cols == b->cols &&
                      out->rows == r->rows && out->cols == r->cols,
             "luminance: R, G, B and the output must share dims");
    AB_HIP(ctx, hipSetDevice(ctx->device));
    StagedPlane sr, sg, sb;
    AB_TRY(ab_stage_in(ctx, r, &sr));
    int rc = ab_stage_in(ctx, g, &sg);
    if (rc == AB_OK) {
        rc = ab_stage_in(ctx, b, &sb);
        if (rc == AB_OK) {
            StagedOut so;
            rc = ab_stage_out_begin(ctx, out, &so);
            if (rc == AB_OK) {
                const int64_t n = r->rows * r->cols;
                hipLaunchKernelGGL(luminance_kernel, dim3(stream_grid(ctx, n)), dim3(kBlock), 0, ctx->stream, sr.dptr, sg.dptr,
                                   sb.dptr, n, so.dptr);
                hipError_t e = hipGetLastError();
                if (e != hipSuccess) rc = ab_set_error(ctx, AB_ERR_HIP, "luminance: %s", hipGetErrorString(e));
                if (rc == AB_OK)
                    rc = ab_stage_out_finish(ctx, &so);
                else
                    ab_stage_out_abort(ctx, &so);
            }
            ab_stage_release(ctx, &sb);
        }
        ab_stage_release(ctx, &sg);
    }
    ab_stage_release(ctx, &sr);
    return rc;
} AB_CATCH(ctx)

int ab_calibrate_image(ab_ctx *ctx, const ab_plane *raw, const ab_plane *bias, const ab_plane *dark, const ab_plane *flat,
                       float dark_exposure_ratio, ab_plane_mut *out) try {
    if (!ctx) return AB_ERR_INVALID;
    AB_CHECK(ctx, raw && out, "null plane");
    const ab_plane *opt[3] = {bias, dark, flat};
    for (int i = 0; i < 3; ++i)
        if (opt[i]) AB_CHECK(ctx, opt[i]->rows == raw->rows && opt[i]->cols == raw->cols, "master frame dims differ from the raw frame");
    AB_HIP(ctx, hipSetDevice(ctx->device));
    StagedPlane so_[3];
    const float *d[3] = {nullptr, nullptr, nullptr};
    int staged = 0, rc = AB_OK;
    for (; staged < 3; ++staged) {
        if (!opt[staged]) continue;
        rc = ab_stage_in(ctx, opt[staged], &so_[staged]);
        if (rc != AB_OK) break;
        d[staged] = so_[staged].dptr;
    }
    if (rc == AB_OK)
        rc = unary_map(ctx, raw, out, [&](const float *in, int64_t n, float *o) {
            hipLaunchKernelGGL(calibrate_kernel, dim3(stream_grid(ctx, n)), dim3(kBlock), 0, ctx->stream, in, d[0], d[1], d[2],
                               dark_exposure_ratio, n, o);
        });
    for (int i = 0; i < staged; ++i)
        if (opt[i]) ab_stage_release(ctx, &so_[i]);
    return rc;
} AB_CATCH(ctx)

}  // extern "C"
