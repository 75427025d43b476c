// Preview / tile renderers up to (not including) the PNG encoder, on gfx950.
//
// Replaces the pixel loops of cmd/helpers.rs:204-322 (render_rgb_preview, render_rgb_preview_with_stf),
// infra/render/rgb.rs:7-34 (render_rgb), infra/render/tiles.rs (downsample_2x :41-70, render_tile :72-113,
// percentile_bounds :149-178, generate_tile_pyramid :180-255, render_tile_rgb(_stf) :257-341,
// generate_tile_pyramid_rgb_inner :383-481) and infra/ipc.rs:36-148 (raw-f32 IPC buffer + 16-byte header).
//
// All of it is streaming byte work: the planes already sit in HBM after the stack / compose step, so a preview costs
// one strided read of the sources and a write 4x..16x smaller; only the u8 buffers cross PCIe.  Outputs are written
// 4 pixels per lane as whole dwords.  Arithmetic is the reference's (f32 for the linear maps, f64 for the STF, f64 for
// the nearest-neighbour source index), so every byte is bit-exact against the CPU restatement.
#include "ab_common.hpp"
#include "stf_device.hpp"

#include <algorithm>
#include <cfloat>
#include <cmath>

namespace {

constexpr int kBlock = 256;

// `f32 as u8` (saturating, NaN -> 0)
__device__ __forceinline__ unsigned char sat_u8(float v) { return !(v > 0.0f) ? 0 : (v >= 255.0f ? 255 : (unsigned char)v); }
__device__ __forceinline__ float clamp01(float v) { return v < 0.0f ? 0.0f : (v > 1.0f ? 1.0f : v); }  // f32::clamp: NaN stays NaN

struct Rgb {
    const float *r, *g, *b;
};
struct Stf3 {
    StfTx t[3];
    int on;  // 0: the plain [0, 1] -> u8 map
};

// nearest-neighbour source index: ((d as f64) * ratio).min((len - 1) as f64) as usize  (helpers.rs:306,309)
__device__ __forceinline__ int64_t nn_index(int64_t d, double ratio, int64_t len) {
    const double s = (double)d * ratio, hi = (double)(len - 1);
    return (int64_t)(s < hi ? s : hi);
}

template <bool ROUND>
__device__ __forceinline__ unsigned char channel_u8(float v, const Stf3 &s, int c) {
    if (s.on) return to_u8(v, s.t[c]);
    const float x = clamp01(v) * 255.0f;
    return sat_u8(ROUND ? roundf(x) : x);  // tiles round (tiles.rs:290), previews truncate (helpers.rs:243)
}

// ph x pw x 3 interleaved bytes, 4 pixels (12 bytes = 3 dwords) per lane
__global__ __launch_bounds__(kBlock) void preview_rgb_kernel(Rgb in, int64_t rows, int64_t cols, int64_t ph, int64_t pw, double y_ratio,
                                                             double x_ratio, Stf3 stf, unsigned char *__restrict__ out) {
    const int64_t npix = ph * pw;
    const int64_t p0 = ((int64_t)blockIdx.x * kBlock + threadIdx.x) * 4;
    if (p0 >= npix) return;
    unsigned char px[12];
    int64_t dy = p0 / pw, dx = p0 - dy * pw;
    const int m = (int)min((int64_t)4, npix - p0);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        if (k < m) {
            const int64_t si = nn_index(dy, y_ratio, rows) * cols + nn_index(dx, x_ratio, cols);
            px[3 * k + 0] = channel_u8<false>(in.r[si], stf, 0);
            px[3 * k + 1] = channel_u8<false>(in.g[si], stf, 1);
            px[3 * k + 2] = channel_u8<false>(in.b[si], stf, 2);
            if (++dx == pw) {
                dx = 0;
                ++dy;
            }
        } else {
            px[3 * k] = px[3 * k + 1] = px[3 * k + 2] = 0;
        }
    }
    if (m == 4) {
        unsigned int *o = reinterpret_cast<unsigned int *>(out + p0 * 3);  // p0 % 4 == 0: 12-byte records are dword aligned
#pragma unroll
        for (int w = 0; w < 3; ++w) o[w] = px[4 * w] | (px[4 * w + 1] << 8) | (px[4 * w + 2] << 16) | ((unsigned int)px[4 * w + 3] << 24);
    } else {
        for (int k = 0; k < 3 * m; ++k) out[p0 * 3 + k] = px[k];
    }
}

// ipc.rs:36-148: cleaned little-endian f32 (+ per-block min / max partials).  full: min / max over the finite inputs
// (:44-52); downsampled: over the cleaned samples (:133-135).
__global__ __launch_bounds__(kBlock) void ipc_encode_kernel(const float *__restrict__ src, int64_t rows, int64_t cols, int64_t ph, int64_t pw,
                                                            double y_ratio, double x_ratio, int full, float *__restrict__ out,
                                                            float2 *__restrict__ part) {
    const int64_t npix = ph * pw, stride = (int64_t)gridDim.x * kBlock;
    float mn = FLT_MAX, mx = -FLT_MAX;
    for (int64_t p = (int64_t)blockIdx.x * kBlock + threadIdx.x; p < npix; p += stride) {
        float v;
        if (full) {
            v = src[p];
        } else {
            const int64_t dy = p / pw, dx = p - dy * pw;
            v = src[nn_index(dy, y_ratio, rows) * cols + nn_index(dx, x_ratio, cols)];
        }
        const bool fin = __builtin_isfinite(v);
        const float clean = fin ? v : 0.0f;
        if (fin || !full) {
            mn = clean < mn ? clean : mn;
            mx = clean > mx ? clean : mx;
        }
        out[p] = clean;
    }
    __shared__ float smn[kBlock], smx[kBlock];
    smn[threadIdx.x] = mn;
    smx[threadIdx.x] = mx;
    __syncthreads();
    for (int s = kBlock / 2; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) {
            smn[threadIdx.x] = fminf(smn[threadIdx.x], smn[threadIdx.x + s]);
            smx[threadIdx.x] = fmaxf(smx[threadIdx.x], smx[threadIdx.x + s]);
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) part[blockIdx.x] = make_float2(smn[0], smx[0]);
}

__global__ __launch_bounds__(kBlock) void minmax_finite_kernel(const float *__restrict__ src, int64_t n, float2 *__restrict__ part) {
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    float mn = FLT_MAX, mx = -FLT_MAX;
    for (int64_t p = (int64_t)blockIdx.x * kBlock + threadIdx.x; p < n; p += stride) {
        const float v = src[p];
        if (__builtin_isfinite(v)) {
            mn = fminf(mn, v);
            mx = fmaxf(mx, v);
        }
    }
    __shared__ float smn[kBlock], smx[kBlock];
    smn[threadIdx.x] = mn;
    smx[threadIdx.x] = mx;
    __syncthreads();
    for (int s = kBlock / 2; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) {
            smn[threadIdx.x] = fminf(smn[threadIdx.x], smn[threadIdx.x + s]);
            smx[threadIdx.x] = fmaxf(smx[threadIdx.x], smx[threadIdx.x + s]);
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) part[blockIdx.x] = make_float2(smn[0], smx[0]);
}

// tiles.rs:41-70: mean of the finite samples of each 2x2 cell (edge cells repeat the last row / column), f64 sum in a, b, c, d order
__global__ __launch_bounds__(kBlock) void downsample2x_kernel(const float *__restrict__ src, int rows, int cols, int nr, int nc,
                                                              float *__restrict__ out) {
    const int nx = blockIdx.x * kBlock + threadIdx.x, ny = blockIdx.y;
    if (nx >= nc) return;
    const int y0 = ny * 2, y1 = min(y0 + 1, rows - 1), x0 = nx * 2, x1 = min(x0 + 1, cols - 1);
    const float q[4] = {src[(int64_t)y0 * cols + x0], src[(int64_t)y0 * cols + x1], src[(int64_t)y1 * cols + x0], src[(int64_t)y1 * cols + x1]};
    double sum = 0.0;
    int count = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k)
        if (__builtin_isfinite(q[k])) {
            sum += (double)q[k];
            ++count;
        }
    out[(int64_t)ny * nc + nx] = count > 0 ? (float)(sum / (double)count) : 0.0f;
}

// One level of the pyramid: grid (tiles, slices); every tile buffer is ts x ts (x CH) bytes, zero outside the image.
// MONO: ((v - gmin) * inv_range).round().clamp(0, 255) (tiles.rs:103-107); RGB: render_tile_rgb(_stf).
template <int CH>
__global__ __launch_bounds__(kBlock) void tile_level_kernel(Rgb in, int rows, int cols, int ts, int tile_cols, float gmin, float inv_range,
                                                            Stf3 stf, unsigned char *__restrict__ out) {
    const int tile = blockIdx.x, ty = tile / tile_cols, tx = tile - ty * tile_cols;
    const int tpix = ts * ts;
    unsigned char *dst = out + (int64_t)tile * tpix * CH;
    const bool vec = (ts & 1) == 0;  // ts^2 % 4 == 0: every tile buffer starts dword aligned
    for (int i = (blockIdx.y * kBlock + threadIdx.x) * 4; i < tpix; i += gridDim.y * kBlock * 4) {
        unsigned char px[4 * CH];
        const int m = min(4, tpix - i);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int w = i + k, ly = w / ts, lx = w - ly * ts;
            const int y = ty * ts + ly, x = tx * ts + lx;
            const bool inside = k < m && y < rows && x < cols;
            const int64_t si = inside ? (int64_t)y * cols + x : 0;
            if (CH == 1) {
                const float v = in.r[si];
                px[k] = (inside && __builtin_isfinite(v)) ? sat_u8(fminf(fmaxf(roundf((v - gmin) * inv_range), 0.0f), 255.0f)) : 0;
            } else {
                px[CH * k + 0] = inside ? channel_u8<true>(in.r[si], stf, 0) : 0;
                px[CH * k + 1] = inside ? channel_u8<true>(in.g[si], stf, 1) : 0;
                px[CH * k + 2] = inside ? channel_u8<true>(in.b[si], stf, 2) : 0;
            }
        }
        if (vec && m == 4) {
            unsigned int *o = reinterpret_cast<unsigned int *>(dst + (int64_t)i * CH);
#pragma unroll
            for (int w = 0; w < CH; ++w) o[w] = px[4 * w] | (px[4 * w + 1] << 8) | (px[4 * w + 2] << 16) | ((unsigned int)px[4 * w + 3] << 24);
        } else {
            for (int k = 0; k < CH * m; ++k) dst[(int64_t)i * CH + k] = px[k];
        }
    }
}

int grid_for(ab_ctx *ctx, int64_t items, int per_block) {
    const int64_t cap = (int64_t)(ctx->cu_count > 0 ? ctx->cu_count : 256) * 8;
    return (int)std::max<int64_t>(1, std::min<int64_t>((items + per_block - 1) / per_block, cap));
}

void preview_dims(int64_t rows, int64_t cols, int64_t max_dim, int64_t *ph, int64_t *pw, double *y_ratio, double *x_ratio) {  // helpers.rs:283-290
    if (rows <= max_dim && cols <= max_dim) {
        *ph = rows;
        *pw = cols;
        *y_ratio = *x_ratio = 1.0;
        return;
    }
    const double scale = (double)max_dim / (double)std::max(rows, cols);
    *pw = (int64_t)std::fmax(std::round((double)cols * scale), 1.0);
    *ph = (int64_t)std::fmax(std::round((double)rows * scale), 1.0);
    *y_ratio = (double)rows / (double)*ph;
    *x_ratio = (double)cols / (double)*pw;
}

int make_stf3(ab_ctx *ctx, const ab_stf_params *stf, const ab_image_stats *stats, Stf3 *out) {
    out->on = 0;
    if (!stf) return AB_OK;
    AB_CHECK(ctx, stats, "an STF needs the three channels' image stats as well");
    for (int c = 0; c < 3; ++c) out->t[c] = make_tx(&stf[c], &stats[c]);
    out->on = 1;
    return AB_OK;
}

// byte output staging: device buffers are written in place, host buffers through a temporary
struct BytesOut {
    unsigned char *dptr = nullptr;
    void *owned = nullptr;
    void *host = nullptr;
    size_t bytes = 0;
};
int bytes_begin(ab_ctx *ctx, void *out, int on_device, size_t bytes, BytesOut *b) {
    b->bytes = bytes;
    if (on_device) {
        b->dptr = (unsigned char *)out;
        return AB_OK;
    }
    AB_HIP(ctx, hipMalloc(&b->owned, std::max<size_t>(bytes, 16)));
    b->dptr = (unsigned char *)b->owned;
    b->host = out;
    return AB_OK;
}
int bytes_finish(ab_ctx *ctx, BytesOut *b, int rc) {
    if (b->host && rc == AB_OK) {
        if (hipMemcpyAsync(b->host, b->dptr, b->bytes, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess ||
            hipStreamSynchronize(ctx->stream) != hipSuccess)
            rc = ab_set_error(ctx, AB_ERR_HIP, "download of the rendered bytes failed");
    }
    if (b->owned) {
        (void)hipStreamSynchronize(ctx->stream);
        (void)hipFree(b->owned);
    }
    return rc;
}

struct Staged3 {
    StagedPlane p[3];
    int n = 0;
};
int stage3(ab_ctx *ctx, const ab_plane *r, const ab_plane *g, const ab_plane *b, Staged3 *s) {
    const ab_plane *ch[3] = {r, g, b};
    for (int c = 0; c < 3; ++c) {
        AB_CHECK(ctx, ch[c] && ch[c]->rows == r->rows && ch[c]->cols == r->cols, "the three channels must share their dims");
        const int rc = ab_stage_in(ctx, ch[c], &s->p[c]);
        if (rc != AB_OK) return rc;
        s->n = c + 1;
    }
    return AB_OK;
}
void release3(ab_ctx *ctx, Staged3 *s) {
    for (int c = 0; c < s->n; ++c) ab_stage_release(ctx, &s->p[c]);
}

int num_levels(int64_t width, int64_t height, int64_t ts) {  // tiles.rs:137-147
    const double max_dim = (double)std::max(width, height), t = (double)ts;
    if (max_dim <= t) return 1;
    return std::max(1, (int)std::ceil(std::log2(max_dim / t)) + 1);
}

size_t layout(int64_t rows, int64_t cols, int64_t ts, int channels, ab_tile_level *levels, int *n_out) {  // tiles.rs:203-246
    const int nl = num_levels(cols, rows, ts);
    int64_t r = rows, c = cols;
    for (int k = 0; k < nl; ++k) {  // k = stack index = max_level - level
        ab_tile_level &L = levels[nl - 1 - k];
        L.level = (uint64_t)(nl - 1 - k);
        L.width = (uint64_t)c;
        L.height = (uint64_t)r;
        L.cols = (uint64_t)((c + ts - 1) / ts);
        L.rows = (uint64_t)((r + ts - 1) / ts);
        L.scale_factor = 1.0 / (double)((uint64_t)1 << k);
        r = (r + 1) / 2;
        c = (c + 1) / 2;
    }
    size_t total = 0;
    for (int k = 0; k < nl; ++k) {
        levels[k].offset = total;
        total += (size_t)(levels[k].cols * levels[k].rows) * (size_t)(ts * ts) * (size_t)channels;
    }
    *n_out = nl;
    return total;
}

int percentile_bounds(ab_ctx *ctx, const float *data, int64_t n, double low_pct, double high_pct, float *lo, float *hi) {  // tiles.rs:149-178
    ab_plane_sel s;
    s.data = data;
    s.n = n;
    s.min_valid = 1e-7f;
    uint64_t count = 0;
    float v[2] = {0.0f, 0.0f};
    AB_TRY(ab_plane_select_ranks(
        ctx, s, 2,
        [&](uint64_t m, uint64_t *ranks) {
            ranks[0] = std::min<uint64_t>((uint64_t)((double)m * low_pct), m - 1);
            ranks[1] = std::min<uint64_t>((uint64_t)((double)m * high_pct), m - 1);
            return 2;
        },
        &count, v));
    if (count) {
        *lo = v[0];
        *hi = v[1];
        return AB_OK;
    }
    // no pixel above the padding threshold: find_minmax_simd's portable branch (math/simd.rs:263-271)
    const int grid = grid_for(ctx, n, kBlock * 8);
    void *d = nullptr;
    AB_TRY(ab_scratch(ctx, grid * sizeof(float2), &d));
    hipLaunchKernelGGL(minmax_finite_kernel, dim3(grid), dim3(kBlock), 0, ctx->stream, data, n, (float2 *)d);
    AB_HIP(ctx, hipGetLastError());
    std::vector<float2> part(grid);
    AB_HIP(ctx, hipMemcpyAsync(part.data(), d, grid * sizeof(float2), hipMemcpyDeviceToHost, ctx->stream));
    AB_HIP(ctx, hipStreamSynchronize(ctx->stream));
    float mn = FLT_MAX, mx = -FLT_MAX;
    for (const float2 &p : part) {
        mn = std::fmin(mn, p.x);
        mx = std::fmax(mx, p.y);
    }
    *lo = mn;
    *hi = mx;
    return AB_OK;
}

// every level of a 1- or 3-channel pyramid into the packed buffer; the 2x reductions ping-pong through one workspace
template <int CH>
int pyramid(ab_ctx *ctx, const float *const *base, const ab_tile_level *levels, int nl, int64_t ts, float gmin, float gmax, const Stf3 &stf,
            unsigned char *tiles) {
    const ab_tile_level &fine = levels[nl - 1];
    const size_t half = nl > 1 ? (size_t)levels[nl - 2].width * levels[nl - 2].height : 0;
    const size_t quarter = nl > 2 ? (size_t)levels[nl - 3].width * levels[nl - 3].height : 0;
    float *ws = nullptr;
    if (half) AB_TRY(ab_workspace(ctx, AB_WS_RENDER, (half + quarter) * CH * sizeof(float), (void **)&ws));
    const float range = std::fmax(gmax - gmin, 1e-10f), inv_range = 255.0f / range;  // tiles.rs:95-96 (f32)
    const float *cur[3] = {base[0], CH == 3 ? base[1] : base[0], CH == 3 ? base[2] : base[0]};
    (void)fine;
    for (int k = 0; k < nl; ++k) {
        const ab_tile_level &L = levels[nl - 1 - k];
        const int n_tiles = (int)(L.cols * L.rows);
        const int slices = (int)std::max<int64_t>(1, std::min<int64_t>((ts * ts + kBlock * 4 - 1) / (kBlock * 4), 2048 / std::max(n_tiles, 1) + 1));
        hipLaunchKernelGGL(tile_level_kernel<CH>, dim3(n_tiles, slices), dim3(kBlock), 0, ctx->stream, Rgb{cur[0], cur[1], cur[2]}, (int)L.height,
                           (int)L.width, (int)ts, (int)L.cols, gmin, inv_range, stf, tiles + L.offset);
        AB_HIP(ctx, hipGetLastError());
        if (k + 1 < nl) {
            const int nr = (int)((L.height + 1) / 2), nc = (int)((L.width + 1) / 2);
            float *dst = ws + ((k & 1) ? half * CH : 0);  // level k + 1 never exceeds `half` (k even) or `quarter` (k odd) pixels
            for (int c = 0; c < CH; ++c) {
                hipLaunchKernelGGL(downsample2x_kernel, dim3(ab_div_up(nc, kBlock), nr), dim3(kBlock), 0, ctx->stream, cur[c], (int)L.height, (int)L.width,
                                   nr, nc, dst + (size_t)c * nr * nc);
                AB_HIP(ctx, hipGetLastError());
            }
            for (int c = 0; c < CH; ++c) cur[c] = dst + (size_t)c * nr * nc;
        }
    }
    return AB_OK;
}

}  // namespace

extern "C" {

int ab_preview_dims(int64_t rows, int64_t cols, int64_t max_dim, int64_t *out_rows, int64_t *out_cols) try {
    if (rows <= 0 || cols <= 0 || max_dim <= 0 || !out_rows || !out_cols) return AB_ERR_INVALID;
    double yr, xr;
    preview_dims(rows, cols, max_dim, out_rows, out_cols, &yr, &xr);
    return AB_OK;
} AB_CATCH_NOCTX

int ab_render_rgb_preview(ab_ctx *ctx, const ab_plane *r, const ab_plane *g, const ab_plane *b, int64_t max_dim, const ab_stf_params *stf,
                          const ab_image_stats *stats, uint8_t *out_rgb, int32_t out_on_device) try {
    if (!ctx) return AB_ERR_INVALID;
    AB_CHECK(ctx, r && g && b && out_rgb, "null argument");
    AB_CHECK(ctx, max_dim > 0, "max_dim must be > 0");
    Stf3 s3;
    AB_TRY(make_stf3(ctx, stf, stats, &s3));
    Staged3 in;
    int rc = stage3(ctx, r, g, b, &in);
    BytesOut bo;
    if (rc == AB_OK) {
        int64_t ph, pw;
        double yr, xr;
        preview_dims(r->rows, r->cols, max_dim, &ph, &pw, &yr, &xr);
        rc = bytes_begin(ctx, out_rgb, out_on_device, (size_t)(ph * pw * 3), &bo);
        if (rc == AB_OK) {
            if (((uintptr_t)bo.dptr & 3) != 0) rc = ab_set_error(ctx, AB_ERR_INVALID, "the preview buffer must be 4-byte aligned");
        }
        if (rc == AB_OK) {
            hipLaunchKernelGGL(preview_rgb_kernel, dim3(ab_div_up(ph * pw, kBlock * 4)), dim3(kBlock), 0, ctx->stream,
                               Rgb{in.p[0].dptr, in.p[1].dptr, in.p[2].dptr}, r->rows, r->cols, ph, pw, yr, xr, s3, bo.dptr);
            if (hipGetLastError() != hipSuccess) rc = ab_set_error(ctx, AB_ERR_HIP, "preview launch failed");
        }
        rc = bytes_finish(ctx, &bo, rc);
    }
    release3(ctx, &in);
    return rc;
} AB_CATCH(ctx)

int ab_ipc_encode_with_header(ab_ctx *ctx, const ab_plane *img, int64_t max_dim, void *out, int32_t out_on_device, size_t *out_len) try {
    if (!ctx) return AB_ERR_INVALID;
    AB_CHECK(ctx, img && out, "null argument");
    AB_CHECK(ctx, max_dim >= 0, "max_dim must be >= 0 (0 = full resolution)");
    StagedPlane in;
    AB_TRY(ab_stage_in(ctx, img, &in));
    int64_t ph = in.rows, pw = in.cols;
    double yr = 1.0, xr = 1.0;
    const int full = max_dim == 0 || (in.rows <= max_dim && in.cols <= max_dim);
    if (!full) preview_dims(in.rows, in.cols, max_dim, &ph, &pw, &yr, &xr);
    const size_t bytes = 16 + (size_t)(ph * pw) * 4;
    BytesOut bo;
    int rc = bytes_begin(ctx, out, out_on_device, bytes, &bo);
    if (rc == AB_OK && ((uintptr_t)bo.dptr & 3) != 0) rc = ab_set_error(ctx, AB_ERR_INVALID, "the IPC buffer must be 4-byte aligned");
    if (rc == AB_OK) {
        const int grid = grid_for(ctx, ph * pw, kBlock * 8);
        void *d = nullptr;
        rc = ab_scratch(ctx, grid * sizeof(float2), &d);
        std::vector<float2> part(grid);
        if (rc == AB_OK) {
            hipLaunchKernelGGL(ipc_encode_kernel, dim3(grid), dim3(kBlock), 0, ctx->stream, in.dptr, in.rows, in.cols, ph, pw, yr, xr, full,
                               (float *)(bo.dptr + 16), (float2 *)d);
            if (hipGetLastError() != hipSuccess || hipMemcpyAsync(part.data(), d, grid * sizeof(float2), hipMemcpyDeviceToHost, ctx->stream) != hipSuccess ||
                hipStreamSynchronize(ctx->stream) != hipSuccess)
                rc = ab_set_error(ctx, AB_ERR_HIP, "IPC encode failed");
        }
        if (rc == AB_OK) {
            float mn = FLT_MAX, mx = -FLT_MAX;
            for (const float2 &p : part) {
                mn = std::fmin(mn, p.x);
                mx = std::fmax(mx, p.y);
            }
            struct {
                uint32_t w, h;
                float dmin, dmax;
            } hdr = {(uint32_t)pw, (uint32_t)ph, mn > mx ? 0.0f : mn, mn > mx ? 1.0f : mx};  // ipc.rs:56-57,84-91
            if (hipMemcpyAsync(bo.dptr, &hdr, 16, hipMemcpyHostToDevice, ctx->stream) != hipSuccess ||
                hipStreamSynchronize(ctx->stream) != hipSuccess)
                rc = ab_set_error(ctx, AB_ERR_HIP, "IPC header upload failed");
        }
    }
    rc = bytes_finish(ctx, &bo, rc);
    ab_stage_release(ctx, &in);
    if (rc == AB_OK && out_len) *out_len = bytes;
    return rc;
} AB_CATCH(ctx)

int ab_tile_compute_num_levels(int64_t width, int64_t height, int64_t tile_size) try {
    if (width <= 0 || height <= 0 || tile_size <= 0) return 0;
    return num_levels(width, height, tile_size);
} AB_CATCH_NOCTX

int ab_tile_pyramid_layout(int64_t rows, int64_t cols, int64_t tile_size, int32_t channels, ab_tile_level *levels, int32_t *num_levels_out,
                           size_t *total_bytes) try {
    if (rows <= 0 || cols <= 0 || tile_size <= 0 || (channels != 1 && channels != 3) || !levels || !num_levels_out) return AB_ERR_INVALID;
    if (num_levels(cols, rows, tile_size) > AB_MAX_TILE_LEVELS) return AB_ERR_INVALID;
    int nl = 0;
    const size_t total = layout(rows, cols, tile_size, channels, levels, &nl);
    *num_levels_out = nl;
    if (total_bytes) *total_bytes = total;
    return AB_OK;
} AB_CATCH_NOCTX

int ab_tile_downsample_2x(ab_ctx *ctx, const ab_plane *img, ab_plane_mut *out) try {
    if (!ctx) return AB_ERR_INVALID;
    AB_CHECK(ctx, img && out, "null argument");
    AB_CHECK(ctx, out->rows == (img->rows + 1) / 2 && out->cols == (img->cols + 1) / 2, "downsample_2x writes ((rows + 1) / 2) x ((cols + 1) / 2)");
    StagedPlane in;
    AB_TRY(ab_stage_in(ctx, img, &in));
    StagedOut so;
    int rc = ab_stage_out_begin(ctx, out, &so);
    if (rc == AB_OK) {
        hipLaunchKernelGGL(downsample2x_kernel, dim3(ab_div_up(out->cols, kBlock), (unsigned)out->rows), dim3(kBlock), 0, ctx->stream, in.dptr,
                           (int)in.rows, (int)in.cols, (int)out->rows, (int)out->cols, so.dptr);
        if (hipGetLastError() != hipSuccess) {
            ab_stage_out_abort(ctx, &so);
            rc = ab_set_error(ctx, AB_ERR_HIP, "downsample launch failed");
        } else {
            rc = ab_stage_out_finish(ctx, &so);
        }
    }
    ab_stage_release(ctx, &in);
    return rc;
} AB_CATCH(ctx)

int ab_tile_percentile_bounds(ab_ctx *ctx, const ab_plane *img, double low_pct, double high_pct, float *lo, float *hi) try {
    if (!ctx) return AB_ERR_INVALID;
    AB_CHECK(ctx, img && lo && hi, "null argument");
    StagedPlane in;
    AB_TRY(ab_stage_in(ctx, img, &in));
    const int rc = percentile_bounds(ctx, in.dptr, in.rows * in.cols, low_pct, high_pct, lo, hi);
    ab_stage_release(ctx, &in);
    return rc;
} AB_CATCH(ctx)

int ab_generate_tile_pyramid(ab_ctx *ctx, const ab_plane *normalized, int64_t tile_size, uint8_t *tiles, int32_t tiles_on_device,
                             ab_tile_level *levels, int32_t *num_levels_out, float *global_min, float *global_max) try {
    if (!ctx) return AB_ERR_INVALID;
    AB_CHECK(ctx, normalized && tiles && levels && num_levels_out, "null argument");
    AB_CHECK(ctx, tile_size > 0 && tile_size <= 4096, "tile_size must be in 1..4096");
    AB_CHECK(ctx, normalized->rows > 0 && normalized->cols > 0, "plane is null or has a zero dimension");
    AB_CHECK(ctx, num_levels(normalized->cols, normalized->rows, tile_size) <= AB_MAX_TILE_LEVELS, "more than AB_MAX_TILE_LEVELS pyramid levels");
    int nl = 0;
    const size_t total = layout(normalized->rows, normalized->cols, tile_size, 1, levels, &nl);
    *num_levels_out = nl;
    StagedPlane in;
    AB_TRY(ab_stage_in(ctx, normalized, &in));
    float gmin = 0.0f, gmax = 0.0f;
    int rc = percentile_bounds(ctx, in.dptr, in.rows * in.cols, 0.001, 0.999, &gmin, &gmax);  // tiles.rs:190
    BytesOut bo;
    if (rc == AB_OK) {
        rc = bytes_begin(ctx, tiles, tiles_on_device, total, &bo);
        if (rc == AB_OK && ((uintptr_t)bo.dptr & 3) != 0) rc = ab_set_error(ctx, AB_ERR_INVALID, "the tile buffer must be 4-byte aligned");
        if (rc == AB_OK) {
            Stf3 none;
            none.on = 0;
            const float *base[1] = {in.dptr};
            rc = pyramid<1>(ctx, base, levels, nl, tile_size, gmin, gmax, none, bo.dptr);
        }
        rc = bytes_finish(ctx, &bo, rc);
    }
    ab_stage_release(ctx, &in);
    if (global_min) *global_min = gmin;
    if (global_max) *global_max = gmax;
    return rc;
} AB_CATCH(ctx)

int ab_generate_tile_pyramid_rgb(ab_ctx *ctx, const ab_plane *r, const ab_plane *g, const ab_plane *b, int64_t tile_size, const ab_stf_params *stf,
                                 const ab_image_stats *stats, uint8_t *tiles, int32_t tiles_on_device, ab_tile_level *levels,
                                 int32_t *num_levels_out) try {
    if (!ctx) return AB_ERR_INVALID;
    AB_CHECK(ctx, r && g && b && tiles && levels && num_levels_out, "null argument");
    AB_CHECK(ctx, tile_size > 0 && tile_size <= 4096, "tile_size must be in 1..4096");
    AB_CHECK(ctx, r->rows > 0 && r->cols > 0, "plane is null or has a zero dimension");
    AB_CHECK(ctx, num_levels(r->cols, r->rows, tile_size) <= AB_MAX_TILE_LEVELS, "more than AB_MAX_TILE_LEVELS pyramid levels");
    Stf3 s3;
    AB_TRY(make_stf3(ctx, stf, stats, &s3));
    int nl = 0;
    const size_t total = layout(r->rows, r->cols, tile_size, 3, levels, &nl);
    *num_levels_out = nl;
    Staged3 in;
    int rc = stage3(ctx, r, g, b, &in);
    BytesOut bo;
    if (rc == AB_OK) {
        rc = bytes_begin(ctx, tiles, tiles_on_device, total, &bo);
        if (rc == AB_OK && ((uintptr_t)bo.dptr & 3) != 0) rc = ab_set_error(ctx, AB_ERR_INVALID, "the tile buffer must be 4-byte aligned");
        if (rc == AB_OK) {
            const float *base[3] = {in.p[0].dptr, in.p[1].dptr, in.p[2].dptr};
            rc = pyramid<3>(ctx, base, levels, nl, tile_size, 0.0f, 1.0f, s3, bo.dptr);
        }
        rc = bytes_finish(ctx, &bo, rc);
    }
    release3(ctx, &in);
    return rc;
} AB_CATCH(ctx)

}  // extern "C"
