// Sub-pixel registration resampling on gfx950: bicubic (Catmull-Rom) shift and affine warp.
//
// Replaces core/stacking/align.rs:36-57 (shift_image_subpixel),
// core/alignment/affine.rs:663-690 (warp_image),
// core/imaging/resample.rs:25-61 (resample_image) and the sampler they share,
// core/imaging/sampling.rs:4-14,51-80 (catmull_rom, bicubic_sample) with
// core/imaging/boundary.rs:9-20 (clamp_index).
//
// One lane per OUTPUT pixel, consecutive lanes on consecutive x: the 4x4 source footprint of
// neighbouring lanes overlaps almost completely, so the 16 taps are served by L1/L2 and HBM sees
// each source line about once (4*P read + 4*P written per frame).  Coordinates, weights and the
// accumulation are f64 in the reference's evaluation order, so results are bit-identical to the
// CPU restatement; only the final store is f32.
#include "ab_common.hpp"
#include <algorithm>
#include <climits>
#include <cmath>

namespace {

// catmull_rom (sampling.rs:4-14) for the four taps of one axis, f in [0, 1]:
//   w0 = cr(f + 1), w1 = cr(f), w2 = cr(f - 1), w3 = cr(f - 2).
// |f + 1| and |f - 2| lie in [1, 2], |f| and |f - 1| in [0, 1], so the reference's two branches
// are known statically; where the two ranges touch (|t| = 1) both polynomials give exactly 0.0,
// so taking the other branch there is still bit-identical.  Operation order as in the reference.
__device__ __forceinline__ double cr_inner(double t) { return t * t * (1.5 * t - 2.5) + 1.0; }            // |t| <= 1
__device__ __forceinline__ double cr_outer(double t) { return t * (t * (2.5 - 0.5 * t) - 4.0) + 2.0; }    // 1 <= |t| <= 2

__device__ __forceinline__ void catmull_weights(double f, double &w0, double &w1, double &w2, double &w3) {
    w0 = cr_outer(fabs(f + 1.0));
    w1 = cr_inner(fabs(f));
    w2 = cr_inner(fabs(f - 1.0));
    w3 = cr_outer(fabs(f - 2.0));
}

__device__ __forceinline__ int clamp_i32(int idx, int len) {  // boundary.rs:9-20
    return idx < 0 ? 0 : (idx >= len ? len - 1 : idx);
}

// sampling.rs:51-80 with 32-bit indexing (callers guarantee rows*cols < 2^31 and |x|,|y| sane).
// `row_val += s*w` starts from 0.0, so the first add is exact and is skipped.
__device__ __forceinline__ float bicubic_taps(const float *__restrict__ src, int rows, int cols, int ld, int ix, int iy,
                                              double wx0, double wx1, double wx2, double wx3, double wy0, double wy1,
                                              double wy2, double wy3) {
    const int c0 = clamp_i32(ix - 1, cols), c1 = clamp_i32(ix, cols), c2 = clamp_i32(ix + 1, cols),
              c3 = clamp_i32(ix + 2, cols);
    const int r0 = clamp_i32(iy - 1, rows) * ld, r1 = clamp_i32(iy, rows) * ld, r2 = clamp_i32(iy + 1, rows) * ld,
              r3 = clamp_i32(iy + 2, rows) * ld;
    const double wy[4] = {wy0, wy1, wy2, wy3};
    const int rr[4] = {r0, r1, r2, r3};
    double val = 0.0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const float *row = src + rr[j];
        double row_val = (double)row[c0] * wx0;
        row_val += (double)row[c1] * wx1;
        row_val += (double)row[c2] * wx2;
        row_val += (double)row[c3] * wx3;
        const double t = row_val * wy[j];
        val = (j == 0) ? t : val + t;
    }
    return (float)val;
}

// the same sum when the whole 4 x 4 footprint is inside the image: no index clamps, one address per row
__device__ __forceinline__ float bicubic_taps_interior(const float *__restrict__ src, int ld, int ix, int iy, double wx0, double wx1,
                                                       double wx2, double wx3, double wy0, double wy1, double wy2, double wy3) {
    const float *p = src + (iy - 1) * ld + (ix - 1);
    const double wy[4] = {wy0, wy1, wy2, wy3};
    double val = 0.0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const float *row = p + j * ld;
        double row_val = (double)row[0] * wx0;
        row_val += (double)row[1] * wx1;
        row_val += (double)row[2] * wx2;
        row_val += (double)row[3] * wx3;
        const double t = row_val * wy[j];
        val = (j == 0) ? t : val + t;
    }
    return (float)val;
}

// `inside` (wave-uniform): every active lane's footprint [ix-1, ix+2] x [iy-1, iy+2] lies inside the image
__device__ __forceinline__ float bicubic_sample(const float *__restrict__ src, int rows, int cols, int ld, double y, double x,
                                                bool sample = true) {
    const double xf = floor(x), yf = floor(y);
    double wx0, wx1, wx2, wx3, wy0, wy1, wy2, wy3;
    catmull_weights(x - xf, wx0, wx1, wx2, wx3);
    catmull_weights(y - yf, wy0, wy1, wy2, wy3);
    const int ix = (int)xf, iy = (int)yf;
    const bool interior = ix >= 1 && ix + 2 < cols && iy >= 1 && iy + 2 < rows;
    if (__all(interior || !sample)) {  // the usual case away from the frame edges
        if (!sample) return 0.0f;
        return bicubic_taps_interior(src, ld, ix, iy, wx0, wx1, wx2, wx3, wy0, wy1, wy2, wy3);
    }
    if (!sample) return 0.0f;
    return bicubic_taps(src, rows, cols, ld, ix, iy, wx0, wx1, wx2, wx3, wy0, wy1, wy2, wy3);
}

// Which piece of which row a workgroup takes.  The hardware hands workgroup ids round the eight XCDs (id % 8), and a piece needs four
// source rows of which the piece below shares three.  When a row is a multiple of 8 pieces wide (4096 columns in 512-pixel pieces)
// the row-major ids are XCD-coherent by themselves -- every XCD walks down one column strip -- and nothing is done.  Otherwise
// (12 451 columns = 25 pieces) the piece below lands in another XCD's L2 and every source row is fetched into up to four of them:
// the row-major sequence of pieces is then cut into eight contiguous runs and XCD c walks run c, a band of rows (a bijection: XCD j
// receives ids j, j + 8, ...: (total - j + 7) / 8 of them).  Measured, interleaved on one box (profiles/r06_warp_xcd_bands.txt): C3
// 22.8 / 23.4 / 23.5 ms per step against 23.6 / 23.8 / 23.6 in id order.  Applied to the 8-pieces-wide bench step as well it cost 0.1 ms
// (the index arithmetic is ~5 % of this kernel's workgroup, and eight XCDs reading rows 512 apart meet in the same memory channels);
// column-major runs -- the 8-wide order generalised -- gained nothing on C3.
__device__ __forceinline__ void xcd_band_piece(unsigned int &piece_x, unsigned int &piece_y) {
    piece_x = blockIdx.x;
    piece_y = blockIdx.y;
#ifndef AB_WARP_ROW_MAJOR
    if ((gridDim.x & 7u) != 0) {  // (uniform)
        const unsigned int gx = gridDim.x, total = gx * gridDim.y, id = blockIdx.y * gx + blockIdx.x;
        const unsigned int c = id & 7u, k = id >> 3;
        unsigned int start = 0;
        for (unsigned int j = 0; j < c; ++j) start += (total - j + 7u) >> 3;
        const unsigned int t = start + k;
        piece_y = t / gx;
        piece_x = t - piece_y * gx;
    }
#endif
}

// align.rs:46-55
__global__ __launch_bounds__(256) void shift_kernel(const float *__restrict__ src, int rows, int cols, int ld, double dy,
                                                    double dx, float *__restrict__ out) {
    unsigned int piece_x, piece_y;
    xcd_band_piece(piece_x, piece_y);
    const int x = piece_x * 256 + threadIdx.x;
    const int y = piece_y;
    if (x >= cols) return;
    const double sy = (double)y + dy;
    const double sx = (double)x + dx;
    float r = 0.0f;
    if (!(sy < -0.5 || sy > (double)rows - 0.5 || sx < -0.5 || sx > (double)cols - 0.5))
        r = bicubic_sample(src, rows, cols, ld, sy, sx);
    out[(size_t)y * cols + x] = r;
}

// The four Catmull-Rom weights of one axis with one operation less per outer weight: 0.5 * t is exact (a power of two), so
// RN(2.5 - RN(0.5 t)) = RN(2.5 - 0.5 t) = fma(-0.5, t, 2.5) -- the only fusion in this file, and it changes no bit.
__device__ __forceinline__ double cr_outer_fused(double t) { return t * (t * __builtin_fma(-0.5, t, 2.5) - 4.0) + 2.0; }
__device__ __forceinline__ void catmull_weights_fused(double f, double &w0, double &w1, double &w2, double &w3) {
    w0 = cr_outer_fused(fabs(f + 1.0));
    w1 = cr_inner(fabs(f));
    w2 = cr_inner(fabs(f - 1.0));
    w3 = cr_outer_fused(fabs(f - 2.0));
}

// affine.rs:674-687; map() is affine.rs:74-80.
// The reference's test `sx >= 0 && sy >= 0 && sx < cols - 1 && sy < rows - 1` is taken on the floors the sampler needs anyway:
// for finite sx, 0 <= sx < cols - 1  <=>  0 <= floor(sx) <= cols - 2 (cols - 1 is an integer; floor(-0.0) = -0.0 converts to
// 0 and -0.0 >= 0.0 holds; v_cvt_i32_f64 saturates, so a floor outside the int range fails the unsigned compare like the
// f64 compare would) -- two integer compares instead of four f64 compares and eight selects on the coordinates.  A NaN
// coordinate would convert to 0 and pass: GUARD adds the two self-compares, and the host takes that instance whenever the
// coefficients could overflow (|coefficient| > 1e150 or non-finite), so that products and sums of the plain one are finite.
template <bool GUARD, bool NOLOAD = false, bool NOCVT = false>
__global__ __launch_bounds__(256) void warp_kernel(const float *__restrict__ src, int src_rows, int src_cols, double a,
                                                   double b, double tx, double c, double d, double ty, int out_rows,
                                                   int out_cols, float *__restrict__ out, int row0) {
    // two output pixels per lane (x and x + 256): their f64 chains are independent, which is the only instruction-level
    // parallelism this f64-bound kernel can get
#ifdef AB_WARP_WAVE_PRIO
    __builtin_amdgcn_s_setprio(AB_WARP_WAVE_PRIO);
#endif
    unsigned int piece_x, piece_y;  // (the pieces of a row band per XCD: xcd_band_piece)
    xcd_band_piece(piece_x, piece_y);
    const int x0 = piece_x * 512 + threadIdx.x;
    // row0: the band of output rows [row0, row0 + gridDim.y) this launch produces (row-band sharding); the
    // coordinate arithmetic uses the row's index in the WHOLE output, so a band is bit-identical to the same rows of a full warp
    const int y = piece_y + row0;
    const double yf = (double)y;
    float r[2];
    bool live[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int x = x0 + 256 * u;
        live[u] = x < out_cols;
        const double xf = (double)x;
        const double sx = a * xf + b * yf + tx;
        const double sy = c * xf + d * yf + ty;
        const double fx = floor(sx), fy = floor(sy);
        int ix, iy;  // the instruction itself (saturating; NaN -> 0): a C++ cast of an out-of-range double is undefined
        asm("v_cvt_i32_f64 %0, %1" : "=v"(ix) : "v"(fx));
        asm("v_cvt_i32_f64 %0, %1" : "=v"(iy) : "v"(fy));
        bool in = live[u] && (unsigned)ix < (unsigned)(src_cols - 1) && (unsigned)iy < (unsigned)(src_rows - 1);
        if constexpr (GUARD) in = in && sx == sx && sy == sy;
        double wx0, wx1, wx2, wx3, wy0, wy1, wy2, wy3;
        catmull_weights_fused(sx - fx, wx0, wx1, wx2, wx3);  // (of no consequence where `in` is false)
        catmull_weights_fused(sy - fy, wy0, wy1, wy2, wy3);
        const bool interior = ix >= 1 && ix + 2 < src_cols && iy >= 1 && iy + 2 < src_rows;
        r[u] = 0.0f;
        if constexpr (NOLOAD) {  // developer timing experiment (AB_ABLATE_WARP=2): the same arithmetic on taps that come from registers
            double val = 0.0;
            const double wy[4] = {wy0, wy1, wy2, wy3};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float s0 = __int_as_float(0x3f800000 | ((ix + 4 * j) & 0xffff)), s1 = __int_as_float(0x3f800000 | ((ix + 4 * j + 1) & 0xffff)),
                      s2 = __int_as_float(0x3f800000 | ((iy + 4 * j + 2) & 0xffff)), s3 = __int_as_float(0x3f800000 | ((iy + 4 * j + 3) & 0xffff));
                double d0 = (double)s0, d1 = (double)s1, d2 = (double)s2, d3 = (double)s3;
                if constexpr (NOCVT) {  // (AB_ABLATE_WARP=3: the upper bound of an LDS-staged f64 tile -- 64-bit words straight from registers)
                    d0 = __longlong_as_double(0x3ff0000000000000ll | (long long)(unsigned)(ix + 4 * j));
                    d1 = __longlong_as_double(0x3ff0000000000000ll | (long long)(unsigned)(ix + 4 * j + 1));
                    d2 = __longlong_as_double(0x3ff0000000000000ll | (long long)(unsigned)(iy + 4 * j + 2));
                    d3 = __longlong_as_double(0x3ff0000000000000ll | (long long)(unsigned)(iy + 4 * j + 3));
                }
                double row_val = d0 * wx0;
                row_val += d1 * wx1;
                row_val += d2 * wx2;
                row_val += d3 * wx3;
                const double t = row_val * wy[j];
                val = (j == 0) ? t : val + t;
            }
            r[u] = in ? (float)val : 0.0f;
        } else if (__all(interior || !in)) {  // the usual case away from the frame edges
            if (in) r[u] = bicubic_taps_interior(src, src_cols, ix, iy, wx0, wx1, wx2, wx3, wy0, wy1, wy2, wy3);
        } else if (in) {
            r[u] = bicubic_taps(src, src_rows, src_cols, src_cols, ix, iy, wx0, wx1, wx2, wx3, wy0, wy1, wy2, wy3);
        }
    }
#pragma unroll
    for (int u = 0; u < 2; ++u)
        if (live[u]) out[(size_t)piece_y * out_cols + x0 + 256 * u] = r[u];
}

// resample.rs:41-58: target pixel centres mapped onto the source grid
__global__ __launch_bounds__(256) void resample_kernel(const float *__restrict__ src, int src_rows, int src_cols, double scale_y,
                                                       double scale_x, double half_shift_y, double half_shift_x, int out_cols,
                                                       float *__restrict__ out) {
    unsigned int piece_x, piece_y;
    xcd_band_piece(piece_x, piece_y);
    const int x = piece_x * 256 + threadIdx.x;
    const int y = piece_y;
    if (x >= out_cols) return;
    const double sy = (double)y * scale_y + half_shift_y;
    const double sx = (double)x * scale_x + half_shift_x;
    out[(size_t)y * out_cols + x] = bicubic_sample(src, src_rows, src_cols, src_cols, sy, sx);
}

}  // namespace

// resample_image (core/imaging/resample.rs:25-61) on device planes
int ab_resample_device(ab_ctx *ctx, const float *src, int64_t src_rows, int64_t src_cols, int64_t out_rows, int64_t out_cols,
                       float *out) {
    AB_HIP(ctx, hipSetDevice(ctx->device));
    if (out_rows == 0 || out_cols == 0) return ab_set_error(ctx, AB_ERR_INVALID, "Target dimensions must be > 0");  // :32-34
    if (out_rows == src_rows && out_cols == src_cols) {  // :36-38
        if (src != out) AB_HIP(ctx, hipMemcpyAsync(out, src, (size_t)src_rows * src_cols * sizeof(float), hipMemcpyDeviceToDevice, ctx->stream));
        return AB_OK;
    }
    AB_CHECK(ctx, src != out, "resample_image cannot run in place");
    AB_CHECK(ctx, src_rows > 0 && src_cols > 0, "resample_image: empty source");
    AB_CHECK(ctx, out_rows <= 65535 && out_rows * out_cols < (int64_t(1) << 31) && src_rows * src_cols < (int64_t(1) << 31),
             "image of %lld x %lld needs a tiled launch (not in this build)", (long long)out_rows, (long long)out_cols);
    const double scale_y = (double)src_rows / (double)out_rows, scale_x = (double)src_cols / (double)out_cols;
    const dim3 grid((unsigned)((out_cols + 255) / 256), (unsigned)out_rows), block(256);
    hipLaunchKernelGGL(resample_kernel, grid, block, 0, ctx->stream, src, (int)src_rows, (int)src_cols, scale_y, scale_x,
                       (scale_y - 1.0) * 0.5, (scale_x - 1.0) * 0.5, (int)out_cols, out);
    AB_HIP(ctx, hipGetLastError());
    return AB_OK;
}

// src is a rows x cols window with row stride src_ld (>= cols); out is contiguous rows x cols
int ab_shift_device(ab_ctx *ctx, const float *src, int64_t rows, int64_t cols, int64_t src_ld, double dy, double dx, float *out) {
    AB_HIP(ctx, hipSetDevice(ctx->device));
    if (fabs(dy) < 1e-12 && fabs(dx) < 1e-12) {  // align.rs:37-39: image.clone()
        if (src != out)
            AB_HIP(ctx, hipMemcpy2DAsync(out, cols * sizeof(float), src, src_ld * sizeof(float), cols * sizeof(float), rows,
                                         hipMemcpyDeviceToDevice, ctx->stream));
        return AB_OK;
    }
    AB_CHECK(ctx, src != out, "shift_image_subpixel cannot run in place");
    AB_CHECK(ctx, rows <= 65535 && rows * src_ld < (int64_t(1) << 31),
             "image of %lld x %lld needs a tiled launch (not in this build)", (long long)rows, (long long)cols);
    const dim3 grid((unsigned)((cols + 255) / 256), (unsigned)rows), block(256);
    hipLaunchKernelGGL(shift_kernel, grid, block, 0, ctx->stream, src, (int)rows, (int)cols, (int)src_ld, dy, dx, out);
    AB_HIP(ctx, hipGetLastError());
    return AB_OK;
}

int ab_warp_device(ab_ctx *ctx, const float *src, int64_t src_rows, int64_t src_cols, const double t[6], int64_t out_rows,
                   int64_t out_cols, float *out) {
    return ab_warp_rows_device(ctx, src, src_rows, src_cols, t, out_rows, out_cols, 0, out_rows, out);
}

// rows [row0, row0 + nrows) of warp_image(src, t, out_rows, out_cols) into the contiguous nrows x out_cols plane `out`
constexpr long kWarpLdsKB = 0;  // (see ab_warp_rows_device)
int ab_warp_rows_device(ab_ctx *ctx, const float *src, int64_t src_rows, int64_t src_cols, const double t[6], int64_t out_rows,
                        int64_t out_cols, int64_t row0, int64_t nrows, float *out) {
    AB_HIP(ctx, hipSetDevice(ctx->device));
    AB_CHECK(ctx, row0 >= 0 && nrows >= 0 && row0 + nrows <= out_rows, "row band [%lld, %lld) leaves the %lld output rows", (long long)row0,
             (long long)(row0 + nrows), (long long)out_rows);
    if (nrows == 0) return AB_OK;
    AB_CHECK(ctx, src != out, "warp_image cannot run in place");
    AB_CHECK(ctx, out_rows <= 65535 && out_rows * out_cols < (int64_t(1) << 31) && src_rows * src_cols < (int64_t(1) << 31),
             "image of %lld x %lld needs a tiled launch (not in this build)", (long long)out_rows, (long long)out_cols);
    const dim3 grid((unsigned)((out_cols + 511) / 512), (unsigned)nrows), block(256);
#ifdef AB_DEV_ABLATION  // developer timing experiments, compiled only into a -DAB_DEV_ABLATION build (profiles/r04_warp_ablation.txt)
    static const int ablate = ab_dev_env("AB_ABLATE_WARP") ? atoi(ab_dev_env("AB_ABLATE_WARP")) : 0;  // developer timing experiments
    if (ablate == 1) return AB_OK;  // what the registration stage takes without the warps
    if (ablate == 3) {              // ... and without the sixteen conversions and the address arithmetic
        const dim3 g2((unsigned)((out_cols + 511) / 512), (unsigned)nrows);
        hipLaunchKernelGGL((warp_kernel<false, true, true>), g2, dim3(256), 0, ctx->stream, src, (int)src_rows, (int)src_cols, t[0], t[1], t[2], t[3],
                           t[4], t[5], (int)out_rows, (int)out_cols, out, (int)row0);
        return AB_OK;
    }
    if (ablate == 2) {              // what the kernel takes without its loads
        const dim3 g2((unsigned)((out_cols + 511) / 512), (unsigned)nrows);
        hipLaunchKernelGGL((warp_kernel<false, true>), g2, dim3(256), 0, ctx->stream, src, (int)src_rows, (int)src_cols, t[0], t[1], t[2], t[3],
                           t[4], t[5], (int)out_rows, (int)out_cols, out, (int)row0);
        return AB_OK;
    }
#endif
    bool tame = true;  // every product and sum of the coordinate arithmetic stays finite (x, y < 2^16)
    for (int i = 0; i < 6; ++i) tame = tame && std::isfinite(t[i]) && fabs(t[i]) <= 1e150;
    // Occupancy cap (AB_WARP_LDS_KB, default below): the kernel is bound by the f64 pipe, which two to four waves per SIMD with two
    // independent chains per lane keep busy; left alone its 28-VGPR waves take EVERY wave slot of the chip and the registration
    // batch's latency-bound estimate kernels (dependent loads, a few VALU cycles between them) wait for slots instead of running in
    // the f64 instructions' shadow.  Unused dynamic LDS per workgroup is the cap: 160 KB / n KB workgroups of 4 waves per CU.
    static const unsigned warp_lds = [] {
        const char *e = ab_dev_env("AB_WARP_LDS_KB");
        const long kb = e ? atol(e) : kWarpLdsKB;
        return (unsigned)(kb < 0 ? 0 : (kb > 64 ? 64 : kb)) * 1024u;
    }();
    if (tame)
        hipLaunchKernelGGL(warp_kernel<false>, grid, block, warp_lds, ctx->stream, src, (int)src_rows, (int)src_cols, t[0], t[1], t[2], t[3],
                           t[4], t[5], (int)out_rows, (int)out_cols, out, (int)row0);
    else
        hipLaunchKernelGGL(warp_kernel<true>, grid, block, warp_lds, ctx->stream, src, (int)src_rows, (int)src_cols, t[0], t[1], t[2], t[3],
                           t[4], t[5], (int)out_rows, (int)out_cols, out, (int)row0);
    AB_HIP(ctx, hipGetLastError());
    return AB_OK;
}

extern "C" {

int ab_shift_image_subpixel(ab_ctx *ctx, const ab_plane *src, double dy, double dx, ab_plane_mut *out) try {
    if (!ctx) return AB_ERR_INVALID;
    AB_CHECK(ctx, src && out, "null plane");
    AB_CHECK(ctx, src->rows == out->rows && src->cols == out->cols, "shift_image_subpixel keeps the image dims");
    StagedPlane in;
    AB_TRY(ab_stage_in(ctx, src, &in));
    StagedOut so;
    int rc = ab_stage_out_begin(ctx, out, &so);
    if (rc == AB_OK) {
        rc = ab_shift_device(ctx, in.dptr, in.rows, in.cols, in.cols, dy, dx, so.dptr);
        if (rc == AB_OK)
            rc = ab_stage_out_finish(ctx, &so);
        else
            ab_stage_out_abort(ctx, &so);
    }
    ab_stage_release(ctx, &in);
    return rc;
} AB_CATCH(ctx)

// rows [row0, row0 + out_band->rows) of warp_image(src, transform, out_rows, out_band->cols) (device planes)
int ab_warp_image_rows(ab_ctx *ctx, const ab_plane *src, const double transform[6], int64_t out_rows, int64_t row0, ab_plane_mut *out_band) try {
    if (!ctx) return AB_ERR_INVALID;
    AB_CHECK(ctx, src && out_band && transform && src->data && out_band->data, "null plane or transform");
    AB_CHECK(ctx, src->on_device && out_band->on_device, "ab_warp_image_rows takes device-resident planes");
    return ab_warp_rows_device(ctx, src->data, src->rows, src->cols, transform, out_rows, out_band->cols, row0, out_band->rows, out_band->data);
} AB_CATCH(ctx)

// Source rows that rows [row0, row0 + nrows) of warp_image(src, transform, out_rows, out_cols) read (affine.rs:663-690 +
// sampling.rs:48-80: an output pixel whose source point passes 0 <= sy < rows - 1 reads rows floor(sy) - 1 .. floor(sy) + 2,
// clamped to the frame).  sy = c x + d y + ty is evaluated here with the kernel's own expression; every operation of it is
// monotone in x and in y (a correctly rounded product and sum are monotone in each operand), so its extremes over the band's
// rectangle are taken at the four corners and the interval below is exactly the hull of what the kernel touches -- not an
// estimate with a safety margin.  A non-finite coefficient asks for the whole frame.
static void warp_source_rows(const double t[6], int64_t src_rows, int64_t src_cols, int64_t out_cols, int64_t row0, int64_t nrows,
                             int64_t *s0, int64_t *sn) {
    *s0 = 0;
    *sn = 0;
    if (nrows <= 0 || out_cols <= 0 || src_rows < 2 || src_cols < 2) return;  // (nothing passes `sy < rows - 1` on a one-row frame)
    for (int i = 0; i < 6; ++i)
        if (!std::isfinite(t[i]) || fabs(t[i]) > 1e150) {
            *sn = src_rows;
            return;
        }
    double lo = INFINITY, hi = -INFINITY;
    const double xs[2] = {0.0, (double)(out_cols - 1)}, ys[2] = {(double)row0, (double)(row0 + nrows - 1)};
    for (double xf : xs)
        for (double yf : ys) {
            const double sy = t[3] * xf + t[4] * yf + t[5];  // the kernel's `c * xf + d * yf + ty` (-ffp-contract=off on both sides)
            lo = std::min(lo, sy);
            hi = std::max(hi, sy);
        }
    const double last = (double)(src_rows - 2);  // floor(sy) of a pixel that samples lies in [0, rows - 2]
    if (!(hi >= 0.0) || !(floor(lo) <= last)) return;
    const int64_t f_lo = (int64_t)std::max(floor(lo), 0.0), f_hi = (int64_t)std::min(floor(hi), last);
    const int64_t a = std::max<int64_t>(f_lo - 1, 0), b = std::min<int64_t>(f_hi + 2, src_rows - 1);
    *s0 = a;
    *sn = b - a + 1;
}

int ab_warp_source_rows(const double transform[6], int64_t src_rows, int64_t src_cols, int64_t out_cols, int64_t row0, int64_t nrows,
                        int64_t *src_row0, int64_t *src_nrows) try {
    if (!transform || !src_row0 || !src_nrows || src_rows < 0 || src_cols < 0 || row0 < 0 || nrows < 0) return AB_ERR_INVALID;
    warp_source_rows(transform, src_rows, src_cols, out_cols, row0, nrows, src_row0, src_nrows);
    return AB_OK;
} AB_CATCH_NOCTX

// the hull of ab_warp_source_rows over n transforms for rank `rank`'s band of the output (ab_shard_rows): what a rank of the
// row-band scheme must hold of every target frame ("rows [g R / G, (g + 1) R / G) of every frame (+ halo)")
int ab_shard_source_rows(const double *transforms, size_t n, int64_t src_rows, int64_t src_cols, int64_t out_rows, int64_t out_cols, int nranks,
                         int rank, int64_t *src_row0, int64_t *src_nrows) try {
    if ((!transforms && n) || !src_row0 || !src_nrows) return AB_ERR_INVALID;
    int64_t row0 = 0, nrows = 0;
    if (ab_shard_rows(out_rows, nranks, rank, &row0, &nrows) != AB_OK) return AB_ERR_INVALID;
    int64_t lo = INT64_MAX, hi = INT64_MIN;
    for (size_t i = 0; i < n; ++i) {
        int64_t s0 = 0, sn = 0;
        warp_source_rows(transforms + 6 * i, src_rows, src_cols, out_cols, row0, nrows, &s0, &sn);
        if (sn > 0) {
            lo = std::min(lo, s0);
            hi = std::max(hi, s0 + sn);
        }
    }
    *src_row0 = lo <= hi ? lo : 0;
    *src_nrows = lo <= hi ? hi - lo : 0;
    return AB_OK;
} AB_CATCH_NOCTX

// ab_warp_image_rows with the SOURCE given as a band: src_band holds rows [src_row0, src_row0 + src_band->rows) of a frame of
// src_rows rows.  The band must cover ab_warp_source_rows of the request; the result is then bit-identical to the same rows of
// ab_warp_image on the whole frame (same kernel, same coordinates: the band's base pointer is moved back by src_row0 rows, so
// every tap address is the one the whole frame would give, and none outside the band is formed).
int ab_warp_image_rows_from_band(ab_ctx *ctx, const ab_plane *src_band, int64_t src_row0, int64_t src_rows, const double transform[6],
                                 int64_t out_rows, int64_t row0, ab_plane_mut *out_band) try {
    if (!ctx) return AB_ERR_INVALID;
    AB_CHECK(ctx, src_band && out_band && transform && (out_band->rows == 0 || out_band->data), "null plane or transform");
    AB_CHECK(ctx, src_row0 >= 0 && src_band->rows >= 0 && src_row0 + src_band->rows <= src_rows, "source band [%lld, %lld) leaves the frame's %lld rows",
             (long long)src_row0, (long long)(src_row0 + src_band->rows), (long long)src_rows);
    if (out_band->rows == 0) return AB_OK;
    AB_CHECK(ctx, out_band->on_device && (src_band->rows == 0 || (src_band->data && src_band->on_device)), "ab_warp_image_rows_from_band takes device-resident planes");
    int64_t need0 = 0, need_n = 0;
    warp_source_rows(transform, src_rows, src_band->cols, out_band->cols, row0, out_band->rows, &need0, &need_n);
    AB_CHECK(ctx, need_n == 0 || (need0 >= src_row0 && need0 + need_n <= src_row0 + src_band->rows),
             "output rows [%lld, %lld) read source rows [%lld, %lld); the band holds [%lld, %lld)", (long long)row0, (long long)(row0 + out_band->rows),
             (long long)need0, (long long)(need0 + need_n), (long long)src_row0, (long long)(src_row0 + src_band->rows));
    // (no source row needed: every pixel of the band is 0.0; the kernel then forms no address at all, whatever the base)
    const float *base = src_band->data ? src_band->data - src_row0 * src_band->cols : nullptr;
    if (!base) {  // an empty band and nothing to read: hand the kernel any valid pointer
        base = out_band->data + 1;
    }
    return ab_warp_rows_device(ctx, base, src_rows, src_band->cols, transform, out_rows, out_band->cols, row0, out_band->rows, out_band->data);
} AB_CATCH(ctx)

int ab_warp_image(ab_ctx *ctx, const ab_plane *src, const double transform[6], ab_plane_mut *out) try {
    if (!ctx) return AB_ERR_INVALID;
    AB_CHECK(ctx, src && out && transform, "null plane or transform");
    StagedPlane in;
    AB_TRY(ab_stage_in(ctx, src, &in));
    StagedOut so;
    int rc = ab_stage_out_begin(ctx, out, &so);
    if (rc == AB_OK) {
        rc = ab_warp_device(ctx, in.dptr, in.rows, in.cols, transform, out->rows, out->cols, so.dptr);
        if (rc == AB_OK)
            rc = ab_stage_out_finish(ctx, &so);
        else
            ab_stage_out_abort(ctx, &so);
    }
    ab_stage_release(ctx, &in);
    return rc;
} AB_CATCH(ctx)

int ab_resample_image(ab_ctx *ctx, const ab_plane *src, ab_plane_mut *out) try {
    if (!ctx) return AB_ERR_INVALID;
    AB_CHECK(ctx, src && out, "null plane");
    if (out->rows == 0 || out->cols == 0) return ab_set_error(ctx, AB_ERR_INVALID, "Target dimensions must be > 0");  // :32-34
    StagedPlane in;
    AB_TRY(ab_stage_in(ctx, src, &in));
    StagedOut so;
    int rc = ab_stage_out_begin(ctx, out, &so);
    if (rc == AB_OK) {
        rc = ab_resample_device(ctx, in.dptr, in.rows, in.cols, out->rows, out->cols, so.dptr);
        if (rc == AB_OK)
            rc = ab_stage_out_finish(ctx, &so);
        else
            ab_stage_out_abort(ctx, &so);
    }
    ab_stage_release(ctx, &in);
    return rc;
} AB_CATCH(ctx)

}  // extern "C"
