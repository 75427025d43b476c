// FITS pixel codecs on gfx950.
//
// Replaces the per-pixel loops of infra/fits/reader.rs (decode_pixels :42-101, is_identity_scaling :36-39) and
// infra/fits/writer.rs (write_f32 / i16 / f64_slice_as_be :82-135, compute_bzero_bscale :143-159).  The data unit of a
// FITS HDU is big-endian; the reference decodes it on the host (rayon) before anything else can start.  Here the raw
// bytes are uploaded as they sit in the file (2 B per pixel for BITPIX 16 instead of 4) and decoded at HBM rate;
// the stacking kernel can also consume raw planes directly (ab_stack_sigma_clip_raw, stack_sigma_clip.hip), which
// removes the decoded copy altogether.  Arithmetic: `v as f64 * bscale + bzero` as two separately rounded f64
// operations, then `as f32`, and the identity fast path -- bit-identical to the reference.
// Header parsing, mmap and file IO stay with the caller.
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

__device__ __forceinline__ uint32_t bswap32(uint32_t v) { return __builtin_bswap32(v); }

template <int BITPIX>
__global__ __launch_bounds__(kBlock) void fits_decode_kernel(const uint8_t *__restrict__ raw, int64_t n, int identity, double bscale, double bzero,
                                                             float *__restrict__ out) {
    AB_GRID_LOOP(i, n) {
        float r;
        if constexpr (BITPIX == 8) {
            const uint8_t b = raw[i];
            r = identity ? (float)b : (float)((double)b * bscale + bzero);
        } else if constexpr (BITPIX == 16) {
            const uint16_t u = ((const uint16_t *)raw)[i];
            const int16_t v = (int16_t)(uint16_t)((u << 8) | (u >> 8));
            r = identity ? (float)v : (float)((double)v * bscale + bzero);
        } else if constexpr (BITPIX == 32) {
            const int32_t v = (int32_t)bswap32(((const uint32_t *)raw)[i]);
            r = identity ? (float)v : (float)((double)v * bscale + bzero);
        } else if constexpr (BITPIX == -32) {
            const float v = __uint_as_float(bswap32(((const uint32_t *)raw)[i]));
            r = identity ? v : (float)((double)v * bscale + bzero);
        } else {  // -64
            const uint64_t u = __builtin_bswap64(((const uint64_t *)raw)[i]);
            const double v = __longlong_as_double((long long)u);
            r = identity ? (float)v : (float)(v * bscale + bzero);
        }
        out[i] = r;
    }
}

template <int BITPIX>
__global__ __launch_bounds__(kBlock) void fits_encode_kernel(const float *__restrict__ in, int64_t n, double bzero, double bscale,
                                                             uint8_t *__restrict__ out) {
    AB_GRID_LOOP(i, n) {
        const float val = in[i];
        if constexpr (BITPIX == -32) {
            ((uint32_t *)out)[i] = bswap32(__float_as_uint(val));  // writer.rs:82-98
        } else if constexpr (BITPIX == 16) {                       // writer.rs:100-118
            const double physical = ((double)val - bzero) / bscale;
            const double c = physical < -32768.0 ? -32768.0 : (physical > 32767.0 ? 32767.0 : physical);  // f64::clamp (NaN stays)
            const double r = round(c);                                                                    // half away from zero
            const int16_t v = (r != r) ? (int16_t)0 : (int16_t)r;                                         // NaN as i16 = 0
            const uint16_t u = (uint16_t)v;
            ((uint16_t *)out)[i] = (uint16_t)((u << 8) | (u >> 8));
        } else {  // -64, writer.rs:120-135
            ((uint64_t *)out)[i] = __builtin_bswap64((uint64_t)__double_as_longlong((double)val));
        }
    }
}

// compute_bzero_bscale's scan (writer.rs:144-152): min / max of the finite pixels (order-independent, exact)
__global__ __launch_bounds__(kBlock) void finite_minmax_kernel(const float *__restrict__ in, int64_t n, double *__restrict__ part /* 2 per block */) {
    __shared__ double s_min[kBlock], s_max[kBlock];
    double mn = INFINITY, mx = -INFINITY;
    AB_GRID_LOOP(i, n) {
        const double v = (double)in[i];
        if (__builtin_isfinite(v)) {
            mn = v < mn ? v : mn;
            mx = v > mx ? v : mx;
        }
    }
    s_min[threadIdx.x] = mn;
    s_max[threadIdx.x] = mx;
    __syncthreads();
    for (int s = kBlock / 2; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) {
            s_min[threadIdx.x] = fmin(s_min[threadIdx.x], s_min[threadIdx.x + s]);
            s_max[threadIdx.x] = fmax(s_max[threadIdx.x], s_max[threadIdx.x + s]);
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        part[2 * blockIdx.x] = s_min[0];
        part[2 * blockIdx.x + 1] = s_max[0];
    }
}

int bytes_per_pixel(int64_t bitpix) {
    switch (bitpix) {
        case 8: return 1;
        case 16: return 2;
        case 32: case -32: return 4;
        case -64: return 8;
        default: return 0;
    }
}

}  // namespace

// device-to-device decode (raw must be aligned to its element size)
int ab_fits_decode_device(ab_ctx *ctx, const uint8_t *raw, int64_t n, int64_t bitpix, double bscale, double bzero, float *out) {
    if (n <= 0) return AB_OK;
    const int identity = std::fabs(bscale - 1.0) < 1e-15 && std::fabs(bzero) < 1e-15;  // reader.rs:36-39
    const dim3 grid(stream_grid(ctx, n)), block(kBlock);
    switch (bitpix) {
        case 8: hipLaunchKernelGGL(fits_decode_kernel<8>, grid, block, 0, ctx->stream, raw, n, identity, bscale, bzero, out); break;
        case 16: hipLaunchKernelGGL(fits_decode_kernel<16>, grid, block, 0, ctx->stream, raw, n, identity, bscale, bzero, out); break;
        case 32: hipLaunchKernelGGL(fits_decode_kernel<32>, grid, block, 0, ctx->stream, raw, n, identity, bscale, bzero, out); break;
        case -32: hipLaunchKernelGGL(fits_decode_kernel<-32>, grid, block, 0, ctx->stream, raw, n, identity, bscale, bzero, out); break;
        case -64: hipLaunchKernelGGL(fits_decode_kernel<-64>, grid, block, 0, ctx->stream, raw, n, identity, bscale, bzero, out); break;
        default: return ab_set_error(ctx, AB_ERR_INVALID, "unsupported BITPIX %lld", (long long)bitpix);
    }
    AB_HIP(ctx, hipGetLastError());
    return AB_OK;
}

extern "C" {

int ab_fits_decode_pixels(ab_ctx *ctx, const void *data, size_t nbytes, int32_t data_on_device, int64_t bitpix, double bscale, double bzero,
                          ab_plane_mut *out) try {
    if (!ctx) return AB_ERR_INVALID;
    AB_CHECK(ctx, out && (data || nbytes == 0), "null argument");
    const int bpp = bytes_per_pixel(bitpix);
    if (bpp == 0) return ab_set_error(ctx, AB_ERR_INVALID, "unsupported BITPIX %lld", (long long)bitpix);  // reader.rs:99: empty Vec
    const int64_t n = (int64_t)(nbytes / (size_t)bpp);  // chunks_exact drops a ragged tail
    AB_CHECK(ctx, out->rows * out->cols == n, "output plane has %lld pixels, the data unit decodes to %lld", (long long)(out->rows * out->cols),
             (long long)n);
    AB_HIP(ctx, hipSetDevice(ctx->device));
    const uint8_t *raw = (const uint8_t *)data;
    void *tmp = nullptr;
    if (!data_on_device && n > 0) {
        AB_HIP(ctx, hipMalloc(&tmp, (size_t)n * bpp));
        const hipError_t e = hipMemcpyAsync(tmp, data, (size_t)n * bpp, hipMemcpyHostToDevice, ctx->stream);
        if (e != hipSuccess) {
            (void)hipFree(tmp);
            return ab_set_error(ctx, AB_ERR_HIP, "H2D of the data unit failed: %s", hipGetErrorString(e));
        }
        raw = (const uint8_t *)tmp;
    }
    AB_CHECK(ctx, ((uintptr_t)raw % (size_t)bpp) == 0, "raw data must be aligned to its element size");
    StagedOut so;
    int rc = ab_stage_out_begin(ctx, out, &so);
    if (rc == AB_OK) {
        rc = ab_fits_decode_device(ctx, raw, n, bitpix, bscale, bzero, so.dptr);
        if (rc == AB_OK)
            rc = ab_stage_out_finish(ctx, &so);
        else
            ab_stage_out_abort(ctx, &so);
    }
    if (tmp) {
        (void)hipStreamSynchronize(ctx->stream);
        (void)hipFree(tmp);
    }
    return rc;
} AB_CATCH(ctx)

int ab_fits_compute_bzero_bscale(ab_ctx *ctx, const ab_plane *img, double *bzero, double *bscale) try {
    if (!ctx) return AB_ERR_INVALID;
    AB_CHECK(ctx, img && bzero && bscale, "null argument");
    AB_HIP(ctx, hipSetDevice(ctx->device));
    StagedPlane in;
    AB_TRY(ab_stage_in(ctx, img, &in));
    const int64_t n = in.rows * in.cols;
    const int grid = stream_grid(ctx, n);
    void *d = nullptr;
    int rc = ab_scratch(ctx, (size_t)grid * 2 * sizeof(double), &d);
    std::vector<double> h((size_t)grid * 2);
    if (rc == AB_OK) {
        hipLaunchKernelGGL(finite_minmax_kernel, dim3(grid), dim3(kBlock), 0, ctx->stream, in.dptr, n, (double *)d);
        hipError_t e = hipGetLastError();
        if (e == hipSuccess) e = hipMemcpyAsync(h.data(), d, h.size() * sizeof(double), hipMemcpyDeviceToHost, ctx->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
        if (e != hipSuccess) rc = ab_set_error(ctx, AB_ERR_HIP, "min/max scan failed: %s", hipGetErrorString(e));
    }
    ab_stage_release(ctx, &in);
    if (rc != AB_OK) return rc;
    double dmin = INFINITY, dmax = -INFINITY;
    for (int i = 0; i < grid; ++i) {
        dmin = std::fmin(dmin, h[2 * i]);
        dmax = std::fmax(dmax, h[2 * i + 1]);
    }
    if (!std::isfinite(dmin) || !std::isfinite(dmax) || std::fabs(dmax - dmin) < 1e-30) {  // writer.rs:153-155
        *bzero = 32768.0;
        *bscale = 1.0;
    } else {
        *bscale = (dmax - dmin) / 65535.0;
        *bzero = dmin + *bscale * 32768.0;
    }
    return AB_OK;
} AB_CATCH(ctx)

int ab_fits_encode_pixels(ab_ctx *ctx, const ab_plane *img, int32_t bitpix, double bzero, double bscale, void *out, int32_t out_on_device) try {
    if (!ctx) return AB_ERR_INVALID;
    AB_CHECK(ctx, img && out, "null argument");
    AB_CHECK(ctx, bitpix == -32 || bitpix == 16 || bitpix == -64, "the writer supports BITPIX -32, 16 and -64 (got %d)", (int)bitpix);
    AB_HIP(ctx, hipSetDevice(ctx->device));
    StagedPlane in;
    AB_TRY(ab_stage_in(ctx, img, &in));
    const int64_t n = in.rows * in.cols;
    const size_t bytes = (size_t)n * (size_t)bytes_per_pixel(bitpix);
    uint8_t *dst = (uint8_t *)out;
    void *tmp = nullptr;
    hipError_t e = hipSuccess;
    if (!out_on_device) {
        e = hipMalloc(&tmp, bytes);
        dst = (uint8_t *)tmp;
    }
    if (e == hipSuccess) {
        const dim3 grid(stream_grid(ctx, n)), block(kBlock);
        if (bitpix == -32)
            hipLaunchKernelGGL(fits_encode_kernel<-32>, grid, block, 0, ctx->stream, in.dptr, n, bzero, bscale, dst);
        else if (bitpix == 16)
            hipLaunchKernelGGL(fits_encode_kernel<16>, grid, block, 0, ctx->stream, in.dptr, n, bzero, bscale, dst);
        else
            hipLaunchKernelGGL(fits_encode_kernel<-64>, grid, block, 0, ctx->stream, in.dptr, n, bzero, bscale, dst);
        e = hipGetLastError();
    }
    if (e == hipSuccess && tmp) {
        e = hipMemcpyAsync(out, tmp, bytes, hipMemcpyDeviceToHost, ctx->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    }
    if (tmp) {
        (void)hipStreamSynchronize(ctx->stream);
        (void)hipFree(tmp);
    }
    ab_stage_release(ctx, &in);
    if (e != hipSuccess) return ab_set_error(ctx, AB_ERR_HIP, "FITS encode failed: %s", hipGetErrorString(e));
    return AB_OK;
} AB_CATCH(ctx)

}  // extern "C"
