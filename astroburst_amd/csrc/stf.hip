// Screen transfer function (midtone stretch) on gfx950.
//
// Replaces core/imaging/stf.rs: auto_stf (:13-39), mtf_balance (:41-47), mtf (:50-58),
// StfTransform (:60-87), apply_stf -> u8 (:89-102), apply_stf_f32 (:104-120) and
// apply_stf_inplace (:147-155).
//
// Pure streaming map: 4 B read, 1 B (u8) or 4 B (f32) written per pixel, float4 / uchar4 wide.
// The per-pixel arithmetic is f64 in the reference's evaluation order, so the u8 output is
// bit-exact against the CPU restatement.
#include "ab_common.hpp"
#include "stf_device.hpp"

#include <cmath>

namespace {

__global__ __launch_bounds__(256) void stf_u8_kernel(const float *__restrict__ in, int64_t n, StfTx t,
                                                     unsigned char *__restrict__ out) {
    const int64_t n4 = n >> 2;
    const int64_t stride = (int64_t)gridDim.x * 256;
    const float4 *in4 = reinterpret_cast<const float4 *>(in);
    uchar4 *out4 = reinterpret_cast<uchar4 *>(out);
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) {
        const float4 v = in4[i];
        uchar4 r;
        r.x = to_u8(v.x, t);
        r.y = to_u8(v.y, t);
        r.z = to_u8(v.z, t);
        r.w = to_u8(v.w, t);
        out4[i] = r;
    }
    if (blockIdx.x == 0) {
        const int64_t i = (n4 << 2) + threadIdx.x;
        if (i < n) out[i] = to_u8(in[i], t);
    }
}

// the same map with the transform read from HBM (written by the statistics chain, stats.hip): no host round trip between
// compute_image_stats -> auto_stf -> apply_stf (cmd/common.rs:18-22)
__global__ __launch_bounds__(256) void stf_u8_tx_kernel(const float *__restrict__ in, int64_t n, const StfTx *__restrict__ tx,
                                                        unsigned char *__restrict__ out) {
    const StfTx t = *tx;
    const int64_t n4 = n >> 2;
    const int64_t stride = (int64_t)gridDim.x * 256;
    const float4 *in4 = reinterpret_cast<const float4 *>(in);
    uchar4 *out4 = reinterpret_cast<uchar4 *>(out);
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) {
        const float4 v = in4[i];
        uchar4 r;
        r.x = to_u8(v.x, t);
        r.y = to_u8(v.y, t);
        r.z = to_u8(v.z, t);
        r.w = to_u8(v.w, t);
        out4[i] = r;
    }
    if (blockIdx.x == 0) {
        const int64_t i = (n4 << 2) + threadIdx.x;
        if (i < n) out[i] = to_u8(in[i], t);
    }
}

__global__ __launch_bounds__(256) void stf_f32_kernel(const float *in, int64_t n, StfTx t, float *out) {
    const int64_t n4 = n >> 2;
    const int64_t stride = (int64_t)gridDim.x * 256;
    const float4 *in4 = reinterpret_cast<const float4 *>(in);
    float4 *out4 = reinterpret_cast<float4 *>(out);
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) {
        const float4 v = in4[i];
        float4 r;
        r.x = to_f32(v.x, t);
        r.y = to_f32(v.y, t);
        r.z = to_f32(v.z, t);
        r.w = to_f32(v.w, t);
        out4[i] = r;
    }
    if (blockIdx.x == 0) {
        const int64_t i = (n4 << 2) + threadIdx.x;
        if (i < n) out[i] = to_f32(in[i], t);
    }
}

// the streaming ceiling probe behind bench.py's `measured_copy_GBs`: four 16-byte loads in flight per lane (a workgroup moves 16 KiB
// per iteration), streaming loads and stores (nothing of a 2 x 256 MB copy is worth keeping in L2)
typedef float copy_f4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void copy_kernel(const float4 *__restrict__ src4, float4 *__restrict__ dst4, int64_t n4) {
    const copy_f4 *src = reinterpret_cast<const copy_f4 *>(src4);
    copy_f4 *dst = reinterpret_cast<copy_f4 *>(dst4);
    const int64_t stride = (int64_t)gridDim.x * 1024;
    for (int64_t base = (int64_t)blockIdx.x * 1024; base < n4; base += stride) {
        const int64_t i = base + threadIdx.x;
        if (base + 1024 <= n4) {
            const copy_f4 a = __builtin_nontemporal_load(&src[i]), b = __builtin_nontemporal_load(&src[i + 256]),
                          c = __builtin_nontemporal_load(&src[i + 512]), d = __builtin_nontemporal_load(&src[i + 768]);
            __builtin_nontemporal_store(a, &dst[i]);
            __builtin_nontemporal_store(b, &dst[i + 256]);
            __builtin_nontemporal_store(c, &dst[i + 512]);
            __builtin_nontemporal_store(d, &dst[i + 768]);
        } else {
            for (int k = 0; k < 4; ++k)
                if (i + 256 * k < n4) dst[i + 256 * k] = src[i + 256 * k];
        }
    }
}

int stream_grid(ab_ctx *ctx, int64_t n4) {
    const int64_t want = (n4 + 255) / 256;
    const int64_t cap = (int64_t)(ctx->cu_count > 0 ? ctx->cu_count : 256) * 8;
    return (int)std::max<int64_t>(1, std::min(want, cap));
}

}  // namespace

int ab_stf_u8_device(ab_ctx *ctx, const float *in, int64_t n, const ab_stf_params *p, const ab_image_stats *st,
                     uint8_t *out) {
    AB_HIP(ctx, hipSetDevice(ctx->device));
    AB_CHECK(ctx, ((uintptr_t)in & 15) == 0 && ((uintptr_t)out & 3) == 0, "apply_stf: planes must be 16-byte aligned");
    hipLaunchKernelGGL(stf_u8_kernel, dim3(stream_grid(ctx, n >> 2)), dim3(256), 0, ctx->stream, in, n, make_tx(p, st), out);
    AB_HIP(ctx, hipGetLastError());
    return AB_OK;
}

int ab_stf_u8_device_tx(ab_ctx *ctx, const float *in, int64_t n, const void *tx_dev, uint8_t *out) {
    AB_HIP(ctx, hipSetDevice(ctx->device));
    AB_CHECK(ctx, ((uintptr_t)in & 15) == 0 && ((uintptr_t)out & 3) == 0, "apply_stf: planes must be 16-byte aligned");
    hipLaunchKernelGGL(stf_u8_tx_kernel, dim3(stream_grid(ctx, n >> 2)), dim3(256), 0, ctx->stream, in, n, (const StfTx *)tx_dev, out);
    AB_HIP(ctx, hipGetLastError());
    return AB_OK;
}

int ab_stf_f32_device(ab_ctx *ctx, const float *in, int64_t n, const ab_stf_params *p, const ab_image_stats *st,
                      float *out) {
    AB_HIP(ctx, hipSetDevice(ctx->device));
    AB_CHECK(ctx, ((uintptr_t)in & 15) == 0 && ((uintptr_t)out & 15) == 0, "apply_stf: planes must be 16-byte aligned");
    hipLaunchKernelGGL(stf_f32_kernel, dim3(stream_grid(ctx, n >> 2)), dim3(256), 0, ctx->stream, in, n, make_tx(p, st), out);
    AB_HIP(ctx, hipGetLastError());
    return AB_OK;
}

extern "C" {

// stf.rs:13-39 (host scalar maths)
int ab_auto_stf(const ab_image_stats *stats, const ab_auto_stf_config *cfg, ab_stf_params *out) try {
    if (!stats || !cfg || !out) return AB_ERR_INVALID;
    ab_auto_stf_hd(stats, cfg, out);
    return AB_OK;
} AB_CATCH_NOCTX

int ab_apply_stf_u8(ab_ctx *ctx, const ab_plane *img, const ab_stf_params *p, const ab_image_stats *st, uint8_t *out,
                    int32_t out_on_device) try {
    if (!ctx) return AB_ERR_INVALID;
    AB_CHECK(ctx, img && p && st && out, "null argument");
    StagedPlane in;
    AB_TRY(ab_stage_in(ctx, img, &in));
    const int64_t n = in.rows * in.cols;
    int rc = AB_OK;
    if (out_on_device) {
        rc = ab_stf_u8_device(ctx, in.dptr, n, p, st, out);
    } else {
        void *d = nullptr;
        hipError_t e = hipMalloc(&d, (size_t)n);
        if (e != hipSuccess) {
            rc = ab_set_error(ctx, AB_ERR_HIP, "hipMalloc: %s", hipGetErrorString(e));
        } else {
            rc = ab_stf_u8_device(ctx, in.dptr, n, p, st, (uint8_t *)d);
            if (rc == AB_OK) {
                e = hipMemcpyAsync(out, d, (size_t)n, hipMemcpyDeviceToHost, ctx->stream);
                if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
                if (e != hipSuccess) rc = ab_set_error(ctx, AB_ERR_HIP, "D2H: %s", hipGetErrorString(e));
            }
            (void)hipFree(d);
        }
    }
    ab_stage_release(ctx, &in);
    return rc;
} AB_CATCH(ctx)

int ab_apply_stf_f32(ab_ctx *ctx, const ab_plane *img, const ab_stf_params *p, const ab_image_stats *st,
                     ab_plane_mut *out) try {
    if (!ctx) return AB_ERR_INVALID;
    AB_CHECK(ctx, img && p && st && out, "null argument");
    AB_CHECK(ctx, img->rows == out->rows && img->cols == out->cols, "apply_stf_f32 keeps the image dims");
    StagedPlane in;
    AB_TRY(ab_stage_in(ctx, img, &in));
    StagedOut so;
    int rc = ab_stage_out_begin(ctx, out, &so);
    if (rc == AB_OK) {
        rc = ab_stf_f32_device(ctx, in.dptr, in.rows * in.cols, p, st, so.dptr);
        if (rc == AB_OK)
            rc = ab_stage_out_finish(ctx, &so);
        else
            ab_stage_out_abort(ctx, &so);
    }
    ab_stage_release(ctx, &in);
    return rc;
} AB_CATCH(ctx)

int ab_bench_copy(ab_ctx *ctx, const float *src_dev, float *dst_dev, size_t n_floats) try {
    if (!ctx) return AB_ERR_INVALID;
    AB_CHECK(ctx, src_dev && dst_dev && (n_floats % 4) == 0, "copy needs 16-byte multiples");
    AB_HIP(ctx, hipSetDevice(ctx->device));
    const int64_t n4 = (int64_t)(n_floats / 4);
    hipLaunchKernelGGL(copy_kernel, dim3(stream_grid(ctx, (n4 + 3) / 4)), dim3(256), 0, ctx->stream, (const float4 *)src_dev,
                       (float4 *)dst_dev, n4);
    AB_HIP(ctx, hipGetLastError());
    return AB_OK;
} AB_CATCH(ctx)

}  // extern "C"
