// One 256-thread workgroup sorting / summing a pixel's samples that live in GLOBAL memory (a per-workgroup scratch segment):
// the any-N fallback behind the deep stacks (stack_deep.hip, batch_pipeline.hip) once a pixel no longer fits one wave's
// registers (> 4096 frames, i.e. frames of at most a few megapixels: 4097 x 4096^2 x 4 B would not fit the HBM).  Nothing
// here is tuned: every compare-exchange stage is a round trip to L2 and a barrier.  It exists so that the library takes
// whatever frame count the reference takes (combine.rs:94-193 gathers a Vec per pixel: no limit).
#pragma once
#include <hip/hip_runtime.h>

namespace {

// ascending bitonic sort of np2 (a power of two) floats, no NaNs among them; every thread of the workgroup calls it
__device__ inline void block_bitonic_sort(float *x, int np2) {
    const int tid = threadIdx.x, nt = blockDim.x, half = np2 >> 1;
    for (int size = 2; size <= np2; size <<= 1) {
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            __syncthreads();
            for (int i = tid; i < half; i += nt) {
                const int lo = ((i & ~(stride - 1)) << 1) | (i & (stride - 1));  // bit `stride` clear
                const int hi = lo | stride;
                const bool up = (lo & size) == 0;
                const float a = x[lo], b = x[hi];
                const float mn = fminf(a, b), mx = fmaxf(a, b);
                x[lo] = up ? mn : mx;
                x[hi] = up ? mx : mn;
            }
        }
    }
    __syncthreads();
}

// f64 sum over x[a .. b] (inclusive), ONE addition per element in index order -- the order the oracle's ORC_ORDER_ASCENDING sums
// in when x is sorted, and frame order when it is not.  MODE 0: the values; MODE 1: squared deviations from `mean`.
// Called by every lane of ONE wave (64 elements are fetched side by side, the additions are serial): returns the same value in all.
template <int MODE>
__device__ inline double wave_serial_sum_f64(const float *x, int a, int b, double mean, int lane) {
    double s = 0.0;
    for (int base = a; base <= b; base += 64) {
        const int e = base + lane;
        const float mine = e <= b ? x[e] : 0.0f;
        const int cnt = min(64, b - base + 1);
        for (int l = 0; l < cnt; ++l) {
            const double v = (double)__builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, mine), l));
            if (MODE == 0) {
                s += v;
            } else {
                const double d = v - mean;
                s += d * d;
            }
        }
    }
    return s;
}

// the same in f32 (the batch stack's frame-order mean, calibration_pipeline.rs:369-376)
__device__ inline float wave_serial_sum_f32(const float *x, int a, int b, int lane) {
    float s = 0.0f;
    for (int base = a; base <= b; base += 64) {
        const int e = base + lane;
        const float mine = e <= b ? x[e] : 0.0f;
        const int cnt = min(64, b - base + 1);
        for (int l = 0; l < cnt; ++l) s += __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, mine), l));
    }
    return s;
}

}  // namespace
