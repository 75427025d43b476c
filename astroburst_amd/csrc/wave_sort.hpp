// Sorting 64 K values held K per lane across one wavefront (gfx950), and rank lookups in the result.  Shared by the deep
// stack (stack_wide.hip) and the deep batch stack (batch_pipeline.hip).
#pragma once
#include <hip/hip_runtime.h>

namespace {

template <int K>
struct Log2;
template <>
struct Log2<2> {
    static constexpr int v = 1;
};
template <>
struct Log2<4> {
    static constexpr int v = 2;
};
template <>
struct Log2<8> {
    static constexpr int v = 3;
};
template <>
struct Log2<16> {
    static constexpr int v = 4;
};
template <>
struct Log2<32> {
    static constexpr int v = 5;
};
template <>
struct Log2<64> {
    static constexpr int v = 6;
};

// ascending bitonic sort of the wave's 64 K values; element index e = lane * K + k
template <int K>
__device__ __forceinline__ void wave_sort(float (&x)[K], int lane) {
    constexpr int N = 64 * K;
#pragma unroll
    for (int size = 2; size <= N; size <<= 1) {
#pragma unroll
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            if (stride < K) {  // both elements in this lane: constant register indices
#pragma unroll
                for (int k = 0; k < K; ++k) {
                    if ((k & stride) == 0) {
                        const bool up = ((lane * K + k) & size) == 0;
                        const float lo = fminf(x[k], x[k | stride]), hi = fmaxf(x[k], x[k | stride]);
                        x[k] = up ? lo : hi;
                        x[k | stride] = up ? hi : lo;
                    }
                }
            } else {  // partner in lane ^ (stride / K), same register
                const int ls = stride / K;
                const bool lower = (lane & ls) == 0;
#pragma unroll
                for (int k = 0; k < K; ++k) {
                    const float y = __shfl_xor(x[k], ls, 64);
                    const bool up = ((lane * K + k) & size) == 0;
                    x[k] = (up == lower) ? fminf(x[k], y) : fmaxf(x[k], y);
                }
            }
        }
    }
}

// sorted element of wave-uniform rank r
template <int K>
__device__ __forceinline__ float elem(const float (&x)[K], int r) {
    const int src = r >> Log2<K>::v, k = r & (K - 1);
    float v = 0.0f;
#pragma unroll
    for (int j = 0; j < K; ++j)
        if (k == j) v = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, x[j]), src));
    return v;
}

}  // namespace
