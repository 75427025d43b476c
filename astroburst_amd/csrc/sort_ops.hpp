// The operations of the rewritten sorting networks (SortNet<NP>::sort_fused, tools/gen_sortnet.py), shared by the stacking kernels
// (stack_sigma_clip.hip, batch_pipeline.hip).  Include BEFORE sortnet_gen.hpp.
#pragma once
#include <hip/hip_runtime.h>

// 789 instructions for 64 samples instead of Batcher's 1038.  The operations are spelled as inline assembly: fminf() on a freshly loaded sample is preceded by a canonicalising v_max_f32 x, x, x,
// a med3 with a literal infinity is folded back to fminf(), and a chain of two fminf() is only sometimes selected as v_min3_f32.
// No NaN reaches the network (non-finite samples are replaced by +inf above it) and denormals are not flushed, so each of these
// returns one of its inputs bit for bit.
__device__ __forceinline__ float ab_v_min(float a, float b) {
    float r;
    asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ float ab_v_max(float a, float b) {
    float r;
    asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ float ab_v_min3(float a, float b, float c) {
    float r;
    asm("v_min3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
__device__ __forceinline__ float ab_v_max3(float a, float b, float c) {
    float r;
    asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
__device__ __forceinline__ float ab_v_med3(float a, float b, float c) {
    float r;
    asm("v_med3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
#define AB_SN_MIN2(a, b) ab_v_min(a, b)
#define AB_SN_MAX2(a, b) ab_v_max(a, b)
#define AB_SN_MIN3(a, b, c) ab_v_min3(a, b, c)
#define AB_SN_MAX3(a, b, c) ab_v_max3(a, b, c)
#define AB_SN_MED3(a, b, c) ab_v_med3(a, b, c)
#ifdef AB_STACK_CE_XOR  // a plain exchange (both outputs kept): the maximum is x ^ y ^ min, one full-rate v_bitop3_b32
#define AB_SN_CE(lo, hi, x, y)                                                                                          \
    {                                                                                                                   \
        lo = ab_v_min(x, y);                                                                                            \
        hi = __uint_as_float(__builtin_amdgcn_bitop3_b32(__float_as_uint(x), __float_as_uint(y), __float_as_uint(lo), 0x96)); \
    }
#endif
