// Shared by the two-lanes-per-pixel stacking kernels (stack_pair.hip: the oracle's arithmetic word for word; stack_duo.hip: the fast
// engine's running sums): H samples per lane, lanes 2k and 2k + 1 share pixel k of the wave's 32.
#pragma once
#include "ab_common.hpp"

#include "sort_ops.hpp"
#define AB_CE(a, b)                       \
    {                                     \
        T lo_ = ab_v_min(v[a], v[b]);     \
        T hi_ = ab_v_max(v[a], v[b]);     \
        v[a] = lo_;                       \
        v[b] = hi_;                       \
    }
#define AB_SORT4(a, b, c, d)                                                                          \
    {                                                                                                 \
        const T x0_ = v[a], x1_ = v[b], x2_ = v[c], x3_ = v[d];                                       \
        const T s0_ = ab_v_min3(x0_, x1_, x2_), s1_ = ab_v_med3(x0_, x1_, x2_), s2_ = ab_v_max3(x0_, x1_, x2_); \
        v[a] = ab_v_min(s0_, x3_);                                                                    \
        v[b] = ab_v_med3(s0_, s1_, x3_);                                                              \
        v[c] = ab_v_med3(s1_, s2_, x3_);                                                              \
        v[d] = ab_v_max(s2_, x3_);                                                                    \
    }
#include "sortnet_gen.hpp"

namespace abpair {

constexpr double kMadToSigma = 1.4826;  // types/constants.rs:7
constexpr int kRejSlots = AB_REJ_SLOTS;
// pixels the fast kernel hands to the oracle-arithmetic kernel: lists by wave index (one same-address atomic per deferring wave
// would serialise at ~12 ns each: stack_sigma_clip.hip), kListWaves one-wave workgroups walking each list
constexpr int kListSlots = 2048;
constexpr int kListWaves = 2;

struct PairArgs {
    const float *const *p;  // plane pointers (device array): the even lane's slot f is p[f], the odd lane's p[half + f]
    int n;                  // frames
    int half;               // frames per lane (the table's stride): H for the exact kernels, the frame-count class R <= H for the fast ones
    int64_t total;          // pixels
    float sigma_low, sigma_high;
    uint32_t max_iter;
    float *out;
    unsigned long long *rejected;
    int median_only;  // median_combine_row_major (calibration.rs:84-125): [len/2] of the finite samples
    // the fast kernel appends to these; the exact kernel in LIST mode (list != nullptr in its copy of the arguments) walks them
    unsigned int *list_count, *list_ticket;
    int *list;
    unsigned int list_cap;
    int walk_lists;
};

// the partner lane's value (lanes 2k <-> 2k+1): DPP quad_perm [1,0,3,2]
__device__ __forceinline__ float swapf(float x) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0xB1, 0xf, 0xf, false));
}
__device__ __forceinline__ int swapi(int x) { return __builtin_amdgcn_update_dpp(0, x, 0xB1, 0xf, 0xf, false); }
__device__ __forceinline__ double swapd(double x) {
    const unsigned long long u = __builtin_bit_cast(unsigned long long, x);
    const unsigned int lo = (unsigned int)swapi((int)(unsigned int)u), hi = (unsigned int)swapi((int)(unsigned int)(u >> 32));
    return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
}

// The network's min / max are inline assembly, and the compiler's hazard recogniser does not see a VALU write inside an asm
// statement: a DPP read of such a register within two wait states returns the OLD value.  Every sample therefore passes through
// one of these (volatile, `s_nop 1` inside, the sample an in/out operand) between the network that wrote it and the first DPP
// that reads it.
template <int H>
__device__ __forceinline__ void dpp_fence(float (&v)[H]) {
#pragma unroll
    for (int i = 0; i < H; i += 8)
        asm volatile("s_nop 1" : "+v"(v[i]), "+v"(v[i + 1]), "+v"(v[i + 2]), "+v"(v[i + 3]), "+v"(v[i + 4]), "+v"(v[i + 5]), "+v"(v[i + 6]),
                     "+v"(v[i + 7]));
}

// Compiler fence (no instructions): the sample vector looks rewritten, so LLVM does not hoist H f32->f64 conversions out of
// the clipping loop (stack_sigma_clip.hip: launder)
template <int H>
__device__ __forceinline__ void launder(float (&v)[H]) {
#pragma unroll
    for (int i = 0; i < H; i += 8)
        asm volatile("" : "+v"(v[i]), "+v"(v[i + 1]), "+v"(v[i + 2]), "+v"(v[i + 3]), "+v"(v[i + 4]), "+v"(v[i + 5]), "+v"(v[i + 6]),
                     "+v"(v[i + 7]));
}
__device__ __forceinline__ void opaque(int &a, int &b) { asm volatile("" : "+v"(a), "+v"(b)); }

// in-lane bitonic merge of a bitonic sequence of H (ascending result): log2 H half-cleaner stages.  (One function per stage: as
// two nested `#pragma unroll` loops the body exceeds the pragma's size limit, the outer loop stays a loop, and the samples live
// in scratch memory.)
template <int H, int D>
__device__ __forceinline__ void half_cleaner(float (&v)[H]) {
    using T = float;
#pragma unroll
    for (int i = 0; i < H; ++i)
        if ((i & D) == 0) AB_CE(i, i + D)
}
template <int H>
__device__ __forceinline__ void bitonic_merge(float (&v)[H]) {
    static_assert(H == 128 || H == 256, "two lanes per pixel: 128 or 256 samples each");
    if constexpr (H == 256) half_cleaner<H, 128>(v);
    half_cleaner<H, 64>(v);
    half_cleaner<H, 32>(v);
    half_cleaner<H, 16>(v);
    half_cleaner<H, 8>(v);
    half_cleaner<H, 4>(v);
    half_cleaner<H, 2>(v);
    half_cleaner<H, 1>(v);
}

// One cross step -- v[i] against the partner's v[H - 1 - i] -- leaves the H smallest of the pair's 2H sorted samples in the even
// lane and the H largest in the odd lane, each a bitonic sequence.
// (An exchange keeps v_med3(x, partner, -inf) = the minimum in the even lane, v_med3(x, partner, +inf) = the maximum in the odd
// lane: one instruction by a per-lane constant instead of min, max and a select.)
template <int H>
__device__ __forceinline__ void cross_step(float (&v)[H], bool odd) {
    const float sel = odd ? __builtin_inff() : -__builtin_inff();
#pragma unroll
    for (int i = 0; i < H / 2; ++i) {
        const float t1 = swapf(v[H - 1 - i]), t2 = swapf(v[i]);
        const float a = ab_v_med3(v[i], t1, sel), b = ab_v_med3(v[H - 1 - i], t2, sel);
        v[i] = a;
        v[H - 1 - i] = b;
    }
}

}  // namespace abpair

// stack_duo.hip: the fast pass of a 129 .. 512-frame stack (H = 128 or 256 samples per lane, the class R of frames per lane);
// the arguments' table holds 2 R pointers
int ab_stack_duo_launch(ab_ctx *ctx, int H, int R, const abpair::PairArgs &args);
// stack_quad.hip: the fast pass of a 257 .. 512-frame (L = 4 lanes per pixel) or 513 .. 1024-frame (L = 8) stack, 128 samples per lane
// (class R of frames per lane); the table holds L R pointers
int ab_stack_quad_launch(ab_ctx *ctx, int L, int R, const abpair::PairArgs &args);
