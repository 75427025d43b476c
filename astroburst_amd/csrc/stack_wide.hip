// Per-pixel kappa-sigma stacking of MORE than 64 frames (65 .. 512) on gfx950.
//
// stack_sigma_clip.hip keeps a pixel's samples in one lane's registers, which ends at 64.  Here one WAVE owns a pixel:
// lane l holds the samples of frames l, l + 64, ... (K = 2, 4 or 8 registers), the wave sorts its 64 K values with a
// bitonic network whose intra-lane stages use constant register indices and whose cross-lane stages are shuffles, and
// everything sigma_clip_combine (core/stacking/combine.rs:14-92) needs is then a rank lookup (v_readlane) or a
// wave-wide count.  The f64 sums run over the sorted survivors one by one in ascending order, exactly as the CPU
// restatement's ORC_ORDER_ASCENDING does, so the result is bit-identical to the <= 64-frame kernel's definition.
// ~2000 instructions per pixel and uncoalesced gathers (one lane per plane): tens of milliseconds for 4096^2 x 128,
// a fallback for deep stacks rather than a roofline kernel.
#include "ab_common.hpp"
#include "wave_sort.hpp"

#include <algorithm>
#include <cmath>

namespace {

constexpr double kMadToSigma = 1.4826;  // types/constants.rs:7
constexpr int kRejSlots = AB_REJ_SLOTS;

struct WideArgs {
    const float *const *p;  // n plane pointers (device array)
    const int64_t *ld;      // n row strides
    int n, contiguous;
    int64_t rows, cols;
    float sigma_low, sigma_high;
    uint32_t max_iter;
    float *out;       // full mode
    double *out_sum;  // partial mode
    uint32_t *out_cnt;
    unsigned long long *rejected;
    int median_only;  // median_combine_row_major (calibration.rs:84-125): [len/2] of the finite samples
};

// sum over the sorted ranks a..b, ascending, one f64 add per element (the oracle's order); mode 1: squared deviations
template <int K, int MODE>
__device__ __forceinline__ double ranked_sum(const float (&x)[K], int a, int b, double mean) {
    double s = 0.0;
    if (a > b) return s;
    for (int src = a >> Log2<K>::v; src <= (b >> Log2<K>::v); ++src) {
#pragma unroll
        for (int j = 0; j < K; ++j) {
            const int e = src * K + j;
            if (e < a || e > b) continue;  // wave-uniform
            const double v = (double)__builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, x[j]), src));
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

// The same sums as a TREE (round 6, the default engine of the > 512-frame stacks): every lane adds its own K sorted elements inside
// [a, b] in f64, then six butterfly steps.  The serial ascending chain above is what made a 1024-frame pixel cost ~32 000
// instructions (2 x 1024 dependent f64 adds per iteration); the tree is ~100.  It is the <= 64-frame fast engine's contract, not
// the oracle's order: f64 sums of <= 4096 f32 values in another order differ in their last bits, the f32 result almost never
// (tests: 1e-5 relative, at most 1e-4 of the pixels may differ at all, measured none); AB_STACK_EXACT=1 keeps the chain.
template <int K, int MODE>
__device__ __forceinline__ double ranked_sum_tree(const float (&x)[K], int a, int b, double mean, int lane) {
    double s = 0.0;
#pragma unroll
    for (int k = 0; k < K; ++k) {
        const int e = lane * K + k;
        const double v = (double)x[k];
        const double t = MODE == 0 ? v : (v - mean) * (v - mean);
        s += (e >= a && e <= b) ? t : 0.0;
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) s += __shfl_xor(s, off, 64);
    return s;
}
template <int K, int MODE, bool TREE>
__device__ __forceinline__ double ranked_sum_any(const float (&x)[K], int a, int b, double mean, int lane) {
    if constexpr (TREE) return ranked_sum_tree<K, MODE>(x, a, b, mean, lane);
    return ranked_sum<K, MODE>(x, a, b, mean);
}

// sigma_clip_combine of one pixel whose samples sit in x[] (non-finite ones already replaced by +inf, `fin` of this lane's are
// finite); returns the rejected-sample count and writes the outputs from lane 0
// `row` (nullable): 64 K floats of LDS that belong to this wave alone (the staged pixel's row, already consumed).  With it the MAD
// needs no second sort: among SORTED samples the deviations left and right of the median are two sorted runs, and the m-th smallest
// of their merge is min over p = 0 .. m of max(med - x[p], x[p + m] - med) (x[>= n] = +inf) -- stack_sigma_clip.hip's med_mad_at,
// evaluated here by the whole wave at once: the sorted samples go to the row, every lane reads its elements' partners p + m back.
template <int K, bool TREE = false>
__device__ __forceinline__ uint32_t wide_pixel(const WideArgs &a, int64_t g, float (&x)[K], int fin, int lane, float *row = nullptr) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) fin += __shfl_xor(fin, off, 64);
    const int n = __builtin_amdgcn_readfirstlane(fin);  // finite samples: sorted ranks [0, n); uniform by construction
    wave_sort<K>(x, lane);

    float value = 0.0f;
    double S = 0.0;
    int len = n;
    uint32_t rej = 0;
    if (n == 1) {
        value = elem<K>(x, 0);  // combine.rs:24-26
        S = (double)value;
    } else if (n >= 2) {
        const float med = elem<K>(x, n >> 1);  // combine.rs:38-40
        if (a.median_only) {
            value = med;
        } else {
            // MAD (combine.rs:42-46): the n/2-th smallest |v - med| -- sort the deviations the same way
            float mad;
            if (row) {
                const int m = n >> 1;
#pragma unroll
                for (int k = 0; k < K; ++k) row[lane * K + k] = x[k];
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                float best = __builtin_inff();
#pragma unroll
                for (int k = 0; k < K; ++k) {
                    const int p = lane * K + k, q = p + m;
                    const float xq = q < 64 * K ? row[q < 64 * K ? q : 0] : __builtin_inff();
                    const float term = fmaxf(med - x[k], xq - med);
                    best = p <= m ? fminf(best, term) : best;
                }
#pragma unroll
                for (int off = 32; off >= 1; off >>= 1) best = fminf(best, __shfl_xor(best, off, 64));
                mad = best;
            } else {
                float d[K];
#pragma unroll
                for (int k = 0; k < K; ++k) d[k] = (lane * K + k) < n ? fabsf(x[k] - med) : __builtin_inff();
                wave_sort<K>(d, lane);
                mad = elem<K>(d, n >> 1);
            }
            float sigma = (float)fmax((double)mad * kMadToSigma, 1e-10);
            float center = med, last_center = __builtin_nanf("");
            int lo_r = 0, hi_r = n - 1;  // survivors = sorted ranks lo_r..hi_r (`v - center` is monotone in v)
            for (uint32_t it = 0; it < a.max_iter; ++it) {
                if (len < 2) break;  // combine.rs:33-35
                if (it > 0) {        // mean / sample variance of the survivors, f64, ascending (combine.rs:50-60)
                    const double nn = (double)len;
                    const double mean = ranked_sum_any<K, 0, TREE>(x, lo_r, hi_r, 0.0, lane) / nn;
                    const double q = ranked_sum_any<K, 1, TREE>(x, lo_r, hi_r, mean, lane);
                    const double variance = q / fmax(nn - 1.0, 1.0);
                    center = (float)mean;
                    sigma = (float)fmax(sqrt(variance), 1e-10);
                }
                last_center = center;                 // combine.rs:63
                const float lo = -a.sigma_low * sigma;  // combine.rs:65-66
                const float hi = a.sigma_high * sigma;
                int cl = 0, ch = 0;
#pragma unroll
                for (int k = 0; k < K; ++k) {
                    const int e = lane * K + k;
                    const bool in = e >= lo_r && e <= hi_r;
                    const float dev = x[k] - center;
                    cl += (int)__builtin_popcountll(__ballot(in && !(dev >= lo)));
                    ch += (int)__builtin_popcountll(__ballot(in && !(dev <= hi)));
                }
                // a sample can fail both tests only if nothing survives (lo > hi or NaN thresholds)
                const int removed = (cl + ch > len) ? len : (cl + ch);
                rej += (uint32_t)removed;  // combine.rs:76-78
                len -= removed;
                if (len > 0) {
                    lo_r += cl;
                    hi_r -= ch;
                } else {
                    lo_r = 1;
                    hi_r = 0;
                }
                if (removed == 0) break;  // combine.rs:80-82
            }
            if (len == 0) {  // combine.rs:85-88
                value = __builtin_isfinite(last_center) ? last_center : 0.0f;
            } else {
                S = ranked_sum_any<K, 0, TREE>(x, lo_r, hi_r, 0.0, lane);  // combine.rs:90-91
                value = (float)(S / (double)len);
            }
        }
    }
    if (lane == 0) {
        if (a.out_sum) {
            a.out_sum[g] = len > 0 ? S : 0.0;
            a.out_cnt[g] = (uint32_t)(len > 0 ? len : 0);
        } else {
            a.out[g] = value;
        }
    }
    return rej;
}

template <int K>
__global__ __launch_bounds__(256) void stack_wide_kernel(const WideArgs a) {
    const int lane = threadIdx.x & 63;
    const int64_t total = a.rows * a.cols, nwaves = (int64_t)gridDim.x * 4;
    const unsigned int wave_id = blockIdx.x * 4u + (threadIdx.x >> 6);
    unsigned long long rej_total = 0;
    for (int64_t g = wave_id; g < total; g += nwaves) {  // wave-uniform pixel
        int64_t y = 0, xcol = g;
        if (!a.contiguous) {
            y = g / a.cols;
            xcol = g - y * a.cols;
        }
        // ---- gather (combine.rs:170-175): only finite samples take part; the rest are +inf pads on top of the order ----
        float x[K];
        int fin = 0;
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const int f = lane + 64 * k;
            float s = __builtin_inff();
            if (f < a.n) {
                const float v = a.p[f][a.contiguous ? g : (y * a.ld[f] + xcol)];
                if (__builtin_isfinite(v)) {
                    s = v;
                    ++fin;
                }
            }
            x[k] = s;
        }
        rej_total += wide_pixel<K>(a, g, x, fin, lane);
    }
    if (lane == 0 && rej_total) atomicAdd(&a.rejected[wave_id & (kRejSlots - 1)], rej_total);
}

// contiguous 16-byte aligned planes of 4 k pixels: a lane fetches FOUR consecutive pixels of its frames with one 16-byte
// load (a quarter of the load instructions, four times the use of every 64-byte sector) and the wave then combines the
// four pixels one after the other
template <int K>
__global__ __launch_bounds__(256) void stack_wide_quad_kernel(const WideArgs a) {
    const int lane = threadIdx.x & 63;
    const int64_t quads = (a.rows * a.cols) >> 2, nwaves = (int64_t)gridDim.x * 4;
    const unsigned int wave_id = blockIdx.x * 4u + (threadIdx.x >> 6);
    unsigned long long rej_total = 0;
    for (int64_t q = wave_id; q < quads; q += nwaves) {
        float4 v[K];
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const int f = lane + 64 * k;
            v[k] = make_float4(__builtin_inff(), __builtin_inff(), __builtin_inff(), __builtin_inff());
            if (f < a.n) v[k] = reinterpret_cast<const float4 *>(a.p[f])[q];
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float x[K];
            int fin = 0;
#pragma unroll
            for (int k = 0; k < K; ++k) {
                const float s = j == 0 ? v[k].x : (j == 1 ? v[k].y : (j == 2 ? v[k].z : v[k].w));
                const bool ok = __builtin_isfinite(s);  // (the +inf of an absent frame is not finite either)
                x[k] = ok ? s : __builtin_inff();
                fin += ok ? 1 : 0;
            }
            rej_total += wide_pixel<K>(a, q * 4 + j, x, fin, lane);
        }
    }
    if (lane == 0 && rej_total) atomicAdd(&a.rejected[wave_id & (kRejSlots - 1)], rej_total);
}

// 513 .. 4096 contiguous frames (round 6): a pixel's 4 bytes are all a wave of stack_wide_kernel uses of every
// 64-byte line it touches -- 16 x the algorithmic traffic, and that, not the ~3000 instructions per pixel, made 1024 x 2048^2 take
// 242 ms (0.07 TB/s of samples).  Here a workgroup stages G = 16 / 8 / 4 ADJACENT pixels of all n frames through LDS (K = 16 / 32 / 64:
// 64 KB) with 16-byte loads -- every byte fetched is used -- and its four waves then combine the pixels one after the other with
// the same wide_pixel (bit-identical results).  Needs contiguous 16-byte aligned planes and a pixel count that is a multiple of 16.
// 512 threads: two workgroups of 64 KB per CU are then four waves per SIMD.  K = 64 (2049 .. 4096 frames, four pixels per group):
// FOUR waves, one per pixel -- with eight, a thread may hold 256 registers and the two 64-register sort arrays spilled 279 of them
// (2100 x 1024^2: 590 ms, twenty times the per-pixel time of 2048 frames).
template <int K>
constexpr int tile_waves() { return K == 64 ? 4 : 8; }
template <int K, bool TREE>
__global__ __launch_bounds__(64 * tile_waves<K>()) void stack_wide_tile_kernel(const WideArgs a) {
    constexpr int kTileWaves = tile_waves<K>();
    constexpr int G = 256 / K, NPAD = 64 * K;  // pixels per group (G x NPAD floats = 64 KB), frames padded to the wave's 64 K slots
    static_assert(G >= 4 && G % 4 == 0 && G * NPAD * 4 == 65536, "32 / 16 / 8 / 4 adjacent pixels in 64 KB");
    extern __shared__ __attribute__((aligned(16))) float tile[];  // [G][NPAD]
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
    const int64_t groups = (a.rows * a.cols) / G;
    for (int i = t; i < G * (NPAD - a.n); i += 64 * kTileWaves) {  // the slots past the last frame: +inf, once
        const int row = i / (NPAD - a.n), col = a.n + i % (NPAD - a.n);
        tile[row * NPAD + col] = __builtin_inff();
    }
    unsigned long long rej_total = 0;
    // A group is 64 / 32 / 16 bytes of every plane: a half, a quarter, an eighth of a cache line, the rest being the next groups'.
    // Workgroup ids go round the eight XCDs (id % 8): in id order the groups that share a line would pull it into as many L2s.
    // Ids id, id + 8, ... -- one XCD -- take neighbouring groups instead (any grid that is a multiple of 8; else id order).
    const unsigned int first = (gridDim.x & 7u) == 0 ? (blockIdx.x & 7u) * (gridDim.x >> 3) + (blockIdx.x >> 3) : blockIdx.x;
    for (int64_t grp = first; grp < groups; grp += gridDim.x) {
        const int64_t base = grp * G;
        constexpr int QPF = G / 4;  // 16-byte quads per frame and group
        for (int i0 = 0; i0 < a.n * QPF; i0 += 64 * kTileWaves * 4) {
            float4 v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int i = i0 + 64 * kTileWaves * u + t;
                if (i < a.n * QPF) v[u] = *reinterpret_cast<const float4 *>(a.p[i / QPF] + base + 4 * (i % QPF));
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int i = i0 + 64 * kTileWaves * u + t;
                if (i < a.n * QPF) {
                    const int f = i / QPF, q = i % QPF;
                    tile[(4 * q + 0) * NPAD + f] = v[u].x;
                    tile[(4 * q + 1) * NPAD + f] = v[u].y;
                    tile[(4 * q + 2) * NPAD + f] = v[u].z;
                    tile[(4 * q + 3) * NPAD + f] = v[u].w;
                }
            }
        }
        __syncthreads();
        for (int j = wv; j < G; j += kTileWaves) {
            float x[K];
            int fin = 0;
#pragma unroll
            for (int k = 0; k < K; ++k) {
                const float sv = tile[j * NPAD + lane + 64 * k];
                const bool ok = __builtin_isfinite(sv);  // (the +inf of an absent frame is not finite either)
                x[k] = ok ? sv : __builtin_inff();
                fin += ok ? 1 : 0;
            }
            rej_total += wide_pixel<K, TREE>(a, base + j, x, fin, lane, tile + j * NPAD);
        }
        __syncthreads();
    }
    if (lane == 0 && rej_total) atomicAdd(&a.rejected[(blockIdx.x * (unsigned int)kTileWaves + (unsigned int)wv) & (kRejSlots - 1)], rej_total);
}

// The pixels a multi-lane fast pass (stack_quad.hip, 513 .. 1024 frames) handed over: `slots` lists of up to `cap` pixel indices, one
// workgroup per list, a wave per pixel, the same wide_pixel (the oracle's ascending sums).  Leaves its list empty.
template <int K>
__global__ __launch_bounds__(256) void stack_wide_list_kernel(const WideArgs a, unsigned int *__restrict__ list_count, const int *__restrict__ list,
                                                              unsigned int cap) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const unsigned int cnt = list_count[blockIdx.x];
    const int *mine = list + (size_t)blockIdx.x * cap;
    unsigned long long rej_total = 0;
    for (unsigned int i = (unsigned int)wv; i < cnt; i += 4) {  // wave-uniform pixel
        const int64_t g = (int64_t)mine[i];
        float x[K];
        int fin = 0;
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const int f = lane + 64 * k;
            float s = __builtin_inff();
            if (f < a.n) {
                const float v = a.p[f][g];
                if (__builtin_isfinite(v)) {
                    s = v;
                    ++fin;
                }
            }
            x[k] = s;
        }
        rej_total += wide_pixel<K>(a, g, x, fin, lane);
    }
    if (lane == 0 && rej_total) atomicAdd(&a.rejected[(blockIdx.x * 4u + (unsigned int)wv) & (kRejSlots - 1)], rej_total);
    __syncthreads();  // (every wave has read the count)
    if (threadIdx.x == 0) list_count[blockIdx.x] = 0;
}

}  // namespace

// dplanes / ld are HOST arrays of n entries (64 < n <= 512); counters already cleared by the caller
int ab_stack_wide_device(ab_ctx *ctx, const float *const *dplanes, const int64_t *ld, size_t n, int64_t rows, int64_t cols,
                         const ab_stack_config *cfg, float *out_dev, double *out_sum_dev, uint32_t *out_cnt_dev, bool median_only) {
    AB_CHECK(ctx, n > 64 && n <= 4096, "the wave-per-pixel stack takes 65 .. 4096 frames (got %zu)", n);
    void *ws = nullptr;
    AB_TRY(ab_workspace(ctx, AB_WS_STACK_WIDE, n * (sizeof(float *) + sizeof(int64_t)), &ws));
    // the tables are tiny; a blocking copy keeps the host arrays' lifetime out of the picture
    AB_HIP(ctx, hipStreamSynchronize(ctx->stream));
    AB_HIP(ctx, hipMemcpy(ws, dplanes, n * sizeof(float *), hipMemcpyHostToDevice));
    AB_HIP(ctx, hipMemcpy((char *)ws + n * sizeof(float *), ld, n * sizeof(int64_t), hipMemcpyHostToDevice));
    WideArgs a;
    a.p = (const float *const *)ws;
    a.ld = (const int64_t *)((char *)ws + n * sizeof(float *));
    a.n = (int)n;
    a.contiguous = 1;
    for (size_t i = 0; i < n; ++i)
        if (ld[i] != cols) a.contiguous = 0;
    a.rows = rows;
    a.cols = cols;
    a.sigma_low = cfg->sigma_low;
    a.sigma_high = cfg->sigma_high;
    a.max_iter = cfg->max_iterations;
    a.out = out_dev;
    a.out_sum = out_sum_dev;
    a.out_cnt = out_cnt_dev;
    a.rejected = ctx->counters;
    a.median_only = median_only ? 1 : 0;
    const int64_t total = rows * cols;
    const int grid = (int)std::max<int64_t>(1, std::min<int64_t>((total + 3) / 4, (int64_t)(ctx->cu_count > 0 ? ctx->cu_count : 256) * 8));
    bool quad = a.contiguous && (total & 3) == 0;
    for (size_t i = 0; i < n && quad; ++i) quad = ((uintptr_t)dplanes[i] & 15) == 0;
    // 513 .. 4096 frames (K = 16, 32, 64 registers per lane and array): the same wave-per-pixel definition, one pixel at a time
    // (four pixels' worth of 16-byte loads would not fit the register file beside two sort arrays)
    // (the LDS-staged form: contiguous 16-byte aligned planes, whole groups of 16 pixels; two workgroups of 64 KB per CU)
    const int64_t gpx = n > 2048 ? 4 : (n > 1024 ? 8 : (n > 512 ? 16 : 32));  // pixels per group of stack_wide_tile_kernel<64 / 32 / 16 / 8>
    const bool tiled = quad && total % gpx == 0 && n > 256;  // (257 .. 512 frames reach this file only where stack_pair.hip does not take them)
    const int tgrid = (int)std::max<int64_t>(1, std::min<int64_t>(total / 4, (int64_t)(ctx->cu_count > 0 ? ctx->cu_count : 256) * 2));
    // the tree sums: the default engine here; the partial sums of the sharded estimator and AB_STACK_EXACT=1 keep the ascending chain
    const bool tree = !ctx->stack_exact && !out_sum_dev;
#define AB_WIDE_TILE(KK, TT)                                                                                                                   \
    do {                                                                                                                                       \
        AB_HIP(ctx, hipFuncSetAttribute((const void *)stack_wide_tile_kernel<KK, TT>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536));  \
        hipLaunchKernelGGL((stack_wide_tile_kernel<KK, TT>), dim3(tgrid), dim3(64 * tile_waves<KK>()), 65536, ctx->stream, a);                                 \
    } while (0)
    if (tiled && n > 2048) {
        if (tree) AB_WIDE_TILE(64, true); else AB_WIDE_TILE(64, false);
    } else if (tiled && n > 1024) {
        if (tree) AB_WIDE_TILE(32, true); else AB_WIDE_TILE(32, false);
    } else if (tiled && n > 512) {
        if (tree) AB_WIDE_TILE(16, true); else AB_WIDE_TILE(16, false);
    } else if (tiled) {
        if (tree) AB_WIDE_TILE(8, true); else AB_WIDE_TILE(8, false);
#undef AB_WIDE_TILE
    } else if (n > 2048) {
        hipLaunchKernelGGL(stack_wide_kernel<64>, dim3(grid), dim3(256), 0, ctx->stream, a);
    } else if (n > 1024) {
        hipLaunchKernelGGL(stack_wide_kernel<32>, dim3(grid), dim3(256), 0, ctx->stream, a);
    } else if (n > 512) {
        hipLaunchKernelGGL(stack_wide_kernel<16>, dim3(grid), dim3(256), 0, ctx->stream, a);
    } else if (quad) {
        if (n <= 128)
            hipLaunchKernelGGL(stack_wide_quad_kernel<2>, dim3(grid), dim3(256), 0, ctx->stream, a);
        else if (n <= 256)
            hipLaunchKernelGGL(stack_wide_quad_kernel<4>, dim3(grid), dim3(256), 0, ctx->stream, a);
        else
            hipLaunchKernelGGL(stack_wide_quad_kernel<8>, dim3(grid), dim3(256), 0, ctx->stream, a);
    } else if (n <= 128) {
        hipLaunchKernelGGL(stack_wide_kernel<2>, dim3(grid), dim3(256), 0, ctx->stream, a);
    } else if (n <= 256) {
        hipLaunchKernelGGL(stack_wide_kernel<4>, dim3(grid), dim3(256), 0, ctx->stream, a);
    } else {
        hipLaunchKernelGGL(stack_wide_kernel<8>, dim3(grid), dim3(256), 0, ctx->stream, a);
    }
    AB_HIP(ctx, hipGetLastError());
    return AB_OK;
}


// table_dev: DEVICE array of at least n plane pointers (contiguous planes of rows x cols); 2048 lists (stack_pair.hpp: kListSlots)
int ab_stack_wide_list_device(ab_ctx *ctx, const float *const *table_dev, size_t n, int64_t rows, int64_t cols, const ab_stack_config *cfg, float *out_dev,
                              bool median_only, unsigned int *list_count, const int *list, unsigned int cap) {
    AB_CHECK(ctx, n > 512 && n <= 1024, "internal: the list pass of the wave-per-pixel kernel takes 513 .. 1024 frames (got %zu)", n);
    WideArgs a;
    a.p = table_dev;
    a.ld = nullptr;
    a.n = (int)n;
    a.contiguous = 1;
    a.rows = rows;
    a.cols = cols;
    a.sigma_low = cfg->sigma_low;
    a.sigma_high = cfg->sigma_high;
    a.max_iter = cfg->max_iterations;
    a.out = out_dev;
    a.out_sum = nullptr;
    a.out_cnt = nullptr;
    a.rejected = ctx->counters;
    a.median_only = median_only ? 1 : 0;
    hipLaunchKernelGGL(stack_wide_list_kernel<16>, dim3(2048), dim3(256), 0, ctx->stream, a, list_count, list, cap);
    AB_HIP(ctx, hipGetLastError());
    return AB_OK;
}
