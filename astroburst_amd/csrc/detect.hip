// Star detection / segmentation on gfx950.
//
// Replaces core/analysis/star_detection.rs: estimate_background (:32-84, with
// math/sigma_clip.rs:4-34 and math/median.rs:27-73 per tile), detect_stars (:86-258),
// core/analysis/confidence.rs:3-8, and core/alignment/affine.rs:24-53 (normalize_for_detection).
//
// Mapping (integer / index work, HBM- and latency-bound; no GEMM shapes):
//   * tile background: one 1024-thread workgroup per tile (<= 256 x 256 px), loaded once and kept on chip
//     (LDS + registers).  The reference's median / MAD selects become 11/11/10-bit radix selects over the
//     tile's valid pixels, whose f32 bit patterns are monotone (valid means > 1e-7).  Even-count medians
//     average the two middle order statistics exactly as exact_median_mut / median_f32_mut do, so tile
//     medians and sigmas are bit-identical (block_select.hpp has the cost model).
//   * labelling: the reference's sequential raster scan + 8-connected BFS is replaced by a lock-free
//     union-find over the above-threshold pixels (4 forward neighbours per pixel, atomicMin hooking, then
//     path flattening).  The root of a component is its minimum raster index, which makes the label
//     canonical; a component is reported only if it owns an INTERIOR pixel, because the reference seeds
//     from 1..rows-1 x 1..cols-1 only.  Only the thresholding pass touches every pixel: it appends the
//     labelled ones (a fraction of a percent of the frame) to a list that the later kernels iterate.
//   * per-component size / bounding box / first interior pixel are integer atomics (order-independent);
//     the flux-weighted moments are f64 sums over at most 5000 pixels, taken by one wave per component
//     over its bounding box with a fixed-shape lane reduction (reproducible; the reference sums in BFS
//     order: the two agree to ~1e-15 relative; counts, npix and the component set are exact).  The host
//     finishes O(#components) work: discovery order, star parameters, flux sort, 3 px dedup.
//   * a percentile normalisation (affine.rs:24-53) can ride along: consumers apply it per pixel on load
//     (ab_px), so the registration path never materialises the normalised frame.
#include "ab_common.hpp"
#include "block_select.hpp"
#include "tile_bucket.hpp"
#include "tile_stream.hpp"

#include <algorithm>
#include <cfloat>
#include <cmath>
#include <chrono>
#include <condition_variable>
#include <thread>

namespace {

constexpr double kMadToSigma = 1.4826;

// Per-frame parameters that one kernel of the detection chain produces and the next consumes, kept in device memory so that the
// host need not sit between them: the percentile kernel writes `xf`, the tile kernel reads it, bg_threshold_kernel reduces the
// tiles to (bg_median, bg_sigma, threshold), the labelling and moment kernels read those.  A kernel given a null FrameDev uses
// its by-value arguments instead (the stand-alone entry points, and the default registration path: see `chained` in
// ab_detect_stars_device for why the host still joins after the percentiles and after the tiles).
struct FrameDev {
    ab_pixel_xf xf;
    double bg_median, bg_sigma, threshold;
    unsigned int finite;  // finite subsample values (< 100: xf.on = 0)
    unsigned int pad;
};

// the candidate lists a tile launch leaves for the labelling pass (tile_stream.hpp: CandSink), planes x tiles of the LAUNCH:
// ent[(plane * tiles + tile) * kCandCap ..], cnt / cut[plane * tiles + tile].  ent == nullptr: no lists.
constexpr int kCandCap = 2048;  // entries per 256 x 256 tile (~1000 on a sky tile at 2.5 sigma: the Gaussian tail + the stars' pixels)
struct TileCand {
    uint2 *ent = nullptr;
    unsigned int *cnt = nullptr;
    float *cut = nullptr;
};

// ---- per-tile sigma-clipped statistics ---------------------------------------------------------
struct TileOut {
    double median, sigma;
    int valid;  // 1 if the tile had >= 8 valid pixels
    int pad;
};

// estimate_background's per-tile body (star_detection.rs:47-68) = sigma_clipped_stats(vals, 3.0, 2)
// (math/sigma_clip.rs:4-34); order statistics by workgroup radix select (block_select.hpp)
// Every select knows where its rank will be: all sky pixels share their top 11 bits, the median barely moves between
// clipping iterations and the MAD shrinks by a percent.  So a select's first sweep histograms LEVEL 1 of that top-level
// bin directly and only counts the candidates in lower / higher bins (prepare_lean); when the wanted ranks do lie in the
// bin (lean_holds: always, on sky tiles) two sweeps finish the select instead of three.  Otherwise the ordinary
// three-level select runs; either way the result is exact.
template <class S>
__device__ __forceinline__ TileOut tile_stats(const S &src, absel::Window &t, unsigned int *hist0, unsigned int *hist, unsigned int *tally,
                                              bool have_guess, uint32_t first_guess) {
    const absel::Keying by_value = {0, 0.0, 0.0f};
    // count + (speculated) histograms of keying k; spec is reset when the ordinary path had to be taken
    auto prepare_for = [&](const absel::Keying &k, bool guess_ok, uint32_t guess, absel::Spec *spec) -> unsigned int {
        *spec = absel::Spec();
        if (guess_ok) {
            const unsigned int n = absel::prepare_lean(src, t, k, hist, guess, spec, tally);
            if (absel::lean_holds(*spec, n)) return n;
            *spec = absel::Spec();
        }
        return absel::prepare(src, t, k, hist0);
    };
    absel::Spec spec_v;
    unsigned int n = prepare_for(by_value, have_guess, first_guess, &spec_v);
    TileOut res = {0.0, 1.0, 0, 0};
    if (n >= 8) {
        res.valid = 1;
        double median = 0.0, sigma = 1.0;
        float prev_mad = 0.0f;
        for (int it = 0; it < 3; ++it) {  // 2 clipping iterations + the final statistics (sigma_clip.rs:7-33)
            if (it < 2 && n < 3) continue;
            if (n == 0) {  // sigma_clip.rs:26-28
                median = 0.0;
                sigma = 1.0;
                break;
            }
            median = absel::exact_median_from(src, t, by_value, hist0, n, hist, spec_v);      // median.rs:27-44
            const absel::Keying by_dev = {1, median, 0.0f};
            absel::Spec spec_d;
            prepare_for(by_dev, it > 0, __float_as_uint(prev_mad) >> 21, &spec_d);
            const float mad_f32 = absel::median_f32_from(src, t, by_dev, hist0, n, hist, spec_d);  // sigma_clip.rs:14-16
            prev_mad = mad_f32;
            const double sig = fmax((double)mad_f32 * kMadToSigma, 1e-30);
            if (it == 2) {
                sigma = sig;
                break;
            }
            // retain v in [lo, hi] (sigma_clip.rs:19-23); kappa = 3.0f32 as f64
            const float lo = (float)(median - 3.0 * sig), hi = (float)(median + 3.0 * sig);
            t.lo = fmaxf(t.lo, lo);
            t.hi = fminf(t.hi, hi);
            if (!(lo <= hi)) {  // NaN bounds or empty interval: nothing is retained
                t.lo = __builtin_inff();
                t.hi = -__builtin_inff();
            }
            n = prepare_for(by_value, true, __float_as_uint((float)median) >> 21, &spec_v);
        }
        res.median = median;
        res.sigma = sigma;
    }
    return res;
}

__global__ __launch_bounds__(absel::kBlock) void tile_background_kernel(const float *__restrict__ img, int rows, int cols,
                                                                        int64_t ld, int step, int ntx, const ab_pixel_xf xf,
                                                                        TileOut *__restrict__ out) {
    __shared__ unsigned int hist0[2048], hist[2048], tally[2];
    __shared__ float cache[35 * absel::kBlock];  // 140 KiB of the CU's 160 KiB
    const int ty = blockIdx.x / ntx, tx = blockIdx.x % ntx;
    absel::Window t;
    t.img = img;
    t.ld = ld;
    t.y0 = ty * step;
    t.x0 = tx * step;
    t.y1 = min(t.y0 + step, rows);
    t.x1 = min(t.x0 + step, cols);
    t.min_valid = 1e-7f;  // star_detection.rs:56
    t.lo = -__builtin_inff();
    t.hi = __builtin_inff();
    t.xf = xf;
    absel::TileSource<35, 29> src;  // the whole tile on chip: one read of the frame per estimate_background
    src.lds = cache;
    src.load(t);
    __syncthreads();
    // a first guess for the very first median's top-level bin: the first candidate wave 0 holds in its first sweep
    __shared__ uint32_t s_guess;
    if (threadIdx.x == 0) s_guess = 0xffffffffu;
    __syncthreads();
    if (threadIdx.x < 64) {
        const float v0 = cache[threadIdx.x];
        const bool ok = absel::candidate(t, v0);
        const unsigned long long m = __ballot(ok);
        if (m && threadIdx.x == (unsigned)__builtin_ctzll(m)) s_guess = __float_as_uint(v0) >> 21;
    }
    __syncthreads();
    const uint32_t guess = s_guess;
    const TileOut res = tile_stats(src, t, hist0, hist, tally, guess != 0xffffffffu, guess);
    if (threadIdx.x == 0) out[blockIdx.x] = res;
}

// The same statistics from ONE histogram of the tile with the tile's keys held in registers (tile_bucket.hpp): round 2's product
// kernel, now the FALLBACK of the streaming kernel below (and the whole job behind AB_TILE_RESIDENT=1).  tile_background_kernel
// above is the round-1 radix-select version, kept behind AB_TILE_LEGACY=1 as an in-library cross-check
// (tests/test_gpu_tile_stats.py runs all three).
// (capping the registers at 168 so that another kernel's wave fits beside a tile on every SIMD -- amdgpu_waves_per_eu(3, 3) --
// spills 65 registers: 104 -> 130 us alone and the registration stage 19.8 -> 21.4 ms)
// `fail` given: the launch works off the list of tiles the streaming kernel declined ({count, finished blocks, ids ...}; id =
// plane * tiles-per-plane + tile) with however many blocks it has, and the last block to finish leaves the list empty again.
__global__ __launch_bounds__(tb::kThreads) void tile_background_bucket_kernel(const float *__restrict__ img_arg, int rows, int cols, int64_t ld,
                                                                              int step, int ntx, const ab_pixel_xf xf_arg, TileOut *__restrict__ out_arg,
                                                                              const FrameDev *__restrict__ fd,
                                                                              const float *const *__restrict__ many_planes = nullptr,
                                                                              const ab_pixel_xf *__restrict__ many_xf = nullptr,
                                                                              unsigned int *__restrict__ fail = nullptr, int tiles_per_plane = 0) {
    __shared__ tb::Shared sh;
    const unsigned int nwork = fail ? fail[0] : 1u;
#pragma unroll 1
    for (unsigned int wi = fail ? blockIdx.x : 0u; wi < nwork; wi += gridDim.x) {
#ifdef AB_TILE_LOOP_SYNC
        __syncthreads();
#endif
        const float *__restrict__ img = img_arg;
        TileOut *__restrict__ out = out_arg;
        unsigned int tile = blockIdx.x, plane = blockIdx.y, per_plane = gridDim.x;
        if (fail) {
            const unsigned int id = fail[2 + wi];
            per_plane = (unsigned int)tiles_per_plane;
            plane = id / per_plane;
            tile = id % per_plane;
        }
        // many_planes: `plane` names the plane (all of one size), its transform and its row of `out` -- the tiles of a whole
        // registration batch in ONE launch
        if (many_planes) {
            img = many_planes[plane];
            out += (size_t)plane * per_plane;
        }
        const ab_pixel_xf xf = many_planes ? many_xf[plane] : (fd ? fd->xf : xf_arg);
        const int ty0 = (int)(tile / (unsigned int)ntx) * step, tx0 = (int)(tile % (unsigned int)ntx) * step;
        const int y1 = min(ty0 + step, rows), x1 = min(tx0 + step, cols);
        // thread (tx, ty) of the 256 x 2 layout walks column tx0 + tx downwards, two rows per slot: consecutive lanes read
        // consecutive pixels; 32 loads are in flight before the first key is formed
        constexpr int kRowPhases = tb::kThreads / 256;
        const int tx = threadIdx.x & 255, ty = threadIdx.x >> 8;
        const int c = tx0 + tx;
        const bool col_ok = c < x1;
        tb::Keys K;
        tb::KeyRange kr;
        constexpr int kBatch = 32;
#pragma unroll
        for (int h = 0; h < tb::kSlots / kBatch; ++h) {
            float raw[kBatch];
#pragma unroll
            for (int i = 0; i < kBatch; ++i) {
                const int r = ty0 + ty + kRowPhases * (h * kBatch + i);
                raw[i] = (col_ok && r < y1) ? img[(int64_t)r * ld + c] : __builtin_nanf("");
            }
#pragma unroll
            for (int i = 0; i < kBatch; ++i) {
                const float v = ab_px(xf, raw[i]);
                const uint32_t key = (__builtin_isfinite(v) && v > 1e-7f) ? __float_as_uint(v) : 0u;  // star_detection.rs:56
                K.v[(h * kBatch + i) >> 5][(h * kBatch + i) & 31] = key;
                kr.add(key);
            }
        }
        const tb::TileResult r = tb::tile_stats(K, sh, kr);
        if (threadIdx.x == 0) {
            TileOut o;
            o.median = r.median;
            o.sigma = r.sigma;
            o.valid = r.valid;
            o.pad = fail ? 1 : 0;  // (1: a tile the streaming kernel declined -- the host counts them, AB_FB_TILES_DECLINED)
            out[tile] = o;
        }
        __syncthreads();  // (the next tile reuses the shared block)
    }
    if (fail && threadIdx.x == 0) {
        __threadfence();
        if (atomicAdd(&fail[1], 1u) == gridDim.x - 1u) {  // the last block: everything has been read
            fail[0] = 0u;
            fail[1] = 0u;
        }
    }
}

// The product kernel (round 4, tile_stream.hpp): nothing per pixel stays on chip -- histogram pass, a plan of the buckets the
// three clipping rounds will need, a second pass that collects those buckets' keys, the rounds on that list.  Three tiles per
// CU in flight at ~100 VGPRs each; a tile it cannot settle exactly is appended to `fail` for the kernel above.
__global__ __launch_bounds__(ts::kThreads) void tile_background_stream_kernel(const float *__restrict__ img_arg, int rows, int cols, int64_t ld, int step,
                                                                              int ntx, const ab_pixel_xf xf_arg, TileOut *__restrict__ out,
                                                                              const FrameDev *__restrict__ fd, const float *const *__restrict__ many_planes,
                                                                              const ab_pixel_xf *__restrict__ many_xf, unsigned int *__restrict__ fail,
                                                                              const TileCand cand) {
    __shared__ ts::Shared sh;
#ifdef AB_TILE_VGPR_FLOOR
    // developer experiment: a register floor caps the tile workgroups per CU (four of them take 156 of the 160 KB of LDS, and no
    // kernel that needs LDS -- label_init, the triangle kernels -- fits beside them)
#define AB_STR2(x) #x
#define AB_STR(x) AB_STR2(x)
    asm volatile("" ::: "v" AB_STR(AB_TILE_VGPR_FLOOR));
#endif
    const float *__restrict__ img = img_arg;
    if (many_planes) {
        img = many_planes[blockIdx.y];
        out += (size_t)blockIdx.y * gridDim.x;
    }
    const ab_pixel_xf xf = many_planes ? many_xf[blockIdx.y] : (fd ? fd->xf : xf_arg);
    ts::TileRect r;
    r.img = img;
    r.ld = ld;
    r.y0 = (int)(blockIdx.x / (unsigned int)ntx) * step;
    r.x0 = (int)(blockIdx.x % (unsigned int)ntx) * step;
    r.y1 = min(r.y0 + step, rows);
    r.x1 = min(r.x0 + step, cols);
    r.vec = (step & 3) == 0 && ((r.x1 - r.x0) & 3) == 0;
    ts::CandSink cs;
    if (cand.ent) {  // (the registration batch: whole tiles also leave their candidate lists for the labelling pass)
        const size_t tl = (size_t)blockIdx.y * gridDim.x + blockIdx.x;
        cs.ent = cand.ent + tl * (size_t)kCandCap;
        cs.cnt = cand.cnt + tl;
        cs.cut = cand.cut + tl;
        cs.cap = (unsigned int)kCandCap;
    }
    const ts::TileResult res = ts::tile_stats(sh, r, xf, cs);
    if ((int)threadIdx.x == 64 * ts::rounds_wave()) {
        if (res.declined) {
            const unsigned int at = atomicAdd(&fail[0], 1u);
            fail[2 + at] = blockIdx.y * gridDim.x + blockIdx.x;
        } else {
            TileOut o;
            o.median = res.median;
            o.sigma = res.sigma;
            o.valid = res.valid;
            o.pad = 0;
            out[blockIdx.x] = o;
        }
    }
}

// ---- threshold + union-find labelling -------------------------------------------------------------
__device__ __forceinline__ bool above(float v, double threshold) { return __builtin_isfinite(v) && (double)v > threshold; }

// threshold every pixel (one mask bit each; parent = self where labelled) and append the labelled ones -- a fraction of a
// percent of the frame -- to a list, so that the merge / numbering / statistics kernels touch only those.
// "is pixel j labelled": one bit per pixel (2 MiB for 4096^2, L2-resident for the kernels that test neighbours) instead of
// a -1 in the 64 MiB forest: the threshold pass writes P/8 bytes, not 4 P, and `parent` is only defined at labelled pixels
__device__ __forceinline__ bool labelled(const unsigned int *__restrict__ mask, int j) { return (mask[j >> 5] >> (j & 31)) & 1u; }

// A 256-thread block owns kInitRounds consecutive sub-spans of 1024 pixels, collects their labelled indices in LDS and
// reserves list space with ONE global atomic at the end (or whenever the LDS list could overflow, which needs > 75 % of
// the pixels above threshold).  With one reservation per 8192 pixels the 2048 same-address atomics-with-return of a
// 4096^2 frame took 25 of the kernel's 36 us.
// Round 4: 256 threads and 16 KB of LDS per block (was 1024 threads, 64 KB).  Alone the kernel takes the same 22 us, but inside a
// registration batch a 16-wave workgroup waited for a CU with sixteen free wave slots and 64 KB of LDS while the warp's and the
// tile kernel's small workgroups kept slipping in ahead of it: 110 us on average.  Same-box A/B of the stage: 12.9 -> 12.4 ms;
// roots_kernel likewise (kRootsBlock 1024 -> 256): another 0.5 - 0.8 ms.
constexpr int kInitBlock = 256, kInitSub = 4 * kInitBlock, kInitRounds = 16, kInitCap = 4 * kInitSub;
__device__ __forceinline__ void label_init_body(const float *__restrict__ img, int rows, int cols, int64_t ld,
                                                                double threshold_arg, const ab_pixel_xf xf_arg, int *__restrict__ parent,
                                                                unsigned int *__restrict__ mask, int *__restrict__ plist, unsigned int *nlab,
                                                                int vec_ok, const FrameDev *__restrict__ fd) {
    const double threshold = fd ? fd->threshold : threshold_arg;
    const ab_pixel_xf xf = fd ? fd->xf : xf_arg;
    __shared__ int found[kInitCap];
    __shared__ unsigned int nfound, base;
    if (threadIdx.x == 0) nfound = 0;
    __syncthreads();
    const int P = rows * cols;
    const int lane = threadIdx.x & 63;
    auto flush = [&]() {  // block-uniform call sites only
        __syncthreads();
        const unsigned int n = nfound;
        if (n) {
            if (threadIdx.x == 0) base = atomicAdd(nlab, n);
            __syncthreads();
            for (unsigned int k = threadIdx.x; k < n; k += kInitBlock) plist[base + k] = found[k];
            __syncthreads();
            if (threadIdx.x == 0) nfound = 0;
        }
        __syncthreads();
    };
    // every loop below has a block-uniform trip count and predicates instead of early exits: the mask words are assembled
    // across lanes
    for (int round = 0; round < kInitRounds; ++round) {
        const int64_t start64 = ((int64_t)blockIdx.x * kInitRounds + round) * kInitSub;
        if (start64 >= P) break;  // uniform
        const int start = (int)start64;
        if (vec_ok) {  // contiguous 16-byte aligned plane of 4 k pixels: no row / column arithmetic, 16-byte loads
            const int i = start + threadIdx.x * 4;  // i % 32 == 4 * (lane % 8): eight lanes make one mask word
            const bool in = i < P;
            float4 v = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
            if (in) v = reinterpret_cast<const float4 *>(img)[i >> 2];
            const bool b0 = in && above(ab_px(xf, v.x), threshold), b1 = in && above(ab_px(xf, v.y), threshold),
                       b2 = in && above(ab_px(xf, v.z), threshold), b3 = in && above(ab_px(xf, v.w), threshold);
            unsigned int w = ((unsigned int)b0 | ((unsigned int)b1 << 1) | ((unsigned int)b2 << 2) | ((unsigned int)b3 << 3)) << (4 * (lane & 7));
            w |= __shfl_xor(w, 1, 64);
            w |= __shfl_xor(w, 2, 64);
            w |= __shfl_xor(w, 4, 64);
            if ((lane & 7) == 0 && in) mask[i >> 5] = w;
            const int cnt = (int)b0 + (int)b1 + (int)b2 + (int)b3;
            if (cnt) {  // ascending order inside the thread; the list's order across threads is irrelevant
                unsigned int at = atomicAdd(&nfound, (unsigned int)cnt);
                if (b0) { parent[i] = i; found[at++] = i; }
                if (b1) { parent[i + 1] = i + 1; found[at++] = i + 1; }
                if (b2) { parent[i + 2] = i + 2; found[at++] = i + 2; }
                if (b3) { parent[i + 3] = i + 3; found[at++] = i + 3; }
            }
        } else {
            for (int off = threadIdx.x; off < kInitSub; off += kInitBlock) {
                const int i = start + off;  // a wave covers the 64 consecutive pixels from i - lane (a multiple of 64)
                bool is = false;
                if (i < P) {
                    const int r = i / cols, c = i - r * cols;
                    is = above(ab_px(xf, img[r * ld + c]), threshold);
                }
                const unsigned long long m = __ballot(is);
                const int i0 = i - lane;
                if (lane == 0 && i0 < P) mask[i0 >> 5] = (unsigned int)m;
                if (lane == 1 && i0 + 32 < P) mask[(i0 >> 5) + 1] = (unsigned int)(m >> 32);
                if (is) {
                    parent[i] = i;
                    found[atomicAdd(&nfound, 1u)] = i;
                }
            }
        }
        __syncthreads();
        if (nfound > (unsigned int)(kInitCap - kInitSub)) flush();  // uniform: nfound is read after the barrier
    }
    flush();
}

__device__ __forceinline__ int uf_find(int *parent, int x) {
    while (true) {
        const int p = __hip_atomic_load(&parent[x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (p == x) return x;
        x = p;
    }
}

__device__ __forceinline__ void uf_union(int *parent, int a, int b) {
    while (true) {
        a = uf_find(parent, a);
        b = uf_find(parent, b);
        if (a == b) return;
        if (a < b) {
            const int t = a;
            a = b;
            b = t;
        }
        const int old = atomicMin(&parent[a], b);  // hook the larger root under the smaller
        if (old == a) return;
        a = old;
    }
}

__device__ __forceinline__ void label_merge_body(int rows, int cols, int *parent, const unsigned int *__restrict__ mask,
                                                          const int *__restrict__ plist, const unsigned int *__restrict__ nlab) {
    const unsigned int n = *nlab;
    for (unsigned int k = blockIdx.x * 256 + threadIdx.x; k < n; k += gridDim.x * 256) {
        const int i = plist[k];
        const int r = i / cols, c = i - r * cols;
        // forward half of the 8-neighbourhood (star_detection.rs:120): E, SW, S, SE
        if (c + 1 < cols && labelled(mask, i + 1)) uf_union(parent, i, i + 1);
        if (r + 1 < rows) {
            const int d = i + cols;
            if (c > 0 && labelled(mask, d - 1)) uf_union(parent, i, d - 1);
            if (labelled(mask, d)) uf_union(parent, i, d);
            if (c + 1 < cols && labelled(mask, d + 1)) uf_union(parent, i, d + 1);
        }
    }
}

struct CompStat {  // filled by atomics (order-independent integers)
    int npix, x0, x1, y0, y1, first_interior;
    // group path only (comp_stats_body<true>): sum of max(v - background, 0) over the member pixels, added by f64 atomics in whatever
    // order the waves arrive.  It only RANKS components (comp_select_kernel); every number a star is made of is recomputed by
    // comp_moments' fixed butterfly.  (For normalised pixels the sum is exact in f64, hence the same in any order: DESIGN 4.3.)
    double flux;
};
static_assert(sizeof(CompStat) == 32, "CompStat layout");

struct CompRec {  // what the host needs to finish one star (star_detection.rs:147-213)
    int first_interior, npix;
    double sum_flux, sum_x, sum_y, peak, sum_r2, sum_xx, sum_yy, sum_xy;
};

// ---- threshold + TILE-LOCAL union-find in LDS (round 5) ----------------------------------------------------------
// label_init + label_merge as above are two passes over global memory: the threshold pass writes parent = self, the merge pass then
// walks the list and hooks roots with global atomics -- 170 000 labelled pixels of a 4096^2 frame, every union a chain of dependent
// L2 round trips (16 us per frame alone, 110 us per group inside a batch).  Stars are a few pixels wide, so almost every union joins
// two pixels of one small neighbourhood.  Here a 256-thread workgroup owns a tile of 32 rows x 128 columns: it thresholds its 4096
// pixels (one 16-byte load x 4 per thread), keeps the tile's mask bits and a label per pixel in LDS, runs the SAME union-find
// (atomicMin hooking: the root is the smallest raster index) on the LDS labels over the forward half of the 8-neighbourhood
// (star_detection.rs:120) inside the tile, and writes for every labelled pixel parent = the global index of its tile-local root.
// Only pixels on the tile's last row, first and last column can have forward neighbours in another tile: they go on a BORDER list
// (~8 % of the labelled pixels) and label_border_kernel does their cross-tile unions on the global forest.  The forest that
// results has the same roots as before (a component's minimum raster index), so everything downstream is unchanged.
// Needs contiguous planes whose width is a multiple of 32 (mask words then never straddle a tile); other planes keep the two-pass form.
constexpr int kTileH = 32, kTileW = 128, kTileThreads = 256;
__device__ __forceinline__ int lds_find(int *lab, int x) {
    while (true) {
        const int p = __hip_atomic_load(&lab[x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (p == x) return x;
        x = p;
    }
}
__device__ __forceinline__ void lds_union(int *lab, int a, int b) {
    while (true) {
        a = lds_find(lab, a);
        b = lds_find(lab, b);
        if (a == b) return;
        if (a < b) {
            const int t = a;
            a = b;
            b = t;
        }
        const int old = atomicMin(&lab[a], b);
        if (old == a) return;
        a = old;
    }
}
// The lists are split into kRegions REGIONS (tile t appends to region t mod kRegions, each with its own counter 256 bytes from the
// next and its own segment of the list): one reservation per tile on ONE counter was 8192 same-address atomics-with-return per
// frame -- ~12 ns each, serialised in the L2: 100 of the kernel's 110 us (the first version measured 53 us per frame against 34
// for the two passes it replaces).  The consumers walk region by region (blockIdx.x mod kRegions).
constexpr int kRegions = 64, kRegionPitch = 64;  // counters: lcnt[r * kRegionPitch] = labelled, lcnt[(kRegions + r) * kRegionPitch] = border

// One row of a tile's mask (128 bits) and the RUNS in it.  The unions go run by run (AB_LABEL_PIXELWISE=1 keeps the pixel-by-pixel
// form): a horizontal run of labelled pixels is one node of the tile's forest, named by its first pixel, so the E neighbour needs no
// union at all, and the SW / S / SE neighbours of all pixels of a run [s, e] are the runs of the next row that touch columns
// [s - 1, e + 1] -- one union per touching run instead of up to three per pixel.  A 13 x 13 star is ~13 + 12 nodes and ~12 unions
// where the pixel form makes ~500 with every find a chain of dependent LDS round trips (tools/label_bench.hip: 30 of the kernel's
// 42 us per frame).  The forest has the same roots: a component's smallest raster index always starts a run.
struct RowBits {
    unsigned long long lo, hi;  // columns 0 .. 63, 64 .. 127
};
__device__ __forceinline__ unsigned long long ones64(int a, int b) {  // bits a .. b, 0 <= a <= b <= 63
    return (b == 63 ? ~0ull : ((1ull << (b + 1)) - 1ull)) & ~((1ull << a) - 1ull);
}
__device__ __forceinline__ RowBits row_bits(const unsigned int (*tmask)[kTileW / 32], int r) {
    const uint4 w = *reinterpret_cast<const uint4 *>(tmask[r]);
    return {(unsigned long long)w.x | ((unsigned long long)w.y << 32), (unsigned long long)w.z | ((unsigned long long)w.w << 32)};
}
__device__ __forceinline__ RowBits row_range(const RowBits &m, int a, int b) {  // m restricted to columns a .. b (0 <= a <= b <= 127)
    RowBits o = {0ull, 0ull};
    if (a < 64) o.lo = m.lo & ones64(a, b < 63 ? b : 63);
    if (b >= 64) o.hi = m.hi & ones64(a > 64 ? a - 64 : 0, b - 64);
    return o;
}
__device__ __forceinline__ int run_start(const RowBits &m, int c) {  // first column of the run that holds column c
    if (c >= 64) {
        const unsigned long long z = ~m.hi & ((1ull << (c - 64)) - 1ull);
        if (z) return 128 - __builtin_clzll(z);
        const unsigned long long zl = ~m.lo;
        return zl ? 64 - __builtin_clzll(zl) : 0;
    }
    const unsigned long long z = ~m.lo & ((1ull << c) - 1ull);
    return z ? 64 - __builtin_clzll(z) : 0;
}
__device__ __forceinline__ int run_end(const RowBits &m, int c) {  // last column of the run that holds column c
    if (c < 64) {
        const unsigned long long z = ~m.lo >> c;  // (bit 0 is clear: column c is labelled)
        if (z) return c + __builtin_ctzll(z) - 1;
        const unsigned long long zh = ~m.hi;
        return zh ? 63 + __builtin_ctzll(zh) : 127;
    }
    const unsigned long long z = ~m.hi >> (c - 64);
    return z ? c + __builtin_ctzll(z) - 1 : 127;
}

// RECS (with RUNS; the registration batch's chained form): the tile also gathers what comp_stats would -- size, bounding box, first
// interior pixel and approximate flux of every tile-local component, accumulated in LDS from one contribution per run piece -- and
// emits ONE RECORD per tile-local component: st[pos] (CompStat), roots[pos] = its root pixel, cid[root] = pos, pos taken from the
// record segment of the tile's record region.  No pixel list is written: roots_many / comp_stats_many (two walks over the frame's
// ~170 000 labelled pixels, 7 global atomics per run of equal roots) are replaced by comp_merge_many over the ~10 000 records, of
// which only those whose tile root was hooked under another tile's (a star on a tile border) have anything to do.  A tile with more
// than kTileSlots components raises the frame's overflow flag: the host redoes that frame through the full path.
constexpr int kTileSlots = 64;
constexpr int kRecRegions = 16;  // record regions (tile t -> region t mod kRecRegions), counters at lcnt[(2 kRegions + r) * kRegionPitch]
struct TileRecOut {
    CompStat *st;
    int *roots, *cid;
    size_t stride;        // records per region segment (= tiles of a region x kTileSlots: a segment cannot overflow)
    unsigned int *flags;  // [0] |= 1: some tile had more than kTileSlots components
    double bg_median;
};
#ifdef AB_LABEL_TIMING  // (tools/label_bench.hip: s_memtime of every wave at each phase boundary)
__device__ long long *g_label_marks = nullptr;  // [tile][wave][8]
#define LT_MARK(i)                                                                                                   \
    do {                                                                                                             \
        if (g_label_marks && (threadIdx.x & 63) == 0 && blockIdx.y == 0)                                             \
            g_label_marks[((size_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * 8 + (i)] = (long long)__builtin_amdgcn_s_memtime(); \
    } while (0)
#else
#define LT_MARK(i) \
    do {           \
    } while (0)
#endif
template <bool RUNS, bool RECS = false>
__device__ __forceinline__ void label_tile_body(const float *__restrict__ img, int rows, int cols, double threshold, const ab_pixel_xf xf,
                                                int *__restrict__ parent, unsigned int *__restrict__ mask, int *__restrict__ plist_all, size_t plist_stride,
                                                int *__restrict__ blist_all, size_t blist_stride, unsigned int *lcnt, const TileRecOut ro = TileRecOut(), int mpitch = 0) {
    static_assert(RUNS || !RECS, "records need the run form");
    // mpitch: bits per row of the mask (a multiple of 32 >= cols; 0 = cols, which must then be one): with padded mask rows a tile's 32-
    // column words never straddle rows, whatever the plane's width (round 5: any width takes this path; cols % 32 == 0 used to be required)
    const int64_t mp = mpitch ? mpitch : cols;
    const int region = (int)(blockIdx.x % kRegions);
    int *__restrict__ plist = plist_all + (size_t)region * plist_stride, *__restrict__ blist = blist_all + (size_t)region * blist_stride;
    unsigned int *nlab = lcnt + region * kRegionPitch, *nborder = lcnt + (kRegions + region) * kRegionPitch;
    __shared__ __attribute__((aligned(16))) unsigned int tmask[kTileH][kTileW / 32];
    __shared__ int lab[kTileH * kTileW];
    __shared__ unsigned int n_found, n_edge, base_found, base_edge;
    __shared__ int acc_i[RECS ? 7 : 1][RECS ? kTileSlots : 1];  // npix, x0, x1, y0, y1, first_interior, root (tile index)
    __shared__ double acc_flux[RECS ? kTileSlots : 1];
    __shared__ unsigned int n_slots, base_rec;
    const int tid = threadIdx.x, lane = tid & 63;
    const int tiles_x = (cols + kTileW - 1) / kTileW;
    const int ty0 = (int)(blockIdx.x / tiles_x) * kTileH, tx0 = (int)(blockIdx.x % tiles_x) * kTileW;
    const int q = tid & 31, r0 = tid >> 5;  // this thread: columns 4 q .. 4 q + 3 of rows r0, r0 + 8, r0 + 16, r0 + 24
    if (tid == 0) n_found = n_edge = n_slots = 0;
    LT_MARK(0);
    unsigned int bits = 0;  // bit 4 j + k: pixel (r0 + 8 j, 4 q + k)
    float4 v[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int r = ty0 + r0 + 8 * j, c = tx0 + 4 * q;
        v[j] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        if (r < rows && c + 3 < cols) {
            v[j] = ts::load4u(img + (int64_t)r * cols + c);  // (any dword address)
        } else if (r < rows && c < cols) {  // the ragged quad at the end of a row whose width is not a multiple of 4
            const float *p = img + (int64_t)r * cols + c;
            v[j].x = p[0];
            if (c + 1 < cols) v[j].y = p[1];
            if (c + 2 < cols) v[j].z = p[2];
        }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int r = ty0 + r0 + 8 * j, c = tx0 + 4 * q;
        const bool in = r < rows && c < cols;
        v[j] = make_float4(ab_px(xf, v[j].x), ab_px(xf, v[j].y), ab_px(xf, v[j].z), ab_px(xf, v[j].w));  // (RECS adds these up further down)
        const unsigned int b = (unsigned int)(in && above(v[j].x, threshold)) | ((unsigned int)(in && c + 1 < cols && above(v[j].y, threshold)) << 1) |
                               ((unsigned int)(in && c + 2 < cols && above(v[j].z, threshold)) << 2) | ((unsigned int)(in && c + 3 < cols && above(v[j].w, threshold)) << 3);
        bits |= b << (4 * j);
        unsigned int w = b << (4 * (lane & 7));  // eight lanes make one mask word
        w |= __shfl_xor(w, 1, 64);
        w |= __shfl_xor(w, 2, 64);
        w |= __shfl_xor(w, 4, 64);
        if ((lane & 7) == 0) {
            tmask[r0 + 8 * j][q >> 3] = w;
            if (in) mask[((int64_t)r * mp + c) >> 5] = w;  // (mp % 32 == 0 and tx0 % 32 == 0: a whole word of this row)
        }
    }
    LT_MARK(1);  // loads + threshold + mask
    // flatten's bookkeeping can start now: list + border list positions (one LDS atomic per thread, one global atomic per workgroup)
    const int cnt = __builtin_popcount(bits);
    unsigned int edge_bits = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int r = r0 + 8 * j, c = 4 * q + k;
            const bool on_edge = r == kTileH - 1 || c == 0 || c == kTileW - 1;
            if (((bits >> (4 * j + k)) & 1u) && on_edge) edge_bits |= 1u << (4 * j + k);
        }
    const int ecnt = __builtin_popcount(edge_bits);
    unsigned int starts = 0;  // RUNS: the pixels of this thread that start a run (the pixel to their left is not labelled)
    if constexpr (RUNS) {
        const unsigned int left = __shfl_up(bits, 1, 64);  // lane - 1 holds columns 4 q - 4 .. 4 q - 1 of the same rows when q > 0
        const unsigned int prev = q > 0 ? ((left >> 3) & 0x1111u) : 0u;  // bit 4 j: pixel (r0 + 8 j, 4 q - 1)
        starts = bits & ~(((bits << 1) & 0xeeeeu) | prev);
        unsigned int todo = starts;
        while (todo) {
            const int bpos = __builtin_ctz(todo);
            todo &= todo - 1;
            const int li = (r0 + 8 * (bpos >> 2)) * kTileW + 4 * q + (bpos & 3);
            lab[li] = li;
        }
    } else {
        // labels: own local index where labelled
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if ((bits >> (4 * j + k)) & 1u) {
                    const int li = (r0 + 8 * j) * kTileW + 4 * q + k;
                    lab[li] = li;
                }
    }
    LT_MARK(2);  // run starts
    __syncthreads();
    unsigned int at = 0, eat = 0;
    if (cnt && !RECS) at = atomicAdd(&n_found, (unsigned int)cnt);
    if (ecnt) eat = atomicAdd(&n_edge, (unsigned int)ecnt);
    if constexpr (RUNS) {
        // unions: every run with the runs of the next row that touch [s - 1, e + 1]
        unsigned int todo = starts;
        while (todo) {
            const int bpos = __builtin_ctz(todo);
            todo &= todo - 1;
            const int r = r0 + 8 * (bpos >> 2), s0 = 4 * q + (bpos & 3);
            if (r + 1 >= kTileH) continue;
            const RowBits m = row_bits(tmask, r), below = row_bits(tmask, r + 1);
            const int e0 = run_end(m, s0);
            RowBits nb = row_range(below, s0 > 0 ? s0 - 1 : 0, e0 + 1 < kTileW ? e0 + 1 : kTileW - 1);
            while (nb.lo | nb.hi) {
                const int p = nb.lo ? __builtin_ctzll(nb.lo) : 64 + __builtin_ctzll(nb.hi);
                const int s1 = run_start(below, p), e1 = run_end(below, p);
                lds_union(lab, r * kTileW + s0, (r + 1) * kTileW + s1);
                // (that run is done: drop its columns p .. e1 from the set)
                if (p < 64) nb.lo &= ~ones64(p, e1 < 63 ? e1 : 63);
                if (e1 >= 64) nb.hi &= ~ones64(p > 64 ? p - 64 : 0, e1 - 64);
            }
        }
        LT_MARK(3);  // unions
        __syncthreads();
        // every run's node -> its root, so that a pixel reads its root in one step
        todo = starts;
        while (todo) {
            const int bpos = __builtin_ctz(todo);
            todo &= todo - 1;
            const int li = (r0 + 8 * (bpos >> 2)) * kTileW + 4 * q + (bpos & 3);
            if constexpr (RECS) {
                // ... and a root takes a slot of the tile's record table and marks itself with -1 - slot (walkers stop at a mark)
                int x = li;
                while (true) {
                    const int pnt = __hip_atomic_load(&lab[x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    if (pnt < 0 || pnt == x) break;
                    x = pnt;
                }
                if (x == li) {  // (only a node's owner marks it, and this loop visits every run once)
                    const unsigned int slot = atomicAdd(&n_slots, 1u);
                    if (slot < (unsigned int)kTileSlots) {
                        acc_i[0][slot] = 0;
                        acc_i[1][slot] = 0x7fffffff;
                        acc_i[2][slot] = -1;
                        acc_i[3][slot] = 0x7fffffff;
                        acc_i[4][slot] = -1;
                        acc_i[5][slot] = 0x7fffffff;
                        acc_i[6][slot] = li;
                        acc_flux[slot] = 0.0;
                    }
                    __hip_atomic_store(&lab[li], -1 - (int)slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                } else {
                    __hip_atomic_store(&lab[li], x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                }
            } else {
                const int root = lds_find(lab, li);
                if (root != li) __hip_atomic_store(&lab[li], root, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);  // (a node others walk through: it still points at an ancestor)
            }
        }
    } else {
        auto lbl = [&](int r, int c) -> bool { return (tmask[r][c >> 5] >> (c & 31)) & 1u; };
        // unions inside the tile, forward half of the 8-neighbourhood: E, SW, S, SE
        unsigned int todo = bits;
        while (todo) {
            const int bpos = __builtin_ctz(todo);
            todo &= todo - 1;
            const int r = r0 + 8 * (bpos >> 2), c = 4 * q + (bpos & 3), li = r * kTileW + c;
            if (c + 1 < kTileW && lbl(r, c + 1)) lds_union(lab, li, li + 1);
            if (r + 1 < kTileH) {
                const int d = li + kTileW;
                if (c > 0 && lbl(r + 1, c - 1)) lds_union(lab, li, d - 1);
                if (lbl(r + 1, c)) lds_union(lab, li, d);
                if (c + 1 < kTileW && lbl(r + 1, c + 1)) lds_union(lab, li, d + 1);
            }
        }
    }
    LT_MARK(4);  // roots, slots
    __syncthreads();
    if constexpr (RECS) {
        const unsigned int rec_region = blockIdx.x % kRecRegions;
        if (tid == 0) {
            base_edge = n_edge ? atomicAdd(nborder, n_edge) : 0u;
            const unsigned int ns = n_slots;
            base_rec = ns ? atomicAdd(lcnt + (2 * kRegions + rec_region) * kRegionPitch, ns < (unsigned int)kTileSlots ? ns : (unsigned int)kTileSlots) : 0u;
            if (ns > (unsigned int)kTileSlots) atomicOr(ro.flags, 1u);
        }
        // every run piece of this thread: its pixels' parents, and one contribution to its component's slot
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            unsigned int bb = (bits >> (4 * j)) & 15u;
            if (!bb) continue;
            const int r = r0 + 8 * j, gy = ty0 + r;
            const RowBits m = row_bits(tmask, r);
            const float f4[4] = {v[j].x, v[j].y, v[j].z, v[j].w};
            while (bb) {  // (at most two pieces in four pixels)
                const int k0 = __builtin_ctz(bb), len = __builtin_ctz(~(bb >> k0));
                const int c0 = 4 * q + k0, s0 = k0 ? c0 : run_start(m, c0);
                const int node = r * kTileW + s0, pnt = lab[node];
                const int rt = pnt >= 0 ? pnt : node;
                const int slot = -1 - (pnt >= 0 ? lab[pnt] : pnt);  // (every root is marked since the barrier)
                const int groot = (ty0 + (rt >> 7)) * cols + tx0 + (rt & (kTileW - 1));
                double fl = 0.0;
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    if (k >= k0 && k < k0 + len) {
                        fl += fmax((double)f4[k] - ro.bg_median, 0.0);
                        parent[gy * cols + tx0 + 4 * q + k] = groot;
                    }
                if (slot < kTileSlots) {
                    const int x0 = tx0 + c0, x1 = x0 + len - 1;
                    atomicAdd(&acc_i[0][slot], len);
                    atomicMin(&acc_i[1][slot], x0);
                    atomicMax(&acc_i[2][slot], x1);
                    atomicMin(&acc_i[3][slot], gy);
                    atomicMax(&acc_i[4][slot], gy);
                    // BFS seeds are interior (star_detection.rs:107-110): the piece's first pixel off the frame's border columns, on an interior row
                    const int cf = x0 > 1 ? x0 : 1, cl = x1 < cols - 2 ? x1 : cols - 2;
                    if (gy >= 1 && gy < rows - 1 && cf <= cl) atomicMin(&acc_i[5][slot], gy * cols + cf);
                    if (fl > 0.0) unsafeAtomicAdd(&acc_flux[slot], fl);
                }
                bb &= ~(((1u << len) - 1u) << k0);
            }
        }
        LT_MARK(5);  // contributions + parents
        __syncthreads();
        eat += base_edge;
        unsigned int todo = edge_bits;
        while (todo) {
            const int bpos = __builtin_ctz(todo);
            todo &= todo - 1;
            blist[eat++] = (ty0 + r0 + 8 * (bpos >> 2)) * cols + tx0 + 4 * q + (bpos & 3);
        }
        const unsigned int ns = n_slots < (unsigned int)kTileSlots ? n_slots : (unsigned int)kTileSlots;
        if ((unsigned int)tid < ns) {
            const size_t pos = (size_t)rec_region * ro.stride + base_rec + (unsigned int)tid;
            const int rt = acc_i[6][tid], groot = (ty0 + (rt >> 7)) * cols + tx0 + (rt & (kTileW - 1));
            ro.st[pos] = CompStat{acc_i[0][tid], acc_i[1][tid], acc_i[2][tid], acc_i[3][tid], acc_i[4][tid], acc_i[5][tid], acc_flux[tid]};
            ro.roots[pos] = groot;
            ro.cid[groot] = (int)pos;
        }
        LT_MARK(6);  // border list + records
        return;
    }
    if (tid == 0) {
        base_found = n_found ? atomicAdd(nlab, n_found) : 0u;
        base_edge = n_edge ? atomicAdd(nborder, n_edge) : 0u;
    }
    __syncthreads();
    at += base_found;
    eat += base_edge;
    // flatten -> global forest
    if constexpr (RUNS) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const unsigned int b = (bits >> (4 * j)) & 15u;
            if (!b) continue;
            const int r = r0 + 8 * j;
            const RowBits m = row_bits(tmask, r);
            int root[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) root[k] = ((b >> k) & 1u) ? lab[r * kTileW + run_start(m, 4 * q + k)] : 0;
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if ((b >> k) & 1u) {
                    const int rt = root[k];  // (every node points at its root since the barrier)
                    const int gi = (ty0 + r) * cols + tx0 + 4 * q + k, groot = (ty0 + (rt >> 7)) * cols + tx0 + (rt & (kTileW - 1));
                    parent[gi] = groot;
                    plist[at++] = gi;
                    if ((edge_bits >> (4 * j + k)) & 1u) blist[eat++] = gi;
                }
        }
    } else {
        unsigned int todo = bits;
        while (todo) {
            const int bpos = __builtin_ctz(todo);
            todo &= todo - 1;
            const int r = r0 + 8 * (bpos >> 2), c = 4 * q + (bpos & 3), li = r * kTileW + c;
            const int root = lds_find(lab, li);
            const int gi = (ty0 + r) * cols + tx0 + c, groot = (ty0 + (root >> 7)) * cols + tx0 + (root & (kTileW - 1));
            parent[gi] = groot;
            plist[at++] = gi;
            if ((edge_bits >> bpos) & 1u) blist[eat++] = gi;
        }
    }
}

// ---- round 6: a frame labelled from its background tiles' CANDIDATE LISTS --------------------------------------
// label_tile_body above streams the whole frame from HBM to find the ~1 % of its pixels above the threshold (loads + threshold + mask
// words: 11.7 of its 24 us per 4096^2 frame) and pays the fixed cost of its LDS phases 4096 times per frame.  The tile pass of the
// background estimate has had every pixel in hand already: whole 256 x 256 tiles leave {position, raw value} of everything above a
// conservative cut (tile_stream.hpp: CandSink).  Here ONE workgroup labels one BACKGROUND tile from that list -- the frame is not read
// at all -- with the same outputs as label_tile_body<true, true>: mask words, parent = global index of the tile-local root, the border
// list, one record per tile-local component.  The forest is the same forest (a component's root is its smallest raster index; here
// a tile-local root is the tile's smallest, and label_border joins tiles on the global forest as before), so everything downstream
// (label_border with the tile's dimensions, comp_merge, the selection, comp_moments) is unchanged and the stars are identical.
//   sparse tile: count <= kCandCap, cut not NaN, and xf(cut) <= threshold (xf non-decreasing: what is not on the list cannot be above)
//   dense tile:  anything else (a partial tile, a tile the stream kernel declined, a tile brighter than the frame's threshold, an
//                overflowed list): the tile's pixels are read from the frame and tested one by one -- same code after the test.
// Nodes of the tile's forest are RUNS of labelled pixels numbered in raster order (row prefix + popcount of the run-start bits), so the
// union-find needs kBgRunCap labels, not 65 536; more runs or more than kBgSlots components raise the frame's overflow flag (the host
// redoes the frame through the full path, like a crowded 32 x 128 tile before).
// LDS is what this kernel waits for inside a registration batch: the background-tile workgroups of the tile pipeline hold 39 KB each, four
// per CU (157 of 160 KB), so a labelling workgroup only finds room where a tile workgroup has just retired.  The first version (whole
// 256 x 256 tiles, 62 KB: run descriptors, 4096 labels, 512 slots) waited ten times longer than it ran; at 36 KB the step fell by 0.43
// ms (profiles/r06_lds_ab.txt).  Hence: a workgroup labels a SUB-TILE of kBgRows rows x 256 columns of its background tile (it reads
// the tile's whole list, 8 entries per thread, and keeps its rows'), there are no run descriptors (a run's owner knows its row and
// first column where they are needed; a root leaves its pixel index in its slot), and the caps are per sub-tile.
#ifndef AB_BG_ROWS
#define AB_BG_ROWS 128
#endif
#ifndef AB_BG_RUNCAP
#define AB_BG_RUNCAP (12 * AB_BG_ROWS)
#endif
#ifndef AB_BG_SLOTS
#define AB_BG_SLOTS (2 * AB_BG_ROWS)
#endif
constexpr int kBgT = 256, kBgRows = AB_BG_ROWS, kBgSub = kBgT / kBgRows, kBgThreads = 256, kBgRunCap = AB_BG_RUNCAP, kBgSlots = AB_BG_SLOTS;
static_assert(kBgRows == 256 || kBgRows == 128 || kBgRows == 64 || kBgRows == 32, "sub-tiles: whole rows of the 256-px background tile");
struct BgShared {
    unsigned int tmask[kBgRows][8];     // one bit per pixel of the sub-tile: above the threshold
    unsigned char wpre[kBgRows][8];     // runs of row r that start in the words before word w
    unsigned int row_base[kBgRows + 4]; // runs in the rows above r; [kBgRows] = the sub-tile's runs
    int lab[kBgRunCap];                 // the forest over run ids; after the flattening: root id, or -1 - slot at a root
    int acc_i[7][kBgSlots];             // npix, x0, x1, y0, y1, first_interior, the root's global pixel index
    double acc_flux[kBgSlots];
    unsigned int wave_tot[kBgThreads / 64];
    unsigned int n_edge, base_edge, n_slots, base_rec;
};
static_assert(sizeof(BgShared) <= 40 * 1024, "a labelling workgroup must fit where ONE tile workgroup (39.3 KB) has retired");
__device__ __forceinline__ unsigned int bg_start_word(const unsigned int (*tm)[8], int r, int w) {
    const unsigned int m = tm[r][w], prev = w ? tm[r][w - 1] >> 31 : 0u;
    return m & ~((m << 1) | prev);
}
// runs of row r that start at or before column c
__device__ __forceinline__ unsigned int bg_rank_incl(const BgShared &sh, int r, int c) {
    const int w = c >> 5;
    return (unsigned int)sh.wpre[r][w] + (unsigned int)__builtin_popcount(bg_start_word(sh.tmask, r, w) & (0xffffffffu >> (31 - (c & 31))));
}
struct FrameCandDev {
    const uint2 *ent;
    const unsigned int *cnt;
    const float *cut;
};
// grid.x = background tiles x kBgSub: workgroup b labels rows [sub * kBgRows, (sub + 1) * kBgRows) of background tile b / kBgSub, sub = b % kBgSub
__device__ __forceinline__ void label_bgtile_body(BgShared &sh, const float *__restrict__ img, int rows, int cols, double threshold, const ab_pixel_xf xf,
                                                  int *__restrict__ parent, unsigned int *__restrict__ mask, int *__restrict__ blist_all, size_t blist_stride,
                                                  unsigned int *lcnt, const TileRecOut ro, int mpitch, const FrameCandDev cand) {
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int tiles_x = (cols + kBgT - 1) / kBgT;
    const unsigned int bgtile = blockIdx.x / kBgSub;
    const int sub = (int)(blockIdx.x % kBgSub);
    const int ty0 = (int)(bgtile / tiles_x) * kBgT + sub * kBgRows, tx0 = (int)(bgtile % tiles_x) * kBgT;
    const int region = (int)(blockIdx.x % kRegions), rec_region = (int)(blockIdx.x % kRecRegions);
    int *__restrict__ blist = blist_all + (size_t)region * blist_stride;
    unsigned int *nborder = lcnt + (kRegions + region) * kRegionPitch;
    const int64_t mp = mpitch;
    if (ty0 >= rows) return;  // (the last background tile's lower sub-tiles may lie below the frame; block-uniform)
    // ---- which form ----
    const unsigned int n = cand.ent ? cand.cnt[bgtile] : 0xffffffffu;
    const float cut = cand.ent ? cand.cut[bgtile] : __builtin_nanf("");
    const bool sparse = cand.ent && n <= (unsigned int)kCandCap && cut == cut && !((double)ab_px(xf, cut) > threshold);  // (block-uniform)
    if (tid == 0) {
        sh.n_edge = sh.n_slots = 0;
        if (!sparse && sub == 0) atomicAdd(ro.flags, 2u);  // (bit 0 is the overflow flag; the host reports flags >> 1 = background tiles read from the frame)
    }
    constexpr int kPerThread = kCandCap / kBgThreads;  // 8 list entries per thread
    uint2 ent[kPerThread];
    unsigned int pass = 0;     // sparse: bit j = entry j of this thread lies in this sub-tile and is above the threshold
    unsigned int ecnt = 0;     // this thread's labelled pixels on the sub-tile's last row, first or last column
    auto on_edge = [](int r, int c) { return r == kBgRows - 1 || c == 0 || c == kBgT - 1; };
    auto quad = [&](int gy, int gx) -> float4 {  // four pixels of the frame at (gy, gx ..), zeros past the row's end
        float4 v = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        const float *p = img + (int64_t)gy * cols + gx;
        if (gx + 3 < cols) {
            v = ts::load4u(p);
        } else {
            v.x = p[0];
            if (gx + 1 < cols) v.y = p[1];
            if (gx + 2 < cols) v.z = p[2];
        }
        return v;
    };
    if (sparse) {
#pragma unroll
        for (int j = 0; j < kPerThread; ++j) {
            const unsigned int k = (unsigned int)(tid + kBgThreads * j);
            ent[j] = k < n ? cand.ent[(size_t)bgtile * kCandCap + k] : make_uint2(0xffffffffu, 0u);  // (row 0xffffff: in no sub-tile)
        }
        if (tid < kBgRows) {
#pragma unroll
            for (int w = 0; w < 8; ++w) sh.tmask[tid][w] = 0u;
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < kPerThread; ++j) {
            const int rt = (int)(ent[j].x >> 8), c = (int)(ent[j].x & 255u), r = rt - sub * kBgRows;
            const float nv = ab_px(xf, __uint_as_float(ent[j].y));
            if (r >= 0 && r < kBgRows && above(nv, threshold)) {
                atomicOr(&sh.tmask[r][c >> 5], 1u << (c & 31));
                pass |= 1u << j;
                ecnt += on_edge(r, c) ? 1u : 0u;
            }
        }
    } else {
        // thread: columns 4 lane .. 4 lane + 3 of rows wv, wv + 4, ... (a wave reads whole rows); eight lanes make one mask word
#pragma unroll 4
        for (int j = 0; j < kBgRows / 4; ++j) {
            const int r = wv + 4 * j, gy = ty0 + r, gx = tx0 + 4 * lane;
            unsigned int nib = 0;
            if (gy < rows && gx < cols) {
                const float4 v = quad(gy, gx);
                nib = (unsigned int)above(ab_px(xf, v.x), threshold) | ((unsigned int)(gx + 1 < cols && above(ab_px(xf, v.y), threshold)) << 1) |
                      ((unsigned int)(gx + 2 < cols && above(ab_px(xf, v.z), threshold)) << 2) | ((unsigned int)(gx + 3 < cols && above(ab_px(xf, v.w), threshold)) << 3);
            }
            unsigned int ebits = r == kBgRows - 1 ? 15u : 0u;
            if (lane == 0) ebits |= 1u;
            if (lane == 63) ebits |= 8u;
            ecnt += (unsigned int)__builtin_popcount(nib & ebits);
            unsigned int w = nib << (4 * (lane & 7));
            w |= __shfl_xor(w, 1, 64);
            w |= __shfl_xor(w, 2, 64);
            w |= __shfl_xor(w, 4, 64);
            if ((lane & 7) == 0) sh.tmask[r][lane >> 3] = w;
        }
    }
    __syncthreads();
    const unsigned int eat0 = ecnt ? atomicAdd(&sh.n_edge, ecnt) : 0u;
    // ---- row tid (tid < kBgRows): its runs are counted ----
    unsigned int nruns = 0;
    if (tid < kBgRows) {
        const int r = tid;
        unsigned int m[8];
        *reinterpret_cast<uint4 *>(&m[0]) = *reinterpret_cast<const uint4 *>(&sh.tmask[r][0]);
        *reinterpret_cast<uint4 *>(&m[4]) = *reinterpret_cast<const uint4 *>(&sh.tmask[r][4]);
        unsigned char pre[8];
#pragma unroll
        for (int w = 0; w < 8; ++w) {
            const unsigned int st = m[w] & ~((m[w] << 1) | (w ? m[w - 1] >> 31 : 0u));
            pre[w] = (unsigned char)nruns;
            nruns += (unsigned int)__builtin_popcount(st);
        }
        *reinterpret_cast<uint2 *>(&sh.wpre[r][0]) = make_uint2((unsigned int)pre[0] | ((unsigned int)pre[1] << 8) | ((unsigned int)pre[2] << 16) | ((unsigned int)pre[3] << 24),
                                                                 (unsigned int)pre[4] | ((unsigned int)pre[5] << 8) | ((unsigned int)pre[6] << 16) | ((unsigned int)pre[7] << 24));
    }
    // exclusive scan of the rows' run counts over the workgroup
    unsigned int incl = nruns;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const unsigned int o = __shfl_up(incl, off, 64);
        if (lane >= off) incl += o;
    }
    if (lane == 63) sh.wave_tot[wv] = incl;
    __syncthreads();
    unsigned int before = 0;
#pragma unroll
    for (int i = 0; i < kBgThreads / 64; ++i) before += i < wv ? sh.wave_tot[i] : 0u;
    const unsigned int my_base = before + incl - nruns;
    if (tid < kBgRows) sh.row_base[tid] = my_base;
    const unsigned int total_runs = sh.wave_tot[0] + sh.wave_tot[1] + sh.wave_tot[2] + sh.wave_tot[3];
    const bool too_many = total_runs > (unsigned int)kBgRunCap;  // (block-uniform) more runs than the forest holds: the host redoes the frame in full
    if (tid < kBgRows) {
        // the sub-tile's mask words (row tid).  An overflowed sub-tile writes ZEROS and nothing else: its neighbours' border pixels must not
        // be united with pixels whose parents were never written (the frame's results are discarded, its kernels still run)
        const int gy = ty0 + tid;
        if (gy < rows) {
#pragma unroll
            for (int w = 0; w < 8; ++w)
                if (tx0 + 32 * w < mpitch) mask[((int64_t)gy * mp + tx0 + 32 * w) >> 5] = too_many ? 0u : sh.tmask[tid][w];
        }
    }
    if (too_many) {
        if (tid == 0) atomicOr(ro.flags, 1u);
        return;
    }
    if (tid == 0) {
        sh.row_base[kBgRows] = total_runs;
        sh.base_edge = sh.n_edge ? atomicAdd(nborder, sh.n_edge) : 0u;
    }
    // for every run [s0, e0] of row r, in raster order: f(id, s0, e0)
    auto for_runs_of_row = [&](int r, unsigned int id0, auto &&f) {
        unsigned int id = id0;
#pragma unroll
        for (int w = 0; w < 8; ++w) {
            unsigned int st = bg_start_word(sh.tmask, r, w);
            const unsigned int mw = sh.tmask[r][w];
            while (st) {
                const int b = __builtin_ctz(st);
                st &= st - 1;
                const int s0 = 32 * w + b;
                const unsigned int z = b == 31 ? 0u : ((~mw) >> (b + 1));  // zeros after column s0 inside the word
                int e0;
                if (z) {
                    e0 = s0 + __builtin_ctz(z);
                } else {
                    int w2 = w + 1;
                    while (w2 < 8 && sh.tmask[r][w2] == 0xffffffffu) ++w2;
                    e0 = w2 == 8 ? kBgT - 1 : 32 * w2 + __builtin_ctz(~sh.tmask[r][w2]) - 1;
                }
                f(id, s0, e0);
                ++id;
            }
        }
    };
    // the runs of row tid: a node of their own
    if (tid < kBgRows)
        for (unsigned int id = my_base; id < my_base + nruns; ++id) sh.lab[id] = (int)id;
    __syncthreads();
    // ---- unions: every run of row tid with the runs of the next row that touch [s - 1, e + 1] ----
    if (tid + 1 < kBgRows && nruns) {
        const int r = tid;
        const unsigned int nb = sh.row_base[r + 1];
        for_runs_of_row(r, my_base, [&](unsigned int id, int s0, int e0) {
            const int lo = s0 > 0 ? s0 - 1 : 0, hi = e0 < kBgT - 1 ? e0 + 1 : kBgT - 1;
            const unsigned int lo_set = (sh.tmask[r + 1][lo >> 5] >> (lo & 31)) & 1u;
            const unsigned int first = bg_rank_incl(sh, r + 1, lo) - lo_set, last = bg_rank_incl(sh, r + 1, hi);
            for (unsigned int k = first; k < last; ++k) lds_union(sh.lab, (int)id, (int)(nb + k));
        });
    }
    __syncthreads();
    // ---- every run's node -> its root; a root takes a record slot, leaves its pixel's global index there and marks itself with
    // -1 - slot (walkers stop at a mark) ----
    if (tid < kBgRows && nruns) {
        for_runs_of_row(tid, my_base, [&](unsigned int id, int s0, int) {
            int x = (int)id;
            while (true) {
                const int pnt = __hip_atomic_load(&sh.lab[x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                if (pnt < 0 || pnt == x) break;
                x = pnt;
            }
            if (x == (int)id) {  // (only a node's owner marks it)
                const unsigned int slot = atomicAdd(&sh.n_slots, 1u);
                if (slot < (unsigned int)kBgSlots) {
                    sh.acc_i[0][slot] = 0;
                    sh.acc_i[1][slot] = 0x7fffffff;
                    sh.acc_i[2][slot] = -1;
                    sh.acc_i[3][slot] = 0x7fffffff;
                    sh.acc_i[4][slot] = -1;
                    sh.acc_i[5][slot] = 0x7fffffff;
                    sh.acc_i[6][slot] = (ty0 + tid) * cols + tx0 + s0;
                    sh.acc_flux[slot] = 0.0;
                }
                __hip_atomic_store(&sh.lab[id], -1 - (int)slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            } else {
                __hip_atomic_store(&sh.lab[id], x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
        });
    }
    __syncthreads();
    if (tid == 0) {
        const unsigned int ns = sh.n_slots;
        sh.base_rec = ns ? atomicAdd(lcnt + (2 * kRegions + rec_region) * kRegionPitch, ns < (unsigned int)kBgSlots ? ns : (unsigned int)kBgSlots) : 0u;
        if (ns > (unsigned int)kBgSlots) atomicOr(ro.flags, 1u);
    }
    // ---- every labelled pixel: its parent, one contribution to its component's slot, the border list ----
    unsigned int eat = sh.base_edge + eat0;
    auto contribute = [&](int r, int c, float nv) {
        const unsigned int id = sh.row_base[r] + bg_rank_incl(sh, r, c) - 1u;
        const int pnt = sh.lab[id];
        const int slot = -1 - (pnt >= 0 ? sh.lab[pnt] : pnt);  // (every root is marked since the barrier)
        const int gy = ty0 + r, gx = tx0 + c, gi = gy * cols + gx;
        if (slot < kBgSlots) {
            parent[gi] = sh.acc_i[6][slot];
            atomicAdd(&sh.acc_i[0][slot], 1);
            atomicMin(&sh.acc_i[1][slot], gx);
            atomicMax(&sh.acc_i[2][slot], gx);
            atomicMin(&sh.acc_i[3][slot], gy);
            atomicMax(&sh.acc_i[4][slot], gy);
            // BFS seeds are interior (star_detection.rs:107-110)
            if (gy >= 1 && gy < rows - 1 && gx >= 1 && gx <= cols - 2) atomicMin(&sh.acc_i[5][slot], gi);
            const double fl = fmax((double)nv - ro.bg_median, 0.0);
            if (fl > 0.0) unsafeAtomicAdd(&sh.acc_flux[slot], fl);
        } else {
            parent[gi] = gi;  // (more components than slots: the frame is redone in full; the forest the other kernels walk stays valid)
        }
        if (on_edge(r, c)) blist[eat++] = gi;
    };
    if (sparse) {
#pragma unroll
        for (int j = 0; j < kPerThread; ++j)
            if ((pass >> j) & 1u) contribute((int)(ent[j].x >> 8) - sub * kBgRows, (int)(ent[j].x & 255u), ab_px(xf, __uint_as_float(ent[j].y)));
    } else {
        for (int j = 0; j < kBgRows / 4; ++j) {
            const int r = wv + 4 * j, gy = ty0 + r, gx = tx0 + 4 * lane;
            const unsigned int nib = (sh.tmask[r][lane >> 3] >> (4 * (lane & 7))) & 15u;
            if (!nib) continue;
            const float4 v = quad(gy, gx);  // (labelled: inside the frame)
            const float f4[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if ((nib >> k) & 1u) contribute(r, 4 * lane + k, ab_px(xf, f4[k]));
        }
    }
    __syncthreads();
    const unsigned int ns = sh.n_slots < (unsigned int)kBgSlots ? sh.n_slots : (unsigned int)kBgSlots;
    for (unsigned int sl = (unsigned int)tid; sl < ns; sl += kBgThreads) {
        const size_t pos = (size_t)rec_region * ro.stride + sh.base_rec + sl;
        const int groot = sh.acc_i[6][sl];
        ro.st[pos] = CompStat{sh.acc_i[0][sl], sh.acc_i[1][sl], sh.acc_i[2][sl], sh.acc_i[3][sl], sh.acc_i[4][sl], sh.acc_i[5][sl], sh.acc_flux[sl]};
        ro.roots[pos] = groot;
        ro.cid[groot] = (int)pos;
    }
}

// cross-tile unions of the border pixels (the forward neighbours that lie in another tile), on the global forest
__device__ __forceinline__ void label_border_body(int rows, int cols, int *parent, const unsigned int *__restrict__ mask, const int *__restrict__ blist,
                                                  const unsigned int *__restrict__ nborder, unsigned int bid, unsigned int nblk, int mpitch,
                                                  int tile_h = kTileH, int tile_w = kTileW) {
    const unsigned int n = *nborder;
    auto lab_at = [&](int rr, int cc) -> bool {  // (mask rows are mpitch bits apart)
        const int64_t j = (int64_t)rr * mpitch + cc;
        return (mask[j >> 5] >> (j & 31)) & 1u;
    };
    for (unsigned int k = bid * 256 + threadIdx.x; k < n; k += nblk * 256) {
        const int i = blist[k];
        const int r = i / cols, c = i - r * cols;
        const int tr = r / tile_h, tc = c / tile_w;
        auto other = [&](int rr, int cc) { return rr / tile_h != tr || cc / tile_w != tc; };
        if (c + 1 < cols && other(r, c + 1) && lab_at(r, c + 1)) uf_union(parent, i, i + 1);
        if (r + 1 < rows) {
            const int d = i + cols;
            if (c > 0 && other(r + 1, c - 1) && lab_at(r + 1, c - 1)) uf_union(parent, i, d - 1);
            if (other(r + 1, c) && lab_at(r + 1, c)) uf_union(parent, i, d);
            if (c + 1 < cols && other(r + 1, c + 1) && lab_at(r + 1, c + 1)) uf_union(parent, i, d + 1);
        }
    }
}

// ---- per-component statistics and moments -------------------------------------------------------------
// number the component roots (parent[i] == i) among the labelled pixels; one atomic per 1024-thread block and round on the
// tail (one per WAVE serialised on that single counter: ~2500 x 12 ns per frame)
constexpr int kRootsBlock = 256;
__device__ __forceinline__ void roots_body(const int *__restrict__ parent, const int *__restrict__ plist,
                                                            const unsigned int *__restrict__ nlab, int *__restrict__ roots, int *__restrict__ cid,
                                                            unsigned int *nroots, unsigned int cap, CompStat *__restrict__ st = nullptr,
                                                            unsigned int bid = blockIdx.x, unsigned int nblk = gridDim.x) {
    __shared__ unsigned int wave_cnt[kRootsBlock / 64], block_base;
    const unsigned int n = *nlab;
    const unsigned int rounds = (n + nblk * kRootsBlock - 1) / (nblk * kRootsBlock);  // uniform trip count (barriers inside)
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    for (unsigned int it = 0; it < rounds; ++it) {
        const unsigned int k = (it * nblk + bid) * kRootsBlock + threadIdx.x;
        int i = -1;
        bool is = false;
        if (k < n) {
            i = plist[k];
            is = parent[i] == i;
        }
        const unsigned long long m = __ballot(is);
        if (lane == 0) wave_cnt[wv] = (unsigned int)__builtin_popcountll(m);
        __syncthreads();
        if (threadIdx.x == 0) {
            unsigned int tot = 0;
            for (int w = 0; w < kRootsBlock / 64; ++w) tot += wave_cnt[w];
            block_base = tot ? atomicAdd(nroots, tot) : 0u;
        }
        __syncthreads();
        if (is) {
            unsigned int pos = block_base + (unsigned int)__builtin_popcountll(m & ((1ull << lane) - 1ull));
            for (int w = 0; w < wv; ++w) pos += wave_cnt[w];
            if (pos < cap) {
                roots[pos] = i;
                cid[i] = (int)pos;
                if (st) st[pos] = CompStat{0, 0x7fffffff, -1, 0x7fffffff, -1, 0x7fffffff, 0.0};  // (the chain without a host join: no comp_init launch)
            }
        }
        __syncthreads();  // wave_cnt / block_base are rewritten next round
    }
}

__device__ __forceinline__ void comp_init_body(CompStat *st, unsigned int n) {
    const unsigned int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) st[i] = CompStat{0, 0x7fffffff, -1, 0x7fffffff, -1, 0x7fffffff, 0.0};
}

// RECS: fold the records of tile-local components whose root was hooked under another tile's (label_border) into their component's
// record, and point their root pixel straight at the component's root -- a member pixel is then at most two steps from it
// (pixel -> tile root -> root; comp_moments looks twice).  The folded record is left with npix = 0: not a component any more.
__device__ __forceinline__ void comp_merge_body(int *parent, const int *__restrict__ roots, const int *__restrict__ cid, CompStat *st, unsigned int n, size_t seg_base,
                                                unsigned int bid, unsigned int nblk) {
    for (unsigned int j = bid * 256 + threadIdx.x; j < n; j += nblk * 256) {
        const size_t pos = seg_base + j;
        const int groot = roots[pos];
        if (__hip_atomic_load(&parent[groot], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == groot) continue;
        const int root = uf_find(parent, groot);
        __hip_atomic_store(&parent[groot], root, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const CompStat me = st[pos];
        CompStat *t = &st[cid[root]];
        atomicAdd(&t->npix, me.npix);
        atomicMin(&t->x0, me.x0);
        atomicMax(&t->x1, me.x1);
        atomicMin(&t->y0, me.y0);
        atomicMax(&t->y1, me.y1);
        if (me.first_interior != 0x7fffffff) atomicMin(&t->first_interior, me.first_interior);
        if (me.flux > 0.0) unsafeAtomicAdd(&t->flux, me.flux);
        st[pos].npix = 0;
    }
}

// flatten the forest and gather size / bounding box / first interior pixel of every component
template <bool FLUX>
__device__ __forceinline__ void comp_stats_body(int rows, int cols, int *parent, const int *__restrict__ cid, CompStat *st,
                                                         const int *__restrict__ plist, const unsigned int *__restrict__ nlab,
                                                         const float *__restrict__ img = nullptr, int64_t ld = 0, const ab_pixel_xf xf = ab_pixel_xf(),
                                                         double bg_median = 0.0, unsigned int bid = blockIdx.x, unsigned int nblk = gridDim.x) {
    const unsigned int n = *nlab;
    const int lane = threadIdx.x & 63;
    // wave-uniform trip count (the shuffles below need every lane); consecutive list entries are mostly row neighbours
    // of one component, so each RUN of equal roots inside a wave is folded by shuffles and only its first lane issues
    // the six atomics (per-pixel atomics on a component's record serialise: 46 us per frame)
    for (unsigned int k0 = bid * 256 + (threadIdx.x & ~63); k0 < n; k0 += nblk * 256) {
        const unsigned int k = k0 + lane;
        const bool valid = k < n;
        int root = -1 - lane, i = 0;  // invalid lanes: pairwise distinct pseudo-roots, never equal to a real one
        if (valid) {
            i = plist[k];
            root = uf_find(parent, i);
            __hip_atomic_store(&parent[i], root, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        const int r = i / cols, c = i - r * cols;
        const bool interior = r >= 1 && r < rows - 1 && c >= 1 && c < cols - 1;  // BFS seeds are interior (:107-110)
        int npix = 1, x0 = c, x1 = c, y0 = r, y1 = r, fi = interior ? i : 0x7fffffff;
        double fl = 0.0;
        if constexpr (FLUX) fl = valid ? fmax((double)ab_px(xf, img[(int64_t)r * ld + c]) - bg_median, 0.0) : 0.0;
        const int prev = __shfl_up(root, 1, 64);
        const bool head = lane == 0 || prev != root;
        // run length to the right of every lane, by doubling: a lane folds in its right neighbour block only while that
        // block still belongs to the same run (run id = number of heads up to the lane)
        const unsigned long long heads = __ballot(head);
        const int run = (int)__builtin_popcountll(heads & ((2ull << lane) - 1ull));
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const int o_run = __shfl_down(run, d, 64), o_n = __shfl_down(npix, d, 64), o_x0 = __shfl_down(x0, d, 64), o_x1 = __shfl_down(x1, d, 64),
                      o_y0 = __shfl_down(y0, d, 64), o_y1 = __shfl_down(y1, d, 64), o_fi = __shfl_down(fi, d, 64);
            double o_fl = 0.0;
            if constexpr (FLUX) o_fl = __shfl_down(fl, d, 64);
            if (lane + d < 64 && o_run == run) {
                npix += o_n;
                x0 = min(x0, o_x0);
                x1 = max(x1, o_x1);
                y0 = min(y0, o_y0);
                y1 = max(y1, o_y1);
                fi = min(fi, o_fi);
                if constexpr (FLUX) fl += o_fl;
            }
        }
        if (valid && head) {
            CompStat *s = &st[cid[root]];
            atomicAdd(&s->npix, npix);
            atomicMin(&s->x0, x0);
            atomicMax(&s->x1, x1);
            atomicMin(&s->y0, y0);
            atomicMax(&s->y1, y1);
            if (fi != 0x7fffffff) atomicMin(&s->first_interior, fi);
            if constexpr (FLUX)
                if (fl > 0.0) unsafeAtomicAdd(&s->flux, fl);  // (global_atomic_add_f64; HBM is fine-grained enough for this: plain hipMalloc memory)
        }
    }
}

__device__ __forceinline__ double wave_sum(double x) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) x += __shfl_xor(x, off, 64);
    return x;
}
__device__ __forceinline__ double wave_max(double x) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) x = fmax(x, __shfl_xor(x, off, 64));
    return x;
}

// one wave per component: flux-weighted moments over the bounding box (star_detection.rs:147-189); lane partials are combined
// by a fixed butterfly, so results are reproducible (the reference accumulates in BFS order: the two agree to ~1e-15 relative;
// the first-order sums are exact in f64 for normalised pixels, hence order-free).
// Round 4: the lanes form a PATCH of pw columns x 64 / pw rows (pw = 8, 16, 32 or 64, the smallest that spans the box) instead
// of one row of 64 columns: a 9 x 9 star was 9 trips with 9 live lanes each, every trip a chain of three dependent loads (mask
// bit -> parent -> pixel), and the whole walk was taken twice (moments about the centroid need the centroid first).  Now the
// three loads of a trip are independent (all taken for every pixel of the box: the mask bit decides afterwards), the first
// four trips are issued together, and boxes of up to eight trips keep their pixel values in registers for the second pass.
// Inside a registration batch the kernel held a hardware queue for 190-260 us per group of four frames (23 us per frame alone:
// pure latency, which grows under load).
constexpr int kMomKeep = 8;
constexpr int kMomPerWave = 4;  // components per wave: their headers and first trips are in flight together
constexpr int kMomFirst = 2;    // trips of each of them issued before anything is consumed
#ifdef AB_DEV_ABLATION
__device__ int g_mom_ablate = 0;  // developer timing experiment (AB_ABLATE_MOMENTS, -DAB_DEV_ABLATION builds only): 1 = no box walk, 2 = no record store either
#endif

struct MomGeom {  // the lane patch of one component's box
    int sh, pw, rpt, dc, dr, nrb, w;
    bool eligible, small;
};
__device__ __forceinline__ MomGeom mom_geom(const CompStat &s, int lane, bool live) {
    MomGeom g;
    g.eligible = live && s.npix >= 3 && s.npix <= 5000 && s.first_interior != 0x7fffffff;  // :142-145
    g.w = s.x1 - s.x0 + 1;
    const int h = s.y1 - s.y0 + 1;
    g.sh = g.w <= 8 ? 3 : (g.w <= 16 ? 4 : (g.w <= 32 ? 5 : 6));
    g.pw = 1 << g.sh;
    g.rpt = 64 >> g.sh;
    g.dc = lane & (g.pw - 1);
    g.dr = lane >> g.sh;
    g.nrb = (h + g.rpt - 1) / g.rpt;  // trips down the box (per column block)
    g.small = g.eligible && g.w <= 64 && g.nrb <= kMomKeep;
    return g;
}

__device__ __forceinline__ void comp_moments_body(const float *__restrict__ img, int cols, int64_t ld, const int *__restrict__ parent,
                                                           const unsigned int *__restrict__ mask, const int *__restrict__ roots, const CompStat *__restrict__ st, unsigned int ncomp,
                                                           double bg_median_arg, const ab_pixel_xf xf_arg, CompRec *__restrict__ rec,
                                                           const FrameDev *__restrict__ fd, const unsigned int *__restrict__ sel = nullptr,
                                                           const unsigned int *__restrict__ nsel = nullptr, bool two_hop = false, int mpitch = 0) {
    const double bg_median = fd ? fd->bg_median : bg_median_arg;
    const ab_pixel_xf xf = fd ? fd->xf : xf_arg;
    const unsigned int base = (blockIdx.x * 4 + (threadIdx.x >> 6)) * kMomPerWave;
    const int lane = threadIdx.x & 63;
    // with a selection (comp_select_many_kernel) the wave's components are sel[base ..] of the *nsel selected ones and their
    // records are written densely in selection order; without, components base .. of all ncomp
    if (sel) ncomp = *nsel;
    if (base >= ncomp) return;
#ifdef AB_DEV_ABLATION
    const int ablate = g_mom_ablate;
#else
    constexpr int ablate = 0;
#endif
    // member pixel's background-subtracted value, 0 for everything else (adding +0.0 changes no sum, max(pk, 0) no peak); the
    // three loads are independent: all are taken for every pixel of the box and the mask bit decides afterwards
    auto value = [&](const CompStat &s, int root, int r, int c, bool valid) -> double {
        const int rr = valid ? r : s.y0, cc = valid ? c : s.x0, idx = rr * cols + cc;  // invalid lanes re-read the box's corner
        const int64_t midx = mpitch ? (int64_t)rr * mpitch + cc : (int64_t)idx;  // (the tiled labelling pads the mask's rows)
        const unsigned int m = mask[midx >> 5];
        const int p = parent[idx];  // (defined at labelled pixels only: the bit decides)
        const float px = img[rr * ld + cc];
        const bool bit = (m >> (midx & 31)) & 1u;
        // (records form: a pixel points at its TILE's root, which points at the component's root when that lies in another tile)
        const int p2 = (two_hop && valid && bit && p != root) ? parent[p] : p;
        const bool member = valid && bit && p2 == root;
        return member ? fmax((double)ab_px(xf, px) - bg_median, 0.0) : 0.0;
    };
    CompStat S[kMomPerWave];
    int root[kMomPerWave];
    MomGeom G[kMomPerWave];
    double v[kMomPerWave][kMomKeep];
#pragma unroll
    for (int j = 0; j < kMomPerWave; ++j) {  // headers: independent loads
        const bool live = base + j < ncomp;
        const unsigned int ci = sel ? sel[live ? base + j : base] : (live ? base + j : base);
        S[j] = st[ci];
        root[j] = roots[ci];
    }
#pragma unroll
    for (int j = 0; j < kMomPerWave; ++j) {  // the first kMomFirst trips of every small box together: 24 loads in flight (a 9 x 9 star
                                             // is two trips of a 16 x 4 patch; four trips each cost 168 registers for the kernel)
        G[j] = mom_geom(S[j], lane, base + j < ncomp);
        const bool go = G[j].small && ablate == 0;
        const int c = S[j].x0 + G[j].dc;
#pragma unroll
        for (int k = 0; k < kMomFirst; ++k) {
            const int r = S[j].y0 + k * G[j].rpt + G[j].dr;
            v[j][k] = value(S[j], root[j], r, c, go && G[j].dc < G[j].w && r <= S[j].y1);
        }
    }
#pragma unroll
    for (int j = 0; j < kMomPerWave; ++j) {
        if (base + j >= ncomp) break;  // wave-uniform
        const CompStat &s = S[j];
        const MomGeom &g = G[j];
        CompRec out;
        out.first_interior = s.first_interior;
        out.npix = s.npix;
        out.sum_flux = out.sum_x = out.sum_y = out.peak = out.sum_r2 = out.sum_xx = out.sum_yy = out.sum_xy = 0.0;
        if (g.eligible && ablate == 0) {
            double f = 0.0, sx = 0.0, sy = 0.0, pk = 0.0;
            if (g.small) {  // wave-uniform; the usual star: its values stay in registers for the second pass
                const int c = s.x0 + g.dc;
                if (g.nrb > kMomFirst) {
#pragma unroll
                    for (int k = kMomFirst; k < 4; ++k) {
                        const int r = s.y0 + k * g.rpt + g.dr;
                        v[j][k] = value(s, root[j], r, c, g.dc < g.w && r <= s.y1);
                    }
                } else {
#pragma unroll
                    for (int k = kMomFirst; k < 4; ++k) v[j][k] = 0.0;
                }
                if (g.nrb > 4) {
#pragma unroll
                    for (int k = 4; k < kMomKeep; ++k) {
                        const int r = s.y0 + k * g.rpt + g.dr;
                        v[j][k] = value(s, root[j], r, c, g.dc < g.w && r <= s.y1);
                    }
                } else {
#pragma unroll
                    for (int k = 4; k < kMomKeep; ++k) v[j][k] = 0.0;
                }
#pragma unroll
                for (int k = 0; k < kMomKeep; ++k) {
                    const int r = s.y0 + k * g.rpt + g.dr;
                    f += v[j][k];
                    sx += (double)c * v[j][k];
                    sy += (double)r * v[j][k];
                    pk = fmax(pk, v[j][k]);
                }
                f = wave_sum(f);
                sx = wave_sum(sx);
                sy = wave_sum(sy);
                pk = wave_max(pk);
                out.sum_flux = f;
                out.sum_x = sx;
                out.sum_y = sy;
                out.peak = pk;
                if (f > 0.0) {
                    const double cx = sx / f, cy = sy / f;
                    double r2 = 0.0, xx = 0.0, yy = 0.0, xy = 0.0;
                    const double dx = (double)c - cx;
#pragma unroll
                    for (int k = 0; k < kMomKeep; ++k) {
                        const double dy = (double)(s.y0 + k * g.rpt + g.dr) - cy;
                        r2 += (dx * dx + dy * dy) * v[j][k];
                        xx += dx * dx * v[j][k];
                        yy += dy * dy * v[j][k];
                        xy += dx * dy * v[j][k];
                    }
                    out.sum_r2 = wave_sum(r2);
                    out.sum_xx = wave_sum(xx);
                    out.sum_yy = wave_sum(yy);
                    out.sum_xy = wave_sum(xy);
                }
            } else {  // a large or wide component: walk the box twice, kMomKeep trips' loads in flight at a time
                const int ncb = (g.w + g.pw - 1) >> g.sh, ntrips = g.nrb * ncb;
                auto trip_rc = [&](int t, int &r, int &c) -> bool {
                    const int rb = t / ncb, cbk = t - rb * ncb;
                    r = s.y0 + rb * g.rpt + g.dr;
                    c = s.x0 + (cbk << g.sh) + g.dc;
                    return t < ntrips && r <= s.y1 && c <= s.x1;
                };
                for (int t0 = 0; t0 < ntrips; t0 += kMomKeep) {
                    double u[kMomKeep];
                    int rr[kMomKeep], cc[kMomKeep];
#pragma unroll
                    for (int k = 0; k < kMomKeep; ++k) {
                        const bool ok = trip_rc(t0 + k, rr[k], cc[k]);
                        u[k] = value(s, root[j], rr[k], cc[k], ok);
                    }
#pragma unroll
                    for (int k = 0; k < kMomKeep; ++k) {
                        f += u[k];
                        sx += (double)cc[k] * u[k];
                        sy += (double)rr[k] * u[k];
                        pk = fmax(pk, u[k]);
                    }
                }
                f = wave_sum(f);
                sx = wave_sum(sx);
                sy = wave_sum(sy);
                pk = wave_max(pk);
                out.sum_flux = f;
                out.sum_x = sx;
                out.sum_y = sy;
                out.peak = pk;
                if (f > 0.0) {
                    const double cx = sx / f, cy = sy / f;
                    double r2 = 0.0, xx = 0.0, yy = 0.0, xy = 0.0;
                    for (int t0 = 0; t0 < ntrips; t0 += kMomKeep) {
                        double u[kMomKeep];
                        int rr[kMomKeep], cc[kMomKeep];
#pragma unroll
                        for (int k = 0; k < kMomKeep; ++k) {
                            const bool ok = trip_rc(t0 + k, rr[k], cc[k]);
                            u[k] = value(s, root[j], rr[k], cc[k], ok);
                        }
#pragma unroll
                        for (int k = 0; k < kMomKeep; ++k) {
                            const double dx = (double)cc[k] - cx, dy = (double)rr[k] - cy;
                            r2 += (dx * dx + dy * dy) * u[k];
                            xx += dx * dx * u[k];
                            yy += dy * dy * u[k];
                            xy += dx * dy * u[k];
                        }
                    }
                    out.sum_r2 = wave_sum(r2);
                    out.sum_xx = wave_sum(xx);
                    out.sum_yy = wave_sum(yy);
                    out.sum_xy = wave_sum(xy);
                }
            }
        }
        // the record goes to pinned HOST memory: one 72-byte store by 18 lanes (every lane holds the wave's sums) instead of
        // five 16-byte stores by lane 0; a component that cannot become a star sends its 8-byte head only (the host reads the
        // sums behind `npix` and `first_interior` in range, finish_stars) -- PCIe writes are what this kernel's duration is made of
        static_assert(sizeof(CompRec) == 72, "18 dwords");
        const double dd[8] = {out.sum_flux, out.sum_x, out.sum_y, out.peak, out.sum_r2, out.sum_xx, out.sum_yy, out.sum_xy};
        unsigned int word = lane == 0 ? (unsigned int)out.first_interior : (unsigned int)out.npix;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const unsigned long long b = (unsigned long long)__double_as_longlong(dd[k]);
            word = lane == 2 + 2 * k ? (unsigned int)b : word;
            word = lane == 3 + 2 * k ? (unsigned int)(b >> 32) : word;
        }
        const bool head_only = !(s.npix >= 3 && s.npix <= 5000 && s.first_interior != 0x7fffffff);
        if (lane < (head_only ? 2 : 18) && ablate != 2) reinterpret_cast<unsigned int *>(&rec[base + j])[lane] = word;
    }
}

// ---- the kernels above as launches: one frame, or a GROUP of frames of one size (blockIdx.y = frame) ------------------------------
// Round 4: inside a registration batch every frame used to bring its own six small launches, two copies and a fill; four hardware
// queues serialise them, so the stage was the sum of ~1000 launch durations over four.  A worker now takes kGroup frames through
// the chain in lockstep: one launch per step for all of them (their blocks side by side), one copy of their counters, one of
// their records.  The per-frame arguments travel by value in DetGroup.
constexpr int kGroupMax = 8;
struct DetGroup {
    int n;
    const float *img[kGroupMax];
    double threshold[kGroupMax], bg_median[kGroupMax];
    ab_pixel_xf xf[kGroupMax];
    int *parent[kGroupMax], *cid[kGroupMax], *plist[kGroupMax], *roots[kGroupMax];
    unsigned int *mask[kGroupMax], *counters[kGroupMax];  // counters[f][0] = components, [1] = labelled pixels
    CompStat *st[kGroupMax];
    CompRec *rec[kGroupMax];
    unsigned int ncomp[kGroupMax];
    unsigned int comp_cap;            // chained form (no host join before the component kernels): capacity of st[f]; ncomp is read on the device
    int chained;
    int *blist[kGroupMax];            // border pixels of the tile-local labelling (label_tile_many_kernel), kRegions segments like plist
    unsigned int *lcnt[kGroupMax];    // its 2 x kRegions list counters, kRegionPitch words apart
    size_t plist_stride, blist_stride;  // ints per region segment
    int tiled;
    int mask_pitch;                   // tiled: bits per row of mask[f] (cols rounded up to 32)
    int recs;                         // label_tile_body<true, true>: st / roots hold one record per TILE-LOCAL component, in kRecRegions segments of rec_stride
    size_t rec_stride;
    unsigned int *sel[kGroupMax];     // indices of the selected components (comp_select_many_kernel), kSelCap each
    unsigned int *selout[kGroupMax];  // PINNED HOST: {selected, candidates} of the frame
    // round 6: the frames' candidate lists (label_bgtile_many_kernel; ent == nullptr: that frame's tiles are read from the frame) and
    // the labelling tile's dimensions (label_border_many_kernel: 32 x 128, or 256 x 256 with the background tiles)
    const uint2 *cand_ent[kGroupMax];
    const unsigned int *cand_cnt[kGroupMax];
    const float *cand_cut[kGroupMax];
    int tile_h, tile_w;
};

// The selection cuts on CompStat::flux -- an f64 sum whose order of additions varies from run to run -- while finish_stars ranks by the
// moments kernel's exact sum: a candidate just below the cut may hold a larger exact flux than a selected one.  The two
// sums differ by a few ulp(f64); a key step (the high word of the f64) is 2^-20 of the flux.  So: when the list was cut
// (candidates > selected) and the faintest star that was KEPT lies within two key steps of the cut, what was left out might have
// been ranked before it and the frame is redone through the full path.  A kept star clear of the cut by two steps is brighter,
// exactly, than everything left out.  (Registration frames: the 120th star is ~4x brighter than the 480th candidate.)
static inline bool selection_cut_too_close(const std::vector<ab_detected_star> &stars, size_t max_keep, unsigned int selected, unsigned int candidates,
                                           unsigned int cut_key) {
    if (candidates <= selected || cut_key == 0 || stars.empty() || stars.size() < max_keep) return false;  // (fewer than max_keep: redone anyway)
    unsigned long long bits;
    const double fl = stars.back().flux;
    memcpy(&bits, &fl, sizeof bits);
    const unsigned long long key = (bits >> 32) + 1ull;
    return key < (unsigned long long)cut_key + 2ull;
}

// ---- the brightest few hundred, chosen on the device ----------------------------------------------------------
// The matcher takes the first 120 stars of the flux-ordered, 3 px-deduplicated list (affine.rs:272-277 after star_detection.rs:
// 215-248); a 4096^2 frame has ~10 000 components.  finish_stars already orders only the brightest 4 x 120 candidates first because
// the dedup never compares a star with a fainter one; this kernel makes the same cut BEFORE the box walks and the records: one
// workgroup per frame finds the kSelKeep-th largest approximate flux (CompStat::flux) among the components that can become stars
// (size, interior seed, positive flux: star_detection.rs:142-152) by an 11 / 11 / 10-bit radix select on the high word of the f64,
// and lists every component at or above it.  comp_moments then walks kSelKeep boxes instead of ten thousand and 35 KB of records
// cross PCIe instead of 750 KB per frame.  The host finishes those exactly as before; if the dedup eats them all before 120
// survive and more candidates exist (crowded or degenerate fields), it re-runs the frame through the full path: same result always.
constexpr unsigned int kSelKeep = 480, kSelCap = 544;
__device__ __forceinline__ unsigned int sel_key(const CompStat &c) {  // 0 = cannot become a star; larger flux -> larger key
    const bool ok = c.npix >= 3 && c.npix <= 5000 && c.first_interior != 0x7fffffff && c.flux > 0.0;
    return ok ? (unsigned int)((unsigned long long)__double_as_longlong(c.flux) >> 32) + 1u : 0u;
}
// 256 threads (a 1024-thread workgroup waits for sixteen free wave slots on one CU while the batch's other kernels hold them: 100 - 170 us
// per launch inside a batch for ~20 us of work) and TWO passes over the component table: pass A histograms a coarse, monotone digit
// of every candidate's key (sel_digit: 2048 bins of 1/16 octave of flux between 2^-64 and 2^64, clamped outside) and finds the bin d0
// that holds the kSelKeep-th brightest; pass B emits everything in a brighter bin straight away and copies the (key, index) pairs of
// bin d0 -- a few hundred -- into LDS, where a radix select on the full key (11 / 11 / 10 bits) finds the cut.  A bin d0 with more
// than kSelSub members (thousands of components within 4 % of the cut) is left out altogether: fewer than kSelKeep are selected
// and the host's rule (candidates remain, fewer than max_keep stars) decides whether the frame is redone in full.
constexpr int kSelThreads = 256, kSelSub = 2048;
__device__ __forceinline__ unsigned int sel_digit(unsigned int k) {  // non-decreasing in k; k != 0
    constexpr int kBase = (1023 - 64) << 20;  // the high word of 2^-64
    const int d = ((int)k - kBase) >> 16;
    return (unsigned int)(d < 0 ? 0 : (d > 2047 ? 2047 : d));
}
__device__ __forceinline__ void sel_find_digit(const unsigned int *hist, int nb, unsigned int want, int tid, unsigned int *digit, unsigned int *above) {
    // one wave walks the digits from the top: the digit that holds the `want`-th largest key and the count strictly above it
    unsigned int run = 0, found = 0, ab = 0;
    bool done = false;
    for (int base = nb - 64; base >= 0 && !done; base -= 64) {
        const unsigned int c = hist[base + 63 - tid];  // lane 0 = the highest digit of the chunk
        unsigned int incl = c;                          // inclusive prefix over lanes 0 .. tid
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const unsigned int o = __shfl_up(incl, off, 64);
            if (tid >= off) incl += o;
        }
        const unsigned long long hit = __ballot(run + incl >= want);
        if (hit) {
            const int l = __builtin_ctzll(hit);
            ab = run + __shfl(incl, l, 64) - __shfl(c, l, 64);
            found = (unsigned int)(base + 63 - l);
            done = true;
        } else {
            run += __shfl(incl, 63, 64);
        }
    }
    *digit = found;
    *above = ab;
}
__global__ __launch_bounds__(kSelThreads) void comp_select_many_kernel(const DetGroup g) { AB_LATENCY_KERNEL_PRIO();
    const int f = blockIdx.x, tid = threadIdx.x;
    unsigned int n = g.ncomp[f];
    if (g.chained && !g.recs) {
        n = g.counters[f][0];
        if (tid == 0) g.selout[f][2] = n;     // the host's only look at the component count
        if (n > g.comp_cap) n = 0;            // table overflow: nothing is selected, the host redoes the frame in full
    }
    const CompStat *__restrict__ st = g.st[f];
    unsigned int *__restrict__ sel = g.sel[f];
    __shared__ unsigned int hist[2048];
    __shared__ unsigned int sub_key[kSelSub], sub_idx[kSelSub];
    __shared__ unsigned int s_ncand, s_d0, s_above, s_count, s_nsub, s_prefix, s_want, s_ge, s_take;
    // the table as segments: one of n entries, or (records form) kRecRegions segments rec_stride apart, each filled to its own count
    __shared__ unsigned int seg_n[kRecRegions], seg_t0[kRecRegions + 1];
    const int nseg = g.recs ? kRecRegions : 1;
    const unsigned int seg_stride = g.recs ? (unsigned int)g.rec_stride : 0u;
    for (unsigned int b = tid; b < 2048; b += kSelThreads) hist[b] = 0;
    if (tid == 0) s_ncand = s_d0 = s_above = s_count = s_nsub = s_prefix = s_want = s_ge = s_take = 0;
    if (g.recs) {
        const unsigned int flagw = g.counters[f][3];   // bit 0: a tile with more components than slots (nothing is selected, the host redoes the frame in
        const bool overflow = (flagw & 1u) != 0;       // full); the rest: 2 x the tiles label_bgtile read from the frame for want of a usable candidate list
        if (tid == 0 && g.chained) g.selout[f][4] = flagw >> 1;
        if (tid < kRecRegions) {
            const unsigned int cnt = g.lcnt[f][(2 * kRegions + tid) * kRegionPitch];
            seg_n[tid] = overflow ? 0u : (cnt < seg_stride ? cnt : seg_stride);
        }
        __syncthreads();
        if (tid == 0) {
            unsigned int t = 0, total = 0;
            for (int sg = 0; sg < kRecRegions; ++sg) {
                seg_t0[sg] = t;
                t += (seg_n[sg] + kSelThreads - 1) / kSelThreads;
                total += seg_n[sg];
            }
            seg_t0[kRecRegions] = t;
            g.selout[f][2] = overflow ? g.comp_cap + 1u : total;  // (records, folded ones included: the host only compares it with the capacity)
        }
    } else if (tid == 0) {
        seg_n[0] = n;
        seg_t0[0] = 0;
        seg_t0[1] = (n + kSelThreads - 1) / kSelThreads;
    }
    __syncthreads();
    const unsigned int trips = seg_t0[nseg];  // block-uniform (the ballots below need whole waves)
    // every entry of the table through consume(key, index), a batch of independent loads at a time (one thread's dependent round
    // trips to L2 were the kernel -- 150 us inside a batch): eight consecutive trips of the one segment, or one trip of all
    // kRecRegions segments at once (the records form: a segment holds two or three trips' worth)
    constexpr unsigned int kSelIlp = 8;
    auto scan = [&](auto &&consume) {
        if (g.recs) {
            unsigned int most = 0;
#pragma unroll
            for (int sg = 0; sg < kRecRegions; ++sg) most = max(most, seg_n[sg]);
            for (unsigned int j = tid; j < most; j += kSelThreads) {  // (block-uniform trip count)
                unsigned int k[kRecRegions];
#pragma unroll
                for (int sg = 0; sg < kRecRegions; ++sg) k[sg] = j < seg_n[sg] ? sel_key(st[(unsigned int)sg * seg_stride + j]) : 0u;
#pragma unroll
                for (int sg = 0; sg < kRecRegions; ++sg) consume(k[sg], (unsigned int)sg * seg_stride + j);
            }
            return;
        }
        for (unsigned int t0 = 0; t0 < trips; t0 += kSelIlp) {
            unsigned int k[kSelIlp];
#pragma unroll
            for (unsigned int u = 0; u < kSelIlp; ++u) {
                const unsigned int i = (t0 + u) * kSelThreads + tid;
                k[u] = i < n ? sel_key(st[i]) : 0u;
            }
#pragma unroll
            for (unsigned int u = 0; u < kSelIlp; ++u) consume(k[u], (t0 + u) * kSelThreads + tid);
        }
    };
    // ---- pass A: level-0 histogram of the candidates ----
    unsigned int mine = 0;
    scan([&](unsigned int k, unsigned int) {
        mine += k ? 1u : 0u;
        if (k) atomicAdd(&hist[sel_digit(k)], 1u);  // (bins of 1/16 octave: a wave's 64 keys rarely share one -- plain LDS atomics;
                                                    // one ballot round per DISTINCT bin, hist_add_matched, was 150 us of this kernel)
    });
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) mine += __shfl_xor(mine, off, 64);
    if ((tid & 63) == 0 && mine) atomicAdd(&s_ncand, mine);
    __syncthreads();
    const unsigned int ncand = s_ncand;
    const bool all = ncand <= kSelKeep;  // block-uniform: select every candidate
    if (!all && tid < 64) {
        unsigned int d0, above;
        sel_find_digit(hist, 2048, kSelKeep, tid, &d0, &above);
        if (tid == 0) {
            s_d0 = d0;
            s_above = above;
            // the whole bin of the cut fits beside what is brighter (the usual case: a bin is 4 % of flux wide): it is taken as it is --
            // a few candidates more than kSelKeep, none of the brightest kSelKeep missing -- and the radix select inside it is skipped
            s_take = above + hist[d0] <= kSelCap ? 1u : 0u;
        }
    }
    __syncthreads();
    const unsigned int d0 = s_d0, above = s_above;
    const bool take_bin = s_take != 0;
    // ---- pass B: emit what is brighter than bin d0, collect bin d0 ----
    scan([&](unsigned int k, unsigned int i) {
        if (!k) return;
        const unsigned int dg = sel_digit(k);
        if (all || dg > d0 || (take_bin && dg == d0)) {
            const unsigned int at = atomicAdd(&s_count, 1u);
            if (at < kSelCap) sel[at] = i;
        } else if (dg == d0) {
            const unsigned int at = atomicAdd(&s_nsub, 1u);
            if (at < (unsigned int)kSelSub) {
                sub_key[at] = k;
                sub_idx[at] = i;
            }
        }
    });
    __syncthreads();
    const unsigned int nsub = s_nsub;
    // the smallest key the selection admits (selout[3]): whatever was left out has an APPROXIMATE key below it.  0: nothing was left out
    constexpr unsigned int kDigitBase = (unsigned int)((1023 - 64) << 20);
    unsigned int cut_key = all ? 0u : (take_bin ? (d0 ? kDigitBase + (d0 << 16) : 1u) : kDigitBase + ((d0 + 1u) << 16));
    if (!all && !take_bin && nsub <= (unsigned int)kSelSub) {  // block-uniform: resolve the remaining 21 bits inside bin d0, in LDS
        const unsigned int want0 = kSelKeep - above;  // >= 1: the kSelKeep-th brightest lies in bin d0
        const unsigned int strips = (nsub + kSelThreads - 1) / kSelThreads;
        unsigned int prefix = 0, mask = 0, want = want0;
        const int shifts[3] = {21, 10, 0}, bits[3] = {11, 11, 10};
        for (int lv = 0; lv < 3; ++lv) {  // radix select of the want0-th largest FULL key among bin d0's members
            const unsigned int nb = 1u << bits[lv];
            for (unsigned int b = tid; b < nb; b += kSelThreads) hist[b] = 0;
            __syncthreads();
            for (unsigned int t = 0; t < strips; ++t) {
                const unsigned int j = t * kSelThreads + tid;
                const unsigned int k = j < nsub ? sub_key[j] : 0u;
                if (j < nsub && (k & mask) == prefix) atomicAdd(&hist[(k >> shifts[lv]) & (nb - 1)], 1u);
            }
            __syncthreads();
            if (tid < 64) {
                unsigned int dg, ab;
                sel_find_digit(hist, (int)nb, want, tid, &dg, &ab);
                if (tid == 0) {
                    s_prefix = prefix | (dg << shifts[lv]);
                    s_want = want - ab;
                }
            }
            __syncthreads();
            prefix = s_prefix;
            want = s_want;
            mask |= (nb - 1) << shifts[lv];
        }
        unsigned int thr = prefix;  // the key of the kSelKeep-th brightest candidate
        unsigned int ge = 0;
        for (unsigned int j = tid; j < nsub; j += kSelThreads) ge += sub_key[j] >= thr ? 1u : 0u;
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) ge += __shfl_xor(ge, off, 64);
        if ((tid & 63) == 0 && ge) atomicAdd(&s_ge, ge);
        __syncthreads();
        if (above + s_ge > kSelCap) thr += 1;  // a crowd of equal keys at the cut: take what is strictly brighter (fewer than kSelKeep)
        cut_key = thr;
        for (unsigned int j = tid; j < nsub; j += kSelThreads) {
            if (sub_key[j] >= thr) {
                const unsigned int at = atomicAdd(&s_count, 1u);
                if (at < kSelCap) sel[at] = sub_idx[j];
            }
        }
    }
    __syncthreads();
    if (tid == 0) {
        const unsigned int m = min(s_count, kSelCap);
        g.counters[f][2] = m;
        g.selout[f][0] = m;
        g.selout[f][3] = cut_key;
        g.selout[f][1] = s_count > kSelCap ? 0xffffffffu : ncand;  // (cannot happen: the cuts above keep it below kSelCap; kept as a loud fallback)
    }
}

__global__ __launch_bounds__(kInitBlock) void label_init_kernel(const float *__restrict__ img, int rows, int cols, int64_t ld, double threshold_arg,
                                                                const ab_pixel_xf xf_arg, int *__restrict__ parent, unsigned int *__restrict__ mask,
                                                                int *__restrict__ plist, unsigned int *nlab, int vec_ok, const FrameDev *__restrict__ fd) { AB_LATENCY_KERNEL_PRIO();
    label_init_body(img, rows, cols, ld, threshold_arg, xf_arg, parent, mask, plist, nlab, vec_ok, fd);
}
__global__ __launch_bounds__(kInitBlock) void label_init_many_kernel(const DetGroup g, int rows, int cols, int64_t ld, int vec_ok) { AB_LATENCY_KERNEL_PRIO();
    const int f = blockIdx.y;
    label_init_body(g.img[f], rows, cols, ld, g.threshold[f], g.xf[f], g.parent[f], g.mask[f], g.plist[f], g.counters[f] + 1, vec_ok, nullptr);
}
__global__ __launch_bounds__(256) void label_merge_kernel(int rows, int cols, int *parent, const unsigned int *__restrict__ mask, const int *__restrict__ plist,
                                                          const unsigned int *__restrict__ nlab) { AB_LATENCY_KERNEL_PRIO();
    label_merge_body(rows, cols, parent, mask, plist, nlab);
}
__global__ __launch_bounds__(256) void label_merge_many_kernel(const DetGroup g, int rows, int cols) { AB_LATENCY_KERNEL_PRIO();
    const int f = blockIdx.y;
    label_merge_body(rows, cols, g.parent[f], g.mask[f], g.plist[f], g.counters[f] + 1);
}
template <bool RUNS, bool RECS>
__global__ __launch_bounds__(kTileThreads) void label_tile_many_kernel(const DetGroup g, int rows, int cols) { AB_LATENCY_KERNEL_PRIO();
    const int f = blockIdx.y;
    TileRecOut ro;
    if constexpr (RECS) ro = TileRecOut{g.st[f], g.roots[f], g.cid[f], g.rec_stride, g.counters[f] + 3, g.bg_median[f]};
    label_tile_body<RUNS, RECS>(g.img[f], rows, cols, g.threshold[f], g.xf[f], g.parent[f], g.mask[f], g.plist[f], g.plist_stride, g.blist[f], g.blist_stride, g.lcnt[f], ro,
                                g.mask_pitch);
}
__global__ __launch_bounds__(kBgThreads) void label_bgtile_many_kernel(const DetGroup g, int rows, int cols) { AB_LATENCY_KERNEL_PRIO();
    __shared__ BgShared sh;
    const int f = blockIdx.y;
    const TileRecOut ro = TileRecOut{g.st[f], g.roots[f], g.cid[f], g.rec_stride, g.counters[f] + 3, g.bg_median[f]};
    label_bgtile_body(sh, g.img[f], rows, cols, g.threshold[f], g.xf[f], g.parent[f], g.mask[f], g.blist[f], g.blist_stride, g.lcnt[f], ro, g.mask_pitch,
                      FrameCandDev{g.cand_ent[f], g.cand_cnt[f], g.cand_cut[f]});
}
// (grid: a multiple of kRecRegions blocks; block b works on record region b mod kRecRegions as sub-block b / kRecRegions)
__global__ __launch_bounds__(256) void comp_merge_many_kernel(const DetGroup g) { AB_LATENCY_KERNEL_PRIO();
    const int f = blockIdx.y, r = blockIdx.x % kRecRegions;
    if (g.counters[f][3] & 1u) return;  // (a tile overflowed its slots: the host redoes the frame in full)
    const unsigned int cnt = g.lcnt[f][(2 * kRegions + r) * kRegionPitch];
    comp_merge_body(g.parent[f], g.roots[f], g.cid[f], g.st[f], cnt < g.rec_stride ? cnt : (unsigned int)g.rec_stride, (size_t)r * g.rec_stride, blockIdx.x / kRecRegions,
                    gridDim.x / kRecRegions);
}
// (grids of the region walkers: a multiple of kRegions blocks; block b works on region b mod kRegions as sub-block b / kRegions)
__global__ __launch_bounds__(256) void label_border_many_kernel(const DetGroup g, int rows, int cols) { AB_LATENCY_KERNEL_PRIO();
    const int f = blockIdx.y, r = blockIdx.x % kRegions;
    label_border_body(rows, cols, g.parent[f], g.mask[f], g.blist[f] + (size_t)r * g.blist_stride, g.lcnt[f] + (kRegions + r) * kRegionPitch, blockIdx.x / kRegions,
                      gridDim.x / kRegions, g.mask_pitch, g.tile_h, g.tile_w);
}
__global__ __launch_bounds__(kRootsBlock) void roots_kernel(const int *__restrict__ parent, const int *__restrict__ plist, const unsigned int *__restrict__ nlab,
                                                            int *__restrict__ roots, int *__restrict__ cid, unsigned int *nroots, unsigned int cap) { AB_LATENCY_KERNEL_PRIO();
    roots_body(parent, plist, nlab, roots, cid, nroots, cap);
}
__global__ __launch_bounds__(kRootsBlock) void roots_many_kernel(const DetGroup g, unsigned int cap) { AB_LATENCY_KERNEL_PRIO();
    const int f = blockIdx.y;
    if (g.tiled) {
        const int r = blockIdx.x % kRegions;
        roots_body(g.parent[f], g.plist[f] + (size_t)r * g.plist_stride, g.lcnt[f] + r * kRegionPitch, g.roots[f], g.cid[f], g.counters[f], cap,
                   g.chained ? g.st[f] : nullptr, blockIdx.x / kRegions, gridDim.x / kRegions);
        return;
    }
    roots_body(g.parent[f], g.plist[f], g.counters[f] + 1, g.roots[f], g.cid[f], g.counters[f], cap, g.chained ? g.st[f] : nullptr);
}
__global__ __launch_bounds__(256) void comp_init_kernel(CompStat *st, unsigned int n) { AB_LATENCY_KERNEL_PRIO(); comp_init_body(st, n); }
__global__ __launch_bounds__(256) void comp_init_many_kernel(const DetGroup g) { AB_LATENCY_KERNEL_PRIO();
    const int f = blockIdx.y;
    comp_init_body(g.st[f], g.ncomp[f]);
}
__global__ __launch_bounds__(256) void comp_stats_kernel(int rows, int cols, int *parent, const int *__restrict__ cid, CompStat *st, const int *__restrict__ plist,
                                                         const unsigned int *__restrict__ nlab) { AB_LATENCY_KERNEL_PRIO();
    comp_stats_body<false>(rows, cols, parent, cid, st, plist, nlab);
}
__global__ __launch_bounds__(256) void comp_stats_many_kernel(const DetGroup g, int rows, int cols, int64_t ld) { AB_LATENCY_KERNEL_PRIO();
    const int f = blockIdx.y;
    if (g.chained) {
        const unsigned int nc = g.counters[f][0];
        if (nc == 0 || nc > g.comp_cap) return;  // (more components than the table holds: the host redoes the frame in full)
    } else if (g.ncomp[f] == 0) {
        return;
    }
    if (g.tiled) {
        const int r = blockIdx.x % kRegions;
        comp_stats_body<true>(rows, cols, g.parent[f], g.cid[f], g.st[f], g.plist[f] + (size_t)r * g.plist_stride, g.lcnt[f] + r * kRegionPitch, g.img[f], ld, g.xf[f],
                              g.bg_median[f], blockIdx.x / kRegions, gridDim.x / kRegions);
        return;
    }
    comp_stats_body<true>(rows, cols, g.parent[f], g.cid[f], g.st[f], g.plist[f], g.counters[f] + 1, g.img[f], ld, g.xf[f], g.bg_median[f]);
}
__global__ __launch_bounds__(256) void comp_moments_kernel(const float *__restrict__ img, int cols, int64_t ld, const int *__restrict__ parent,
                                                           const unsigned int *__restrict__ mask, const int *__restrict__ roots, const CompStat *__restrict__ st,
                                                           unsigned int ncomp, double bg_median_arg, const ab_pixel_xf xf_arg, CompRec *__restrict__ rec,
                                                           const FrameDev *__restrict__ fd) { AB_LATENCY_KERNEL_PRIO();
    comp_moments_body(img, cols, ld, parent, mask, roots, st, ncomp, bg_median_arg, xf_arg, rec, fd);
}
__global__ __launch_bounds__(256) void comp_moments_many_kernel(const DetGroup g, int cols, int64_t ld) { AB_LATENCY_KERNEL_PRIO();
    const int f = blockIdx.y;
    comp_moments_body(g.img[f], cols, ld, g.parent[f], g.mask[f], g.roots[f], g.st[f], g.ncomp[f], g.bg_median[f], g.xf[f], g.rec[f], nullptr,
                      g.sel[f], g.sel[f] ? g.counters[f] + 2 : nullptr, g.recs != 0, g.tiled ? g.mask_pitch : 0);
}

// ---- normalize_for_detection (affine.rs:24-53) ------------------------------------------------------
__global__ __launch_bounds__(256) void subsample_kernel(const float *__restrict__ img, int64_t len, int64_t step,
                                                        float *__restrict__ out, int64_t nout) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < nout) out[i] = img[i * step];
}

__global__ __launch_bounds__(256) void normalize_kernel(const float *__restrict__ img, int64_t len, double lo, double inv_range,
                                                        float *__restrict__ out) {
    const int64_t stride = (int64_t)gridDim.x * 256;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < len; i += stride) {
        double t = ((double)img[i] - lo) * inv_range;
        t = t < 0.0 ? 0.0 : (t > 1.0 ? 1.0 : t);  // f64::clamp: NaN stays NaN
        out[i] = (float)t;
    }
}

// normalize_for_detection's parameters (affine.rs:24-47): the 1st and 99.9th percentile of an every-step subsample.
// xf->on = 0 when the reference returns image.clone() (too few finite samples / flat range).
// The 1 % / 99.9 % order statistics of the finite subsample values (affine.rs:33-42 sorts the subsample; only these two
// elements of the sorted list are read), by ONE 1024-thread workgroup: an 11 / 11 / 10-bit radix select on the monotone image
// of the float bit patterns (any sign), both ranks descending together -- three sweeps over ~100 000 values in L2.  The first
// version copied the subsample to the host and ran std::nth_element twice: 0.7 .. 1.2 ms of a worker thread per frame, the
// largest single item of the registration stage's per-frame host time.
struct PercentileOut {
    float lo, hi;
    unsigned int finite;  // count of finite subsample values
    unsigned int pad;
};
__device__ __forceinline__ uint32_t ordered_key(float v) {  // monotone for every finite float (-0.0 sorts just below +0.0)
    const uint32_t b = __float_as_uint(v);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__device__ __forceinline__ float ordered_value(uint32_t k) { return __uint_as_float((k & 0x80000000u) ? (k & 0x7fffffffu) : ~k); }

// histogram update with the wave's first bin tallied by ballot: sky pixels share their top bits, and 64 lanes adding to one LDS
// address serialise
__device__ __forceinline__ void tally(unsigned int *hist, bool on, uint32_t bin) {
    const unsigned long long act = __builtin_amdgcn_ballot_w64(on);
    if (!act) return;
    const uint32_t mode = (uint32_t)__builtin_amdgcn_readlane((int)bin, (int)__builtin_ctzll(act));
    const unsigned long long same = __builtin_amdgcn_ballot_w64(on && bin == mode);
    if ((threadIdx.x & 63) == (int)__builtin_ctzll(act)) atomicAdd(&hist[mode], (unsigned int)__builtin_popcountll(same));
    if (on && bin != mode) atomicAdd(&hist[bin], 1u);
}

// the bin of `hist[0 .. 2048)` that holds 0-based rank r, and r's rank inside it; every thread gets the answer (block of 1024)
__device__ __forceinline__ void find_rank_2048(const unsigned int *hist, unsigned int *wave_tot /* 16 */, unsigned int *bcast /* 2 */,
                                               unsigned int r, unsigned int *bin, unsigned int *within) {
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
    const unsigned int h0 = hist[2 * t], h1 = hist[2 * t + 1], mine = h0 + h1;
    unsigned int inc = mine;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const unsigned int u = __shfl_up(inc, o, 64);
        if (lane >= o) inc += u;
    }
    __syncthreads();  // (wave_tot / bcast may still be read from the previous call)
    if (lane == 63) wave_tot[wv] = inc;
    __syncthreads();
    unsigned int base = 0;
    for (int i = 0; i < wv; ++i) base += wave_tot[i];
    const unsigned int excl = base + inc - mine;
    if (r >= excl && r < excl + mine) {  // exactly one thread
        const bool second = r - excl >= h0;
        bcast[0] = 2u * t + (second ? 1u : 0u);
        bcast[1] = r - excl - (second ? h0 : 0u);
    }
    __syncthreads();
    *bin = bcast[0];
    *within = bcast[1];
}

// SRC supplies sample i; the sweeps are written once for both sources below
template <class SRC>
__device__ __forceinline__ void percentiles_body(const SRC &src, unsigned int ns, PercentileOut *__restrict__ out, FrameDev *__restrict__ fd,
                                                 ab_pixel_xf *__restrict__ dxf = nullptr /* the transform itself, into a device table ... */,
                                                 ab_pixel_xf *__restrict__ hxf = nullptr /* ... and into its pinned host mirror (fed pipeline) */) {
    __shared__ unsigned int hist[2][2048];
    __shared__ unsigned int wave_tot[16], bcast[2];
    const int t = threadIdx.x;
    for (int i = t; i < 2 * 2048; i += 1024) (&hist[0][0])[i] = 0;
    __syncthreads();
    // level 0: key bits 31..21 of every finite value; the total is the finite count
    src.for_each(ns, [&](float v) {
        const bool ok = fabsf(v) <= 3.4028234663852886e38f;  // finite (NaN fails)
        tally(hist[0], ok, ordered_key(v) >> 21);
    });
    __syncthreads();
    unsigned int m = 0;
    {
        const unsigned int mine = hist[0][2 * t] + hist[0][2 * t + 1];
        unsigned int x = mine;
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) x += __shfl_xor(x, o, 64);
        if ((t & 63) == 0) wave_tot[t >> 6] = x;
        __syncthreads();
        for (int i = 0; i < 16; ++i) m += wave_tot[i];
    }
    if (m < 100u) {  // affine.rs:34-36: too few samples, the frame is used as it is
        if (t == 0) {
            *out = PercentileOut{0.0f, 0.0f, m, 0u};
            if (fd) {
                fd->xf = ab_pixel_xf();
                fd->finite = m;
            }
            if (dxf) *dxf = ab_pixel_xf();
            if (hxf) *hxf = ab_pixel_xf();
        }
        return;
    }
    unsigned int r[2] = {m / 100u, (unsigned int)((uint64_t)m * 999u / 1000u)};  // affine.rs:41-42
    uint32_t prefix[2];
    {
        unsigned int b0, w0, b1, w1;
        find_rank_2048(hist[0], wave_tot, bcast, r[0], &b0, &w0);
        find_rank_2048(hist[0], wave_tot, bcast, r[1], &b1, &w1);
        prefix[0] = b0 << 21;
        prefix[1] = b1 << 21;
        r[0] = w0;
        r[1] = w1;
    }
    // levels 1 (bits 20..10) and 2 (bits 9..0): both ranks in the same sweep, one histogram each
    for (int level = 1; level <= 2; ++level) {
        const int shift = level == 1 ? 10 : 0;
        const uint32_t digit_mask = level == 1 ? 2047u : 1023u, prefix_mask = level == 1 ? 0xffe00000u : 0xfffffc00u;
        __syncthreads();
        for (int i = t; i < 2 * 2048; i += 1024) (&hist[0][0])[i] = 0;
        __syncthreads();
        const uint32_t p0 = prefix[0], p1 = prefix[1];
        src.for_each(ns, [&](float v) {
            const bool ok = fabsf(v) <= 3.4028234663852886e38f;
            const uint32_t k = ordered_key(v), d = (k >> shift) & digit_mask;
            if (ok && (k & prefix_mask) == p0) atomicAdd(&hist[0][d], 1u);
            if (ok && (k & prefix_mask) == p1) atomicAdd(&hist[1][d], 1u);
        });
        __syncthreads();
        for (int q = 0; q < 2; ++q) {
            unsigned int b, w;
            find_rank_2048(hist[q], wave_tot, bcast, r[q], &b, &w);
            prefix[q] |= b << shift;
            r[q] = w;
        }
    }
    if (t == 0) {
        const float lo = ordered_value(prefix[0]), hi = ordered_value(prefix[1]);
        *out = PercentileOut{lo, hi, m, 0u};
        if (fd || dxf || hxf) {  // the host's arithmetic (xf_from_percentiles), on the device
            ab_pixel_xf xf;
            const double range = (double)hi - (double)lo;
            if (!(range < 1e-15)) {
                xf.lo = (double)lo;
                xf.inv = 1.0 / range;
                xf.on = 1;
            }
            if (fd) {
                fd->xf = xf;
                fd->finite = m;
            }
            if (dxf) *dxf = xf;
            if (hxf) *hxf = xf;
        }
    }
}

// the subsample held in registers: <= kPctPer values per thread, all loads in flight at once; the three sweeps then read
// registers.  (Sweeping the buffer in memory cost one dependent L2 round trip per value and sweep -- the LDS atomics keep the
// compiler from pipelining the loads: 92 us for the workgroup.  Gathering every `step`-th pixel of the plane in this kernel,
// without subsample_kernel, was worse still: 100 000 scattered cache lines through ONE compute unit, 0.3 ms.)
constexpr int kPctPer = 100;
struct RegSample {
    float v[kPctPer];
    template <class F>
    __device__ __forceinline__ void for_each(unsigned int ns, F f) const {
#pragma unroll
        for (int j = 0; j < kPctPer; ++j)
            if ((unsigned int)(j * 1024) < ns) f(v[j]);  // block-uniform; slots past ns hold NaN
    }
};
__global__ __launch_bounds__(1024) void percentiles_reg_kernel(const float *__restrict__ sub, unsigned int ns, PercentileOut *__restrict__ out,
                                                                FrameDev *__restrict__ fd) {
    RegSample s;
#pragma unroll
    for (int j = 0; j < kPctPer; ++j) {
        const unsigned int i = (unsigned int)(j * 1024) + threadIdx.x;
        s.v[j] = i < ns ? sub[i] : __builtin_nanf("");
    }
    percentiles_body(s, ns, out, fd);
}

// larger subsamples (planes of 100 000 .. 200 000 pixels are sampled at step 1): swept from a buffer
struct MemSample {
    const float *s;
    template <class F>
    __device__ __forceinline__ void for_each(unsigned int ns, F f) const {
        for (unsigned int i0 = 0; i0 < ns; i0 += 1024) {
            const unsigned int i = i0 + threadIdx.x;
            f(i < ns ? s[i] : __builtin_nanf(""));
        }
    }
};
__global__ __launch_bounds__(1024) void percentiles_mem_kernel(const float *__restrict__ sub, unsigned int ns, PercentileOut *__restrict__ out,
                                                                FrameDev *__restrict__ fd) {
    percentiles_body(MemSample{sub}, ns, out, fd);
}

// The same for MANY planes of one size in two launches (the registration batch: the percentiles of all 64 frames before any
// worker starts).  One percentile workgroup per plane -- 64 of them side by side take what one takes, where each used to sit in
// its frame's chain: 60 .. 100 us of ONE compute unit during which the three other streams of its hardware queue waited.
constexpr int kManyPlanes = 128;
struct PlaneList {
    const float *p[kManyPlanes];
};
__global__ __launch_bounds__(256) void subsample_many_kernel(const PlaneList pl, int64_t len, int64_t step, float *__restrict__ out, int64_t nout) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < nout) out[(int64_t)blockIdx.y * nout + i] = pl.p[blockIdx.y][i * step];
}
__global__ __launch_bounds__(1024) void percentiles_many_reg_kernel(const float *__restrict__ sub, unsigned int ns, PercentileOut *__restrict__ out,
                                                                     ab_pixel_xf *__restrict__ dxf, ab_pixel_xf *__restrict__ hxf) {
    const float *mine = sub + (size_t)blockIdx.x * ns;
    RegSample s;
#pragma unroll
    for (int j = 0; j < kPctPer; ++j) {
        const unsigned int i = (unsigned int)(j * 1024) + threadIdx.x;
        s.v[j] = i < ns ? mine[i] : __builtin_nanf("");
    }
    percentiles_body(s, ns, out + blockIdx.x, (FrameDev *)nullptr, dxf ? dxf + blockIdx.x : nullptr, hxf ? hxf + blockIdx.x : nullptr);
}
__global__ __launch_bounds__(1024) void percentiles_many_mem_kernel(const float *__restrict__ sub, unsigned int ns, PercentileOut *__restrict__ out,
                                                                     ab_pixel_xf *__restrict__ dxf, ab_pixel_xf *__restrict__ hxf) {
    percentiles_body(MemSample{sub + (size_t)blockIdx.x * ns}, ns, out + blockIdx.x, (FrameDev *)nullptr, dxf ? dxf + blockIdx.x : nullptr,
                     hxf ? hxf + blockIdx.x : nullptr);
}


// estimate_background's reduction of the tiles (star_detection.rs:70-83) on the device: the upper median of the valid tiles'
// medians and of their sigmas (sorted[len / 2]; any order of equal values gives the same element), then detect_stars'
// threshold (:103).  One workgroup; a tile's rank is counted against all others (<= kBgTiles tiles: 256 for a 4096^2 frame).
constexpr int kBgTiles = 2048;
struct BgOut {
    double bg_median, bg_sigma;
};
__global__ __launch_bounds__(1024) void bg_threshold_kernel(const TileOut *__restrict__ tiles, int ntiles, double sigma_threshold,
                                                            FrameDev *__restrict__ fd, BgOut *__restrict__ out) {
    __shared__ double med[kBgTiles], sig[kBgTiles];
    __shared__ unsigned int nv_s;
    __shared__ double res[2];
    if (threadIdx.x == 0) {
        nv_s = 0;
        res[0] = 0.0;  // no valid tile: (0, 1) (:70-72)
        res[1] = 1.0;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < ntiles; i += 1024) {
        const TileOut t = tiles[i];
        if (t.valid) {
            const unsigned int at = atomicAdd(&nv_s, 1u);
            med[at] = t.median;
            sig[at] = t.sigma;
        }
    }
    __syncthreads();
    const unsigned int nv = nv_s, want = nv / 2;
    for (unsigned int i = threadIdx.x; i < nv; i += 1024) {
        const double mi = med[i], si = sig[i];
        unsigned int rm = 0, rs = 0;
        for (unsigned int j = 0; j < nv; ++j) {
            const double mj = med[j], sj = sig[j];
            rm += (mj < mi || (mj == mi && j < i)) ? 1u : 0u;
            rs += (sj < si || (sj == si && j < i)) ? 1u : 0u;
        }
        if (rm == want) res[0] = mi;
        if (rs == want) res[1] = si;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        const double bg_median = res[0], bg_sigma = nv ? fmax(res[1], 1e-10) : 1.0;  // :81
        fd->bg_median = bg_median;
        fd->bg_sigma = bg_sigma;
        fd->threshold = bg_median + sigma_threshold * bg_sigma;  // :103
        out->bg_median = bg_median;
        out->bg_sigma = bg_sigma;
    }
}

// affine.rs:34-47 from the two order statistics: unchanged frame for < 100 finite samples or a flat range
ab_pixel_xf xf_from_percentiles(const PercentileOut &po) {
    ab_pixel_xf xf;
    if (po.finite < 100) return xf;
    const double lo = (double)po.lo, hi = (double)po.hi;
    const double range = hi - lo;
    if (range < 1e-15) return xf;
    xf.lo = lo;
    xf.inv = 1.0 / range;
    xf.on = 1;
    return xf;
}

int f64_cmp(double a, double b) {  // math/median.rs:15-25
    if (a < b) return -1;
    if (a > b) return 1;
    if (a == b) return 0;
    const bool an = std::isnan(a), bn = std::isnan(b);
    return an && bn ? 0 : (an ? 1 : -1);
}

}  // namespace

// {count, finished blocks, ids[tiles]} of the tiles the streaming kernel declines, one per stream the tile kernels run on
static int tile_fail_buffer(ab_ctx *ctx, int which, size_t tiles, unsigned int **out) {
    if (tiles + 2 > ctx->tile_fail_cap[which]) {
        if (ctx->tile_fail[which]) {
            AB_HIP(ctx, hipDeviceSynchronize());
            AB_HIP(ctx, hipFree(ctx->tile_fail[which]));
            ctx->tile_fail[which] = nullptr;
            ctx->tile_fail_cap[which] = 0;
        }
        const size_t cap = tiles + 2 + (tiles >> 1);
        AB_HIP(ctx, hipMalloc((void **)&ctx->tile_fail[which], cap * sizeof(unsigned int)));
        // once: the kernels keep it zeroed.  hipMemset on device memory is NOT ordered with the context's non-blocking streams and may
        // return before it has run (round 6: a fresh worker context's first tile launch could append its declined tiles to a list
        // that the fill then wiped -- a stale tile statistic in 1 of ~10 first calls, tests/test_gpu_subframe.py): fill, then wait
        AB_HIP(ctx, hipMemset(ctx->tile_fail[which], 0, 2 * sizeof(unsigned int)));
        AB_HIP(ctx, hipDeviceSynchronize());
        ctx->tile_fail_cap[which] = cap;
    }
    *out = ctx->tile_fail[which];
    return AB_OK;
}

// The per-tile statistics of `nplanes` planes (grid: tiles x planes) on `stream`: the streaming kernel, then the resident one
// over whatever it declined (a launch of a few hundred blocks that find an empty list and leave).  AB_TILE_RESIDENT=1: the
// resident kernel alone (round 2 / 3's arrangement).  which: 0 = the context's stream, 1 = its auxiliary stream.
static int launch_tile_kernels(ab_ctx *ctx, hipStream_t stream, int which, const float *img, int64_t rows, int64_t cols, int64_t ld, int step, int ntx,
                               int ntiles, int nplanes, const ab_pixel_xf &xf, TileOut *out, const FrameDev *fd, const float *const *many_planes,
                               const ab_pixel_xf *many_xf, const TileCand cand = TileCand()) {
    static const bool resident = ab_dev_env("AB_TILE_RESIDENT") != nullptr;
    if (resident) {
        hipLaunchKernelGGL(tile_background_bucket_kernel, dim3((unsigned)ntiles, (unsigned)nplanes), dim3(tb::kThreads), 0, stream, img, (int)rows, (int)cols,
                           ld, step, ntx, xf, out, fd, many_planes, many_xf, (unsigned int *)nullptr, 0);
        return AB_OK;
    }
    unsigned int *fail = nullptr;
    AB_TRY(tile_fail_buffer(ctx, which, (size_t)ntiles * (size_t)nplanes, &fail));
    // AB_TILE_PAD_KB (developer knob): unused dynamic LDS per tile workgroup.  Four of them take 156 of a CU's 160 KB, so no kernel that
    // needs LDS (the tile labelling, the votes, the selection) runs beside a tile launch; any padding leaves three and 43 KB free.
    static const unsigned int tile_pad = ab_dev_env("AB_TILE_PAD_KB") ? (unsigned int)std::min(std::max(atoi(ab_dev_env("AB_TILE_PAD_KB")), 0), 100) * 1024u : 0u;
    hipLaunchKernelGGL(tile_background_stream_kernel, dim3((unsigned)ntiles, (unsigned)nplanes), dim3(ts::kThreads), tile_pad, stream, img, (int)rows, (int)cols, ld,
                       step, ntx, xf, out, fd, many_planes, many_xf, fail, cand);
    // (the fallback's workgroups need 55 KB of LDS each before they can even look at the -- almost always empty -- list, and inside a batch
    // that room has to be waited for: 256 of them delayed the next tile launch by 16 - 80 us; 64 find it sooner and still work a long list
    // off side by side)
#ifndef AB_TILE_FALLBACK_BLOCKS
#define AB_TILE_FALLBACK_BLOCKS 64
#endif
    const unsigned int blocks = (unsigned int)std::min<int64_t>((int64_t)ntiles * nplanes, AB_TILE_FALLBACK_BLOCKS);
    hipLaunchKernelGGL(tile_background_bucket_kernel, dim3(blocks), dim3(tb::kThreads), 0, stream, img, (int)rows, (int)cols, ld, step, ntx, xf, out, fd,
                       many_planes, many_xf, fail, ntiles);
    return AB_OK;
}

// estimate_background (star_detection.rs:32-84) on a device plane
// per-tile sigma-clipped (median, sigma, valid) of estimate_background's tiling (star_detection.rs:36-68), row-major tiles
static int tile_stats_host(ab_ctx *ctx, const float *img, int64_t rows, int64_t cols, int64_t ld, int64_t tile_size, ab_pixel_xf xf,
                           std::vector<TileOut> *out, int *ntx_out) {
    AB_HIP(ctx, hipSetDevice(ctx->device));
    const int step = (int)std::max<int64_t>(tile_size, 16);
    AB_CHECK(ctx, step <= 256, "background tiles larger than 256 px are not supported (tile_size %lld)", (long long)tile_size);
    const int nty = (int)((rows + step - 1) / step), ntx = (int)((cols + step - 1) / step);
    const int ntiles = nty * ntx;
    // small results go straight into the context's pinned host buffer (device-visible): no copy command, no bounce
    // through the runtime's staging pages for a pageable destination, just the stream sync
    void *pin = nullptr;
    AB_TRY(ab_pinned(ctx, (size_t)ntiles * sizeof(TileOut), &pin));
    static const bool legacy = ab_dev_env("AB_TILE_LEGACY") != nullptr;
    if (legacy)
        hipLaunchKernelGGL(tile_background_kernel, dim3(ntiles), dim3(absel::kBlock), 0, ctx->stream, img, (int)rows, (int)cols, ld, step,
                           ntx, xf, (TileOut *)pin);
    else
        AB_TRY(launch_tile_kernels(ctx, ctx->stream, 0, img, rows, cols, ld, step, ntx, ntiles, 1, xf, (TileOut *)pin, nullptr, nullptr, nullptr));
    AB_HIP(ctx, hipGetLastError());
    AB_HIP(ctx, hipStreamSynchronize(ctx->stream));
    out->assign((const TileOut *)pin, (const TileOut *)pin + ntiles);
    if (ntx_out) *ntx_out = ntx;
    return AB_OK;
}

// (median, sigma) of estimate_background's tiles (star_detection.rs:70-83)
static void background_from_tiles(const TileOut *t, int ntiles, double *out_median, double *out_sigma, ab_ctx *count_in = nullptr) {
    std::vector<double> med, sig;
    uint64_t declined = 0;
    for (int i = 0; i < ntiles; ++i) declined += t[i].pad != 0;
    ab_count_fallback(count_in, AB_FB_TILES_DECLINED, declined);
    for (int i = 0; i < ntiles; ++i)
        if (t[i].valid) {
            med.push_back(t[i].median);
            sig.push_back(t[i].sigma);
        }
    if (med.empty()) {  // :70-72
        *out_median = 0.0;
        *out_sigma = 1.0;
        return;
    }
    auto lt = [](double a, double b) { return f64_cmp(a, b) < 0; };
    std::sort(med.begin(), med.end(), lt);
    std::sort(sig.begin(), sig.end(), lt);
    *out_median = med[med.size() / 2];
    *out_sigma = std::fmax(sig[sig.size() / 2], 1e-10);
}

// The candidate lists of a pipeline's planes (TileCand; only for 256-px tiles -- frames of 2048 px and more on the short side -- whose
// tiles the labelling pass can take one for one): n x ntiles x kCandCap entries + counts + cuts in a workspace of the pipeline's context
static_assert(kCandCap == 2048, "ab_common.hpp: ab_bg_pipeline_cand assumes 2048 entries per tile");
static int pipeline_cand_lists(ab_ctx *ctx, size_t n, int ntiles, int step, TileCand *out) {
    *out = TileCand();
    if (step != 256 || n == 0) return AB_OK;
    const size_t tiles = n * (size_t)ntiles, ent_bytes = tiles * (size_t)kCandCap * sizeof(uint2);
    char *base = nullptr;
    AB_TRY(ab_workspace(ctx, AB_WS_DETECT_CAND, ent_bytes + tiles * (sizeof(unsigned int) + sizeof(float)) + 64, (void **)&base));
    out->ent = (uint2 *)base;
    out->cnt = (unsigned int *)(base + ent_bytes);
    out->cut = (float *)(out->cnt + tiles);
    return AB_OK;
}
static inline TileCand cand_at(const TileCand &c, size_t plane, int ntiles) {
    TileCand o;
    if (c.ent) {
        o.ent = c.ent + plane * (size_t)ntiles * (size_t)kCandCap;
        o.cnt = c.cnt + plane * (size_t)ntiles;
        o.cut = c.cut + plane * (size_t)ntiles;
    }
    return o;
}

// estimate_background of the n contiguous planes of a registration batch, each with its own load transform, as a PIPELINE: the
// tile kernel runs on the context's auxiliary stream, `chunk` planes per launch (grid: tiles x planes), an event after every
// launch, results in a pinned buffer of its own; ab_bg_pipeline_get blocks on a plane's event and reduces its tiles.  Inside a
// batch the tile kernel owns every compute unit while it runs; launched per frame from sixteen worker streams it kept pushing
// the other frames' small kernels aside and shared their in-order queues.  (One launch for all 64 frames followed by a
// synchronisation was slower than the per-frame form -- 5.9 ms during which nothing else runs: 19.0 against 18.0 ms for the
// stage; the pipeline measures 17.1.)
int ab_bg_pipeline_begin(ab_ctx *ctx, const float *const *planes, size_t n, int64_t rows, int64_t cols, const ab_pixel_xf *xf, int chunk,
                         ab_bg_pipeline *p, bool want_cand) {
    *p = ab_bg_pipeline();
    static const bool legacy = ab_dev_env("AB_TILE_LEGACY") != nullptr;
    if (legacy || n == 0 || rows < 3 || cols < 3 || chunk < 1) return AB_OK;
    AB_HIP(ctx, hipSetDevice(ctx->device));
    const int64_t m = std::min(rows, cols);
    const int64_t tile_size = std::min<int64_t>(std::max<int64_t>(m / 8, 32), 256);  // detect_stars' choice (:100)
    const int step = (int)std::max<int64_t>(tile_size, 16);
    const int nty = (int)((rows + step - 1) / step), ntx = (int)((cols + step - 1) / step), ntiles = nty * ntx;
    if (!ctx->aux_stream) {
        const hipError_t stream_rc = ab_stream_create_masked(ctx, &ctx->aux_stream, AB_DEV_NAME("AB_TILE_CU_MASK"), AB_DEV_NAME("AB_TILE_PRIO"), 0);  // (outside AB_HIP: its message would carry the developer variables' names)
        AB_HIP(ctx, stream_rc);
    }
    // the first launch holds the reference and the first group's targets only (AB_TILE_FIRST, default 5; 0 = `chunk` like the rest):
    // nothing else can run until a group's tiles are done, and eight frames' tiles are ~300 us of an otherwise idle chip
    static const int first_env = ab_dev_env("AB_TILE_FIRST") ? atoi(ab_dev_env("AB_TILE_FIRST")) : 5;
    const size_t first = (first_env > 0 && first_env < chunk && (size_t)first_env < n) ? (size_t)first_env : 0;
    const size_t nchunks = first ? 1 + (n - first + (size_t)chunk - 1) / (size_t)chunk : (n + (size_t)chunk - 1) / (size_t)chunk;
    while (ctx->aux_events.size() < nchunks) {
        hipEvent_t e;
        AB_HIP(ctx, hipEventCreateWithFlags(&e, hipEventDisableTiming));
        ctx->aux_events.push_back(e);
    }
    // the tile results, and behind them the staging copy of the plane-pointer and transform tables: the callers' tables are locals
    // in pageable memory, and an asynchronous copy out of those is only safe while the runtime happens to stage it synchronously
    const size_t ptr_bytes = n * sizeof(const float *), xf_bytes = n * sizeof(ab_pixel_xf);
    const size_t tiles_bytes = (n * (size_t)ntiles * sizeof(TileOut) + 63) & ~(size_t)63;
    const size_t need = tiles_bytes + ((ptr_bytes + 15) & ~(size_t)15) + xf_bytes;
    if (need > ctx->aux_pinned_bytes) {
        if (ctx->aux_pinned) {
            AB_HIP(ctx, hipStreamSynchronize(ctx->aux_stream));
            AB_HIP(ctx, hipHostFree(ctx->aux_pinned));
            ctx->aux_pinned = nullptr;
            ctx->aux_pinned_bytes = 0;
        }
        AB_HIP(ctx, hipHostMalloc(&ctx->aux_pinned, need, hipHostMallocDefault));
        ctx->aux_pinned_bytes = need;
    }
    char *dv = nullptr;  // device copies of the plane pointers and transforms
    AB_TRY(ab_workspace(ctx, AB_WS_DETECT_DEV, ptr_bytes + xf_bytes + 64, (void **)&dv));
    const float **dplanes = (const float **)dv;
    ab_pixel_xf *dxf = (ab_pixel_xf *)(dv + ((ptr_bytes + 15) & ~(size_t)15));
    TileCand cand;
    if (want_cand) AB_TRY(pipeline_cand_lists(ctx, n, ntiles, step, &cand));
    char *stage = (char *)ctx->aux_pinned + tiles_bytes;
    memcpy(stage, planes, ptr_bytes);
    memcpy(stage + ((ptr_bytes + 15) & ~(size_t)15), xf, xf_bytes);
    // from the first enqueue on, a failure drains the auxiliary stream before it returns: the launches read the caller's frames
    auto enqueue = [&]() -> int {
        AB_HIP(ctx, hipMemcpyAsync(dplanes, stage, ptr_bytes, hipMemcpyHostToDevice, ctx->aux_stream));
        AB_HIP(ctx, hipMemcpyAsync(dxf, stage + ((ptr_bytes + 15) & ~(size_t)15), xf_bytes, hipMemcpyHostToDevice, ctx->aux_stream));
        if (cand.ent) AB_HIP(ctx, hipMemsetAsync(cand.cut, 0xff, n * (size_t)ntiles * sizeof(float), ctx->aux_stream));  // NaN: "no list" until a tile says otherwise
        for (size_t c = 0; c < nchunks; ++c) {
            const size_t begin = first ? (c == 0 ? 0 : first + (c - 1) * (size_t)chunk) : c * (size_t)chunk;
            const size_t cnt = (first && c == 0) ? first : std::min<size_t>((size_t)chunk, n - begin);
            AB_TRY(launch_tile_kernels(ctx, ctx->aux_stream, 1, nullptr, rows, cols, cols, step, ntx, ntiles, (int)cnt, ab_pixel_xf(),
                                       (TileOut *)ctx->aux_pinned + begin * (size_t)ntiles, nullptr, (const float *const *)(dplanes + begin),
                                       (const ab_pixel_xf *)(dxf + begin), cand_at(cand, begin, ntiles)));
            AB_HIP(ctx, hipEventRecord(ctx->aux_events[c], ctx->aux_stream));
        }
        AB_HIP(ctx, hipGetLastError());
        return AB_OK;
    };
    const int rc = enqueue();
    if (rc != AB_OK) {
        (void)hipStreamSynchronize(ctx->aux_stream);
        return rc;
    }
    p->on = true;
    p->tiles = ctx->aux_pinned;
    p->events = ctx->aux_events.data();
    p->ntiles = ntiles;
    p->chunk = chunk;
    p->first = (int)first;
    p->n = n;
    p->cand_ent = cand.ent;
    p->cand_cnt = cand.cnt;
    p->cand_cut = cand.cut;
    p->cand_step = cand.ent ? step : 0;
    return AB_OK;
}

// The same pipeline FED chunk by chunk (round 4): nothing is known about a plane on the host before its chunk has run.  The
// percentiles of normalize_for_detection (subsample + radix select, one workgroup per plane) run per chunk on a third stream and
// leave each plane's transform in a device table (read by the tile kernel) and in a pinned mirror (read by the workers after
// ab_bg_pipeline_get); a chunk's tile launch waits for its percentiles by event.
// `landed` (nullable): one event per plane that is still being written when this returns (an upload from the host in flight).
// A chunk is then enqueued by a FEEDER THREAD once its planes' events have completed on the host.  (Enqueueing every chunk up
// front behind hipStreamWaitEvent was measured first: the runtime multiplexes all streams onto four hardware queues, a wait
// packet at the head of a queue holds back every stream that shares it, and a third of the workers' kernels ran only after the
// LAST frame had landed -- 10 ms of registration behind a 76 ms upload instead of ~2.)
struct FedPlan {  // everything a chunk's launches need, by value (the feeder outlives ab_bg_pipeline_begin_fed)
    ab_ctx *ctx = nullptr;
    std::vector<const float *> planes;
    std::vector<hipEvent_t> landed;
    size_t n = 0, nchunks = 0;
    int chunk = 1, step = 0, ntx = 0, ntiles = 0;
    int64_t rows = 0, cols = 0, len = 0, sstep = 1, ns = 0;
    float *sub = nullptr;
    const float **dplanes = nullptr;
    ab_pixel_xf *dxf = nullptr, *hxf = nullptr;
    PercentileOut *hpo = nullptr;
    TileCand cand;  // the candidate lists of all n planes (pipeline_cand_lists; their cuts are preset to NaN)
    int enqueue(size_t c) const {
        const size_t first = c * (size_t)chunk, cnt = std::min<size_t>((size_t)chunk, n - first);
        PlaneList pl;
        for (size_t i = 0; i < (size_t)kManyPlanes; ++i) pl.p[i] = planes[first + (i < cnt ? i : 0)];
        float *csub = sub + first * (size_t)ns;
        hipLaunchKernelGGL(subsample_many_kernel, dim3((unsigned)((ns + 255) / 256), (unsigned)cnt), dim3(256), 0, ctx->pct_stream, pl, len, sstep, csub, ns);
        if (ns <= (int64_t)kPctPer * 1024)
            hipLaunchKernelGGL(percentiles_many_reg_kernel, dim3((unsigned)cnt), dim3(1024), 0, ctx->pct_stream, (const float *)csub, (unsigned int)ns,
                               hpo + first, dxf + first, hxf + first);
        else
            hipLaunchKernelGGL(percentiles_many_mem_kernel, dim3((unsigned)cnt), dim3(1024), 0, ctx->pct_stream, (const float *)csub, (unsigned int)ns,
                               hpo + first, dxf + first, hxf + first);
        AB_HIP(ctx, hipEventRecord(ctx->pct_events[c], ctx->pct_stream));
        AB_HIP(ctx, hipStreamWaitEvent(ctx->aux_stream, ctx->pct_events[c], 0));
        AB_TRY(launch_tile_kernels(ctx, ctx->aux_stream, 1, nullptr, rows, cols, cols, step, ntx, ntiles, (int)cnt, ab_pixel_xf(),
                                   (TileOut *)ctx->aux_pinned + first * (size_t)ntiles, nullptr, (const float *const *)(dplanes + first),
                                   (const ab_pixel_xf *)(dxf + first), cand_at(cand, first, ntiles)));
        AB_HIP(ctx, hipEventRecord(ctx->aux_events[c], ctx->aux_stream));
        AB_HIP(ctx, hipGetLastError());
        return AB_OK;
    }
};
struct ab_bg_feed_impl {
    FedPlan plan;
    std::mutex m;
    std::condition_variable cv;
    size_t enqueued = 0;  // chunks whose launches are in the streams
    int rc = AB_OK;
    std::string err;             // the feeder's own error message (it never writes the caller's context: ab_tls_error_sink)
    std::atomic<bool> stop{false};  // the batch was abandoned (a worker failed, a cancel): no further chunk is enqueued
    std::thread th;
    void give_up() {  // (nobody should be waiting once the batch is abandoned; if somebody is, it gets an answer)
        {
            std::lock_guard<std::mutex> g(m);
            if (rc == AB_OK) rc = AB_ERR_CANCELLED;
            err = "the batch was abandoned";
        }
        cv.notify_all();
    }
    void run() {
        ab_tls_error_sink = &err;
        (void)hipSetDevice(plan.ctx->device);
        for (size_t c = 0; c < plan.nchunks; ++c) {
            if (stop.load(std::memory_order_acquire)) return give_up();
            int r = AB_OK;
            const size_t first = c * (size_t)plan.chunk, cnt = std::min<size_t>((size_t)plan.chunk, plan.n - first);
            for (size_t i = 0; i < cnt && r == AB_OK; ++i)
                if (plan.landed[first + i] && hipEventSynchronize(plan.landed[first + i]) != hipSuccess)
                    r = ab_set_error(plan.ctx, AB_ERR_HIP, "waiting for an uploaded frame failed");
            ab_upload_trace("chunk landed", (long)c);
            if (stop.load(std::memory_order_acquire)) return give_up();
            if (r == AB_OK) r = plan.enqueue(c);
            ab_upload_trace("chunk enqueued", (long)c);
            {
                std::lock_guard<std::mutex> g(m);
                if (r == AB_OK)
                    enqueued = c + 1;
                else
                    rc = r;
            }
            cv.notify_all();
            if (r != AB_OK) return;
        }
    }
};

int ab_bg_pipeline_begin_fed(ab_ctx *ctx, const float *const *planes, size_t n, int64_t rows, int64_t cols, int chunk, const hipEvent_t *landed,
                             ab_bg_pipeline *p, bool want_cand) {
    *p = ab_bg_pipeline();
    static const bool legacy = ab_dev_env("AB_TILE_LEGACY") != nullptr;
    if (legacy || n == 0 || rows < 3 || cols < 3 || chunk < 1) return AB_OK;
    if (chunk > kManyPlanes) chunk = kManyPlanes;
    AB_HIP(ctx, hipSetDevice(ctx->device));
    FedPlan f;
    f.ctx = ctx;
    f.n = n;
    f.chunk = chunk;
    f.rows = rows;
    f.cols = cols;
    const int64_t m = std::min(rows, cols);
    const int64_t tile_size = std::min<int64_t>(std::max<int64_t>(m / 8, 32), 256);  // detect_stars' choice (:100)
    f.step = (int)std::max<int64_t>(tile_size, 16);
    const int nty = (int)((rows + f.step - 1) / f.step);
    f.ntx = (int)((cols + f.step - 1) / f.step);
    f.ntiles = nty * f.ntx;
    f.len = rows * cols;
    f.sstep = std::max<int64_t>(f.len / 100000, 1);  // affine.rs:28-30
    f.ns = (f.len + f.sstep - 1) / f.sstep;
    if (!ctx->aux_stream) {
        const hipError_t stream_rc = ab_stream_create_masked(ctx, &ctx->aux_stream, AB_DEV_NAME("AB_TILE_CU_MASK"), AB_DEV_NAME("AB_TILE_PRIO"), 0);  // (outside AB_HIP: its message would carry the developer variables' names)
        AB_HIP(ctx, stream_rc);
    }
    if (!ctx->pct_stream) AB_HIP(ctx, hipStreamCreateWithFlags(&ctx->pct_stream, hipStreamNonBlocking));
    f.nchunks = (n + (size_t)chunk - 1) / (size_t)chunk;
    while (ctx->aux_events.size() < f.nchunks) {
        hipEvent_t e;
        AB_HIP(ctx, hipEventCreateWithFlags(&e, hipEventDisableTiming));
        ctx->aux_events.push_back(e);
    }
    while (ctx->pct_events.size() < f.nchunks) {
        hipEvent_t e;
        AB_HIP(ctx, hipEventCreateWithFlags(&e, hipEventDisableTiming));
        ctx->pct_events.push_back(e);
    }
    // pinned: tile results | plane pointers (staging copy) | transforms (the mirror the workers read) | percentile records
    const size_t ptr_bytes = (n * sizeof(const float *) + 15) & ~(size_t)15, xf_bytes = (n * sizeof(ab_pixel_xf) + 15) & ~(size_t)15;
    const size_t tiles_bytes = (n * (size_t)f.ntiles * sizeof(TileOut) + 63) & ~(size_t)63;
    const size_t need = tiles_bytes + ptr_bytes + xf_bytes + n * sizeof(PercentileOut);
    if (need > ctx->aux_pinned_bytes) {
        if (ctx->aux_pinned) {
            AB_HIP(ctx, hipStreamSynchronize(ctx->aux_stream));
            AB_HIP(ctx, hipStreamSynchronize(ctx->pct_stream));
            AB_HIP(ctx, hipHostFree(ctx->aux_pinned));
            ctx->aux_pinned = nullptr;
            ctx->aux_pinned_bytes = 0;
        }
        AB_HIP(ctx, hipHostMalloc(&ctx->aux_pinned, need, hipHostMallocDefault));
        ctx->aux_pinned_bytes = need;
    }
    char *dv = nullptr;  // device: plane pointers | transforms
    AB_TRY(ab_workspace(ctx, AB_WS_PIPE_TABLES, ptr_bytes + xf_bytes + 64, (void **)&dv));
    AB_TRY(ab_workspace(ctx, AB_WS_PIPE_SUBSAMPLE, n * (size_t)f.ns * sizeof(float), (void **)&f.sub));
    unsigned int *fail = nullptr;  // (sized once, here: the feeder's launches must not reallocate it)
    AB_TRY(tile_fail_buffer(ctx, 1, (size_t)f.ntiles * (size_t)chunk, &fail));
    f.dplanes = (const float **)dv;
    f.dxf = (ab_pixel_xf *)(dv + ptr_bytes);
    char *stage = (char *)ctx->aux_pinned + tiles_bytes;
    f.hxf = (ab_pixel_xf *)(stage + ptr_bytes);
    f.hpo = (PercentileOut *)(stage + ptr_bytes + xf_bytes);
    memcpy(stage, planes, n * sizeof(const float *));
    f.planes.assign(planes, planes + n);
    if (landed) f.landed.assign(landed, landed + n);
    AB_HIP(ctx, hipMemcpyAsync(f.dplanes, stage, n * sizeof(const float *), hipMemcpyHostToDevice, ctx->aux_stream));
    if (want_cand) AB_TRY(pipeline_cand_lists(ctx, n, f.ntiles, f.step, &f.cand));
    if (f.cand.ent) AB_HIP(ctx, hipMemsetAsync(f.cand.cut, 0xff, n * (size_t)f.ntiles * sizeof(float), ctx->aux_stream));  // NaN: "no list"
    p->tiles = ctx->aux_pinned;
    p->events = ctx->aux_events.data();
    p->ntiles = f.ntiles;
    p->chunk = chunk;
    p->n = n;
    p->xf_host = f.hxf;
    p->cand_ent = f.cand.ent;
    p->cand_cnt = f.cand.cnt;
    p->cand_cut = f.cand.cut;
    p->cand_step = f.cand.ent ? f.step : 0;
    if (landed) {
        ab_bg_feed_impl *feed = new ab_bg_feed_impl();
        feed->plan = std::move(f);
        p->feed = feed;
        p->on = true;
        feed->th = std::thread([feed] { feed->run(); });
        return AB_OK;
    }
    // frames that are complete already: every chunk enqueued here.  From the first enqueue on, a failure drains both streams
    // before it returns: the launches read the caller's frames
    for (size_t c = 0; c < f.nchunks; ++c) {
        const int rc = f.enqueue(c);
        if (rc != AB_OK) {
            (void)hipStreamSynchronize(ctx->pct_stream);
            (void)hipStreamSynchronize(ctx->aux_stream);
            return rc;
        }
    }
    p->on = true;
    return AB_OK;
}

// joins the feeder (if any): call before the pipeline's streams are drained and before the planes go away
void ab_bg_pipeline_end(ab_bg_pipeline *p) {
    if (p && p->feed) {
        p->feed->stop.store(true, std::memory_order_release);  // (a finished feeder never looks; an abandoned batch stops at the next chunk)
        if (p->feed->th.joinable()) p->feed->th.join();
        delete p->feed;
        p->feed = nullptr;
    }
}

int ab_bg_pipeline_get(ab_ctx *ctx, const ab_bg_pipeline *p, size_t i, double *bg) {
    AB_CHECK(ctx, p && p->on && i < p->n, "background pipeline: no such plane");
    const size_t c = p->first > 0 ? (i < (size_t)p->first ? 0 : 1 + (i - (size_t)p->first) / (size_t)p->chunk) : i / (size_t)p->chunk;
    if (p->feed) {  // the chunk's launches are in the streams only once its planes have landed
        std::unique_lock<std::mutex> g(p->feed->m);
        p->feed->cv.wait(g, [&] { return p->feed->enqueued > c || p->feed->rc != AB_OK; });
        if (p->feed->enqueued <= c) return ab_set_error(ctx, p->feed->rc, "the background pipeline's feeder failed: %s", p->feed->err.c_str());
    }
    AB_HIP(ctx, hipEventSynchronize(p->events[c]));
    ab_upload_trace("tiles ready, plane", (long)i);
    background_from_tiles((const TileOut *)p->tiles + i * (size_t)p->ntiles, p->ntiles, &bg[0], &bg[1], ctx);
    return AB_OK;
}

int ab_estimate_background_device(ab_ctx *ctx, const float *img, int64_t rows, int64_t cols, int64_t ld, int64_t tile_size,
                                  double *out_median, double *out_sigma, ab_pixel_xf xf = ab_pixel_xf()) {
    std::vector<TileOut> h;
    AB_TRY(tile_stats_host(ctx, img, rows, cols, ld, tile_size, xf, &h, nullptr));
    std::vector<double> med, sig;
    uint64_t declined = 0;
    for (const auto &t : h) declined += t.pad != 0;
    ab_count_fallback(ctx, AB_FB_TILES_DECLINED, declined);
    for (const auto &t : h)
        if (t.valid) {
            med.push_back(t.median);
            sig.push_back(t.sigma);
        }
    if (med.empty()) {  // :70-72
        *out_median = 0.0;
        *out_sigma = 1.0;
        return AB_OK;
    }
    auto lt = [](double a, double b) { return f64_cmp(a, b) < 0; };
    std::sort(med.begin(), med.end(), lt);
    std::sort(sig.begin(), sig.end(), lt);
    *out_median = med[med.size() / 2];
    *out_sigma = std::fmax(sig[sig.size() / 2], 1e-10);
    return AB_OK;
}

// The host's share of detect_stars (star_detection.rs:147-248) on the components' records: star parameters in discovery order,
// the stable sort by descending flux, the 3 px dedup, at most max_keep survivors.
static void finish_stars(const CompRec *recs_begin, unsigned int ncomp, double bg_sigma, size_t max_keep, std::vector<ab_detected_star> *stars) {
    const CompRec *recs_end = recs_begin + ncomp;
    // ---- host: the reference finishes the stars in discovery order (ascending first interior pixel = BFS seed order) and then
    // sorts them stably by descending flux (:215).  The two orders are one: (flux descending, first interior pixel ascending).
    struct Cand {
        ab_detected_star s;
        int first;
    };
    std::vector<Cand> cand;
    cand.reserve(ncomp);
    for (const CompRec *pc = recs_begin; pc != recs_end; ++pc) {
        const CompRec *c = pc;
        if (!(c->first_interior != 0x7fffffff && c->npix >= 3 && c->npix <= 5000 && c->sum_flux > 0.0)) continue;
        const double sum_flux = c->sum_flux;
        const double cx = c->sum_x / sum_flux, cy = c->sum_y / sum_flux;
        const double sigma_star = std::sqrt(c->sum_r2 / (2.0 * sum_flux));
        const double fwhm = sigma_star * 2.3548200450309493;
        if (fwhm < 0.5 || fwhm > 30.0) continue;
        const double ixx = c->sum_xx / sum_flux, iyy = c->sum_yy / sum_flux, ixy = c->sum_xy / sum_flux;
        const double trace = ixx + iyy;
        const double det = std::fmax(ixx * iyy - ixy * ixy, 0.0);
        const double disc = std::sqrt(std::fmax((trace * trace / 4.0) - det, 0.0));
        const double l1 = trace / 2.0 + disc, l2 = std::fmax(trace / 2.0 - disc, 0.0);
        double ecc = 0.0;
        if (l1 > 1e-15) {
            ecc = std::sqrt(1.0 - l2 / l1);
            ecc = ecc < 0.0 ? 0.0 : (ecc > 1.0 ? 1.0 : ecc);
        }
        Cand k;
        k.s.x = cx;
        k.s.y = cy;
        k.s.flux = sum_flux;
        k.s.fwhm = fwhm;
        k.s.eccentricity = ecc;
        k.s.peak = c->peak;
        k.s.npix = (uint64_t)c->npix;
        k.s.snr = bg_sigma <= DBL_EPSILON ? 0.0 : c->peak / bg_sigma;  // confidence.rs:3-8
        k.first = c->first_interior;
        cand.push_back(k);
    }
    // the order is taken on 16-byte keys, not on the 80-byte candidates (the mask builders and SPCC keep EVERY star: 10 000 candidates
    // of an 8192^2 plane took 0.45 ms of host time per detection, the GPU idle beside it); the key is total (`first` is unique), so
    // the result is the same sequence
    struct Ord {
        double flux;
        int first;
        uint32_t idx;
    };
    std::vector<Ord> ord(cand.size());
    for (size_t i = 0; i < cand.size(); ++i) ord[i] = Ord{cand[i].s.flux, cand[i].first, (uint32_t)i};
    auto before = [](const Ord &a, const Ord &b) { return a.flux != b.flux ? b.flux < a.flux : a.first < b.first; };
    // a caller that wants only the max_keep brightest survivors of the 3 px dedup (registration: 120) does not need the faint
    // thousands in order: the dedup only ever compares a star with brighter ones, so the brightest 4 max_keep are split off and
    // sorted first, and the rest only if the dedup ate so many that they are needed after all
    size_t sorted_upto = ord.size();
    if (max_keep < ord.size() / 4) {
        sorted_upto = 4 * max_keep;
        std::nth_element(ord.begin(), ord.begin() + sorted_upto, ord.end(), before);
    }
    std::sort(ord.begin(), ord.begin() + sorted_upto, before);
    // dedup within 3 px, comparing only against kept stars in the 3 x 3 neighbourhood of 3 px grid cells (:217-248)
    // (kept stars live in a chained hash table over the 3 px cells: no per-cell allocations)
    size_t nbuckets = 64;
    while (nbuckets < 2 * std::min(ord.size(), std::max<size_t>(4 * std::min(max_keep, ord.size()), 64))) nbuckets <<= 1;
    std::vector<int> head(nbuckets, -1), next(ord.size(), -1);
    std::vector<uint64_t> cell_of(ord.size());
    auto key = [](uint64_t gy, uint64_t gx) { return (gy << 32) | gx; };
    auto bucket = [&](uint64_t k) { return (size_t)((k * 0x9E3779B97F4A7C15ull) >> 32) & (nbuckets - 1); };
    stars->reserve(std::min(ord.size(), max_keep));
    for (size_t i = 0; i < ord.size() && stars->size() < max_keep; ++i) {
        if (i == sorted_upto) {  // the brightest block did not yield max_keep survivors: order the rest too
            std::sort(ord.begin() + sorted_upto, ord.end(), before);
            sorted_upto = ord.size();
        }
        const ab_detected_star &fi = cand[ord[i].idx].s;
        const uint64_t gx = (uint64_t)(fi.x / 3.0), gy = (uint64_t)(fi.y / 3.0);
        bool too_close = false;
        for (uint64_t ny = gy ? gy - 1 : 0; ny <= gy + 1 && !too_close; ++ny)
            for (uint64_t nx = gx ? gx - 1 : 0; nx <= gx + 1 && !too_close; ++nx) {
                const uint64_t k = key(ny, nx);
                for (int j = head[bucket(k)]; j >= 0; j = next[j]) {
                    if (cell_of[j] != k) continue;
                    const ab_detected_star &kept = cand[ord[j].idx].s;
                    const double dx = fi.x - kept.x, dy = fi.y - kept.y;
                    if (dx * dx + dy * dy < 9.0) {
                        too_close = true;
                        break;
                    }
                }
            }
        if (!too_close) {
            const uint64_t k = key(gy, gx);
            cell_of[i] = k;
            next[i] = head[bucket(k)];
            head[bucket(k)] = (int)i;
            stars->push_back(fi);
        }
    }
}

// detect_stars (star_detection.rs:86-258) on a device plane; stars sorted by flux, deduplicated
int ab_detect_stars_device(ab_ctx *ctx, const float *img, int64_t rows, int64_t cols, int64_t ld, double sigma_threshold,
                           std::vector<ab_detected_star> *stars, double *bg_median_out, double *bg_sigma_out, ab_pixel_xf xf, size_t max_keep,
                           bool normalize_first, const double *bg_known) {
    stars->clear();
    *bg_median_out = 0.0;
    *bg_sigma_out = 1.0;
    if (rows < 3 || cols < 3) return AB_OK;  // :89-98
    AB_CHECK(ctx, rows * cols < (int64_t(1) << 31), "detect_stars: image too large for 32-bit labels");
    AB_HIP(ctx, hipSetDevice(ctx->device));
    ab_trace trace("detect_stars");
    const int64_t m = std::min(rows, cols);
    const int64_t tile_size = std::min<int64_t>(std::max<int64_t>(m / 8, 32), 256);  // :100
    const int64_t P = rows * cols;
    // The chained form (normalize_first: the registration path): percentiles -> tiles -> background -> threshold -> labels are
    // enqueued back to back, every parameter travelling through FrameDev; the host joins at the component count.
    const int step = (int)std::max<int64_t>(tile_size, 16);
    const int nty = (int)((rows + step - 1) / step), ntx = (int)((cols + step - 1) / step), ntiles = nty * ntx;
    static const bool legacy_tiles = ab_dev_env("AB_TILE_LEGACY") != nullptr;
    // OFF by default (AB_DETECT_CHAIN=1 turns it on): measured on the bench, same box, three runs each -- registration stage 18.4 /
    // 18.6 / 18.7 ms with the two host joins against 21.0 / 21.9 / 22.2 ms chained, although one frame alone gets faster (0.39 ->
    // 0.37 ms).  Sixteen streams share four in-order hardware queues; a stream that enqueues nine packets in one go holds its
    // queue until they have all run, and the three streams behind it wait -- the host joins were what interleaved them.
    static const bool want_chain = ab_dev_env("AB_DETECT_CHAIN") != nullptr;
    const bool chained = normalize_first && !bg_known && ld == cols && ntiles <= kBgTiles && !legacy_tiles && want_chain;
    double bg_median = 0.0, bg_sigma = 1.0, threshold = 0.0;
    FrameDev *fd = nullptr;
    struct Joined {  // what the host reads at the first synchronisation of the chained form (pinned)
        PercentileOut po;
        BgOut bg;
        unsigned int ncomp;
    };
    void *pin = nullptr;
    if (chained) {
        char *dv = nullptr;
        AB_TRY(ab_workspace(ctx, AB_WS_DETECT_DEV, sizeof(FrameDev) + (size_t)ntiles * sizeof(TileOut), (void **)&dv));
        fd = (FrameDev *)dv;
        TileOut *tiles = (TileOut *)(dv + sizeof(FrameDev));
        AB_TRY(ab_pinned(ctx, sizeof(Joined), &pin));
        Joined *jn = (Joined *)pin;
        const int64_t sstep = std::max<int64_t>(P / 100000, 1), ns = (P + sstep - 1) / sstep;
        float *sub = nullptr;
        AB_TRY(ab_workspace(ctx, AB_WS_SUBSAMPLE, (size_t)ns * sizeof(float), (void **)&sub));
        hipLaunchKernelGGL(subsample_kernel, dim3((unsigned)((ns + 255) / 256)), dim3(256), 0, ctx->stream, img, P, sstep, sub, ns);
        if (ns <= (int64_t)kPctPer * 1024)
            hipLaunchKernelGGL(percentiles_reg_kernel, dim3(1), dim3(1024), 0, ctx->stream, sub, (unsigned int)ns, &jn->po, fd);
        else
            hipLaunchKernelGGL(percentiles_mem_kernel, dim3(1), dim3(1024), 0, ctx->stream, sub, (unsigned int)ns, &jn->po, fd);
        AB_TRY(launch_tile_kernels(ctx, ctx->stream, 0, img, rows, cols, ld, step, ntx, ntiles, 1, ab_pixel_xf(), tiles, fd, nullptr, nullptr));
        hipLaunchKernelGGL(bg_threshold_kernel, dim3(1), dim3(1024), 0, ctx->stream, (const TileOut *)tiles, ntiles, sigma_threshold, fd, &jn->bg);
        AB_HIP(ctx, hipGetLastError());
    } else if (bg_known) {  // the caller has estimate_background's result for this plane and transform already
        bg_median = bg_known[0];
        bg_sigma = bg_known[1];
        threshold = bg_median + sigma_threshold * bg_sigma;  // :103
    } else {
        if (normalize_first) AB_TRY(ab_normalize_params_device(ctx, img, P, &xf));
        AB_TRY(ab_estimate_background_device(ctx, img, rows, cols, ld, tile_size, &bg_median, &bg_sigma, xf));
        trace.mark("background");
        threshold = bg_median + sigma_threshold * bg_sigma;  // :103
    }

    const unsigned int root_cap = (unsigned int)(P / 4 + 1);  // 8-connected components cannot be denser
    int *parent = nullptr, *cid = nullptr, *roots = nullptr;
    AB_TRY(ab_workspace(ctx, AB_WS_DETECT_PARENT, (size_t)P * sizeof(int), (void **)&parent));
    AB_TRY(ab_workspace(ctx, AB_WS_DETECT_CID, (size_t)P * sizeof(int), (void **)&cid));
    AB_TRY(ab_workspace(ctx, AB_WS_DETECT_ROOTS, ((size_t)root_cap + 4) * sizeof(int), (void **)&roots));
    unsigned int *nroots = (unsigned int *)(roots + root_cap), *nlab = nroots + 1;
    int *plist = nullptr;
    AB_TRY(ab_workspace(ctx, AB_WS_DETECT_LIST, (size_t)P * sizeof(int), (void **)&plist));
    unsigned int *mask = nullptr;
    AB_TRY(ab_workspace(ctx, AB_WS_DETECT_MASK, ((size_t)P / 32 + 2) * sizeof(unsigned int), (void **)&mask));
    const int gl = (ctx->cu_count > 0 ? ctx->cu_count : 256) * 8;  // list kernels: grid-stride over *nlab entries
    AB_HIP(ctx, hipMemsetAsync(nroots, 0, 2 * sizeof(unsigned int), ctx->stream));
    hipLaunchKernelGGL(label_init_kernel, dim3((unsigned)((P + kInitSub * kInitRounds - 1) / (kInitSub * kInitRounds))), dim3(kInitBlock), 0, ctx->stream, img, (int)rows,
                       (int)cols, ld, threshold, xf, parent, mask, plist, nlab,
                       (int)(ld == cols && (P & 3) == 0 && ((uintptr_t)img & 15) == 0), (const FrameDev *)fd);
    hipLaunchKernelGGL(label_merge_kernel, dim3(gl), dim3(256), 0, ctx->stream, (int)rows, (int)cols, parent, mask, plist, nlab);
    hipLaunchKernelGGL(roots_kernel, dim3(gl / 4), dim3(kRootsBlock), 0, ctx->stream, parent, plist, nlab, roots, cid, nroots, root_cap);
    AB_HIP(ctx, hipGetLastError());
    unsigned int ncomp = 0;
    if (chained) {
        Joined *jn = (Joined *)pin;
        AB_HIP(ctx, hipMemcpyAsync(&jn->ncomp, nroots, sizeof(unsigned int), hipMemcpyDeviceToHost, ctx->stream));
        AB_HIP(ctx, hipStreamSynchronize(ctx->stream));
        ncomp = jn->ncomp;
        bg_median = jn->bg.bg_median;
        bg_sigma = jn->bg.bg_sigma;
    } else {
        AB_TRY(ab_pinned(ctx, 64, &pin));
        AB_HIP(ctx, hipMemcpyAsync(pin, nroots, sizeof(unsigned int), hipMemcpyDeviceToHost, ctx->stream));
        AB_HIP(ctx, hipStreamSynchronize(ctx->stream));
        ncomp = *(const unsigned int *)pin;
    }
    *bg_median_out = bg_median;
    *bg_sigma_out = bg_sigma;
    trace.mark(chained ? "percentiles+background+label+roots" : "label+roots");
    AB_CHECK(ctx, ncomp <= root_cap, "detect_stars: %u components exceed the table capacity", ncomp);
    if (ncomp == 0) return AB_OK;
    void *cbuf = nullptr;
    AB_TRY(ab_workspace(ctx, AB_WS_DETECT_COMPS, (size_t)ncomp * (sizeof(CompStat) + sizeof(CompRec)), &cbuf));
    CompRec *drec = (CompRec *)cbuf;  // 80-byte records from one lane per wave: written to HBM and copied in one piece (straight
    CompStat *dstat = (CompStat *)(drec + ncomp);  // PCIe stores made the kernel 12 us slower than the copy costs)
    AB_TRY(ab_pinned(ctx, (size_t)ncomp * sizeof(CompRec), &pin));
    hipLaunchKernelGGL(comp_init_kernel, dim3((ncomp + 255) / 256), dim3(256), 0, ctx->stream, dstat, ncomp);
    hipLaunchKernelGGL(comp_stats_kernel, dim3(gl), dim3(256), 0, ctx->stream, (int)rows, (int)cols, parent, cid, dstat, plist, nlab);
    hipLaunchKernelGGL(comp_moments_kernel, dim3((ncomp + 4 * kMomPerWave - 1) / (4 * kMomPerWave)), dim3(256), 0, ctx->stream, img, (int)cols, ld, parent, mask, roots, dstat, ncomp,
                       bg_median, xf, drec, (const FrameDev *)fd);
    AB_HIP(ctx, hipGetLastError());
    AB_HIP(ctx, hipMemcpyAsync(pin, drec, (size_t)ncomp * sizeof(CompRec), hipMemcpyDeviceToHost, ctx->stream));
    AB_HIP(ctx, hipStreamSynchronize(ctx->stream));
    const CompRec *recs_begin = (const CompRec *)pin;

    trace.mark("moments+D2H");
    if (trace.on) fprintf(stderr, " (%u components)", ncomp);
    finish_stars(recs_begin, ncomp, bg_sigma, max_keep, stars);
    trace.mark("sort+dedup");
    return AB_OK;
}

// detect_stars (star_detection.rs:86-258) of G <= kGroupMax frames of ONE size in lockstep: the registration batch's form.  The
// frames' background (estimate_background's median and sigma under their load transform xf[f]) comes from the tile pipeline; every
// step of the chain is ONE launch for all of them (blockIdx.y = frame), the counters and the component records come back in one
// copy each.  Results equal G calls of ab_detect_stars_device.
int ab_detect_stars_group_device(ab_ctx *ctx, const float *const *imgs, int G, int64_t rows, int64_t cols, double sigma_threshold, const ab_pixel_xf *xf,
                                 const double (*bg)[2], size_t max_keep, std::vector<ab_detected_star> *stars /* [G] */, const ab_frame_cand *cand) {
    for (int f = 0; f < G; ++f) stars[f].clear();
    if (rows < 3 || cols < 3 || G <= 0) return AB_OK;  // :89-98
    AB_CHECK(ctx, G <= kGroupMax, "detect_stars: groups of at most %d frames", kGroupMax);
    AB_CHECK(ctx, rows * cols < (int64_t(1) << 31), "detect_stars: image too large for 32-bit labels");
    AB_HIP(ctx, hipSetDevice(ctx->device));
    const int64_t P = rows * cols;
    const unsigned int root_cap = (unsigned int)(P / 4 + 1);
    int *parent = nullptr, *cid = nullptr, *roots = nullptr, *plist = nullptr;
    unsigned int *mask = nullptr;
    const int mask_pitch = (int)((cols + 31) / 32 * 32);  // (the tiled labelling's; the two-pass form indexes the mask by pixel number)
    const size_t mask_words = (size_t)rows * (size_t)(mask_pitch / 32) + 2, roots_words = (size_t)root_cap + 4;
    AB_TRY(ab_workspace(ctx, AB_WS_DETECT_PARENT, (size_t)G * P * sizeof(int), (void **)&parent));
    AB_TRY(ab_workspace(ctx, AB_WS_DETECT_CID, (size_t)G * P * sizeof(int), (void **)&cid));
    AB_TRY(ab_workspace(ctx, AB_WS_DETECT_ROOTS, (size_t)G * roots_words * sizeof(int), (void **)&roots));
    // tile-local labelling (label_tile_many_kernel): contiguous planes (always, here) of any width -- the mask's rows are padded to
    // whole words, a row's ragged last quad is loaded float by float; AB_LABEL_LEGACY=1 keeps the two-pass form (the GPU tests run both)
    const bool tiled = !ctx->label_legacy;  // (any width, any dword alignment: round 5)
    // Round 6: with the tile pass's candidate lists (a registration batch's frames: ab_bg_pipeline_cand) the labelling tile IS the 256 x 256
    // background tile and the frame is not read at all (label_bgtile_many_kernel); a frame without lists keeps the 32 x 128 tiles.  The
    // records form only (the chained detection with the device-side selection: decided below with the same conditions).
    const bool bg_tiles = tiled && cand && cand[0].ent && !ctx->label_pixelwise && !ctx->detect_no_recs && !ctx->detect_full_records && !ctx->detect_midjoin &&
                          4 * max_keep <= (size_t)kSelKeep;
    const int tile_h = bg_tiles ? kBgRows : kTileH, tile_w = bg_tiles ? kBgT : kTileW, tile_slots = bg_tiles ? kBgSlots : kTileSlots;
    // (background tiles: one workgroup per sub-tile of kBgRows rows, kBgSub of them per 256 x 256 tile -- the last tile row's may lie below the frame)
    const int64_t tiles_x = (cols + tile_w - 1) / tile_w, tiles_y = bg_tiles ? ((rows + kBgT - 1) / kBgT) * kBgSub : (rows + tile_h - 1) / tile_h, ntile = tiles_x * tiles_y;
    // (tiled: the lists come in kRegions segments, each sized for the tiles that append to it)
    const size_t tiles_per_region = ((size_t)ntile + kRegions - 1) / kRegions;
    const size_t plist_stride = bg_tiles ? 0 : tiles_per_region * (size_t)(kTileH * kTileW), blist_stride = tiles_per_region * (size_t)(tile_w + 2 * tile_h);
    const size_t plist_ints = tiled ? (size_t)kRegions * plist_stride : (size_t)P, blist_ints = tiled ? (size_t)kRegions * blist_stride : 0;
    const size_t lcnt_words = tiled ? (size_t)(2 * kRegions + kRecRegions) * kRegionPitch : 0;
    AB_TRY(ab_workspace(ctx, AB_WS_DETECT_LIST, (size_t)G * (plist_ints + blist_ints + lcnt_words) * sizeof(int), (void **)&plist));
    AB_TRY(ab_workspace(ctx, AB_WS_DETECT_MASK, (size_t)G * mask_words * sizeof(unsigned int), (void **)&mask));
    DetGroup g;
    memset(&g, 0, sizeof g);
    g.n = G;
    bool vec_ok = (P & 3) == 0;
    unsigned int *counters = (unsigned int *)(roots + (size_t)G * root_cap);  // G x 4 words behind the G root tables
    for (int f = 0; f < G; ++f) {
        g.img[f] = imgs[f];
        g.bg_median[f] = bg[f][0];
        g.threshold[f] = bg[f][0] + sigma_threshold * bg[f][1];  // :103
        g.xf[f] = xf[f];
        g.parent[f] = parent + (size_t)f * P;
        g.cid[f] = cid + (size_t)f * P;
        g.plist[f] = plist + (size_t)f * plist_ints;
        g.blist[f] = plist + (size_t)G * plist_ints + (size_t)f * blist_ints;
        g.lcnt[f] = (unsigned int *)(plist + (size_t)G * (plist_ints + blist_ints)) + (size_t)f * lcnt_words;
        g.roots[f] = roots + (size_t)f * root_cap;
        g.mask[f] = mask + (size_t)f * mask_words;
        g.counters[f] = counters + 4 * f;
        vec_ok = vec_ok && ((uintptr_t)imgs[f] & 15) == 0;
    }
    const int cus = ctx->cu_count > 0 ? ctx->cu_count : 256;
    // list kernels: grid-stride over the frame's labelled pixels (a fraction of a percent of the frame); tiled: a multiple of kRegions
    const int gl = tiled ? std::max(kRegions, (cus * 2 / kRegions) * kRegions) : cus * 2;
    // The matcher's 120 (max_keep <= kSelKeep / 4): the brightest kSelKeep candidates are chosen on the device and only they are
    // walked and sent (comp_select_many_kernel).  A caller that wants every star, or AB_DETECT_FULL_RECORDS=1, takes all records.
    const bool select = !ctx->detect_full_records && 4 * max_keep <= (size_t)kSelKeep;
    // With the selection the component kernels' grids no longer depend on the component count, so the whole detection is ONE chain
    // with one host join at its end (labels -> roots (which also initialise the component table) -> statistics -> selection ->
    // moments); the table holds comp_cap components per frame, a frame with more is redone in full.  AB_DETECT_MIDJOIN=1 keeps the
    // join after the root numbering (round 4's shape: the host sizes the table and the grids).
    const bool chained = select && !ctx->detect_midjoin;
    // records form (label_tile_body<true, true>): the tiles gather their components' statistics themselves; roots_many and
    // comp_stats_many give way to comp_merge_many.  AB_DETECT_NO_RECS=1 keeps the pixel-list chain (the GPU tests run both).
    const size_t rec_stride = (((size_t)ntile + kRecRegions - 1) / kRecRegions) * (size_t)tile_slots, rec_cap = (size_t)kRecRegions * rec_stride;
    const bool recs = chained && tiled && !ctx->label_pixelwise && !ctx->detect_no_recs && rec_cap <= (size_t)root_cap;
    AB_CHECK(ctx, !bg_tiles || recs, "detect_stars: the background-tile labelling needs the records form (%zu records for %u roots)", rec_cap, root_cap);
    const unsigned int comp_cap = recs ? (unsigned int)rec_cap : (unsigned int)std::min<int64_t>(root_cap, (int64_t)1 << 18);
    void *pin = nullptr;
    unsigned int *selout = nullptr;
    if (chained) {
        void *cbuf = nullptr;
        AB_TRY(ab_workspace(ctx, AB_WS_DETECT_COMPS, (size_t)G * comp_cap * sizeof(CompStat) + (size_t)G * kSelCap * sizeof(unsigned int), &cbuf));
        CompStat *dstat = (CompStat *)cbuf;
        unsigned int *dsel = (unsigned int *)(dstat + (size_t)G * comp_cap);
        AB_TRY(ab_pinned(ctx, (size_t)G * kSelCap * sizeof(CompRec) + (size_t)G * 8 * sizeof(unsigned int), &pin));
        selout = (unsigned int *)((CompRec *)pin + (size_t)G * kSelCap);  // 8 words per frame: selected, candidates, components, cut key, dense tiles
        for (int f = 0; f < G; ++f) {
            g.rec[f] = (CompRec *)pin + (size_t)f * kSelCap;
            g.st[f] = dstat + (size_t)f * comp_cap;
            g.sel[f] = dsel + (size_t)f * kSelCap;
            g.selout[f] = selout + 8 * f;
            for (int k = 0; k < 8; ++k) selout[8 * f + k] = 0;
        }
        g.comp_cap = comp_cap;
        g.chained = 1;
    }
    AB_HIP(ctx, hipMemsetAsync(counters, 0, (size_t)G * 4 * sizeof(unsigned int), ctx->stream));
    g.tiled = tiled ? 1 : 0;
    g.tile_h = tile_h;
    g.tile_w = tile_w;
    for (int f = 0; f < G; ++f) {
        const bool has = bg_tiles && cand[f].ent;
        g.cand_ent[f] = has ? (const uint2 *)cand[f].ent : nullptr;
        g.cand_cnt[f] = has ? cand[f].cnt : nullptr;
        g.cand_cut[f] = has ? cand[f].cut : nullptr;
    }
    g.mask_pitch = mask_pitch;
    g.recs = recs ? 1 : 0;
    g.rec_stride = rec_stride;
    g.plist_stride = plist_stride;
    g.blist_stride = blist_stride;
    if (tiled) AB_HIP(ctx, hipMemsetAsync(g.lcnt[0], 0, (size_t)G * lcnt_words * sizeof(unsigned int), ctx->stream));
    if (tiled) {
        // (one workgroup per tile.  Tried: fewer workgroups that walk several tiles with the next tile's loads in flight -- 128 / 142 us
        // per group of four 4096^2 frames with 1024 / 2048 workgroups against 101 us, profiles/r05_label_tile_variants.txt; the loop
        // alone, one trip per workgroup, cost 40 us: 36 VGPRs instead of 20 and the prefetch's predication.  The stage did not move.)
        if (bg_tiles)
            hipLaunchKernelGGL(label_bgtile_many_kernel, dim3((unsigned)ntile, (unsigned)G), dim3(kBgThreads), 0, ctx->stream, g, (int)rows, (int)cols);
        else if (recs)
            hipLaunchKernelGGL((label_tile_many_kernel<true, true>), dim3((unsigned)ntile, (unsigned)G), dim3(kTileThreads), 0, ctx->stream, g, (int)rows, (int)cols);
        else if (ctx->label_pixelwise)
            hipLaunchKernelGGL((label_tile_many_kernel<false, false>), dim3((unsigned)ntile, (unsigned)G), dim3(kTileThreads), 0, ctx->stream, g, (int)rows, (int)cols);
        else
            hipLaunchKernelGGL((label_tile_many_kernel<true, false>), dim3((unsigned)ntile, (unsigned)G), dim3(kTileThreads), 0, ctx->stream, g, (int)rows, (int)cols);
        hipLaunchKernelGGL(label_border_many_kernel, dim3(gl, G), dim3(256), 0, ctx->stream, g, (int)rows, (int)cols);
    } else {
        hipLaunchKernelGGL(label_init_many_kernel, dim3((unsigned)((P + kInitSub * kInitRounds - 1) / (kInitSub * kInitRounds)), (unsigned)G), dim3(kInitBlock), 0, ctx->stream, g,
                           (int)rows, (int)cols, cols, (int)vec_ok);
        hipLaunchKernelGGL(label_merge_many_kernel, dim3(gl, G), dim3(256), 0, ctx->stream, g, (int)rows, (int)cols);
    }
    if (recs) {
        hipLaunchKernelGGL(comp_merge_many_kernel, dim3(4 * kRecRegions, G), dim3(256), 0, ctx->stream, g);
    } else {
        hipLaunchKernelGGL(roots_many_kernel, dim3(gl, G), dim3(kRootsBlock), 0, ctx->stream, g, chained ? comp_cap : root_cap);
    }
    AB_HIP(ctx, hipGetLastError());
    if (chained) {
        if (!recs) hipLaunchKernelGGL(comp_stats_many_kernel, dim3(gl, G), dim3(256), 0, ctx->stream, g, (int)rows, (int)cols, cols);
        hipLaunchKernelGGL(comp_select_many_kernel, dim3(G), dim3(kSelThreads), 0, ctx->stream, g);
        hipLaunchKernelGGL(comp_moments_many_kernel, dim3((kSelCap + 4 * kMomPerWave - 1) / (4 * kMomPerWave), G), dim3(256), 0, ctx->stream, g, (int)cols, cols);
        AB_HIP(ctx, hipGetLastError());
        AB_HIP(ctx, hipStreamSynchronize(ctx->stream));
        bool redo[kGroupMax] = {};
        // (what the trace prints is copied out BEFORE any frame is redone: the full path may regrow -- and free -- the pinned buffer
        // selout lives in: ADVICE r5)
        unsigned int sel_m[kGroupMax] = {}, sel_ncand[kGroupMax] = {}, sel_ncomp[kGroupMax] = {};
        size_t sel_stars[kGroupMax] = {};
        for (int f = 0; f < G; ++f) {
            const unsigned int m = selout[8 * f], ncand = selout[8 * f + 1], ncomp = selout[8 * f + 2], cut_key = selout[8 * f + 3];
            if (bg_tiles) ab_count_fallback(ctx, AB_FB_LABEL_TILES_DENSE, selout[8 * f + 4]);
            sel_m[f] = m;
            sel_ncand[f] = ncand;
            sel_ncomp[f] = ncomp;
            AB_CHECK(ctx, m <= kSelCap, "detect_stars: the selection of frame %d holds %u components", f, m);
            AB_CHECK(ctx, ncomp <= root_cap, "detect_stars: %u components exceed the table capacity", ncomp);
            if (ncomp > comp_cap) {  // more components than the chained table holds (or a tile with more components than record slots)
                redo[f] = true;
                ab_count_fallback(ctx, recs && ncomp == comp_cap + 1u ? AB_FB_TILE_SLOTS : AB_FB_COMPONENT_TABLE);
                continue;
            }
            finish_stars(g.rec[f], m, bg[f][1], max_keep, &stars[f]);
            sel_stars[f] = stars[f].size();
            redo[f] = stars[f].size() < max_keep && ncand > m;  // (see below)
            if (redo[f]) ab_count_fallback(ctx, AB_FB_SELECTION_SHORT);
            if (!redo[f] && selection_cut_too_close(stars[f], max_keep, m, ncand, cut_key)) {
                redo[f] = true;
                ab_count_fallback(ctx, AB_FB_SELECTION_CUT);
            }
        }
        for (int f = 0; f < G; ++f) {
            if (!redo[f]) continue;
            static const bool trace = ab_env("AB_TRACE") != nullptr;
            if (trace)
                fprintf(stderr, "[ab_trace] detect_stars: frame %d of the group redone in full (%u components, %u candidates, %u selected, %zu stars of them)\n", f,
                        sel_ncomp[f], sel_ncand[f], sel_m[f], sel_stars[f]);
            double m0 = 0.0, s0 = 0.0;
            AB_TRY(ab_detect_stars_device(ctx, imgs[f], rows, cols, cols, sigma_threshold, &stars[f], &m0, &s0, xf[f], max_keep, false, bg[f]));
            ab_count_fallback(ctx, AB_FB_FRAMES_REDONE);
        }
        return AB_OK;
    }
    AB_TRY(ab_pinned(ctx, (size_t)G * 4 * sizeof(unsigned int), &pin));
    AB_HIP(ctx, hipMemcpyAsync(pin, counters, (size_t)G * 4 * sizeof(unsigned int), hipMemcpyDeviceToHost, ctx->stream));
    AB_HIP(ctx, hipStreamSynchronize(ctx->stream));
    size_t total = 0, off[kGroupMax + 1];
    unsigned int max_nc = 0;
    for (int f = 0; f < G; ++f) {
        g.ncomp[f] = ((const unsigned int *)pin)[4 * f];
        AB_CHECK(ctx, g.ncomp[f] <= root_cap, "detect_stars: %u components exceed the table capacity", g.ncomp[f]);
        off[f] = total;
        total += g.ncomp[f];
        max_nc = std::max(max_nc, g.ncomp[f]);
    }
    off[G] = total;
    if (total == 0) return AB_OK;
    void *cbuf = nullptr;
    AB_TRY(ab_workspace(ctx, AB_WS_DETECT_COMPS, total * sizeof(CompStat) + (size_t)G * kSelCap * sizeof(unsigned int), &cbuf));
    CompStat *dstat = (CompStat *)cbuf;
    unsigned int *dsel = (unsigned int *)(dstat + total);
    // the component records (72 B each, written once, by one lane, never read on the device) go STRAIGHT into pinned host memory
    const size_t rec_count = select ? (size_t)G * kSelCap : total;
    AB_TRY(ab_pinned(ctx, rec_count * sizeof(CompRec) + (size_t)G * 4 * sizeof(unsigned int), &pin));
    selout = (unsigned int *)((CompRec *)pin + rec_count);
    for (int f = 0; f < G; ++f) {
        g.rec[f] = (CompRec *)pin + (select ? (size_t)f * kSelCap : off[f]);
        g.st[f] = dstat + off[f];
        g.sel[f] = select ? dsel + (size_t)f * kSelCap : nullptr;
        g.selout[f] = selout + 4 * f;
        selout[4 * f] = selout[4 * f + 1] = selout[4 * f + 3] = 0;
    }
    hipLaunchKernelGGL(comp_init_many_kernel, dim3((max_nc + 255) / 256, G), dim3(256), 0, ctx->stream, g);
    hipLaunchKernelGGL(comp_stats_many_kernel, dim3(gl, G), dim3(256), 0, ctx->stream, g, (int)rows, (int)cols, cols);
#ifdef AB_DEV_ABLATION
    if (ab_dev_env("AB_ABLATE_MOMENTS")) {
        const int v = atoi(ab_dev_env("AB_ABLATE_MOMENTS"));
        AB_HIP(ctx, hipMemcpyToSymbol(HIP_SYMBOL(g_mom_ablate), &v, sizeof v));
    }
#endif
    if (select) {
        hipLaunchKernelGGL(comp_select_many_kernel, dim3(G), dim3(kSelThreads), 0, ctx->stream, g);
        hipLaunchKernelGGL(comp_moments_many_kernel, dim3((kSelCap + 4 * kMomPerWave - 1) / (4 * kMomPerWave), G), dim3(256), 0, ctx->stream, g, (int)cols, cols);
    } else {
        hipLaunchKernelGGL(comp_moments_many_kernel, dim3((max_nc + 4 * kMomPerWave - 1) / (4 * kMomPerWave), G), dim3(256), 0, ctx->stream, g, (int)cols, cols);
    }
    AB_HIP(ctx, hipGetLastError());
    AB_HIP(ctx, hipStreamSynchronize(ctx->stream));
    bool redo[kGroupMax] = {};
    for (int f = 0; f < G; ++f) {
        if (!select) {
            finish_stars((const CompRec *)pin + off[f], g.ncomp[f], bg[f][1], max_keep, &stars[f]);
            continue;
        }
        const unsigned int m = selout[4 * f], ncand = selout[4 * f + 1], cut_key = selout[4 * f + 3];
        AB_CHECK(ctx, m <= kSelCap, "detect_stars: the selection of frame %d holds %u components", f, m);
        finish_stars(g.rec[f], m, bg[f][1], max_keep, &stars[f]);
        // the brightest kSelKeep did not yield max_keep survivors and fainter candidates exist (a crowded field whose dedup eats
        // hundreds, or a crowd of equal fluxes at the cut): the whole list decides
        redo[f] = stars[f].size() < max_keep && ncand > m;
        if (redo[f]) ab_count_fallback(ctx, AB_FB_SELECTION_SHORT);
        if (!redo[f] && selection_cut_too_close(stars[f], max_keep, m, ncand, cut_key)) {
            redo[f] = true;
            ab_count_fallback(ctx, AB_FB_SELECTION_CUT);
        }
    }
    for (int f = 0; f < G; ++f) {  // (after every frame's records have been read: the full path re-carves the workspaces and the pinned buffer)
        if (!redo[f]) continue;
        double m0 = 0.0, s0 = 0.0;
        AB_TRY(ab_detect_stars_device(ctx, imgs[f], rows, cols, cols, sigma_threshold, &stars[f], &m0, &s0, xf[f], max_keep, false, bg[f]));
        ab_count_fallback(ctx, AB_FB_FRAMES_REDONE);
    }
    return AB_OK;
}

int ab_normalize_params_device(ab_ctx *ctx, const float *img, int64_t len, ab_pixel_xf *xf) {
    AB_HIP(ctx, hipSetDevice(ctx->device));
    *xf = ab_pixel_xf();
    if (len == 0) return AB_OK;
    const int64_t step = std::max<int64_t>(len / 100000, 1);
    const int64_t ns = (len + step - 1) / step;
    void *pin = nullptr;  // the kernel writes its 16 bytes straight into pinned host memory
    AB_TRY(ab_pinned(ctx, sizeof(PercentileOut), &pin));
    ab_trace trace("normalize_params");
    float *sub = nullptr;
    AB_TRY(ab_workspace(ctx, AB_WS_SUBSAMPLE, (size_t)ns * sizeof(float), (void **)&sub));
    hipLaunchKernelGGL(subsample_kernel, dim3((unsigned)((ns + 255) / 256)), dim3(256), 0, ctx->stream, img, len, step, sub, ns);
    if (ns <= (int64_t)kPctPer * 1024)
        hipLaunchKernelGGL(percentiles_reg_kernel, dim3(1), dim3(1024), 0, ctx->stream, sub, (unsigned int)ns, (PercentileOut *)pin, (FrameDev *)nullptr);
    else
        hipLaunchKernelGGL(percentiles_mem_kernel, dim3(1), dim3(1024), 0, ctx->stream, sub, (unsigned int)ns, (PercentileOut *)pin, (FrameDev *)nullptr);
    AB_HIP(ctx, hipGetLastError());
    AB_HIP(ctx, hipStreamSynchronize(ctx->stream));
    trace.mark("subsample+percentiles+sync");
    *xf = xf_from_percentiles(*(const PercentileOut *)pin);
    return AB_OK;
}

// the same for n planes of `len` pixels each (device pointers), two launches and one synchronisation for all of them
int ab_normalize_params_many_device(ab_ctx *ctx, const float *const *planes, size_t n, int64_t len, ab_pixel_xf *xf) {
    AB_HIP(ctx, hipSetDevice(ctx->device));
    for (size_t i = 0; i < n; ++i) xf[i] = ab_pixel_xf();
    if (len == 0 || n == 0) return AB_OK;
    const int64_t step = std::max<int64_t>(len / 100000, 1);
    const int64_t ns = (len + step - 1) / step;
    for (size_t first = 0; first < n; first += kManyPlanes) {
        const size_t m = std::min<size_t>(kManyPlanes, n - first);
        PlaneList pl;
        for (size_t i = 0; i < (size_t)kManyPlanes; ++i) pl.p[i] = planes[first + (i < m ? i : 0)];
        void *pin = nullptr;
        AB_TRY(ab_pinned(ctx, m * sizeof(PercentileOut), &pin));
        float *sub = nullptr;
        AB_TRY(ab_workspace(ctx, AB_WS_SUBSAMPLE, m * (size_t)ns * sizeof(float), (void **)&sub));
        hipLaunchKernelGGL(subsample_many_kernel, dim3((unsigned)((ns + 255) / 256), (unsigned)m), dim3(256), 0, ctx->stream, pl, len, step, sub, ns);
        if (ns <= (int64_t)kPctPer * 1024)
            hipLaunchKernelGGL(percentiles_many_reg_kernel, dim3((unsigned)m), dim3(1024), 0, ctx->stream, (const float *)sub, (unsigned int)ns, (PercentileOut *)pin,
                               (ab_pixel_xf *)nullptr, (ab_pixel_xf *)nullptr);
        else
            hipLaunchKernelGGL(percentiles_many_mem_kernel, dim3((unsigned)m), dim3(1024), 0, ctx->stream, (const float *)sub, (unsigned int)ns, (PercentileOut *)pin,
                               (ab_pixel_xf *)nullptr, (ab_pixel_xf *)nullptr);
        AB_HIP(ctx, hipGetLastError());
        AB_HIP(ctx, hipStreamSynchronize(ctx->stream));
        for (size_t i = 0; i < m; ++i) xf[first + i] = xf_from_percentiles(((const PercentileOut *)pin)[i]);
    }
    return AB_OK;
}

// normalize_for_detection (affine.rs:24-53) into a contiguous device buffer; *cloned = 1 when the
// reference returns image.clone() (too few samples / flat range): out then equals the input
int ab_normalize_for_detection_device(ab_ctx *ctx, const float *img, int64_t len, float *out, int *cloned) {
    ab_pixel_xf xf;
    AB_TRY(ab_normalize_params_device(ctx, img, len, &xf));
    *cloned = xf.on ? 0 : 1;
    if (len == 0) return AB_OK;
    if (!xf.on) {
        if (out != img) AB_HIP(ctx, hipMemcpyAsync(out, img, (size_t)len * sizeof(float), hipMemcpyDeviceToDevice, ctx->stream));
        return AB_OK;
    }
    const int g = (int)std::min<int64_t>((len + 255) / 256, (int64_t)(ctx->cu_count > 0 ? ctx->cu_count : 256) * 8);
    hipLaunchKernelGGL(normalize_kernel, dim3(g), dim3(256), 0, ctx->stream, img, len, xf.lo, xf.inv, out);
    AB_HIP(ctx, hipGetLastError());
    return AB_OK;
}

extern "C" {

// the tile map behind estimate_background: per tile (row-major, ceil(rows / step) x ceil(cols / step), step = max(tile_size, 16))
// the sigma-clipped median and sigma, and whether the tile had the 8 valid pixels it needs (star_detection.rs:47-68)
int ab_background_tile_stats(ab_ctx *ctx, const ab_plane *img, int64_t tile_size, double *out_median, double *out_sigma, int32_t *out_valid,
                             size_t cap, size_t *out_tiles, size_t *out_tiles_x) try {
    if (!ctx) return AB_ERR_INVALID;
    AB_CHECK(ctx, img && out_tiles, "null argument");
    StagedPlane in;
    AB_TRY(ab_stage_in(ctx, img, &in));
    std::vector<TileOut> h;
    int ntx = 0;
    const int rc = tile_stats_host(ctx, in.dptr, in.rows, in.cols, in.cols, tile_size, ab_pixel_xf(), &h, &ntx);
    ab_stage_release(ctx, &in);
    if (rc != AB_OK) return rc;
    *out_tiles = h.size();
    if (out_tiles_x) *out_tiles_x = (size_t)ntx;
    for (size_t i = 0; i < h.size() && i < cap; ++i) {
        if (out_median) out_median[i] = h[i].median;
        if (out_sigma) out_sigma[i] = h[i].sigma;
        if (out_valid) out_valid[i] = h[i].valid;
    }
    return AB_OK;
} AB_CATCH(ctx)

int ab_estimate_background(ab_ctx *ctx, const ab_plane *img, int64_t tile_size, double *out_median, double *out_sigma) try {
    if (!ctx) return AB_ERR_INVALID;
    AB_CHECK(ctx, img && out_median && out_sigma, "null argument");
    StagedPlane in;
    AB_TRY(ab_stage_in(ctx, img, &in));
    const int rc = ab_estimate_background_device(ctx, in.dptr, in.rows, in.cols, in.cols, tile_size, out_median, out_sigma);
    ab_stage_release(ctx, &in);
    return rc;
} AB_CATCH(ctx)

int ab_detect_stars(ab_ctx *ctx, const ab_plane *img, double sigma_threshold, ab_detected_star *out, size_t cap, size_t *out_count,
                    size_t *out_total, double *bg_median, double *bg_sigma) try {
    if (!ctx) return AB_ERR_INVALID;
    AB_CHECK(ctx, img && out_count && (out || cap == 0), "null argument");
    AB_CHECK(ctx, img->data && img->rows > 0 && img->cols > 0, "plane is null or has a zero dimension");
    StagedPlane in;
    AB_TRY(ab_stage_in(ctx, img, &in));
    std::vector<ab_detected_star> stars;
    double m = 0.0, s = 1.0;
    const int rc = ab_detect_stars_device(ctx, in.dptr, in.rows, in.cols, in.cols, sigma_threshold, &stars, &m, &s);
    ab_stage_release(ctx, &in);
    if (rc != AB_OK) return rc;
    const size_t n = std::min(cap, stars.size());
    for (size_t i = 0; i < n; ++i) out[i] = stars[i];
    *out_count = n;
    if (out_total) *out_total = stars.size();
    if (bg_median) *bg_median = m;
    if (bg_sigma) *bg_sigma = s;
    return AB_OK;
} AB_CATCH(ctx)

int ab_normalize_for_detection(ab_ctx *ctx, const ab_plane *img, ab_plane_mut *out) try {
    if (!ctx) return AB_ERR_INVALID;
    AB_CHECK(ctx, img && out && img->rows == out->rows && img->cols == out->cols, "null plane or mismatched dims");
    StagedPlane in;
    AB_TRY(ab_stage_in(ctx, img, &in));
    StagedOut so;
    int rc = ab_stage_out_begin(ctx, out, &so);
    if (rc == AB_OK) {
        int cloned = 0;
        rc = ab_normalize_for_detection_device(ctx, in.dptr, in.rows * in.cols, so.dptr, &cloned);
        if (rc == AB_OK)
            rc = ab_stage_out_finish(ctx, &so);
        else
            ab_stage_out_abort(ctx, &so);
    }
    ab_stage_release(ctx, &in);
    return rc;
} AB_CATCH(ctx)

}  // extern "C"
