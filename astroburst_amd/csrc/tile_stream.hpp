// Sigma-clipped median / MAD of one background tile (<= 256 x 256 px), exactly, by a STREAMING workgroup (gfx950).
//
// Replaces sigma_clipped_stats(values, 3.0, 2) of estimate_background's per-tile body
// (core/analysis/star_detection.rs:47-68, math/sigma_clip.rs:4-34, math/median.rs:27-63) -- the same arithmetic as
// tile_bucket.hpp, which stays in the library as the fallback for the tiles this kernel declines.
//
// tile_bucket.hpp keeps a tile's 65 536 keys in the register file (240 VGPRs x 512 threads: one tile per CU, nothing else
// resident beside it) and answers every order statistic by a sweep through the VGPR index register: ~190 000 cycles per
// tile, 89 us per 4096^2 frame, and while it runs it owns the chip.  Here nothing per pixel stays on chip:
//
//   pass 1  the tile streams through once (float4 rows, HBM) into a 4096-bucket LDS histogram, zoomed on the quartiles of a
//           256-pixel sample exactly as tile_bucket.hpp zooms;
//   plan    the three rounds of sigma_clipped_stats are SIMULATED on the histogram alone, at bucket resolution: where the
//           median's bucket will be, where the two rings of the MAD's candidates will be, where the clipping bounds will
//           fall -- each with a margin of a few buckets.  Those bucket ranges (a few dozen buckets, 1 - 3 thousand keys)
//           are the HOT ZONES;
//   pass 2  the tile streams through a second time (L2 / Infinity Cache) and the keys of hot buckets are written to an LDS
//           list, BUCKET-SORTED: a bucket's slice of the list is known from the prefix sum, so a key's position is one LDS
//           atomic on its bucket's cursor;
//   stats   the three rounds run for real on prefix sum + list: the median's bucket is a slice of the list, the MAD's
//           candidates are two slices (guess-and-verify with the same proof as tile_bucket.hpp's fast path), the exact
//           number of keys cut by a clipping bound is a prefix-sum entry plus a count inside ONE bucket's slice.
//
// Every answer is either exact or the tile is DECLINED (a request outside the hot zones, a list that would overflow, a
// guess whose proof fails, a rank inside a catch-all bucket): declined tiles are appended to a list that
// tile_background_bucket_kernel works off afterwards.  On sky tiles nothing is declined; heavy ties, flat and multi-modal
// tiles are (tests/test_gpu_tile_stats.py compares all of them with the oracle bit for bit, through both kernels).
//
// 256 threads, ~48 KB of LDS and < 128 VGPRs per workgroup: three tiles per CU in flight, other kernels' waves beside them.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

#include "ab_common.hpp"
#include "tile_bucket.hpp"

namespace ts {

using tb::dev_key;
using tb::wave_max;
using tb::wave_min;
using tb::wave_scan_incl;
using tb::wave_sort64;
using tb::wave_sum;
using tb::Window;
using tb::OP_MAX;
using tb::OP_MIN;
using tb::OP_SUM;

constexpr int kThreads = 256;
constexpr int kWaves = kThreads / 64;
constexpr int kBuckets = 4096;
constexpr int kTop = kBuckets - 1;           // catch-all above the zoom window (bucket 0: below it)
constexpr int kPer = kBuckets / kThreads;    // buckets a thread owns in the scans
constexpr int kHotCap = 4096;                // keys the hot list holds
constexpr int kSlotCap = 512;                // hot buckets
constexpr int kSelCap = 2048;                // keys one select works on
constexpr int kMine = kSelCap / kThreads;
constexpr int kMaxZones = 16;
constexpr int kRing = 2;      // the MAD's candidates: buckets within kRing bucket widths of median -+ t* (as tile_bucket.hpp)
constexpr int kRingPlan = 3;  // + what the plan may be off by
constexpr int kEdgePlan = 12; // a clipping bound: predicted to +- this many bucket widths
constexpr unsigned int kNone = 0xffffffffu;

// reasons a tile is declined (debug / statistics; the fallback does not care)
enum Decline {
    D_NONE = 0, D_ZONES, D_HOT_OVERFLOW, D_MED_CATCHALL, D_MED_COLD, D_MED_BIG, D_DEV_BM, D_DEV_NOFIRST, D_DEV_SHAPE, D_DEV_COLD,
    D_DEV_BIG, D_DEV_COUNT, D_DEV_PROOF, D_EDGE_COLD, D_EDGE_CATCHALL
};

struct Shared {
    unsigned int prefix[kBuckets];  // bucket counts, then their inclusive prefix sum
    union {
        unsigned short lut[kBuckets];  // pass 2: bucket -> slot (0xffff: cold)
        unsigned int tmp[kSelCap];     // afterwards: the keys a select works on
    };
    unsigned int list[kHotCap];     // the hot keys, bucket-sorted
    unsigned int cursor[kSlotCap];  // pass 2: next free position of the slot's slice; afterwards: its end
    unsigned int lstart[kSlotCap];  // the slice's start
    unsigned int sel_hist[3][256];
    unsigned int sel_part[2 * kWaves];
    unsigned int part[4 * kWaves];
    unsigned int scal[8];
    int zone_lo[kMaxZones], zone_hi[kMaxZones], zone_slot[kMaxZones];
    int nzones;
    int decline;
#ifdef AB_TILE_TIMING
    long long t_phase[8];  // 0 sample 1 pass1 2 scan 3 plan 4 zones+lut 5 pass2 6 stats
    long long t_mark;
#endif
};

#ifdef AB_TILE_TIMING
#define TS_MARK(sh, i)                             \
    do {                                           \
        if (threadIdx.x == 0) {                    \
            const long long now_ = clock64();      \
            (sh).t_phase[i] += now_ - (sh).t_mark; \
            (sh).t_mark = now_;                    \
        }                                          \
    } while (0)
#else
#define TS_MARK(sh, i) \
    do {               \
    } while (0)
#endif

// ---- histogram geometry: bucket 0 = keys below zlo | buckets 1 .. 4094 = 2^shift keys each from zlo | bucket 4095 = the rest ----
struct Geo {
    uint32_t zlo, base;  // base = zlo - 2^shift (zlo >= 2^shift)
    int shift;           // <= 19: zlo + (4094 << shift) cannot wrap
    __device__ __forceinline__ int bucket_of(uint32_t key) const {
        const uint32_t b = __builtin_elementwise_sub_sat(key, base) >> shift;
        return (int)(b < (uint32_t)kTop ? b : (uint32_t)kTop);
    }
    __device__ __forceinline__ uint32_t first_key(int b) const {  // smallest key of bucket b
        if (b <= 0) return 0x33d6bf96u;  // the smallest candidate: v > 1e-7f
        if (b >= kTop) return zlo + ((uint32_t)(kTop - 1) << shift);
        return zlo + ((uint32_t)(b - 1) << shift);
    }
    __device__ __forceinline__ uint32_t last_key(int b) const {  // largest key of bucket b (candidates are finite)
        if (b <= 0) return zlo - 1u;
        if (b >= kTop) return 0x7f7fffffu;
        return zlo + ((uint32_t)b << shift) - 1u;
    }
};

template <int OP>
__device__ __forceinline__ unsigned int wred(unsigned int x) {
    return OP == OP_SUM ? wave_sum(x) : (OP == OP_MIN ? wave_min(x) : wave_max(x));
}
template <int OP>
__device__ __forceinline__ unsigned int comb(unsigned int x, unsigned int y) {
    return OP == OP_SUM ? x + y : (OP == OP_MIN ? min(x, y) : max(x, y));
}
// two values reduced over the workgroup; every thread gets the results (two barriers)
template <int OP0, int OP1>
__device__ __forceinline__ void block_reduce2(Shared &sh, unsigned int &a, unsigned int &b) {
    a = wred<OP0>(a);
    b = wred<OP1>(b);
    const int w = threadIdx.x >> 6;
    __syncthreads();  // (the slots may still be read from the previous reduction)
    if ((threadIdx.x & 63) == 0) {
        sh.part[2 * w] = a;
        sh.part[2 * w + 1] = b;
    }
    __syncthreads();
    unsigned int ra = sh.part[0], rb = sh.part[1];
#pragma unroll
    for (int i = 1; i < kWaves; ++i) {
        ra = comb<OP0>(ra, sh.part[2 * i]);
        rb = comb<OP1>(rb, sh.part[2 * i + 1]);
    }
    a = ra;
    b = rb;
}

// inclusive prefix sum of sh.prefix[0 .. 4096) in place
__device__ __forceinline__ void scan_prefix(Shared &sh) {
    const int t = threadIdx.x, b0 = t * kPer, w = t >> 6, lane = t & 63;
    unsigned int v[kPer];
#pragma unroll
    for (int j = 0; j < kPer; j += 4) {
        const uint4 q = *reinterpret_cast<const uint4 *>(&sh.prefix[b0 + j]);
        v[j] = q.x;
        v[j + 1] = q.y;
        v[j + 2] = q.z;
        v[j + 3] = q.w;
    }
#pragma unroll
    for (int j = 1; j < kPer; ++j) v[j] += v[j - 1];
    const unsigned int tot = v[kPer - 1], incl = wave_scan_incl(tot);
    if (lane == 63) sh.part[w] = incl;
    __syncthreads();
    unsigned int before = incl - tot;
#pragma unroll
    for (int i = 0; i < kWaves; ++i) before += i < w ? sh.part[i] : 0u;
#pragma unroll
    for (int j = 0; j < kPer; j += 4) {
        uint4 q;
        q.x = v[j] + before;
        q.y = v[j + 1] + before;
        q.z = v[j + 2] + before;
        q.w = v[j + 3] + before;
        *reinterpret_cast<uint4 *>(&sh.prefix[b0 + j]) = q;
    }
    __syncthreads();
}

__device__ __forceinline__ unsigned int below(const Shared &sh, int b) { return b <= 0 ? 0u : sh.prefix[b - 1]; }  // keys in buckets < b

// the buckets of the global ranks g_lo <= g_hi (both < the candidate count): bucket b covers ranks [prefix[b-1], prefix[b])
__device__ __forceinline__ void find_ranks(Shared &sh, unsigned int g_lo, unsigned int g_hi, int *b_lo, int *b_hi) {
    const int t = threadIdx.x;
    const unsigned int lo = below(sh, kPer * t), hi = sh.prefix[kPer * t + kPer - 1];
    auto scan = [&](unsigned int g, int slot) {
        if (g >= lo && g < hi) {  // exactly one thread
            int b = kPer * t;
#pragma unroll
            for (int j = 0; j < kPer - 1; ++j) b += g >= sh.prefix[kPer * t + j] ? 1 : 0;
            sh.scal[slot] = (unsigned int)b;
        }
    };
    scan(g_lo, 0);
    scan(g_hi, 1);
    __syncthreads();
    *b_lo = (int)sh.scal[0];
    *b_hi = (int)sh.scal[1];
    __syncthreads();
}

// window candidates with key < first key of bucket b  (b in [0, 4096])
__device__ __forceinline__ unsigned int wprefix(const Shared &sh, const Window &w, int b) {
    const unsigned int p = below(sh, b);
    const unsigned int q = p < w.c_lo ? w.c_lo : (p > w.c_le_hi ? w.c_le_hi : p);
    return q - w.c_lo;
}

// ---- the deviation side: where the MAD's candidates are ---------------------------------------------------------------------
struct DevGeo {
    Geo g;
    float mf;     // the median as f32
    float delta;  // width of the median's bucket in value units
    int bm;       // the bucket of the median
    __device__ __forceinline__ int bucket_at(float x) const { return x > 0.0f ? g.bucket_of(x < 3.0e38f ? __float_as_uint(x) : 0x7f7fffffu) : 0; }
    // the buckets that hold median - t and median + t, either side of bm
    __device__ __forceinline__ void ring_of(float t, int *bl, int *br) const {
        const int l = bucket_at(mf - t), r = bucket_at(mf + t);
        *bl = l < bm ? l : bm;
        *br = r > bm ? r : bm;
    }
};
__device__ __forceinline__ bool make_devgeo(const Geo &g, double median, DevGeo *d) {
    d->g = g;
    d->mf = (float)median;  // may round up past the median: step back below
    int b = d->mf <= 0.0f ? 0 : g.bucket_of(__float_as_uint(d->mf));
    auto left_of_median = [&](uint32_t key) { return (double)__uint_as_float(key) <= median; };
    while (b > 0 && !left_of_median(g.first_key(b))) --b;
    while (b < kTop && left_of_median(g.first_key(b + 1))) ++b;
    d->bm = b;
    if (b <= 0 || b >= kTop) return false;
    d->delta = __uint_as_float(g.first_key(b) + (1u << g.shift)) - __uint_as_float(g.first_key(b));
    return d->delta > 0.0f;
}

// F~(t) = window candidates in the buckets that meet [median - t, median + t]: the first threshold index i (t = (i + 1) delta)
// with F~ >= need; kNone if none of the 4096 does.  One threshold per thread in a window of 256 around `guess` first (the
// window's answer is THE first iff it lies strictly inside the window: F~ is monotone), else all 4096, 16 per thread.
__device__ __forceinline__ unsigned int mad_first(Shared &sh, const Window &w, const DevGeo &d, unsigned int need, unsigned int guess) {
    unsigned int first = kNone, z = 0;
    auto reaches = [&](unsigned int i) {
        int bl, br;
        d.ring_of((float)(i + 1u) * d.delta, &bl, &br);
        return wprefix(sh, w, br + 1) - wprefix(sh, w, bl) >= need;
    };
    if (guess != kNone) {
        const unsigned int gi = guess < 3900u ? guess : 3900u, i_lo = gi > 128u ? gi - 128u : 0u;
        const unsigned int i = i_lo + threadIdx.x;
        if (reaches(i)) first = i;
        block_reduce2<OP_MIN, OP_SUM>(sh, first, z);
        if (first != kNone && (first > i_lo || i_lo == 0u)) return first;
    }
    first = kNone;
#pragma unroll 1
    for (int j = 0; j < kPer; ++j) {
        const unsigned int i = (unsigned int)(j * kThreads) + threadIdx.x;
        if (first == kNone && reaches(i)) first = i;
    }
    block_reduce2<OP_MIN, OP_SUM>(sh, first, z);
    return first;
}

// ---- zones ------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ int zone_of(const Shared &sh, int b) {
    for (int z = 0; z < sh.nzones; ++z)
        if (b >= sh.zone_lo[z] && b <= sh.zone_hi[z]) return z;
    return -1;
}
__device__ __forceinline__ int slot_of(const Shared &sh, int b) {
    const int z = zone_of(sh, b);
    return z < 0 ? -1 : sh.zone_slot[z] + (b - sh.zone_lo[z]);
}
// the list slice of the hot buckets [ba, bb] (one zone): false if they are not all hot
__device__ __forceinline__ bool slice_of(const Shared &sh, int ba, int bb, unsigned int *from, unsigned int *to) {
    const int z = zone_of(sh, ba);
    if (z < 0 || bb > sh.zone_hi[z] || bb < ba) return false;
    const int s = sh.zone_slot[z] - sh.zone_lo[z];
    *from = sh.lstart[s + ba];
    *to = sh.cursor[s + bb];
    return true;
}

// The plan: sigma_clipped_stats simulated on the prefix sum at bucket resolution; writes the zones it expects the real rounds
// to ask for (thread 0 -> sh.zone_*, unsorted, possibly overlapping) and returns how many, and per round the threshold index
// the MAD search should start from.
__device__ __forceinline__ int plan_zones(Shared &sh, const Geo &g, unsigned int cnt, float mad_guess, unsigned int (&first_guess)[3]) {
    int nz = 0;
    auto add = [&](int lo, int hi) {
        lo = lo < 1 ? 1 : lo;
        hi = hi > kTop - 1 ? kTop - 1 : hi;
        if (lo <= hi && nz < kMaxZones) {
            if (threadIdx.x == 0) {
                sh.zone_lo[nz] = lo;
                sh.zone_hi[nz] = hi;
            }
            ++nz;
        }
    };
    Window w;
    w.lo = 1;
    w.hi = 0x7f7fffffu;
    w.c_lo = 0;
    w.c_le_hi = cnt;
    w.n = cnt;
    first_guess[0] = first_guess[1] = first_guess[2] = kNone;
    const uint32_t gran = 1u << g.shift;
#pragma unroll 1
    for (int it = 0; it < 3; ++it) {
        if (w.n < 3) break;
        const unsigned int mid = w.n / 2, g_hi = w.c_lo + mid, g_lo = w.n % 2 == 0 ? g_hi - 1 : g_hi;
        int b_lo, b_hi;
        find_ranks(sh, g_lo, g_hi, &b_lo, &b_hi);
        add(b_lo - 1, b_hi + 1);
        if (b_hi <= 0 || b_hi >= kTop) break;
        // the median, interpolated inside its bucket
        const unsigned int ex = below(sh, b_hi), in_b = sh.prefix[b_hi] - ex;
        const float frac = ((float)(g_hi - ex) + 0.5f) / (float)(in_b ? in_b : 1u);
        const uint32_t kmed = g.first_key(b_hi) + (uint32_t)(frac * (float)gran);
        const double med = (double)__uint_as_float(kmed);
        DevGeo d;
        if (!make_devgeo(g, med, &d)) break;
        unsigned int guess = kNone;
        if (it == 0) {
            if (mad_guess > 0.0f) {
                const float gi = mad_guess / d.delta;
                guess = gi < 3900.0f ? (unsigned int)gi : 3900u;
            }
        } else {
            guess = first_guess[it - 1];
        }
        const unsigned int first = mad_first(sh, w, d, mid + 1u, guess);
        if (first == kNone) break;
        first_guess[it] = first;
        {
            int bLo, bRo, bLi = d.bm, bRi = d.bm;
            d.ring_of((float)(first + 1u + kRing + kRingPlan) * d.delta, &bLo, &bRo);
            if (first + 1u > (unsigned int)(kRing + kRingPlan)) d.ring_of((float)(first + 1u - kRing - kRingPlan) * d.delta, &bLi, &bRi);
            if (bLi + 1 >= bRi) {
                add(bLo, bRo);
            } else {
                add(bLo, bLi);
                add(bRi, bRo);
            }
        }
        if (it == 2) break;
        const double mad = (double)(((float)first + 0.5f) * d.delta);
        const double sig = fmax(mad * 1.4826, 1e-30);
        const float lo = (float)(med - 3.0 * sig), hi = (float)(med + 3.0 * sig);
        if (!(lo <= hi)) break;
        const float e = (float)kEdgePlan * d.delta;
        add(d.bucket_at(lo - e), d.bucket_at(lo + e));
        add(d.bucket_at(hi - e), d.bucket_at(hi + e));
        const uint32_t klo = lo > 0.0f ? __float_as_uint(lo) : 1u;
        const uint32_t khi = hi > 0.0f ? (__float_as_uint(hi) > 0x7f7fffffu ? 0x7f7fffffu : __float_as_uint(hi)) : 0u;
        w.lo = w.lo > klo ? w.lo : klo;
        w.hi = w.hi < khi ? w.hi : khi;
        if (w.lo > w.hi) break;
        // bucket-resolution counts: half of the bound's own bucket on either side
        const int bl = g.bucket_of(w.lo), bh = g.bucket_of(w.hi);
        unsigned int c_lo = below(sh, bl) + (sh.prefix[bl] - below(sh, bl)) / 2u;
        unsigned int c_hi = sh.prefix[bh] - (sh.prefix[bh] - below(sh, bh)) / 2u;
        if (w.lo <= g.first_key(bl)) c_lo = below(sh, bl);
        if (w.hi >= g.last_key(bh)) c_hi = sh.prefix[bh];
        c_lo = c_lo > w.c_lo ? c_lo : w.c_lo;
        c_hi = c_hi < w.c_le_hi ? c_hi : w.c_le_hi;
        if (c_hi <= c_lo) break;
        w.c_lo = c_lo;
        w.c_le_hi = c_hi;
        w.n = c_hi - c_lo;
    }
    return nz;
}

// zones sorted, merged and numbered (thread 0), slots and slices laid out, the lookup table filled.  False: too many hot
// buckets or keys.
__device__ __forceinline__ bool build_zones(Shared &sh, int nz) {
    const int t = threadIdx.x;
    __syncthreads();
    if (t == 0) {
        // insertion sort by lower end, then merge what touches
        for (int i = 1; i < nz; ++i) {
            const int lo = sh.zone_lo[i], hi = sh.zone_hi[i];
            int j = i;
            while (j > 0 && sh.zone_lo[j - 1] > lo) {
                sh.zone_lo[j] = sh.zone_lo[j - 1];
                sh.zone_hi[j] = sh.zone_hi[j - 1];
                --j;
            }
            sh.zone_lo[j] = lo;
            sh.zone_hi[j] = hi;
        }
        int m = 0;
        for (int i = 0; i < nz; ++i) {
            if (m > 0 && sh.zone_lo[i] <= sh.zone_hi[m - 1] + 1) {
                if (sh.zone_hi[i] > sh.zone_hi[m - 1]) sh.zone_hi[m - 1] = sh.zone_hi[i];
            } else {
                sh.zone_lo[m] = sh.zone_lo[i];
                sh.zone_hi[m] = sh.zone_hi[i];
                ++m;
            }
        }
        int slots = 0;
        unsigned int keys = 0;
        for (int i = 0; i < m; ++i) {
            sh.zone_slot[i] = slots;
            slots += sh.zone_hi[i] - sh.zone_lo[i] + 1;
            keys += sh.prefix[sh.zone_hi[i]] - below(sh, sh.zone_lo[i]);
        }
        sh.nzones = m;
        sh.scal[2] = (unsigned int)slots;
        sh.scal[3] = keys;
    }
    // the lookup table: cold everywhere ...
    {
        uint4 *l = reinterpret_cast<uint4 *>(sh.lut);
        const uint4 ff = {kNone, kNone, kNone, kNone};
#pragma unroll
        for (int i = 0; i < (int)(sizeof(sh.lut) / 16) / kThreads; ++i) l[i * kThreads + t] = ff;
    }
    __syncthreads();
    const int slots = (int)sh.scal[2];
    const unsigned int keys = sh.scal[3];
    if (slots > kSlotCap || keys > (unsigned int)kHotCap || sh.nzones == 0) return false;
    // ... hot where a zone says so; zone z's slice starts where the keys of the zones before it end
    for (int s = t; s < slots; s += kThreads) {
        int z = 0;
        unsigned int base = 0;
        while (z + 1 < sh.nzones && s >= sh.zone_slot[z + 1]) {
            base += sh.prefix[sh.zone_hi[z]] - below(sh, sh.zone_lo[z]);
            ++z;
        }
        const int b = sh.zone_lo[z] + (s - sh.zone_slot[z]);
        const unsigned int at = base + (below(sh, b) - below(sh, sh.zone_lo[z]));
        sh.lut[b] = (unsigned short)s;
        sh.lstart[s] = at;
        sh.cursor[s] = at;
    }
    __syncthreads();
    return true;
}

// ---- select: the ranks r_hi and r_lo (= r_hi or r_hi - 1; 0-based) of sh.tmp[0 .. n), n <= kSelCap (tile_bucket.hpp's
// block_select2 for 256 threads: 8-bit radix descent over the bytes in which the keys differ, one barrier per byte) ----------
__device__ __forceinline__ void block_select2(Shared &sh, unsigned int n, unsigned int r_lo, unsigned int r_hi, uint32_t *k_lo, uint32_t *k_hi) {
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
    uint32_t mine[kMine];
    const uint32_t k0 = sh.tmp[0];
    uint32_t diff = 0;
#pragma unroll
    for (int q = 0; q < kMine; ++q) {
        const unsigned int i = (unsigned int)(q * kThreads + t);
        mine[q] = i < n ? sh.tmp[i] : k0;  // pads repeat a real key and are skipped by index below
        diff |= mine[q] ^ k0;
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) diff |= (uint32_t)__shfl_xor((int)diff, off, 64);
    if (lane == 0) sh.sel_part[wv] = diff;
    sh.sel_hist[0][t] = 0;
    __syncthreads();
    diff = 0;
#pragma unroll
    for (int i = 0; i < kWaves; ++i) diff |= sh.sel_part[i];
    int top = 24;
    while (top > 0 && (diff >> top) == 0) top -= 8;  // the highest byte that varies
    uint32_t mask = top == 24 ? 0u : (0xffffffffu << (top + 8));
    uint32_t prefix = k0 & mask;
    unsigned int rank = r_hi;
    int h = 0;
    for (int shift = top; shift >= 0; shift -= 8) {
        unsigned int *const hist = sh.sel_hist[h], *const next = sh.sel_hist[h == 2 ? 0 : h + 1];
        next[t] = 0;  // last read two bytes ago: a barrier lies in between
#pragma unroll
        for (int q = 0; q < kMine; ++q)
            if ((unsigned int)(q * kThreads + t) < n && (mine[q] & mask) == prefix) atomicAdd(&hist[(mine[q] >> shift) & 255u], 1u);
        __syncthreads();
        // lane l owns digits 4l .. 4l+3; every wave scans for itself
        const unsigned int c0 = hist[4 * lane], c1 = hist[4 * lane + 1], c2 = hist[4 * lane + 2], c3 = hist[4 * lane + 3];
        const unsigned int own = c0 + c1 + c2 + c3;
        const unsigned int incl = wave_scan_incl(own);
        const unsigned int excl = incl - own;
        const bool owner = rank >= excl && rank < incl;
        uint32_t digit = 0;
        unsigned int bel = 0;
        if (owner) {
            const unsigned int r = rank - excl;
            if (r < c0) {
                digit = 4 * lane;
                bel = excl;
            } else if (r < c0 + c1) {
                digit = 4 * lane + 1;
                bel = excl + c0;
            } else if (r < c0 + c1 + c2) {
                digit = 4 * lane + 2;
                bel = excl + c0 + c1;
            } else {
                digit = 4 * lane + 3;
                bel = excl + c0 + c1 + c2;
            }
        }
        const unsigned long long om = __ballot(owner);
        const int ol = om ? (int)__builtin_ctzll(om) : 0;
        digit = (uint32_t)__builtin_amdgcn_readlane((int)digit, ol);
        bel = (unsigned int)__builtin_amdgcn_readlane((int)bel, ol);
        rank -= bel;
        prefix |= digit << shift;
        mask |= 255u << shift;
        h = h == 2 ? 0 : h + 1;
    }
    *k_hi = prefix;
    *k_lo = prefix;
    // `rank` is now r_hi's position among the keys equal to it.  The element before r_hi is the same key unless r_hi is its first
    // occurrence; then it is the largest key below.
    if (r_lo != r_hi && rank == 0) {
        uint32_t best = 0;
#pragma unroll
        for (int q = 0; q < kMine; ++q)
            if ((unsigned int)(q * kThreads + t) < n && mine[q] < prefix) best = best > mine[q] ? best : mine[q];
        best = wave_max(best);
        if (lane == 0) sh.sel_part[kWaves + wv] = best;
        __syncthreads();
#pragma unroll
        for (int i = 0; i < kWaves; ++i) best = best > sh.sel_part[kWaves + i] ? best : sh.sel_part[kWaves + i];
        *k_lo = best;
    }
    __syncthreads();  // tmp and the histograms are free again
}

// ---- the three requests of a round, answered from prefix sum + hot list; false = declined ------------------------------------
// order statistics of the VALUE keys: global ranks g_lo <= g_hi (adjacent or equal)
__device__ __forceinline__ bool select_values_hot(Shared &sh, unsigned int g_lo, unsigned int g_hi, uint32_t *k_lo, uint32_t *k_hi) {
    int b_lo, b_hi;
    find_ranks(sh, g_lo, g_hi, &b_lo, &b_hi);
    if (b_lo <= 0 || b_hi >= kTop) {
        sh.decline = D_MED_CATCHALL;
        return false;
    }
    unsigned int f0, t0, f1 = 0, t1 = 0;
    if (!slice_of(sh, b_lo, b_lo, &f0, &t0) || (b_hi != b_lo && !slice_of(sh, b_hi, b_hi, &f1, &t1))) {
        sh.decline = D_MED_COLD;
        return false;
    }
    const unsigned int n0 = t0 - f0, n = n0 + (t1 - f1);
    if (n > (unsigned int)kSelCap) {
        sh.decline = D_MED_BIG;
        return false;
    }
    for (unsigned int i = threadIdx.x; i < n; i += kThreads) sh.tmp[i] = i < n0 ? sh.list[f0 + i] : sh.list[f1 + (i - n0)];
    __syncthreads();
    const unsigned int ex = below(sh, b_lo);  // (everything between the two buckets is empty: the ranks are adjacent)
    block_select2(sh, n, g_lo - ex, g_hi - ex, k_lo, k_hi);
    return true;
}

// order statistics of the DEVIATION keys |v - median| of the window's candidates: ranks r_lo <= r_hi (adjacent or equal).
// tile_bucket.hpp's guess-and-verify: the first threshold t* with F~(t*) >= r_hi + 1 places two rings of buckets; their keys
// come from the hot list; afterwards the guess is PROVED (the largest deviation possible between the rings <= the smaller
// selected one, the smallest possible outside >= the larger).
__device__ __forceinline__ bool select_devs_hot(Shared &sh, const Geo &g, const Window &w, double median, unsigned int r_lo, unsigned int r_hi,
                                                unsigned int guess, uint32_t *d_lo, uint32_t *d_hi) {
    DevGeo d;
    if (!make_devgeo(g, median, &d)) {
        sh.decline = D_DEV_BM;
        return false;
    }
    const unsigned int first = mad_first(sh, w, d, r_hi + 1u, guess);
    if (first == kNone) {
        sh.decline = D_DEV_NOFIRST;
        return false;
    }
    int bLo, bRo, bLi = 0, bRi = 0;
    d.ring_of((float)(first + 1u + kRing) * d.delta, &bLo, &bRo);
    const bool has_inner_t = first + 1u > (unsigned int)kRing;
    if (has_inner_t) d.ring_of((float)(first + 1u - kRing) * d.delta, &bLi, &bRi);
    const bool has_inner = has_inner_t && bLi + 1 <= bRi - 1;  // inner buckets: strictly between the two rings
    const unsigned int c_in = has_inner ? wprefix(sh, w, bRi) - wprefix(sh, w, bLi + 1) : 0u;
    const unsigned int n_cand = wprefix(sh, w, bRo + 1) - wprefix(sh, w, bLo) - c_in;
    if (!(bLo > 0 && bRo < kTop && c_in <= r_lo && c_in + n_cand >= r_hi + 1u && n_cand <= (unsigned int)kSelCap)) {
        sh.decline = D_DEV_SHAPE;
        return false;
    }
    const uint32_t ga_lo = g.first_key(bLo) > w.lo ? g.first_key(bLo) : w.lo;
    const uint32_t ga_hi = g.last_key(bRo) < w.hi ? g.last_key(bRo) : w.hi;
    uint32_t x_lo = 1, x_hi = 0;  // the excluded inner run of keys (empty)
    unsigned int f0, t0, f1 = 0, t1 = 0;
    bool hot;
    if (has_inner) {
        x_lo = g.first_key(bLi + 1);
        x_hi = g.last_key(bRi - 1);
        hot = slice_of(sh, bLo, bLi, &f0, &t0) && slice_of(sh, bRi, bRo, &f1, &t1);
    } else {
        hot = slice_of(sh, bLo, bRo, &f0, &t0);
    }
    if (!hot) {
        sh.decline = D_DEV_COLD;
        return false;
    }
    const unsigned int n0 = t0 - f0, n = n0 + (t1 - f1);
    if (n > (unsigned int)kSelCap) {
        sh.decline = D_DEV_BIG;
        return false;
    }
    // the slices as deviation keys; a key outside the window sorts last (0xffffffff) and is never selected
    unsigned int in = 0, z = 0;
    for (unsigned int i = threadIdx.x; i < n; i += kThreads) {
        const uint32_t k = i < n0 ? sh.list[f0 + i] : sh.list[f1 + (i - n0)];
        const bool ok = k >= ga_lo && k <= ga_hi;
        sh.tmp[i] = ok ? dev_key(k, median) : kNone;
        in += ok ? 1u : 0u;
    }
    block_reduce2<OP_SUM, OP_SUM>(sh, in, z);  // (its barriers also publish tmp)
    if (in != n_cand) {
        sh.decline = D_DEV_COUNT;
        return false;
    }
    uint32_t sel_lo, sel_hi;
    block_select2(sh, n, r_lo - c_in, r_hi - c_in, &sel_lo, &sel_hi);
    // the proof: inner deviations <= sel_lo, outer deviations >= sel_hi
    uint32_t t_in = 0, t_out = kNone;
    if (has_inner) {
        const uint32_t ka = x_lo > w.lo ? x_lo : w.lo, kb = x_hi < w.hi ? x_hi : w.hi;
        if (ka <= kb) {
            const uint32_t da = dev_key(ka, median), db = dev_key(kb, median);
            t_in = da > db ? da : db;
        }
    }
    if (ga_lo > w.lo) t_out = dev_key(ga_lo - 1u, median);  // (ga_lo - 1 lies left of the median: bLo <= bm)
    if (ga_hi < w.hi) {
        const uint32_t dd = dev_key(ga_hi + 1u, median);
        t_out = t_out < dd ? t_out : dd;
    }
    if (!(t_in <= sel_lo && sel_hi <= t_out)) {
        sh.decline = D_DEV_PROOF;
        return false;
    }
    *d_lo = sel_lo;
    *d_hi = sel_hi;
    return true;
}

// candidates with key < klo and with key <= khi, exactly
__device__ __forceinline__ bool count_edges(Shared &sh, const Geo &g, uint32_t klo, uint32_t khi, unsigned int *lt_lo, unsigned int *le_hi) {
    const int bl = g.bucket_of(klo), bh = g.bucket_of(khi);
    const unsigned int bel_l = below(sh, bl), in_l = sh.prefix[bl] - bel_l, bel_h = below(sh, bh), in_h = sh.prefix[bh] - bel_h;
    // a bound at (or beyond) its bucket's end, or an empty bucket, needs no keys
    const bool easy_l = in_l == 0 || klo <= g.first_key(bl), easy_h = in_h == 0 || khi >= g.last_key(bh);
    unsigned int fl = 0, tl = 0, fh = 0, th = 0;
    if (!easy_l) {
        if (bl <= 0 || bl >= kTop) {
            sh.decline = D_EDGE_CATCHALL;
            return false;
        }
        if (!slice_of(sh, bl, bl, &fl, &tl)) {
            sh.decline = D_EDGE_COLD;
            return false;
        }
    }
    if (!easy_h) {
        if (bh <= 0 || bh >= kTop) {
            sh.decline = D_EDGE_CATCHALL;
            return false;
        }
        if (!slice_of(sh, bh, bh, &fh, &th)) {
            sh.decline = D_EDGE_COLD;
            return false;
        }
    }
    unsigned int a = 0, b = 0;
    for (unsigned int i = fl + threadIdx.x; i < tl; i += kThreads) a += sh.list[i] < klo ? 1u : 0u;
    for (unsigned int i = fh + threadIdx.x; i < th; i += kThreads) b += sh.list[i] <= khi ? 1u : 0u;
    if (!easy_l || !easy_h) block_reduce2<OP_SUM, OP_SUM>(sh, a, b);
    *lt_lo = bel_l + (easy_l ? 0u : a);
    *le_hi = in_h == 0 ? bel_h : (khi >= g.last_key(bh) ? sh.prefix[bh] : bel_h + b);
    return true;
}

// sigma_clipped_stats(values, 3.0, 2) (sigma_clip.rs:4-34) on prefix sum + hot list; false = declined
__device__ __forceinline__ bool run_rounds(Shared &sh, const Geo &g, unsigned int cnt, const unsigned int (&first_guess)[3], double *median_out,
                                           double *sigma_out) {
    constexpr double kMadToSigma = 1.4826;
    Window w;
    w.lo = 1;
    w.hi = 0x7f7fffffu;
    w.c_lo = 0;
    w.c_le_hi = cnt;
    w.n = cnt;
    double median = 0.0, sigma = 1.0;
#pragma unroll 1
    for (int it = 0; it < 3; ++it) {  // 2 clipping iterations + the final statistics (sigma_clip.rs:7-33)
        if (it < 2 && w.n < 3) continue;  // `if values.len() < 3 { break }`: no more clipping, the final statistics still run
        if (w.n == 0) {                   // sigma_clip.rs:26-28
            median = 0.0;
            sigma = 1.0;
            break;
        }
        // exact_median_mut (median.rs:27-44): element n/2, averaged in f64 with the largest element below it for even n
        const unsigned int mid = w.n / 2;
        uint32_t ka, kb;
        if (!select_values_hot(sh, w.c_lo + (w.n % 2 == 0 ? mid - 1 : mid), w.c_lo + mid, &ka, &kb)) return false;
        median = w.n % 2 == 0 ? ((double)__uint_as_float(ka) + (double)__uint_as_float(kb)) / 2.0 : (double)__uint_as_float(kb);
        // median_f32_mut of the deviations (median.rs:46-63): f32 average for even n
        uint32_t da, db;
        if (!select_devs_hot(sh, g, w, median, w.n % 2 == 0 ? mid - 1 : mid, mid, first_guess[it], &da, &db)) return false;
        const float mad_f32 = w.n % 2 == 0 ? (__uint_as_float(da) + __uint_as_float(db)) / 2.0f : __uint_as_float(db);
        const double sig = fmax((double)mad_f32 * kMadToSigma, 1e-30);
        if (it == 2) {
            sigma = sig;
            break;
        }
        // retain v in [lo, hi] (sigma_clip.rs:19-23); kappa = 3.0f32 as f64
        const float lo = (float)(median - 3.0 * sig), hi = (float)(median + 3.0 * sig);
        if (!(lo <= hi)) {  // NaN bounds: nothing is retained
            w.n = 0;
            w.c_lo = w.c_le_hi = 0;
            w.lo = 1;
            w.hi = 0;
            continue;
        }
        const uint32_t klo = lo > 0.0f ? __float_as_uint(lo) : 1u;
        const uint32_t khi = hi > 0.0f ? (__float_as_uint(hi) > 0x7f7fffffu ? 0x7f7fffffu : __float_as_uint(hi)) : 0u;
        w.lo = w.lo > klo ? w.lo : klo;
        w.hi = w.hi < khi ? w.hi : khi;
        if (w.lo > w.hi) {
            w.n = 0;
            w.c_lo = w.c_le_hi = 0;
            continue;
        }
        if (!count_edges(sh, g, w.lo, w.hi, &w.c_lo, &w.c_le_hi)) return false;
        w.n = w.c_le_hi - w.c_lo;
    }
    *median_out = median;
    *sigma_out = sigma;
    return true;
}

// ---- streaming a tile -------------------------------------------------------------------------------------------------------
struct TileRect {
    const float *img;
    int64_t ld;
    int y0, y1, x0, x1;
    bool vec;  // x0, x1, ld multiples of 4 and img 16-byte aligned: rows as float4
};
__device__ __forceinline__ uint32_t make_key(const ab_pixel_xf &xf, float raw) {
    const float v = ab_px(xf, raw);
    return (__builtin_isfinite(v) && v > 1e-7f) ? __float_as_uint(v) : 0u;  // star_detection.rs:56
}
// f(key) for every pixel of the tile (0 = not a candidate); the order is the same for every pass
template <class F>
__device__ __forceinline__ void stream_tile(const TileRect &r, const ab_pixel_xf &xf, F f) {
    const int t = threadIdx.x;
    if (r.vec) {
        constexpr int U = 8;
        const int lane = t & 63, wv = t >> 6, c = r.x0 + 4 * lane;
        const bool col_ok = c < r.x1;
        const float *p = r.img + c;
#pragma unroll 1
        for (int r0 = r.y0 + wv; r0 < r.y1; r0 += kWaves * U) {
            float4 raw[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int row = r0 + kWaves * u;
                raw[u] = (col_ok && row < r.y1) ? *reinterpret_cast<const float4 *>(p + (int64_t)row * r.ld)
                                                : make_float4(__builtin_nanf(""), __builtin_nanf(""), __builtin_nanf(""), __builtin_nanf(""));
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                f(make_key(xf, raw[u].x));
                f(make_key(xf, raw[u].y));
                f(make_key(xf, raw[u].z));
                f(make_key(xf, raw[u].w));
            }
        }
    } else {
        constexpr int U = 16;
        const int c = r.x0 + t;
        const bool col_ok = c < r.x1;
        const float *p = r.img + c;
#pragma unroll 1
        for (int r0 = r.y0; r0 < r.y1; r0 += U) {
            float raw[U];
#pragma unroll
            for (int u = 0; u < U; ++u) raw[u] = (col_ok && r0 + u < r.y1) ? p[(int64_t)(r0 + u) * r.ld] : __builtin_nanf("");
#pragma unroll
            for (int u = 0; u < U; ++u) f(make_key(xf, raw[u]));
        }
    }
}

struct TileResult {
    double median, sigma;
    int valid;     // 1 if the tile had >= 8 valid pixels
    int declined;  // != 0: the tile needs tile_bucket.hpp (median / sigma / valid are not set)
};

// where to zoom (efficiency only: any geometry is exact): the quartiles of 4 x 64 sample pixels, as tile_bucket.hpp
__device__ __forceinline__ Geo zoom_from_sample(Shared &sh, const TileRect &r, const ab_pixel_xf &xf, float *mad_guess) {
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
    const int h = r.y1 - r.y0, wd = r.x1 - r.x0;
    // sample t: row t * h / 256, a column that walks across the tile
    const int sr = r.y0 + (int)(((unsigned int)t * (unsigned int)h) >> 8), sc = r.x0 + (int)(((unsigned int)(t * 67 + 13) % 256u * (unsigned int)wd) >> 8);
    const uint32_t sk = make_key(xf, r.img[(int64_t)sr * r.ld + sc]);
    const uint32_t sorted = wave_sort64(sk);
    const int nz = __builtin_popcountll(__builtin_amdgcn_ballot_w64(sorted == 0u)), nv = 64 - nz;
    const uint32_t q1 = (uint32_t)__shfl((int)sorted, nz + nv / 4 < 64 ? nz + nv / 4 : 63, 64);
    const uint32_t q2 = (uint32_t)__shfl((int)sorted, nz + nv / 2 < 64 ? nz + nv / 2 : 63, 64);
    const uint32_t q3 = (uint32_t)__shfl((int)sorted, nz + (3 * nv) / 4 < 64 ? nz + (3 * nv) / 4 : 63, 64);
    if (lane == 0) {
        sh.part[4 * wv + 0] = q1;
        sh.part[4 * wv + 1] = q2;
        sh.part[4 * wv + 2] = q3;
        sh.part[4 * wv + 3] = nv >= 8 ? 1u : 0u;
    }
    __syncthreads();
    float s1 = 0.0f, s2 = 0.0f, s3 = 0.0f, sn = 0.0f;
#pragma unroll
    for (int i = 0; i < kWaves; ++i) {
        const bool ok = sh.part[4 * i + 3] != 0u;
        s1 += ok ? __uint_as_float(sh.part[4 * i + 0]) : 0.0f;
        s2 += ok ? __uint_as_float(sh.part[4 * i + 1]) : 0.0f;
        s3 += ok ? __uint_as_float(sh.part[4 * i + 2]) : 0.0f;
        sn += ok ? 1.0f : 0.0f;
    }
    __syncthreads();
    Geo g;
    *mad_guess = 0.0f;
    uint32_t kc = 0x3f000000u;  // no usable sample: coarse buckets over everything
    int shift = 19;
    if (sn > 0.0f) {
        const float Q1 = s1 / sn, Q2 = s2 / sn, Q3 = s3 / sn;
        const float sig = (Q3 - Q1) * (1.0f / 1.349f);
        *mad_guess = 0.6745f * sig;
        kc = __float_as_uint(Q2);
        const float up = Q2 + sig;
        const uint32_t wkeys = (sig > 0.0f && up < 3.0e38f) ? __float_as_uint(up) - __float_as_uint(Q2) : 0u;  // sigma in key units
        const uint32_t bk = wkeys / 300u;
        shift = bk >= 1u ? 31 - __builtin_clz(bk) : 0;
        shift = shift > 19 ? 19 : shift;
    }
    const uint32_t gran = 1u << shift, half = (uint32_t)((kBuckets - 2) / 2) << shift;
    g.shift = shift;
    g.zlo = kc > half + gran ? kc - half : gran;
    g.base = g.zlo - gran;
    return g;
}

// the whole tile; every thread returns the same result
__device__ __forceinline__ TileResult tile_stats(Shared &sh, const TileRect &r, const ab_pixel_xf &xf) {
    const int t = threadIdx.x;
    TileResult res = {0.0, 1.0, 0, 0};
#ifdef AB_TILE_TIMING
    if (t == 0) {
        for (int i = 0; i < 8; ++i) sh.t_phase[i] = 0;
        sh.t_mark = clock64();
    }
#endif
    if (t == 0) sh.decline = D_NONE;
#pragma unroll
    for (int i = 0; i < kPer; i += 4) *reinterpret_cast<uint4 *>(&sh.prefix[t * kPer + i]) = make_uint4(0u, 0u, 0u, 0u);
    float mad_guess;
    const Geo g = zoom_from_sample(sh, r, xf, &mad_guess);  // (its barriers also publish the cleared histogram)
    TS_MARK(sh, 0);
    // ---- pass 1: the histogram ----
    stream_tile(r, xf, [&](uint32_t k) {
        if (k) atomicAdd(&sh.prefix[g.bucket_of(k)], 1u);
    });
    __syncthreads();
    TS_MARK(sh, 1);
    scan_prefix(sh);
    const unsigned int cnt = sh.prefix[kTop];
    TS_MARK(sh, 2);
    if (cnt < 8) return res;  // star_detection.rs:61
    res.valid = 1;
    // ---- plan, zones ----
    unsigned int first_guess[3];
    const int nz = plan_zones(sh, g, cnt, mad_guess, first_guess);
    TS_MARK(sh, 3);
    if (!build_zones(sh, nz)) {
        res.declined = sh.nzones == 0 ? D_ZONES : D_HOT_OVERFLOW;
        return res;
    }
    TS_MARK(sh, 4);
    // ---- pass 2: the hot keys, bucket-sorted ----
    stream_tile(r, xf, [&](uint32_t k) {
        if (k) {
            const unsigned int s = sh.lut[g.bucket_of(k)];
            if (s != 0xffffu) {
                const unsigned int at = atomicAdd(&sh.cursor[s], 1u);
                if (at < (unsigned int)kHotCap) sh.list[at] = k;
            }
        }
    });
    __syncthreads();
    TS_MARK(sh, 5);
    // ---- the rounds ----
    if (!run_rounds(sh, g, cnt, first_guess, &res.median, &res.sigma)) {
        __syncthreads();
        res.declined = sh.decline ? sh.decline : 1;
    }
    TS_MARK(sh, 6);
    return res;
}

}  // namespace ts
