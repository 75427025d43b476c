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
//   pass 1  the tile streams through once (float4 rows, HBM) into a 4096-bucket LDS histogram.  The buckets are uniform in
//           the RAW pixel value (one fma + clamp + convert per pixel), zoomed on the quartiles of a 256-pixel sample; the
//           percentile normalisation the registration path applies on load (ab_px, f64) is monotone, so order statistics
//           can be taken on raw values and normalised afterwards -- the passes never evaluate it;
//   plan    the three rounds of sigma_clipped_stats are SIMULATED on the histogram alone, at bucket resolution: where the
//           median's bucket will be, where the two rings of the MAD's candidates will be, where the clipping bounds will
//           fall -- each with a margin of a few buckets.  Those bucket ranges (a few dozen buckets, 1 - 3 thousand pixels)
//           are the HOT ZONES;
//   pass 2  the tile streams through a second time (L2 / Infinity Cache) and the raw values of hot buckets are appended to
//           their zone's segment of an LDS list (segment sizes are known exactly from the prefix sum);
//   rounds  ONE wave runs the three rounds for real on prefix sum + list (the other three leave): the median's bucket is
//           picked out of its zone, the MAD's candidates are two rings out of theirs (guess-and-verify with the same proof
//           as tile_bucket.hpp's fast path), the exact number of pixels cut by a clipping bound is a prefix-sum entry plus a
//           count inside ONE bucket.  Only here are values normalised (exactly, ab_px) -- a few hundred per request.
//
// Every answer is either exact or the tile is DECLINED (a request outside the hot zones, a list that would overflow, a
// guess whose proof fails, a rank inside a catch-all bucket, a sample without spread): declined tiles are appended to a
// list that tile_background_bucket_kernel works off afterwards.  On sky tiles nothing is declined; heavy ties, flat and
// multi-modal tiles are (tests/test_gpu_tile_stats.py compares all of them with the oracle bit for bit, through both kernels).
//
// Why raw buckets are sound.  xf = ab_px is non-decreasing, so sorting the candidates by raw value sorts their normalised
// values too (ties in any order): bucket b holds the ranks [prefix[b-1], prefix[b]) of the NORMALISED order, a clipping
// window [lo, hi] on normalised values is the rank range [#{v < lo}, #{v <= hi}), and a bucket's normalised values lie
// between xf(its smallest raw value) and xf(its largest) -- which is all the counting and the proofs below use.
//
// 256 threads, ~39 KB of LDS and ~100 VGPRs per workgroup: four tiles per CU in flight, other kernels' waves beside them.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

#include "ab_common.hpp"
#include "tile_bucket.hpp"

namespace ts {

using tb::dev_key;
using tb::wave_scan_incl;
using tb::wave_sort64;
using tb::Window;

constexpr int kThreads = 256;
constexpr int kWaves = kThreads / 64;
constexpr int kBuckets = 4096;
constexpr int kTop = kBuckets - 1;         // catch-all above the zoom window (bucket 0: below it)
constexpr int kPer = kBuckets / kThreads;  // buckets a thread owns in the scan
constexpr int kHotCap = 3072;              // values the hot list holds
constexpr int kTabCap = 384;               // bucket-boundary table entries (hot buckets + one either side of every zone)
constexpr int kSelCap = 1024;              // keys one select works on
constexpr int kMaxZones = 16;
constexpr int kRing = 2;        // the MAD's candidates: buckets within kRing bucket widths of median -+ t* (as tile_bucket.hpp)
constexpr int kRingPlan = 3;    // + what the plan may be off by
constexpr int kEdgePlan = 12;   // a clipping bound: predicted to +- this many bucket widths
#ifndef TS_PER_SIGMA
#define TS_PER_SIGMA 480.0f
#endif
constexpr float kPerSigma = TS_PER_SIGMA;  // buckets per sigma of the sample: 4094 buckets = +- 4.3 sigma (the sample's sigma is good to ~7 %; at 550 one tile in a hundred had a clipping bound outside)
constexpr unsigned int kNone = 0xffffffffu;

// reasons a tile is declined (debug / statistics; the fallback does not care)
enum Decline {
    D_NONE = 0, D_ZONES, D_HOT_OVERFLOW, D_MED_CATCHALL, D_MED_COLD, D_MED_BIG, D_DEV_BM, D_DEV_NOFIRST, D_DEV_SHAPE, D_DEV_COLD,
    D_DEV_BIG, D_DEV_COUNT, D_DEV_PROOF, D_EDGE_COLD, D_EDGE_CATCHALL, D_SAMPLE
};

struct alignas(16) Shared {
    unsigned int prefix[kBuckets + kThreads];  // bucket counts, then their inclusive prefix sum | one scratch word per thread (where the
                                               // passes send what they do not count; later the select's 256-bin histogram)
    float list[kHotCap + kThreads];            // the raw values of hot buckets, one segment per zone | scratch
    union {
        unsigned char lut[kBuckets];  // pass 2: bucket -> zone + 1 (0: cold)
        unsigned int tmp[kSelCap];    // afterwards: the keys a select works on
    };
    unsigned int zcur[kMaxZones + kThreads];  // pass 2: next free position of the zone's segment | scratch
    uint32_t tab_first[kTabCap], tab_last[kTabCap];  // normalised keys of the smallest / largest raw value of the buckets around the zones
    unsigned int part[4 * kWaves];
    unsigned int zone_at[kMaxZones + 1];  // zone z's segment: list[zone_at[z] .. zone_at[z + 1])
    int zone_lo[kMaxZones], zone_hi[kMaxZones];
    int zone_tab[kMaxZones];              // table index of bucket zone_lo[z] - 1
    int nzones;
    int decline;
    unsigned int ncand;  // pass 2: entries appended to the tile's candidate list (CandSink)
#ifdef AB_TILE_TIMING
    long long t_phase[16];  // 0 sample 1 pass1 2 scan 3 plan 4 zones+lut 5 pass2 6 rest of rounds; rounds: 8 v:find+gather 9 v:select 10 d:geo 11 d:first 12 d:gather 13 d:select 14 d:proof 15 edges
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

// floats of either sign as monotone unsigned integers (for stepping to the neighbouring float and for sorting the sample)
__device__ __forceinline__ uint32_t ord_of(float x) {
    const uint32_t b = __float_as_uint(x);
    return b ^ ((b >> 31) ? 0xffffffffu : 0x80000000u);
}
__device__ __forceinline__ float from_ord(uint32_t u) { return __uint_as_float(u ^ ((u >> 31) ? 0x80000000u : 0xffffffffu)); }

// ---- geometry: bucket 0 = values below the zoom window | buckets 1 .. 4094 uniform in the raw value | bucket 4095 = above ----
struct Geo {
    float scale, off;    // bucket(v) = clamp(fma(v, scale, off), 0, 4095) truncated: non-decreasing in v
    float r_min, r_max;  // the candidates (star_detection.rs:56 on the normalised value): r_min <= raw <= r_max
    ab_pixel_xf xf;
    double range;        // 1 / xf.inv: the approximate inverse of the normalisation
    __device__ __forceinline__ int bucket(float v) const { return (int)__builtin_amdgcn_fmed3f(__builtin_fmaf(v, scale, off), 0.0f, (float)kTop); }
    __device__ __forceinline__ bool is_cand(float v) const { return v >= r_min && v <= r_max; }
    __device__ __forceinline__ uint32_t norm_key(float raw) const { return __float_as_uint(ab_px(xf, raw)); }  // (candidates: positive)
    __device__ __forceinline__ float unnorm(float x) const { return xf.on ? (float)((double)x * range + xf.lo) : x; }  // approximately
    __device__ __forceinline__ int bucket_at_norm(float x) const { return bucket(unnorm(x)); }
    // the smallest float of bucket b (1 <= b <= 4095): an estimate from the inverse map, then stepped float by float -- exact
    // with respect to bucket(); false if the steps do not settle (degenerate geometry)
    __device__ __forceinline__ bool first_raw(int b, float *out) const {
        uint32_t u = ord_of(((float)b - off) / scale);
        int guard = 0;
        while (bucket(from_ord(u)) >= b && ++guard < 64) --u;
        while (bucket(from_ord(u)) < b && ++guard < 128) ++u;
        *out = from_ord(u);
        return guard < 128 && bucket(from_ord(u)) >= b && bucket(from_ord(u - 1u)) < b;
    }
    // the normalised keys that bound bucket b's pixels: xf(smallest raw value), xf(largest raw value).  b = 0 / 4095: the
    // candidates' own bounds.
    __device__ __forceinline__ bool first_norm(int b, uint32_t *key) const {
        if (b <= 0) {
            *key = 0x33d6bf96u;  // the smallest candidate: v > 1e-7f
            return true;
        }
        float r;
        if (!first_raw(b, &r)) return false;
        *key = norm_key(r < r_min ? r_min : r);
        return true;
    }
    __device__ __forceinline__ bool last_norm(int b, uint32_t *key) const {
        if (b >= kTop) {
            *key = norm_key(r_max);
            return true;
        }
        float r;
        if (!first_raw(b + 1, &r)) return false;
        r = from_ord(ord_of(r) - 1u);
        *key = norm_key(r < r_min ? r_min : r);
        return true;
    }
};

// ---- workgroup pieces ---------------------------------------------------------------------------------------------------------
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

// ---- one-wave pieces (the plan and the rounds run on wave 0 alone: no barriers, LDS operations of a wave retire in order) ------
__device__ __forceinline__ void wave_sync() { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); }
__device__ __forceinline__ int first_lane(unsigned long long m) { return (int)__builtin_ctzll(m); }
__device__ __forceinline__ unsigned int lanes_below(unsigned long long m) {  // set bits of m below this lane
    return __builtin_amdgcn_mbcnt_hi((unsigned int)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned int)m, 0u));
}
__device__ __forceinline__ unsigned int below(const Shared &sh, int b) { return b <= 0 ? 0u : sh.prefix[b - 1]; }  // pixels in buckets < b

// the bucket of global rank g (< the candidate count): bucket b covers ranks [prefix[b-1], prefix[b])
__device__ __forceinline__ int wfind_rank(const Shared &sh, unsigned int g) {
    const int lane = threadIdx.x & 63;
    const unsigned long long m1 = __builtin_amdgcn_ballot_w64(g < sh.prefix[64 * lane + 63]);
    const int L = m1 ? first_lane(m1) : 63;
    const unsigned long long m2 = __builtin_amdgcn_ballot_w64(g < sh.prefix[64 * L + lane]);
    return 64 * L + (m2 ? first_lane(m2) : 63);
}

// window candidates in buckets < b  (b in [0, 4096])
__device__ __forceinline__ unsigned int wprefix(const Shared &sh, const Window &w, int b) {
    const unsigned int p = below(sh, b);
    const unsigned int q = p < w.c_lo ? w.c_lo : (p > w.c_le_hi ? w.c_le_hi : p);
    return q - w.c_lo;
}

// The zones as the rounds see them: lane z holds zone z (bucket range, list segment, where its rows of the boundary table start).
// Lookups are a compare + ballot; walking the zone arrays in LDS cost a dependent LDS round trip per zone and field (~2 400 cycles
// per lookup, thirty lookups per tile).
struct ZReg {
    int lo, hi, tab;       // buckets [lo, hi]; table index of bucket lo - 1 (-1: no table rows)
    unsigned int at, end;  // list[at .. end)
    int n;                 // zones
    __device__ __forceinline__ void load(const Shared &sh) {
        const int lane = threadIdx.x & 63;
        n = sh.nzones;
        const bool have = lane < n;
        lo = have ? sh.zone_lo[lane] : 0;
        hi = have ? sh.zone_hi[lane] : -1;
        tab = have ? sh.zone_tab[lane] : -1;
        at = have ? sh.zone_at[lane] : 0u;
        end = have ? sh.zone_at[lane + 1] : 0u;
    }
    __device__ __forceinline__ int zone_of(int b) const {
        const unsigned long long m = __builtin_amdgcn_ballot_w64(b >= lo && b <= hi);
        return m ? first_lane(m) : -1;
    }
    __device__ __forceinline__ int lo_of(int z) const { return __builtin_amdgcn_readlane(lo, z); }
    __device__ __forceinline__ int hi_of(int z) const { return __builtin_amdgcn_readlane(hi, z); }
    __device__ __forceinline__ unsigned int at_of(int z) const { return (unsigned int)__builtin_amdgcn_readlane((int)at, z); }
    __device__ __forceinline__ unsigned int end_of(int z) const { return (unsigned int)__builtin_amdgcn_readlane((int)end, z); }
    // the table row of bucket b (in or next to a zone), -1 if it has none
    __device__ __forceinline__ int tab_index(int b) const {
        const unsigned long long m = __builtin_amdgcn_ballot_w64(tab >= 0 && b >= lo - 1 && b <= hi + 1);
        if (!m) return -1;
        const int z = first_lane(m);
        return __builtin_amdgcn_readlane(tab, z) + (b - __builtin_amdgcn_readlane(lo, z) + 1);
    }
};
// first_norm / last_norm through the table (a direct call costs ~200 dependent instructions -- a division, float-by-float steps,
// the f64 normalisation)
__device__ __forceinline__ bool tfirst(const Shared &sh, const ZReg &zr, const Geo &g, int b, uint32_t *key) {
    const int i = zr.tab_index(b);
    if (i < 0) return g.first_norm(b, key);
    *key = sh.tab_first[i];
    return *key != 0u;
}
__device__ __forceinline__ bool tlast(const Shared &sh, const ZReg &zr, const Geo &g, int b, uint32_t *key) {
    const int i = zr.tab_index(b);
    if (i < 0) return g.last_norm(b, key);
    *key = sh.tab_last[i];
    return *key != 0u;
}

struct DevGeo {
    float mf;     // the median as f32 (normalised)
    float delta;  // width of the median's bucket in normalised units
    int bm;       // the bucket of the median
};
// the buckets that hold median - t and median + t (normalised), either side of bm
__device__ __forceinline__ void ring_of(const Geo &g, const DevGeo &d, float t, int *bl, int *br) {
    const int l = g.bucket_at_norm(d.mf - t), r = g.bucket_at_norm(d.mf + t);
    *bl = l < d.bm ? l : d.bm;
    *br = r > d.bm ? r : d.bm;
}
// bm = the last bucket whose smallest normalised value is <= median
__device__ __forceinline__ bool make_devgeo(const Shared &sh, const ZReg &zr, const Geo &g, double median, DevGeo *d) {
    d->mf = (float)median;
    int b = g.bucket_at_norm(d->mf), guard = 0;
    uint32_t k;
    auto first_le = [&](int bb, bool *le) {
        if (!tfirst(sh, zr, g, bb, &k)) return false;
        *le = (double)__uint_as_float(k) <= median;
        return true;
    };
    bool le;
    while (b > 0) {
        if (!first_le(b, &le)) return false;
        if (le || ++guard > 8) break;
        --b;
    }
    while (b < kTop) {
        if (!first_le(b + 1, &le)) return false;
        if (!le || ++guard > 16) break;
        ++b;
    }
    d->bm = b;
    if (b <= 0 || b >= kTop || guard > 16) return false;
    uint32_t k0, k1;
    if (!tfirst(sh, zr, g, b, &k0) || !tfirst(sh, zr, g, b + 1, &k1)) return false;
    d->delta = __uint_as_float(k1) - __uint_as_float(k0);
    return d->delta > 0.0f;
}

// F~(t) = window candidates in the buckets that meet [median - t, median + t]: the first threshold index i (t = (i + 1) delta)
// with F~ >= need; kNone if none of the 4096 does.  The 256 thresholds around `guess` first (the window's answer is THE first
// iff it lies strictly inside the window: F~ is monotone), else all 4096.
__device__ __forceinline__ unsigned int wmad_first(const Shared &sh, const Geo &g, const Window &w, const DevGeo &d, unsigned int need, unsigned int guess) {
    const int lane = threadIdx.x & 63;
    auto reaches = [&](unsigned int i) {
        int bl, br;
        ring_of(g, d, (float)(i + 1u) * d.delta, &bl, &br);
        return wprefix(sh, w, br + 1) - wprefix(sh, w, bl) >= need;
    };
    if (guess != kNone) {
        const unsigned int gi = guess < 3900u ? guess : 3900u, i_lo = gi > 96u ? gi - 96u : 0u;
#pragma unroll 1
        for (int j = 0; j < 4; ++j) {
            const unsigned long long m = __builtin_amdgcn_ballot_w64(reaches(i_lo + 64u * j + lane));
            if (m) {
                const unsigned int first = i_lo + 64u * j + (unsigned int)first_lane(m);
                if (first > i_lo || i_lo == 0u) return first;
                break;
            }
        }
    }
#pragma unroll 1
    for (int j = 0; j < kBuckets / 64; ++j) {
        const unsigned long long m = __builtin_amdgcn_ballot_w64(reaches(64u * j + lane));
        if (m) return 64u * j + (unsigned int)first_lane(m);
    }
    return kNone;
}

// The plan: sigma_clipped_stats simulated on the prefix sum at bucket resolution; lane 0 writes the zones it expects the real
// rounds to ask for (unsorted, possibly overlapping); returns how many, and per round the threshold index the MAD search should
// start from.
__device__ __forceinline__ int plan_zones(Shared &sh, const Geo &g, unsigned int cnt, float mad_guess, unsigned int (&first_guess)[3]) {
    int nz = 0;
    auto add = [&](int lo, int hi) {
        lo = lo < 1 ? 1 : lo;
        hi = hi > kTop - 1 ? kTop - 1 : hi;
        if (lo <= hi && nz < kMaxZones) {
            if ((threadIdx.x & 63) == 0) {
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
#pragma unroll 1
    for (int it = 0; it < 3; ++it) {
        if (w.n < 3) break;
        const unsigned int mid = w.n / 2, g_hi = w.c_lo + mid, g_lo = w.n % 2 == 0 ? g_hi - 1 : g_hi;
        const int b_hi = wfind_rank(sh, g_hi), b_lo = g_lo == g_hi ? b_hi : wfind_rank(sh, g_lo);
        add(b_lo - 1, b_hi + 1);
        if (b_hi <= 0 || b_hi >= kTop) break;
        // the median, interpolated inside its bucket
        uint32_t k0, k1;
        if (!g.first_norm(b_hi, &k0) || !g.first_norm(b_hi + 1, &k1)) break;
        const unsigned int ex = below(sh, b_hi), in_b = sh.prefix[b_hi] - ex;
        const float frac = ((float)(g_hi - ex) + 0.5f) / (float)(in_b ? in_b : 1u);
        const float v0 = __uint_as_float(k0), v1 = __uint_as_float(k1);
        const double med = (double)(v0 + frac * (v1 - v0));
        DevGeo d;
        d.mf = (float)med;
        d.bm = b_hi;
        d.delta = v1 - v0;
        if (!(d.delta > 0.0f)) break;
        unsigned int guess = kNone;
        if (it == 0) {
            if (mad_guess > 0.0f) {
                const float gi = mad_guess / d.delta;
                guess = gi < 3900.0f ? (unsigned int)gi : 3900u;
            }
        } else {
            guess = first_guess[it - 1];
        }
        const unsigned int first = wmad_first(sh, g, w, d, mid + 1u, guess);
        if (first == kNone) break;
        first_guess[it] = first;
        {
            int bLo, bRo, bLi = d.bm, bRi = d.bm;
            ring_of(g, d, (float)(first + 1u + kRing + kRingPlan) * d.delta, &bLo, &bRo);
            if (first + 1u > (unsigned int)(kRing + kRingPlan)) ring_of(g, d, (float)(first + 1u - kRing - kRingPlan) * d.delta, &bLi, &bRi);
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
        int bl = 0, bh = kTop;
        if (lo > 0.0f) {
            add(g.bucket_at_norm(lo - e), g.bucket_at_norm(lo + e));
            bl = g.bucket_at_norm(lo);
        }
        if (hi < 3.0e38f) {
            add(g.bucket_at_norm(hi - e), g.bucket_at_norm(hi + e));
            bh = g.bucket_at_norm(hi);
        }
        // bucket-resolution counts: half of the bound's own bucket on either side
        unsigned int c_lo = lo > 0.0f ? below(sh, bl) + (sh.prefix[bl] - below(sh, bl)) / 2u : 0u;
        unsigned int c_hi = hi < 3.0e38f ? sh.prefix[bh] - (sh.prefix[bh] - below(sh, bh)) / 2u : cnt;
        c_lo = c_lo > w.c_lo ? c_lo : w.c_lo;
        c_hi = c_hi < w.c_le_hi ? c_hi : w.c_le_hi;
        if (c_hi <= c_lo) break;
        w.c_lo = c_lo;
        w.c_le_hi = c_hi;
        w.n = c_hi - c_lo;
    }
    return nz;
}

// The zones sorted and merged (one zone per lane, sorted across the lanes; the merge walks them with readlane), their list
// segments laid out (wave 0).  False: no zone, or more hot pixels than the list holds.
__device__ __forceinline__ bool build_zones(Shared &sh, const Geo &g, int nz) {
    const int lane = threadIdx.x & 63;
    wave_sync();
    // key = lo << 16 | hi (both < 4096); lanes without a zone sort last
    uint32_t key = lane < nz ? ((uint32_t)sh.zone_lo[lane] << 16) | (uint32_t)sh.zone_hi[lane] : kNone;
    key = wave_sort64(key);
    int m = 0, cur_lo = 0, cur_hi = -2;
    int my_lo = 0, my_hi = -1;  // lane z keeps merged zone z
    for (int i = 0; i < nz; ++i) {
        const uint32_t k = (uint32_t)__builtin_amdgcn_readlane((int)key, i);
        const int lo = (int)(k >> 16), hi = (int)(k & 0xffffu);
        if (m > 0 && lo <= cur_hi + 1) {
            cur_hi = hi > cur_hi ? hi : cur_hi;
        } else {
            if (m > 0 && lane == m - 1) {
                my_lo = cur_lo;
                my_hi = cur_hi;
            }
            cur_lo = lo;
            cur_hi = hi;
            ++m;
        }
    }
    if (m > 0 && lane == m - 1) {
        my_lo = cur_lo;
        my_hi = cur_hi;
    }
    // segment sizes, table sizes: exclusive scans over the lanes
    const unsigned int size = lane < m ? sh.prefix[my_hi] - below(sh, my_lo) : 0u;
    const unsigned int tlen = lane < m ? (unsigned int)(my_hi - my_lo + 3) : 0u;
    const unsigned int at = wave_scan_incl(size) - size, tat = wave_scan_incl(tlen) - tlen;
    if (lane < m) {
        sh.zone_lo[lane] = my_lo;
        sh.zone_hi[lane] = my_hi;
        sh.zone_at[lane] = at;
        sh.zcur[lane] = at;
        sh.zone_tab[lane] = tat + tlen <= (unsigned int)kTabCap ? (int)tat : -1;
        if (lane == m - 1) sh.zone_at[m] = at + size;
    }
    if (lane == 0) sh.nzones = m;
    wave_sync();
    return m != 0 && sh.zone_at[m] <= (unsigned int)kHotCap;
}
// ... and by the whole workgroup: the lookup table marked (cleared beforehand), the boundary table filled
__device__ __forceinline__ void fill_zone_tables(Shared &sh, const Geo &g) {
    const int t = threadIdx.x, m = sh.nzones;
    for (int z = 0; z < m; ++z) {
        const int lo = sh.zone_lo[z], hi = sh.zone_hi[z], tb0 = sh.zone_tab[z];
        for (int b = lo + t; b <= hi; b += kThreads) sh.lut[b] = (unsigned char)(z + 1);
        if (tb0 >= 0)
            for (int b = lo - 1 + t; b <= hi + 1; b += kThreads) {
                uint32_t kf = 0u, kl = 0u;
                if (!g.first_norm(b, &kf)) kf = 0u;  // (0: "does not settle", reported by tfirst / tlast)
                if (!g.last_norm(b, &kl)) kl = 0u;
                sh.tab_first[tb0 + (b - lo + 1)] = kf;
                sh.tab_last[tb0 + (b - lo + 1)] = kl;
            }
    }
}

// ---- select: the ranks r_hi and r_lo (= r_hi or r_hi - 1; 0-based) of sh.tmp[0 .. n), n <= kSelCap, by one wave.  n <= 64:
// sorted across the lanes.  Else an 8-bit radix descent over the bytes in which the keys differ, the keys held in registers ----
__device__ __forceinline__ void wselect2(Shared &sh, unsigned int n, unsigned int r_lo, unsigned int r_hi, uint32_t *k_lo, uint32_t *k_hi) {
    constexpr int kMine = kSelCap / 64;
    const int lane = threadIdx.x & 63;
    unsigned int *const hist = &sh.prefix[kBuckets];  // 256 scratch words
    wave_sync();
    if (n <= 64u) {
        const uint32_t sorted = wave_sort64((unsigned int)lane < n ? sh.tmp[lane] : kNone);
        *k_lo = (uint32_t)__builtin_amdgcn_readlane((int)sorted, (int)r_lo);
        *k_hi = (uint32_t)__builtin_amdgcn_readlane((int)sorted, (int)r_hi);
        return;
    }
    uint32_t mine[kMine];
    const uint32_t k0 = sh.tmp[0];
    uint32_t diff = 0;
#pragma unroll
    for (int q = 0; q < kMine; ++q) {
        const unsigned int i = (unsigned int)(q * 64 + lane);
        mine[q] = i < n ? sh.tmp[i] : k0;  // pads repeat a real key and are skipped by index below
        diff |= mine[q] ^ k0;
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) diff |= (uint32_t)__shfl_xor((int)diff, off, 64);
    int top = 24;
    while (top > 0 && (diff >> top) == 0) top -= 8;  // the highest byte that varies
    uint32_t mask = top == 24 ? 0u : (0xffffffffu << (top + 8));
    uint32_t prefix = k0 & mask;
    unsigned int rank = r_hi;
    for (int shift = top; shift >= 0; shift -= 8) {
        *reinterpret_cast<uint4 *>(&hist[4 * lane]) = make_uint4(0u, 0u, 0u, 0u);
        wave_sync();
#pragma unroll
        for (int q = 0; q < kMine; ++q)
            if ((unsigned int)(q * 64) < n) {  // (uniform)
                if ((unsigned int)(q * 64 + lane) < n && (mine[q] & mask) == prefix) atomicAdd(&hist[(mine[q] >> shift) & 255u], 1u);
            }
        wave_sync();
        const uint4 c = *reinterpret_cast<const uint4 *>(&hist[4 * lane]);  // lane l owns digits 4l .. 4l+3
        const unsigned int own = c.x + c.y + c.z + c.w;
        const unsigned int incl = wave_scan_incl(own);
        const unsigned int excl = incl - own;
        const bool owner = rank >= excl && rank < incl;
        uint32_t digit = 0;
        unsigned int bel = 0;
        if (owner) {
            const unsigned int r = rank - excl;
            if (r < c.x) {
                digit = 4 * lane;
                bel = excl;
            } else if (r < c.x + c.y) {
                digit = 4 * lane + 1;
                bel = excl + c.x;
            } else if (r < c.x + c.y + c.z) {
                digit = 4 * lane + 2;
                bel = excl + c.x + c.y;
            } else {
                digit = 4 * lane + 3;
                bel = excl + c.x + c.y + c.z;
            }
        }
        const unsigned long long om = __builtin_amdgcn_ballot_w64(owner);
        const int ol = om ? first_lane(om) : 0;
        digit = (uint32_t)__builtin_amdgcn_readlane((int)digit, ol);
        bel = (unsigned int)__builtin_amdgcn_readlane((int)bel, ol);
        rank -= bel;
        prefix |= digit << shift;
        mask |= 255u << shift;
    }
    *k_hi = prefix;
    *k_lo = prefix;
    // `rank` is now r_hi's position among the keys equal to it.  The element before r_hi is the same key unless r_hi is its first
    // occurrence; then it is the largest key below.
    if (r_lo != r_hi && rank == 0) {
        uint32_t best = 0;
#pragma unroll
        for (int q = 0; q < kMine; ++q)
            if ((unsigned int)(q * 64 + lane) < n && mine[q] < prefix) best = best > mine[q] ? best : mine[q];
        *k_lo = tb::wave_max(best);
    }
}

// The keys of list[from .. to) that lie in [k_from, k_to], mapped by key_of (kNone = drop), appended to sh.tmp from `at` on.  Four
// entries per lane and step (independent chains: a step is bound by the LDS round trip, not by its arithmetic).  Returns the
// new fill, or kNone if tmp would overflow.
template <class KeyOf>
__device__ __forceinline__ unsigned int gather_keys(Shared &sh, unsigned int from, unsigned int to, uint32_t k_from, uint32_t k_to, unsigned int at,
                                                    KeyOf key_of) {
    const int lane = threadIdx.x & 63;
    const uint32_t *const list = reinterpret_cast<const uint32_t *>(sh.list);
#pragma unroll 1
    for (unsigned int i0 = from; i0 < to; i0 += 256) {
        uint32_t key[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const unsigned int i = i0 + 64u * j + lane;
            const uint32_t k = i < to ? list[i] : 0u;
            key[j] = (k >= k_from && k <= k_to) ? key_of(k) : kNone;
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const unsigned long long m = __builtin_amdgcn_ballot_w64(key[j] != kNone);
            const unsigned int pos = at + lanes_below(m);
            if (key[j] != kNone && pos < (unsigned int)kSelCap) sh.tmp[pos] = key[j];
            at += (unsigned int)__builtin_popcountll(m);
        }
        if (at > (unsigned int)kSelCap) return kNone;
    }
    return at;
}

// ---- the three requests of a round, answered from prefix sum + hot list (normalised keys by now); false = declined ------------
struct Rounds {
    Shared &sh;
    const Geo &g;
    ZReg zr;
    __device__ __forceinline__ bool fail(int why) {
        sh.decline = why;
        return false;
    }
    // The normalised keys that delimit the buckets [ba, bb] inside zone z.  Membership by key is exact unless the neighbouring
    // bucket ends on the very key this range starts with (the normalisation is coarser than the raw values there) and holds
    // pixels: then the request is declined.
    __device__ __forceinline__ bool key_range(int ba, int bb, uint32_t *k_from, uint32_t *k_to) {
        uint32_t prev, next;
        if (!tfirst(sh, zr, g, ba, k_from) || !tlast(sh, zr, g, bb, k_to) || !tlast(sh, zr, g, ba - 1, &prev) || !tfirst(sh, zr, g, bb + 1, &next)) return false;
        if (prev >= *k_from && sh.prefix[ba - 1] != below(sh, ba - 1)) return false;
        if (next <= *k_to && sh.prefix[bb + 1] != sh.prefix[bb]) return false;
        return true;
    }
    // order statistics of the VALUES: global ranks g_lo <= g_hi (adjacent or equal), as normalised keys
    __device__ __forceinline__ bool select_values(unsigned int g_lo, unsigned int g_hi, uint32_t *k_lo, uint32_t *k_hi) {
        const int b_hi = wfind_rank(sh, g_hi), b_lo = g_lo == g_hi ? b_hi : wfind_rank(sh, g_lo);
        if (b_lo <= 0 || b_hi >= kTop) return fail(D_MED_CATCHALL);
        const int z_lo = zr.zone_of(b_lo), z_hi = zr.zone_of(b_hi);
        if (z_lo < 0 || z_hi < 0) return fail(D_MED_COLD);
        auto same = [](uint32_t k) { return k; };
        uint32_t kf, kt;
        if (!key_range(b_lo, b_lo, &kf, &kt)) return fail(D_MED_COLD);
        unsigned int n = gather_keys(sh, zr.at_of(z_lo), zr.end_of(z_lo), kf, kt, 0u, same);
        if (n != kNone && b_hi != b_lo) {
            if (!key_range(b_hi, b_hi, &kf, &kt)) return fail(D_MED_COLD);
            n = gather_keys(sh, zr.at_of(z_hi), zr.end_of(z_hi), kf, kt, n, same);
        }
        if (n == kNone) return fail(D_MED_BIG);
        const unsigned int ex = below(sh, b_lo);  // (everything between the two buckets is empty: the ranks are adjacent)
        if (n != sh.prefix[b_hi] - ex) return fail(D_MED_COLD);
        TS_MARK(sh, 8);
        wselect2(sh, n, g_lo - ex, g_hi - ex, k_lo, k_hi);
        TS_MARK(sh, 9);
        return true;
    }
    // order statistics of the DEVIATIONS |v - median| of the window's candidates: ranks r_lo <= r_hi (adjacent or equal).
    // tile_bucket.hpp's guess-and-verify: the first threshold t* with F~(t*) >= r_hi + 1 places two rings of buckets; their
    // pixels come from the hot list; afterwards the guess is PROVED (the largest deviation possible between the rings <= the
    // smaller selected one, the smallest possible outside >= the larger).
    __device__ __forceinline__ bool select_devs(const Window &w, double median, unsigned int r_lo, unsigned int r_hi, unsigned int guess, uint32_t *d_lo,
                                                uint32_t *d_hi) {
        DevGeo d;
        if (!make_devgeo(sh, zr, g, median, &d)) return fail(D_DEV_BM);
        TS_MARK(sh, 10);
        const unsigned int first = wmad_first(sh, g, w, d, r_hi + 1u, guess);
        TS_MARK(sh, 11);
        if (first == kNone) return fail(D_DEV_NOFIRST);
        int bLo, bRo, bLi = 0, bRi = 0;
        ring_of(g, d, (float)(first + 1u + kRing) * d.delta, &bLo, &bRo);
        const bool has_inner_t = first + 1u > (unsigned int)kRing;
        if (has_inner_t) ring_of(g, d, (float)(first + 1u - kRing) * d.delta, &bLi, &bRi);
        const bool has_inner = has_inner_t && bLi + 1 <= bRi - 1;  // inner buckets: strictly between the two rings
        const unsigned int c_in = has_inner ? wprefix(sh, w, bRi) - wprefix(sh, w, bLi + 1) : 0u;
        const unsigned int n_cand = wprefix(sh, w, bRo + 1) - wprefix(sh, w, bLo) - c_in;
        if (!(bLo > 0 && bRo < kTop && c_in <= r_lo && c_in + n_cand >= r_hi + 1u && n_cand <= (unsigned int)kSelCap)) return fail(D_DEV_SHAPE);
        // the rings' pixels inside the window, as deviation keys
        auto key_of = [&](uint32_t k) { return (k >= w.lo && k <= w.hi) ? dev_key(k, median) : kNone; };
        unsigned int n;
        uint32_t kf, kt;
        if (has_inner) {
            const int zl = zr.zone_of(bLo), zq = zr.zone_of(bRi);
            if (zl < 0 || zq < 0 || bLi > zr.hi_of(zl) || bRo > zr.hi_of(zq)) return fail(D_DEV_COLD);
            if (!key_range(bLo, bLi, &kf, &kt)) return fail(D_DEV_COLD);
            n = gather_keys(sh, zr.at_of(zl), zr.end_of(zl), kf, kt, 0u, key_of);
            if (n != kNone) {
                if (!key_range(bRi, bRo, &kf, &kt)) return fail(D_DEV_COLD);
                n = gather_keys(sh, zr.at_of(zq), zr.end_of(zq), kf, kt, n, key_of);
            }
        } else {
            const int z = zr.zone_of(bLo);
            if (z < 0 || bRo > zr.hi_of(z)) return fail(D_DEV_COLD);
            if (!key_range(bLo, bRo, &kf, &kt)) return fail(D_DEV_COLD);
            n = gather_keys(sh, zr.at_of(z), zr.end_of(z), kf, kt, 0u, key_of);
        }
        if (n == kNone) return fail(D_DEV_BIG);
        if (n != n_cand) return fail(D_DEV_COUNT);
        TS_MARK(sh, 12);
        uint32_t sel_lo, sel_hi;
        wselect2(sh, n, r_lo - c_in, r_hi - c_in, &sel_lo, &sel_hi);
        TS_MARK(sh, 13);
        // the proof: inner deviations <= sel_lo, outer deviations >= sel_hi.  A bucket's normalised values lie between those of
        // its smallest and largest raw value; deviations fall towards the median and rise away from it, so the extremes sit at
        // the ends of the runs.
        uint32_t t_in = 0, t_out = kNone;
        if (has_inner) {
            uint32_t ka, kb;
            if (!tfirst(sh, zr, g, bLi + 1, &ka) || !tlast(sh, zr, g, bRi - 1, &kb)) return fail(D_DEV_PROOF);
            ka = ka > w.lo ? ka : w.lo;
            kb = kb < w.hi ? kb : w.hi;
            if (ka <= kb) {
                const uint32_t da = dev_key(ka, median), db = dev_key(kb, median);
                t_in = da > db ? da : db;
            }
        }
        if (wprefix(sh, w, bLo) > 0u) {  // window candidates left of the rings: all <= the largest value of bucket bLo - 1
            uint32_t k;
            if (!tlast(sh, zr, g, bLo - 1, &k) || !((double)__uint_as_float(k) <= median)) return fail(D_DEV_PROOF);
            t_out = dev_key(k, median);
        }
        if (wprefix(sh, w, kBuckets) - wprefix(sh, w, bRo + 1) > 0u) {  // ... right of them: all >= the smallest value of bucket bRo + 1
            uint32_t k;
            if (!tfirst(sh, zr, g, bRo + 1, &k) || !((double)__uint_as_float(k) >= median)) return fail(D_DEV_PROOF);
            const uint32_t dd = dev_key(k, median);
            t_out = t_out < dd ? t_out : dd;
        }
        TS_MARK(sh, 14);
        if (!(t_in <= sel_lo && sel_hi <= t_out)) return fail(D_DEV_PROOF);
        *d_lo = sel_lo;
        *d_hi = sel_hi;
        return true;
    }
    // candidates with normalised key < klo (HI = false) or <= khi (HI = true), exactly: whole buckets from the prefix sum, one
    // bucket counted key by key.  The bucket: the first one whose largest value reaches the bound.
    template <bool HI>
    __device__ __forceinline__ bool count_edge(uint32_t bound, unsigned int *out) {
        const int lane = threadIdx.x & 63;
        // first bucket b whose last normalised value is >= bound (HI: > bound): buckets before it lie entirely below, buckets
        // after it entirely at or above (HI: above)
        int b = g.bucket_at_norm(__uint_as_float(bound)), guard = 0;
        auto reaches = [&](int bb, bool *r) {
            uint32_t k;
            if (!tlast(sh, zr, g, bb, &k)) return false;
            *r = HI ? k > bound : k >= bound;
            return true;
        };
        bool r;
        while (b > 0) {  // step down while the bucket before also reaches the bound
            if (!reaches(b - 1, &r)) return fail(D_EDGE_COLD);
            if (!r || ++guard > 8) break;
            --b;
        }
        while (b < kTop) {
            if (!reaches(b, &r)) return fail(D_EDGE_COLD);
            if (r || ++guard > 16) break;
            ++b;
        }
        if (guard > 16) return fail(D_EDGE_COLD);
        const unsigned int bel = below(sh, b), in_b = sh.prefix[b] - bel;
        if (in_b == 0) {
            *out = bel;
            return true;
        }
        if (b <= 0 || b >= kTop) {
            // a catch-all bucket: settled only if it lies entirely on one side of the bound
            uint32_t kf, kl;
            if (!tfirst(sh, zr, g, b, &kf) || !tlast(sh, zr, g, b, &kl)) return fail(D_EDGE_CATCHALL);
            if (HI ? kl <= bound : kl < bound) {
                *out = bel + in_b;
                return true;
            }
            if (HI ? kf > bound : kf >= bound) {
                *out = bel;
                return true;
            }
            return fail(D_EDGE_CATCHALL);
        }
        const int z = zr.zone_of(b);
        uint32_t kf, kt;
        if (z < 0 || !key_range(b, b, &kf, &kt)) return fail(D_EDGE_COLD);
        const uint32_t *const list = reinterpret_cast<const uint32_t *>(sh.list);
        unsigned int c = 0, seen = 0;
        const unsigned int to = zr.end_of(z);
        for (unsigned int i0 = zr.at_of(z); i0 < to; i0 += 256) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const unsigned int i = i0 + 64u * j + lane;
                const uint32_t k = i < to ? list[i] : 0u;
                const bool mine = k >= kf && k <= kt;
                seen += mine ? 1u : 0u;
                c += (mine && (HI ? k <= bound : k < bound)) ? 1u : 0u;
            }
        }
        seen = tb::wave_sum(seen);
        c = tb::wave_sum(c);
        if (seen != in_b) return fail(D_EDGE_COLD);
        *out = bel + c;
        return true;
    }

    // sigma_clipped_stats(values, 3.0, 2) (sigma_clip.rs:4-34)
    __device__ __forceinline__ bool run(unsigned int cnt, const unsigned int (&first_guess)[3], double *median_out, double *sigma_out) {
        constexpr double kMadToSigma = 1.4826;
        zr.load(sh);
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
            if (!select_values(w.c_lo + (w.n % 2 == 0 ? mid - 1 : mid), w.c_lo + mid, &ka, &kb)) return false;
            median = w.n % 2 == 0 ? ((double)__uint_as_float(ka) + (double)__uint_as_float(kb)) / 2.0 : (double)__uint_as_float(kb);
            // median_f32_mut of the deviations (median.rs:46-63): f32 average for even n
            uint32_t da, db;
            if (!select_devs(w, median, w.n % 2 == 0 ? mid - 1 : mid, mid, first_guess[it], &da, &db)) return false;
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
            // as keys: candidates are positive floats, so v >= lo <=> key >= bits(lo) for lo > 0 (anything for lo <= 0) and
            // v <= hi <=> key <= bits(hi) for hi > 0 (nothing for hi <= 0: candidates exceed 1e-7)
            const uint32_t klo = lo > 0.0f ? __float_as_uint(lo) : 1u;
            const uint32_t khi = hi > 0.0f ? (__float_as_uint(hi) > 0x7f7fffffu ? 0x7f7fffffu : __float_as_uint(hi)) : 0u;
            const bool new_lo = klo > w.lo, new_hi = khi < w.hi;
            w.lo = new_lo ? klo : w.lo;
            w.hi = new_hi ? khi : w.hi;
            if (w.lo > w.hi) {
                w.n = 0;
                w.c_lo = w.c_le_hi = 0;
                continue;
            }
            TS_MARK(sh, 6);
            if (new_lo && w.lo > 0x33d6bf96u && !count_edge<false>(w.lo, &w.c_lo)) return false;
            if (new_hi && !count_edge<true>(w.hi, &w.c_le_hi)) return false;
            w.n = w.c_le_hi - w.c_lo;
            TS_MARK(sh, 15);
        }
        *median_out = median;
        *sigma_out = sigma;
        return true;
    }
};

// ---- streaming a tile -------------------------------------------------------------------------------------------------------
struct TileRect {
    const float *img;
    int64_t ld;
    int y0, y1, x0, x1;
    bool vec;  // the tile's width is a multiple of 4: rows as four-float loads (dword-aligned is enough: global_load_dwordx4 takes any
               // dword address; round 5 -- rows of an odd-width plane start at every alignment, and 16-byte alignment used to be required)
};
struct __attribute__((packed, aligned(4))) F4u {  // four floats at a dword-aligned address
    float x, y, z, w;
};
__device__ __forceinline__ float4 load4u(const float *p) {
    const F4u t = *reinterpret_cast<const F4u *>(p);
    return make_float4(t.x, t.y, t.z, t.w);
}
__device__ __forceinline__ bool whole_tile(const TileRect &r) { return r.vec && r.y1 - r.y0 == 256 && r.x1 - r.x0 == 256; }
// A WHOLE tile in BATCHES of 16 pixels: f(v[16], position of v[0], ...) -- the same loads in the same order as stream_tile's first
// branch; pixel i of a batch sits at position base + (i / 4) * (kWaves * 256) + (i % 4) of the tile (256 * row + column).  Pass 2 wants
// the batch: per pixel it is three DEPENDENT LDS round trips (zone table -> atomic with return -> list store), and written pixel by
// pixel the compiler must keep them in order across pixels too (the byte-typed table may alias anything): 512 round trips per thread,
// 58 000 of a tile's cycles.  Sixteen reads, then sixteen atomics, then sixteen stores are three round trips per batch.
template <class F>
__device__ __forceinline__ void stream_whole_tile_batches(const TileRect &r, F f) {
    constexpr int U = 4, kRounds = 64 / (2 * U);
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
    const float *p = r.img + (int64_t)(r.y0 + wv) * r.ld + r.x0 + 4 * lane;
    const int64_t step = (int64_t)kWaves * r.ld;
    float4 A[U], B[U];
    auto load = [&](float4(&X)[U]) {
#pragma unroll
        for (int u = 0; u < U; ++u) X[u] = load4u(p + u * step);
        p += U * step;
    };
    unsigned int at = (unsigned int)(wv * 256 + 4 * lane);  // position of the batch's first pixel
    auto eat = [&](const float4(&X)[U]) {
        float v[4 * U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            v[4 * u + 0] = X[u].x;
            v[4 * u + 1] = X[u].y;
            v[4 * u + 2] = X[u].z;
            v[4 * u + 3] = X[u].w;
        }
        f(v, at);
        at += (unsigned int)(U * kWaves * 256);
    };
    load(A);
#pragma unroll 1
    for (int it = 0; it < kRounds; ++it) {
        load(B);
        eat(A);
        if (it + 1 < kRounds) load(A);
        eat(B);
    }
}
// f(raw value) for every pixel of the tile (NaN where a lane has no pixel)
template <class F>
__device__ __forceinline__ void stream_tile(const TileRect &r, F f) {
    const int t = threadIdx.x;
    if (whole_tile(r)) {
        // A whole tile: wave w reads rows w, w + 4, ... -- 64 rows of 256 px, one float4 per lane.  Four rows are processed while
        // the next four are in flight (two register sets, taking turns): a wave that waits for every batch it has just asked for
        // exposes the memory latency eight times per pass.
        constexpr int U = 4, kRounds = 64 / (2 * U);
        const int lane = t & 63, wv = t >> 6;
        const float *p = r.img + (int64_t)(r.y0 + wv) * r.ld + r.x0 + 4 * lane;
        const int64_t step = (int64_t)kWaves * r.ld;  // between a wave's rows
        float4 A[U], B[U];
        auto load = [&](float4(&X)[U]) {
#pragma unroll
            for (int u = 0; u < U; ++u) X[u] = load4u(p + u * step);
            p += U * step;
        };
        auto eat = [&](const float4(&X)[U]) {
#pragma unroll
            for (int u = 0; u < U; ++u) {
                f(X[u].x);
                f(X[u].y);
                f(X[u].z);
                f(X[u].w);
            }
        };
        load(A);
#pragma unroll 1
        for (int it = 0; it < kRounds; ++it) {
            load(B);
            eat(A);
            if (it + 1 < kRounds) load(A);
            eat(B);
        }
    } else if (r.vec) {
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
                raw[u] = (col_ok && row < r.y1) ? load4u(p + (int64_t)row * r.ld)
                                                : make_float4(__builtin_nanf(""), __builtin_nanf(""), __builtin_nanf(""), __builtin_nanf(""));
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                f(raw[u].x);
                f(raw[u].y);
                f(raw[u].z);
                f(raw[u].w);
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
            for (int u = 0; u < U; ++u) f(raw[u]);
        }
    }
}

struct TileResult {
    double median, sigma;
    int valid;     // 1 if the tile had >= 8 valid pixels
    int declined;  // != 0: the tile needs tile_bucket.hpp (median / sigma / valid are not set)
};

// which raw values are candidates: xf(raw) finite and > 1e-7 (star_detection.rs:56).  xf is non-decreasing, so they are an
// interval [r_min, r_max]; r_min by bisection over the floats in order.
__device__ __forceinline__ void candidate_range(const ab_pixel_xf &xf, float *r_min, float *r_max) {
    if (!xf.on) {
        *r_min = __uint_as_float(0x33d6bf96u);  // the float after 1e-7f
        *r_max = 3.4028234663852886e38f;
        return;
    }
    *r_max = __builtin_inff();  // clamps to 1.0
    uint32_t lo = ord_of(-__builtin_inff()), hi = ord_of(__builtin_inff());  // invariant: pred(hi) (1.0 > 1e-7), !pred(lo) (0.0)
    while (hi - lo > 1u) {
        const uint32_t mid = lo + ((hi - lo) >> 1);
        if (ab_px(xf, from_ord(mid)) > 1e-7f)
            hi = mid;
        else
            lo = mid;
    }
    *r_min = from_ord(hi);
}

// where to zoom (efficiency only: any geometry is exact): the quartiles of 4 x 64 sample pixels, as tile_bucket.hpp.  False:
// the sample has no spread to speak of (flat / empty tiles: declined before a pixel is streamed).
__device__ __forceinline__ bool zoom_from_sample(Shared &sh, const TileRect &r, const ab_pixel_xf &xf, Geo *g, float *mad_guess) {
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
    const int h = r.y1 - r.y0, wd = r.x1 - r.x0;
    g->xf = xf;
    g->range = xf.on ? 1.0 / xf.inv : 1.0;
    candidate_range(xf, &g->r_min, &g->r_max);
    // sample t: row t * h / 256, a column that walks across the tile
    const int sr = r.y0 + (int)(((unsigned int)t * (unsigned int)h) >> 8), sc = r.x0 + (int)(((unsigned int)(t * 67 + 13) % 256u * (unsigned int)wd) >> 8);
    const float sv = r.img[(int64_t)sr * r.ld + sc];
    const uint32_t sorted = wave_sort64(g->is_cand(sv) ? ord_of(sv) : 0u);
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
        s1 += ok ? from_ord(sh.part[4 * i + 0]) : 0.0f;
        s2 += ok ? from_ord(sh.part[4 * i + 1]) : 0.0f;
        s3 += ok ? from_ord(sh.part[4 * i + 2]) : 0.0f;
        sn += ok ? 1.0f : 0.0f;
    }
    __syncthreads();
    if (!(sn > 0.0f)) return false;
    const float Q1 = s1 / sn, Q2 = s2 / sn, Q3 = s3 / sn;
    const float sig = (Q3 - Q1) * (1.0f / 1.349f);
    const float wb = sig / kPerSigma;  // bucket width (raw units)
    g->scale = 1.0f / wb;
    const float vlo = Q2 - (float)((kBuckets - 2) / 2) * wb;
    g->off = 1.0f - vlo * g->scale;
    *mad_guess = 0.6745f * sig * (float)(xf.on ? xf.inv : 1.0);  // normalised units
    return sig > 0.0f && __builtin_isfinite(g->scale) && __builtin_isfinite(g->off) && g->scale > 0.0f;
}

// ---- the tile's CANDIDATE LIST (round 6) ------------------------------------------------------------------------
// The registration's labelling pass used to stream every frame from HBM a second time to find the ~1 % of its pixels above the
// detection threshold T = median of tile medians + 3.5 x median of tile sigmas (star_detection.rs:70-84, :103) -- a number nobody
// knows before the last tile of the frame is done.  Pass 2 has every pixel in hand, so a WHOLE tile now also appends
// {position in the tile, raw value} of every pixel in bucket >= cb to a list in global memory, cb = the sample's median + kCandSigmas
// sample sigmas (buckets are uniform in the raw value, so "bucket >= cb" is "raw >= cut", cut = the smallest float of bucket cb,
// exactly).  The labelling kernel (detect.hip: label_bgtile_body) then checks xf(cut) <= T per tile -- xf is non-decreasing, so
// every pixel NOT on the list has xf(raw) <= xf(cut) <= T: the list is a superset of the thresholded set and what it labels is
// bit-identical -- and labels the tile from the list alone (1 - 2 % of the pixels, 8 bytes each).  A tile whose list is missing
// (cut = NaN: not a whole tile, declined before pass 2, no usable geometry), overflowed (count > cap) or cut too high (a tile
// brighter than the frame's threshold) is read from the frame, as before.
struct CandSink {
    uint2 *ent = nullptr;         // this tile's segment of `cap` entries {position = 256 * row + column, raw bits}
    unsigned int *cnt = nullptr;  // this tile's count (may exceed cap: the list is then not usable)
    float *cut = nullptr;         // this tile's cut (NaN: no list)
    unsigned int cap = 0;
};
#ifndef TS_CAND_SIGMAS
#define TS_CAND_SIGMAS 2.5f
#endif
constexpr float kCandSigmas = TS_CAND_SIGMAS;  // 3.5 is the matcher's threshold; 2.5 leaves a tile one sigma of offset against the frame's median

// Which of the workgroup's four waves plans the zones and runs the rounds (alone: the other three leave): tile b's wave b mod 4, not
// always wave 0 -- a workgroup's wave i tends to land on SIMD i, and the four tiles of a CU ran their single-wave rounds (half of a
// tile's lifetime) on ONE SIMD.  Round 6, interleaved A/B of the bench step: 9.54 / 9.50 / 9.52 / 9.47 ms rotated against 9.68 / 9.54 /
// 9.62 / 9.52 with wave 0 (-DTS_ROUNDS_WAVE0 keeps wave 0).
__device__ __forceinline__ int rounds_wave() {
#ifdef TS_ROUNDS_WAVE0
    return 0;
#else
    return (int)(blockIdx.x & 3u);
#endif
}

// the whole tile; the threads of wave rounds_wave() return the result (the other waves: `declined` only)
__device__ __forceinline__ TileResult tile_stats(Shared &sh, const TileRect &r, const ab_pixel_xf &xf, const CandSink cs = CandSink()) {
    const int t = threadIdx.x;
    TileResult res = {0.0, 1.0, 0, 0};
    if (cs.cut && t == 0) {
        *cs.cut = __builtin_nanf("");  // (until pass 2 has run: every early return leaves "no list")
        *cs.cnt = 0u;
    }
#ifdef AB_TILE_TIMING
    if (t == 0) {
        for (int i = 0; i < 16; ++i) sh.t_phase[i] = 0;
        sh.t_mark = clock64();
    }
#endif
    if (t == 0) {
        sh.decline = D_NONE;
        sh.nzones = 0;
        sh.ncand = 0u;
    }
#pragma unroll
    for (int i = 0; i < kPer; i += 4) *reinterpret_cast<uint4 *>(&sh.prefix[t * kPer + i]) = make_uint4(0u, 0u, 0u, 0u);
    *reinterpret_cast<uint4 *>(&sh.lut[16 * t]) = make_uint4(0u, 0u, 0u, 0u);
    Geo g;
    float mad_guess;
    if (!zoom_from_sample(sh, r, xf, &g, &mad_guess)) {  // (its barriers also publish the cleared histogram and table)
        res.declined = D_SAMPLE;
        return res;
    }
    TS_MARK(sh, 0);
    // ---- pass 1: the histogram ----
    // (no branches: what is not a candidate is counted into the thread's scratch word)
    stream_tile(r, [&](float v) { atomicAdd(&sh.prefix[g.is_cand(v) ? g.bucket(v) : kBuckets + t], 1u); });
    __syncthreads();
    TS_MARK(sh, 1);
    scan_prefix(sh);
    const unsigned int cnt = sh.prefix[kTop];
    TS_MARK(sh, 2);
    if (cnt < 8) return res;  // star_detection.rs:61
    res.valid = 1;
    // ---- plan, zones (wave 0), their tables (everybody) ----
    unsigned int first_guess[3] = {kNone, kNone, kNone};
    const int rw = rounds_wave();
    if ((t >> 6) == rw) {
        const int nz = plan_zones(sh, g, cnt, mad_guess, first_guess);
        TS_MARK(sh, 3);
        if (!build_zones(sh, g, nz)) sh.decline = sh.nzones == 0 ? D_ZONES : D_HOT_OVERFLOW;
    }
    __syncthreads();
    if (sh.decline) {
        res.declined = sh.decline;
        return res;
    }
    fill_zone_tables(sh, g);
    __syncthreads();
    TS_MARK(sh, 4);
    // ---- pass 2: the hot pixels, by zone (no branches: what is not hot lands in the thread's scratch words) ----
    // the candidate list's cut: bucket cb (the sample's median sits in bucket 2048, a sample sigma is kPerSigma buckets) and the
    // smallest float in it; without an exact cut there is no list
    const int cb = (kBuckets - 2) / 2 + 1 + (int)(kCandSigmas * kPerSigma);
    float cut_raw = 0.0f;
    const bool emit = cs.ent != nullptr && cb > 1 && cb < kTop && whole_tile(r) && g.first_raw(cb, &cut_raw);  // (block-uniform)
    if (whole_tile(r)) {
        // sixteen pixels at a time, phase by phase (see stream_whole_tile_batches); `emit` is block-uniform
        stream_whole_tile_batches(r, [&](const float(&v)[16], unsigned int pos0) {
            int b[16];
            unsigned int z[16], at[16];
#pragma unroll
            for (int i = 0; i < 16; ++i) b[i] = g.bucket(v[i]);
#pragma unroll
            for (int i = 0; i < 16; ++i) z[i] = sh.lut[b[i]];
#pragma unroll
            for (int i = 0; i < 16; ++i) z[i] = g.is_cand(v[i]) ? z[i] : 0u;
#pragma unroll
            for (int i = 0; i < 16; ++i) at[i] = atomicAdd(&sh.zcur[z[i] ? z[i] - 1u : (unsigned int)(kMaxZones + t)], 1u);
#pragma unroll
            for (int i = 0; i < 16; ++i) sh.list[(z[i] && at[i] < (unsigned int)kHotCap) ? at[i] : (unsigned int)(kHotCap + t)] = v[i];
            if (emit) {
                unsigned int hit = 0;
#pragma unroll
                for (int i = 0; i < 16; ++i) hit |= (b[i] >= cb && v[i] <= g.r_max) ? 1u << i : 0u;  // (NaN fails the second test)
                while (hit) {  // rare: 1 - 2 % of the pixels
                    const int i = __builtin_ctz(hit);
                    hit &= hit - 1u;
                    const unsigned int ca = atomicAdd(&sh.ncand, 1u);
                    float vi = v[0];
#pragma unroll
                    for (int k = 1; k < 16; ++k) vi = i == k ? v[k] : vi;  // (no runtime index into the register array)
                    if (ca < cs.cap) cs.ent[ca] = make_uint2(pos0 + (unsigned int)((i >> 2) * (kWaves * 256) + (i & 3)), __float_as_uint(vi));
                }
            }
        });
    } else {
        stream_tile(r, [&](float v) {
            unsigned int z = sh.lut[g.bucket(v)];
            z = g.is_cand(v) ? z : 0u;
            const unsigned int at = atomicAdd(&sh.zcur[z ? z - 1u : (unsigned int)(kMaxZones + t)], 1u);
            sh.list[(z && at < (unsigned int)kHotCap) ? at : (unsigned int)(kHotCap + t)] = v;
        });
    }
    __syncthreads();
    if (emit && t == 0) {
        *cs.cnt = sh.ncand;
        *cs.cut = cut_raw;
    }
    // the list as normalised keys, in place (the only place the normalisation is evaluated for more than a handful of values)
    {
        const unsigned int n = sh.zone_at[sh.nzones];
        for (unsigned int i = t; i < n; i += kThreads) sh.list[i] = __uint_as_float(g.norm_key(sh.list[i]));
    }
    __syncthreads();
    TS_MARK(sh, 5);
#ifdef AB_TILE_TIMING
    if (t == 0) sh.t_phase[7] = sh.zone_at[sh.nzones] * 1000 + sh.nzones;  // (hot pixels, zones: printed by tools/tile_stream_bench)
#endif
    if ((t >> 6) != rw) return res;  // (declined / valid of these waves are not read)
    // ---- the rounds (wave 0) ----
    Rounds rounds = {sh, g, ZReg()};
    if (!rounds.run(cnt, first_guess, &res.median, &res.sigma)) res.declined = sh.decline ? sh.decline : 1;
    TS_MARK(sh, 6);
    return res;
}

}  // namespace ts
