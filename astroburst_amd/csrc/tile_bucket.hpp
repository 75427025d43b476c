// Sigma-clipped median / MAD of one background tile (<= 256 x 256 px) by ONE workgroup, exactly (gfx950).
//
// Replaces sigma_clipped_stats(values, 3.0, 2) of estimate_background's per-tile body
// (core/analysis/star_detection.rs:47-68, math/sigma_clip.rs:4-34, math/median.rs:27-63).
//
// Round 1 ran an 11/11/10-bit radix select per order statistic: 13 LDS-histogram sweeps over the tile (~18 VALU
// instructions and one LDS atomic per pixel and sweep; 180 us per 4096^2 frame, 36 % of a frame's registration time).
// Here the tile is histogrammed ONCE and everything else is answered from that histogram plus cheap register sweeps:
//
//   * the 64 pixels of a thread stay in VGPRs as monotone u32 keys (valid pixels are positive finite floats, so the
//     IEEE bit pattern is the key; 0 = not a candidate);
//   * one sweep with LDS atomics builds a 4096-bucket histogram of the keys over [min key, max key] (bucket =
//     (key - base) >> shift, shift chosen so that the span fits), turned into an inclusive prefix sum;
//   * the clipping windows of sigma_clipped_stats only ever cut tails, so "elements of the current window below bucket
//     b" is a closed form of that one prefix sum (clamped between the counts below the window's ends) -- the histogram
//     is never rebuilt;
//   * an order statistic of the VALUES: the prefix sum names its bucket, a register sweep (two compares per pixel, no
//     atomics except for the few matches) gathers that bucket's keys into an LDS list, one wave selects in the list;
//   * an order statistic of the DEVIATIONS |v - median| (the MAD): the deviations of a bucket's pixels lie between the
//     deviations of its two boundary keys, so every bucket boundary yields a lower and an upper bound on the count of
//     deviations below it; the tightest pair brackets the wanted deviation between two boundaries, only the pixels of
//     the buckets straddling that bracket -- a few hundred -- are gathered (as deviation keys) and selected from;
//   * a list that would overflow (heavy ties: a flat tile is one bucket of 65 536 equal keys) falls back to bisection
//     on the key with counting sweeps; exact as well, just slower.
//
// Everything is integer comparison on keys plus the reference's own f64 / f32 arithmetic for the deviation
// (`(v as f64 - median).abs() as f32`), so tile medians and sigmas are bit-identical to the oracle's.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

#include "ab_common.hpp"

namespace tb {

constexpr int kThreads = 1024;  // one workgroup per tile
constexpr int kSlots = 64;      // pixels per thread: 256 x 256 / 1024
constexpr int kBuckets = 4096;
constexpr int kListCap = 6144;  // gathered candidates (u32 keys)

struct Shared {
    unsigned int prefix[kBuckets];  // bucket counts, then their inclusive prefix sum
    unsigned int list[kListCap];
    unsigned int wave_part[16 * 4];  // per-wave partials of the block reductions
    unsigned int hist[256];          // wave 0's radix-select histogram
    unsigned int scal[16];           // broadcast slots
};

struct Keys {
    uint32_t k[kSlots];  // 0 = not a candidate
};

struct Frame {  // histogram geometry
    uint32_t base;   // key of bucket 0's first element
    int shift;       // bucket = (key - base) >> shift
    unsigned int total;  // valid pixels
};

// An opaque copy of a key (at most one v_mov): what a sweep computes from it cannot be hoisted out of the enclosing loops and
// kept alive across the other sweeps (64 SGPR-pair masks per hoisted predicate, spilled lane by lane), and -- unlike marking
// the key array itself as rewritten -- the 64 keys stay loop-invariant values (no 64-wide PHIs at every loop header).
__device__ __forceinline__ uint32_t fresh(uint32_t k) {
    uint32_t o;
    asm volatile("v_mov_b32 %0, %1" : "=v"(o) : "v"(k));
    return o;
}

__device__ __forceinline__ unsigned int wave_sum(unsigned int x) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) x += __shfl_xor(x, off, 64);
    return x;
}
__device__ __forceinline__ unsigned int wave_min(unsigned int x) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) x = min(x, (unsigned int)__shfl_xor(x, off, 64));
    return x;
}
__device__ __forceinline__ unsigned int wave_max(unsigned int x) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) x = max(x, (unsigned int)__shfl_xor(x, off, 64));
    return x;
}

// up to four values reduced over the workgroup; every thread gets the results (two barriers)
enum { OP_SUM = 0, OP_MIN = 1, OP_MAX = 2 };
template <int OP0, int OP1, int OP2, int OP3>
__device__ __forceinline__ void block_reduce4(Shared &sh, unsigned int &a, unsigned int &b, unsigned int &c, unsigned int &d) {
    auto wred = [](unsigned int x, int op) { return op == OP_SUM ? wave_sum(x) : (op == OP_MIN ? wave_min(x) : wave_max(x)); };
    auto comb = [](unsigned int x, unsigned int y, int op) { return op == OP_SUM ? x + y : (op == OP_MIN ? min(x, y) : max(x, y)); };
    a = wred(a, OP0);
    b = wred(b, OP1);
    c = wred(c, OP2);
    d = wred(d, OP3);
    const int w = threadIdx.x >> 6;
    __syncthreads();  // (the partial slots may still be read from the previous reduction)
    if ((threadIdx.x & 63) == 0) {
        sh.wave_part[4 * w + 0] = a;
        sh.wave_part[4 * w + 1] = b;
        sh.wave_part[4 * w + 2] = c;
        sh.wave_part[4 * w + 3] = d;
    }
    __syncthreads();
    unsigned int ra = sh.wave_part[0], rb = sh.wave_part[1], rc = sh.wave_part[2], rd = sh.wave_part[3];
#pragma unroll
    for (int i = 1; i < kThreads / 64; ++i) {
        ra = comb(ra, sh.wave_part[4 * i + 0], OP0);
        rb = comb(rb, sh.wave_part[4 * i + 1], OP1);
        rc = comb(rc, sh.wave_part[4 * i + 2], OP2);
        rd = comb(rd, sh.wave_part[4 * i + 3], OP3);
    }
    a = ra;
    b = rb;
    c = rc;
    d = rd;
}

// (v as f64 - median).abs() as f32, as a monotone key (sigma_clip.rs:15,31)
__device__ __forceinline__ uint32_t dev_key(uint32_t key, double median) {
    return __float_as_uint((float)fabs((double)__uint_as_float(key) - median));
}

// count of candidates with key < a and with key <= b (one register sweep, no atomics)
__device__ __forceinline__ void count_below(const Keys &K, Shared &sh, uint32_t a, uint32_t b, unsigned int *lt_a, unsigned int *le_b) {
    unsigned int ca = 0, cb = 0;
#pragma unroll
    for (int i = 0; i < kSlots; ++i) {
        const uint32_t k = fresh(K.k[i]);
        ca += (k != 0 && k < a) ? 1u : 0u;
        cb += (k != 0 && k <= b) ? 1u : 0u;
    }
    unsigned int z0 = 0, z1 = 0;
    block_reduce4<OP_SUM, OP_SUM, OP_SUM, OP_SUM>(sh, ca, cb, z0, z1);
    *lt_a = ca;
    *le_b = cb;
}

// count of window candidates whose deviation key is <= d
__device__ __forceinline__ unsigned int count_dev_le(const Keys &K, Shared &sh, uint32_t wlo, uint32_t whi, double median, uint32_t d) {
    unsigned int c = 0;
#pragma unroll
    for (int i = 0; i < kSlots; ++i) {
        const uint32_t k = fresh(K.k[i]);
        if (k >= wlo && k <= whi && k != 0) c += dev_key(k, median) <= d ? 1u : 0u;
    }
    unsigned int z0 = 0, z1 = 0, z2 = 0;
    block_reduce4<OP_SUM, OP_SUM, OP_SUM, OP_SUM>(sh, c, z0, z1, z2);
    return c;
}

// append this lane's value to the LDS list (wave-aggregated: one atomic per wave); entries past the capacity are dropped,
// the counter keeps counting
__device__ __forceinline__ void list_push(Shared &sh, bool take, uint32_t v) {
    const unsigned long long m = __ballot(take);
    if (m) {
        const int lane = threadIdx.x & 63, leader = (int)__builtin_ctzll(m);
        unsigned int base = 0;
        if (lane == leader) base = atomicAdd(&sh.scal[0], (unsigned int)__builtin_popcountll(m));
        base = (unsigned int)__builtin_amdgcn_readlane((int)base, leader);
        const unsigned int at = base + (unsigned int)__builtin_popcountll(m & ((1ull << lane) - 1ull));
        if (take && at < (unsigned int)kListCap) sh.list[at] = v;
    }
}

// wave 0: the rank-th smallest (0-based) of sh.list[0 .. n) by an 8-bit radix select (4 digits), n <= kListCap.
// Returns the key in every lane of wave 0; other waves must not call.
__device__ __forceinline__ uint32_t wave_select(Shared &sh, unsigned int n, unsigned int rank) {
    const int lane = threadIdx.x & 63;
    uint32_t prefix = 0, mask = 0;
    for (int shift = 24; shift >= 0; shift -= 8) {
#pragma unroll
        for (int i = 0; i < 4; ++i) sh.hist[lane + 64 * i] = 0;
        __builtin_amdgcn_wave_barrier();
        for (unsigned int i = lane; i < n; i += 64) {
            const uint32_t k = sh.list[i];
            if ((k & mask) == prefix) atomicAdd(&sh.hist[(k >> shift) & 255u], 1u);
        }
        __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0): the LDS atomics have landed (same wave)
        __builtin_amdgcn_wave_barrier();
        // lane l owns digits 4l .. 4l+3
        const unsigned int c0 = sh.hist[4 * lane], c1 = sh.hist[4 * lane + 1], c2 = sh.hist[4 * lane + 2], c3 = sh.hist[4 * lane + 3];
        const unsigned int mine = c0 + c1 + c2 + c3;
        unsigned int incl = mine;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const unsigned int up = __shfl_up(incl, off, 64);
            if (lane >= off) incl += up;
        }
        const unsigned int excl = incl - mine;
        const bool owner = rank >= excl && rank < incl;
        uint32_t digit = 0;
        unsigned int below = 0;
        if (owner) {
            unsigned int r = rank - excl;
            if (r < c0) {
                digit = 4 * lane;
                below = excl;
            } else if (r < c0 + c1) {
                digit = 4 * lane + 1;
                below = excl + c0;
            } else if (r < c0 + c1 + c2) {
                digit = 4 * lane + 2;
                below = excl + c0 + c1;
            } else {
                digit = 4 * lane + 3;
                below = excl + c0 + c1 + c2;
            }
        }
        const unsigned long long om = __ballot(owner);
        const int ol = om ? (int)__builtin_ctzll(om) : 0;
        digit = (uint32_t)__builtin_amdgcn_readlane((int)digit, ol);
        below = (unsigned int)__builtin_amdgcn_readlane((int)below, ol);
        rank -= below;
        prefix |= digit << shift;
        mask |= 255u << shift;
        __builtin_amdgcn_wave_barrier();
    }
    return prefix;
}

// ---- order statistics of the VALUE keys ---------------------------------------------------------------------------------
// global ranks g_lo <= g_hi (g_hi - g_lo <= 1) among all candidates, sorted by key.  Every thread returns both keys.
__device__ __forceinline__ void select_values(const Keys &K, Shared &sh, const Frame &f, unsigned int g_lo, unsigned int g_hi, uint32_t *k_lo,
                                              uint32_t *k_hi) {
    // the buckets that hold the two ranks: bucket b covers ranks [prefix[b-1], prefix[b])
    unsigned int b_lo = 0xffffffffu, b_hi = 0, ex_lo = 0xffffffffu, z = 0;
#pragma unroll
    for (int j = 0; j < kBuckets / kThreads; ++j) {
        const int b = threadIdx.x * (kBuckets / kThreads) + j;
        const unsigned int incl = sh.prefix[b], excl = b ? sh.prefix[b - 1] : 0u;
        if (g_lo >= excl && g_lo < incl) {
            b_lo = (unsigned int)b;
            ex_lo = excl;
        }
        if (g_hi >= excl && g_hi < incl) b_hi = (unsigned int)b;
    }
    block_reduce4<OP_MIN, OP_MAX, OP_MIN, OP_SUM>(sh, b_lo, b_hi, ex_lo, z);
    // key range of the two buckets (everything in between is empty: the ranks are adjacent)
    const uint64_t r_lo = (uint64_t)f.base + ((uint64_t)b_lo << f.shift);
    const uint64_t r_hi = (uint64_t)f.base + (((uint64_t)b_hi + 1) << f.shift) - 1;
    const uint32_t key_lo = (uint32_t)r_lo, key_hi = r_hi > 0xffffffffull ? 0xffffffffu : (uint32_t)r_hi;
    const unsigned int in_range = sh.prefix[b_hi] - ex_lo;
    if (in_range <= (unsigned int)kListCap) {
        if (threadIdx.x == 0) sh.scal[0] = 0;
        __syncthreads();
    #pragma unroll
        for (int i = 0; i < kSlots; ++i) {
            const uint32_t k = fresh(K.k[i]);
            list_push(sh, k >= key_lo && k <= key_hi && k != 0, k);
        }
        __syncthreads();
        if (threadIdx.x < 64) {
            const uint32_t a = wave_select(sh, in_range, g_lo - ex_lo);
            const uint32_t b = g_hi == g_lo ? a : wave_select(sh, in_range, g_hi - ex_lo);
            if (threadIdx.x == 0) {
                sh.scal[1] = a;
                sh.scal[2] = b;
            }
        }
        __syncthreads();
        *k_lo = sh.scal[1];
        *k_hi = sh.scal[2];
        __syncthreads();
        return;
    }
    // too many equal-ish keys for the list: bisect on the key with counting sweeps (smallest key x with count(<= x) > rank)
    uint32_t res[2];
    for (int which = 0; which < 2; ++which) {
        const unsigned int g = which ? g_hi : g_lo;
        if (which && g_hi == g_lo) {
            res[1] = res[0];
            break;
        }
        uint32_t lo = key_lo, hi = key_hi;  // invariant: count(<= lo - 1) <= g < count(<= hi)
        while (lo < hi) {
            const uint32_t mid = lo + ((hi - lo) >> 1);
            unsigned int lt, le;
            count_below(K, sh, 0, mid, &lt, &le);
            if (le > g)
                hi = mid;
            else
                lo = mid + 1;
        }
        res[which] = lo;
    }
    *k_lo = res[0];
    *k_hi = res[1];
}

// ---- order statistics of the DEVIATION keys ---------------------------------------------------------------------------------
struct Window {
    uint32_t lo, hi;            // candidate keys retained so far: lo <= key <= hi
    unsigned int c_lo, c_le_hi;  // candidates with key < lo / key <= hi
    unsigned int n;              // = c_le_hi - c_lo
};

// window candidates with key < first key of bucket b  (b in [0, kBuckets])
__device__ __forceinline__ unsigned int wprefix(const Shared &sh, const Window &w, int b) {
    const unsigned int p = b <= 0 ? 0u : sh.prefix[b - 1];
    const unsigned int q = p < w.c_lo ? w.c_lo : (p > w.c_le_hi ? w.c_le_hi : p);
    return q - w.c_lo;
}

struct DevCtx {
    Frame f;
    double median;
    int nb;  // buckets in use: ((max key - base) >> shift) + 1
    __device__ __forceinline__ uint32_t first_key(int b) const { return f.base + ((uint32_t)b << f.shift); }  // b < nb
    __device__ __forceinline__ uint32_t last_key(int b) const {
        const uint64_t k = (uint64_t)f.base + (((uint64_t)b + 1) << f.shift) - 1;
        return k > 0x7f7fffffull ? 0x7f7fffffu : (uint32_t)k;
    }
    __device__ __forceinline__ bool left_of_median(uint32_t key) const { return (double)__uint_as_float(key) <= median; }
    __device__ __forceinline__ int bucket_of(uint32_t key) const {
        if (key < f.base) return 0;
        const uint32_t b = (key - f.base) >> f.shift;
        return b >= (uint32_t)nb ? nb - 1 : (int)b;
    }
    // Smallest bucket b in [0, bm] whose boundary key (FIRST or LAST key of the bucket) has deviation <= t, scanning the
    // side left of the median; bm + 1 if none.  The deviation of a key left of the median falls as the key rises, so the
    // predicate is monotone in b; the start is the bucket of the float nearest median - t, corrected by stepping.
    template <bool LAST>
    __device__ __forceinline__ int left_edge(uint32_t t, int bm) const {
        const double x = median - (double)__uint_as_float(t);
        int b = x <= 0.0 ? 0 : bucket_of(__float_as_uint((float)x));
        if (b > bm) b = bm;
        auto ok = [&](int bb) {
            const uint32_t key = LAST ? last_key(bb) : first_key(bb);
            return !left_of_median(key) || dev_key(key, median) <= t;  // a boundary key right of the median: the bucket holds the median
        };
        while (b > 0 && ok(b - 1)) --b;
        while (b <= bm && !ok(b)) ++b;
        return b;
    }
    // Largest bucket b in [bm, nb) whose boundary key has deviation <= t on the right side; bm - 1 if none.
    template <bool LAST>
    __device__ __forceinline__ int right_edge(uint32_t t, int bm) const {
        const double x = median + (double)__uint_as_float(t);
        int b = bucket_of(x >= 3.4028234663852886e38 ? 0x7f7fffffu : __float_as_uint((float)x));
        if (b < bm) b = bm;
        auto ok = [&](int bb) {
            const uint32_t key = LAST ? last_key(bb) : first_key(bb);
            return left_of_median(key) || dev_key(key, median) <= t;
        };
        while (b < nb - 1 && ok(b + 1)) ++b;
        while (b >= bm && !ok(b)) --b;
        return b;
    }
};

// window candidates in buckets that lie ENTIRELY at deviation <= t (full) / that reach down to deviation <= t (any);
// also the two bucket ranges themselves ([fl, fr] full, [al, ar] any; empty ranges have l > r)
struct Bounds {
    unsigned int n_full, n_any;
    int fl, fr, al, ar;
};
__device__ __forceinline__ Bounds dev_bounds(const Shared &sh, const Window &w, const DevCtx &c, int bm, uint32_t t) {
    Bounds o;
    // left of the median a bucket's largest deviation sits at its FIRST key, its smallest at its LAST key; mirrored on the right
    o.fl = c.left_edge<false>(t, bm);
    o.fr = c.right_edge<true>(t, bm);
    o.al = c.left_edge<true>(t, bm);
    o.ar = c.right_edge<false>(t, bm);
    // the bucket that holds the median has deviations on both sides of it: full only if both of its boundary keys qualify
    const bool bm_full = o.fl <= bm && o.fr >= bm;
    if (!bm_full) {
        // full buckets are then [fl, bm - 1] on the left and [bm + 1, fr] on the right: two runs
        const unsigned int left = o.fl <= bm - 1 ? wprefix(sh, w, bm) - wprefix(sh, w, o.fl) : 0u;
        const unsigned int right = o.fr >= bm + 1 ? wprefix(sh, w, o.fr + 1) - wprefix(sh, w, bm + 1) : 0u;
        o.n_full = left + right;
    } else {
        o.n_full = wprefix(sh, w, o.fr + 1) - wprefix(sh, w, o.fl);
    }
    // "any": the median's bucket always reaches deviation ~0 <= t ... unless it has no candidates; counting it is still a valid
    // upper bound
    if (o.al > bm) o.al = bm;
    if (o.ar < bm) o.ar = bm;
    o.n_any = wprefix(sh, w, o.ar + 1) - wprefix(sh, w, o.al);
    return o;
}

// ranks r_lo <= r_hi (adjacent or equal, 0-based) of the deviation keys of the window's candidates
__device__ __forceinline__ void select_devs(const Keys &K, Shared &sh, const Frame &f, int nb, const Window &w, double median, unsigned int r_lo,
                                            unsigned int r_hi, uint32_t *d_lo, uint32_t *d_hi) {
    DevCtx c;
    c.f = f;
    c.median = median;
    c.nb = nb;
    // the bucket of the median: the last bucket whose first key is <= median
    int bm;
    {
        const float mf = (float)median;  // may round up past the median: step back below
        int b = mf <= 0.0f ? 0 : c.bucket_of(__float_as_uint(mf));
        while (b > 0 && !c.left_of_median(c.first_key(b))) --b;
        while (b < nb - 1 && c.left_of_median(c.first_key(b + 1))) ++b;
        bm = b;
    }
    // every non-empty bucket proposes its largest deviation as a threshold t:
    //   N_any(t) <= r_lo      =>  the wanted deviations are > t        (best such t: the largest)
    //   N_full(t) >= r_hi + 1 =>  the wanted deviations are <= t       (best such t: the smallest)
    unsigned int t_lo = 0, have_lo = 0, t_hi = 0xffffffffu, z = 0;
#pragma unroll 1
    for (int j = 0; j < kBuckets / kThreads; ++j) {
        const int b = threadIdx.x * (kBuckets / kThreads) + j;
        if (b >= nb) continue;
        if (wprefix(sh, w, b + 1) == wprefix(sh, w, b)) continue;  // no window candidates in this bucket
        const uint32_t dl = dev_key(c.first_key(b), median), dh = dev_key(c.last_key(b), median);
        const uint32_t t = dl > dh ? dl : dh;
        const Bounds o = dev_bounds(sh, w, c, bm, t);
        if (o.n_any <= r_lo) {
            t_lo = t_lo > t + 1 ? t_lo : t + 1;  // stored + 1 so that 0 means "none"
            have_lo = 1;
        }
        if (o.n_full >= r_hi + 1) t_hi = t_hi < t ? t_hi : t;
    }
    block_reduce4<OP_MAX, OP_MAX, OP_MIN, OP_SUM>(sh, t_lo, have_lo, t_hi, z);
    // candidates: window pixels outside the buckets entirely below t_lo and inside the buckets reaching below t_hi
    Bounds lo_b, hi_b;
    unsigned int c0 = 0;
    bool lo_bm_full = false;
    if (have_lo) {
        lo_b = dev_bounds(sh, w, c, bm, t_lo - 1);
        c0 = lo_b.n_full;
        lo_bm_full = lo_b.fl <= bm && lo_b.fr >= bm;
    }
    uint32_t a_lo_key = w.lo, a_hi_key = w.hi;  // outer range: everything (t_hi unknown: cannot happen for r_hi < n, kept for safety)
    unsigned int n_cand = w.n - c0;
    if (t_hi != 0xffffffffu) {
        hi_b = dev_bounds(sh, w, c, bm, t_hi);
        a_lo_key = c.first_key(hi_b.al);
        a_hi_key = c.last_key(hi_b.ar);
        n_cand = hi_b.n_any - c0;
    }
    // excluded runs (entirely below t_lo): [fl, min(fr, bm - 1)] and [max(fl, bm + 1), fr], plus bm itself when full
    uint32_t x1_lo = 1, x1_hi = 0, x2_lo = 1, x2_hi = 0;  // empty
    if (have_lo) {
        const int l_end = lo_bm_full ? lo_b.fr : (bm - 1 < lo_b.fr ? bm - 1 : lo_b.fr);
        if (lo_b.fl <= bm && lo_b.fl <= l_end) {
            x1_lo = c.first_key(lo_b.fl);
            x1_hi = c.last_key(l_end);
        }
        if (!lo_bm_full && lo_b.fr >= bm + 1) {
            x2_lo = c.first_key(bm + 1);
            x2_hi = c.last_key(lo_b.fr);
        }
    }
    const unsigned int want_lo = r_lo - c0, want_hi = r_hi - c0;
    if (n_cand <= (unsigned int)kListCap) {
        if (threadIdx.x == 0) sh.scal[0] = 0;
        __syncthreads();
    #pragma unroll
        for (int i = 0; i < kSlots; ++i) {
            const uint32_t k = fresh(K.k[i]);
            const bool in = k != 0 && k >= w.lo && k <= w.hi && k >= a_lo_key && k <= a_hi_key && !(k >= x1_lo && k <= x1_hi) &&
                            !(k >= x2_lo && k <= x2_hi);
            list_push(sh, in, in ? dev_key(k, median) : 0u);
        }
        __syncthreads();
        if (threadIdx.x < 64) {
            const unsigned int n_list = sh.scal[0];  // == n_cand
            const uint32_t a = wave_select(sh, n_list, want_lo);
            const uint32_t b = r_hi == r_lo ? a : wave_select(sh, n_list, want_hi);
            if (threadIdx.x == 0) {
                sh.scal[1] = a;
                sh.scal[2] = b;
            }
        }
        __syncthreads();
        *d_lo = sh.scal[1];
        *d_hi = sh.scal[2];
        __syncthreads();
        return;
    }
    // heavy ties: bisect on the deviation key inside the bracket (smallest d with count(dev <= d) > rank)
    uint32_t res[2];
    for (int which = 0; which < 2; ++which) {
        const unsigned int r = which ? r_hi : r_lo;
        if (which && r_hi == r_lo) {
            res[1] = res[0];
            break;
        }
        uint32_t lo = have_lo ? t_lo : 0u, hi = t_hi == 0xffffffffu ? 0x7f800000u : t_hi;  // (t_lo is stored + 1: the first key above it)
        while (lo < hi) {
            const uint32_t mid = lo + ((hi - lo) >> 1);
            if (count_dev_le(K, sh, w.lo, w.hi, median, mid) > r)
                hi = mid;
            else
                lo = mid + 1;
        }
        res[which] = lo;
    }
    *d_lo = res[0];
    *d_hi = res[1];
}

struct TileResult {
    double median, sigma;
    int valid;
};

// sigma_clipped_stats(values, 3.0, 2) of the candidates in K (sigma_clip.rs:4-34); needs >= 8 candidates (star_detection.rs:61)
__device__ __forceinline__ TileResult tile_stats(const Keys &K, Shared &sh) {
    constexpr double kMadToSigma = 1.4826;
    // ---- histogram geometry: min / max / count of the candidates ----
    unsigned int kmin = 0xffffffffu, kmax = 0, cnt = 0, z = 0;
#pragma unroll
    for (int i = 0; i < kSlots; ++i) {
        const uint32_t k = fresh(K.k[i]);
        if (k) {
            kmin = min(kmin, k);
            kmax = max(kmax, k);
            ++cnt;
        }
    }
    block_reduce4<OP_MIN, OP_MAX, OP_SUM, OP_SUM>(sh, kmin, kmax, cnt, z);
    TileResult res = {0.0, 1.0, 0};
    if (cnt < 8) return res;
    res.valid = 1;
    Frame f;
    f.base = kmin;
    f.total = cnt;
    const uint32_t span = kmax - kmin;
    f.shift = 0;
    while ((span >> f.shift) >= (uint32_t)kBuckets) ++f.shift;
    const int nb = (int)(span >> f.shift) + 1;
    // ---- the one histogram sweep ----
    for (int i = threadIdx.x; i < kBuckets; i += kThreads) sh.prefix[i] = 0;
    __syncthreads();
#pragma unroll
    for (int i = 0; i < kSlots; ++i) {
        const uint32_t k = fresh(K.k[i]);
        if (k) atomicAdd(&sh.prefix[(k - f.base) >> f.shift], 1u);
    }
    __syncthreads();
    {  // inclusive prefix sum over 4096 buckets: 4 per thread + a scan of the 1024 thread totals in sh.list
        const int b0 = threadIdx.x * (kBuckets / kThreads);
        unsigned int v0 = sh.prefix[b0], v1 = sh.prefix[b0 + 1], v2 = sh.prefix[b0 + 2], v3 = sh.prefix[b0 + 3];
        v1 += v0;
        v2 += v1;
        v3 += v2;
        sh.list[threadIdx.x] = v3;
        __syncthreads();
        for (int off = 1; off < kThreads; off <<= 1) {
            const unsigned int add = threadIdx.x >= (unsigned)off ? sh.list[threadIdx.x - off] : 0u;
            __syncthreads();
            sh.list[threadIdx.x] += add;
            __syncthreads();
        }
        const unsigned int before = sh.list[threadIdx.x] - v3;
        sh.prefix[b0] = v0 + before;
        sh.prefix[b0 + 1] = v1 + before;
        sh.prefix[b0 + 2] = v2 + before;
        sh.prefix[b0 + 3] = v3 + before;
        __syncthreads();
    }

    Window w;
    w.lo = 1;
    w.hi = 0x7f7fffffu;
    w.c_lo = 0;
    w.c_le_hi = cnt;
    w.n = cnt;
    double median = 0.0, sigma = 1.0;
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
        select_values(K, sh, f, w.c_lo + (w.n % 2 == 0 ? mid - 1 : mid), w.c_lo + mid, &ka, &kb);
        median = w.n % 2 == 0 ? ((double)__uint_as_float(ka) + (double)__uint_as_float(kb)) / 2.0 : (double)__uint_as_float(kb);
        // median_f32_mut of the deviations (median.rs:46-63): f32 average for even n
        uint32_t da, db;
        select_devs(K, sh, f, nb, w, median, w.n % 2 == 0 ? mid - 1 : mid, mid, &da, &db);
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
        w.lo = w.lo > klo ? w.lo : klo;
        w.hi = w.hi < khi ? w.hi : khi;
        if (w.lo > w.hi) {
            w.n = 0;
            w.c_lo = w.c_le_hi = 0;
            continue;
        }
        count_below(K, sh, w.lo, w.hi, &w.c_lo, &w.c_le_hi);
        w.n = w.c_le_hi - w.c_lo;
    }
    res.median = median;
    res.sigma = sigma;
    return res;
}

}  // namespace tb
