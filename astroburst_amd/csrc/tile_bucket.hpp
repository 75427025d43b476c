// Sigma-clipped median / MAD of one background tile (<= 256 x 256 px) by ONE workgroup, exactly (gfx950).
//
// Replaces sigma_clipped_stats(values, 3.0, 2) of estimate_background's per-tile body
// (core/analysis/star_detection.rs:47-68, math/sigma_clip.rs:4-34, math/median.rs:27-63).
//
// Round 1 ran an 11/11/10-bit radix select per order statistic: 13 LDS-histogram sweeps over the tile (~18 VALU
// instructions and one LDS atomic per pixel and sweep; 180 us per 4096^2 frame, 36 % of a frame's registration time).
// Here the tile is histogrammed ONCE and everything else is answered from that histogram plus cheap register sweeps:
//
//   * the 128 pixels of a thread stay in VGPRs as monotone u32 keys (valid pixels are positive finite floats, so the
//     IEEE bit pattern is the key; 0 = not a candidate);
//   * one sweep with LDS atomics builds a 4096-bucket histogram of the keys, turned into an inclusive prefix sum.  The
//     buckets are uniform in key space over a ZOOM window [mean - 6 d, mean + 6 d] of a 512-pixel sample (d = its mean
//     absolute deviation) -- a few dozen pixels per bucket around the median -- with one catch-all bucket on each side for
//     the tails (star pixels, the clamped 0s and 1s of a normalised frame);
//   * the clipping windows of sigma_clipped_stats only ever cut tails, so "elements of the current window below bucket
//     b" is a closed form of that one prefix sum (clamped between the counts below the window's ends) -- the histogram
//     is never rebuilt;
//   * an order statistic of the VALUES: the prefix sum names its bucket, a register sweep (one compare per pixel, no
//     atomics: every wave appends to its own segment of an LDS list) gathers that bucket's keys, one wave selects in the list;
//   * an order statistic of the DEVIATIONS |v - median| (the MAD): the deviations of a bucket's pixels lie between the
//     deviations of its two boundary keys, so a threshold at a bucket boundary yields a lower and an upper bound on the
//     count of deviations below it from the prefix sum alone; a two-round search over the boundaries brackets the wanted
//     deviation between two of them, and only the pixels of the buckets straddling that bracket -- a few dozen -- are
//     gathered (as deviation keys) and selected from;
//   * a list that would overflow (heavy ties: a flat tile is one bucket of 65 536 equal keys; a rank inside a catch-all
//     bucket) falls back to bisection on the key with counting sweeps; exact as well, just slower.
//
// Everything is integer comparison on keys plus the reference's own f64 / f32 arithmetic for the deviation
// (`(v as f64 - median).abs() as f32`), so tile medians and sigmas are bit-identical to the oracle's
// (tests/test_gpu_tile_stats.py compares every tile of adversarial images).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

#include "ab_common.hpp"

namespace tb {

// 512 threads x 128 pixels, not 1024 x 64: with four waves per SIMD the register budget is 128 VGPRs, and 64 resident keys
// plus the working set of the sweeps spilled ~90 registers -- every spilled key cost a serialised scratch round trip per sweep
// (measured: 330 us per frame, slower than round 1).  Two waves per SIMD have 256 VGPRs; the sweeps are ALU work, not latency.
constexpr int kThreads = 512;  // one workgroup per tile
constexpr int kWaves = kThreads / 64;
constexpr int kSlots = 128;    // pixels per thread: 256 x 256 / 512
constexpr int kBuckets = 4096;
constexpr int kOwn = 8;         // gathered candidates a THREAD may hold (its private column of the list); more = overflow
constexpr int kListCap = 2048;  // gathered candidates a select works on (u32 keys)

struct Shared {
    unsigned int prefix[kBuckets];  // bucket counts, then their inclusive prefix sum
    unsigned int list[(kOwn + 1 + kSlots / 32) * kThreads];  // while gathering: row r, column t = the r-th candidate of thread t (+ overflow rows)
    unsigned int tmp[kListCap];     // the candidates compacted to the front (what the select reads) / scan scratch
    unsigned int wave_part[kWaves * 4];  // per-wave partials of the block reductions
    unsigned int found_part[kWaves];     // per-wave totals of a gather (compact_list)
    unsigned int sel_hist[3][256];       // block_select2: rotating digit histograms
    unsigned int sel_part[2 * kWaves];   // block_select2: per-wave partials
    unsigned int scal[16];   // broadcast slots
    long long t_phase[16];   // AB_TILE_TIMING: cycles per phase (thread 0): 0 setup + histogram + scan + window counts; value select:
                             // 1 bucket search 2 gather 3 select; deviation select: 4 round 1 5 round 2 + bounds 6 gather 7 select
    long long t_mark;
};

#ifdef AB_TILE_TIMING
#define TB_MARK(sh, i)                             \
    do {                                           \
        if (threadIdx.x == 0) {                    \
            const long long now_ = clock64();      \
            (sh).t_phase[i] += now_ - (sh).t_mark; \
            (sh).t_mark = now_;                    \
        }                                          \
    } while (0)
#else
#define TB_MARK(sh, i) \
    do {               \
    } while (0)
#endif

// The keys live in VGPRs as four 32-element vectors, and every sweep is a REAL loop of 32 iterations over a uniform index j
// that reads v[0][j] .. v[3][j] through the VGPR index register (s_set_gpr_idx_on: no scratch).  The first version unrolled
// its sweeps fully -- 128 slots x 6 sweeps x ~30 instructions = 180 KB of straight-line code, each line of it executed once:
// the sweeps ran at the instruction-fetch rate (~100 cycles per 64-byte line: 14 000 cycles for a sweep whose ALU work is
// 2 500), and the whole kernel was no faster than round 1's.  Looped, a sweep's body is a few hundred bytes.
typedef uint32_t u32x32 __attribute__((ext_vector_type(32)));
constexpr int kVecs = kSlots / 32;
struct Keys {
    u32x32 v[kVecs];  // key of slot s = v[s >> 5][s & 31]; 0 = not a candidate
};
// k[c] = v[c][j] for the four vectors inside ONE index-register window.  The compiler opens a window (s_set_gpr_idx_on / _off, two
// SALU instructions with their hazards: ~20 cycles) around every indexed move it emits, four per iteration of a sweep -- 40 % of a
// sweep (tools/sweep_bench.hip).  Inline assembly cannot name "element j of this operand", but it can name a register: the four
// vectors are tied to v[104:231] by the operand constraints (the allocator then keeps them there for good: no copies appear), and
// the window's moves address v104 / v136 / v168 / v200 + j.  The rest of the kernel lives below and just above them (the kernel
// must stay at 240 registers: at two waves per SIMD that leaves 32 per SIMD for a wave of the f64 warp beside a tile workgroup).
#ifndef AB_TILE_NO_PINNED_KEYS
static_assert(kVecs == 4, "read_keys names four register ranges");
__device__ __forceinline__ void read_keys(const Keys &K, int j, uint32_t (&k)[kVecs]) {
    // s_set_gpr_idx_on writes M0 (and MODE.gpr_idx_en, which _off restores).  M0 cannot be named in the clobber list (a reserved
    // register: the compiler only warns and ignores it), so the statement saves and restores it itself: whatever the compiler keeps
    // in M0 -- LDS-direct bases, readlane selectors, its own indirect indexing -- survives.
    uint32_t m0_save;
    asm volatile(
        "s_mov_b32 %4, m0\n\t"
        "s_set_gpr_idx_on %5, gpr_idx(SRC0)\n\t"
        "v_mov_b32 %0, v104\n\t"
        "v_mov_b32 %1, v136\n\t"
        "v_mov_b32 %2, v168\n\t"
        "v_mov_b32 %3, v200\n\t"
        "s_set_gpr_idx_off\n\t"
        "s_mov_b32 m0, %4"
        : "=&v"(k[0]), "=&v"(k[1]), "=&v"(k[2]), "=&v"(k[3]), "=&s"(m0_save)
        : "s"(j), "{v[104:135]}"(K.v[0]), "{v[136:167]}"(K.v[1]), "{v[168:199]}"(K.v[2]), "{v[200:231]}"(K.v[3]));
}
#else
__device__ __forceinline__ void read_keys(const Keys &K, int j, uint32_t (&k)[kVecs]) {
#pragma unroll
    for (int c = 0; c < kVecs; ++c) k[c] = K.v[c][j];
}
#endif
// f(key) for every key of the thread
template <class F>
__device__ __forceinline__ void for_each_key(const Keys &K, F f) {
#pragma unroll 1
    for (int j = 0; j < 32; ++j) {
        uint32_t k[kVecs];
        read_keys(K, j, k);
#pragma unroll
        for (int c = 0; c < kVecs; ++c) f(k[c]);
    }
}

// Wave reductions on DPP moves (row_shr 1/2/4/8 inside the rows of 16, then row_bcast 15 / 31 across them; lanes without a source
// take the identity): twelve VALU instructions and a v_readlane.  The __shfl_xor butterfly these replaced is six dependent
// ds_bpermute round trips per value -- four values per workgroup reduction, nine reductions per tile: ~13 000 of a tile's
// 230 000 cycles.
enum { OP_SUM = 0, OP_MIN = 1, OP_MAX = 2 };
template <int OP>
__device__ __forceinline__ unsigned int wave_reduce(unsigned int x) {
    constexpr unsigned int id = OP == OP_MIN ? 0xffffffffu : 0u;
    auto op = [](unsigned int a, unsigned int b) { return OP == OP_SUM ? a + b : (OP == OP_MIN ? min(a, b) : max(a, b)); };
    x = op(x, (unsigned int)__builtin_amdgcn_update_dpp((int)id, (int)x, 0x111, 0xf, 0xf, false));
    x = op(x, (unsigned int)__builtin_amdgcn_update_dpp((int)id, (int)x, 0x112, 0xf, 0xf, false));
    x = op(x, (unsigned int)__builtin_amdgcn_update_dpp((int)id, (int)x, 0x114, 0xf, 0xf, false));
    x = op(x, (unsigned int)__builtin_amdgcn_update_dpp((int)id, (int)x, 0x118, 0xf, 0xf, false));
    x = op(x, (unsigned int)__builtin_amdgcn_update_dpp((int)id, (int)x, 0x142, 0xa, 0xf, false));  // row_bcast:15 into rows 1 and 3
    x = op(x, (unsigned int)__builtin_amdgcn_update_dpp((int)id, (int)x, 0x143, 0xc, 0xf, false));  // row_bcast:31 into rows 2 and 3
    return (unsigned int)__builtin_amdgcn_readlane((int)x, 63);
}
__device__ __forceinline__ unsigned int wave_sum(unsigned int x) { return wave_reduce<OP_SUM>(x); }
__device__ __forceinline__ unsigned int wave_min(unsigned int x) { return wave_reduce<OP_MIN>(x); }
__device__ __forceinline__ unsigned int wave_max(unsigned int x) { return wave_reduce<OP_MAX>(x); }

// inclusive prefix sum over the 64 lanes with DPP moves (row_shr 1/2/4/8, row_bcast 15/31: the gfx9 scan idiom) -- six VALU
// instructions instead of six ds_bpermute round trips
__device__ __forceinline__ unsigned int wave_scan_incl(unsigned int x) {
    const int lane = threadIdx.x & 63;
    unsigned int t;
    t = (unsigned int)__builtin_amdgcn_update_dpp(0, (int)x, 0x111, 0xf, 0xf, false);
    if ((lane & 15) >= 1) x += t;
    t = (unsigned int)__builtin_amdgcn_update_dpp(0, (int)x, 0x112, 0xf, 0xf, false);
    if ((lane & 15) >= 2) x += t;
    t = (unsigned int)__builtin_amdgcn_update_dpp(0, (int)x, 0x114, 0xf, 0xf, false);
    if ((lane & 15) >= 4) x += t;
    t = (unsigned int)__builtin_amdgcn_update_dpp(0, (int)x, 0x118, 0xf, 0xf, false);
    if ((lane & 15) >= 8) x += t;
    t = (unsigned int)__builtin_amdgcn_update_dpp(0, (int)x, 0x142, 0xf, 0xf, false);
    if ((lane & 31) >= 16) x += t;
    t = (unsigned int)__builtin_amdgcn_update_dpp(0, (int)x, 0x143, 0xf, 0xf, false);
    if (lane >= 32) x += t;
    return x;
}

// the wave's 64 values sorted ascending across its lanes (bitonic network on lane shuffles: 21 exchanges)
__device__ __forceinline__ uint32_t wave_sort64(uint32_t x) {
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int k = 2; k <= 64; k <<= 1) {
#pragma unroll
        for (int j = k >> 1; j >= 1; j >>= 1) {
            const uint32_t y = (uint32_t)__shfl_xor((int)x, j, 64);
            const bool keep_min = ((lane & k) == 0) == ((lane & j) == 0);
            x = keep_min ? min(x, y) : max(x, y);
        }
    }
    return x;
}

// up to four values reduced over the workgroup; every thread gets the results (two barriers)
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
    for (int i = 1; i < kWaves; ++i) {
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

// ---- histogram geometry: catch-all bucket 0 | nc uniform buckets over [zlo, zhi] | catch-all bucket nb - 1 -----------------------------
struct Frame {
    uint32_t kmin, kmax;  // extreme candidate keys
    uint32_t zlo, zhi;    // zoom window (kmin <= zlo <= zhi <= kmax)
    int shift;            // central bucket = 1 + ((key - zlo) >> shift)
    int nb;               // buckets in use (nc + 2)
    __device__ __forceinline__ int bucket_of(uint32_t key) const {
        if (key < zlo) return 0;
        if (key > zhi) return nb - 1;
        return 1 + (int)((key - zlo) >> shift);
    }
    __device__ __forceinline__ uint32_t first_key(int b) const {  // smallest key that maps to bucket b (b < nb)
        if (b <= 0) return kmin;
        if (b >= nb - 1) return zhi + 1u;  // zhi <= kmax <= 0x7f7fffff: no wrap
        return zlo + ((uint32_t)(b - 1) << shift);
    }
    __device__ __forceinline__ uint32_t last_key(int b) const {  // largest key that maps to bucket b
        if (b <= 0) return zlo - 1u;  // zlo >= kmin >= 1
        if (b >= nb - 1) return kmax > zhi ? kmax : zhi + 1u;
        const uint64_t k = (uint64_t)zlo + ((uint64_t)b << shift) - 1u;
        return k > (uint64_t)zhi ? zhi : (uint32_t)k;
    }
};

// count of candidates with key < a and with key <= b (one register sweep, no atomics)
__device__ __forceinline__ void count_below(const Keys &K, Shared &sh, uint32_t a, uint32_t b, unsigned int *lt_a, unsigned int *le_b) {
    unsigned int ca = 0, cb = 0;
    // k != 0 && k < a  <=>  k - 1 < a - 1 (unsigned, a >= 1; the non-candidate 0 wraps to the top); k != 0 && k <= b  <=>  k - 1 < b
    const uint32_t a1 = a ? a - 1u : 0u;
    for_each_key(K, [&](uint32_t k) {
        const uint32_t k1 = k - 1u;
        ca += (k1 < a1) ? 1u : 0u;
        cb += (k1 < b) ? 1u : 0u;
    });
    unsigned int z0 = 0, z1 = 0;
    block_reduce4<OP_SUM, OP_SUM, OP_SUM, OP_SUM>(sh, ca, cb, z0, z1);
    *lt_a = ca;
    *le_b = cb;
}

// count of window candidates whose deviation key is <= d
__device__ __forceinline__ unsigned int count_dev_le(const Keys &K, Shared &sh, uint32_t wlo, uint32_t whi, double median, uint32_t d) {
    unsigned int c = 0;
    for_each_key(K, [&](uint32_t k) {
        if (k >= wlo && k <= whi && k != 0) c += dev_key(k, median) <= d ? 1u : 0u;
    });
    unsigned int z0 = 0, z1 = 0, z2 = 0;
    block_reduce4<OP_SUM, OP_SUM, OP_SUM, OP_SUM>(sh, c, z0, z1, z2);
    return c;
}

struct GatherArgs {
    uint32_t lo, hi;                      // outer key range (lo <= hi, lo >= 1)
    uint32_t x1_lo, x1_hi, x2_lo, x2_hi;  // excluded runs (empty: lo > hi)
    double median;
};
// One gather sweep over the thread's keys: every key in [lo, hi] outside the excluded runs goes to the thread's OWN column of
// the list -- row = how many it has found so far: no ballots, no lane ranks, a predicated store per key.  (The first version
// appended per wave through ballot + mbcnt under a per-group branch; the rings of a deviation gather hold one key in a hundred,
// so nine groups in ten took the slow path: 25 000 cycles per sweep.)  Returns the thread's count, kOwn + 1 meaning "more than
// kOwn" (compact_list reports the overflow).
template <bool DEV>
__device__ __forceinline__ unsigned int gather_sweep(const Keys &K, Shared &sh, const GatherArgs &g) {
    // the wanted keys as (at most) two plain runs [l1, l1 + w1] and [l2, l2 + w2]: the outer range with ONE excluded run cut
    // out of it -- the usual shape of a deviation gather (two rings around the median).  With two excluded runs (the median's
    // own bucket left between them) the outer range is tested and the runs are taken out explicitly.
    constexpr uint32_t kNever = 0xffffffffu;  // run [kNever, kNever]: k - kNever <= 0 never holds for a key
    const bool two_excl = DEV && g.x1_lo <= g.x1_hi && g.x2_lo <= g.x2_hi;
    uint32_t l1 = g.lo, w1 = g.hi - g.lo, l2 = kNever, w2 = 0;
    if (DEV && !two_excl) {
        const uint32_t ex_lo = g.x1_lo <= g.x1_hi ? g.x1_lo : g.x2_lo, ex_hi = g.x1_lo <= g.x1_hi ? g.x1_hi : g.x2_hi;
        if (ex_lo <= ex_hi && ex_hi >= g.lo && ex_lo <= g.hi) {  // an excluded run that meets the outer range
            if (ex_lo > g.lo) {
                l1 = g.lo;
                w1 = ex_lo - 1u - g.lo;
            } else {
                l1 = kNever;
                w1 = 0;
            }
            if (ex_hi < g.hi) {
                l2 = ex_hi + 1u;
                w2 = g.hi - l2;
            }
        }
    }
    // the thread's write cursor: a row index; rows 0 .. kOwn - 1 are what compact_list reads, row kOwn marks "more than kOwn"
    // (the cursor is clamped there once per group of kVecs keys, so up to kVecs - 1 further rows can be written in between)
    unsigned int cnt = 0;
    unsigned int *const col = &sh.list[threadIdx.x];
    auto put = [&](uint32_t k, bool in) {
        if (in) {
            col[cnt * kThreads] = k;
            ++cnt;
        }
    };
    if constexpr (!DEV) {
        // a value gather wants one bucket's keys -- one key in a thousand: four keys share a wave-wide test, and only the one
        // group in five that holds a hit anywhere in the wave goes on to the stores (6 instructions per key instead of 10)
#pragma unroll 1
        for (int j = 0; j < 32; ++j) {
            uint32_t k[kVecs];
            unsigned long long any = 0;
            read_keys(K, j, k);
#pragma unroll
            for (int c = 0; c < kVecs; ++c) any |= __builtin_amdgcn_ballot_w64(k[c] - l1 <= w1);
            if (any) {
#pragma unroll
                for (int c = 0; c < kVecs; ++c) put(k[c], k[c] - l1 <= w1);
                cnt = cnt < (unsigned int)(kOwn + 1) ? cnt : (unsigned int)(kOwn + 1);
            }
        }
        return cnt;
    }
#pragma unroll 1
    for (int j = 0; j < 32; ++j) {
        uint32_t kk[kVecs];
        read_keys(K, j, kk);
#pragma unroll
        for (int c = 0; c < kVecs; ++c) {
            const uint32_t k = kk[c];
            bool in = (k - l1 <= w1) | (DEV & (k - l2 <= w2));  // lo <= k <= hi in one unsigned compare
            if (two_excl) in = in & !((k >= g.x1_lo) & (k <= g.x1_hi)) & !((k >= g.x2_lo) & (k <= g.x2_hi));
            put(k, in);
        }
        cnt = cnt < (unsigned int)(kOwn + 1) ? cnt : (unsigned int)(kOwn + 1);
    }
    return cnt;
}

// After the sweep: the columns compacted to the front of sh.tmp in thread order (as deviation keys if DEV).  Returns the total
// (every thread), or 0xffffffff if a thread found more than kOwn or the total exceeds kListCap.
template <bool DEV>
__device__ __forceinline__ unsigned int compact_list(Shared &sh, unsigned int cnt, double median) {
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const unsigned int incl = wave_scan_incl(cnt);
    const unsigned int over = __builtin_amdgcn_ballot_w64(cnt > (unsigned int)kOwn) ? 1u : 0u;
    if (lane == 63) sh.found_part[w] = incl | (over << 31);
    __syncthreads();
    unsigned int before = 0, total = 0, bad = 0;
#pragma unroll
    for (int i = 0; i < kWaves; ++i) {
        const unsigned int v = sh.found_part[i];
        const unsigned int n = v & 0x7fffffffu;
        bad |= v >> 31;
        before += i < w ? n : 0u;
        total += n;
    }
    if (bad || total > (unsigned int)kListCap) {
        __syncthreads();
        return 0xffffffffu;
    }
    const unsigned int at = before + incl - cnt;
    for (unsigned int i = 0; i < cnt; ++i) {
        const uint32_t k = sh.list[i * kThreads + threadIdx.x];
        sh.tmp[at + i] = DEV ? dev_key(k, median) : k;
    }
    __syncthreads();
    return total;
}

// The ranks r_hi and r_lo (= r_hi or r_hi - 1; 0-based) of sh.tmp[0 .. n), n <= kListCap, by the whole workgroup: a thread keeps
// its <= 4 keys in registers; an 8-bit radix descent for r_hi over the bytes in which the keys differ at all (the keys of one
// bucket share their high bytes), one barrier per byte -- every wave scans the 256-bin histogram for itself, so nothing is
// broadcast, and three histograms rotate so that clearing the next one never meets a reader of the last.  Then the predecessor
// of r_hi, if it is a different key.  Every thread returns both keys.  (One wave doing all of this alone took 9 400 cycles per
// deviation select.)  The caller's last barrier lies after tmp was filled; none is needed after the call before tmp is refilled,
// because a workgroup reduction always comes first.
__device__ __forceinline__ void block_select2(Shared &sh, unsigned int n, unsigned int r_lo, unsigned int r_hi, uint32_t *k_lo, uint32_t *k_hi) {
    constexpr int kMine = kListCap / kThreads;
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
    if (t < 256) sh.sel_hist[0][t] = 0;
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
        if (t < 256) next[t] = 0;  // last read two bytes ago: a barrier lies in between
#pragma unroll
        for (int q = 0; q < kMine; ++q)
            if ((unsigned int)(q * kThreads + t) < n && (mine[q] & mask) == prefix) atomicAdd(&hist[(mine[q] >> shift) & 255u], 1u);
        __syncthreads();
        // lane l owns digits 4l .. 4l+3
        const unsigned int c0 = hist[4 * lane], c1 = hist[4 * lane + 1], c2 = hist[4 * lane + 2], c3 = hist[4 * lane + 3];
        const unsigned int own = c0 + c1 + c2 + c3;
        const unsigned int incl = wave_scan_incl(own);
        const unsigned int excl = incl - own;
        const bool owner = rank >= excl && rank < incl;
        uint32_t digit = 0;
        unsigned int below = 0;
        if (owner) {
            const unsigned int r = rank - excl;
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
        h = h == 2 ? 0 : h + 1;
    }
    *k_hi = prefix;
    *k_lo = prefix;
    // `rank` is now r_hi's position among the keys equal to it: r_hi - rank keys are smaller.  The element before r_hi is the same
    // key unless r_hi is its first occurrence; then it is the largest key below.
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
    TB_MARK(sh, 1);
    // key range of the two buckets (everything in between is empty: the ranks are adjacent)
    const uint32_t key_lo = f.first_key((int)b_lo), key_hi = f.last_key((int)b_hi);
    const unsigned int in_range = sh.prefix[b_hi] - ex_lo;
    if (in_range <= (unsigned int)kListCap) {
        GatherArgs ga;
        ga.lo = key_lo;  // (>= 1: a non-candidate never matches)
        ga.hi = key_hi >= key_lo ? key_hi : key_lo;
        ga.x1_lo = ga.x2_lo = 1;
        ga.x1_hi = ga.x2_hi = 0;
        ga.median = 0.0;
        TB_MARK(sh, 2);
        const unsigned int found = gather_sweep<false>(K, sh, ga);
        TB_MARK(sh, 14);
        const unsigned int n = compact_list<false>(sh, found, 0.0);
        TB_MARK(sh, 2);
        if (n != 0xffffffffu) {
            block_select2(sh, n, g_lo - ex_lo, g_hi - ex_lo, k_lo, k_hi);
            TB_MARK(sh, 3);
            return;
        }
    }
    // too many keys in those buckets for the list: bisect on the key with counting sweeps (smallest key x with count(<= x) > rank)
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
    TB_MARK(sh, 3);
}

// ---- order statistics of the DEVIATION keys ---------------------------------------------------------------------------------
struct Window {
    uint32_t lo, hi;             // candidate keys retained so far: lo <= key <= hi
    unsigned int c_lo, c_le_hi;  // candidates with key < lo / key <= hi
    unsigned int n;              // = c_le_hi - c_lo
};

// window candidates with key < first key of bucket b  (b in [0, nb])
__device__ __forceinline__ unsigned int wprefix(const Shared &sh, const Window &w, int b) {
    const unsigned int p = b <= 0 ? 0u : sh.prefix[b - 1];
    const unsigned int q = p < w.c_lo ? w.c_lo : (p > w.c_le_hi ? w.c_le_hi : p);
    return q - w.c_lo;
}

struct DevCtx {
    Frame f;
    double median;
    int bm;  // the bucket of the median: the last bucket whose first key is <= median
    __device__ __forceinline__ bool left_of_median(uint32_t key) const { return (double)__uint_as_float(key) <= median; }
    // Smallest bucket b in [0, bm] whose boundary key (FIRST or LAST key of the bucket) has deviation <= t, scanning the
    // side left of the median; bm + 1 if none.  The deviation of a key left of the median falls as the key rises, so the
    // predicate is monotone in b; the search starts at `hint` (or at the bucket of the float nearest median - t) and steps.
    template <bool LAST>
    __device__ __forceinline__ int left_edge(uint32_t t, int hint) const {
        int b = hint;
        if (b < 0) {
            const double x = median - (double)__uint_as_float(t);
            b = x <= 0.0 ? 0 : f.bucket_of(__float_as_uint((float)x));
        }
        if (b > bm) b = bm;
        auto ok = [&](int bb) {
            const uint32_t key = LAST ? f.last_key(bb) : f.first_key(bb);
            return !left_of_median(key) || dev_key(key, median) <= t;  // a boundary key right of the median: the bucket holds the median
        };
        while (b > 0 && ok(b - 1)) --b;
        while (b <= bm && !ok(b)) ++b;
        return b;
    }
    // Largest bucket b in [bm, nb) whose boundary key has deviation <= t on the right side; bm - 1 if none.
    template <bool LAST>
    __device__ __forceinline__ int right_edge(uint32_t t, int hint) const {
        int b = hint;
        if (b < 0) {
            const double x = median + (double)__uint_as_float(t);
            b = f.bucket_of(x >= 3.4028234663852886e38 ? 0x7f7fffffu : __float_as_uint((float)x));
        }
        if (b < bm) b = bm;
        auto ok = [&](int bb) {
            const uint32_t key = LAST ? f.last_key(bb) : f.first_key(bb);
            return left_of_median(key) || dev_key(key, median) <= t;
        };
        while (b < f.nb - 1 && ok(b + 1)) ++b;
        while (b >= bm && !ok(b)) --b;
        return b;
    }
};

// window candidates in buckets that lie ENTIRELY at deviation <= t (full) / that reach down to deviation <= t (any);
// also the two bucket ranges themselves ([fl, fr] full with bm possibly left out, [al, ar] any)
struct Bounds {
    unsigned int n_full, n_any;
    int fl, fr, al, ar;
    bool bm_full;
};
__device__ __forceinline__ Bounds dev_bounds(const Shared &sh, const Window &w, const DevCtx &c, uint32_t t, int hint_l, int hint_r) {
    const int bm = c.bm;
    Bounds o;
    // left of the median a bucket's largest deviation sits at its FIRST key, its smallest at its LAST key; mirrored on the right
    o.fl = c.left_edge<false>(t, hint_l);
    o.fr = c.right_edge<true>(t, hint_r);
    o.al = c.left_edge<true>(t, o.fl > 0 ? o.fl - 1 : 0);  // the any-run starts at most a bucket or so further out than the full run
    o.ar = c.right_edge<false>(t, o.fr < c.f.nb - 1 ? o.fr + 1 : o.fr);
    // the bucket that holds the median has deviations on both sides of it: full only if both of its boundary keys qualify
    o.bm_full = o.fl <= bm && o.fr >= bm;
    if (!o.bm_full) {
        // full buckets are then [fl, bm - 1] on the left and [bm + 1, fr] on the right: two runs
        const unsigned int left = o.fl <= bm - 1 ? wprefix(sh, w, bm) - wprefix(sh, w, o.fl) : 0u;
        const unsigned int right = o.fr >= bm + 1 ? wprefix(sh, w, o.fr + 1) - wprefix(sh, w, bm + 1) : 0u;
        o.n_full = left + right;
    } else {
        o.n_full = wprefix(sh, w, o.fr + 1) - wprefix(sh, w, o.fl);
    }
    // "any": counting the median's bucket even when it does not reach below t keeps n_any a valid upper bound
    if (o.al > bm) o.al = bm;
    if (o.ar < bm) o.ar = bm;
    o.n_any = wprefix(sh, w, o.ar + 1) - wprefix(sh, w, o.al);
    return o;
}

// ranks r_lo <= r_hi (adjacent or equal, 0-based) of the deviation keys of the window's candidates
__device__ __forceinline__ void select_devs(const Keys &K, Shared &sh, const Frame &f, const Window &w, double median, unsigned int r_lo,
                                            unsigned int r_hi, float mad_guess, uint32_t *d_lo, uint32_t *d_hi) {
    DevCtx c;
    c.f = f;
    c.median = median;
    {
        const float mf = (float)median;  // may round up past the median: step back below
        int b = mf <= 0.0f ? 0 : f.bucket_of(__float_as_uint(mf));
        while (b > 0 && !c.left_of_median(f.first_key(b))) --b;
        while (b < f.nb - 1 && c.left_of_median(f.first_key(b + 1))) ++b;
        c.bm = b;
    }
    // ---- fast path: guess the ring, verify afterwards ------------------------------------------------------------------------
    // F~(t) = window candidates in the buckets that meet [median - t, median + t] is a step-function approximation of
    // F(t) = #{deviation <= t}, good to a bucket or so on either side.  512 threads evaluate it at 4096 thresholds one bucket
    // width apart (integer arithmetic on the prefix sum: no deviations are formed) and the first threshold t* with
    // F~(t*) >= r_hi + 1 places the wanted deviations.  The candidates are then the buckets within kRing (2) bucket widths of
    // median -+ t* (two rings), the buckets between the rings count as "smaller" and everything outside as "larger" -- a GUESS,
    // turned into a proof after the select: the largest deviation possible between the rings must not exceed the smaller
    // selected deviation and the smallest possible outside must not fall below the larger one (deviations fall towards the
    // median and rise away from it, so both extremes sit at run ends).  A guess that fails, overflows the list or meets a
    // catch-all bucket drops to the rigorous bracket search below (binade boundaries inside the window, bimodal tiles).
    // Measured: the bracket search cost 32 000 of a tile's 130 000 cycles per clipping iteration, this path ~2 000.
    if (c.bm > 0 && c.bm < f.nb - 1) {
        constexpr int kRing = 2;
        constexpr int PERT = kBuckets / kThreads;
        const float mf = (float)median;
        const float delta = __uint_as_float(f.first_key(c.bm) + (1u << f.shift)) - __uint_as_float(f.first_key(c.bm));
        auto bucket_at = [&](float x) { return x > 0.0f ? f.bucket_of(x < 3.0e38f ? __float_as_uint(x) : 0x7f7fffffu) : 0; };
        auto ring_of = [&](float t, int *bl, int *br) {  // the buckets that hold median - t and median + t, either side of bm
            const int l = bucket_at(mf - t), r = bucket_at(mf + t);
            *bl = l < c.bm ? l : c.bm;
            *br = r > c.bm ? r : c.bm;
        };
        unsigned int first = 0xffffffffu, z0 = 0, z1 = 0, z2 = 0;
        // where to look first: one threshold per thread in a window of 512 around the expected MAD (the sample's, then the
        // previous iteration's).  The window's answer is THE first threshold iff it lies strictly inside the window (F~ is
        // monotone: below the window's start it is smaller still); otherwise all 4096 thresholds are evaluated, 8 per thread.
        bool windowed = false;
        if (mad_guess > 0.0f && delta > 0.0f) {
            const float gi = mad_guess / delta;
            const unsigned int guess = gi < 3500.0f ? (unsigned int)gi : 3500u, i_lo = guess > 256u ? guess - 256u : 0u;
            const unsigned int i = i_lo + threadIdx.x;
            int bl, br;
            ring_of((float)(i + 1u) * delta, &bl, &br);
            const unsigned int F = wprefix(sh, w, br + 1) - wprefix(sh, w, bl);
            if (F >= r_hi + 1u) first = i;
            block_reduce4<OP_MIN, OP_SUM, OP_SUM, OP_SUM>(sh, first, z0, z1, z2);
            windowed = first != 0xffffffffu && (first > i_lo || i_lo == 0u);
        }
        if (!windowed) {
            first = 0xffffffffu;
            z0 = z1 = z2 = 0;
#pragma unroll
            for (int j = 0; j < PERT; ++j) {
                const unsigned int i = (unsigned int)(j * kThreads) + threadIdx.x;
                int bl, br;
                ring_of((float)(i + 1u) * delta, &bl, &br);
                const unsigned int F = wprefix(sh, w, br + 1) - wprefix(sh, w, bl);
                if (F >= r_hi + 1u) first = first < i ? first : i;
            }
            block_reduce4<OP_MIN, OP_SUM, OP_SUM, OP_SUM>(sh, first, z0, z1, z2);
        }
        TB_MARK(sh, 4);
        if (first != 0xffffffffu && delta > 0.0f) {
            int bLo, bRo, bLi = 0, bRi = 0;
            ring_of((float)(first + 1u + kRing) * delta, &bLo, &bRo);
            const bool has_inner_t = first + 1u > (unsigned int)kRing;
            if (has_inner_t) ring_of((float)(first + 1u - kRing) * delta, &bLi, &bRi);
            const bool has_inner = has_inner_t && bLi + 1 <= bRi - 1;  // inner buckets: strictly between the two rings
            const unsigned int c_in = has_inner ? wprefix(sh, w, bRi) - wprefix(sh, w, bLi + 1) : 0u;
            const unsigned int n_cand = wprefix(sh, w, bRo + 1) - wprefix(sh, w, bLo) - c_in;
            if (bLo > 0 && bRo < f.nb - 1 && c_in <= r_lo && c_in + n_cand >= r_hi + 1u && n_cand <= (unsigned int)kListCap) {
                GatherArgs ga;
                ga.lo = f.first_key(bLo) > w.lo ? f.first_key(bLo) : w.lo;
                ga.hi = f.last_key(bRo) < w.hi ? f.last_key(bRo) : w.hi;
                ga.x1_lo = ga.x2_lo = 1;
                ga.x1_hi = ga.x2_hi = 0;
                if (has_inner) {
                    ga.x1_lo = f.first_key(bLi + 1);
                    ga.x1_hi = f.last_key(bRi - 1);
                }
                ga.median = median;
                TB_MARK(sh, 6);
                const unsigned int found = gather_sweep<true>(K, sh, ga);
                TB_MARK(sh, 14);
                const unsigned int n = compact_list<true>(sh, found, median);
                TB_MARK(sh, 6);
                if (n == n_cand) {
                    uint32_t sel_lo, sel_hi;
                    block_select2(sh, n, r_lo - c_in, r_hi - c_in, &sel_lo, &sel_hi);
                    // the proof: inner deviations <= sel_lo, outer deviations >= sel_hi
                    uint32_t t_in = 0, t_out = 0xffffffffu;
                    if (has_inner) {
                        const uint32_t ka = ga.x1_lo > w.lo ? ga.x1_lo : w.lo, kb = ga.x1_hi < w.hi ? ga.x1_hi : w.hi;
                        if (ka <= kb) {
                            const uint32_t da = dev_key(ka, median), db = dev_key(kb, median);
                            t_in = da > db ? da : db;
                        }
                    }
                    if (ga.lo > w.lo) t_out = dev_key(ga.lo - 1u, median);  // (ga.lo - 1 lies left of the median: bLo <= bm)
                    if (ga.hi < w.hi) {
                        const uint32_t d = dev_key(ga.hi + 1u, median);
                        t_out = t_out < d ? t_out : d;
                    }
                    TB_MARK(sh, 7);
                    if (t_in <= sel_lo && sel_hi <= t_out) {
                        *d_lo = sel_lo;
                        *d_hi = sel_hi;
                        return;
                    }
                }
            }
        }
    }
    TB_MARK(sh, 13);
    // ---- rigorous path ----------------------------------------------------------------------------------------------------
    // A bucket's largest deviation t is a threshold with two bounds on F(t) = #{deviation <= t} from the prefix sum alone:
    //   N_any(t) <= r_lo      =>  the wanted deviations are > t        (best such t: the largest)
    //   N_full(t) >= r_hi + 1 =>  the wanted deviations are <= t       (best such t: the smallest)
    // Both bounds grow with t, so a probe that cannot tighten the bracket found so far is skipped.
    unsigned int t_lo = 0, t_hi = 0xffffffffu, z0 = 0, z1 = 0;  // t_lo is stored + 1 (0 = none): the first key above the bound
    auto probe = [&](int b) {
        if (b < 0 || b >= f.nb) return;
        const uint32_t dl = dev_key(f.first_key(b), median), dh = dev_key(f.last_key(b), median);
        const uint32_t t = dl > dh ? dl : dh;
        if (t + 1 <= t_lo || t >= t_hi) return;  // cannot improve the bracket
        const Bounds o = dev_bounds(sh, w, c, t, b <= c.bm ? b : -1, b >= c.bm ? b : -1);
        if (o.n_any <= r_lo) t_lo = t + 1;
        if (o.n_full >= r_hi + 1) t_hi = t;
    };
    // Round 1: one bucket in eight, one per thread.
    constexpr int PER = kBuckets / kThreads;
    probe(threadIdx.x * PER + PER / 2);
    block_reduce4<OP_MAX, OP_MIN, OP_SUM, OP_SUM>(sh, t_lo, t_hi, z0, z1);
    TB_MARK(sh, 4);
    // Round 2: the buckets whose threshold lies inside round 1's bracket are a run of at most PER on either side of the
    // median.  Their probes go to the lanes of wave 0, one bucket each, instead of queueing up in the threads that own them.
    {
        const uint32_t hi_t = t_hi == 0xffffffffu ? 0x7f800000u : t_hi;
        const int l_from = c.left_edge<false>(hi_t, -1);                             // left run: largest deviation <= hi_t ...
        const int l_to = t_lo ? c.left_edge<false>(t_lo - 1, -1) - 1 : c.bm;        // ... and > t_lo - 1
        const int r_to = c.right_edge<true>(hi_t, -1);
        const int r_from = t_lo ? c.right_edge<true>(t_lo - 1, -1) + 1 : c.bm;
        if (threadIdx.x < 64) {
            const int lane = threadIdx.x;
            // lanes 0 .. 31 walk the left run outwards from its inner end, lanes 32 .. 63 the right run; a run longer than 32
            // buckets (never, after round 1) is simply not refined beyond its first 32
            const int b = lane < 32 ? l_to - lane : r_from + (lane - 32);
            const bool mine = lane < 32 ? (b >= l_from && b <= l_to) : (b >= r_from && b <= r_to);
            if (mine) probe(b);
            t_lo = wave_max(t_lo);
            t_hi = wave_min(t_hi);
            if (lane == 0) {
                sh.scal[4] = t_lo;
                sh.scal[5] = t_hi;
            }
        }
        __syncthreads();
        t_lo = sh.scal[4];
        t_hi = sh.scal[5];
        __syncthreads();
    }
    // candidates: window pixels outside the buckets entirely below t_lo and inside the buckets reaching below t_hi
    const bool have_lo = t_lo != 0;
    Bounds lo_b, hi_b;
    unsigned int c0 = 0;
    uint32_t x1_lo = 1, x1_hi = 0, x2_lo = 1, x2_hi = 0;  // excluded key runs (entirely below t_lo); empty
    if (have_lo) {
        lo_b = dev_bounds(sh, w, c, t_lo - 1, -1, -1);
        c0 = lo_b.n_full;
        const int bm = c.bm;
        const int l_end = lo_b.bm_full ? lo_b.fr : (bm - 1 < lo_b.fr ? bm - 1 : lo_b.fr);
        if (lo_b.fl <= bm && lo_b.fl <= l_end) {
            x1_lo = f.first_key(lo_b.fl);
            x1_hi = f.last_key(l_end);
        }
        if (!lo_b.bm_full && lo_b.fr >= bm + 1) {
            x2_lo = f.first_key(bm + 1);
            x2_hi = f.last_key(lo_b.fr);
        }
    }
    uint32_t a_lo_key = w.lo, a_hi_key = w.hi;  // outer range (t_hi always exists for r_hi < n; the window is the safe default)
    unsigned int n_cand = w.n - c0;
    if (t_hi != 0xffffffffu) {
        hi_b = dev_bounds(sh, w, c, t_hi, -1, -1);
        a_lo_key = f.first_key(hi_b.al);
        a_hi_key = f.last_key(hi_b.ar);
        n_cand = hi_b.n_any - c0;
    }
    if (a_lo_key < w.lo) a_lo_key = w.lo;
    if (a_hi_key > w.hi) a_hi_key = w.hi;
    TB_MARK(sh, 5);
    if (n_cand <= (unsigned int)kListCap && a_lo_key <= a_hi_key) {
        GatherArgs ga;
        ga.lo = a_lo_key;
        ga.hi = a_hi_key;
        ga.x1_lo = x1_lo;
        ga.x1_hi = x1_hi;
        ga.x2_lo = x2_lo;
        ga.x2_hi = x2_hi;
        ga.median = median;
        const unsigned int found = gather_sweep<true>(K, sh, ga);
        const unsigned int n = compact_list<true>(sh, found, median);  // == n_cand
        TB_MARK(sh, 6);
        if (n != 0xffffffffu) {
            block_select2(sh, n, r_lo - c0, r_hi - c0, d_lo, d_hi);
            TB_MARK(sh, 7);
            return;
        }
    }
    // heavy ties / a bracket inside a catch-all bucket: bisect on the deviation key (smallest d with count(dev <= d) > rank)
    uint32_t res[2];
    for (int which = 0; which < 2; ++which) {
        const unsigned int r = which ? r_hi : r_lo;
        if (which && r_hi == r_lo) {
            res[1] = res[0];
            break;
        }
        uint32_t lo = t_lo, hi = t_hi == 0xffffffffu ? 0x7f800000u : t_hi;  // (t_lo is stored + 1: the first key above the bound)
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
    TB_MARK(sh, 7);
}

struct TileResult {
    double median, sigma;
    int valid;
};

// one of this thread's keys as a sample: waves take different slots (different rows of the tile); wave-uniform switch
__device__ __forceinline__ uint32_t sample_key(const Keys &K) {
    switch ((threadIdx.x >> 6) & 7) {
        case 0: return K.v[0][8];
        case 1: return K.v[0][24];
        case 2: return K.v[1][8];
        case 3: return K.v[1][24];
        case 4: return K.v[2][8];
        case 5: return K.v[2][24];
        case 6: return K.v[3][8];
        default: return K.v[3][24];
    }
}

// min / max / count of a thread's candidate keys, accumulated while the keys are formed (static register indices: three
// instructions per key, where a separate sweep through the index register cost 30 000 cycles)
struct KeyRange {
    unsigned int kmin1 = 0xffffffffu, kmax = 0, cnt = 0;  // kmin1 = min(key - 1): the non-candidate 0 wraps to the top and never wins
    __device__ __forceinline__ void add(uint32_t k) {
        kmin1 = min(kmin1, k - 1u);
        kmax = max(kmax, k);
        cnt += k ? 1u : 0u;
    }
};

// sigma_clipped_stats(values, 3.0, 2) of the candidates in K (sigma_clip.rs:4-34); needs >= 8 candidates (star_detection.rs:61)
__device__ __forceinline__ TileResult tile_stats(const Keys &K, Shared &sh, const KeyRange &kr) {
    constexpr double kMadToSigma = 1.4826;
#ifdef AB_TILE_TIMING
    if (threadIdx.x == 0) {
        for (int i = 0; i < 16; ++i) sh.t_phase[i] = 0;
        sh.t_mark = clock64();
    }
#endif
    // ---- min / max / count of the candidates ----
    unsigned int kmin = kr.kmin1, kmax = kr.kmax, cnt = kr.cnt, z = 0;
    kmin += 1u;  // 0xffffffff + 1 = 0 when the thread holds no candidate: fixed up after the reduction
    if (kmin == 0) kmin = 0xffffffffu;
    block_reduce4<OP_MIN, OP_MAX, OP_SUM, OP_SUM>(sh, kmin, kmax, cnt, z);
    TB_MARK(sh, 9);
    TileResult res = {0.0, 1.0, 0};
    if (cnt < 8) return res;
    res.valid = 1;
    // ---- where to zoom (efficiency only: any window is exact) ----
    // Every wave sorts 64 sample keys (one per thread, different rows per wave) across its lanes and reads off the quartiles of its
    // valid ones; the block averages them: a median and a spread that stars and clamped pixels cannot drag about.  (Round 2's first
    // version used mean +- 6 mean-absolute-deviations of the sample: one percent of star pixels tripled the deviation, and a sky
    // three sigma above zero stretched the window over twenty binades of key space -- buckets of sigma / 15 .. sigma / 60 with
    // 400 .. 1700 pixels each, which is what the selects then had to chew through.)  The buckets are 2^shift keys wide with
    // sigma / 300 .. sigma / 600 per bucket (40 .. 90 pixels in the central ones), 4094 of them centred on the sample median: about
    // +- 3.4 .. 6.8 sigma; the rest of the key range falls into the two catch-all buckets.
    Frame f;
    f.kmin = kmin;
    f.kmax = kmax;
    float mad_guess = 0.0f;
    {
        const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
        const uint32_t sorted = wave_sort64(sample_key(K));
        const int nz = __builtin_popcountll(__builtin_amdgcn_ballot_w64(sorted == 0u)), nv = 64 - nz;
        const uint32_t q1 = (uint32_t)__shfl((int)sorted, nz + nv / 4 < 64 ? nz + nv / 4 : 63, 64);
        const uint32_t q2 = (uint32_t)__shfl((int)sorted, nz + nv / 2 < 64 ? nz + nv / 2 : 63, 64);
        const uint32_t q3 = (uint32_t)__shfl((int)sorted, nz + (3 * nv) / 4 < 64 ? nz + (3 * nv) / 4 : 63, 64);
        __syncthreads();  // (the partial slots may still be read from the reduction above)
        if (lane == 0) {
            sh.wave_part[4 * wv + 0] = q1;
            sh.wave_part[4 * wv + 1] = q2;
            sh.wave_part[4 * wv + 2] = q3;
            sh.wave_part[4 * wv + 3] = nv >= 8 ? 1u : 0u;
        }
        __syncthreads();
        float s1 = 0.0f, s2 = 0.0f, s3 = 0.0f, sn = 0.0f;
#pragma unroll
        for (int i = 0; i < kWaves; ++i) {
            const bool ok = sh.wave_part[4 * i + 3] != 0u;
            s1 += ok ? __uint_as_float(sh.wave_part[4 * i + 0]) : 0.0f;
            s2 += ok ? __uint_as_float(sh.wave_part[4 * i + 1]) : 0.0f;
            s3 += ok ? __uint_as_float(sh.wave_part[4 * i + 2]) : 0.0f;
            sn += ok ? 1.0f : 0.0f;
        }
        uint32_t zlo = kmin, zhi = kmax;
        if (sn > 0.0f) {
            const float Q1 = s1 / sn, Q2 = s2 / sn, Q3 = s3 / sn;
            const float sig = (Q3 - Q1) * (1.0f / 1.349f);
            mad_guess = 0.6745f * sig;  // where select_devs looks first
            uint32_t kc = __float_as_uint(Q2);
            kc = kc < kmin ? kmin : (kc > kmax ? kmax : kc);
            const float up = Q2 + sig;
            const uint32_t wkeys = (sig > 0.0f && up < 3.0e38f) ? __float_as_uint(up) - __float_as_uint(Q2) : 0u;  // sigma in key units
            const uint32_t bk = wkeys / 300u;
            const int shift = bk >= 1u ? 31 - __builtin_clz(bk) : 0;
            const uint64_t half = (uint64_t)(kBuckets - 2) / 2 << shift, width = ((uint64_t)(kBuckets - 2) << shift) - 1u;
            zlo = (uint64_t)kc > half + kmin ? (uint32_t)(kc - half) : kmin;
            zhi = (uint64_t)zlo + width < (uint64_t)kmax ? (uint32_t)(zlo + width) : kmax;
        }
        f.zlo = zlo;
        f.zhi = zhi;
        const uint32_t span = zhi - zlo;
        f.shift = 0;
        while ((span >> f.shift) >= (uint32_t)(kBuckets - 2)) ++f.shift;
        f.nb = (int)(span >> f.shift) + 3;
    }
    TB_MARK(sh, 10);
    // ---- the one histogram sweep ----
    for (int i = threadIdx.x; i < kBuckets; i += kThreads) sh.prefix[i] = 0;
    __syncthreads();
    // Straight-line bucket index for the sweep: with base = zlo - 2^shift, min((key -sat base) >> shift, ..) is 0 below zlo and
    // 1 + ((key - zlo) >> shift) from zlo on; only "above zhi" keeps its compare (a key just above zhi may share the last
    // central bucket's granule).  Non-candidates (key 0) simply land in bucket 0 and are taken out again afterwards -- no
    // predicate, no nested exec regions: 9 instructions per key instead of ~20.  Not when most of the tile is empty (64 lanes
    // adding to one LDS address serialise) or when base would underflow: then the plain form.
    const uint32_t gran = 1u << f.shift;
    if (f.zlo >= gran && cnt * 2u >= (unsigned int)(kThreads * kSlots)) {
        const uint32_t base = f.zlo - gran, zhi = f.zhi, top = (uint32_t)(f.nb - 1);
        const int shift = f.shift;
        for_each_key(K, [&](uint32_t k) {
            const uint32_t t = k > base ? k - base : 0u;
            const uint32_t b = k > zhi ? top : (t >> shift);
            atomicAdd(&sh.prefix[b], 1u);
        });
        __syncthreads();
        if (threadIdx.x == 0) sh.prefix[0] -= (unsigned int)(kThreads * kSlots) - cnt;  // the non-candidates
    } else {
        for_each_key(K, [&](uint32_t k) {
            if (k) atomicAdd(&sh.prefix[f.bucket_of(k)], 1u);
        });
    }
    __syncthreads();
    TB_MARK(sh, 11);
    {  // inclusive prefix sum over 4096 buckets: PER per thread + a scan of the thread totals in sh.tmp
        constexpr int PER = kBuckets / kThreads;
        const int b0 = threadIdx.x * PER;
        unsigned int v[PER];
#pragma unroll
        for (int j = 0; j < PER; ++j) v[j] = sh.prefix[b0 + j] + (j ? v[j - 1] : 0u);
        sh.tmp[threadIdx.x] = v[PER - 1];
        __syncthreads();
        for (int off = 1; off < kThreads; off <<= 1) {
            const unsigned int add = threadIdx.x >= (unsigned)off ? sh.tmp[threadIdx.x - off] : 0u;
            __syncthreads();
            sh.tmp[threadIdx.x] += add;
            __syncthreads();
        }
        const unsigned int before = sh.tmp[threadIdx.x] - v[PER - 1];
#pragma unroll
        for (int j = 0; j < PER; ++j) sh.prefix[b0 + j] = v[j] + before;
        __syncthreads();
    }
    TB_MARK(sh, 12);

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
        select_values(K, sh, f, w.c_lo + (w.n % 2 == 0 ? mid - 1 : mid), w.c_lo + mid, &ka, &kb);
#ifdef AB_TILE_REPEAT  // developer experiment (tools/tile_bench.hip): the same select again, its code now warm in the instruction cache
        select_values(K, sh, f, w.c_lo + (w.n % 2 == 0 ? mid - 1 : mid), w.c_lo + mid, &ka, &kb);
#endif
        median = w.n % 2 == 0 ? ((double)__uint_as_float(ka) + (double)__uint_as_float(kb)) / 2.0 : (double)__uint_as_float(kb);
        // median_f32_mut of the deviations (median.rs:46-63): f32 average for even n
        uint32_t da, db;
        select_devs(K, sh, f, w, median, w.n % 2 == 0 ? mid - 1 : mid, mid, mad_guess, &da, &db);
        const float mad_f32 = w.n % 2 == 0 ? (__uint_as_float(da) + __uint_as_float(db)) / 2.0f : __uint_as_float(db);
        mad_guess = mad_f32;
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
        TB_MARK(sh, 0);
    }
    res.median = median;
    res.sigma = sigma;
    return res;
}

}  // namespace tb
