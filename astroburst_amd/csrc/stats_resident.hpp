// compute_image_stats' histogram path (stats.rs:85-210) + auto_stf + apply_stf with the PLANE HELD IN THE REGISTER FILE.
//
// The chain in stats.hip reads the plane five times (scan, three histogram passes, stretch), every pass a kernel of its own
// that starts cold: 0.30 ms for a 4096 x 4096 plane, 15 % of what HBM could deliver.  An MI355X has 256 CUs x 512 KiB of
// vector registers = 128 MiB: a 64 MiB plane fits, 64 pixels per thread of 256 workgroups of 1024.  So ONE kernel loads the plane
// once, every later pass sweeps registers, and the passes are separated by grid-wide barriers instead of kernel boundaries;
// the scalar bookkeeping between the passes (percentile bin, in-bin interpolation) is done by EVERY workgroup on an LDS copy of
// the state -- the same deterministic arithmetic on the same counts, so nothing is broadcast -- and the stretch to u8 reads the
// registers too.  HBM traffic: 4 P read + P written instead of 20 P + P.
//
// What is counted is unchanged -- the same f64 bin index expressions as dense_hist_kernel (the reference's) -- but not how.  All the
// reference takes from a 65 536-bin histogram is the bin where the cumulative count reaches a rank, that bin's count and the
// cumulative count there (find_percentile_bin / interpolate_percentile / resolve_rank_in_hist, stats.rs:302-353).  With the
// pixels in registers a second sweep costs a few microseconds, so each of those is a TWO-LEVEL descent: 256 coarse counts
// (bin >> 8), pick the coarse bin, 256 fine counts inside it (bin & 255).  A workgroup's 256 counts go to its own row of a slab
// in HBM (plain stores), and after the barrier every workgroup adds up the rows it needs (G x 1 KiB, L2-resident): no atomics on
// shared addresses anywhere.  That matters more than the passes themselves: the first version of this kernel kept the chain's
// 65 536-bin LDS histograms and flushed them with global atomics, and took as long as the chain (270 us) -- a sky image puts
// its pixels into a few hundred bins, so 256 workgroups x 2 flushes land on the same few hundred addresses, and same-address
// device-scope atomics retire at one per ~15 .. 50 ns (4096 adds to ONE counter: 60 us; measured with the phase stamps below).
// The chain's passes are bound by the same thing.
//
// In LDS a wave counts into its own 256-bin copy; the bin most of its pixels share (the sky's coarse bin, deviation bin 0) is
// counted in a scalar register by ballot, because 64 lanes adding to one LDS address serialise.
//
// The grid barrier needs all workgroups resident: grid <= number of CUs, one workgroup (1024 threads) per CU.  A barrier that is
// not reached within ~0.3 s (CU mask, another resident kernel of this kind) raises an abort flag, every workgroup leaves, and
// the host re-runs the chain: slow, never wrong.  Included by stats.hip inside its namespace.
#pragma once

constexpr int kResBlock = 1024;                 // == kBookBlock
constexpr int kResVec = 16;                     // float4 loads per thread
constexpr int kResPer = 4 * kResVec;            // pixels per thread
constexpr int kResTile = kResBlock * kResPer;   // 65 536 pixels per workgroup
constexpr int kResWaves = kResBlock / 64;
// A thread's 64 pixels are two 32-element register vectors, and every sweep is a REAL loop of 32 iterations that reads a[j] and
// b[j] through the VGPR index register (cf. tile_bucket.hpp).  Four reads per index window (the vectors tied to fixed registers,
// as tile_bucket.hpp's read_keys does) measured 365 000 cycles against 374 000 for the kernel: not worth the assembly here -- a
// sweep is 17 VALU and ~10 SALU instructions per pixel, and the scalar unit serves a SIMD once in four cycles.  Unrolled sweeps
// would need neither index windows nor loop control, but every attempt spilled 130 .. 630 registers (the optimiser shares sub-expressions between the sweeps and
// parks their ballot masks; laundering the values and pinning the counts brought it to 133, not to 0), and 25 000 instructions
// of straight-line code run at the instruction-fetch rate.
// a scalar count is final where it is written (the optimiser otherwise collects ballot masks and counts them later)
#define AB_RES_PIN(c) asm volatile("" : "+s"(c))
typedef float f32x32 __attribute__((ext_vector_type(32)));
struct Pix {
    f32x32 a, b;  // pixel 4 k + c of the thread (k-th float4 load, component c): k < 8 in a[4 k + c], else in b[4 (k - 8) + c]
};
template <class F>
__device__ __forceinline__ void for_each_pixel(const Pix &P, F f) {
#pragma unroll 1
    for (int j = 0; j < 32; ++j) {
        const float x = P.a[j], y = P.b[j];
        f(x);
        f(y);
    }
}
static_assert(kResBlock == kBookBlock, "the resident kernel shares the chain's bookkeeping bodies");

// All workgroups of the grid; epoch = the number of this barrier (from 1).  false: aborted (uniform in the workgroup).
// Every workgroup publishes the epoch in a flag of its own and polls everybody's (G <= 512 flags: 8 per lane of one wave).  The
// first version counted arrivals in ONE word: 256 same-address atomics retire in ~8 us, which was then the cost of every barrier
// for every workgroup.
//
// What crosses the barrier (a workgroup's slab row, its partial) is published with device-scope write-through stores, each
// waited for (s_waitcnt) before the workgroup's flag is stored, and every published item sits on cache lines that only its
// workgroup writes and that nobody reads before the barrier it belongs to (rows are per level, partials per purpose and 128 bytes
// apart).  A reader therefore cannot hold a stale copy -- caches are clean at kernel start, nothing prefetches -- and reads them
// with plain loads after the poll, sharing them through its XCD's L2.  The formally fenced version (release = L2 write-back
// before the flag, acquire = L2 invalidate after the poll) measured 2 + 2 us per barrier and workgroup on top of the poll.
constexpr int kBarAbort = kResMaxGrid;        // u32 index of the abort flag in `bar`
constexpr int kBarStamps = kResMaxGrid + 32;  // u32 index of the timing stamps (8-byte aligned)
__device__ __forceinline__ bool grid_barrier(unsigned int *bar, unsigned int epoch, unsigned long long *dbg = nullptr, bool fake_timeout = false) {
    __shared__ int s_ok;
    __builtin_amdgcn_s_waitcnt(0);  // this wave's published stores have been written through
    __syncthreads();
    const int t = threadIdx.x;
    if (t < 64) {
        if (dbg && t == 0) dbg[0] = __builtin_amdgcn_s_memtime();
        if (t == 0) __hip_atomic_store(&bar[blockIdx.x], epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (dbg && t == 0) dbg[1] = __builtin_amdgcn_s_memtime();
        int ok = 1;
        unsigned int spins = fake_timeout ? 0xfffffff0u : 0u;  // (test hook: this workgroup has arrived -- and gives up at once)
        for (;;) {
            unsigned int m = 0xffffffffu;
            for (unsigned int b = t; b < gridDim.x; b += 64) m = min(m, __hip_atomic_load(&bar[b], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
            if (!fake_timeout && __builtin_amdgcn_ballot_w64(m < epoch) == 0) break;
            if (++spins > 300000u || __hip_atomic_load(&bar[kBarAbort], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) {
                __hip_atomic_store(&bar[kBarAbort], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                ok = 0;
                break;
            }
            __builtin_amdgcn_s_sleep(1);
        }
        if (dbg && t == 0) dbg[2] = __builtin_amdgcn_s_memtime();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        if (dbg && t == 0) dbg[3] = __builtin_amdgcn_s_memtime();
        if (t == 0) s_ok = ok;
    }
    __syncthreads();
    const bool ok = s_ok != 0;
    __syncthreads();
    return ok;
}

// bin_index() without branches: v_cvt_u32_f64 saturates (NaN and negatives -> 0, too large -> 2^32 - 1), like Rust's `as usize`
__device__ __forceinline__ uint32_t bin16(double t) {
    uint32_t r;
    asm("v_cvt_u32_f64_e32 %0, %1" : "=v"(r) : "v"(t));
    return r < 65535u ? r : 65535u;
}

// A wave's LDS region: histogram A [0, 256), histogram B [256, 512), 64 per-lane dummy counters [512, 576).
constexpr int kWaveLds = 576;
// One of the wave's 256-bin histograms, with its hottest bin counted in a scalar register (64 lanes adding to one LDS address
// serialise).  add() has no branches and never touches exec: a lane with nothing to count adds to its own dummy counter, so the
// pixels of one loop iteration interleave freely (with an exec-masked ds_add per pixel the chains ran one after the other, each
// with four VALU -> SALU -> exec hops: 130 cycles per pixel and wave).
struct WaveHist {
    unsigned int *region;
    uint32_t off, dummy, hot, hotcount;
    __device__ __forceinline__ void begin(unsigned int *wave_region, uint32_t hist_off, uint32_t guess) {
        region = wave_region;
        off = hist_off;
        dummy = 512u + (threadIdx.x & 63u);
        hot = (uint32_t)__builtin_amdgcn_readfirstlane((int)guess) & 255u;
        hotcount = 0;
    }
    __device__ __forceinline__ void add(bool take, uint32_t bin) {  // bin < 256
        const bool hit = bin == hot;
        hotcount += (uint32_t)__popcll(__builtin_amdgcn_ballot_w64(take) & __builtin_amdgcn_ballot_w64(hit));
        AB_RES_PIN(hotcount);
        atomicAdd(&region[(take && !hit) ? off + bin : dummy], 1u);
    }
    __device__ __forceinline__ void end() {
        if ((threadIdx.x & 63) == 0 && hotcount) atomicAdd(&region[off + hot], hotcount);
    }
};

// after the sweep: the workgroup's kResWaves copies of `nh` (1 or 2) histograms -> its slab row (and the copies cleared)
__device__ __forceinline__ void publish_counts(unsigned int *lds, unsigned int *row, int nh) {
    __syncthreads();
    const int t = threadIdx.x;
    if (t < 256 * nh) {
        unsigned int c = 0;
#pragma unroll
        for (int k = 0; k < kResWaves; ++k) {
            c += lds[k * kWaveLds + t];
            lds[k * kWaveLds + t] = 0;
        }
        __hip_atomic_store(&row[t], c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}
// a workgroup's partial (block_reduce_scan's result), published the same way
__device__ __forceinline__ void publish_partial(double mn, double mx, double sum, unsigned long long cnt, ScanPartial *out) {
    __shared__ ScanPartial s_p;
    block_reduce_scan(mn, mx, sum, cnt, &s_p);
    if (threadIdx.x == 0) {
        unsigned long long *o = reinterpret_cast<unsigned long long *>(out);
        __hip_atomic_store(&o[0], (unsigned long long)__double_as_longlong(s_p.mn), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&o[1], (unsigned long long)__double_as_longlong(s_p.mx), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&o[2], (unsigned long long)__double_as_longlong(s_p.sum), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&o[3], s_p.cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// after the barrier: column totals over the grid's rows for `nh` (1 or 2) histograms, into s_tot (LDS, 512 x u64).  Thread
// (q, l) = (t >> 6, t & 63) reads columns 4 l .. 4 l + 3 (uint4) of the rows q, q + 16, ...: for 256 workgroups 16 (32) independent
// loads per thread, all in flight together (four at a time, the first version spent 10 us here on 32 dependent round trips).
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void grid_columns(const unsigned int *slab, unsigned int G, int nh, unsigned long long *s_tot, unsigned int *s_red) {
    const int t = threadIdx.x, l = t & 63, q = t >> 6;
    for (int h = 0; h < nh; ++h) {
        uint4 a = {0, 0, 0, 0};  // (a plane has fewer than 2^31 pixels)
        const uint4 *p = reinterpret_cast<const uint4 *>(slab) + 64 * h + l;
        // device-scope (sc1) loads, the pair of publish_counts' sc1 stores: the rows are rewritten by every launch, and a plain load
        // may be served a line this XCD's L2 kept from the launch before.  An agent-scope atomic load is sc1 only up to 8 bytes (two
        // per 16-byte piece: +25 us per launch), so the 16-byte loads are written out, eight in flight per wait.
        unsigned int r = q;
        for (; r + 7u * kResWaves < G; r += 8u * kResWaves) {
            u32x4 x0, x1, x2, x3, x4, x5, x6, x7;
            const uint4 *b = p + (size_t)r * (kSlabRow / 4);
            constexpr size_t S = (size_t)kResWaves * (kSlabRow / 4);
            asm volatile(
                "global_load_dwordx4 %0, %8, off sc1\n\t"
                "global_load_dwordx4 %1, %9, off sc1\n\t"
                "global_load_dwordx4 %2, %10, off sc1\n\t"
                "global_load_dwordx4 %3, %11, off sc1\n\t"
                "global_load_dwordx4 %4, %12, off sc1\n\t"
                "global_load_dwordx4 %5, %13, off sc1\n\t"
                "global_load_dwordx4 %6, %14, off sc1\n\t"
                "global_load_dwordx4 %7, %15, off sc1\n\t"
                "s_waitcnt vmcnt(0)"
                : "=&v"(x0), "=&v"(x1), "=&v"(x2), "=&v"(x3), "=&v"(x4), "=&v"(x5), "=&v"(x6), "=&v"(x7)
                : "v"(b), "v"(b + S), "v"(b + 2 * S), "v"(b + 3 * S), "v"(b + 4 * S), "v"(b + 5 * S), "v"(b + 6 * S), "v"(b + 7 * S)
                : "memory");
            const u32x4 t = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
            a.x += t.x, a.y += t.y, a.z += t.z, a.w += t.w;
        }
        for (; r < G; r += kResWaves) {
            unsigned long long *src = reinterpret_cast<unsigned long long *>(const_cast<uint4 *>(p + (size_t)r * (kSlabRow / 4)));
            const unsigned long long lo = __hip_atomic_load(&src[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned long long hi = __hip_atomic_load(&src[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            a.x += (unsigned int)lo, a.y += (unsigned int)(lo >> 32), a.z += (unsigned int)hi, a.w += (unsigned int)(hi >> 32);
        }
        reinterpret_cast<uint4 *>(s_red)[q * 128 + 64 * h + l] = a;
    }
    __syncthreads();
    if (t < 256 * nh) {
        unsigned long long c = 0;
#pragma unroll
        for (int k = 0; k < kResWaves; ++k) c += s_red[k * 512 + t];
        s_tot[t] = c;
    }
    __syncthreads();
}

// One level of a two-level descent: 256 counts (one per thread t < 256, zero elsewhere), the rank wanted (>= 1) and the count
// before this level.  Every thread returns the same: found, bin, its count, inclusive cumulative count there.
__device__ __forceinline__ RankHit find_in_256(unsigned long long mine, unsigned long long before, unsigned long long rank) {
    __shared__ unsigned long long s_w[4];
    __shared__ RankHit s_h;
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    unsigned long long cum = mine;
    if (t < 256) {
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const unsigned long long up = __shfl_up(cum, off, 64);
            if (lane >= off) cum += up;
        }
        if (lane == 63) s_w[w] = cum;
    }
    if (t == 0) s_h.found = 0;
    __syncthreads();
    if (t < 256) {
        unsigned long long base = before;
        for (int i = 0; i < w; ++i) base += s_w[i];
        cum += base;
        const bool hit = cum >= rank && cum - mine < rank;  // exactly one bin: counts are >= 0 and the hit bin's is >= 1
        if (hit) {
            s_h.found = 1;
            s_h.bin = (uint32_t)t;
            s_h.count = mine;
            s_h.cum = cum;
        }
    }
    __syncthreads();
    const RankHit r = s_h;
    __syncthreads();
    return r;
}
// the fine level's hit as a hit in the 65 536-bin histogram (the coarse bin's count is the sum of its fine counts and the rank is
// reached inside it, so the fine level always finds)
__device__ __forceinline__ RankHit join_levels(const RankHit &coarse, const RankHit &fine) {
    RankHit r = fine;
    r.bin = coarse.bin * 256u + fine.bin;
    return r;
}

__global__ __launch_bounds__(kResBlock) void stats_resident_kernel(const float *__restrict__ data, int64_t n, StatsDev *st_out, const ResWs w,
                                                                   unsigned int epoch_base, int known, double kmin, double kmax, ab_auto_stf_config cfg,
                                                                   unsigned char *__restrict__ u8, int abort_at) {
    __shared__ unsigned int lds[kResWaves * kWaveLds];  // the waves' counting regions
    __shared__ unsigned long long s_tot[512];
    __shared__ __attribute__((aligned(16))) unsigned int s_red[kResWaves * 512];
    __shared__ StatsDev s_st;
    const int t = threadIdx.x;
    const unsigned int G = gridDim.x;
    unsigned int nbar = epoch_base;  // the barrier flags keep the previous launch's epochs: this launch's are all larger
    int level = 0;
    auto my_row = [&](int lv) { return w.slab + ((size_t)lv * kResMaxGrid + blockIdx.x) * kSlabRow; };
    auto rows_of = [&](int lv) { return w.slab + (size_t)lv * kResMaxGrid * kSlabRow; };
    // AB_STATS_TIMING=1 (host side) prints workgroup 0's s_memtime stamps between the phases
    unsigned long long *stamps = reinterpret_cast<unsigned long long *>(w.bar + kBarStamps);
    int nstamp = 0;
    auto stamp = [&]() {
        if (blockIdx.x == 0 && t == 0) stamps[nstamp] = __builtin_amdgcn_s_memtime();
        ++nstamp;
    };
    stamp();
    auto barrier = [&](unsigned long long *dbg = nullptr) {
        ++nbar;
        return grid_barrier(w.bar, nbar, dbg, abort_at != 0 && blockIdx.x == G - 1u && nbar - epoch_base == (unsigned int)abort_at);
    };

    // ---- the workgroup's 65 536 pixels, all loads in flight at once; beyond the plane: NaN (not a valid pixel) ----
    Pix P;
    const int64_t base4 = (int64_t)blockIdx.x * (kResTile / 4) + t;
    const bool full = (int64_t)(blockIdx.x + 1) * kResTile <= n;  // (uniform) every workgroup but possibly the last
    if (full) {
        const float4 *d4 = reinterpret_cast<const float4 *>(data) + base4;
#pragma unroll
        for (int k = 0; k < kResVec; ++k) {
            const float4 q = d4[(int64_t)k * kResBlock];
            if (k < 8)
                P.a[4 * k] = q.x, P.a[4 * k + 1] = q.y, P.a[4 * k + 2] = q.z, P.a[4 * k + 3] = q.w;
            else
                P.b[4 * k - 32] = q.x, P.b[4 * k - 31] = q.y, P.b[4 * k - 30] = q.z, P.b[4 * k - 29] = q.w;
        }
    } else {
#pragma unroll 1
        for (int j = 0; j < 32; ++j) {  // element j of a / b = pixel 4 k + c with k = j >> 2 (+ 8), c = j & 3
            const int64_t pa = (base4 + (int64_t)(j >> 2) * kResBlock) * 4 + (j & 3), pb = pa + (int64_t)8 * kResBlock * 4;
            P.a[j] = pa < n ? data[pa] : __builtin_nanf("");
            P.b[j] = pb < n ? data[pb] : __builtin_nanf("");
        }
    }
    for (int i = t; i < (int)(sizeof(StatsDev) / 4); i += kResBlock) reinterpret_cast<unsigned int *>(&s_st)[i] = 0;
    for (int i = t; i < kResWaves * kWaveLds; i += kResBlock) lds[i] = 0;
    stamp();  // 1: loads issued
    // every pixel that is not valid (padding, non-finite) becomes NaN: "valid" is then x == x in every sweep, a NaN fails every
    // range test by itself, and apply_stf maps either to 0; the range (f32 min / max, widened afterwards: (double) is monotone and
    // exact; minnum / maxnum ignore a NaN operand) rides along
    float mnf = __builtin_inff(), mxf = -__builtin_inff();
#pragma unroll 1
    for (int j = 0; j < 32; ++j) {
        const float x = P.a[j], y = P.b[j];
        const float xc = is_valid_pixel(x) ? x : __builtin_nanf(""), yc = is_valid_pixel(y) ? y : __builtin_nanf("");
        P.a[j] = xc;
        P.b[j] = yc;
        mnf = fminf(mnf, fminf(xc, yc));
        mxf = fmaxf(mxf, fmaxf(xc, yc));
    }
    __syncthreads();
    unsigned int *const mine = lds + (t >> 6) * kWaveLds;

    // ---- range (stats.rs:212-258), unless the caller knows it (:25-41) ----
    if (!known) {
        const bool any = mnf <= mxf;
        publish_partial(any ? (double)mnf : DBL_MAX, any ? (double)mxf : -DBL_MAX, 0.0, 0ull, &w.p1[4 * blockIdx.x]);
        stamp();  // 2: plane in registers, range swept
        if (!barrier()) return;
        stamp();  // 3: barrier
        double mn, mx, sum;
        unsigned long long cnt;
        reduce_partials<true>(w.p1, (int)G, &mn, &mx, &sum, &cnt, 4);
        if (t == 0) {
            s_st.negmin_max[0] = -mn;
            s_st.negmin_max[1] = mx;
        }
    } else {
        nstamp = 4;
        if (t == 0) {
            s_st.negmin_max[0] = -kmin;
            s_st.negmin_max[1] = kmax;
        }
    }
    if (t == 0) book_range_body(&s_st);
    __syncthreads();

    if (!s_st.empty) {
        // ---- VALUE histogram (stats.rs:260-300): coarse level, with the sum and the count of the valid pixels ----
        const double origin = s_st.gmin, inv = s_st.inv;
        {
            double sum = 0.0;
            uint32_t wave_cnt = 0;
            WaveHist A;
            A.begin(mine, 0u, bin16(((double)P.a[0] - origin) * inv) >> 8);
            for_each_pixel(P, [&](float x) {
                const bool ok = x == x;
                const double vf = (double)x;
                sum += ok ? vf : 0.0;
                wave_cnt += (uint32_t)__popcll(__builtin_amdgcn_ballot_w64(ok));
                AB_RES_PIN(wave_cnt);
                A.add(ok, bin16((vf - origin) * inv) >> 8);
            });
            A.end();
            publish_partial(0.0, 0.0, sum, (t & 63) == 0 ? (unsigned long long)wave_cnt : 0ull, &w.p2[4 * blockIdx.x]);
            publish_counts(lds, my_row(level), 1);
        }
        stamp();  // 4: VALUE coarse sweep
        if (!barrier(blockIdx.x == 0 ? stamps + 19 : nullptr)) return;  // (+ the barrier's own stamps: 19 .. 22)
        stamp();  // 5: barrier
        unsigned long long total;
        {
            double mn, mx, sum;
            reduce_partials<true>(w.p2, (int)G, &mn, &mx, &sum, &total, 4);
            if (t == 0) s_st.sum = sum;
        }
        grid_columns(rows_of(level++), G, 1, s_tot, s_red);
        if (total == 0) {  // stats.rs:95-97 (a known range and no valid pixel)
            if (t == 0) write_zero_result(&s_st);
            __syncthreads();
        } else {
            const unsigned long long half = half_of(total);
            const RankHit cv = find_in_256(t < 256 ? s_tot[t] : 0ull, 0ull, half);  // always found: half <= total
            stamp();  // 6: coarse counts added up, coarse bin found
            // ---- VALUE histogram, fine level: the 256 bins of the coarse bin ----
            {
                WaveHist A;
                A.begin(mine, 0u, bin16(((double)P.a[0] - origin) * inv));
                for_each_pixel(P, [&](float x) {
                    const uint32_t b = bin16(((double)x - origin) * inv);
                    A.add(x == x && (b >> 8) == cv.bin, b & 255u);
                });
                A.end();
                publish_counts(lds, my_row(level), 1);
            }
            stamp();  // 7: VALUE fine sweep
            if (!barrier()) return;
            stamp();  // 8: barrier
            grid_columns(rows_of(level++), G, 1, s_tot, s_red);
            const RankHit fv = find_in_256(t < 256 ? s_tot[t] : 0ull, cv.cum - cv.count, half);
            if (t == 0) book_value_apply(&s_st, total, half, join_levels(cv, fv), 0ull);
            __syncthreads();
        }
    }
    if (!s_st.empty) {
        level = 2;
        nstamp = 9;
        // ---- DEV pass (stats.rs:119-146): deviations from the coarse median (A), sub-bins of the median's bin (B); coarse level ----
        const double refine_lo = s_st.median_bin_lo, refine_hi = s_st.median_bin_hi, refine_inv = s_st.refine_inv, dev_inv = s_st.dev_inv;
        const float cmed = s_st.coarse_med_f32;
        const unsigned long long half = s_st.half_count, before = s_st.count_before_median;
        const unsigned long long rank_in_bin = half > before ? half - before : 0;  // saturating_sub (:148)
        {
            WaveHist A, B;
            A.begin(mine, 0u, 0u);
            B.begin(mine, 256u, 128u);
            for_each_pixel(P, [&](float x) {
                const double vf = (double)x;
                B.add(vf >= refine_lo && vf < refine_hi, bin16((vf - refine_lo) * refine_inv) >> 8);
                A.add(x == x, bin16((double)fabsf(x - cmed) * dev_inv) >> 8);
            });
            A.end();
            B.end();
            publish_counts(lds, my_row(level), 2);
        }
        stamp();  // 9: DEV coarse sweep
        if (!barrier()) return;
        stamp();  // 10: barrier
        grid_columns(rows_of(level++), G, 2, s_tot, s_red);
        const RankHit cd = find_in_256(t < 256 ? s_tot[t] : 0ull, 0ull, half);  // every valid pixel has a deviation bin: found
        RankHit cr;
        cr.found = 0;
        cr.count = cr.cum = 0;
        if (rank_in_bin != 0) cr = find_in_256(t < 256 ? s_tot[256 + t] : 0ull, 0ull, rank_in_bin);
        if (!cr.found) cr.bin = 0xffffffffu;  // no pixel's coarse sub-bin: the fine sweep then counts nothing for B
        // ---- DEV pass, fine level ----
        {
            WaveHist A, B;
            A.begin(mine, 0u, 0u);
            B.begin(mine, 256u, 128u);
            for_each_pixel(P, [&](float x) {
                const double vf = (double)x;
                const uint32_t sb = bin16((vf - refine_lo) * refine_inv);
                B.add(vf >= refine_lo && vf < refine_hi && (sb >> 8) == cr.bin, sb & 255u);
                const uint32_t db = bin16((double)fabsf(x - cmed) * dev_inv);
                A.add(x == x && (db >> 8) == cd.bin, db & 255u);
            });
            A.end();
            B.end();
            publish_counts(lds, my_row(level), 2);
        }
        stamp();  // 11: DEV fine sweep
        if (!barrier()) return;
        stamp();  // 12: barrier
        grid_columns(rows_of(level++), G, 2, s_tot, s_red);
        const RankHit fd = find_in_256(t < 256 ? s_tot[t] : 0ull, cd.cum - cd.count, half);
        const double sub_bw_med = s_st.refine_range / (double)kHistBins;
        double median = refine_lo;  // resolve_rank_in_hist (stats.rs:333-353): rank 0
        if (rank_in_bin != 0) {
            RankHit hit = cr;  // not found at the coarse level = not found
            if (cr.found) hit = join_levels(cr, find_in_256(t < 256 ? s_tot[256 + t] : 0ull, cr.cum - cr.count, rank_in_bin));
            median = resolve_from_hit(hit, rank_in_bin, refine_lo, sub_bw_med);
        }
        if (t == 0) book_dev_apply(&s_st, median, join_levels(cd, fd));
        __syncthreads();

        // ---- MAD refine (stats.rs:166-209): the 65 536 sub-bins of the three deviation bins around the MAD; coarse level ----
        const float center = s_st.exact_med_f32, mad_lo = s_st.mad_lo_f32, mad_hi = s_st.mad_hi_f32;
        const double region_lo = s_st.mad_region_lo, mad_inv = s_st.mad_refine_inv, sub_bw = s_st.mad_refine_range / (double)kHistBins;
        {
            uint32_t below = 0;  // (per wave)
            WaveHist A;
            A.begin(mine, 0u, 128u);
            for_each_pixel(P, [&](float x) {
                const float dev = fabsf(x - center);  // NaN for a pixel that is not valid: fails both tests
                below += (uint32_t)__popcll(__builtin_amdgcn_ballot_w64(dev < mad_lo));
                AB_RES_PIN(below);
                A.add(dev >= mad_lo && dev < mad_hi, bin16(((double)dev - region_lo) * mad_inv) >> 8);
            });
            A.end();
            publish_partial(0.0, 0.0, 0.0, (t & 63) == 0 ? (unsigned long long)below : 0ull, &w.p3[4 * blockIdx.x]);
            publish_counts(lds, my_row(level), 1);
        }
        stamp();  // 13: MAD coarse sweep
        if (!barrier()) return;
        stamp();  // 14: barrier
        grid_columns(rows_of(level++), G, 1, s_tot, s_red);
        unsigned long long nbelow;
        {
            double mn, mx, sum;
            reduce_partials<true>(w.p3, (int)G, &mn, &mx, &sum, &nbelow, 4);
        }
        const unsigned long long rank = half > nbelow ? half - nbelow : 0;
        double mad = region_lo;  // rank 0
        if (rank != 0) {
            const RankHit cm = find_in_256(t < 256 ? s_tot[t] : 0ull, 0ull, rank);
            if (!cm.found) {
                mad = region_lo + (double)kHistBins * sub_bw;
            } else {
                // ---- MAD refine, fine level ----
                {
                    WaveHist A;
                    A.begin(mine, 0u, 128u);
                    for_each_pixel(P, [&](float x) {
                        const float dev = fabsf(x - center);
                        const uint32_t sb = bin16(((double)dev - region_lo) * mad_inv);
                        A.add(dev >= mad_lo && dev < mad_hi && (sb >> 8) == cm.bin, sb & 255u);
                    });
                    A.end();
                    publish_counts(lds, my_row(level), 1);
                }
                stamp();  // 15: MAD fine sweep
                if (!barrier()) return;
                stamp();  // 16: barrier
                grid_columns(rows_of(level++), G, 1, s_tot, s_red);
                const RankHit fm = find_in_256(t < 256 ? s_tot[t] : 0ull, cm.cum - cm.count, rank);
                mad = resolve_from_hit(join_levels(cm, fm), rank, region_lo, sub_bw);
            }
        }
        if (t == 0) finish_result(&s_st, mad, cfg);
    } else if (t == 0) {
        finish_result(&s_st, 0.0, cfg);
    }
    __syncthreads();
    if (blockIdx.x == 0)  // (without the completion marker: that is written last, below)
        for (int i = t; i < (int)(sizeof(StatsDev) / 4); i += kResBlock) reinterpret_cast<unsigned int *>(st_out)[i] = reinterpret_cast<const unsigned int *>(&s_st)[i];
    nstamp = 17;
    stamp();  // 17: result written

    // ---- apply_stf -> u8 (stf.rs:89-102) from the registers ----
    if (u8) {
        const StfTx tx = s_st.tx;
#pragma unroll 1
        for (int k = 0; k < 8; ++k) {
            uchar4 ra, rb;
            ra.x = to_u8(P.a[4 * k], tx), ra.y = to_u8(P.a[4 * k + 1], tx), ra.z = to_u8(P.a[4 * k + 2], tx), ra.w = to_u8(P.a[4 * k + 3], tx);
            rb.x = to_u8(P.b[4 * k], tx), rb.y = to_u8(P.b[4 * k + 1], tx), rb.z = to_u8(P.b[4 * k + 2], tx), rb.w = to_u8(P.b[4 * k + 3], tx);
            const int64_t ia = base4 + (int64_t)k * kResBlock, ib = ia + (int64_t)8 * kResBlock;
            if (full) {
                reinterpret_cast<uchar4 *>(u8)[ia] = ra;
                reinterpret_cast<uchar4 *>(u8)[ib] = rb;
            } else {
                const unsigned char qa[4] = {ra.x, ra.y, ra.z, ra.w}, qb[4] = {rb.x, rb.y, rb.z, rb.w};
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    if (ia * 4 + c < n) u8[ia * 4 + c] = qa[c];
                    if (ib * 4 + c < n) u8[ib * 4 + c] = qb[c];
                }
            }
        }
    }
    stamp();  // 18: stretched
    // ---- completion is a fact about the GRID.  Every workgroup publishes "finished" (epoch base + 8: above this launch's barriers,
    // below the next launch's) once its part of the u8 plane is stored; workgroup 0 writes the host's completion marker only when it
    // has seen all of them.  A workgroup that timed out at the LAST barrier after publishing its arrival there lets its peers pass
    // and finish -- with the marker written by workgroup 0 alone (round 3) the host then took a preview with an unwritten tile for
    // complete.  Now the marker stays away (the abort flag is up, or workgroup 0's wait runs out) and the host re-runs the chain.
    __builtin_amdgcn_s_waitcnt(0);
    __syncthreads();
    const unsigned int fin = epoch_base + 8u;
    if (t == 0) __hip_atomic_store(&w.bar[blockIdx.x], fin, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (blockIdx.x == 0 && t < 64) {
        bool all = false;
        for (unsigned int spins = 0; spins < 300000u; ++spins) {
            unsigned int m = 0xffffffffu;
            for (unsigned int b = t; b < G; b += 64) m = min(m, __hip_atomic_load(&w.bar[b], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
            if (__builtin_amdgcn_ballot_w64(m < fin) == 0) {
                all = true;
                break;
            }
            if (__hip_atomic_load(&w.bar[kBarAbort], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) break;
            __builtin_amdgcn_s_sleep(1);
        }
        if (all && t == 0) st_out->done = (unsigned long long)fin;  // the host's proof that EVERY workgroup of this launch ran to the end
    }
}
