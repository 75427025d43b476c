// Cost of a register sweep over 128 VGPR-resident keys per thread, 512 threads, one workgroup per CU (developer tool).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/sweep_bench.hip -o build/sweep_bench
// Variants: 0 = count (k-1 < a, k-1 < b) through the VGPR index register (what tile_bucket.hpp does), 1 = range test + ballot,
// 2 = count with ONE index-mode window per iteration (inline asm), 3 = count, keys staged 32 at a time through LDS (no indexing)
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
typedef uint32_t u32x32 __attribute__((ext_vector_type(32)));
struct Keys { u32x32 v[4]; };
constexpr int kSweeps = 64;

template <int V>
__global__ __launch_bounds__(512) void k(const uint32_t *in, uint32_t *out, long long *cyc, uint32_t a, uint32_t b) {
    Keys K;
#pragma unroll
    for (int i = 0; i < 128; ++i) K.v[i >> 5][i & 31] = in[(size_t)blockIdx.x * 65536 + i * 512 + threadIdx.x];
    __shared__ uint32_t stage[512 * 4];
    uint32_t ca = 0, cb = 0;
    __syncthreads();
    const long long t0 = clock64();
#pragma unroll 1
    for (int s = 0; s < kSweeps; ++s) {
        const uint32_t aa = a + s, bb = b + s;
        if (V == 0) {
#pragma unroll 1
            for (int j = 0; j < 32; ++j) {
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const uint32_t k1 = K.v[c][j] - 1u;
                    ca += k1 < aa ? 1u : 0u;
                    cb += k1 < bb ? 1u : 0u;
                }
            }
        } else if (V == 1) {
            unsigned long long any = 0;
#pragma unroll 1
            for (int j = 0; j < 32; ++j) {
#pragma unroll
                for (int c = 0; c < 4; ++c) any |= __builtin_amdgcn_ballot_w64(K.v[c][j] - aa <= bb);
            }
            ca += (uint32_t)__builtin_popcountll(any);
        } else if (V == 2) {
#pragma unroll 1
            for (int j = 0; j < 32; ++j) {
                uint32_t k0, k1, k2, k3;
                // one index window: four indexed moves back to back
                asm volatile("s_set_gpr_idx_on %4, gpr_idx(SRC0)\n v_mov_b32 %0, %5\n v_mov_b32 %1, %6\n v_mov_b32 %2, %7\n v_mov_b32 %3, %8\n s_set_gpr_idx_off"
                             : "=&v"(k0), "=&v"(k1), "=&v"(k2), "=&v"(k3)
                             : "s"(j), "v"(K.v[0][0]), "v"(K.v[1][0]), "v"(K.v[2][0]), "v"(K.v[3][0])
                             : "m0");
                k0 -= 1u; k1 -= 1u; k2 -= 1u; k3 -= 1u;
                ca += (k0 < aa) + (k1 < aa) + (k2 < aa) + (k3 < aa);
                cb += (k0 < bb) + (k1 < bb) + (k2 < bb) + (k3 < bb);
            }
        } else if (V == 3) {
            // no indexing at all: static code over 128 keys is too large, so this variant measures the pure ALU cost on 4 registers
#pragma unroll 1
            for (int j = 0; j < 32; ++j) {
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const uint32_t k1 = K.v[c][0] + (uint32_t)j - 1u;
                    ca += k1 < aa ? 1u : 0u;
                    cb += k1 < bb ? 1u : 0u;
                }
            }
        }
    }
    const long long t1 = clock64();
    // keep every key alive
    uint32_t x = 0;
#pragma unroll
    for (int i = 0; i < 128; ++i) x ^= K.v[i >> 5][i & 31];
    out[blockIdx.x * 512 + threadIdx.x] = ca + cb + x + stage[threadIdx.x & 3];
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int V>
void run(const uint32_t *d, uint32_t *out, long long *cyc) {
    k<V><<<256, 512>>>(d, out, cyc, 1000000000u, 1010000000u);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    (void)hipEventRecord(e0);
    k<V><<<256, 512>>>(d, out, cyc, 1000000000u, 1010000000u);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    long long c;
    (void)hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    printf("variant %d: %.1f us per launch, %lld cycles per sweep (workgroup 0)\n", V, ms * 1000.0f, c / kSweeps);
}

int main() {
    uint32_t *d, *out;
    long long *cyc;
    (void)hipMalloc(&d, 256ull * 65536 * 4);
    (void)hipMemset(d, 0x3c, 256ull * 65536 * 4);
    (void)hipMalloc(&out, 256 * 512 * 4);
    (void)hipMalloc(&cyc, 256 * 8);
    run<0>(d, out, cyc);
    run<1>(d, out, cyc);
    run<2>(d, out, cyc);
    run<3>(d, out, cyc);
    return 0;
}
