// Are global atomics faster when every XCD adds to a PRIVATE copy with workgroup scope (served in that XCD's L2) than when all add
// to one array with device scope (served at the memory side)?  And do the private copies add up?  (developer tool, gfx950)
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/xcd_atomic_bench.hip -o build/xcd_atomic_bench
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>
constexpr int kBins = 65536, kPer = 256;
__device__ __forceinline__ unsigned int xcc_id() { return __builtin_amdgcn_s_getreg((3 << 11) | 20) & 15u; }  // HW_REG_XCC_ID[3:0]
template <int MODE>
__global__ __launch_bounds__(1024) void k(unsigned long long *hist, unsigned int *xcc_seen) {
    const unsigned int x = xcc_id();
    if (threadIdx.x == 0) atomicOr(&xcc_seen[0], 1u << x);
    unsigned long long *h = MODE == 0 ? hist : hist + (size_t)x * kBins;
    uint32_t s = blockIdx.x * 1024u + threadIdx.x + 12345u;
    for (int i = 0; i < kPer; ++i) {
        s = s * 1664525u + 1013904223u;
        const uint32_t b = (s >> 8) & (kBins - 1);
        if (MODE == 0) __hip_atomic_fetch_add(&h[b], 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else __hip_atomic_fetch_add(&h[b], 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
}
template <int MODE>
void run(unsigned long long *d, unsigned int *seen, const char *name) {
    const int blocks = 1024;
    (void)hipMemset(d, 0, 16ull * kBins * 8);
    (void)hipMemset(seen, 0, 4);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    (void)hipEventRecord(e0);
    k<MODE><<<blocks, 1024>>>(d, seen);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(16ull * kBins);
    unsigned int sn;
    (void)hipMemcpy(h.data(), d, h.size() * 8, hipMemcpyDeviceToHost);
    (void)hipMemcpy(&sn, seen, 4, hipMemcpyDeviceToHost);
    unsigned long long tot = 0;
    for (auto v : h) tot += v;
    printf("%-44s %8.3f ms  %7.1f M atomics/ms  total %llu (expected %llu)  xcc mask 0x%x\n", name, ms, blocks * 1024.0 * kPer / ms / 1e6, tot,
           (unsigned long long)blocks * 1024 * kPer, sn);
}
int main() {
    unsigned long long *d;
    unsigned int *seen;
    (void)hipMalloc(&d, 16ull * kBins * 8);
    (void)hipMalloc(&seen, 4);
    run<0>(d, seen, "one array, device-scope atomics");
    run<1>(d, seen, "a copy per XCD, workgroup-scope atomics");
    run<0>(d, seen, "one array, device-scope atomics");
    run<1>(d, seen, "a copy per XCD, workgroup-scope atomics");
    return 0;
}
