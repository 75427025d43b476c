// Phase timing of the one-histogram tile statistics (csrc/tile_bucket.hpp), developer tool:
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -DAB_TILE_TIMING -Iastroburst_amd/csrc tools/tile_bench.hip -o build/tile_bench
// Runs 256 tiles of 256 x 256 of a synthetic percentile-normalised sky (values clamped to [0, 1], what the registration path
// feeds the kernel) and prints the cycles thread 0 of tile 0 .. 3 spent per phase, plus the kernel's duration.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <random>
#include <vector>
#include "tile_bucket.hpp"

__global__ __launch_bounds__(tb::kThreads) void k(const float *img, int cols, double *out, long long *phases) {
    __shared__ tb::Shared sh;
    const int ntx = cols / 256;
    const int ty0 = (blockIdx.x / ntx) * 256, tx0 = (blockIdx.x % ntx) * 256;
    const int tx = threadIdx.x & 255, ty = threadIdx.x >> 8;
    tb::Keys K;
    tb::KeyRange kr;
    const long long t_start = clock64();
#pragma unroll
    for (int i = 0; i < tb::kSlots; ++i) {
        const float v = img[(size_t)(ty0 + ty + (tb::kThreads / 256) * i) * cols + tx0 + tx];
        const uint32_t key = (__builtin_isfinite(v) && v > 1e-7f) ? __float_as_uint(v) : 0u;
        K.v[i >> 5][i & 31] = key;
        kr.add(key);
    }
    long long t_load = clock64() - t_start;
    if (K.v[0][0] == 0xdeadbeefu) t_load = 0;  // (keeps the reading after the loads)
    const tb::TileResult r = tb::tile_stats(K, sh, kr);
    const long long t_all = clock64() - t_start;
    if (threadIdx.x == 0) {
        out[2 * blockIdx.x] = r.median;
        out[2 * blockIdx.x + 1] = r.sigma;
#ifdef AB_TILE_TIMING
        for (int i = 0; i < 16; ++i) phases[16 * blockIdx.x + i] = sh.t_phase[i];
        phases[16 * blockIdx.x + 8] = t_load;
        phases[16 * blockIdx.x + 15] = t_all;
#endif
    }
}

int main(int argc, char **argv) {
    const int rows = 4096, cols = 4096, mode = argc > 1 ? atoi(argv[1]) : 0;
    std::vector<float> h((size_t)rows * cols);
    std::mt19937 rng(1);
    std::normal_distribution<float> sky(mode == 0 ? 0.2f : 1300.0f, mode == 0 ? 0.06f : 30.0f);
    std::uniform_real_distribution<float> u(0.f, 1.f);
    for (auto &v : h) {
        float x = sky(rng);
        if (u(rng) < 0.01f) x += (mode == 0 ? 0.5f : 20000.0f) * u(rng);   // star pixels
        if (mode == 0) x = x < 0.f ? 0.f : (x > 1.f ? 1.f : x);            // normalize_for_detection clamps
        v = x;
    }
    float *d;
    double *out;
    long long *ph;
    hipMalloc(&d, h.size() * 4);
    hipMalloc(&out, 256 * 2 * 8);
    hipMalloc(&ph, 256 * 16 * 8);
    hipMemcpy(d, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    k<<<256, tb::kThreads>>>(d, cols, out, ph);
    hipEventRecord(e0);
    for (int i = 0; i < 10; ++i) k<<<256, tb::kThreads>>>(d, cols, out, ph);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    printf("mode %d (%s): %.1f us per launch of 256 tiles\n", mode, mode == 0 ? "normalised [0,1] frame" : "raw ADU frame", ms * 100.0f);
    std::vector<long long> p(256 * 16);
    std::vector<double> o(512);
    hipMemcpy(p.data(), ph, p.size() * 8, hipMemcpyDeviceToHost);
    hipMemcpy(o.data(), out, o.size() * 8, hipMemcpyDeviceToHost);
    const char *names[16] = {"window-counts", "v:bucket", "v:gather", "v:select", "d:round1", "d:round2+bounds", "d:gather", "d:select",
                             "load", "minmax-reduce", "zoom-sums", "hist", "scan", "d:to-rigorous", "sweeps(v+d)", "TOTAL"};
    for (int t = 0; t < 3; ++t) {
        printf("tile %d: median %.6g sigma %.6g |", t, o[2 * t], o[2 * t + 1]);
        for (int i = 0; i < 16; ++i) printf(" %s %lld;", names[i], p[16 * t + i]);
        printf("\n");
    }
    return 0;
}
