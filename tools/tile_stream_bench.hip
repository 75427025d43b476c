// Streaming tile statistics (csrc/tile_stream.hpp) against the register-resident ones (csrc/tile_bucket.hpp), developer tool:
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -DAB_TILE_TIMING -Iastroburst_amd/csrc tools/tile_stream_bench.hip -o build/tile_stream_bench
//   build/tile_stream_bench [mode] [frames]
// mode 0: percentile-normalised sky in [0, 1]; 1: raw ADU sky; 2: raw ADU sky normalised ON LOAD (ab_pixel_xf on: what the
// registration path does); 3: mode 2 with a gradient, NaN patches and a zero border.  Prints both kernels' time per frame of
// 256 tiles (frames > 1: that many frames in one launch), how many tiles the streaming kernel declined and why, whether every
// accepted tile equals the resident kernel's bit for bit, and the streaming kernel's phase cycles.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <random>
#include <vector>
#include "tile_stream.hpp"

__global__ __launch_bounds__(tb::kThreads) void k_resident(const float *img, int cols, ab_pixel_xf xf, double *out) {
    __shared__ tb::Shared sh;
    const int ntx = cols / 256;
    const int tile = blockIdx.x % (ntx * ntx);
    img += (size_t)(blockIdx.x / (ntx * ntx)) * cols * cols;
    const int ty0 = (tile / ntx) * 256, tx0 = (tile % ntx) * 256;
    const int tx = threadIdx.x & 255, ty = threadIdx.x >> 8;
    tb::Keys K;
    tb::KeyRange kr;
#pragma unroll
    for (int i = 0; i < tb::kSlots; ++i) {
        const float v = ab_px(xf, img[(size_t)(ty0 + ty + (tb::kThreads / 256) * i) * cols + tx0 + tx]);
        const uint32_t key = (__builtin_isfinite(v) && v > 1e-7f) ? __float_as_uint(v) : 0u;
        K.v[i >> 5][i & 31] = key;
        kr.add(key);
    }
    const tb::TileResult r = tb::tile_stats(K, sh, kr);
    if (threadIdx.x == 0) {
        out[3 * blockIdx.x] = r.median;
        out[3 * blockIdx.x + 1] = r.sigma;
        out[3 * blockIdx.x + 2] = r.valid;
    }
}

__global__ __launch_bounds__(ts::kThreads) void k_stream(const float *img, int cols, ab_pixel_xf xf, double *out, int *declined, long long *phases) {
    __shared__ ts::Shared sh;
    const int ntx = cols / 256;
    const int tile = blockIdx.x % (ntx * ntx);
    ts::TileRect r;
    r.img = img + (size_t)(blockIdx.x / (ntx * ntx)) * cols * cols;
    r.ld = cols;
    r.y0 = (tile / ntx) * 256;
    r.x0 = (tile % ntx) * 256;
    r.y1 = r.y0 + 256;
    r.x1 = r.x0 + 256;
    r.vec = true;
    const ts::TileResult res = ts::tile_stats(sh, r, xf);
    if ((int)threadIdx.x == 64 * ts::rounds_wave()) {
        out[3 * blockIdx.x] = res.median;
        out[3 * blockIdx.x + 1] = res.sigma;
        out[3 * blockIdx.x + 2] = res.valid;
        declined[blockIdx.x] = res.declined;
#ifdef AB_TILE_TIMING
        for (int i = 0; i < 16; ++i) phases[16 * blockIdx.x + i] = sh.t_phase[i];
#endif
    }
}

int main(int argc, char **argv) {
    const int rows = 4096, cols = 4096, mode = argc > 1 ? atoi(argv[1]) : 0, frames = argc > 2 ? atoi(argv[2]) : 1;
    const size_t P = (size_t)rows * cols;
    std::vector<float> h(P * frames);
    std::mt19937 rng(1);
    const bool raw = mode != 0;
    std::normal_distribution<float> sky(raw ? 1300.0f : 0.2f, raw ? 30.0f : 0.06f);
    std::uniform_real_distribution<float> u(0.f, 1.f);
    for (size_t i = 0; i < h.size(); ++i) {
        const size_t p = i % P;
        const int r = (int)(p / cols), c = (int)(p % cols);
        float x = sky(rng);
        if (u(rng) < 0.01f) x += (raw ? 20000.0f : 0.5f) * u(rng);  // star pixels
        if (mode == 0) x = x < 0.f ? 0.f : (x > 1.f ? 1.f : x);    // normalize_for_detection clamps
        if (mode == 3) {
            x += 0.02f * (float)r + 0.01f * (float)c;                                       // a gradient of 80 + 40 ADU
            if ((r / 300) % 5 == 2 && (c / 500) % 4 == 1) x = std::nanf("");                 // NaN patches
            if (r < 40 || c < 24 || r >= rows - 17 || c >= cols - 60) x = 0.0f;              // zero border
        }
        h[i] = x;
    }
    ab_pixel_xf xf;
    if (mode >= 2) {
        xf.on = 1;
        xf.lo = 1235.0;  // ~ the 1st percentile
        xf.inv = 1.0 / (2400.0 - 1235.0);
    }
    const int ntiles = 256 * frames;
    float *d;
    double *o_res, *o_str;
    int *decl;
    long long *ph;
    hipMalloc(&d, h.size() * 4);
    hipMalloc(&o_res, ntiles * 3 * 8);
    hipMalloc(&o_str, ntiles * 3 * 8);
    hipMalloc(&decl, ntiles * 4);
    hipMalloc(&ph, ntiles * 16 * 8);
    hipMemcpy(d, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    float ms_res, ms_str;
    const int reps = 10;
    k_resident<<<ntiles, tb::kThreads>>>(d, cols, xf, o_res);
    hipEventRecord(e0);
    for (int i = 0; i < reps; ++i) k_resident<<<ntiles, tb::kThreads>>>(d, cols, xf, o_res);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    hipEventElapsedTime(&ms_res, e0, e1);
    k_stream<<<ntiles, ts::kThreads>>>(d, cols, xf, o_str, decl, ph);
    hipEventRecord(e0);
    for (int i = 0; i < reps; ++i) k_stream<<<ntiles, ts::kThreads>>>(d, cols, xf, o_str, decl, ph);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    hipEventElapsedTime(&ms_str, e0, e1);
    if (hipGetLastError() != hipSuccess || hipDeviceSynchronize() != hipSuccess) {
        printf("HIP error\n");
        return 1;
    }
    printf("mode %d, %d frame(s) per launch: resident %.1f us per frame, streaming %.1f us per frame\n", mode, frames, ms_res * 1000.0f / reps / frames,
           ms_str * 1000.0f / reps / frames);
    std::vector<double> a(ntiles * 3), b(ntiles * 3);
    std::vector<int> dc(ntiles);
    std::vector<long long> p(ntiles * 16);
    hipMemcpy(a.data(), o_res, a.size() * 8, hipMemcpyDeviceToHost);
    hipMemcpy(b.data(), o_str, b.size() * 8, hipMemcpyDeviceToHost);
    hipMemcpy(dc.data(), decl, dc.size() * 4, hipMemcpyDeviceToHost);
    hipMemcpy(p.data(), ph, p.size() * 8, hipMemcpyDeviceToHost);
    int ndecl = 0, nbad = 0, why[16] = {};
    for (int t = 0; t < ntiles; ++t) {
        if (dc[t]) {
            ++ndecl;
            ++why[dc[t] & 15];
            continue;
        }
        if (memcmp(&a[3 * t], &b[3 * t], 24) != 0) {
            if (nbad < 5) printf("  MISMATCH tile %d: resident (%.17g, %.17g, %g) streaming (%.17g, %.17g, %g)\n", t, a[3 * t], a[3 * t + 1], a[3 * t + 2], b[3 * t], b[3 * t + 1], b[3 * t + 2]);
            ++nbad;
        }
    }
    printf("tiles %d, declined %d, mismatches %d; decline reasons:", ntiles, ndecl, nbad);
    for (int i = 1; i < 16; ++i)
        if (why[i]) printf(" [%d]=%d", i, why[i]);
    printf("\n");
    const char *names[16] = {"sample", "pass1", "scan", "plan", "zones", "pass2", "r:rest", "-", "v:gather", "v:select", "d:geo", "d:first", "d:gather", "d:select", "d:proof", "edges"};
    for (int t = 0; t < 3; ++t) {
        printf("tile %d: median %.6g sigma %.6g |", t, b[3 * t], b[3 * t + 1]);
        long long tot = 0;
        for (int i = 0; i < 16; ++i) {
            printf(" %s %lld;", names[i], p[16 * t + i]);
            tot += p[16 * t + i];
        }
        printf(" TOTAL %lld\n", tot);
    }
    return nbad ? 2 : 0;
}
