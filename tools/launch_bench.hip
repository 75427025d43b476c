// How long does hipLaunchKernel take when T host threads launch small kernels onto their own streams at once? (developer tool)
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/launch_bench.hip -o build/launch_bench -lpthread
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <thread>
#include <vector>
__global__ void tiny(int *p) { if (threadIdx.x == 0 && blockIdx.x == 0) p[0] += 1; }
__global__ void busy(int *p, int iters) {
    int x = threadIdx.x;
    for (int i = 0; i < iters; ++i) x = x * 1664525 + 1013904223;
    if (x == 42) p[0] = x;
}
int main() {
    int *d;
    (void)hipMalloc(&d, 4096);
    for (int mode = 0; mode < 2; ++mode)
        for (int T : {1, 4, 8, 12, 16}) {
            std::vector<std::thread> th;
            std::vector<double> launch_us(T), sync_us(T);
            const auto t0 = std::chrono::steady_clock::now();
            for (int t = 0; t < T; ++t)
                th.emplace_back([&, t] {
                    (void)hipSetDevice(0);
                    hipStream_t s;
                    (void)hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
                    double lu = 0, su = 0;
                    for (int rep = 0; rep < 60; ++rep) {          // 60 "segments" of 4 launches + a sync, like a frame's chain
                        const auto a = std::chrono::steady_clock::now();
                        for (int k = 0; k < 4; ++k) {
                            if (mode == 0) hipLaunchKernelGGL(tiny, dim3(1), dim3(64), 0, s, d + t);
                            else hipLaunchKernelGGL(busy, dim3(256), dim3(256), 0, s, d + t, 20000);   // ~20 us of a full GPU
                        }
                        const auto b = std::chrono::steady_clock::now();
                        (void)hipStreamSynchronize(s);
                        const auto c = std::chrono::steady_clock::now();
                        lu += std::chrono::duration<double, std::micro>(b - a).count();
                        su += std::chrono::duration<double, std::micro>(c - b).count();
                    }
                    launch_us[t] = lu / 240;
                    sync_us[t] = su / 60;
                    (void)hipStreamDestroy(s);
                });
            for (auto &x : th) x.join();
            const double wall = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
            double l = 0, s = 0;
            for (int t = 0; t < T; ++t) { l += launch_us[t]; s += sync_us[t]; }
            printf("%s kernels, %2d threads: %.1f us per launch, %.1f us per sync, wall %.1f ms (incl. stream create)\n", mode ? "20-us" : "empty", T, l / T, s / T, wall);
        }
    return 0;
}
