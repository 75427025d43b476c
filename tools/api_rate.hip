// Developer tool: how many kernel launches / small async copies per second can T host threads push, each on its own
// non-blocking stream?  (bounds the frame-parallel registration: ~27 runtime calls per frame)
//   hipcc --offload-arch=gfx950 -O2 tools/api_rate.hip -o build/api_rate -lpthread
#include <hip/hip_runtime.h>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <thread>
#include <vector>
__global__ void tiny(int *p) { if (threadIdx.x == 0 && blockIdx.x == 0) p[0] += 1; }
int main(int argc, char **argv) {
    for (int T : {1, 2, 4, 8, 16, 24, 32}) {
        std::vector<std::thread> th;
        std::atomic<long> calls{0};
        const auto t0 = std::chrono::steady_clock::now();
        for (int t = 0; t < T; ++t)
            th.emplace_back([&]() {
                hipSetDevice(0);
                hipStream_t s;
                hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
                int *d, *h;
                hipMalloc(&d, 4096);
                hipHostMalloc(&h, 4096);
                for (int it = 0; it < 300; ++it) {
                    for (int k = 0; k < 6; ++k) hipLaunchKernelGGL(tiny, dim3(64), dim3(256), 0, s, d);
                    hipMemsetAsync(d, 0, 64, s);
                    hipMemcpyAsync(h, d, 64, hipMemcpyDeviceToHost, s);
                    hipStreamSynchronize(s);
                    calls += 9;
                }
                hipFree(d);
                hipHostFree(h);
                hipStreamDestroy(s);
            });
        for (auto &x : th) x.join();
        const double sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        printf("threads %2d: %8.0f runtime calls/s  (%.1f us per call overall, %.1f us per call per thread)\n", T, calls / sec, 1e6 * sec / calls,
               1e6 * sec / (calls / T));
    }
    return 0;
}
