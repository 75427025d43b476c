// Developer tool: where label_tile_many_kernel's time goes (round 5).  Built against the library's own source:
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -Iastroburst_amd/csrc -Iinclude tools/label_bench.hip -Lastroburst_amd -lastroburst_hip -o build/label_bench
// Variants: 8 runs + component records | 0 runs, pixel list | 7 the pixel-by-pixel unions | 1 loads only, 32 x 128 tile pattern | 2 loads only, 16 KB contiguous per workgroup | 3 loads + threshold +
// mask words, no labels | 4 loads only, 8 x 512 pattern | 5 loads only, 16 x 256 pattern | 6 as 3 with the labels' LDS allocated (occupancy)
#define AB_LABEL_TIMING 1
#include "../astroburst_amd/csrc/detect.hip"

#include <random>
#include <vector>

namespace {
template <int V>
__global__ __launch_bounds__(256) void bench_kernel(const float *const *imgs, int rows, int cols, double threshold, ab_pixel_xf xf, int **parent, unsigned int **mask,
                                                    int **plist, size_t plist_stride, int **blist, size_t blist_stride, unsigned int **lcnt, float *sink,
                                                    CompStat **st = nullptr, int **roots = nullptr, int **cid = nullptr, size_t rec_stride = 0, unsigned int *flags = nullptr) {
    const int f = blockIdx.y;
    const float *img = imgs[f];
    if constexpr (V == 8) {
        label_tile_body<true, true>(img, rows, cols, threshold, xf, parent[f], mask[f], plist[f], plist_stride, blist[f], blist_stride, lcnt[f],
                                    TileRecOut{st[f], roots[f], cid[f], rec_stride, flags + 4 * f, 0.0125});
    } else if constexpr (V == 0 || V == 7) {
        label_tile_body<V == 0>(img, rows, cols, threshold, xf, parent[f], mask[f], plist[f], plist_stride, blist[f], blist_stride, lcnt[f]);
    } else {
        const int tid = threadIdx.x, lane = tid & 63;
        float4 v[4];
        int r_of[4], c_of;
        if constexpr (V == 2) {
            const float *p = img + (size_t)blockIdx.x * 4096;
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = *reinterpret_cast<const float4 *>(p + 1024 * j + 4 * tid);
            c_of = 0;
        } else {
            constexpr int TW = (V == 4) ? 512 : (V == 5 ? 256 : 128), TH = 4096 / TW, QW = TW / 4, RS = 256 / QW;  // threads per row, rows per step
            const int tiles_x = cols / TW;
            const int ty0 = (int)(blockIdx.x / tiles_x) * TH, tx0 = (int)(blockIdx.x % tiles_x) * TW;
            const int q = tid % QW, r0 = tid / QW;
            c_of = tx0 + 4 * q;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                r_of[j] = ty0 + r0 + RS * j;
                v[j] = *reinterpret_cast<const float4 *>(img + (int64_t)r_of[j] * cols + c_of);
            }
        }
        if constexpr (V == 3 || V == 6) {
            __shared__ unsigned int tmask[kTileH][kTileW / 32];
            __shared__ int lab[V == 6 ? kTileH * kTileW : 1];
            const int q = tid & 31, r0 = tid >> 5;
            unsigned int bits = 0;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const unsigned int b = (unsigned int)(above(ab_px(xf, v[j].x), threshold)) | ((unsigned int)(above(ab_px(xf, v[j].y), threshold)) << 1) |
                                       ((unsigned int)(above(ab_px(xf, v[j].z), threshold)) << 2) | ((unsigned int)(above(ab_px(xf, v[j].w), threshold)) << 3);
                bits |= b << (4 * j);
                unsigned int w = b << (4 * (lane & 7));
                w |= __shfl_xor(w, 1, 64);
                w |= __shfl_xor(w, 2, 64);
                w |= __shfl_xor(w, 4, 64);
                if ((lane & 7) == 0) {
                    tmask[r0 + 8 * j][q >> 3] = w;
                    mask[f][((int64_t)r_of[j] * cols + c_of) >> 5] = w;
                }
            }
            if (V == 6 && bits) lab[tid] = (int)bits;
            __syncthreads();
            if (bits == 0xdeadbeefu) sink[0] = (float)tmask[0][0] + (float)lab[0];
        } else {
            float s = 0.0f;
#pragma unroll
            for (int j = 0; j < 4; ++j) s += v[j].x + v[j].y + v[j].z + v[j].w;
            if (s == 1234.5678f) sink[0] = s;
        }
    }
}

CompStat **g_st;
int **g_roots, **g_cid;
size_t g_rec_stride;
unsigned int *g_flags;
template <int V>
float run(const float *const *d_imgs, int G, int rows, int cols, double thr, ab_pixel_xf xf, int **parent, unsigned int **mask, int **plist, size_t ps, int **blist, size_t bs,
          unsigned int **lcnt, unsigned int *lcnt_flat, size_t lcnt_words, float *sink, int reps) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    float best = 1e30f, tot = 0.0f;
    for (int r = 0; r < reps + 2; ++r) {
        hipMemsetAsync(lcnt_flat, 0, lcnt_words * 4, 0);
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL(bench_kernel<V>, dim3(4096, G), dim3(256), 0, 0, d_imgs, rows, cols, thr, xf, parent, mask, plist, ps, blist, bs, lcnt, sink, g_st, g_roots, g_cid,
                           g_rec_stride, g_flags);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        if (r >= 2) {
            best = std::min(best, ms);
            tot += ms;
        }
    }
    printf("variant %d  G=%d  min %.1f us  avg %.1f us  = %.1f us/frame, %.2f TB/s of pixels\n", V, G, best * 1e3, tot / reps * 1e3, best * 1e3 / G,
           (double)G * rows * cols * 4 / (best * 1e-3) / 1e12);
    return best;
}
}  // namespace

int main(int argc, char **argv) {
    const int rows = 4096, cols = 4096, G = argc > 1 ? atoi(argv[1]) : 4, reps = 20;
    const size_t P = (size_t)rows * cols;
    std::vector<float> h(P);
    std::mt19937 rng(7);
    std::normal_distribution<float> nz(200.0f, 6.0f);
    for (auto &x : h) x = nz(rng);
    std::uniform_real_distribution<float> u(0.0f, 1.0f);
    for (int s = 0; s < 6000; ++s) {
        const float cy = u(rng) * rows, cx = u(rng) * cols, a = 200.0f + 3000.0f * u(rng) * u(rng);
        for (int dy = -6; dy <= 6; ++dy)
            for (int dx = -6; dx <= 6; ++dx) {
                const int y = (int)cy + dy, x = (int)cx + dx;
                if (y < 0 || y >= rows || x < 0 || x >= cols) continue;
                h[(size_t)y * cols + x] += a * expf(-(dy * dy + dx * dx) / 4.5f);
            }
    }
    std::vector<const float *> imgs(G);
    std::vector<int *> parent(G), plist(G), blist(G);
    std::vector<unsigned int *> mask(G), lcnt(G);
    const size_t ps = P / 16 / kRegions, bs = P / 64 / kRegions, lw = (size_t)(2 * kRegions + kRecRegions) * kRegionPitch;
    unsigned int *lcnt_flat;
    hipMalloc(&lcnt_flat, G * lw * 4);
    for (int f = 0; f < G; ++f) {
        float *d;
        hipMalloc(&d, P * 4);
        hipMemcpy(d, h.data(), P * 4, hipMemcpyHostToDevice);
        imgs[f] = d;
        hipMalloc(&parent[f], P * 4);
        hipMalloc(&mask[f], P / 8);
        hipMalloc(&plist[f], ps * kRegions * 4);
        hipMalloc(&blist[f], bs * kRegions * 4);
        lcnt[f] = lcnt_flat + f * lw;
    }
    const float **d_imgs;
    int **d_parent, **d_plist, **d_blist;
    unsigned int **d_mask, **d_lcnt;
    float *sink;
    hipMalloc(&d_imgs, G * 8);
    hipMalloc(&d_parent, G * 8);
    hipMalloc(&d_plist, G * 8);
    hipMalloc(&d_blist, G * 8);
    hipMalloc(&d_mask, G * 8);
    hipMalloc(&d_lcnt, G * 8);
    hipMalloc(&sink, 64);
    hipMemcpy(d_imgs, imgs.data(), G * 8, hipMemcpyHostToDevice);
    hipMemcpy(d_parent, parent.data(), G * 8, hipMemcpyHostToDevice);
    hipMemcpy(d_plist, plist.data(), G * 8, hipMemcpyHostToDevice);
    hipMemcpy(d_blist, blist.data(), G * 8, hipMemcpyHostToDevice);
    hipMemcpy(d_mask, mask.data(), G * 8, hipMemcpyHostToDevice);
    hipMemcpy(d_lcnt, lcnt.data(), G * 8, hipMemcpyHostToDevice);
    {
        g_rec_stride = (4096 / kRecRegions) * (size_t)kTileSlots;
        std::vector<CompStat *> st(G);
        std::vector<int *> ro(G), ci(G);
        for (int f = 0; f < G; ++f) {
            hipMalloc(&st[f], kRecRegions * g_rec_stride * sizeof(CompStat));
            hipMalloc(&ro[f], kRecRegions * g_rec_stride * 4);
            hipMalloc(&ci[f], P * 4);
        }
        hipMalloc(&g_st, G * 8);
        hipMalloc(&g_roots, G * 8);
        hipMalloc(&g_cid, G * 8);
        hipMalloc(&g_flags, G * 16);
        hipMemset(g_flags, 0, G * 16);
        hipMemcpy(g_st, st.data(), G * 8, hipMemcpyHostToDevice);
        hipMemcpy(g_roots, ro.data(), G * 8, hipMemcpyHostToDevice);
        hipMemcpy(g_cid, ci.data(), G * 8, hipMemcpyHostToDevice);
    }
    ab_pixel_xf xf;
    xf.on = 1;
    xf.lo = 150.0;
    xf.inv = 1.0 / 4000.0;
    const double thr = (200.0 + 5.0 * 6.0 - 150.0) / 4000.0;
#define RUN(V) run<V>(d_imgs, G, rows, cols, thr, xf, d_parent, d_mask, d_plist, ps, d_blist, bs, d_lcnt, lcnt_flat, G * lw, sink, reps)
    for (int round = 0; round < 2; ++round) {
        RUN(8);
        RUN(0);
        RUN(7);
        RUN(1);
        RUN(2);
        RUN(3);
        RUN(6);
        RUN(4);
        RUN(5);
    }
    {  // the two forms leave the same forest: parent of every labelled pixel, the mask, the list lengths
        std::vector<int> pa(P), pb(P);
        std::vector<unsigned int> ma(P / 32), mb(P / 32), ca(lw), cb(lw);
        hipMemset(parent[0], 0xff, P * 4);
        hipMemset(lcnt_flat, 0, G * lw * 4);
        hipLaunchKernelGGL(bench_kernel<7>, dim3(4096, 1), dim3(256), 0, 0, d_imgs, rows, cols, thr, xf, d_parent, d_mask, d_plist, ps, d_blist, bs, d_lcnt, sink);
        hipMemcpy(pa.data(), parent[0], P * 4, hipMemcpyDeviceToHost);
        hipMemcpy(ma.data(), mask[0], P / 8, hipMemcpyDeviceToHost);
        hipMemcpy(ca.data(), lcnt_flat, lw * 4, hipMemcpyDeviceToHost);
        hipMemset(parent[0], 0xff, P * 4);
        hipMemset(lcnt_flat, 0, G * lw * 4);
        hipLaunchKernelGGL(bench_kernel<0>, dim3(4096, 1), dim3(256), 0, 0, d_imgs, rows, cols, thr, xf, d_parent, d_mask, d_plist, ps, d_blist, bs, d_lcnt, sink);
        hipMemcpy(pb.data(), parent[0], P * 4, hipMemcpyDeviceToHost);
        hipMemcpy(mb.data(), mask[0], P / 8, hipMemcpyDeviceToHost);
        hipMemcpy(cb.data(), lcnt_flat, lw * 4, hipMemcpyDeviceToHost);
        size_t diff = 0, lab_px = 0;
        for (size_t i = 0; i < P; ++i) {
            diff += pa[i] != pb[i];
            lab_px += pa[i] >= 0;
        }
        printf("pixelwise vs runs: %zu labelled pixels, %zu parents differ, masks %s, counters %s\n", lab_px, diff, ma == mb ? "equal" : "DIFFER", ca == cb ? "equal" : "DIFFER");
    }
    {
        unsigned int fl[4];
        hipMemcpy(fl, g_flags, 16, hipMemcpyDeviceToHost);
        printf("records form: overflow flag of frame 0 = %u\n", fl[0]);
    }
    {  // phase marks of the records form, frame 0: per tile the slowest wave's time at each mark, and the tile's lifetime
        long long *marks;
        hipMalloc(&marks, 4096 * 4 * 8 * 8);
        hipMemset(marks, 0, 4096 * 4 * 8 * 8);
        hipMemcpyToSymbol(HIP_SYMBOL(g_label_marks), &marks, sizeof marks);
        hipMemset(lcnt_flat, 0, G * lw * 4);
        hipLaunchKernelGGL(bench_kernel<8>, dim3(4096, G), dim3(256), 0, 0, d_imgs, rows, cols, thr, xf, d_parent, d_mask, d_plist, ps, d_blist, bs, d_lcnt, sink, g_st, g_roots, g_cid,
                           g_rec_stride, g_flags);
        hipDeviceSynchronize();
        std::vector<long long> m(4096 * 4 * 8);
        hipMemcpy(m.data(), marks, m.size() * 8, hipMemcpyDeviceToHost);
        long long *none = nullptr;
        hipMemcpyToSymbol(HIP_SYMBOL(g_label_marks), &none, sizeof none);
        double ph[7] = {0}, life = 0;
        const char *names[7] = {"", "loads+threshold+mask", "run starts", "unions", "roots+slots", "contributions+parents", "border list+records"};
        for (int t = 0; t < 4096; ++t) {
            long long t0 = m[(t * 4) * 8], last[7];
            for (int w = 1; w < 4; ++w) t0 = std::min(t0, m[(t * 4 + w) * 8]);
            for (int i = 1; i < 7; ++i) {
                last[i] = 0;
                for (int w = 0; w < 4; ++w) last[i] = std::max(last[i], m[(t * 4 + w) * 8 + i]);
            }
            long long prev = t0;
            for (int i = 1; i < 7; ++i) {
                ph[i] += (double)(last[i] - prev);
                prev = last[i];
            }
            life += (double)(last[6] - t0);
        }
        printf("records form, per tile (s_memtime ticks of 10 ns, slowest wave at each mark): lifetime %.0f;", life / 4096);
        for (int i = 1; i < 7; ++i) printf("  %s %.0f", names[i], ph[i] / 4096);
        printf("\n");
    }
    std::vector<unsigned int> c(G * lw);
    hipMemcpy(c.data(), lcnt_flat, G * lw * 4, hipMemcpyDeviceToHost);
    unsigned int nl = 0;
    for (int r = 0; r < kRegions; ++r) nl += c[r * kRegionPitch];
    printf("labelled pixels of frame 0 after the last variant: %u\n", nl);
    return 0;
}
