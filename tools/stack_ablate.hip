// Ablation micro-bench of the stacking kernel (developer tool, not part of the library):
// times the kernel cut short after each stage on one 64 x 4096 x 4096 synthetic stack.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off tools/stack_ablate.hip -o build/stack_ablate
#include "../astroburst_amd/csrc/ab_context.hip"
#include "../astroburst_amd/csrc/stack_sigma_clip.hip"
// (the > 64-frame path lives in stack_wide.hip, whose file-scope constants collide with stack_sigma_clip.hip's when both are
// pulled into one translation unit; this tool never stacks more than 64 frames)
int ab_stack_wide_device(ab_ctx *ctx, const float *const *, const int64_t *, size_t, int64_t, int64_t, const ab_stack_config *, float *, double *,
                         uint32_t *, bool) {
    return ab_set_error(ctx, AB_ERR_INVALID, "stack_ablate: the wide path is not linked into this tool");
}

__global__ void fill_kernel(float *p, int64_t n, uint32_t seed, float cr_rate) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    // counter-based hash -> Box-Muller normal (true Gaussian tails, like the bench's shot + read noise), sky 1200 +- 12.7,
    // rare x30 outliers
    uint64_t z = (uint64_t)i * 0x9E3779B97F4A7C15ull + seed * 0xD1B54A32D192ED03ull;
    float u[2];
    for (int k = 0; k < 2; ++k) {
        z ^= z >> 30; z *= 0xBF58476D1CE4E5B9ull; z ^= z >> 27; z *= 0x94D049BB133111EBull; z ^= z >> 31;
        u[k] = ((float)(z >> 40) + 0.5f) * (1.0f / 16777216.0f);
    }
    float v = 1200.0f + 12.7f * sqrtf(-2.0f * logf(u[0])) * cosf(6.2831853f * u[1]);
    z ^= z >> 30; z *= 0xBF58476D1CE4E5B9ull;
    if ((float)(z >> 40) * (1.0f / 16777216.0f) < cr_rate) v *= 30.0f;
    p[i] = v;
}

template <int STAGE, bool EXACT>
float time_stage(ab_ctx *ctx, const StackArgs &args, int reps) {
    const int64_t total = args.rows * args.cols;
    const dim3 grid((unsigned)((total + 255) / 256)), block(256);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((stack_sigma_clip_kernel<64, false, EXACT, STAGE, true>), grid, block, 0, ctx->stream, args);
    hipEventRecord(e0, ctx->stream);
    for (int r = 0; r < reps; ++r)
        hipLaunchKernelGGL((stack_sigma_clip_kernel<64, false, EXACT, STAGE, true>), grid, block, 0, ctx->stream, args);
    hipEventRecord(e1, ctx->stream);
    hipEventSynchronize(e1);
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    return ms / reps;
}

int main(int argc, char **argv) {
    ab_ctx *ctx = nullptr;
    if (ab_ctx_create(0, &ctx) != AB_OK) { fprintf(stderr, "no gfx950 device\n"); return 1; }
    const int64_t rows = 4096, cols = 4096, P = rows * cols;
    // usage: stack_ablate [cosmic_ray_rate]   |   stack_ablate --dir DIR   (DIR/frame_%02d.f32: 64 raw 4096^2 f32 planes,
    // e.g. the bench's registered frames written by tools/time_stack_bench_data.py --dump DIR)
    const bool from_dir = argc > 2 && std::string(argv[1]) == "--dir";
    const float cr = (!from_dir && argc > 1) ? atof(argv[1]) : 1e-4f;
    StackArgs args; memset(&args, 0, sizeof args);
    std::vector<float> host(from_dir ? P : 0);
    for (int f = 0; f < 64; ++f) {
        float *p; hipMalloc(&p, P * 4);
        if (from_dir) {
            char path[512];
            snprintf(path, sizeof path, "%s/frame_%02d.f32", argv[2], f);
            FILE *fp = fopen(path, "rb");
            if (!fp || fread(host.data(), 4, P, fp) != (size_t)P) { fprintf(stderr, "cannot read %s\n", path); return 1; }
            fclose(fp);
            hipMemcpy(p, host.data(), P * 4, hipMemcpyHostToDevice);
        } else {
            fill_kernel<<<(unsigned)((P + 255) / 256), 256, 0, ctx->stream>>>(p, P, 1000 + f, cr);
        }
        args.p[f] = p; args.ld[f] = cols;
    }
    float *out; hipMalloc(&out, P * 4);
    args.n = 64; args.n_real = 64; args.contiguous = 1; args.rows = rows; args.cols = cols;
    args.sigma_low = 3.f; args.sigma_high = 3.f; args.max_iter = 5; args.out = out; args.rejected = ctx->counters;
    hipStreamSynchronize(ctx->stream);
    const double gb = (4.0 * 64 * P + 4.0 * P) / 1e9;
    float t1 = time_stage<1, false>(ctx, args, 5);
    float t2 = time_stage<2, false>(ctx, args, 5);
    float t3 = time_stage<3, false>(ctx, args, 5);
    float t4 = time_stage<4, false>(ctx, args, 5);
    float t5 = time_stage<5, false>(ctx, args, 5);
    float t6 = time_stage<6, false>(ctx, args, 5);
    float t7 = time_stage<7, false>(ctx, args, 5);
    args.max_iter = 2;
    float t92 = time_stage<99, false>(ctx, args, 5);
    args.max_iter = 3;
    float t93 = time_stage<99, false>(ctx, args, 5);
    args.max_iter = 5;
    float t9 = time_stage<99, false>(ctx, args, 5);
    float tx = time_stage<99, true>(ctx, args, 5);
    // compute-only estimate: every frame pointer aliases frame 0 (64 MiB, cache resident after the
    // first touch), so HBM traffic drops ~64x and what remains is issue/latency time
    StackArgs alias = args;
    for (int f = 1; f < 64; ++f) alias.p[f] = alias.p[0];
    float ta1 = time_stage<1, false>(ctx, alias, 5);
    float ta2 = time_stage<2, false>(ctx, alias, 5);
    float ta9 = time_stage<99, false>(ctx, alias, 5);
    // the product path: fast pass + general pass through ab_stack_device (what bench.py's roofline times)
    float tprod = 0.f;
    {
        std::vector<const float *> dp(64);
        std::vector<int64_t> ld(64, cols);
        for (int f = 0; f < 64; ++f) dp[f] = args.p[f];
        ab_stack_config cfg = {3.f, 3.f, 5, 0};
        ab_stack_device(ctx, dp.data(), ld.data(), 64, rows, cols, &cfg, out, nullptr, nullptr, nullptr, false);
        hipEvent_t e0, e1;
        hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0, ctx->stream);
        for (int r = 0; r < 10; ++r) ab_stack_device(ctx, dp.data(), ld.data(), 64, rows, cols, &cfg, out, nullptr, nullptr, nullptr, false);
        hipEventRecord(e1, ctx->stream);
        hipEventSynchronize(e1);
        hipEventElapsedTime(&tprod, e0, e1);
        tprod /= 10;
    }
    printf("PRODUCT two-pass stack    %8.3f ms  %7.1f GB/s  (%.3f of 8 TB/s)\n", tprod, gb / tprod * 1e3, gb / tprod * 1e3 / 8000.0);
    printf("cosmic-ray rate %g\n", cr);
    printf("aliased frames (no HBM): loads %.3f ms, +sort %.3f ms, full %.3f ms\n", ta1, ta2, ta9);
    printf("stage 1 loads only        %8.3f ms  %7.1f GB/s\n", t1, gb / t1 * 1e3);
    printf("stage 2 + pads + sort     %8.3f ms  %7.1f GB/s\n", t2, gb / t2 * 1e3);
    printf("stage 3 + median/MAD      %8.3f ms  %7.1f GB/s\n", t3, gb / t3 * 1e3);
    printf("stage 4 + clip 0          %8.3f ms  %7.1f GB/s\n", t4, gb / t4 * 1e3);
    printf("stage 5 + S1/Q1 pass      %8.3f ms  %7.1f GB/s\n", t5, gb / t5 * 1e3);
    printf("stage 6 + iter-1 mean/sigma %6.3f ms  %7.1f GB/s\n", t6, gb / t6 * 1e3);
    printf("stage 7 + iter-1 end walk %8.3f ms  %7.1f GB/s\n", t7, gb / t7 * 1e3);
    printf("full, max_iter=2          %8.3f ms  %7.1f GB/s\n", t92, gb / t92 * 1e3);
    printf("full, max_iter=3          %8.3f ms  %7.1f GB/s\n", t93, gb / t93 * 1e3);
    printf("stage 99 full (fast)      %8.3f ms  %7.1f GB/s\n", t9, gb / t9 * 1e3);
    printf("stage 99 full (exact)     %8.3f ms  %7.1f GB/s\n", tx, gb / tx * 1e3);
    return 0;
}
