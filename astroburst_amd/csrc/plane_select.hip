// Exact order statistics over a whole plane (gfx950): 11/11/10-bit radix select on f32 bit patterns.
//
// Serves every place the reference collects the valid pixels of a full image into a Vec and calls
// select_nth_unstable on it: background.rs:135-146 (global median / MAD), :350-361 (model median),
// masked_stretch.rs:213-229 (median of the unmasked pixels, once per stretch iteration).  Candidates
// are finite and > min_valid >= 0, so their bit patterns (and those of absolute deviations) order
// like the values.  Pass 0 histograms the top 11 bits and yields the candidate count; each requested
// rank then costs two more streaming passes.  All passes are HBM-bound reads of the plane (+ mask).
#include "ab_common.hpp"

#include <algorithm>

namespace {

constexpr int kBlock = 256;

struct SelArgs {
    const float *data, *mask;
    int64_t n;
    float min_valid;
    int use_dev;
    float center;
    uint32_t prefix_mask, prefix_val;
    int shift, nbits;
    unsigned int *hist;
};

__global__ __launch_bounds__(kBlock) void plane_select_hist_kernel(const SelArgs a) {
    __shared__ unsigned int lds[2048];
    const uint32_t nb = 1u << a.nbits;
    for (uint32_t i = threadIdx.x; i < nb; i += kBlock) lds[i] = 0;
    __syncthreads();
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < a.n; i += stride) {
        const float v = a.data[i];
        bool ok = __builtin_isfinite(v) && v > a.min_valid;
        if (a.mask) ok = ok && a.mask[i] < 0.5f;
        if (ok) {
            const float k = a.use_dev ? fabsf(v - a.center) : v;
            const uint32_t key = __float_as_uint(k);
            if ((key & a.prefix_mask) == a.prefix_val) atomicAdd(&lds[(key >> a.shift) & (nb - 1)], 1u);
        }
    }
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < nb; i += kBlock)
        if (lds[i]) atomicAdd(&a.hist[i], lds[i]);
}

int run_pass(ab_ctx *ctx, const ab_plane_sel &s, uint32_t mask, uint32_t val, int shift, int nbits, unsigned int *host) {
    const uint32_t nb = 1u << nbits;
    AB_HIP(ctx, hipMemsetAsync(ctx->sel_hist, 0, nb * sizeof(unsigned int), ctx->stream));
    const int grid = (int)std::max<int64_t>(
        1, std::min<int64_t>((s.n + kBlock - 1) / kBlock, (int64_t)(ctx->cu_count > 0 ? ctx->cu_count : 256) * 8));
    SelArgs a{s.data, s.mask, s.n, s.min_valid, s.use_dev, s.center, mask, val, shift, nbits, ctx->sel_hist};
    hipLaunchKernelGGL(plane_select_hist_kernel, dim3(grid), dim3(kBlock), 0, ctx->stream, a);
    AB_HIP(ctx, hipGetLastError());
    AB_HIP(ctx, hipMemcpyAsync(host, ctx->sel_hist, nb * sizeof(unsigned int), hipMemcpyDeviceToHost, ctx->stream));
    AB_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return AB_OK;
}

// bin containing 0-based rank; rank becomes the rank within the bin
uint32_t locate(const unsigned int *h, uint32_t nb, uint64_t *rank) {
    uint64_t cum = 0;
    for (uint32_t i = 0; i < nb; ++i) {
        if (cum + h[i] > *rank) {
            *rank -= cum;
            return i;
        }
        cum += h[i];
    }
    *rank = 0;
    return nb - 1;
}

}  // namespace

int ab_plane_select_ranks(ab_ctx *ctx, const ab_plane_sel &s, int max_ranks, const std::function<int(uint64_t, uint64_t *)> &ranks_of,
                          uint64_t *count_out, float *vals) {
    *count_out = 0;
    AB_HIP(ctx, hipSetDevice(ctx->device));
    if (!ctx->sel_hist) AB_HIP(ctx, hipMalloc((void **)&ctx->sel_hist, 2048 * sizeof(unsigned int)));
    void *pin = nullptr;
    AB_TRY(ab_pinned(ctx, 3 * 2048 * sizeof(unsigned int), &pin));
    unsigned int *h0 = (unsigned int *)pin, *h1 = h0 + 2048, *h2 = h1 + 2048;
    if (s.n <= 0) return AB_OK;
    AB_TRY(run_pass(ctx, s, 0, 0, 21, 11, h0));
    uint64_t count = 0;
    for (int i = 0; i < 2048; ++i) count += h0[i];
    *count_out = count;
    if (count == 0) return AB_OK;
    std::vector<uint64_t> ranks((size_t)max_ranks);
    const int n_ranks = std::min(max_ranks, ranks_of(count, ranks.data()));
    // The ranks descend TOGETHER: a level is histogrammed once per distinct prefix, and the two middle ranks of an even count (the
    // usual request) share their prefix all the way down except when they straddle a bin edge -- 3 passes instead of 5.
    struct Item {
        int r;
        uint64_t rank;  // within the current prefix
    };
    const int shifts[3] = {21, 10, 0}, bits[3] = {11, 11, 10};
    std::function<int(int, uint32_t, uint32_t, const unsigned int *, std::vector<Item>)> descend =
        [&](int level, uint32_t mask, uint32_t val, const unsigned int *h, std::vector<Item> items) -> int {
        // h = the histogram of `level` under (mask, val); split the items by the bin their rank falls into
        const uint32_t nb = 1u << bits[level];
        std::vector<std::pair<uint32_t, std::vector<Item>>> groups;
        for (Item it : items) {
            const uint32_t b = locate(h, nb, &it.rank);
            if (groups.empty() || groups.back().first != b) groups.push_back({b, {}});
            groups.back().second.push_back(it);
        }
        for (auto &g : groups) {
            const uint32_t v = val | (g.first << shifts[level]);
            if (level == 2) {
                for (const Item &it : g.second) memcpy(&vals[it.r], &v, sizeof(float));
                continue;
            }
            const uint32_t m = mask | (((1u << bits[level]) - 1u) << shifts[level]);
            unsigned int *hn = level == 0 ? h1 : h2;  // (a level's histogram is consumed before the next group overwrites it)
            AB_TRY(run_pass(ctx, s, m, v, shifts[level + 1], bits[level + 1], hn));
            AB_TRY(descend(level + 1, m, v, hn, g.second));
        }
        return AB_OK;
    };
    std::vector<Item> items;
    for (int r = 0; r < n_ranks; ++r) items.push_back({r, std::min(ranks[r], count - 1)});
    std::stable_sort(items.begin(), items.end(), [](const Item &a, const Item &b) { return a.rank < b.rank; });  // equal bins become neighbours
    return descend(0, 0u, 0u, h0, items);
}

int ab_plane_order_stats(ab_ctx *ctx, const ab_plane_sel &s, int want_lower, uint64_t *count_out, float *mid_out, float *lower_out) {
    *mid_out = 0.0f;
    if (lower_out) *lower_out = 0.0f;
    float v[2] = {0.0f, 0.0f};
    int got = 0;
    AB_TRY(ab_plane_select_ranks(
        ctx, s, 2,
        [&](uint64_t count, uint64_t *ranks) {
            ranks[0] = count / 2;
            ranks[1] = count / 2 - (count > 1 ? 1 : 0);
            return got = (want_lower && count % 2 == 0) ? 2 : 1;
        },
        count_out, v));
    if (*count_out == 0) return AB_OK;
    *mid_out = v[0];
    if (lower_out) *lower_out = got == 2 ? v[1] : v[0];
    return AB_OK;
}

int ab_plane_median_f32(ab_ctx *ctx, const ab_plane_sel &s, float *out, uint64_t *count_out) {
    uint64_t cnt;
    float mid, lower;
    AB_TRY(ab_plane_order_stats(ctx, s, 1, &cnt, &mid, &lower));
    if (count_out) *count_out = cnt;
    *out = cnt == 0 ? 0.0f : (cnt % 2 == 0 ? (lower + mid) / 2.0f : mid);
    return AB_OK;
}
