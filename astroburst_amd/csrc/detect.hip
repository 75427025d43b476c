// Star detection / segmentation on gfx950.
//
// Replaces core/analysis/star_detection.rs: estimate_background (:32-84, with
// math/sigma_clip.rs:4-34 and math/median.rs:27-73 per tile), detect_stars (:86-258),
// core/analysis/confidence.rs:3-8, and core/alignment/affine.rs:24-53 (normalize_for_detection).
//
// Mapping (integer / index work, HBM- and latency-bound; no GEMM shapes):
//   * tile background: one 1024-thread workgroup per tile (<= 256 x 256 px).  The reference's
//     median / MAD selects become 11/11/10-bit radix selects over the tile's valid pixels, whose
//     f32 bit patterns are monotone (valid means > 1e-7); histograms live in LDS, the tile itself is
//     re-read from L2.  Even-count medians average the two middle order statistics exactly as
//     exact_median_mut / median_f32_mut do, so tile medians and sigmas are bit-identical.
//   * labelling: the reference's sequential raster scan + 8-connected BFS is replaced by a
//     lock-free union-find over the above-threshold pixels (4 forward neighbours per pixel,
//     atomicMin hooking, then path flattening).  The root of a component is its minimum raster
//     index, which makes the label canonical; a component is reported only if it owns an
//     INTERIOR pixel, because the reference seeds from 1..rows-1 x 1..cols-1 only.
//   * the above-threshold pixels (a fraction of a percent of the frame) are compacted to
//     (index, root, value) triples; the per-component moments are f64 sums over at most 5000 pixels
//     each and are finished on the host in raster order (the reference sums in BFS order: the two
//     agree to ~1e-15 relative; counts, npix and the component set are exact).
#include "ab_common.hpp"
#include "block_select.hpp"

#include <algorithm>
#include <cfloat>
#include <cmath>
#include <unordered_map>

namespace {

constexpr double kMadToSigma = 1.4826;

// ---- per-tile sigma-clipped statistics ---------------------------------------------------------
struct TileOut {
    double median, sigma;
    int valid;  // 1 if the tile had >= 8 valid pixels
    int pad;
};

// estimate_background's per-tile body (star_detection.rs:47-68) = sigma_clipped_stats(vals, 3.0, 2)
// (math/sigma_clip.rs:4-34); order statistics by workgroup radix select (block_select.hpp)
__global__ __launch_bounds__(absel::kBlock) void tile_background_kernel(const float *__restrict__ img, int rows, int cols,
                                                                        int64_t ld, int step, int ntx,
                                                                        TileOut *__restrict__ out) {
    __shared__ unsigned int hist[2048];
    const int ty = blockIdx.x / ntx, tx = blockIdx.x % ntx;
    absel::Window t;
    t.img = img;
    t.ld = ld;
    t.y0 = ty * step;
    t.x0 = tx * step;
    t.y1 = min(t.y0 + step, rows);
    t.x1 = min(t.x0 + step, cols);
    t.min_valid = 1e-7f;  // star_detection.rs:56
    t.lo = -__builtin_inff();
    t.hi = __builtin_inff();

    unsigned int n = absel::count(t, hist);
    TileOut res = {0.0, 1.0, 0, 0};
    if (n >= 8) {
        res.valid = 1;
        double median = 0.0, sigma = 1.0;
        for (int it = 0; it < 3; ++it) {  // 2 clipping iterations + the final statistics (sigma_clip.rs:7-33)
            if (it < 2 && n < 3) continue;
            if (n == 0) {  // sigma_clip.rs:26-28
                median = 0.0;
                sigma = 1.0;
                break;
            }
            median = absel::exact_median(t, n, hist);                                  // median.rs:27-44
            const float mad_f32 = absel::median_f32(t, 1, median, 0.0f, n, hist);      // sigma_clip.rs:14-16
            const double sig = fmax((double)mad_f32 * kMadToSigma, 1e-30);
            if (it == 2) {
                sigma = sig;
                break;
            }
            // retain v in [lo, hi] (sigma_clip.rs:19-23); kappa = 3.0f32 as f64
            const float lo = (float)(median - 3.0 * sig), hi = (float)(median + 3.0 * sig);
            t.lo = fmaxf(t.lo, lo);
            t.hi = fminf(t.hi, hi);
            if (!(lo <= hi)) {  // NaN bounds or empty interval: nothing is retained
                t.lo = __builtin_inff();
                t.hi = -__builtin_inff();
            }
            n = absel::count(t, hist);
        }
        res.median = median;
        res.sigma = sigma;
    }
    if (threadIdx.x == 0) out[blockIdx.x] = res;
}

// ---- threshold + union-find labelling -------------------------------------------------------------
__device__ __forceinline__ bool above(float v, double threshold) { return __builtin_isfinite(v) && (double)v > threshold; }

__global__ __launch_bounds__(256) void label_init_kernel(const float *__restrict__ img, int rows, int cols, int64_t ld,
                                                         double threshold, int *__restrict__ parent) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= rows * cols) return;
    const int r = i / cols, c = i - r * cols;
    parent[i] = above(img[r * ld + c], threshold) ? i : -1;
}

__device__ __forceinline__ int uf_find(int *parent, int x) {
    while (true) {
        const int p = __hip_atomic_load(&parent[x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (p == x) return x;
        x = p;
    }
}

__device__ __forceinline__ void uf_union(int *parent, int a, int b) {
    while (true) {
        a = uf_find(parent, a);
        b = uf_find(parent, b);
        if (a == b) return;
        if (a < b) {
            const int t = a;
            a = b;
            b = t;
        }
        const int old = atomicMin(&parent[a], b);  // hook the larger root under the smaller
        if (old == a) return;
        a = old;
    }
}

__global__ __launch_bounds__(256) void label_merge_kernel(int rows, int cols, int *parent) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= rows * cols) return;
    if (parent[i] < 0) return;
    const int r = i / cols, c = i - r * cols;
    // forward half of the 8-neighbourhood (star_detection.rs:120): E, SW, S, SE
    if (c + 1 < cols && parent[i + 1] >= 0) uf_union(parent, i, i + 1);
    if (r + 1 < rows) {
        const int d = i + cols;
        if (c > 0 && parent[d - 1] >= 0) uf_union(parent, i, d - 1);
        if (parent[d] >= 0) uf_union(parent, i, d);
        if (c + 1 < cols && parent[d + 1] >= 0) uf_union(parent, i, d + 1);
    }
}

struct Triple {
    int idx, root;
    float value;
};

// flatten + compact the labelled pixels; one atomic per wave on the list tail
__global__ __launch_bounds__(256) void label_compact_kernel(const float *__restrict__ img, int rows, int cols, int64_t ld,
                                                            int *parent, Triple *__restrict__ list, unsigned int *count,
                                                            unsigned int cap) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    bool is = false;
    int root = -1;
    if (i < rows * cols && parent[i] >= 0) {
        root = uf_find(parent, i);
        is = true;
    }
    const unsigned long long m = __ballot(is);
    if (m == 0) return;
    const int lane = threadIdx.x & 63;
    unsigned int base = 0;
    if (lane == (int)__builtin_ctzll(m)) base = atomicAdd(count, (unsigned int)__builtin_popcountll(m));
    base = __shfl(base, (int)__builtin_ctzll(m), 64);
    if (is) {
        const unsigned int pos = base + (unsigned int)__builtin_popcountll(m & ((1ull << lane) - 1ull));
        if (pos < cap) {
            const int r = i / cols, c = i - r * cols;
            list[pos].idx = i;
            list[pos].root = root;
            list[pos].value = img[r * ld + c];
        }
    }
}

// ---- normalize_for_detection (affine.rs:24-53) ------------------------------------------------------
__global__ __launch_bounds__(256) void subsample_kernel(const float *__restrict__ img, int64_t len, int64_t step,
                                                        float *__restrict__ out, int64_t nout) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < nout) out[i] = img[i * step];
}

__global__ __launch_bounds__(256) void normalize_kernel(const float *__restrict__ img, int64_t len, double lo, double inv_range,
                                                        float *__restrict__ out) {
    const int64_t stride = (int64_t)gridDim.x * 256;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < len; i += stride) {
        double t = ((double)img[i] - lo) * inv_range;
        t = t < 0.0 ? 0.0 : (t > 1.0 ? 1.0 : t);  // f64::clamp: NaN stays NaN
        out[i] = (float)t;
    }
}

int f64_cmp(double a, double b) {  // math/median.rs:15-25
    if (a < b) return -1;
    if (a > b) return 1;
    if (a == b) return 0;
    const bool an = std::isnan(a), bn = std::isnan(b);
    return an && bn ? 0 : (an ? 1 : -1);
}

}  // namespace

// estimate_background (star_detection.rs:32-84) on a device plane
int ab_estimate_background_device(ab_ctx *ctx, const float *img, int64_t rows, int64_t cols, int64_t ld, int64_t tile_size,
                                  double *out_median, double *out_sigma) {
    AB_HIP(ctx, hipSetDevice(ctx->device));
    const int step = (int)std::max<int64_t>(tile_size, 16);
    AB_CHECK(ctx, step <= 256, "background tiles larger than 256 px are not supported (tile_size %lld)", (long long)tile_size);
    const int nty = (int)((rows + step - 1) / step), ntx = (int)((cols + step - 1) / step);
    const int ntiles = nty * ntx;
    void *d = nullptr;
    AB_TRY(ab_scratch(ctx, (size_t)ntiles * sizeof(TileOut), &d));
    hipLaunchKernelGGL(tile_background_kernel, dim3(ntiles), dim3(absel::kBlock), 0, ctx->stream, img, (int)rows, (int)cols, ld, step,
                       ntx, (TileOut *)d);
    AB_HIP(ctx, hipGetLastError());
    std::vector<TileOut> h(ntiles);
    AB_HIP(ctx, hipMemcpyAsync(h.data(), d, (size_t)ntiles * sizeof(TileOut), hipMemcpyDeviceToHost, ctx->stream));
    AB_HIP(ctx, hipStreamSynchronize(ctx->stream));
    std::vector<double> med, sig;
    for (const auto &t : h)
        if (t.valid) {
            med.push_back(t.median);
            sig.push_back(t.sigma);
        }
    if (med.empty()) {  // :70-72
        *out_median = 0.0;
        *out_sigma = 1.0;
        return AB_OK;
    }
    auto lt = [](double a, double b) { return f64_cmp(a, b) < 0; };
    std::sort(med.begin(), med.end(), lt);
    std::sort(sig.begin(), sig.end(), lt);
    *out_median = med[med.size() / 2];
    *out_sigma = std::fmax(sig[sig.size() / 2], 1e-10);
    return AB_OK;
}

// detect_stars (star_detection.rs:86-258) on a device plane; stars sorted by flux, deduplicated
int ab_detect_stars_device(ab_ctx *ctx, const float *img, int64_t rows, int64_t cols, int64_t ld, double sigma_threshold,
                           std::vector<ab_detected_star> *stars, double *bg_median_out, double *bg_sigma_out) {
    stars->clear();
    *bg_median_out = 0.0;
    *bg_sigma_out = 1.0;
    if (rows < 3 || cols < 3) return AB_OK;  // :89-98
    AB_CHECK(ctx, rows * cols < (int64_t(1) << 31), "detect_stars: image too large for 32-bit labels");
    AB_HIP(ctx, hipSetDevice(ctx->device));
    const int64_t m = std::min(rows, cols);
    const int64_t tile_size = std::min<int64_t>(std::max<int64_t>(m / 8, 32), 256);  // :100
    double bg_median, bg_sigma;
    AB_TRY(ab_estimate_background_device(ctx, img, rows, cols, ld, tile_size, &bg_median, &bg_sigma));
    *bg_median_out = bg_median;
    *bg_sigma_out = bg_sigma;
    const double threshold = bg_median + sigma_threshold * bg_sigma;  // :103

    const int64_t P = rows * cols;
    int *parent = nullptr;
    unsigned int *count = nullptr;
    Triple *list = nullptr;
    const unsigned int cap = (unsigned int)std::min<int64_t>(P, int64_t(1) << 28);
    AB_HIP(ctx, hipMalloc((void **)&parent, (size_t)P * sizeof(int)));
    hipError_t e = hipMalloc((void **)&count, sizeof(unsigned int));
    if (e == hipSuccess) e = hipMalloc((void **)&list, (size_t)cap * sizeof(Triple));
    std::vector<Triple> h;
    int rc = AB_OK;
    if (e == hipSuccess) e = hipMemsetAsync(count, 0, sizeof(unsigned int), ctx->stream);
    if (e == hipSuccess) {
        const int g = (int)((P + 255) / 256);
        hipLaunchKernelGGL(label_init_kernel, dim3(g), dim3(256), 0, ctx->stream, img, (int)rows, (int)cols, ld, threshold, parent);
        hipLaunchKernelGGL(label_merge_kernel, dim3(g), dim3(256), 0, ctx->stream, (int)rows, (int)cols, parent);
        hipLaunchKernelGGL(label_compact_kernel, dim3(g), dim3(256), 0, ctx->stream, img, (int)rows, (int)cols, ld, parent, list, count,
                           cap);
        e = hipGetLastError();
    }
    unsigned int n = 0;
    if (e == hipSuccess) e = hipMemcpyAsync(&n, count, sizeof n, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    if (e == hipSuccess && n > cap) rc = ab_set_error(ctx, AB_ERR_NOMEM, "detect_stars: %u labelled pixels exceed the list capacity", n);
    if (e == hipSuccess && rc == AB_OK && n > 0) {
        h.resize(n);
        e = hipMemcpyAsync(h.data(), list, (size_t)n * sizeof(Triple), hipMemcpyDeviceToHost, ctx->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    }
    if (parent) (void)hipFree(parent);
    if (count) (void)hipFree(count);
    if (list) (void)hipFree(list);
    if (e != hipSuccess) return ab_set_error(ctx, AB_ERR_HIP, "detect_stars: %s", hipGetErrorString(e));
    if (rc != AB_OK) return rc;

    // ---- host: group by root in raster order, keep components that own an interior pixel ----
    std::sort(h.begin(), h.end(), [](const Triple &a, const Triple &b) { return a.idx < b.idx; });
    struct Comp {
        int first_interior = -1;
        std::vector<int> px;  // positions in h (raster order)
    };
    std::unordered_map<int, int> root_to_comp;
    std::vector<Comp> comps;
    for (size_t k = 0; k < h.size(); ++k) {
        auto it = root_to_comp.find(h[k].root);
        int ci;
        if (it == root_to_comp.end()) {
            ci = (int)comps.size();
            root_to_comp.emplace(h[k].root, ci);
            comps.emplace_back();
        } else {
            ci = it->second;
        }
        Comp &c = comps[ci];
        c.px.push_back((int)k);
        const int r = h[k].idx / (int)cols, cc = h[k].idx % (int)cols;
        if (c.first_interior < 0 && r >= 1 && r < rows - 1 && cc >= 1 && cc < cols - 1) c.first_interior = h[k].idx;
    }
    std::vector<int> order;
    for (int i = 0; i < (int)comps.size(); ++i)
        if (comps[i].first_interior >= 0) order.push_back(i);
    std::sort(order.begin(), order.end(), [&](int a, int b) { return comps[a].first_interior < comps[b].first_interior; });  // discovery order

    std::vector<ab_detected_star> found;
    for (int ci : order) {
        const Comp &c = comps[ci];
        const size_t npix = c.px.size();
        if (npix < 3 || npix > 5000) continue;  // :142-145
        double sum_flux = 0.0, sum_x = 0.0, sum_y = 0.0, peak = 0.0;
        for (int k : c.px) {
            const double v = std::fmax((double)h[k].value - bg_median, 0.0);
            const int pr = h[k].idx / (int)cols, pc = h[k].idx % (int)cols;
            sum_flux += v;
            sum_x += (double)pc * v;
            sum_y += (double)pr * v;
            peak = std::fmax(peak, v);
        }
        if (sum_flux <= 0.0) continue;
        const double cx = sum_x / sum_flux, cy = sum_y / sum_flux;
        double sum_r2 = 0.0, sum_xx = 0.0, sum_yy = 0.0, sum_xy = 0.0;
        for (int k : c.px) {
            const double v = std::fmax((double)h[k].value - bg_median, 0.0);
            const int pr = h[k].idx / (int)cols, pc = h[k].idx % (int)cols;
            const double dx = (double)pc - cx, dy = (double)pr - cy;
            sum_r2 += (dx * dx + dy * dy) * v;
            sum_xx += dx * dx * v;
            sum_yy += dy * dy * v;
            sum_xy += dx * dy * v;
        }
        const double sigma_star = std::sqrt(sum_r2 / (2.0 * sum_flux));
        const double fwhm = sigma_star * 2.3548200450309493;
        if (fwhm < 0.5 || fwhm > 30.0) continue;
        const double ixx = sum_xx / sum_flux, iyy = sum_yy / sum_flux, ixy = sum_xy / sum_flux;
        const double trace = ixx + iyy;
        const double det = std::fmax(ixx * iyy - ixy * ixy, 0.0);
        const double disc = std::sqrt(std::fmax((trace * trace / 4.0) - det, 0.0));
        const double l1 = trace / 2.0 + disc, l2 = std::fmax(trace / 2.0 - disc, 0.0);
        double ecc = 0.0;
        if (l1 > 1e-15) {
            ecc = std::sqrt(1.0 - l2 / l1);
            ecc = ecc < 0.0 ? 0.0 : (ecc > 1.0 ? 1.0 : ecc);
        }
        ab_detected_star s;
        s.x = cx;
        s.y = cy;
        s.flux = sum_flux;
        s.fwhm = fwhm;
        s.eccentricity = ecc;
        s.peak = peak;
        s.npix = npix;
        s.snr = bg_sigma <= DBL_EPSILON ? 0.0 : peak / bg_sigma;  // confidence.rs:3-8
        found.push_back(s);
    }
    std::stable_sort(found.begin(), found.end(), [](const ab_detected_star &a, const ab_detected_star &b) { return b.flux < a.flux; });  // :215
    // dedup within 3 px, comparing only against kept stars in the 3 x 3 neighbourhood of 3 px grid cells (:217-248)
    std::unordered_map<uint64_t, std::vector<int>> grid;
    auto key = [](uint64_t gy, uint64_t gx) { return (gy << 32) | gx; };
    for (size_t i = 0; i < found.size(); ++i) {
        const uint64_t gx = (uint64_t)(found[i].x / 3.0), gy = (uint64_t)(found[i].y / 3.0);
        bool too_close = false;
        for (uint64_t ny = gy ? gy - 1 : 0; ny <= gy + 1 && !too_close; ++ny)
            for (uint64_t nx = gx ? gx - 1 : 0; nx <= gx + 1 && !too_close; ++nx) {
                auto it = grid.find(key(ny, nx));
                if (it == grid.end()) continue;
                for (int j : it->second) {
                    const double dx = found[i].x - found[j].x, dy = found[i].y - found[j].y;
                    if (dx * dx + dy * dy < 9.0) {
                        too_close = true;
                        break;
                    }
                }
            }
        if (!too_close) {
            grid[key(gy, gx)].push_back((int)i);
            stars->push_back(found[i]);
        }
    }
    return AB_OK;
}

// normalize_for_detection (affine.rs:24-53) into a contiguous device buffer; *cloned = 1 when the
// reference returns image.clone() (too few samples / flat range): out then equals the input
int ab_normalize_for_detection_device(ab_ctx *ctx, const float *img, int64_t len, float *out, int *cloned) {
    AB_HIP(ctx, hipSetDevice(ctx->device));
    *cloned = 1;
    if (len == 0) return AB_OK;
    const int64_t step = std::max<int64_t>(len / 100000, 1);
    const int64_t ns = (len + step - 1) / step;
    void *d = nullptr;
    AB_TRY(ab_scratch(ctx, (size_t)ns * sizeof(float), &d));
    hipLaunchKernelGGL(subsample_kernel, dim3((unsigned)((ns + 255) / 256)), dim3(256), 0, ctx->stream, img, len, step, (float *)d, ns);
    AB_HIP(ctx, hipGetLastError());
    std::vector<float> s(ns);
    AB_HIP(ctx, hipMemcpyAsync(s.data(), d, (size_t)ns * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
    AB_HIP(ctx, hipStreamSynchronize(ctx->stream));
    s.erase(std::remove_if(s.begin(), s.end(), [](float v) { return !std::isfinite(v); }), s.end());
    auto clone = [&]() -> int {
        if (out != img) AB_HIP(ctx, hipMemcpyAsync(out, img, (size_t)len * sizeof(float), hipMemcpyDeviceToDevice, ctx->stream));
        return AB_OK;
    };
    if (s.size() < 100) return clone();
    std::sort(s.begin(), s.end());
    const double lo = (double)s[s.size() / 100], hi = (double)s[s.size() * 999 / 1000];
    const double range = hi - lo;
    if (range < 1e-15) return clone();
    *cloned = 0;
    const int g = (int)std::min<int64_t>((len + 255) / 256, (int64_t)(ctx->cu_count > 0 ? ctx->cu_count : 256) * 8);
    hipLaunchKernelGGL(normalize_kernel, dim3(g), dim3(256), 0, ctx->stream, img, len, lo, 1.0 / range, out);
    AB_HIP(ctx, hipGetLastError());
    return AB_OK;
}

extern "C" {

int ab_estimate_background(ab_ctx *ctx, const ab_plane *img, int64_t tile_size, double *out_median, double *out_sigma) {
    if (!ctx) return AB_ERR_INVALID;
    AB_CHECK(ctx, img && out_median && out_sigma, "null argument");
    StagedPlane in;
    AB_TRY(ab_stage_in(ctx, img, &in));
    const int rc = ab_estimate_background_device(ctx, in.dptr, in.rows, in.cols, in.cols, tile_size, out_median, out_sigma);
    ab_stage_release(ctx, &in);
    return rc;
}

int ab_detect_stars(ab_ctx *ctx, const ab_plane *img, double sigma_threshold, ab_detected_star *out, size_t cap, size_t *out_count,
                    size_t *out_total, double *bg_median, double *bg_sigma) {
    if (!ctx) return AB_ERR_INVALID;
    AB_CHECK(ctx, img && out_count && (out || cap == 0), "null argument");
    AB_CHECK(ctx, img->data && img->rows > 0 && img->cols > 0, "plane is null or has a zero dimension");
    StagedPlane in;
    AB_TRY(ab_stage_in(ctx, img, &in));
    std::vector<ab_detected_star> stars;
    double m = 0.0, s = 1.0;
    const int rc = ab_detect_stars_device(ctx, in.dptr, in.rows, in.cols, in.cols, sigma_threshold, &stars, &m, &s);
    ab_stage_release(ctx, &in);
    if (rc != AB_OK) return rc;
    const size_t n = std::min(cap, stars.size());
    for (size_t i = 0; i < n; ++i) out[i] = stars[i];
    *out_count = n;
    if (out_total) *out_total = stars.size();
    if (bg_median) *bg_median = m;
    if (bg_sigma) *bg_sigma = s;
    return AB_OK;
}

int ab_normalize_for_detection(ab_ctx *ctx, const ab_plane *img, ab_plane_mut *out) {
    if (!ctx) return AB_ERR_INVALID;
    AB_CHECK(ctx, img && out && img->rows == out->rows && img->cols == out->cols, "null plane or mismatched dims");
    StagedPlane in;
    AB_TRY(ab_stage_in(ctx, img, &in));
    StagedOut so;
    int rc = ab_stage_out_begin(ctx, out, &so);
    if (rc == AB_OK) {
        int cloned = 0;
        rc = ab_normalize_for_detection_device(ctx, in.dptr, in.rows * in.cols, so.dptr, &cloned);
        if (rc == AB_OK)
            rc = ab_stage_out_finish(ctx, &so);
        else
            ab_stage_out_abort(ctx, &so);
    }
    ab_stage_release(ctx, &in);
    return rc;
}

}  // extern "C"
