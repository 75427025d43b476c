// Background extraction (gradient removal) on gfx950.
//
// Replaces core/imaging/background.rs: extract_background (:55-116), auto_sample_grid (:118-210),
// fit_polynomial_surface (:251-290), evaluate_polynomial_surface (:307-341), apply_correction
// (:343-383), compute_rms_residual (:385-415), solve_linear_system (:417-459).
//
// Device work (all HBM-bound passes over the plane):
//   * global median / MAD of the positive finite pixels: whole-image radix select (plane_select.hip;
//     exact order statistics, even counts averaged in f32 as median_f32_mut);
//   * one workgroup per grid cell takes the median of the cell's inner 50 % window
//     (block_select.hpp);
//   * the fitted surface is evaluated per pixel in f64 with the reference's cumulative power
//     tables and term order, then the correction (subtract / divide, re-centred on the model's
//     median) is applied in f32.
// Host work: the <= 32 x 32 samples' kappa-sigma rejection and the <= 21-term least-squares fit
// (Gaussian elimination with partial pivoting, f64) -- scalar maths the reference also runs serially.
#include "ab_common.hpp"
#include "block_select.hpp"

#include <algorithm>
#include <cmath>

namespace {

constexpr int kBlock = 256;
constexpr int kMaxPolyTerms = 21;
constexpr float kMadToSigmaF32 = (float)1.4826;

// ---- grid-cell medians (background.rs:151-190) ---------------------------------------------------------
struct CellOut {
    float median;
    unsigned int count;  // pixels that are finite and > 1e-7
};

__global__ __launch_bounds__(absel::kBlock) void cell_median_kernel(const float *__restrict__ img, int64_t ld, int grid,
                                                                    int cell_h, int cell_w, int margin_h, int margin_w,
                                                                    int inner_h, int inner_w, CellOut *__restrict__ out) {
    __shared__ unsigned int hist0[2048], hist[2048];
    const int gy = blockIdx.x / grid, gx = blockIdx.x % grid;
    absel::Window w;
    w.img = img;
    w.ld = ld;
    w.y0 = gy * cell_h + margin_h;
    w.x0 = gx * cell_w + margin_w;
    w.y1 = w.y0 + inner_h;
    w.x1 = w.x0 + inner_w;
    w.min_valid = 1e-7f;
    w.lo = -__builtin_inff();
    w.hi = __builtin_inff();
    const absel::Keying by_value = {0, 0.0, 0.0f};
    const absel::StreamSource src;  // a cell's inner window is streamed (any size): too many pixels for registers, and only ~4 passes are needed
    const unsigned int n = absel::prepare(src, w, by_value, hist0);
    float med = 0.0f;
    if (n > 0) med = absel::median_f32_from(src, w, by_value, hist0, n, hist);
    if (threadIdx.x == 0) {
        out[blockIdx.x].median = med;
        out[blockIdx.x].count = n;
    }
}

// ---- surface evaluation + correction (background.rs:307-383) ----------------------------------------------
struct PolyArgs {
    double coeffs[kMaxPolyTerms];
    int degree;
    int rows, cols;
    double row_scale, col_scale;
};

// evaluate_polynomial_surface (:307-341), one row per blockIdx.y, kPolyPx pixels of it per thread.  The reference's term is
// (coeffs[idx] * y_pows[yp]) * x_pows[xp]: the first product is the same for every pixel of a row, so a thread forms the row's
// DEG-dependent table once and spends one multiply and one add per term and pixel -- the same operations in the same order, bit
// for bit.  DEG is a template parameter: with a run-time degree the power tables were indexed dynamically and lived in scratch
// memory (459 us for 8192^2, 0.58 TB/s of model written; round 4).  `x / col_scale` stays an IEEE division unless col_scale is a
// power of two, where multiplying by its reciprocal is the same exact scaling.
constexpr int kPolyPx = 4;
template <int DEG>
__global__ __launch_bounds__(kBlock) void poly_model_kernel(const PolyArgs p, float *__restrict__ model, int pow2_cols, double inv_cols) {
    constexpr int T = (DEG + 1) * (DEG + 2) / 2;
    const int y = blockIdx.y;
    const double ny = (double)y / p.row_scale - 0.5;
    double y_pows[DEG + 1];
    y_pows[0] = 1.0;
#pragma unroll
    for (int i = 1; i <= DEG; ++i) y_pows[i] = y_pows[i - 1] * ny;
    double cy[T];  // coeffs[idx] * y_pows[yp] in eval_poly_inline's term order (:230-249)
    {
        int idx = 0;
#pragma unroll
        for (int total = 0; total <= DEG; ++total)
#pragma unroll
            for (int yp = total; yp >= 0; --yp) {
                cy[idx] = p.coeffs[idx] * y_pows[yp];
                ++idx;
            }
    }
#pragma unroll
    for (int u = 0; u < kPolyPx; ++u) {
        const int x = (blockIdx.x * kPolyPx + u) * kBlock + threadIdx.x;
        if (x >= p.cols) continue;
        const double nx = (pow2_cols ? (double)x * inv_cols : (double)x / p.col_scale) - 0.5;
        double x_pows[DEG + 1];
        x_pows[0] = 1.0;
#pragma unroll
        for (int i = 1; i <= DEG; ++i) x_pows[i] = x_pows[i - 1] * nx;
        double val = 0.0;
        int idx = 0;
#pragma unroll
        for (int total = 0; total <= DEG; ++total)
#pragma unroll
            for (int yp = total; yp >= 0; --yp) {
                val += cy[idx] * x_pows[total - yp];
                ++idx;
            }
        model[(size_t)y * p.cols + x] = (float)val;
    }
}

__global__ __launch_bounds__(kBlock) void bg_apply_kernel(const float *__restrict__ img, const float *__restrict__ model, int64_t n,
                                                          int mode, float model_median, float *__restrict__ out) {
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) {
        const float v = img[i], bg = model[i];
        float r;
        if (mode == 0) {
            r = v - bg + model_median;                                    // :367-369
        } else {
            r = fabsf(bg) > 1e-10f ? (v / bg) * model_median : v;        // :370-376
        }
        out[i] = r;
    }
}

// ---- host scalar maths ----------------------------------------------------------------------------------------
double powi(double a, int b) {  // f64::powi = compiler-rt __powidf2 (square-and-multiply, LSB first)
    double r = 1.0;
    for (;;) {
        if (b & 1) r *= a;
        b /= 2;
        if (b == 0) break;
        a *= a;
    }
    return r;
}

int poly_basis_into(double y, double x, int degree, double *out) {  // :217-228
    int idx = 0;
    for (int total = 0; total <= degree; ++total)
        for (int yp = total; yp >= 0; --yp) out[idx++] = powi(y, yp) * powi(x, total - yp);
    return idx;
}

bool solve_linear_system(std::vector<double> &a, std::vector<double> &b, int n) {  // :417-459
    for (int col = 0; col < n; ++col) {
        int max_row = col;
        double max_val = std::fabs(a[col * n + col]);
        for (int row = col + 1; row < n; ++row) {
            const double v = std::fabs(a[row * n + col]);
            if (v > max_val) {
                max_val = v;
                max_row = row;
            }
        }
        if (max_val < 1e-14) return false;
        if (max_row != col) {
            for (int k = 0; k < n; ++k) std::swap(a[col * n + k], a[max_row * n + k]);
            std::swap(b[col], b[max_row]);
        }
        const double pivot = a[col * n + col];
        for (int row = col + 1; row < n; ++row) {
            const double factor = a[row * n + col] / pivot;
            for (int k = col; k < n; ++k) a[row * n + k] -= factor * a[col * n + k];
            b[row] -= factor * b[col];
        }
    }
    for (int col = n - 1; col >= 0; --col) {
        double sum = b[col];
        for (int k = col + 1; k < n; ++k) sum -= a[col * n + k] * b[k];
        b[col] = sum / a[col * n + col];
    }
    return true;
}

float host_median_f32(std::vector<float> v) {  // median_f32_mut on a copy
    const size_t n = v.size();
    if (n == 0) return 0.0f;
    const size_t mid = n / 2;
    std::nth_element(v.begin(), v.begin() + mid, v.end());
    if (n % 2 == 0) {
        const float right = v[mid];
        const float left = *std::max_element(v.begin(), v.begin() + mid);
        return (left + right) / 2.0f;
    }
    return v[mid];
}

struct Sample {
    float y, x, value;
};

}  // namespace

extern "C" int ab_extract_background(ab_ctx *ctx, const ab_plane *img, const ab_background_config *cfg, ab_plane_mut *out_model,
                                     ab_plane_mut *out_corrected, ab_background_info *info) try {
    if (!ctx) return AB_ERR_INVALID;
    AB_CHECK(ctx, img && cfg && out_corrected, "null argument");
    const int64_t rows = img->rows, cols = img->cols, npix = rows * cols;
    AB_CHECK(ctx, out_corrected->rows == rows && out_corrected->cols == cols && (!out_model || (out_model->rows == rows && out_model->cols == cols)),
             "model / corrected planes must have the image's dims");
    const int grid = (int)cfg->grid_size, degree = (int)cfg->poly_degree;
    AB_CHECK(ctx, grid >= 1 && grid <= 64 && degree >= 0 && degree <= 5, "grid_size must be 1..64 and poly_degree 0..5 (<= 21 terms)");
    AB_CHECK(ctx, rows <= 65535 && npix < (int64_t(1) << 31), "image too large for this build");
    const int cell_h = (int)(rows / grid), cell_w = (int)(cols / grid);
    if (cell_h < 4 || cell_w < 4) return ab_set_error(ctx, AB_ERR_INVALID, "Image too small for grid_size=%d", grid);  // :127-129
    AB_CHECK(ctx, (int64_t)cell_h * cell_w <= ((int64_t)1 << 26), "grid cells larger than 2^26 px are not supported");  // u32 ranks, one block per cell
    const int margin_h = cell_h / 4, margin_w = cell_w / 4, inner_h = cell_h - 2 * margin_h, inner_w = cell_w - 2 * margin_w;
    AB_HIP(ctx, hipSetDevice(ctx->device));

    StagedPlane in;
    AB_TRY(ab_stage_in(ctx, img, &in));
    float *model = nullptr;
    CellOut *dcells = nullptr;
    StagedOut so_corr, so_model;
    bool corr_open = false, model_open = false;
    int rc = AB_OK;
    auto cleanup = [&]() {
        if (corr_open) ab_stage_out_abort(ctx, &so_corr);
        if (model_open) ab_stage_out_abort(ctx, &so_model);
        (void)hipStreamSynchronize(ctx->stream);
        ab_stage_release(ctx, &in);
    };
#define BG_TRY(expr)          \
    do {                      \
        rc = (expr);          \
        if (rc != AB_OK) {    \
            cleanup();        \
            return rc;        \
        }                     \
    } while (0)
#define BG_HIP(call)                                                                               \
    do {                                                                                           \
        hipError_t e_ = (call);                                                                    \
        if (e_ != hipSuccess) {                                                                    \
            rc = ab_set_error(ctx, AB_ERR_HIP, "%s failed: %s", #call, hipGetErrorString(e_));     \
            cleanup();                                                                             \
            return rc;                                                                             \
        }                                                                                          \
    } while (0)

    BG_TRY(ab_workspace(ctx, AB_WS_SCOPE1, (size_t)grid * grid * sizeof(CellOut), (void **)&dcells));  // (kept between calls: ab_common.hpp)
    BG_TRY(ab_progress(ctx, "sampling background", 1, 4));  // background.rs:63-66

    // global median / MAD over finite, > 0 pixels (:135-146)
    float global_median, global_mad;
    ab_plane_sel sel;
    sel.data = in.dptr;
    sel.n = npix;
    BG_TRY(ab_plane_median_f32(ctx, sel, &global_median, nullptr));
    sel.use_dev = 1;
    sel.center = global_median;
    BG_TRY(ab_plane_median_f32(ctx, sel, &global_mad, nullptr));
    const float sigma = global_mad * kMadToSigmaF32;

    // grid-cell medians (:150-190)
    hipLaunchKernelGGL(cell_median_kernel, dim3(grid * grid), dim3(absel::kBlock), 0, ctx->stream, in.dptr, in.cols, grid, cell_h,
                       cell_w, margin_h, margin_w, inner_h, inner_w, dcells);
    BG_HIP(hipGetLastError());
    std::vector<CellOut> cells((size_t)grid * grid);
    BG_HIP(hipMemcpyAsync(cells.data(), dcells, cells.size() * sizeof(CellOut), hipMemcpyDeviceToHost, ctx->stream));
    BG_HIP(hipStreamSynchronize(ctx->stream));
    std::vector<Sample> samples;
    const size_t total_cell = (size_t)inner_h * inner_w;
    for (int gy = 0; gy < grid; ++gy)
        for (int gx = 0; gx < grid; ++gx) {
            const CellOut &c = cells[(size_t)gy * grid + gx];
            const size_t zero_count = total_cell - c.count;
            if (c.count == 0 || (double)zero_count / (double)total_cell > 0.3) continue;
            const float lo = global_median - cfg->sigma_clip * sigma, hi = global_median + cfg->sigma_clip * sigma;
            if (c.median >= lo && c.median <= hi) {
                const int y0 = gy * cell_h + margin_h, x0 = gx * cell_w + margin_w;
                samples.push_back({(float)(y0 + inner_h / 2), (float)(x0 + inner_w / 2), c.median});
            }
        }
    const size_t n_terms = (size_t)(degree + 1) * (degree + 2) / 2, min_samples = n_terms + 2;
    for (size_t it = 1; it < cfg->iterations; ++it) {  // :192-207
        if (samples.size() < min_samples) break;
        std::vector<float> values;
        for (const Sample &s : samples) values.push_back(s.value);
        const float med = host_median_f32(values);
        for (float &v : values) v = std::fabs(v - med);
        const float mad = host_median_f32(values);
        const float sig = mad * kMadToSigmaF32;
        const float lo = med - cfg->sigma_clip * sig, hi = med + cfg->sigma_clip * sig;
        samples.erase(std::remove_if(samples.begin(), samples.end(), [&](const Sample &s) { return !(s.value >= lo && s.value <= hi); }),
                      samples.end());
    }
    if (info) info->sample_count = samples.size();
    if (samples.size() < min_samples) {  // :71-77
        rc = ab_set_error(ctx, AB_ERR_INVALID, "Not enough background samples (%zu) for polynomial degree %d", samples.size(), degree);
        cleanup();
        return rc;
    }

    BG_TRY(ab_progress(ctx, "fitting polynomial surface", 2, 4));  // background.rs:79-84 (cancel checked, then tick)
    // fit_polynomial_surface (:251-290)
    PolyArgs pa;
    memset(&pa, 0, sizeof pa);
    pa.degree = degree;
    pa.rows = (int)rows;
    pa.cols = (int)cols;
    pa.row_scale = (double)rows;
    pa.col_scale = (double)cols;
    std::vector<double> ata(n_terms * n_terms, 0.0), atb(n_terms, 0.0);
    double basis[kMaxPolyTerms];
    for (const Sample &s : samples) {
        const double ny = (double)s.y / pa.row_scale - 0.5, nx = (double)s.x / pa.col_scale - 0.5, val = (double)s.value;
        const int cnt = poly_basis_into(ny, nx, degree, basis);
        for (int i = 0; i < cnt; ++i) {
            atb[i] += basis[i] * val;
            for (int j = 0; j < cnt; ++j) ata[i * n_terms + j] += basis[i] * basis[j];
        }
    }
    for (size_t i = 0; i < n_terms; ++i) ata[i * n_terms + i] += 1e-8;
    if (!solve_linear_system(ata, atb, (int)n_terms)) {
        rc = ab_set_error(ctx, AB_ERR_INVALID, "Failed to solve polynomial fit: Singular matrix in polynomial fit");
        cleanup();
        return rc;
    }
    for (size_t i = 0; i < n_terms; ++i) pa.coeffs[i] = atb[i];
    if (info) memcpy(info->coeffs, pa.coeffs, sizeof pa.coeffs);

    BG_TRY(ab_progress(ctx, "generating model", 3, 4));  // background.rs:88-93
    // evaluate_polynomial_surface + apply_correction (:307-383)
    if (out_model) {
        BG_TRY(ab_stage_out_begin(ctx, out_model, &so_model));
        model_open = true;
        model = so_model.dptr;
    } else {
        BG_TRY(ab_workspace(ctx, AB_WS_SCOPE0, (size_t)npix * sizeof(float), (void **)&model));
    }
    {
        const int pow2_cols = (cols & (cols - 1)) == 0 ? 1 : 0;
        const double inv_cols = 1.0 / (double)cols;  // exact when cols is a power of two
        const dim3 pgrid((unsigned)((cols + kBlock * kPolyPx - 1) / (kBlock * kPolyPx)), (unsigned)rows), pblock(kBlock);
        switch (degree) {  // 0 .. 5 (checked above)
        case 0: hipLaunchKernelGGL(poly_model_kernel<0>, pgrid, pblock, 0, ctx->stream, pa, model, pow2_cols, inv_cols); break;
        case 1: hipLaunchKernelGGL(poly_model_kernel<1>, pgrid, pblock, 0, ctx->stream, pa, model, pow2_cols, inv_cols); break;
        case 2: hipLaunchKernelGGL(poly_model_kernel<2>, pgrid, pblock, 0, ctx->stream, pa, model, pow2_cols, inv_cols); break;
        case 3: hipLaunchKernelGGL(poly_model_kernel<3>, pgrid, pblock, 0, ctx->stream, pa, model, pow2_cols, inv_cols); break;
        case 4: hipLaunchKernelGGL(poly_model_kernel<4>, pgrid, pblock, 0, ctx->stream, pa, model, pow2_cols, inv_cols); break;
        default: hipLaunchKernelGGL(poly_model_kernel<5>, pgrid, pblock, 0, ctx->stream, pa, model, pow2_cols, inv_cols); break;
        }
    }
    BG_HIP(hipGetLastError());
    float model_median = 0.0f;
    ab_plane_sel msel;
    msel.data = model;
    msel.n = npix;
    BG_TRY(ab_plane_median_f32(ctx, msel, &model_median, nullptr));
    BG_TRY(ab_progress(ctx, "applying correction", 4, 4));  // background.rs:97-99
    BG_TRY(ab_stage_out_begin(ctx, out_corrected, &so_corr));
    corr_open = true;
    const int g = (int)std::min<int64_t>((npix + kBlock - 1) / kBlock, (int64_t)(ctx->cu_count > 0 ? ctx->cu_count : 256) * 8);
    hipLaunchKernelGGL(bg_apply_kernel, dim3(g), dim3(kBlock), 0, ctx->stream, in.dptr, model, npix, (int)cfg->mode, model_median,
                       so_corr.dptr);
    BG_HIP(hipGetLastError());
    rc = ab_stage_out_finish(ctx, &so_corr);
    corr_open = false;
    if (rc == AB_OK && model_open) {
        rc = ab_stage_out_finish(ctx, &so_model);
        model_open = false;
    }
    // compute_rms_residual (:385-415)
    if (rc == AB_OK && info) {
        double sum_sq = 0.0;
        for (const Sample &s : samples) {
            double yp[7] = {1.0, 0, 0, 0, 0, 0, 0}, xp[7] = {1.0, 0, 0, 0, 0, 0, 0};
            const double ny = (double)s.y / pa.row_scale - 0.5, nx = (double)s.x / pa.col_scale - 0.5;
            for (int i = 1; i <= std::min(degree, 6); ++i) {
                yp[i] = yp[i - 1] * ny;
                xp[i] = xp[i - 1] * nx;
            }
            double val = 0.0;
            int idx = 0;
            for (int total = 0; total <= degree; ++total)
                for (int ypow = total; ypow >= 0; --ypow) {
                    val += pa.coeffs[idx] * yp[ypow] * xp[total - ypow];
                    ++idx;
                }
            const double diff = (double)s.value - val;
            sum_sq += diff * diff;
        }
        info->rms_residual = std::sqrt(sum_sq / (double)samples.size());
    }
    cleanup();
    return rc;
#undef BG_TRY
#undef BG_HIP
} AB_CATCH(ctx)
