// Phase-correlation registration on gfx950.
//
// Replaces core/alignment/phase_correlation.rs (phase_correlate :22-89, correlate_single
// :105-141, is_constant_or_zero :143-160), core/alignment/downsample.rs:6-46 (area_downsample)
// and the math/ pieces they use: window.rs:3-18, fft.rs:136-167,202-226,271-282,
// complex.rs:27-44, subpixel.rs:27-100, normalization.rs:128-170.
//
// The transform sizes are fixed by the algorithm: everything larger than 512 px is first
// area-averaged to 512 x 512 and then refined on a centred 512 x 512 crop, so the work per frame is
// three 512^2 complex-f64 2-D FFTs twice over (4 MiB per buffer, L2 resident).  One workgroup
// runs one line through an LDS-resident radix-2 decimation-in-time FFT (8 KiB of f64 pairs,
// one butterfly per thread per stage, twiddles from a host-made table).  Rows first, then columns,
// exactly the reference's order.  The reference's FFT is rustfft (not reproducible bit for bit,
// its tests pin the shift to +-1 px); the CPU oracle uses the same butterflies as this kernel, so
// the correlation surface is bit-identical between the two and the estimated shift agrees to ~1e-12.
// find_peak's tie-break (schedule dependent in the reference) is the lowest index.
#include "ab_common.hpp"

#include <cfloat>
#include <cmath>

namespace {

constexpr int kCoarseMaxDim = 512;   // phase_correlation.rs:10
constexpr int kRefineCropSize = 512; // :11
constexpr double kEpsilon = 1e-15;   // :13
constexpr int kBlock = 256;

struct MinMaxPartial {
    float mn, mx;
    unsigned long long finite;
};

// phase_correlation.rs:143-160
__global__ __launch_bounds__(kBlock) void minmax_finite_kernel(const float *__restrict__ img, int rows, int cols, int64_t ld,
                                                               MinMaxPartial *__restrict__ partials) {
    float mn = __builtin_inff(), mx = -__builtin_inff();
    unsigned long long cnt = 0;
    auto take = [&](float v) {
        if (__builtin_isfinite(v)) {
            mn = v < mn ? v : mn;
            mx = v > mx ? v : mx;
            cnt += 1;
        }
    };
    // a workgroup walks whole rows (no 64-bit division per pixel: the first version took 540 us for a 8192 x 8192 plane),
    // 16 bytes per lane and load where the rows allow it, two loads in flight
    const bool vec = (((uintptr_t)img) & 15) == 0 && (ld & 3) == 0;
    for (int y = blockIdx.x; y < rows; y += gridDim.x) {
        const float *row = img + (int64_t)y * ld;
        int x = 0;
        if (vec) {
            const float4 *r4 = reinterpret_cast<const float4 *>(row);
            const int c4 = cols >> 2;
            int i = threadIdx.x;
            for (; i + kBlock < c4; i += 2 * kBlock) {
                const float4 a = r4[i], b = r4[i + kBlock];
                take(a.x), take(a.y), take(a.z), take(a.w);
                take(b.x), take(b.y), take(b.z), take(b.w);
            }
            if (i < c4) {
                const float4 a = r4[i];
                take(a.x), take(a.y), take(a.z), take(a.w);
            }
            x = c4 << 2;
        }
        for (int i = x + threadIdx.x; i < cols; i += kBlock) take(row[i]);
    }
    __shared__ float s_mn[kBlock / 64], s_mx[kBlock / 64];
    __shared__ unsigned long long s_c[kBlock / 64];
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        mn = fminf(mn, __shfl_xor(mn, off, 64));
        mx = fmaxf(mx, __shfl_xor(mx, off, 64));
        cnt += __shfl_xor(cnt, off, 64);
    }
    if ((threadIdx.x & 63) == 0) {
        s_mn[threadIdx.x >> 6] = mn;
        s_mx[threadIdx.x >> 6] = mx;
        s_c[threadIdx.x >> 6] = cnt;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int i = 1; i < kBlock / 64; ++i) {
            mn = fminf(mn, s_mn[i]);
            mx = fmaxf(mx, s_mx[i]);
            cnt += s_c[i];
        }
        partials[blockIdx.x].mn = mn;
        partials[blockIdx.x].mx = mx;
        partials[blockIdx.x].finite = cnt;
    }
}

// downsample.rs:18-43: box average of the finite samples, f64 sum in row-major order
__global__ __launch_bounds__(kBlock) void area_downsample_kernel(const float *__restrict__ src, int in_rows, int in_cols,
                                                                 int64_t ld, int out_rows, int out_cols, double scale_y,
                                                                 double scale_x, float *__restrict__ out) {
    const int idx = blockIdx.x * kBlock + threadIdx.x;
    if (idx >= out_rows * out_cols) return;
    const int oy = idx / out_cols, ox = idx - oy * out_cols;
    auto clampi = [](long long v, int len) { return v < 0 ? 0 : (v >= len ? len - 1 : (int)v); };
    const int y0 = clampi((long long)floor((double)oy * scale_y), in_rows);
    const long long y1r = (long long)ceil((double)(oy + 1) * scale_y);
    const int y1 = y1r <= 0 ? 0 : (y1r < in_rows ? (int)y1r : in_rows);
    const int x0 = clampi((long long)floor((double)ox * scale_x), in_cols);
    const long long x1r = (long long)ceil((double)(ox + 1) * scale_x);
    const int x1 = x1r <= 0 ? 0 : (x1r < in_cols ? (int)x1r : in_cols);
    double sum = 0.0;
    unsigned count = 0;
    for (int y = y0; y < y1; ++y)
        for (int x = x0; x < x1; ++x) {
            const float v = src[y * ld + x];
            if (__builtin_isfinite(v)) {
                sum += (double)v;
                count += 1;
            }
        }
    out[idx] = count > 0 ? (float)(sum / (double)count) : 0.0f;
}

// fft.rs:202-226: v * wy * wx for finite v, zero padded to the power-of-two buffer
__global__ __launch_bounds__(kBlock) void window_pad_kernel(const float *__restrict__ img, int rows, int cols, int64_t ld,
                                                            const double *__restrict__ win_y, const double *__restrict__ win_x,
                                                            int fft_rows, int fft_cols, double2 *__restrict__ out) {
    const int idx = blockIdx.x * kBlock + threadIdx.x;
    if (idx >= fft_rows * fft_cols) return;
    const int y = idx / fft_cols, x = idx - y * fft_cols;
    double re = 0.0;
    if (y < rows && x < cols) {
        const double v = (double)img[y * ld + x];
        re = __builtin_isfinite(v) ? v * win_y[y] * win_x[x] : 0.0;
    }
    out[idx] = make_double2(re, 0.0);
}

// one line per workgroup, radix-2 DIT in LDS; tw[k] = exp(-2 pi i k / n)
__global__ __launch_bounds__(kBlock) void fft_lines_kernel(double2 *data, int n, int log2n, int64_t elem_stride,
                                                           int64_t line_stride, const double2 *__restrict__ tw, int inverse) {
    __shared__ double2 s[512];
    double2 *line = data + (int64_t)blockIdx.x * line_stride;
    for (int i = threadIdx.x; i < n; i += kBlock) {
        const int r = log2n ? (int)(__brev((unsigned)i) >> (32 - log2n)) : 0;
        s[r] = line[(int64_t)i * elem_stride];
    }
    __syncthreads();
    for (int m = 2; m <= n; m <<= 1) {
        const int half = m >> 1, step = n / m;
        for (int t = threadIdx.x; t < n / 2; t += kBlock) {
            const int k = (t / half) * m, j = t % half;
            double2 w = tw[j * step];
            if (inverse) w.y = -w.y;
            const double2 x = s[k + j + half];
            const double tr = w.x * x.x - w.y * x.y;
            const double ti = w.x * x.y + w.y * x.x;
            const double2 u = s[k + j];
            s[k + j] = make_double2(u.x + tr, u.y + ti);
            s[k + j + half] = make_double2(u.x - tr, u.y - ti);
        }
        __syncthreads();
    }
    for (int i = threadIdx.x; i < n; i += kBlock) line[(int64_t)i * elem_stride] = s[i];
}

// complex.rs:27-44, in place into fa
__global__ __launch_bounds__(kBlock) void cross_power_kernel(double2 *fa, const double2 *__restrict__ fb, int n, double eps) {
    const int i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    const double2 a = fa[i], b = fb[i];
    const double pr = a.x * b.x + a.y * b.y;
    const double pi = a.y * b.x - a.x * b.y;
    const double mag = sqrt(pr * pr + pi * pi);
    fa[i] = mag > eps ? make_double2(pr / mag, pi / mag) : make_double2(0.0, 0.0);
}

struct PeakPartial {
    double best;
    int best_idx;
    double sum, count;
};

// inverse_2d's 1/(rows*cols) scaling + extract_real + find_peak + first moment, one pass
// (blockIdx.y = pair of a batch: every pointer advances by its pair stride; a single correlation launches one row of workgroups)
__global__ __launch_bounds__(kBlock) void scale_peak_kernel(const double2 *__restrict__ buf, int n, double norm,
                                                            double *__restrict__ corr, PeakPartial *__restrict__ partials, int partials_per_pair = 0) {
    buf += (size_t)blockIdx.y * n;
    corr += (size_t)blockIdx.y * n;
    partials += (size_t)blockIdx.y * partials_per_pair;
    double best = -DBL_MAX, sum = 0.0, count = 0.0;
    int best_idx = 0x7fffffff;
    const int stride = gridDim.x * kBlock;
    for (int i = blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) {
        const double v = buf[i].x * norm;  // c.re * norm (fft.rs:161-166)
        corr[i] = v;
        if (v > best) {  // ascending i per thread: the first maximum is kept
            best = v;
            best_idx = i;
        }
        if (__builtin_isfinite(v)) {
            sum += v;
            count += 1.0;
        }
    }
    __shared__ PeakPartial sh[kBlock];
    sh[threadIdx.x] = {best, best_idx, sum, count};
    __syncthreads();
    for (int off = kBlock / 2; off >= 1; off >>= 1) {
        if ((int)threadIdx.x < off) {
            PeakPartial &a = sh[threadIdx.x];
            const PeakPartial &b = sh[threadIdx.x + off];
            // maximum value, lowest index among equals; NaN never wins (v > best is false)
            if (b.best_idx != 0x7fffffff && (a.best_idx == 0x7fffffff || b.best > a.best || (b.best == a.best && b.best_idx < a.best_idx))) {
                a.best = b.best;
                a.best_idx = b.best_idx;
            }
            a.sum += b.sum;
            a.count += b.count;
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) partials[blockIdx.x] = sh[0];
}

// What the host used to do between the kernels of a correlation -- pick the peak among the workgroups' partials, form the mean,
// add up the variance partials, read the peak's four neighbours -- in the host's own (sequential) order, by one thread each, so
// that a correlation ends in ONE read-back and one synchronisation instead of four (a 4 x 1600 x 1600 stack_images(align) spent
// more time in its 30 host joins than in its kernels).
struct PcFin {
    int best_idx, pad;
    double mean, count, var_sum;
    double sv[5];
};
// (a workgroup fetches the partials into LDS together; one thread then walks them in order: 256 dependent global loads took 60 us)
__global__ __launch_bounds__(256) void peak_finish_kernel(const PeakPartial *__restrict__ pp, int pg, PcFin *fin, int partials_per_pair = 0) {
    pp += (size_t)blockIdx.x * partials_per_pair;  // (one workgroup per pair)
    fin += blockIdx.x;
    __shared__ PeakPartial sh[256];
    if ((int)threadIdx.x < pg) sh[threadIdx.x] = pp[threadIdx.x];
    __syncthreads();
    if (threadIdx.x != 0) return;
    double best = -DBL_MAX, sum = 0.0, count = 0.0;
    int best_idx = 0x7fffffff;
    for (int i = 0; i < pg; ++i) {
        const PeakPartial p = sh[i];
        if (p.best_idx != 0x7fffffff && (best_idx == 0x7fffffff || p.best > best || (p.best == best && p.best_idx < best_idx))) {
            best = p.best;
            best_idx = p.best_idx;
        }
        sum += p.sum;
        count += p.count;
    }
    fin->best_idx = best_idx == 0x7fffffff ? 0 : best_idx;
    fin->count = count;
    fin->mean = count >= 1.0 ? sum / count : 0.0;  // normalization.rs:128-161
}
__global__ __launch_bounds__(256) void var_finish_kernel(const double *__restrict__ vp, int pg, const double *__restrict__ corr, int fr, int fc, PcFin *fin,
                                                         int partials_per_pair = 0) {
    vp += (size_t)blockIdx.x * partials_per_pair;  // (one workgroup per pair)
    corr += (size_t)blockIdx.x * fr * fc;
    fin += blockIdx.x;
    __shared__ double sh[256];
    if ((int)threadIdx.x < pg) sh[threadIdx.x] = vp[threadIdx.x];
    __syncthreads();
    if (threadIdx.x != 0) return;
    double var_sum = 0.0;
    if (fin->count >= 1.0)
        for (int i = 0; i < pg; ++i) var_sum += sh[i];
    fin->var_sum = var_sum;
    const int py = fin->best_idx / fc, px = fin->best_idx % fc;
    // the 5 surface samples of the 3-point refinements (subpixel.rs:27-62), wrap-around neighbours
    fin->sv[0] = corr[py * fc + px];
    fin->sv[1] = corr[(py == 0 ? fr - 1 : py - 1) * fc + px];
    fin->sv[2] = corr[(py == fr - 1 ? 0 : py + 1) * fc + px];
    fin->sv[3] = corr[py * fc + (px == 0 ? fc - 1 : px - 1)];
    fin->sv[4] = corr[py * fc + (px == fc - 1 ? 0 : px + 1)];
}

__global__ __launch_bounds__(kBlock) void var_kernel(const double *__restrict__ corr, int n, const PcFin *__restrict__ fin, double *__restrict__ partials,
                                                     int partials_per_pair = 0) {
    corr += (size_t)blockIdx.y * n;
    fin += blockIdx.y;
    partials += (size_t)blockIdx.y * partials_per_pair;
    const double mean = fin->mean;
    double vs = 0.0;
    const int stride = gridDim.x * kBlock;
    for (int i = blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) {
        const double v = corr[i];
        if (__builtin_isfinite(v)) {
            const double d = v - mean;
            vs += d * d;
        }
    }
    __shared__ double sh[kBlock];
    sh[threadIdx.x] = vs;
    __syncthreads();
    for (int off = kBlock / 2; off >= 1; off >>= 1) {
        if ((int)threadIdx.x < off) sh[threadIdx.x] += sh[threadIdx.x + off];
        __syncthreads();
    }
    if (threadIdx.x == 0) partials[blockIdx.x] = sh[0];
}

// ---- the same kernels with blockIdx.y = one PAIR of a batch (round 5: stack_images(align) correlates every frame with frame 0) ----
// Each is the single-pair kernel's body on the pair's own source view / buffers: the arithmetic and its order are the single
// pair's, so a batch returns bit for bit what n calls of ab_phase_correlate_device return.
struct PcItem {  // the source view of pair b: a plane, a downsampled plane or a crop of one
    const float *p;
    int64_t ld;
    int rows, cols;
};
__global__ __launch_bounds__(kBlock) void minmax_finite_many_kernel(const PcItem *__restrict__ items, MinMaxPartial *__restrict__ partials, int per_item) {
    const PcItem it = items[blockIdx.y];
    const float *img = it.p;
    const int rows = it.rows, cols = it.cols;
    const int64_t ld = it.ld;
    float mn = __builtin_inff(), mx = -__builtin_inff();
    unsigned long long cnt = 0;
    auto take = [&](float v) {
        if (__builtin_isfinite(v)) {
            mn = v < mn ? v : mn;
            mx = v > mx ? v : mx;
            cnt += 1;
        }
    };
    const bool vec = (((uintptr_t)img) & 15) == 0 && (ld & 3) == 0;
    for (int y = blockIdx.x; y < rows; y += gridDim.x) {
        const float *row = img + (int64_t)y * ld;
        int x = 0;
        if (vec) {
            const float4 *r4 = reinterpret_cast<const float4 *>(row);
            const int c4 = cols >> 2;
            int i = threadIdx.x;
            for (; i + kBlock < c4; i += 2 * kBlock) {
                const float4 a = r4[i], b = r4[i + kBlock];
                take(a.x), take(a.y), take(a.z), take(a.w);
                take(b.x), take(b.y), take(b.z), take(b.w);
            }
            if (i < c4) {
                const float4 a = r4[i];
                take(a.x), take(a.y), take(a.z), take(a.w);
            }
            x = c4 << 2;
        }
        for (int i = x + threadIdx.x; i < cols; i += kBlock) take(row[i]);
    }
    __shared__ float s_mn[kBlock / 64], s_mx[kBlock / 64];
    __shared__ unsigned long long s_c[kBlock / 64];
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        mn = fminf(mn, __shfl_xor(mn, off, 64));
        mx = fmaxf(mx, __shfl_xor(mx, off, 64));
        cnt += __shfl_xor(cnt, off, 64);
    }
    if ((threadIdx.x & 63) == 0) {
        s_mn[threadIdx.x >> 6] = mn;
        s_mx[threadIdx.x >> 6] = mx;
        s_c[threadIdx.x >> 6] = cnt;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int i = 1; i < kBlock / 64; ++i) {
            mn = fminf(mn, s_mn[i]);
            mx = fmaxf(mx, s_mx[i]);
            cnt += s_c[i];
        }
        MinMaxPartial *o = partials + (size_t)blockIdx.y * per_item + blockIdx.x;
        o->mn = mn;
        o->mx = mx;
        o->finite = cnt;
    }
}
__global__ __launch_bounds__(kBlock) void area_downsample_many_kernel(const PcItem *__restrict__ items, int out_rows, int out_cols, double scale_y, double scale_x,
                                                                      float *__restrict__ out_all) {
    const PcItem it = items[blockIdx.y];
    const float *src = it.p;
    const int in_rows = it.rows, in_cols = it.cols;
    const int64_t ld = it.ld;
    float *out = out_all + (size_t)blockIdx.y * out_rows * out_cols;
    const int idx = blockIdx.x * kBlock + threadIdx.x;
    if (idx >= out_rows * out_cols) return;
    const int oy = idx / out_cols, ox = idx - oy * out_cols;
    auto clampi = [](long long v, int len) { return v < 0 ? 0 : (v >= len ? len - 1 : (int)v); };
    const int y0 = clampi((long long)floor((double)oy * scale_y), in_rows);
    const long long y1r = (long long)ceil((double)(oy + 1) * scale_y);
    const int y1 = y1r <= 0 ? 0 : (y1r < in_rows ? (int)y1r : in_rows);
    const int x0 = clampi((long long)floor((double)ox * scale_x), in_cols);
    const long long x1r = (long long)ceil((double)(ox + 1) * scale_x);
    const int x1 = x1r <= 0 ? 0 : (x1r < in_cols ? (int)x1r : in_cols);
    double sum = 0.0;
    unsigned count = 0;
    for (int y = y0; y < y1; ++y)
        for (int x = x0; x < x1; ++x) {
            const float v = src[y * ld + x];
            if (__builtin_isfinite(v)) {
                sum += (double)v;
                count += 1;
            }
        }
    out[idx] = count > 0 ? (float)(sum / (double)count) : 0.0f;
}
__global__ __launch_bounds__(kBlock) void window_pad_many_kernel(const PcItem *__restrict__ items, const double *__restrict__ win_y, const double *__restrict__ win_x,
                                                                 int fft_rows, int fft_cols, double2 *__restrict__ out_all) {
    const PcItem it = items[blockIdx.y];
    double2 *out = out_all + (size_t)blockIdx.y * fft_rows * fft_cols;
    const int idx = blockIdx.x * kBlock + threadIdx.x;
    if (idx >= fft_rows * fft_cols) return;
    const int y = idx / fft_cols, x = idx - y * fft_cols;
    double re = 0.0;
    if (y < it.rows && x < it.cols) {
        const double v = (double)it.p[y * it.ld + x];
        re = __builtin_isfinite(v) ? v * win_y[y] * win_x[x] : 0.0;
    }
    out[idx] = make_double2(re, 0.0);
}
__global__ __launch_bounds__(kBlock) void fft_lines_many_kernel(double2 *data_all, int64_t batch_stride, int n, int log2n, int64_t elem_stride, int64_t line_stride,
                                                                const double2 *__restrict__ tw, int inverse) {
    __shared__ double2 s[512];
    double2 *line = data_all + (int64_t)blockIdx.y * batch_stride + (int64_t)blockIdx.x * line_stride;
    for (int i = threadIdx.x; i < n; i += kBlock) {
        const int r = log2n ? (int)(__brev((unsigned)i) >> (32 - log2n)) : 0;
        s[r] = line[(int64_t)i * elem_stride];
    }
    __syncthreads();
    for (int m = 2; m <= n; m <<= 1) {
        const int half = m >> 1, step = n / m;
        for (int t = threadIdx.x; t < n / 2; t += kBlock) {
            const int k = (t / half) * m, j = t % half;
            double2 w = tw[j * step];
            if (inverse) w.y = -w.y;
            const double2 x = s[k + j + half];
            const double tr = w.x * x.x - w.y * x.y;
            const double ti = w.x * x.y + w.y * x.x;
            const double2 u = s[k + j];
            s[k + j] = make_double2(u.x + tr, u.y + ti);
            s[k + j + half] = make_double2(u.x - tr, u.y - ti);
        }
        __syncthreads();
    }
    for (int i = threadIdx.x; i < n; i += kBlock) line[(int64_t)i * elem_stride] = s[i];
}
// complex.rs:27-44 with the reference's spectrum shared by every pair: fb <- A conj(B) / |A conj(B)|
__global__ __launch_bounds__(kBlock) void cross_power_many_kernel(const double2 *__restrict__ fa, double2 *fb_all, int n, double eps) {
    const int i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    double2 *fb = fb_all + (size_t)blockIdx.y * n;
    const double2 a = fa[i], b = fb[i];
    const double pr = a.x * b.x + a.y * b.y;
    const double pi = a.y * b.x - a.x * b.y;
    const double mag = sqrt(pr * pr + pi * pi);
    fb[i] = mag > eps ? make_double2(pr / mag, pi / mag) : make_double2(0.0, 0.0);
}

// ---------------------------------------------------------------------------------------------------
int next_pow2(int n) {
    int p = 1;
    while (p < n) p <<= 1;
    return p;
}
int ilog2(int n) {
    int l = 0;
    while ((1 << l) < n) ++l;
    return l;
}

struct View {  // a rows x cols window of a device plane with row stride ld
    const float *p;
    int rows, cols;
    int64_t ld;
};

constexpr int kPartials = 256;
constexpr int kMinMaxPartials = 2048;  // (min / max / count are order-independent: as many workgroups as fill the chip)

struct PcScratch {
    double2 *fa, *fb;      // 512 x 512 each
    double *corr;          // 512 x 512
    double *hann_y, *hann_x;
    double2 *tw_r, *tw_c;  // 256 each
    float *ds_a, *ds_b;    // 512 x 512 downsampled planes
    void *partials;        // kMinMaxPartials x 32 B (min / max partials of two planes; peak partials)
    double *var_partials;  // kPartials
    PcFin *fin;
};

void hann_periodic(int n, std::vector<double> &w);
void twiddles(int n, std::vector<double> &tw);

int pc_carve(ab_ctx *ctx, PcScratch *s) {
    const size_t n = 512 * 512;
    const size_t bytes = 2 * n * sizeof(double2) + n * sizeof(double) + 2 * 512 * sizeof(double) + 2 * 256 * sizeof(double2) +
                         2 * n * sizeof(float) + 2 * kMinMaxPartials * 32 + kPartials * 8 + sizeof(PcFin) + 512;
    void *p = nullptr;
    AB_TRY(ab_scratch(ctx, bytes, &p));
    char *c = (char *)p;
    s->fa = (double2 *)c; c += n * sizeof(double2);
    s->fb = (double2 *)c; c += n * sizeof(double2);
    s->corr = (double *)c; c += n * sizeof(double);
    s->hann_y = (double *)c; c += 512 * sizeof(double);
    s->hann_x = (double *)c; c += 512 * sizeof(double);
    s->tw_r = (double2 *)c; c += 256 * sizeof(double2);
    s->tw_c = (double2 *)c; c += 256 * sizeof(double2);
    s->ds_a = (float *)c; c += n * sizeof(float);
    s->ds_b = (float *)c; c += n * sizeof(float);
    s->partials = (void *)c; c += 2 * kMinMaxPartials * 32;
    s->var_partials = (double *)c; c += kPartials * 8;
    s->fin = (PcFin *)c;
    return AB_OK;
}

// Hann windows and twiddles for (rows, cols) in table set `set` (0: the coarse / only correlation, 1: the refinement crop): kept
// in a workspace of their own between calls, uploaded only when the dimensions change (every call used to upload four tables from
// pageable memory and wait for them)
int pc_tables(ab_ctx *ctx, PcScratch *s, int set, int rows, int cols, int fr, int fc) {
    const size_t per = 2 * 512 * sizeof(double) + 2 * 256 * sizeof(double2);
    char *t = nullptr;
    AB_TRY(ab_workspace(ctx, AB_WS_PHASE_TABLES, 2 * per, (void **)&t));
    if (ctx->pc_tab_ws != (const void *)t) {
        ctx->pc_tab_ws = t;
        ctx->pc_tab_dims[0][0] = ctx->pc_tab_dims[1][0] = 0;
    }
    t += (size_t)set * per;
    s->hann_y = (double *)t;
    s->hann_x = s->hann_y + 512;
    s->tw_r = (double2 *)(s->hann_x + 512);
    s->tw_c = s->tw_r + 256;
    if (ctx->pc_tab_dims[set][0] == rows && ctx->pc_tab_dims[set][1] == cols) return AB_OK;
    std::vector<double> hy, hx, twr, twc;
    hann_periodic(rows, hy);
    hann_periodic(cols, hx);
    twiddles(fr, twr);
    twiddles(fc, twc);
    AB_HIP(ctx, hipMemcpyAsync(s->hann_y, hy.data(), rows * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    AB_HIP(ctx, hipMemcpyAsync(s->hann_x, hx.data(), cols * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    AB_HIP(ctx, hipMemcpyAsync(s->tw_r, twr.data(), twr.size() * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    AB_HIP(ctx, hipMemcpyAsync(s->tw_c, twc.data(), twc.size() * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    AB_HIP(ctx, hipStreamSynchronize(ctx->stream));  // the host tables are locals
    ctx->pc_tab_dims[set][0] = rows;
    ctx->pc_tab_dims[set][1] = cols;
    return AB_OK;
}

void hann_periodic(int n, std::vector<double> &w) {  // window.rs:3-18
    w.resize(n);
    if (n == 1) {
        w[0] = 1.0;
        return;
    }
    const double two_pi = 2.0 * 3.14159265358979323846, nf = (double)n;
    for (int i = 0; i < n; ++i) w[i] = 0.5 * (1.0 - std::cos(two_pi * (double)i / nf));
}

void twiddles(int n, std::vector<double> &tw) {
    tw.assign(2 * std::max(1, n / 2), 0.0);
    for (int k = 0; k < n / 2; ++k) {
        const double ang = -2.0 * 3.14159265358979323846 * (double)k / (double)n;
        tw[2 * k] = std::cos(ang);
        tw[2 * k + 1] = std::sin(ang);
    }
}

int fft2d(ab_ctx *ctx, double2 *buf, int fr, int fc, const PcScratch &s, int inverse) {
    hipLaunchKernelGGL(fft_lines_kernel, dim3(fr), dim3(kBlock), 0, ctx->stream, buf, fc, ilog2(fc), (int64_t)1, (int64_t)fc,
                       s.tw_c, inverse);  // rows
    hipLaunchKernelGGL(fft_lines_kernel, dim3(fc), dim3(kBlock), 0, ctx->stream, buf, fr, ilog2(fr), (int64_t)fc, (int64_t)1,
                       s.tw_r, inverse);  // columns
    AB_HIP(ctx, hipGetLastError());
    return AB_OK;
}

// is_constant_or_zero (:143-160) of both planes with one read-back (the reference tests the target only if the reference passes:
// the result is the same)
int constant_or_zero2(ab_ctx *ctx, const View &r, const View &t, const PcScratch &s, bool *cr, bool *ct) {
    const View *v[2] = {&r, &t};
    int grid[2];
    MinMaxPartial *part = (MinMaxPartial *)s.partials;
    for (int k = 0; k < 2; ++k) {
        grid[k] = (int)std::min<int64_t>(kMinMaxPartials, v[k]->rows);
        hipLaunchKernelGGL(minmax_finite_kernel, dim3(grid[k]), dim3(kBlock), 0, ctx->stream, v[k]->p, v[k]->rows, v[k]->cols, v[k]->ld,
                           part + (size_t)k * kMinMaxPartials);
    }
    AB_HIP(ctx, hipGetLastError());
    void *pin = nullptr;
    AB_TRY(ab_pinned(ctx, 2 * kMinMaxPartials * sizeof(MinMaxPartial), &pin));
    AB_HIP(ctx, hipMemcpyAsync(pin, part, (size_t)(kMinMaxPartials + grid[1]) * sizeof(MinMaxPartial), hipMemcpyDeviceToHost, ctx->stream));
    AB_HIP(ctx, hipStreamSynchronize(ctx->stream));
    bool out[2];
    for (int k = 0; k < 2; ++k) {
        const MinMaxPartial *h = (const MinMaxPartial *)pin + (size_t)k * kMinMaxPartials;
        float mn = INFINITY, mx = -INFINITY;
        unsigned long long cnt = 0;
        for (int i = 0; i < grid[k]; ++i) {
            mn = std::fmin(mn, h[i].mn);
            mx = std::fmax(mx, h[i].mx);
            cnt += h[i].finite;
        }
        out[k] = cnt < 16 || std::fabs(mx - mn) < 1e-10f;
    }
    *cr = out[0];
    *ct = out[1];
    return AB_OK;
}

// the host's part of correlate_single (:128-141): sigma, confidence, wrap-around, 3-point refinement -- from the PcFin the device leaves
void finish_from_fin(const PcFin &fin, int fr, int fc, double *dx, double *dy, double *conf) {
    const double mean = fin.mean, count = fin.count;
    const double sigma = count >= 1.0 ? std::sqrt(fin.var_sum / (count > 1.0 ? count - 1.0 : 1.0)) : 0.0;
    const int py = fin.best_idx / fc, px = fin.best_idx % fc;
    const double *sv = fin.sv;
    auto refine = [](double center, double prev, double next) {
        const double denom = 2.0 * (2.0 * center - prev - next);
        if (std::fabs(denom) < 1e-15) return 0.0;  // FftFloat::epsilon_val() for f64 is 1e-15 (math/fft.rs)
        const double r = (prev - next) / denom;
        return std::fmin(std::fmax(r, -0.5), 0.5);
    };
    *conf = std::fabs(sigma) < 1e-15 ? 0.0 : (sv[0] - mean) / sigma;                    // normalization.rs:163-168
    const double raw_dy = py > fr / 2 ? (double)py - (double)fr : (double)py;           // subpixel.rs:77-83
    const double raw_dx = px > fc / 2 ? (double)px - (double)fc : (double)px;
    *dy = raw_dy + refine(sv[0], sv[1], sv[2]);
    *dx = raw_dx + refine(sv[0], sv[3], sv[4]);
}

// phase_correlation.rs:105-141
int correlate_single(ab_ctx *ctx, const View &a, const View &b, PcScratch s, int table_set, double *dx, double *dy, double *conf,
                     double *surface_host) {
    const int rows = a.rows, cols = a.cols;
    const int fr = next_pow2(rows), fc = next_pow2(cols), n = fr * fc;
    AB_TRY(pc_tables(ctx, &s, table_set, rows, cols, fr, fc));
    const int g = (n + kBlock - 1) / kBlock;
    hipLaunchKernelGGL(window_pad_kernel, dim3(g), dim3(kBlock), 0, ctx->stream, a.p, rows, cols, a.ld, s.hann_y, s.hann_x, fr, fc, s.fa);
    hipLaunchKernelGGL(window_pad_kernel, dim3(g), dim3(kBlock), 0, ctx->stream, b.p, rows, cols, b.ld, s.hann_y, s.hann_x, fr, fc, s.fb);
    AB_TRY(fft2d(ctx, s.fa, fr, fc, s, 0));
    AB_TRY(fft2d(ctx, s.fb, fr, fc, s, 0));
    hipLaunchKernelGGL(cross_power_kernel, dim3(g), dim3(kBlock), 0, ctx->stream, s.fa, s.fb, n, kEpsilon);
    AB_TRY(fft2d(ctx, s.fa, fr, fc, s, 1));
    const int pg = std::min(kPartials, g);
    hipLaunchKernelGGL(scale_peak_kernel, dim3(pg), dim3(kBlock), 0, ctx->stream, s.fa, n, 1.0 / (double)((size_t)fr * fc), s.corr,
                       (PeakPartial *)s.partials);
    static_assert(kPartials <= 256, "the finish kernels stage the partials in 256 LDS slots");
    hipLaunchKernelGGL(peak_finish_kernel, dim3(1), dim3(256), 0, ctx->stream, (const PeakPartial *)s.partials, pg, s.fin);
    hipLaunchKernelGGL(var_kernel, dim3(pg), dim3(kBlock), 0, ctx->stream, s.corr, n, (const PcFin *)s.fin, s.var_partials);
    hipLaunchKernelGGL(var_finish_kernel, dim3(1), dim3(256), 0, ctx->stream, (const double *)s.var_partials, pg, (const double *)s.corr, fr, fc, s.fin);
    AB_HIP(ctx, hipGetLastError());
    void *pin = nullptr;
    AB_TRY(ab_pinned(ctx, sizeof(PcFin), &pin));
    AB_HIP(ctx, hipMemcpyAsync(pin, s.fin, sizeof(PcFin), hipMemcpyDeviceToHost, ctx->stream));
    if (surface_host) AB_HIP(ctx, hipMemcpyAsync(surface_host, s.corr, (size_t)n * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    AB_HIP(ctx, hipStreamSynchronize(ctx->stream));
    PcFin fin;
    memcpy(&fin, pin, sizeof fin);
    finish_from_fin(fin, fr, fc, dx, dy, conf);
    return AB_OK;
}

int64_t f64_to_i64_sat(double v) {
    if (std::isnan(v)) return 0;
    if (v >= 9223372036854775807.0) return INT64_MAX;
    if (v <= -9223372036854775808.0) return INT64_MIN;
    return (int64_t)v;
}

}  // namespace

// phase_correlation.rs:22-89 on device-resident planes (row strides allowed: crops cost nothing)
int ab_phase_correlate_device(ab_ctx *ctx, const float *ref, int64_t ref_rows, int64_t ref_cols, int64_t ref_ld, const float *tgt,
                              int64_t tgt_rows, int64_t tgt_cols, int64_t tgt_ld, double *dx, double *dy, double *confidence) {
    AB_HIP(ctx, hipSetDevice(ctx->device));
    const int64_t rows = std::min(ref_rows, tgt_rows), cols = std::min(ref_cols, tgt_cols);
    *dx = 0.0;
    *dy = 0.0;
    *confidence = 0.0;
    AB_CHECK(ctx, rows > 0 && cols > 0 && rows < (1 << 30) && cols < (1 << 30), "phase_correlate: bad dims");
    PcScratch s;
    AB_TRY(pc_carve(ctx, &s));
    const View r{ref, (int)rows, (int)cols, ref_ld}, t{tgt, (int)rows, (int)cols, tgt_ld};
    bool cr = false, ct = false;
    AB_TRY(constant_or_zero2(ctx, r, t, s, &cr, &ct));
    if (cr || ct) return AB_OK;                                                    // :42-48
    if (rows <= kCoarseMaxDim && cols <= kCoarseMaxDim) return correlate_single(ctx, r, t, s, 0, dx, dy, confidence, nullptr);  // :50-52
    const double scale_y = (double)rows / (double)kCoarseMaxDim, scale_x = (double)cols / (double)kCoarseMaxDim;
    const int ds_rows = (int)std::min<int64_t>(kCoarseMaxDim, rows), ds_cols = (int)std::min<int64_t>(kCoarseMaxDim, cols);
    // area_downsample's own scale (downsample.rs:13-14) is in/out, which differs from scale_y when a dim <= 512
    const double dsy = (double)rows / (double)ds_rows, dsx = (double)cols / (double)ds_cols;
    const int g = (ds_rows * ds_cols + kBlock - 1) / kBlock;
    hipLaunchKernelGGL(area_downsample_kernel, dim3(g), dim3(kBlock), 0, ctx->stream, r.p, r.rows, r.cols, r.ld, ds_rows, ds_cols,
                       dsy, dsx, s.ds_a);
    hipLaunchKernelGGL(area_downsample_kernel, dim3(g), dim3(kBlock), 0, ctx->stream, t.p, t.rows, t.cols, t.ld, ds_rows, ds_cols,
                       dsy, dsx, s.ds_b);
    AB_HIP(ctx, hipGetLastError());
    double cdx, cdy, cconf;
    AB_TRY(correlate_single(ctx, View{s.ds_a, ds_rows, ds_cols, ds_cols}, View{s.ds_b, ds_rows, ds_cols, ds_cols}, s, 0, &cdx, &cdy,
                            &cconf, nullptr));
    const double coarse_dx = cdx * scale_x, coarse_dy = cdy * scale_y;            // :62-64
    const int64_t half = kRefineCropSize / 2, ref_cy = rows / 2, ref_cx = cols / 2;
    const int64_t tgt_cy = std::min(std::max<int64_t>(f64_to_i64_sat(std::round((double)ref_cy + coarse_dy)), 0), rows - 1);
    const int64_t tgt_cx = std::min(std::max<int64_t>(f64_to_i64_sat(std::round((double)ref_cx + coarse_dx)), 0), cols - 1);
    auto crop = [&](const View &v, int64_t cy, int64_t cx) {                       // extract_crop, :91-103
        const int64_t y0 = cy > half ? cy - half : 0, y1 = std::min(cy + half, rows), x0 = cx > half ? cx - half : 0,
                      x1 = std::min(cx + half, cols);
        return View{v.p + y0 * v.ld + x0, (int)(y1 - y0), (int)(x1 - x0), v.ld};
    };
    const View rc = crop(r, ref_cy, ref_cx), tc = crop(t, tgt_cy, tgt_cx);
    if (rc.rows != tc.rows || rc.cols != tc.cols) {                               // :74-80
        *dx = coarse_dx;
        *dy = coarse_dy;
        *confidence = cconf;
        return AB_OK;
    }
    double rdx, rdy, rconf;
    AB_TRY(correlate_single(ctx, rc, tc, s, 1, &rdx, &rdy, &rconf, nullptr));
    *dx = coarse_dx + rdx;
    *dy = coarse_dy + rdy;
    *confidence = rconf;
    return AB_OK;
}

// ---- phase_correlate(reference, targets[i]) for every i in ONE batch per stage (round 5) ----------------------------
// stack_images(align = true) registers every frame on frame 0 (combine.rs:123-138): the reference's min / max, its downsampled
// plane, its window + coarse spectrum and its crop's spectrum used to be recomputed for every pair, and every pair brought three
// host joins of its own.  Here a stage runs for all pairs at once (blockIdx.y = pair) and ends in one read-back: three joins per
// batch instead of three per pair, the reference's work done once.  Same kernels' arithmetic in the same order: every (dx, dy,
// confidence) equals ab_phase_correlate_device's bit for bit.
namespace {

constexpr size_t kPcBatch = 16;  // pairs per round (4 MiB of spectrum + 2 MiB of surface + 1 MiB of downsampled plane each)

struct PcBatchScratch {
    double2 *fa;       // the reference's spectrum (512 x 512)
    double2 *fb;       // kPcBatch x 512 x 512
    double *corr;      // kPcBatch x 512 x 512
    float *ds;         // (1 + kPcBatch) x 512 x 512: the downsampled reference, then the targets
    MinMaxPartial *mm;  // (1 + kPcBatch) x kMinMaxPartials
    PeakPartial *peak;  // kPcBatch x kPartials
    double *var;        // kPcBatch x kPartials
    PcFin *fin;         // kPcBatch
    PcItem *items;      // 1 + kPcBatch
    unsigned int upload = 0;  // uploads so far: each takes its own 1 KiB slot of the pinned block (two may be pending between joins)
};

int pc_batch_carve(ab_ctx *ctx, PcBatchScratch *s) {
    const size_t n = 512 * 512, B = kPcBatch;
    const size_t bytes = n * sizeof(double2) + B * n * sizeof(double2) + B * n * sizeof(double) + (1 + B) * n * sizeof(float) +
                         (1 + B) * kMinMaxPartials * sizeof(MinMaxPartial) + B * kPartials * sizeof(PeakPartial) + B * kPartials * sizeof(double) +
                         B * sizeof(PcFin) + (1 + B) * sizeof(PcItem) + 1024;
    void *p = nullptr;
    AB_TRY(ab_scratch(ctx, bytes, &p));
    char *c = (char *)p;
    s->fa = (double2 *)c; c += n * sizeof(double2);
    s->fb = (double2 *)c; c += B * n * sizeof(double2);
    s->corr = (double *)c; c += B * n * sizeof(double);
    s->ds = (float *)c; c += (1 + B) * n * sizeof(float);
    s->mm = (MinMaxPartial *)c; c += (1 + B) * kMinMaxPartials * sizeof(MinMaxPartial);
    s->peak = (PeakPartial *)c; c += B * kPartials * sizeof(PeakPartial);
    s->var = (double *)c; c += B * kPartials * sizeof(double);
    s->fin = (PcFin *)c; c += ((B * sizeof(PcFin) + 15) / 16) * 16;
    s->items = (PcItem *)c;
    return AB_OK;
}

// upload `cnt` views as the kernels' item table (through the context's pinned buffer: the host vector is a local)
int pc_upload_items(ab_ctx *ctx, PcBatchScratch &s, const View *v, size_t cnt) {
    void *pin = nullptr;
    AB_TRY(ab_pinned(ctx, 64 * 1024, &pin));  // (also holds the stage's read-back further up: see the offsets below)
    PcItem *h = (PcItem *)((char *)pin + (size_t)(s.upload++ % 16u) * 1024);  // (a slot is reused only many joins later)
    for (size_t i = 0; i < cnt; ++i) h[i] = PcItem{v[i].p, v[i].ld, v[i].rows, v[i].cols};
    AB_HIP(ctx, hipMemcpyAsync(s.items, h, cnt * sizeof(PcItem), hipMemcpyHostToDevice, ctx->stream));
    return AB_OK;
}

// correlate_single(a, b[i]) for i < nb: all views rows x cols (<= 512 x 512)
int correlate_batch(ab_ctx *ctx, const View &a, const View *b, size_t nb, PcBatchScratch &s, int table_set, double *dx, double *dy, double *conf) {
    const int rows = a.rows, cols = a.cols;
    const int fr = next_pow2(rows), fc = next_pow2(cols), n = fr * fc;
    PcScratch t{};
    AB_TRY(pc_tables(ctx, &t, table_set, rows, cols, fr, fc));
    const int g = (n + kBlock - 1) / kBlock;
    std::vector<View> views(b, b + nb);
    AB_TRY(pc_upload_items(ctx, s, views.data(), nb));
    hipLaunchKernelGGL(window_pad_kernel, dim3(g), dim3(kBlock), 0, ctx->stream, a.p, rows, cols, a.ld, t.hann_y, t.hann_x, fr, fc, s.fa);
    hipLaunchKernelGGL(window_pad_many_kernel, dim3(g, (unsigned)nb), dim3(kBlock), 0, ctx->stream, (const PcItem *)s.items, (const double *)t.hann_y, (const double *)t.hann_x, fr, fc,
                       s.fb);
    AB_TRY(fft2d(ctx, s.fa, fr, fc, t, 0));
    auto fft2d_many = [&](int inverse) {
        hipLaunchKernelGGL(fft_lines_many_kernel, dim3(fr, (unsigned)nb), dim3(kBlock), 0, ctx->stream, s.fb, (int64_t)n, fc, ilog2(fc), (int64_t)1, (int64_t)fc,
                           (const double2 *)t.tw_c, inverse);  // rows
        hipLaunchKernelGGL(fft_lines_many_kernel, dim3(fc, (unsigned)nb), dim3(kBlock), 0, ctx->stream, s.fb, (int64_t)n, fr, ilog2(fr), (int64_t)fc, (int64_t)1,
                           (const double2 *)t.tw_r, inverse);  // columns
    };
    fft2d_many(0);
    hipLaunchKernelGGL(cross_power_many_kernel, dim3(g, (unsigned)nb), dim3(kBlock), 0, ctx->stream, (const double2 *)s.fa, s.fb, n, kEpsilon);
    fft2d_many(1);
    const int pg = std::min(kPartials, g);
    hipLaunchKernelGGL(scale_peak_kernel, dim3(pg, (unsigned)nb), dim3(kBlock), 0, ctx->stream, (const double2 *)s.fb, n, 1.0 / (double)((size_t)fr * fc), s.corr, s.peak, kPartials);
    hipLaunchKernelGGL(peak_finish_kernel, dim3((unsigned)nb), dim3(256), 0, ctx->stream, (const PeakPartial *)s.peak, pg, s.fin, kPartials);
    hipLaunchKernelGGL(var_kernel, dim3(pg, (unsigned)nb), dim3(kBlock), 0, ctx->stream, (const double *)s.corr, n, (const PcFin *)s.fin, s.var, kPartials);
    hipLaunchKernelGGL(var_finish_kernel, dim3((unsigned)nb), dim3(256), 0, ctx->stream, (const double *)s.var, pg, (const double *)s.corr, fr, fc, s.fin, kPartials);
    AB_HIP(ctx, hipGetLastError());
    void *pin = nullptr;
    AB_TRY(ab_pinned(ctx, 64 * 1024, &pin));
    PcFin *hf = (PcFin *)((char *)pin + 32 * 1024);
    AB_HIP(ctx, hipMemcpyAsync(hf, s.fin, nb * sizeof(PcFin), hipMemcpyDeviceToHost, ctx->stream));
    AB_HIP(ctx, hipStreamSynchronize(ctx->stream));
    for (size_t i = 0; i < nb; ++i) finish_from_fin(hf[i], fr, fc, &dx[i], &dy[i], &conf[i]);
    return AB_OK;
}

}  // namespace

int ab_phase_correlate_many_device(ab_ctx *ctx, const float *ref, int64_t ref_ld, const float *const *tgts, const int64_t *tgt_ld, size_t n, int64_t rows,
                                   int64_t cols, double *dx, double *dy, double *confidence) {
    AB_HIP(ctx, hipSetDevice(ctx->device));
    for (size_t i = 0; i < n; ++i) dx[i] = dy[i] = confidence[i] = 0.0;
    if (n == 0) return AB_OK;
    AB_CHECK(ctx, rows > 0 && cols > 0 && rows < (1 << 30) && cols < (1 << 30), "phase_correlate: bad dims");
    static_assert(sizeof(PcItem) * (1 + kPcBatch) <= 1024 && sizeof(PcFin) * kPcBatch <= 16 * 1024, "the pinned block's slots");
    PcBatchScratch s;
    AB_TRY(pc_batch_carve(ctx, &s));
    const View r{ref, (int)rows, (int)cols, ref_ld};
    for (size_t base = 0; base < n; base += kPcBatch) {
        const size_t nb = std::min(kPcBatch, n - base);
        std::vector<View> t(nb);
        for (size_t i = 0; i < nb; ++i) t[i] = View{tgts[base + i], (int)rows, (int)cols, tgt_ld[base + i]};
        // ---- is_constant_or_zero (:143-160) of the reference and of every target: one launch, one read-back ----
        std::vector<View> all(1 + nb);
        all[0] = r;
        for (size_t i = 0; i < nb; ++i) all[1 + i] = t[i];
        AB_TRY(pc_upload_items(ctx, s, all.data(), all.size()));
        const int mg = (int)std::min<int64_t>(kMinMaxPartials, rows);
        hipLaunchKernelGGL(minmax_finite_many_kernel, dim3(mg, (unsigned)(1 + nb)), dim3(kBlock), 0, ctx->stream, (const PcItem *)s.items, s.mm, kMinMaxPartials);
        AB_HIP(ctx, hipGetLastError());
        std::vector<MinMaxPartial> hm((1 + nb) * (size_t)mg);
        {
            // (the partials of plane k sit kMinMaxPartials apart: mg of them are copied per plane)
            void *pin = nullptr;
            const size_t need = (1 + nb) * (size_t)kMinMaxPartials * sizeof(MinMaxPartial);
            AB_TRY(ab_pinned(ctx, std::max<size_t>(need, 64 * 1024), &pin));
            AB_HIP(ctx, hipMemcpyAsync(pin, s.mm, need, hipMemcpyDeviceToHost, ctx->stream));
            AB_HIP(ctx, hipStreamSynchronize(ctx->stream));
            for (size_t k = 0; k < 1 + nb; ++k) memcpy(&hm[k * mg], (const MinMaxPartial *)pin + k * kMinMaxPartials, (size_t)mg * sizeof(MinMaxPartial));
        }
        auto degenerate = [&](size_t k) {
            float mn = INFINITY, mx = -INFINITY;
            unsigned long long cnt = 0;
            for (int i = 0; i < mg; ++i) {
                mn = std::fmin(mn, hm[k * mg + i].mn);
                mx = std::fmax(mx, hm[k * mg + i].mx);
                cnt += hm[k * mg + i].finite;
            }
            return cnt < 16 || std::fabs(mx - mn) < 1e-10f;
        };
        if (degenerate(0)) continue;  // :42-44: every pair of this round is (0, 0, 0)
        std::vector<size_t> live;     // indices (within the round) of the pairs that are correlated
        for (size_t i = 0; i < nb; ++i)
            if (!degenerate(1 + i)) live.push_back(i);  // :45-48
        if (live.empty()) continue;
        std::vector<View> lv(live.size());
        std::vector<double> ldx(live.size()), ldy(live.size()), lcf(live.size());
        if (rows <= kCoarseMaxDim && cols <= kCoarseMaxDim) {  // :50-52
            for (size_t k = 0; k < live.size(); ++k) lv[k] = t[live[k]];
            AB_TRY(correlate_batch(ctx, r, lv.data(), lv.size(), s, 0, ldx.data(), ldy.data(), lcf.data()));
            for (size_t k = 0; k < live.size(); ++k) {
                dx[base + live[k]] = ldx[k];
                dy[base + live[k]] = ldy[k];
                confidence[base + live[k]] = lcf[k];
            }
            continue;
        }
        // ---- coarse: area_downsample to <= 512 x 512 (the reference once), correlate (:54-64) ----
        const double scale_y = (double)rows / (double)kCoarseMaxDim, scale_x = (double)cols / (double)kCoarseMaxDim;
        const int ds_rows = (int)std::min<int64_t>(kCoarseMaxDim, rows), ds_cols = (int)std::min<int64_t>(kCoarseMaxDim, cols);
        const double dsy = (double)rows / (double)ds_rows, dsx = (double)cols / (double)ds_cols;
        const size_t dn = (size_t)ds_rows * ds_cols;
        std::vector<View> src(1 + live.size());
        src[0] = r;
        for (size_t k = 0; k < live.size(); ++k) src[1 + k] = t[live[k]];
        AB_TRY(pc_upload_items(ctx, s, src.data(), src.size()));
        hipLaunchKernelGGL(area_downsample_many_kernel, dim3((unsigned)((dn + kBlock - 1) / kBlock), (unsigned)src.size()), dim3(kBlock), 0, ctx->stream, (const PcItem *)s.items, ds_rows,
                           ds_cols, dsy, dsx, s.ds);
        AB_HIP(ctx, hipGetLastError());
        for (size_t k = 0; k < live.size(); ++k) lv[k] = View{s.ds + (1 + k) * dn, ds_rows, ds_cols, ds_cols};
        AB_TRY(correlate_batch(ctx, View{s.ds, ds_rows, ds_cols, ds_cols}, lv.data(), lv.size(), s, 0, ldx.data(), ldy.data(), lcf.data()));
        // ---- refine on centred 512 x 512 crops (:66-88); a pair whose crops differ in size keeps the coarse answer (:74-80) ----
        const int64_t half = kRefineCropSize / 2, ref_cy = rows / 2, ref_cx = cols / 2;
        auto crop = [&](const View &v, int64_t cy, int64_t cx) {  // extract_crop, :91-103
            const int64_t y0 = cy > half ? cy - half : 0, y1 = std::min(cy + half, rows), x0 = cx > half ? cx - half : 0, x1 = std::min(cx + half, cols);
            return View{v.p + y0 * v.ld + x0, (int)(y1 - y0), (int)(x1 - x0), v.ld};
        };
        const View rc = crop(r, ref_cy, ref_cx);
        std::vector<size_t> fine;  // positions in `live`
        std::vector<View> fv;
        std::vector<double> cdx(live.size()), cdy(live.size());
        for (size_t k = 0; k < live.size(); ++k) {
            cdx[k] = ldx[k] * scale_x;
            cdy[k] = ldy[k] * scale_y;
            const int64_t tgt_cy = std::min(std::max<int64_t>(f64_to_i64_sat(std::round((double)ref_cy + cdy[k])), 0), rows - 1);
            const int64_t tgt_cx = std::min(std::max<int64_t>(f64_to_i64_sat(std::round((double)ref_cx + cdx[k])), 0), cols - 1);
            const View tc = crop(t[live[k]], tgt_cy, tgt_cx);
            if (tc.rows != rc.rows || tc.cols != rc.cols) {
                dx[base + live[k]] = cdx[k];
                dy[base + live[k]] = cdy[k];
                confidence[base + live[k]] = lcf[k];
            } else {
                fine.push_back(k);
                fv.push_back(tc);
            }
        }
        if (!fine.empty()) {
            std::vector<double> rdx(fine.size()), rdy(fine.size()), rcf(fine.size());
            AB_TRY(correlate_batch(ctx, rc, fv.data(), fv.size(), s, 1, rdx.data(), rdy.data(), rcf.data()));
            for (size_t j = 0; j < fine.size(); ++j) {
                const size_t k = fine[j];
                dx[base + live[k]] = cdx[k] + rdx[j];
                dy[base + live[k]] = cdy[k] + rdy[j];
                confidence[base + live[k]] = rcf[j];
            }
        }
    }
    return AB_OK;
}

extern "C" {

int ab_phase_correlate(ab_ctx *ctx, const ab_plane *reference, const ab_plane *target, ab_phase_correlation_result *out) try {
    if (!ctx) return AB_ERR_INVALID;
    AB_CHECK(ctx, reference && target && out, "null argument");
    StagedPlane r, t;
    AB_TRY(ab_stage_in(ctx, reference, &r));
    int rc = ab_stage_in(ctx, target, &t);
    if (rc == AB_OK) {
        rc = ab_phase_correlate_device(ctx, r.dptr, r.rows, r.cols, r.cols, t.dptr, t.rows, t.cols, t.cols, &out->dx, &out->dy,
                                       &out->confidence);
        ab_stage_release(ctx, &t);
    }
    ab_stage_release(ctx, &r);
    return rc;
} AB_CATCH(ctx)

// test hook: correlate_single's full correlation surface (dims <= 512) for bit-level parity
int ab_correlate_single(ab_ctx *ctx, const ab_plane *a, const ab_plane *b, ab_phase_correlation_result *out, double *surface_host) try {
    if (!ctx) return AB_ERR_INVALID;
    AB_CHECK(ctx, a && b && out, "null argument");
    AB_CHECK(ctx, a->rows == b->rows && a->cols == b->cols && a->rows <= kCoarseMaxDim && a->cols <= kCoarseMaxDim,
             "correlate_single takes two equal planes of at most 512 x 512");
    AB_HIP(ctx, hipSetDevice(ctx->device));
    StagedPlane sa, sb;
    AB_TRY(ab_stage_in(ctx, a, &sa));
    int rc = ab_stage_in(ctx, b, &sb);
    if (rc == AB_OK) {
        PcScratch s;
        rc = pc_carve(ctx, &s);
        if (rc == AB_OK)
            rc = correlate_single(ctx, View{sa.dptr, (int)sa.rows, (int)sa.cols, sa.cols}, View{sb.dptr, (int)sb.rows, (int)sb.cols, sb.cols},
                                  s, 0, &out->dx, &out->dy, &out->confidence, surface_host);
        ab_stage_release(ctx, &sb);
    }
    ab_stage_release(ctx, &sa);
    return rc;
} AB_CATCH(ctx)

}  // extern "C"
