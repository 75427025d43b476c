// Workgroup-level exact order statistics over a rectangular window of a plane (gfx950).
//
// Used wherever the reference takes a median / MAD of up to 256 x 256 pixels with
// select_nth_unstable: the detection background tiles (core/analysis/star_detection.rs:47-68) and the
// background-extraction grid cells (core/imaging/background.rs:151-190).  All candidate pixels are
// positive finite floats (or absolute deviations), whose IEEE bit patterns are monotone as u32, so
// rank k is found by an 11/11/10-bit radix select: three LDS-histogram passes over the window
// (re-read from L2), one 1024-thread workgroup per window.  Exact for every rank -> medians of
// even counts average the two middle order statistics just as math/median.rs does.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

namespace absel {

constexpr int kBlock = 1024;

struct Window {
    const float *img;
    int64_t ld;
    int y0, y1, x0, x1;
    float min_valid;  // pixel is a candidate iff finite and > min_valid
    float lo, hi;     // and lo <= v <= hi (cumulative `retain` bounds; +-inf when unused)
};

__device__ __forceinline__ bool candidate(const Window &w, float v) {
    return __builtin_isfinite(v) && v > w.min_valid && v >= w.lo && v <= w.hi;
}

// histogram over bits [shift, shift+nbits) of the keys matching the prefix.
// mode 0: key = bits(v); mode 1: key = bits((f32)|(f64)v - center_f64|); mode 2: key = bits(|v - center_f32|)
__device__ inline void window_hist(const Window &w, int mode, double center64, float center32, uint32_t prefix_mask,
                                   uint32_t prefix_val, int shift, int nbits, unsigned int *hist /* LDS, 2048 */) {
    const int nb = 1 << nbits;
    for (int i = threadIdx.x; i < nb; i += kBlock) hist[i] = 0;
    __syncthreads();
    const int ww = w.x1 - w.x0, n = ww * (w.y1 - w.y0);
    for (int i = threadIdx.x; i < n; i += kBlock) {
        const int r = w.y0 + i / ww, c = w.x0 + i % ww;
        const float v = w.img[r * w.ld + c];
        if (candidate(w, v)) {
            float k = v;
            if (mode == 1) k = (float)fabs((double)v - center64);
            if (mode == 2) k = fabsf(v - center32);
            const uint32_t key = __float_as_uint(k);
            if ((key & prefix_mask) == prefix_val) atomicAdd(&hist[(key >> shift) & (nb - 1)], 1u);
        }
    }
    __syncthreads();
}

// bin holding 0-based `rank`, the count before it, and the histogram total (broadcast to all threads)
__device__ inline void find_bin(const unsigned int *hist, int nb, unsigned int rank, unsigned int *bin_out,
                                unsigned int *before_out, unsigned int *total_out) {
    __shared__ unsigned int s_bin, s_before, s_total;
    if (threadIdx.x == 0) {
        unsigned int cum = 0, bin = nb - 1, before = 0;
        bool found = false;
        for (int i = 0; i < nb; ++i) {
            const unsigned int h = hist[i];
            if (!found && cum + h > rank) {
                bin = i;
                before = cum;
                found = true;
            }
            cum += h;
        }
        s_bin = bin;
        s_before = before;
        s_total = cum;
    }
    __syncthreads();
    *bin_out = s_bin;
    *before_out = s_before;
    *total_out = s_total;
    __syncthreads();
}

__device__ inline unsigned int count(const Window &w, unsigned int *hist) {
    window_hist(w, 0, 0.0, 0.0f, 0, 0, 21, 11, hist);
    unsigned int b, bf, n;
    find_bin(hist, 2048, 0xffffffffu, &b, &bf, &n);
    return n;
}

// the rank-th smallest key (0-based) of the window's candidates
__device__ inline float select(const Window &w, int mode, double center64, float center32, unsigned int rank,
                               unsigned int *hist) {
    const int shifts[3] = {21, 10, 0}, bits[3] = {11, 11, 10};
    uint32_t mask = 0, val = 0;
    for (int p = 0; p < 3; ++p) {
        window_hist(w, mode, center64, center32, mask, val, shifts[p], bits[p], hist);
        unsigned int bin, before, total;
        find_bin(hist, 1 << bits[p], rank, &bin, &before, &total);
        rank -= before;
        val |= bin << shifts[p];
        mask |= ((1u << bits[p]) - 1u) << shifts[p];
    }
    return __uint_as_float(val);
}

// median_f32_mut (math/median.rs:46-63) of the candidates (n > 0): f32 average of the two middle values
__device__ inline float median_f32(const Window &w, int mode, double center64, float center32, unsigned int n,
                                   unsigned int *hist) {
    const unsigned int mid = n / 2;
    const float right = select(w, mode, center64, center32, mid, hist);
    if (n % 2 == 0) {
        const float left = select(w, mode, center64, center32, mid - 1, hist);
        return (left + right) / 2.0f;
    }
    return right;
}

// exact_median_mut (math/median.rs:27-44): f64 average of the two middle values
__device__ inline double exact_median(const Window &w, unsigned int n, unsigned int *hist) {
    const unsigned int mid = n / 2;
    const float right = select(w, 0, 0.0, 0.0f, mid, hist);
    if (n % 2 == 0) {
        const float left = select(w, 0, 0.0, 0.0f, mid - 1, hist);
        return ((double)left + (double)right) / 2.0;
    }
    return (double)right;
}

}  // namespace absel
