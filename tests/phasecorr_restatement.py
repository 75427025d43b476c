"""An INDEPENDENT restatement of the reference's phase correlation, in numpy, written from the Rust and from nothing else:

    core/alignment/phase_correlation.rs:22-168   phase_correlate, extract_crop, correlate_single, is_constant_or_zero
    core/alignment/downsample.rs:6-48            area_downsample
    math/fft.rs:150-167, 202-229, 250-252, 272-282   inverse_2d's 1/(rows cols), prepare_windowed_buffer, extract_real, find_peak
    math/window.rs:3-18                          hann_periodic
    math/complex.rs:16-45                        safe_normalize, cross_power_element (A conj(B), normalised)
    math/normalization.rs:128-170                compute_mean_sigma (n - 1), compute_snr
    math/subpixel.rs:27-100                      quadratic_refine_1d (circular neighbours), unwrap_circular_peak, unwrap_and_refine

It shares no code with oracle/orc_phasecorr.c (whose FFT is a hand-written radix-2; this one uses numpy's) and is the only pin
this function can have: the reference's own three tests for it (phase_correlation.rs:205-220, align.rs:216-223, pair.rs:126-156)
assert values its code does not produce -- see tests/test_oracle_phasecorr_cases.py.  The oracle and the HIP path are both held
to this file to 1e-6 px in dx / dy (SURVEY 8c allows that much for FFT-derived shifts: the two FFTs round differently at 1e-13).
"""
import numpy as np

COARSE_MAX_DIM = 512      # phase_correlation.rs:10-13
REFINE_CROP_SIZE = 512
EPSILON = 1e-15


def _is_constant_or_zero(img):                                   # :143-162 (f32 min / max over the finite pixels)
    fin = img[np.isfinite(img)]
    if fin.size < 16:
        return True
    return abs(np.float32(fin.max()) - np.float32(fin.min())) < np.float32(1e-10)


def hann_periodic(n):                                            # window.rs:3-18
    if n == 0:
        return np.zeros(0)
    if n == 1:
        return np.ones(1)
    return 0.5 * (1.0 - np.cos(2.0 * np.pi * np.arange(n, dtype=np.float64) / float(n)))


def _next_pow2(n):
    p = 1
    while p < n:
        p *= 2
    return p


def _windowed(img, wy, wx, fr, fc):                              # fft.rs:202-229
    rows, cols = img.shape
    buf = np.zeros((fr, fc), np.complex128)
    v = img.astype(np.float64)
    w = v * wy[:, None] * wx[None, :]
    buf[:rows, :cols] = np.where(np.isfinite(v), w, 0.0)
    return buf


def _refine_1d(surface, peak_y, peak_x, axis_y):                 # subpixel.rs:27-63
    rows, cols = surface.shape
    c = surface[peak_y, peak_x]
    if axis_y:
        p = surface[rows - 1 if peak_y == 0 else peak_y - 1, peak_x]
        n = surface[0 if peak_y == rows - 1 else peak_y + 1, peak_x]
    else:
        p = surface[peak_y, cols - 1 if peak_x == 0 else peak_x - 1]
        n = surface[peak_y, 0 if peak_x == cols - 1 else peak_x + 1]
    denom = 2.0 * (2.0 * c - p - n)
    if abs(denom) < 1e-15:
        return 0.0
    return min(max((p - n) / denom, -0.5), 0.5)


def _unwrap(peak, size):                                         # subpixel.rs:78-84
    return float(peak) - float(size) if peak > size // 2 else float(peak)


def correlate_single(a, b):                                      # phase_correlation.rs:104-141
    rows, cols = a.shape
    fr, fc = _next_pow2(rows), _next_pow2(cols)
    wy, wx = hann_periodic(rows), hann_periodic(cols)
    fa = np.fft.fft2(_windowed(a, wy, wx, fr, fc))
    fb = np.fft.fft2(_windowed(b, wy, wx, fr, fc))
    prod = fa * np.conj(fb)                                      # complex.rs:27-33: (a.re b.re + a.im b.im, a.im b.re - a.re b.im)
    mag = np.abs(prod)
    cross = np.where(mag > EPSILON, prod / np.where(mag > EPSILON, mag, 1.0), 0.0)
    corr = np.fft.ifft2(cross).real                              # inverse_2d divides by rows * cols (fft.rs:162-166)
    flat = int(np.argmax(corr))                                  # find_peak: the first maximum (strict > in the fold)
    py, px = divmod(flat, fc)
    fin = corr[np.isfinite(corr)]
    mean = fin.sum() / fin.size
    sigma = np.sqrt(((fin - mean) ** 2).sum() / (fin.size - 1 if fin.size > 1 else 1))
    conf = 0.0 if abs(sigma) < 1e-15 else (corr[py, px] - mean) / sigma
    dy = _unwrap(py, fr) + _refine_1d(corr, py, px, True)
    dx = _unwrap(px, fc) + _refine_1d(corr, py, px, False)
    return dx, dy, conf


def area_downsample(img, out_rows, out_cols):                    # downsample.rs:6-48
    in_rows, in_cols = img.shape
    if (in_rows, in_cols) == (out_rows, out_cols):
        return img.copy()
    sy, sx = in_rows / out_rows, in_cols / out_cols
    out = np.zeros((out_rows, out_cols), np.float32)

    def edges(n_out, scale, n_in):
        lo = np.clip(np.floor(np.arange(n_out) * scale).astype(np.int64), 0, n_in - 1)
        hi = np.minimum(np.maximum(np.ceil((np.arange(n_out) + 1) * scale).astype(np.int64), 0), n_in)
        return lo, hi
    y0, y1 = edges(out_rows, sy, in_rows)
    x0, x1 = edges(out_cols, sx, in_cols)
    v = img.astype(np.float64)
    fin = np.isfinite(v)
    for oy in range(out_rows):
        rows_v, rows_f = v[y0[oy]:y1[oy]], fin[y0[oy]:y1[oy]]
        for ox in range(out_cols):
            blk, ok = rows_v[:, x0[ox]:x1[ox]], rows_f[:, x0[ox]:x1[ox]]
            cnt = int(ok.sum())
            # (the reference adds the block's pixels one by one in raster order; np.sum's pairwise order differs in the last bits
            # of an f64 sum that is then rounded to f32 -- the result is the same f32 except on exact rounding ties)
            out[oy, ox] = np.float32(blk[ok].sum() / cnt) if cnt else np.float32(0.0)
    return out


def _crop(img, cy, cx, half):                                    # phase_correlation.rs:89-102
    rows, cols = img.shape
    return img[max(cy - half, 0):min(cy + half, rows), max(cx - half, 0):min(cx + half, cols)]


def _round_half_away(x):                                         # f64::round
    return int(np.floor(abs(x) + 0.5) * (1 if x >= 0 else -1))


def phase_correlate(reference, target):                          # phase_correlation.rs:22-87
    rows, cols = min(reference.shape[0], target.shape[0]), min(reference.shape[1], target.shape[1])
    ref, tgt = np.ascontiguousarray(reference[:rows, :cols], np.float32), np.ascontiguousarray(target[:rows, :cols], np.float32)
    if _is_constant_or_zero(ref) or _is_constant_or_zero(tgt):
        return 0.0, 0.0, 0.0
    if rows <= COARSE_MAX_DIM and cols <= COARSE_MAX_DIM:
        return correlate_single(ref, tgt)
    scale_y, scale_x = rows / COARSE_MAX_DIM, cols / COARSE_MAX_DIM
    ds_rows, ds_cols = min(COARSE_MAX_DIM, rows), min(COARSE_MAX_DIM, cols)
    cdx, cdy, cconf = correlate_single(area_downsample(ref, ds_rows, ds_cols), area_downsample(tgt, ds_rows, ds_cols))
    coarse_dx, coarse_dy = cdx * scale_x, cdy * scale_y
    half = REFINE_CROP_SIZE // 2
    ref_cy, ref_cx = rows // 2, cols // 2
    tgt_cy = min(max(_round_half_away(ref_cy + coarse_dy), 0), rows - 1)
    tgt_cx = min(max(_round_half_away(ref_cx + coarse_dx), 0), cols - 1)
    ref_crop, tgt_crop = _crop(ref, ref_cy, ref_cx, half), _crop(tgt, tgt_cy, tgt_cx, half)
    if ref_crop.shape != tgt_crop.shape:
        return coarse_dx, coarse_dy, cconf
    rdx, rdy, rconf = correlate_single(np.ascontiguousarray(ref_crop), np.ascontiguousarray(tgt_crop))
    return coarse_dx + rdx, coarse_dy + rdy, rconf
