"""Test helper (not product code): compute_image_stats' histogram path (core/imaging/stats.rs:75-210) as the pass /
bookkeeping protocol the row-band sharded implementation follows -- every pass returns INTEGER partials of one band, an
injected `allreduce` joins the bands, and the scalar bookkeeping runs on the joined values.

It is a second, independent restatement of stats.rs in numpy (the first is oracle/orc_stats.c): the reference has no test
for this file, so tests pin the two restatements against each other, and both against libastroburst_hip.so.
"""
import math

import numpy as np

HIST_BINS = 65536            # stats.rs:8
PADDING_THRESHOLD = np.float32(1e-7)   # types/constants.rs:6
MAD_TO_SIGMA = 1.4826        # types/constants.rs:7


def valid(v):  # stats.rs:10-13
    return np.isfinite(v) & (v > PADDING_THRESHOLD)


def _bin(t, last):
    """Rust `f64 as usize` then `.min(last)`: truncation, saturating, NaN -> 0"""
    t = np.where(np.isnan(t), 0.0, t)
    return np.clip(t, 0.0, float(last)).astype(np.int64)


def scan_pass(band):  # stats.rs:212-240 (scan_minmax)
    v = band[valid(band)].astype(np.float64)
    if v.size == 0:
        return np.array([-np.finfo(np.float64).max, -np.finfo(np.float64).max])   # {-min, max} with min = f64::MAX
    return np.array([-v.min(), v.max()])


def value_pass(band, gmin, inv):  # stats.rs:260-300
    v = band[valid(band)].astype(np.float64)
    hist = np.bincount(_bin((v - gmin) * inv, HIST_BINS - 1), minlength=HIST_BINS).astype(np.int64)
    return hist, float(v.sum()), int(v.size)


def dev_pass(band, coarse_med_f32, dev_inv, lo, hi, refine_inv):  # stats.rs:119-146
    v32 = band[valid(band)]
    vf = v32.astype(np.float64)
    m = (vf >= lo) & (vf < hi)
    refine = np.bincount(_bin((vf[m] - lo) * refine_inv, HIST_BINS - 1), minlength=HIST_BINS).astype(np.int64)
    d = np.abs(v32 - np.float32(coarse_med_f32)).astype(np.float64)
    dev = np.bincount(_bin(d * dev_inv, HIST_BINS - 1), minlength=HIST_BINS).astype(np.int64)
    return refine, dev


def mad_pass(band, med_f32, lo_f32, hi_f32, region_lo, inv):  # stats.rs:166-191
    v32 = band[valid(band)]
    dev = np.abs(v32 - np.float32(med_f32))
    below = int((dev < np.float32(lo_f32)).sum())
    m = (dev >= np.float32(lo_f32)) & (dev < np.float32(hi_f32))
    h = np.bincount(_bin((dev[m].astype(np.float64) - region_lo) * inv, HIST_BINS - 1), minlength=HIST_BINS).astype(np.int64)
    return below, h


def find_percentile_bin(hist, total, pct):  # stats.rs:302-311
    target = int(math.ceil(total * pct))
    cum = np.cumsum(hist)
    idx = np.nonzero(cum >= target)[0]
    return int(idx[0]) if idx.size else len(hist) - 1


def interpolate_percentile(hist, total, pct, data_min, bin_width):  # stats.rs:313-331
    target = int(math.ceil(total * pct))
    cum = 0
    for i in np.nonzero(hist)[0] if target > 0 else range(len(hist)):
        count = int(hist[i])
        cum += count
        if cum >= target:
            frac = 1.0 - ((cum - target) / count) if count > 0 else 0.5
            return data_min + (i + frac) * bin_width
    return data_min + len(hist) * bin_width


def resolve_rank_in_hist(hist, rank, region_lo, sub_bw):  # stats.rs:333-353
    if rank == 0:
        return region_lo
    cum = 0
    for i in np.nonzero(hist)[0]:
        count = int(hist[i])
        cum += count
        if cum >= rank:
            return region_lo + (i + (1.0 - ((cum - rank) / count))) * sub_bw
    return region_lo + len(hist) * sub_bw


def stats_hist_sharded(band, allreduce_sum, allreduce_max):
    """compute_image_stats_hist of the image whose row bands are spread over the ranks; band = this rank's rows (f32).
    allreduce_sum(int64 / float64 ndarray) and allreduce_max(float64 ndarray) return the joined arrays."""
    band = np.ascontiguousarray(band, dtype=np.float32)
    nm = allreduce_max(scan_pass(band))
    gmin, gmax = -nm[0], nm[1]
    zero = dict(min=0.0, max=0.0, median=0.0, mad=0.0, sigma=0.0, mean=0.0, valid_count=0)
    if gmin == np.finfo(np.float64).max:
        return zero
    rng = max(gmax - gmin, 1e-30)
    bin_width = rng / HIST_BINS
    inv = HIST_BINS / rng
    hist, s, c = value_pass(band, gmin, inv)
    joined = allreduce_sum(np.concatenate([hist, np.array([c], np.int64)]))
    hist, total = joined[:-1], int(joined[-1])
    s = float(allreduce_sum(np.array([s], np.float64))[0])
    if total == 0:
        return zero
    mean = s / total
    half = int(math.ceil(total * 0.5))
    median_bin = find_percentile_bin(hist, total, 0.5)
    before = int(hist[:median_bin].sum())
    lo = gmin + median_bin * bin_width
    hi = lo + bin_width
    coarse = interpolate_percentile(hist, total, 0.5, gmin, bin_width)
    dev_bw, dev_inv = rng / HIST_BINS, HIST_BINS / rng
    refine_range = max(hi - lo, 1e-30)
    refine_inv = HIST_BINS / refine_range
    refine, dev = dev_pass(band, np.float32(coarse), dev_inv, lo, hi, refine_inv)
    joined = allreduce_sum(np.concatenate([refine, dev]))
    refine, dev = joined[:HIST_BINS], joined[HIST_BINS:]
    median = resolve_rank_in_hist(refine, max(half - before, 0), lo, refine_range / HIST_BINS)
    mad_bin = find_percentile_bin(dev, total, 0.5)
    e_lo, e_hi = max(mad_bin - 1, 0), min(mad_bin + 2, HIST_BINS)
    r_lo, r_hi = e_lo * dev_bw, e_hi * dev_bw
    mad_range = max(r_hi - r_lo, 1e-30)
    below, h = mad_pass(band, np.float32(median), np.float32(r_lo), np.float32(r_hi), r_lo, HIST_BINS / mad_range)
    joined = allreduce_sum(np.concatenate([h, np.array([below], np.int64)]))
    h, below = joined[:-1], int(joined[-1])
    mad = resolve_rank_in_hist(h, max(half - below, 0), r_lo, mad_range / HIST_BINS)
    return dict(min=gmin, max=gmax, median=median, mad=mad, sigma=max(mad * MAD_TO_SIGMA, 1e-30), mean=mean, valid_count=total)
