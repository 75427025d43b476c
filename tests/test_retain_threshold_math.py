"""The batch stack's retain test without its division (csrc/batch_pipeline.hip, retain_thresholds): the claim
    RN32(d / sigma) < h   <=>   d < m * sigma,   m = (pred(h) + h) / 2      (a tie d = m sigma cannot occur: 25 + bits vs 24)
and its mirror image for the lower bound, checked in numpy on values chosen to sit ON and one ulp either side of every
boundary -- the arithmetic the kernel relies on, restated here (f32 division in numpy is IEEE)."""
import numpy as np


def thresholds(nsl, sh):
    nsl, sh = np.float32(nsl), np.float32(sh)
    mid_hi = (np.float64(np.nextafter(sh, np.float32(-np.inf))) + np.float64(sh)) * 0.5
    mid_lo = (np.float64(nsl) + np.float64(np.nextafter(nsl, np.float32(np.inf)))) * 0.5
    return mid_lo, mid_hi


def test_threshold_compare_equals_the_division():
    rng = np.random.default_rng(3)
    bounds = [(-2.5, 3.0), (-1.0, 1.0), (-0.7, 0.7), (-2.9999998, 2.9999998), (-1.7000001, 5.0000005), (-1e-30, 1e-30), (-1e30, 1e30),
              (2.0, 3.0), (-3.0, -2.0), (-4.0, 4.0), (-0.5, 8.0)]
    checked = ties = 0
    for nsl, sh in bounds:
        mid_lo, mid_hi = thresholds(nsl, sh)
        sig = np.concatenate([np.float32(10.0) ** rng.uniform(-9.9, 30.0, 4000).astype(np.float32), np.float32([1e-10, 1.0, 3.0, 0.1, 2.0 ** 100])])
        for mid in (mid_lo, mid_hi):
            t = mid * sig.astype(np.float64)                       # exact: 25 x 24 bits
            with np.errstate(over="ignore"):
                base = t.astype(np.float32)
            base = base[np.isfinite(base)]
            s = sig[: base.size] if base.size == sig.size else sig[np.isfinite((mid * sig.astype(np.float64)).astype(np.float32))]
            cand = [base]
            for _ in range(3):
                cand.append(np.nextafter(cand[-1], np.float32(np.inf)))
            lowc = base
            for _ in range(3):
                lowc = np.nextafter(lowc, np.float32(-np.inf))
                cand.append(lowc)
            for d in cand:
                with np.errstate(over="ignore", under="ignore", invalid="ignore"):
                    z = d / s                                      # f32 / f32: RN32(d / sigma)
                want = (z > np.float32(nsl)) & (z < np.float32(sh))
                d64, s64 = d.astype(np.float64), s.astype(np.float64)
                lo_ok, hi_ok = d64 > mid_lo * s64, d64 < mid_hi * s64
                assert np.array_equal(want, lo_ok & hi_ok), (nsl, sh)
                checked += d.size
                ties += int(np.count_nonzero((d64 == mid_lo * s64) | (d64 == mid_hi * s64)))
        # and far from the boundaries
        d = (rng.normal(0, 3, 20000) * 1.0).astype(np.float32)
        s = np.float32(10.0) ** rng.uniform(-2, 2, 20000).astype(np.float32)
        z = d / s
        d64, s64 = d.astype(np.float64), s.astype(np.float64)
        assert np.array_equal((z > np.float32(nsl)) & (z < np.float32(sh)), (d64 > mid_lo * s64) & (d64 < mid_hi * s64))
    assert checked > 500000 and ties == 0   # m sigma is never an f32 (m is an odd 25-bit integer times a power of two)
