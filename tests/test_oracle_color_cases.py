"""Pin the colour/tone/calibration part of the CPU oracle against the reference's own unit tests
(scnr.rs:63-102, curves.rs:214-277, stretch.rs tests, calibration.rs:344-410)."""
import numpy as np


def test_scnr_removes_dominant_green(oracle):               # scnr.rs:64-73
    r, g, b = (np.full((2, 2), v, np.float32) for v in (0.3, 0.9, 0.3))
    r2, g2, b2 = oracle.apply_scnr(r, g, b, "average", 1.0, False)
    assert abs(g2[0, 0] - 0.3) < 1e-5 and abs(r2[0, 0] - 0.3) < 1e-5 and abs(b2[0, 0] - 0.3) < 1e-5


def test_scnr_preserve_skips_saturated(oracle):             # scnr.rs:75-83
    r, g, b = (np.full((1, 1), v, np.float32) for v in (2.5, 1.8, 1.2))
    r2, _, b2 = oracle.apply_scnr(r, g, b, "maximum", 1.0, True)
    assert abs(r2[0, 0] - 2.5) < 1e-5 and abs(b2[0, 0] - 1.2) < 1e-5


def test_scnr_preserve_boosts(oracle):                      # scnr.rs:85-94
    r, g, b = (np.full((1, 1), v, np.float32) for v in (0.2, 0.6, 0.2))
    r2, g2, b2 = oracle.apply_scnr(r, g, b, "average", 1.0, True)
    assert r2[0, 0] > 0.2 and b2[0, 0] > 0.2 and abs(g2[0, 0] - 0.2) < 1e-5


def test_scnr_amount_zero(oracle):                          # scnr.rs:96-102
    r, g, b = (np.full((1, 1), v, np.float32) for v in (0.3, 0.9, 0.3))
    _, g2, _ = oracle.apply_scnr(r, g, b, "average", 0.0, True)
    assert abs(g2[0, 0] - 0.9) < 1e-5


def test_levels_identity_clip_gamma(oracle):                # curves.rs:214-241
    data = np.add.outer(np.arange(10), np.arange(10)).astype(np.float32) / np.float32(20.0)
    assert np.all(np.abs(oracle.apply_levels(data) - data) < 1e-6)
    res = oracle.apply_levels(np.array([[0.0, 0.1, 0.5, 1.0]], np.float32), 0.2, 1.0, 1.0)
    assert res[0, 0] == 0.0 and res[0, 1] == 0.0 and 0.0 < res[0, 2] < 1.0 and abs(res[0, 3] - 1.0) < 1e-4
    half = np.array([[0.5]], np.float32)
    assert oracle.apply_levels(half, 0.0, 2.0, 1.0)[0, 0] > 0.5 > oracle.apply_levels(half, 0.0, 0.5, 1.0)[0, 0]


def test_spline_identity_scurve_monotonic(oracle):          # curves.rs:243-277
    lut = oracle.spline_lut_from_points([(0.0, 0.0), (1.0, 1.0)])
    v = (np.arange(101, dtype=np.float32) / np.float32(100.0)).reshape(1, -1)
    assert np.all(np.abs(oracle.apply_curve(v, lut) - v) < 0.01)
    lut = oracle.spline_lut_from_points([(0.0, 0.0), (0.25, 0.15), (0.5, 0.5), (0.75, 0.85), (1.0, 1.0)])
    f = lambda x: float(oracle.apply_curve(np.array([[x]], np.float32), lut)[0, 0])
    assert f(0.0) < 0.01 and abs(f(1.0) - 1.0) < 0.01 and f(0.25) < 0.25 and f(0.75) > 0.75
    lut = oracle.spline_lut_from_points([(0.0, 0.0), (0.3, 0.1), (0.5, 0.5), (0.7, 0.9), (1.0, 1.0)])
    out = oracle.apply_curve((np.arange(4096, dtype=np.float32) / np.float32(4095.0)).reshape(1, -1), lut).ravel()
    assert np.all(np.diff(out) >= -1e-6)


def test_spline_unsorted_duplicate_and_missing_endpoints(oracle):
    a = oracle.spline_lut_from_points([(0.5, 0.7), (0.2, 0.1), (0.5 + 1e-10, 0.9)])   # sorted, deduped, (0,0),(1,1) added
    b = oracle.spline_lut_from_points([(0.0, 0.0), (0.2, 0.1), (0.5, 0.7), (1.0, 1.0)])
    assert np.array_equal(a, b)
    assert oracle.spline_lut_from_points([])[0] == 0.0 and oracle.spline_lut_from_points([])[4095] == 1.0


def test_arcsinh_basic(oracle):                             # stretch.rs tests: range, monotone, zero factor, flat
    data = np.linspace(0, 1000, 64, dtype=np.float32).reshape(8, 8)
    out = oracle.arcsinh_stretch_with_stats(data, 0.0, 1000.0, 10.0)
    assert out.min() >= 0.0 and out.max() <= 1.0 + 1e-6 and np.all(np.diff(out.ravel()) >= 0)
    assert np.array_equal(oracle.arcsinh_stretch_with_stats(data, 0.0, 1000.0, 0.0), data)
    assert np.all(oracle.arcsinh_stretch_with_stats(np.full((4, 4), 5.0, np.float32), 5.0, 5.0, 10.0) == 0.0)
    nan = data.copy()
    nan[0, 0] = np.nan
    assert oracle.arcsinh_stretch_with_stats(nan, 0.0, 1000.0, 10.0)[0, 0] == 0.0


def test_calibrate_and_median_combine(oracle):              # calibration.rs:344-410
    raw = np.full((4, 4), 1100.0, np.float32)
    bias = np.full((4, 4), 100.0, np.float32)
    dark = np.full((4, 4), 50.0, np.float32)
    flat = np.full((4, 4), 2.0, np.float32)
    assert np.allclose(oracle.calibrate_image(raw, bias), 1000.0)
    assert np.allclose(oracle.calibrate_image(raw, bias, dark, None, 2.0), 900.0)
    assert np.allclose(oracle.calibrate_image(raw, bias, dark, flat, 2.0), 450.0)
    flat[1, 1] = 0.0                                         # |flat| <= 1e-4 -> no division
    assert oracle.calibrate_image(raw, None, None, flat)[1, 1] == 1100.0
    assert oracle.calibrate_image(bias, raw)[0, 0] == 0.0    # negative clamps to 0
    frames = [np.full((2, 2), v, np.float32) for v in (1.0, 5.0, 3.0, 100.0)]
    assert np.all(oracle.median_combine(frames) == 5.0)      # upper median of an even count
    frames[1][0, 0] = np.nan
    assert oracle.median_combine(frames)[0, 0] == 3.0
    assert oracle.median_combine([np.full((1, 1), np.nan, np.float32)])[0, 0] == 0.0
