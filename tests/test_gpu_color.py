"""GPU parity for the elementwise colour / tone / calibration maps vs the CPU oracle.

Bar: f32 maps built from +, -, *, /, min, max, LUT lookups: bit-exact.  Maps through libm
(pow in apply_levels, asinh/pow in arcsinh_stretch): <= 1e-6 relative (the reference's own Rust
std implementations differ from any C libm at the ulp level; the north-star contract is 1e-5)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def rgb(rng, rows=123, cols=257, lo=-0.1, hi=1.6):
    planes = [rng.uniform(lo, hi, (rows, cols)).astype(np.float32) for _ in range(3)]
    planes[1][5, 7:20] = np.nan
    planes[0][9, 3] = np.inf
    planes[2][11, 11] = -np.inf
    return planes


@pytest.mark.parametrize("method", ["average", "maximum"])
@pytest.mark.parametrize("amount,preserve", [(1.0, False), (0.6, True), (1.0, True), (0.0, True), (5.0, False), (1e-8, True)])
def test_scnr(ctx, oracle, method, amount, preserve):
    r, g, b = rgb(np.random.default_rng(1))
    want = oracle.apply_scnr(r, g, b, method, amount, preserve)
    r2, g2, b2 = r.copy(), g.copy(), b.copy()
    ctx.apply_scnr_inplace(r2, g2, b2, method, amount, preserve)
    for got, ref in zip((r2, g2, b2), want):
        assert np.array_equal(got, ref, equal_nan=True)


def test_scnr_reference_cases_and_mismatched_dims(ctx):
    r, g, b = (np.full((2, 2), v, np.float32) for v in (0.3, 0.9, 0.3))
    ctx.apply_scnr_inplace(r, g, b, "average", 1.0, False)
    assert abs(g[0, 0] - 0.3) < 1e-5 and abs(r[0, 0] - 0.3) < 1e-5
    r, g, b = np.ones((2, 2), np.float32), np.ones((2, 3), np.float32), np.ones((2, 2), np.float32)
    ctx.apply_scnr_inplace(r, g, b)                          # scnr.rs:24-26: silently returns
    assert np.all(g == 1.0)


def test_scnr_device_planes(ctx, oracle):
    import torch
    r, g, b = rgb(np.random.default_rng(2), 64, 256)
    want = oracle.apply_scnr(r, g, b, "average", 0.8, True)
    d = [torch.from_numpy(x).cuda() for x in (r, g, b)]
    ctx.use_torch_stream()
    ctx.apply_scnr_inplace(*d, "average", 0.8, True)
    for got, ref in zip(d, want):
        assert np.array_equal(got.cpu().numpy(), ref, equal_nan=True)


@pytest.mark.parametrize("preset", ["sho", "hoo", "rgb7", "out_of_range"])
def test_blend_channels(ctx, oracle, preset):
    rng = np.random.default_rng(3)
    rows, cols = 97, 131
    k = {"sho": 3, "hoo": 2, "rgb7": 7, "out_of_range": 3}[preset]
    chans = [rng.uniform(0, 2000, (rows, cols)).astype(np.float32) for _ in range(k)]
    chans[0][4, 4] = np.nan                                   # no NaN handling in the reference: propagates
    weights = {
        "sho": [(0, 1.0, 0.0, 0.0), (1, 0.0, 1.0, 0.0), (2, 0.0, 0.0, 1.0)],
        "hoo": [(0, 1.0, 0.0, 0.0), (1, 0.0, 0.5, 0.5), (1, 0.0, 0.5, 0.5)],
        "rgb7": [(i, 0.1 * i + 0.07, 0.9 - 0.11 * i, 0.3 + 0.013 * i * i) for i in range(7)],
        "out_of_range": [(0, 0.3, 0.3, 0.3), (9, 5.0, 5.0, 5.0), (2, 0.7, 0.1, 0.2)],
    }[preset]
    want = oracle.blend_channels(chans, weights, rows, cols)
    got = ctx.blend_channels(chans, weights, rows, cols)
    for a, b in zip(got, want):
        assert np.array_equal(a, b, equal_nan=True)


def test_curve_lut_and_apply(ctx, oracle):
    pts_sets = [[(0.0, 0.0), (1.0, 1.0)], [(0.0, 0.0), (0.25, 0.15), (0.5, 0.5), (0.75, 0.85), (1.0, 1.0)],
                [(0.5, 0.7), (0.2, 0.1), (0.5 + 1e-10, 0.9)], [], [(0.3, 0.9), (0.6, 0.1)], [(0.0, 0.2), (1.0, 0.8)]]
    rng = np.random.default_rng(4)
    img = rng.uniform(-0.2, 1.3, (111, 205)).astype(np.float32)
    img[3, 3], img[4, 4], img[5, 5] = np.nan, np.inf, 1.0
    for pts in pts_sets:
        lut = ctx.spline_lut_from_points(pts)
        assert np.array_equal(lut, oracle.spline_lut_from_points(pts))      # same f64 scalar code
        assert np.array_equal(ctx.apply_curve(img, lut), oracle.apply_curve(img, lut))


@pytest.mark.parametrize("black,gamma,white", [(0.0, 1.0, 1.0), (0.2, 1.0, 1.0), (0.0, 2.0, 1.0), (0.05, 0.5, 0.9),
                                               (0.1, 100.0, 0.1), (0.0, 1.0 + 1e-9, 1.0)])
def test_levels(ctx, oracle, black, gamma, white):
    rng = np.random.default_rng(5)
    img = rng.uniform(-0.2, 1.3, (64, 200)).astype(np.float32)
    img[1, 1], img[2, 2] = np.nan, -np.inf
    got, ref = ctx.apply_levels(img, black, gamma, white), oracle.apply_levels(img, black, gamma, white)
    if abs(gamma - 1.0) < 1e-7 and black == 0.0 and white == 1.0:
        assert np.array_equal(got, ref, equal_nan=True)       # is_identity(): data.clone()
    else:
        np.testing.assert_allclose(got, ref, rtol=1e-6, atol=1e-30)


@pytest.mark.parametrize("factor,gamma", [(10.0, 1.0), (100.0, 1.0), (5.0, 0.8), (0.0, 1.0), (-3.0, 1.0)])
def test_arcsinh(ctx, oracle, factor, gamma):
    rng = np.random.default_rng(6)
    img = rng.uniform(0, 5000, (70, 300)).astype(np.float32)
    img[0, 0], img[0, 1] = np.nan, np.inf
    got = ctx.arcsinh_stretch_with_stats(img, 100.0, 4000.0, factor, gamma)
    ref = oracle.arcsinh_stretch_with_stats(img, 100.0, 4000.0, factor, gamma)
    if factor == 0.0:
        assert np.array_equal(got, ref, equal_nan=True)
    else:
        np.testing.assert_allclose(got, ref, rtol=2e-6, atol=1e-7, equal_nan=True)
    assert np.all(ctx.arcsinh_stretch_with_stats(img, 7.0, 7.0, 10.0) == 0.0)       # flat range -> zeros


def test_luminance_scale(ctx, oracle):
    r, g, b = rgb(np.random.default_rng(7))
    assert np.array_equal(ctx.luminance(r, g, b), oracle.luminance(r, g, b))
    for f in (1.37, 0.0, -2.5):
        assert np.array_equal(ctx.scale(r, f), oracle.scale(r, f), equal_nan=True)


def test_calibrate_image(ctx, oracle):
    rng = np.random.default_rng(8)
    shape = (90, 140)
    raw = rng.uniform(900, 3000, shape).astype(np.float32)
    bias = rng.uniform(90, 110, shape).astype(np.float32)
    dark = rng.uniform(0, 30, shape).astype(np.float32)
    flat = rng.uniform(0.6, 1.2, shape).astype(np.float32)
    flat[2, 2], flat[3, 3], flat[4, 4] = 0.0, np.nan, 5e-5
    raw[5, 5] = np.nan
    for combo in [(bias, None, None), (bias, dark, None), (bias, dark, flat), (None, None, flat), (None, None, None)]:
        assert np.array_equal(ctx.calibrate_image(raw, *combo, dark_exposure_ratio=1.5),
                              oracle.calibrate_image(raw, *combo, dark_exposure_ratio=1.5), equal_nan=True)


@pytest.mark.parametrize("n", [1, 2, 3, 5, 8, 11, 16, 20, 33, 64])
def test_median_combine(ctx, oracle, n):
    rng = np.random.default_rng(100 + n)
    frames = [rng.normal(1000, 20, (40, 77)).astype(np.float32) for _ in range(n)]
    for f in frames[::3]:
        f[rng.random(f.shape) < 0.05] = np.nan
    frames[0][0, 0] = np.inf
    for f in frames:
        f[1, 1] = np.nan                                     # a pixel with no finite sample -> 0
    assert np.array_equal(ctx.median_combine(frames), oracle.median_combine(frames))
