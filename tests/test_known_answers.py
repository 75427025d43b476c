"""Hand-derived known answers for the reference files that have NO unit test upstream (stats.rs,
channel_blend.rs, ...).  Every expected number below is worked out by hand from the reference's code (cited), not produced by
running the oracle; the oracle (CPU) and libastroburst_hip.so (GPU) must both reproduce them.  Small N on purpose."""
import numpy as np
import pytest


# core/imaging/stats.rs:43-73 (exact path) with math/median.rs:27-73
STATS_CASES = [
    # values 1..9: median 5; |v - 5| = 4,3,2,1,0,1,2,3,4 -> sorted 0,1,1,2,2,3,3,4,4 -> [9/2] = 2 = MAD; sigma = 1.4826 * 2
    (np.arange(1, 10, dtype=np.float32).reshape(3, 3), dict(min=1.0, max=9.0, median=5.0, mad=2.0, sigma=2.9652, mean=5.0, n=9)),
    # values 1..8: even count -> median (4 + 5) / 2 = 4.5; deviations 3.5,2.5,1.5,.5,.5,1.5,2.5,3.5 -> middle two 1.5, 2.5 -> 2.0
    (np.arange(1, 9, dtype=np.float32).reshape(2, 4), dict(min=1.0, max=8.0, median=4.5, mad=2.0, sigma=2.9652, mean=4.5, n=8)),
    # invalid pixels never count (stats.rs:10-13): 0, 1e-8 (<= 1e-7), NaN, inf, -3 are skipped; valid = {2, 4, 10}
    # median 4; deviations 2, 0, 6 -> sorted 0, 2, 6 -> [1] = 2; mean 16 / 3
    (np.array([[0.0, 2.0, 1e-8, np.nan], [4.0, np.inf, -3.0, 10.0]], np.float32),
     dict(min=2.0, max=10.0, median=4.0, mad=2.0, sigma=2.9652, mean=16.0 / 3.0, n=3)),
    # one valid pixel: median = the pixel, MAD 0, sigma floored at 1e-30 (stats.rs:66)
    (np.array([[7.0, 0.0]], np.float32), dict(min=7.0, max=7.0, median=7.0, mad=0.0, sigma=1e-30, mean=7.0, n=1)),
]


def check_stats(st, want):
    assert st.valid_count == want["n"]
    assert (st.min, st.max, st.median, st.mad) == (want["min"], want["max"], want["median"], want["mad"])
    assert abs(st.sigma - want["sigma"]) <= 1e-12 * max(want["sigma"], 1e-300)
    assert abs(st.mean - want["mean"]) <= 1e-15 * max(abs(want["mean"]), 1.0)


@pytest.mark.parametrize("case", range(len(STATS_CASES)))
def test_oracle_stats_known_answers(oracle, case):
    img, want = STATS_CASES[case]
    check_stats(oracle.compute_image_stats(img), want)


def test_oracle_stats_hist_path_known_answer(oracle):
    """stats.rs:85-210 on 8 pixels by hand.  values 1..8: min 1, max 8, range 7, 65536 bins of width 7/65536.
    bin(v) = floor((v - 1) * 65536 / 7): 0, 9362, 18724, 28086, 37449, 46811, 56173, 65535(saturated).
    total 8 -> half_count = 4 -> median bin = bin of the 4th value (4.0) = 28086, count_before = 3.
    the refine histogram of that bin holds one pixel (4.0); rank in bin = 4 - 3 = 1 -> sub-bin s of 4.0, frac = 1 - 0/1 = 1:
    median = bin_lo + (s + 1) * sub_width, i.e. within one sub-bin width (7 / 65536^2) above 4.0."""
    img = np.arange(1, 9, dtype=np.float32).reshape(2, 4)
    st = oracle.compute_image_stats(img, path="hist")
    assert (st.min, st.max, st.valid_count, st.mean) == (1.0, 8.0, 8, 4.5)
    sub = 7.0 / 65536.0 / 65536.0
    assert 4.0 <= st.median <= 4.0 + 1.0001 * sub
    # deviations from the coarse median (~4.0): 3, 2, 1, 0, 1, 2, 3, 4 -> sorted 0,1,1,2,2,3,3,4: the 4th smallest is 2 -> MAD ~ 2
    assert abs(st.mad - 2.0) <= 3 * 7.0 / 65536.0
    assert abs(st.sigma - 1.4826 * st.mad) <= 1e-15


def test_oracle_blend_known_answer(oracle):
    """channel_blend.rs:13-70: per weight in list order rv += v * rw (f32 multiply, then f32 add); weights cast f64 -> f32;
    a weight whose channel_idx is out of range is skipped (:21-22)."""
    a = np.array([[1.0, 2.0]], np.float32)
    b = np.array([[10.0, 20.0]], np.float32)
    w = [(0, 0.5, 0.25, 0.0), (1, 0.1, 0.0, 1.0), (7, 9.0, 9.0, 9.0)]     # (channel_idx, r, g, b)
    r, g, bb = oracle.blend_channels([a, b], w, 1, 2)
    f = np.float32
    assert r[0, 0] == f(f(1.0) * f(0.5)) + f(f(10.0) * f(0.1)) and r[0, 1] == f(f(2.0) * f(0.5)) + f(f(20.0) * f(0.1))
    assert np.array_equal(g, np.array([[0.25, 0.5]], f))
    assert np.array_equal(bb, np.array([[10.0, 20.0]], f))


def test_oracle_auto_stf_known_answer(oracle):
    """stf.rs:13-47 by hand: min 0, max 1, median 0.25, sigma 0.05, shadow_k -2.8, target 0.25:
    shadow = 0.25 - 2.8 * 0.05 = 0.11; m = (0.25 - 0.11) / (1 - 0.11) = 0.14 / 0.89;
    midtone = mtf_balance(m, 0.25) = m (t - 1) / (2 t m - t - m)."""
    st = oracle.ImageStats(0.0, 1.0, 0.25, 0.05 / 1.4826, 0.05, 0.3, 100)
    p = oracle.auto_stf(st)
    assert abs(p.shadow - 0.11) < 1e-15 and p.highlight == 1.0
    m, t = 0.14 / 0.89, 0.25
    assert abs(p.midtone - m * (t - 1.0) / (2.0 * t * m - t - m)) < 1e-15
    assert abs(oracle.mtf(m, p.midtone) - 0.25) < 1e-12          # the stretched median lands on the target background


# ---- the same answers from the HIP path -------------------------------------------------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("case", range(len(STATS_CASES)))
def test_hip_stats_known_answers(ctx, case):
    img, want = STATS_CASES[case]
    check_stats(ctx.compute_image_stats(img), want)


@pytest.mark.gpu
def test_hip_blend_and_stf_known_answers(ctx):
    a = np.array([[1.0, 2.0]], np.float32)
    b = np.array([[10.0, 20.0]], np.float32)
    r, g, bb = ctx.blend_channels([a, b], [(0, 0.5, 0.25, 0.0), (1, 0.1, 0.0, 1.0), (7, 9.0, 9.0, 9.0)], 1, 2)
    f = np.float32
    assert r[0, 0] == f(f(1.0) * f(0.5)) + f(f(10.0) * f(0.1)) and r[0, 1] == f(f(2.0) * f(0.5)) + f(f(20.0) * f(0.1))
    assert np.array_equal(g, np.array([[0.25, 0.5]], f)) and np.array_equal(bb, np.array([[10.0, 20.0]], f))
    from astroburst_amd import ImageStats
    p = ctx.auto_stf(ImageStats(0.0, 1.0, 0.25, 0.05 / 1.4826, 0.05, 0.3, 100))
    m, t = 0.14 / 0.89, 0.25
    assert abs(p.shadow - 0.11) < 1e-15 and abs(p.midtone - m * (t - 1.0) / (2.0 * t * m - t - m)) < 1e-15
