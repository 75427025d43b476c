"""Pin the RGB-composition oracle: white_balance.rs:22-95 and resample.rs:146-186 unit tests transcribed,
pair.rs:98-173, plus process_rgb (rgb.rs:209-323) against a numpy composition of already-pinned oracle pieces."""
import math

import numpy as np
import pytest


def make_stats(oracle, median, mad):                                 # white_balance.rs:26-36
    return oracle.ImageStats(0.0, 1.0, median, mad, mad * 1.4826, median, 1000)


def test_wb_equal_channels_return_ones(oracle):                      # :38-45
    s = make_stats(oracle, 0.5, 0.01)
    assert all(abs(v - 1.0) < 1e-12 for v in oracle.select_wb_reference(s, s, s))


def test_wb_most_stable_channel(oracle):                             # :47-81
    sr, sg, sb = make_stats(oracle, 0.5, 0.001), make_stats(oracle, 0.4, 0.02), make_stats(oracle, 0.3, 0.03)
    r, g, b = oracle.select_wb_reference(sr, sg, sb)
    assert abs(r - 1.0) < 1e-12 and abs(g - 0.5 / 0.4) < 1e-12 and abs(b - 0.5 / 0.3) < 1e-12
    sr, sg = make_stats(oracle, 0.5, 0.05), make_stats(oracle, 0.4, 0.001)
    r, g, b = oracle.select_wb_reference(sr, sg, sb)
    assert abs(r - 0.4 / 0.5) < 1e-12 and abs(g - 1.0) < 1e-12 and abs(b - 0.4 / 0.3) < 1e-12
    sg, sb = make_stats(oracle, 0.4, 0.04), make_stats(oracle, 0.3, 0.001)
    r, g, b = oracle.select_wb_reference(sr, sg, sb)
    assert abs(r - 0.3 / 0.5) < 1e-12 and abs(g - 0.3 / 0.4) < 1e-12 and abs(b - 1.0) < 1e-12


def test_wb_near_zero_median_handled(oracle):                        # :83-94
    out = oracle.select_wb_reference(make_stats(oracle, 0.0, 0.0), make_stats(oracle, 0.5, 0.01), make_stats(oracle, 0.3, 0.02))
    assert all(math.isfinite(v) for v in out)


def test_resample_identity_down_up(oracle):                          # resample.rs:146-186
    img = np.add.outer(np.arange(100), np.arange(100)).astype(np.float32)
    assert np.abs(oracle.resample_image(img, 100, 100) - img).max() < 1e-4
    down = oracle.resample_image(np.full((200, 200), 42.0, np.float32), 100, 100)
    assert down.shape == (100, 100) and np.abs(down - 42.0).max() < 1.0
    up = oracle.resample_image(np.full((50, 50), 10.0, np.float32), 100, 100)
    assert up.shape == (100, 100) and np.abs(up - 10.0).max() < 1e-4
    with pytest.raises(ValueError, match="Target dimensions must be > 0"):
        oracle.resample_image(img, 0, 10)


def test_resample_matches_bicubic_sampler(oracle):
    rng = np.random.default_rng(0)
    img = rng.uniform(0, 1, (37, 53)).astype(np.float32)
    out = oracle.resample_image(img, 80, 31)
    sy, sx = 37 / 80, 53 / 31
    for ty, tx in [(0, 0), (79, 30), (40, 15), (3, 29), (78, 1)]:
        want = oracle.bicubic_sample(img, 37, 53, ty * sy + (sy - 1.0) * 0.5, tx * sx + (sx - 1.0) * 0.5)
        assert out[ty, tx] == np.float32(want)


def make_pattern(rows, cols):                                        # pair.rs:103-108 (f32 arithmetic)
    y = np.arange(rows, dtype=np.float32)[:, None]
    x = np.arange(cols, dtype=np.float32)[None, :]
    yi, xi = np.arange(rows)[:, None], np.arange(cols)[None, :]
    return (np.sin(y * np.float32(0.3)) * np.cos(x * np.float32(0.2)) * np.float32(1000.0) + np.float32(500.0)
            + np.sin((yi * 7 + xi * 13).astype(np.float32) * np.float32(0.01)) * np.float32(200.0)).astype(np.float32)


def shift_array(img, dy, dx):                                        # pair.rs:110-124
    out = np.zeros_like(img)
    rows, cols = img.shape
    ys, xs = np.arange(rows) - dy, np.arange(cols) - dx
    vy, vx = (ys >= 0) & (ys < rows), (xs >= 0) & (xs < cols)
    out[np.ix_(vy, vx)] = img[np.ix_(ys[vy], xs[vx])]
    return out


def test_phase_correlation_pair_alignment_follows_the_code(oracle):
    """pair.rs:126-156 expects RMSE < 50 after align_pair(PhaseCorrelation).  phase_correlate's peak sits at
    MINUS the displacement (see test_oracle_phasecorr_cases), so shifting the target by (dy, dx) as align_pair
    does moves it further away; the oracle follows the code and this test records the measured behaviour."""
    reference = make_pattern(128, 128)
    target = shift_array(reference, 6, -4)
    dx, dy, conf = oracle.phase_correlate(reference, target)
    aligned = oracle.shift_image_subpixel(target, dy, dx)
    d = (aligned[20:108, 20:108] - reference[20:108, 20:108]).astype(np.float64)
    rmse_code = math.sqrt((d * d).mean())
    truth = oracle.shift_image_subpixel(target, 6.0, -4.0)           # what a correct (dy, dx) would produce
    assert np.array_equal(truth[20:108, 20:108], reference[20:108, 20:108])
    assert dy < 0.0 < dx                                             # sign opposite to the needed (+6, -4)
    assert rmse_code > 50.0                                          # measured: ~849 (pattern amplitude 1000)


def test_compose_stf_matches_numpy(oracle):
    rng = np.random.default_rng(3)
    img = rng.uniform(-0.1, 1.2, (64, 80)).astype(np.float32)
    img[0, :4] = [np.nan, np.inf, 0.0, 5e-8]
    st = oracle.compute_image_stats(img)
    p = oracle.auto_stf(st)
    got = oracle.compose_apply_stf(img, p, st)
    v = img.astype(np.float64)
    inv = 1.0 / max(st.max - st.min, 1e-30)
    clipped = np.clip(((v - st.min) * inv - p.shadow) / max(p.highlight - p.shadow, 1e-15), 0.0, 1.0)
    with np.errstate(invalid="ignore", divide="ignore"):
        want = ((p.midtone - 1.0) * clipped / ((2.0 * p.midtone - 1.0) * clipped - p.midtone)).astype(np.float32)
    want = np.where(clipped <= 0.0, np.float32(0), np.where(clipped >= 1.0, np.float32(1), want))
    want = np.where(np.isfinite(img) & (img > 1e-7), want, np.float32(0)).astype(np.float32)
    assert np.array_equal(got, want)


def star_field(rng, rows, cols, n_stars, shift=(0.0, 0.0), scale=1.0, background=0.05):
    img = np.full((rows, cols), background, np.float64)
    sig = 3.0 / 2.3548
    for _ in range(n_stars):
        cy, cx, amp = rng.uniform(10, rows - 10), rng.uniform(10, cols - 10), rng.uniform(0.1, 0.9)
        cy, cx = cy + shift[0], cx + shift[1]
        y0, y1, x0, x1 = max(int(cy) - 10, 0), min(int(cy) + 11, rows), max(int(cx) - 10, 0), min(int(cx) + 11, cols)
        yy, xx = np.mgrid[y0:y1, x0:x1]
        img[y0:y1, x0:x1] += amp * np.exp(-((yy - cy) ** 2 + (xx - cx) ** 2) / (2 * sig * sig))
    return (img * scale).astype(np.float32)


def test_process_rgb_composition(oracle):
    rows, cols = 128, 160
    r = star_field(np.random.default_rng(1), rows, cols, 40) + np.random.default_rng(2).normal(0, 0.002, (rows, cols)).astype(np.float32)
    g = star_field(np.random.default_rng(1), rows, cols, 40, shift=(1.5, -2.25), scale=0.8) \
        + np.random.default_rng(3).normal(0, 0.002, (rows, cols)).astype(np.float32)
    b = star_field(np.random.default_rng(1), rows, cols, 40, shift=(-0.75, 1.0), scale=1.3) \
        + np.random.default_rng(4).normal(0, 0.002, (rows, cols)).astype(np.float32)
    res = oracle.process_rgb(r, g, b, scnr=dict(method="average", amount=0.8, preserve_luminance=True))
    # hand composition from the individually pinned pieces
    imgs = [r.copy()]
    offs = []
    for t in (g, b):
        dx, dy, _ = oracle.phase_correlate(r, t)
        imgs.append(oracle.shift_image_subpixel(t, dy, dx))
        offs.append((dy, dx))
    assert res.offset_g == offs[0] and res.offset_b == offs[1]
    full = [oracle.compute_image_stats(x) for x in imgs]
    wb = oracle.select_wb_reference(*full)
    for c in range(3):
        assert res.channel_stats[c] == (full[c].min, full[c].max, full[c].median, full[c].mean)
        m = np.float32(wb[c])
        if abs(m - np.float32(1.0)) >= np.float32(1e-7):
            imgs[c] = (imgs[c] * m).astype(np.float32)
        assert np.array_equal(res.pre_stretch[c], imgs[c])
    sts = [oracle.compute_image_stats(x) for x in imgs]
    stfs = [oracle.auto_stf(s) for s in sts]
    outs = [oracle.compose_apply_stf(x, p, s) for x, p, s in zip(imgs, stfs, sts)]
    outs = oracle.apply_scnr(outs[0], outs[1], outs[2], "average", 0.8, True)
    for c, got in enumerate((res.r, res.g, res.b)):
        assert res.stf[c] == stfs[c] and res.stats_wb[c] == sts[c]
        assert np.array_equal(got, outs[c])
    assert res.scnr_applied and not res.resampled and (res.rows, res.cols) == (rows, cols)


def test_process_rgb_variants_and_errors(oracle):
    rng = np.random.default_rng(7)
    r = rng.uniform(0.05, 0.6, (64, 96)).astype(np.float32)
    g = rng.uniform(0.05, 0.6, (32, 48)).astype(np.float32)
    res = oracle.process_rgb(r, g, None, align=False, white_balance="none", linked_stf=True)
    assert res.resampled and (res.rows, res.cols) == (64, 96)
    g_up = oracle.resample_image(g, 64, 96)
    assert np.array_equal(res.pre_stretch[1], g_up)
    assert np.array_equal(res.pre_stretch[2], ((r + g_up) * np.float32(0.5)).astype(np.float32))   # synthesised blue
    comb = ((r + g_up + res.pre_stretch[2]) * np.float32(1.0 / 3.0)).astype(np.float32)
    p = oracle.auto_stf(oracle.compute_image_stats(comb))
    assert res.stf == (p, p, p)
    manual = oracle.process_rgb(r, None, r, align=False, white_balance=(1.0, 2.0, 0.5), auto_stretch=False,
                                stf=(oracle.StfParams(0.1, 0.3, 0.9), None, None))
    assert manual.stf[0] == oracle.StfParams(0.1, 0.3, 0.9) and manual.stf[1] == oracle.StfParams(0.0, 0.5, 1.0)
    assert np.array_equal(manual.pre_stretch[1], (r * np.float32(2.0)).astype(np.float32))          # G = clone of (r, r) avg... r
    with pytest.raises(ValueError, match=r"Need at least 2 channels for RGB compose \(got 1\)"):
        oracle.process_rgb(r, None, None)
    tiny = np.ones((4, 4), np.float32)
    with pytest.raises(ValueError, match=r"Channel dimension ratio 24\.0x exceeds 8x limit\. R=96x64 G=4x4\. Check channel assignments\."):
        oracle.process_rgb(r, tiny, None)
