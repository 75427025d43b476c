"""The reference ships no tests for core/imaging/calibration_pipeline.rs; the batch oracle is pinned against an
independent, line-by-line numpy / pure-Python restatement of the same functions on small cases."""
import numpy as np

MAD_TO_SIGMA = 1.4826


def py_scms_pixel(vals, sl, sh, max_iter, rej):                       # calibration_pipeline.rs:340-370
    vals = [(np.float32(v), i) for i, v in enumerate(vals)]
    sl, sh = np.float32(sl), np.float32(sh)
    for _ in range(max_iter):
        if len(vals) < 3:
            break
        key = lambda v: (np.isnan(v), v)                              # f32_cmp: NaN last
        scratch = sorted((v for v, _ in vals), key=key)
        mid = len(scratch) // 2
        median = scratch[mid]
        with np.errstate(invalid="ignore"):
            dev = sorted((np.abs(np.float32(v - median)) for v in scratch), key=key)
            sigma = np.float32(np.float64(dev[mid]) * MAD_TO_SIGMA)
            if sigma < np.float32(1e-10):
                break
            kept = []
            for v, f in vals:
                z = np.float32(np.float32(v - median) / sigma)
                if z > -sl and z < sh:
                    kept.append((v, f))
                else:
                    rej[f] += 1
        if len(kept) == len(vals):
            break
        vals = kept
    if not vals:
        return np.float32(0.0)
    s = np.float32(0.0)
    with np.errstate(invalid="ignore"):
        for v, _ in vals:
            s = np.float32(s + v)
        return np.float32(s / np.float32(len(vals)))


def py_scms(frames, sl=2.5, sh=3.0, max_iter=5):
    n = len(frames)
    out = np.zeros_like(frames[0])
    rej = [0] * n
    for idx in np.ndindex(frames[0].shape):
        out[idx] = py_scms_pixel([f[idx] for f in frames], sl, sh, max_iter, rej)
    return out, rej


def frames_with_trouble(n, shape, seed):
    rng = np.random.default_rng(seed)
    fr = [rng.normal(100, 5, shape).astype(np.float32) for _ in range(n)]
    for k in range(n):
        fr[k][rng.random(shape) < 0.03] += 80.0                       # outliers
        fr[k][rng.random(shape) < 0.01] = np.nan
        fr[k][rng.random(shape) < 0.005] = np.inf
        fr[k][rng.random(shape) < 0.005] = -np.inf
    for k in range(n):
        fr[k][0, :4] = 7.0                                            # constant pixels: sigma < 1e-10 stops at once
        fr[k][1, :4] = np.nan                                         # all NaN
    for k in range(n // 2 + 1):
        fr[k][2, :4] = np.nan                                         # NaN median
        fr[k][3, :4] = np.inf                                         # +inf median
    fr[0][0, 0] = np.nan                                              # constant + one NaN: kept, the mean turns NaN
    return fr


def test_sigma_clipped_mean_stack_vs_python(oracle):
    for n, seed in ((1, 0), (2, 1), (3, 2), (4, 3), (7, 4), (16, 5), (33, 6)):
        fr = frames_with_trouble(n, (6, 9), seed)
        for sl, sh, it in ((2.5, 3.0, 5), (1.0, 1.0, 2), (3.0, 2.0, 1), (2.5, 3.0, 0), (0.5, 0.7, 50)):
            want, wrej = py_scms(fr, sl, sh, it)
            got, grej = oracle.sigma_clipped_mean_stack(fr, sl, sh, it)
            assert grej == wrej, (n, sl, sh, it)
            assert np.array_equal(got, want, equal_nan=True), (n, sl, sh, it)
    fr = frames_with_trouble(9, (4, 5), 9)
    got, _ = oracle.sigma_clipped_mean_stack(fr)
    assert np.all(got[0, 1:4] == 7.0) and np.isnan(got[0, 0]) and np.all(got[1, :4] == 0.0) and np.all(got[2, :4] == 0.0)


def test_calibrate_light_vs_numpy(oracle):                           # :74-118
    rng = np.random.default_rng(1)
    light = rng.normal(500, 50, (20, 30)).astype(np.float32)
    bias = rng.normal(100, 2, (20, 30)).astype(np.float32)
    dark = rng.normal(10, 1, (20, 30)).astype(np.float32)
    flat = rng.normal(1.0, 0.1, (20, 30)).astype(np.float32)
    flat[0, 0], flat[0, 1], flat[0, 2], light[1, 1], light[2, 2] = 0.0, np.nan, 5e-5, 50.0, np.nan
    v = (light - bias) - dark
    ok = np.isfinite(flat) & (np.abs(flat) > np.float32(1e-4))
    with np.errstate(divide="ignore", invalid="ignore"):
        want = np.where(ok, v / flat, v)
    want = np.where(want < 0, np.float32(0), want)
    assert np.array_equal(oracle.calibrate_light(light, bias, dark, flat), want, equal_nan=True)
    assert np.array_equal(oracle.calibrate_light(light, None, dark, None), np.where(light - dark < 0, np.float32(0), light - dark), equal_nan=True)
    # a master of another length is skipped (:87-89)
    assert np.array_equal(oracle.calibrate_light(light, bias[:10], None, None), np.where(light < 0, np.float32(0), light), equal_nan=True)
    assert np.array_equal(oracle.calibrate_light(light, bias.reshape(30, 20), None, None), oracle.calibrate_light(light, bias, None, None), equal_nan=True)


def test_normalize_frames_vs_numpy(oracle):                          # :309-319
    rng = np.random.default_rng(2)
    a = rng.normal(300, 20, (40, 50)).astype(np.float32)
    neg = -a
    got = oracle.normalize_frames([a, neg, np.zeros((3, 3), np.float32)])
    mean = float(np.sum(a.astype(np.float64).ravel()))                # pairwise in numpy, sequential in the reference: same to ~1e-13
    inv = np.float32(1.0) / np.float32(mean / a.size)
    assert np.allclose(got[0], a * inv, rtol=1e-6) and abs(float(got[0].mean()) - 1.0) < 1e-5
    assert np.array_equal(got[1], neg) and np.array_equal(got[2], np.zeros((3, 3), np.float32))


def test_run_batch_channel_is_the_three_steps(oracle):               # :157-190
    rng = np.random.default_rng(3)
    shape = (24, 31)
    lights = [rng.normal(400 + 30 * k, 12, shape).astype(np.float32) for k in range(9)]
    lights[4][rng.random(shape) < 0.05] += 500.0
    bias = rng.normal(100, 2, shape).astype(np.float32)
    flat = rng.normal(1.0, 0.05, shape).astype(np.float32)
    cal = [oracle.calibrate_light(l, bias, None, flat) for l in lights]
    norm = oracle.normalize_frames(cal)
    want, wrej = oracle.sigma_clipped_mean_stack(norm)
    got, rej, mean, std = oracle.run_batch_channel(lights, bias, None, flat)
    assert rej == wrej and np.array_equal(got, want)
    assert abs(mean - float(want.astype(np.float64).mean())) < 1e-12 and abs(std - float(want.astype(np.float64).std())) < 1e-12
    assert rej[4] > 2 * max(rej[0], 1)                                # the frame with the planted outliers takes the rejections
    raw, rrej, _, _ = oracle.run_batch_channel(lights, bias, None, flat, normalize=False)
    w2, wr2 = oracle.sigma_clipped_mean_stack(cal)
    assert rrej == wr2 and np.array_equal(raw, w2)


def test_compose_rgb_from_masters_vs_numpy(oracle):                  # :201-307
    rng = np.random.default_rng(4)
    r, g, b, l = (rng.normal(0.5, 0.2, (17, 23)).astype(np.float32) for _ in range(4))

    def norm(ch):
        mn, mx = ch.min(), ch.max()
        return np.clip((ch - mn) * (np.float32(1.0) / (mx - mn)), 0, 1)

    got = oracle.compose_rgb_from_masters(r, g, b)
    assert got.shape == (17, 23, 3) and np.array_equal(got, np.stack([norm(r), norm(g), norm(b)], axis=-1))
    rn, gn, bn, ln = norm(r), norm(g), norm(b), norm(l)
    lum = np.float32(0.2126) * rn + np.float32(0.7152) * gn + np.float32(0.0722) * bn
    with np.errstate(divide="ignore", invalid="ignore"):
        scale = np.where(lum > np.float32(1e-10), ln / lum, np.float32(1.0))
    want = np.stack([np.clip(c * scale, 0, 1) for c in (rn, gn, bn)], axis=-1)
    assert np.array_equal(oracle.compose_rgb_from_masters(r, g, b, l), want)
    crop = oracle.compose_rgb_from_masters(r, g[:15, :20], b, l)      # differing dims: common crop, L ignored (:209-233)
    assert crop.shape == (15, 20, 3)
    assert np.array_equal(crop, np.stack([norm(r[:15, :20]), norm(g[:15, :20]), norm(b[:15, :20])], axis=-1))
    flat = oracle.compose_rgb_from_masters(np.full((4, 4), 3.0, np.float32), g[:4, :4], b[:4, :4])
    assert np.all(flat[:, :, 0] == 0.0)                               # range < 1e-10 -> zeros (:301-303)
