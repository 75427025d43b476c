"""Star mask + masked stretch oracle (star_mask.rs, masked_stretch.rs).

The reference holds no unit tests for these two files, so the oracle is pinned here against an
independent numpy restatement written from the same source lines, plus behavioural checks."""
import math

import numpy as np



def star_field(rng, rows, cols, n_stars, background=0.02, noise=0.002, fwhm=3.5):
    img = rng.normal(background, noise, (rows, cols))
    sig = fwhm / 2.3548
    for _ in range(n_stars):
        cy, cx, amp = rng.uniform(8, rows - 8), rng.uniform(8, cols - 8), rng.uniform(0.05, 0.9)
        y0, y1, x0, x1 = int(cy) - 12, int(cy) + 13, int(cx) - 12, int(cx) + 13
        y0, x0 = max(y0, 0), max(x0, 0)
        yy, xx = np.mgrid[y0:min(y1, rows), x0:min(x1, cols)]
        img[y0:y1, x0:x1] += amp * np.exp(-((yy - cy) ** 2 + (xx - cx) ** 2) / (2 * sig * sig))
    return img.clip(1e-5, None).astype(np.float32)


def np_star_mask(image, stars, growth=2.5, softness=4.0, min_fwhm=1.5, max_fwhm=30.0, protect=False, ceiling=0.85):
    h, w = image.shape
    mask = np.zeros((h, w), np.float32)
    count = 0
    for x, y, fwhm in stars:                                      # star_mask.rs:54-113
        if not (min_fwhm <= fwhm <= max_fwhm):
            continue
        count += 1
        radius = fwhm * growth
        soft = radius + softness
        y0, y1 = int(max(math.floor(y - soft), 0.0)), min(max(int(math.ceil(y + soft)), 0), h - 1)
        x0, x1 = int(max(math.floor(x - soft), 0.0)), min(max(int(math.ceil(x + soft)), 0), w - 1)
        if y0 > y1 or x0 > x1:
            continue
        py, px = np.mgrid[y0:y1 + 1, x0:x1 + 1]
        d2 = (px - x) ** 2 + (py - y) ** 2
        r2i, r2o = radius * radius, soft * soft
        t = ((d2 - r2i) / max(r2o - r2i, 1e-10)).astype(np.float32)
        val = np.float32(1.0) - t * t * (np.float32(3.0) - np.float32(2.0) * t)
        val = np.where(d2 <= r2i, np.float32(1.0), val)
        val = np.where(d2 <= r2o, val, np.float32(-1.0))
        sub = mask[y0:y1 + 1, x0:x1 + 1]
        np.maximum(sub, val.astype(np.float32), out=sub)
    if protect:                                                    # :115-132
        c = np.float32(ceiling)
        inv = np.float32(1.0) / (np.float32(1.0) - c) if c < 1.0 else np.float32(1.0)
        with np.errstate(invalid="ignore"):
            excess = np.clip((image - c) * inv, np.float32(0.0), np.float32(1.0))
            smooth = excess * excess * (np.float32(3.0) - np.float32(2.0) * excess)
            upd = (image > c) & (mask < 1.0) & (smooth > mask)
        mask = np.where(upd, smooth, mask).astype(np.float32)
    return mask, count, float((mask > 0.01).sum()) / (h * w)


def test_star_mask_matches_numpy(oracle):
    rng = np.random.default_rng(0)
    img = rng.uniform(0.0, 1.2, (200, 300)).astype(np.float32)
    img[4, 4] = np.nan
    img[5, 5] = np.inf
    stars = [(150.3, 100.7, 4.0), (2.1, 3.2, 3.0), (298.9, 198.5, 6.5), (100.0, 50.0, 1.0), (60.0, 60.0, 31.0),
             (-30.0, -30.0, 5.0), (400.0, 100.0, 5.0), (152.0, 103.0, 2.0), (20.5, 180.25, 29.0)]
    for protect in (False, True):
        got = oracle.generate_star_mask(img, stars=stars, luminance_protect=protect)
        want, cnt, cov = np_star_mask(img, stars, protect=protect)
        assert got.stars_masked == cnt == 7
        assert np.array_equal(got.mask, want)
        assert got.coverage_fraction == cov
    assert got.mask[100, 150] == 1.0 and got.mask[5, 5] == 1.0      # +inf pixel: full protection


def np_masked_stretch(image, mask, iterations=10, target=0.25, protection=0.85, thresh=1e-5):
    valid = np.isfinite(image) & (image > 1e-7)
    mn, mx = float(image[valid].min()), float(image[valid].max())
    rng_ = np.float32(mx - mn)
    dmin, inv = np.float32(mn), np.float32(1.0) / rng_
    with np.errstate(invalid="ignore"):
        w = np.where(np.isfinite(image) & (image > 0), np.clip((image - dmin) * inv, np.float32(0), np.float32(1)), np.float32(0))
    w = w.astype(np.float32)

    def med(a):
        v = np.sort(a[(mask < 0.5) & (a > 0)])
        return float(v[len(v) // 2]) if len(v) else 0.0

    prev, runs, conv = med(w), 0, False
    p = np.float32(protection)
    for it in range(iterations):
        runs = it + 1
        bg = med(w)
        if abs(bg - target) < thresh:
            conv = True
            break
        if it > 0 and abs(bg - prev) < thresh * 0.1:
            break
        denom = 2.0 * target * bg - target - bg
        m = 0.5 if abs(denom) < 1e-15 else min(max(bg * (target - 1.0) / denom, 0.0001), 0.9999)
        m = np.float32(m)
        den = (np.float32(2.0) * m - np.float32(1.0)) * w - m
        with np.errstate(divide="ignore", invalid="ignore"):
            s = np.clip((m - np.float32(1.0)) * w / den, np.float32(0), np.float32(1))
        s = np.where(np.abs(den) < np.float32(1e-10), w, s)
        s = np.where(w <= 0, np.float32(0), np.where(w >= 1, np.float32(1), s)).astype(np.float32)
        blend = mask * p
        w = (w * blend + s * (np.float32(1.0) - blend)).astype(np.float32)
        prev = bg
    return np.clip(w, 0, 1), runs, med(w), conv


def test_masked_stretch_with_mask_matches_numpy(oracle):
    rng = np.random.default_rng(1)
    img = (rng.normal(0.02, 0.004, (160, 240)).clip(1e-4, None)).astype(np.float32)
    img[40:50, 60:70] += 0.8
    img[0, 0] = np.nan
    img[1, 1] = -1.0
    mask = np.zeros_like(img)
    mask[35:55, 55:75] = 1.0
    mask[30:35, 55:75] = 0.4
    res = oracle.masked_stretch(img, mask=oracle.StarMaskResult(mask, 1, 0.01))
    want, runs, fbg, conv = np_masked_stretch(img, mask)
    assert res.iterations_run == runs and res.converged == conv
    assert res.final_background == fbg
    assert np.array_equal(res.image, want)
    assert res.converged and abs(res.final_background - 0.25) < 1e-5
    assert res.stars_masked == 1 and res.mask_coverage == 0.01
    # protected core is stretched less than the open sky around it
    assert res.image[45, 65] < 1.0 and res.image[100, 100] > 0.15


def test_masked_stretch_degenerate_inputs(oracle):
    flat = np.full((32, 32), 0.5, np.float32)                       # range < 1e-10 -> zeros (masked_stretch.rs:198-200)
    res = oracle.masked_stretch(flat, mask=oracle.StarMaskResult(np.zeros_like(flat), 0, 0.0))
    assert not res.image.any() and res.final_background == 0.0 and not res.converged
    assert res.iterations_run == 2                                  # bg = 0 -> never at target; stagnates on the 2nd pass
    res0 = oracle.masked_stretch(flat, iterations=0, mask=oracle.StarMaskResult(np.zeros_like(flat), 0, 0.0))
    assert res0.iterations_run == 0


def test_masked_stretch_end_to_end_on_star_field(oracle):
    rng = np.random.default_rng(5)
    img = star_field(rng, 256, 384, 60)
    res = oracle.masked_stretch(img)
    assert res.stars_masked > 10 and 0.0 < res.mask_coverage < 0.5
    assert res.image.min() >= 0.0 and res.image.max() <= 1.0
    assert abs(res.final_background - 0.25) < 0.02
    r, g, b, mask = oracle.masked_stretch_rgb_shared(img, img * np.float32(0.8), img * np.float32(1.1))
    assert r.stars_masked == g.stars_masked == b.stars_masked == mask.stars_masked
