"""SPCC oracle (core/astrometry/spcc.rs).  The reference holds no unit tests for this file (parity unpinned):
the restatement is pinned here against independent numpy / closed-form restatements of the same lines."""
import math

import numpy as np
import pytest


def planck_rgb(teff):                                               # spcc.rs:215-243
    def inten(lam_nm):
        lam, h, c, k = lam_nm * 1e-9, 6.626e-34, 2.998e8, 1.381e-23
        e = h * c / (lam * k * teff)
        return 0.0 if e > 500.0 else (2.0 * h * c * c / lam ** 5) / (math.exp(e) - 1.0)
    r, g, b = inten(640.0), inten(530.0), inten(460.0)
    m = max(r, g, b)
    return (1.0, 1.0, 1.0) if m < 1e-30 else (r / m, g / m, b / m)


def test_white_reference_rgb(oracle):                               # :245-255
    assert oracle.spcc_white_reference_rgb("photopic") == (1.0, 1.0, 1.0)
    assert oracle.spcc_white_reference_rgb((0.9, 1.0, 1.1)) == (0.9, 1.0, 1.1)
    assert np.allclose(oracle.spcc_white_reference_rgb("g2v"), planck_rgb(5778.0), rtol=1e-13)
    r, g, b = planck_rgb(5500.0)
    assert np.allclose(oracle.spcc_white_reference_rgb("average_spiral"), (r * 0.98, g, b * 1.02), rtol=1e-13)
    assert max(oracle.spcc_white_reference_rgb("g2v")) == 1.0


def np_aperture(img, x, y, radius):                                 # :341-383
    h, w = img.shape
    outer = radius * 1.8
    y0, y1 = int(max(math.floor(y - outer), 0)), min(max(int(math.ceil(y + outer)), 0), h - 1)
    x0, x1 = int(max(math.floor(x - outer), 0)), min(max(int(math.ceil(x + outer)), 0), w - 1)
    flux = bg = 0.0
    cnt = 0
    for py in range(y0, y1 + 1):
        for px in range(x0, x1 + 1):
            d2 = (px - x) ** 2 + (py - y) ** 2
            v = float(img[py, px])
            if d2 <= radius * radius:
                flux += v
            elif (radius * 1.2) ** 2 <= d2 <= outer * outer:
                bg += v
                cnt += 1
    if cnt:
        flux -= bg / cnt * (math.pi * radius * radius)
    return max(flux, 0.0)


def test_aperture_flux_matches_python(oracle):
    rng = np.random.default_rng(0)
    img = rng.uniform(0.0, 1.0, (60, 80)).astype(np.float32)
    for x, y, rad in [(40.3, 30.7, 4.5), (1.0, 2.0, 3.0), (78.9, 58.2, 6.0), (20.0, 20.0, 3.0), (-5.0, 10.0, 3.0)]:
        assert oracle.aperture_flux_f32(img, x, y, rad) == np_aperture(img, x, y, rad)
    flat = np.full((50, 50), 0.25, np.float32)                       # flat field: aperture - annulus mean * pi r^2
    got = oracle.aperture_flux_f32(flat, 25.0, 25.0, 5.0)
    n_in = sum(1 for py in range(50) for px in range(50) if (px - 25) ** 2 + (py - 25) ** 2 <= 25)
    assert got == pytest.approx(max(0.25 * n_in - 0.25 * math.pi * 25.0, 0.0), rel=1e-12)


def coloured_field(seed, rows, cols, n_stars, gains=(1.0, 1.0, 1.0)):
    rng = np.random.default_rng(seed)
    planes = [np.full((rows, cols), 0.02, np.float64) for _ in range(3)]
    sig = 3.2 / 2.3548
    for _ in range(n_stars):
        cy, cx, amp = rng.uniform(15, rows - 15), rng.uniform(15, cols - 15), rng.uniform(0.05, 0.6)
        col = rng.uniform(0.7, 1.3, 3)
        y0, y1, x0, x1 = int(cy) - 12, int(cy) + 13, int(cx) - 12, int(cx) + 13
        yy, xx = np.mgrid[y0:y1, x0:x1]
        psf = amp * np.exp(-((yy - cy) ** 2 + (xx - cx) ** 2) / (2 * sig * sig))
        for c in range(3):
            planes[c][y0:y1, x0:x1] += psf * col[c] * gains[c]
    nrng = np.random.default_rng(seed + 1000)
    return [(p + nrng.normal(0, 0.0008, p.shape)).astype(np.float32) for p in planes]


def test_spcc_end_to_end_and_from_detection(oracle):
    r, g, b = coloured_field(3, 300, 400, 60)
    res = oracle.spcc_calibrate_rgb(r, g, b, 1.5)
    assert res.g_factor == 1.0 and 5 <= res.stars_matched <= res.stars_total <= 200
    assert 0.3 < res.r_factor < 3.0 and 0.3 < res.b_factor < 3.0 and -0.3 <= res.avg_color_index <= 4.0
    lum = (np.float32(0.2126) * r + np.float32(0.7152) * g) + np.float32(0.0722) * b
    stars, _, _ = oracle.detect_stars(lum, 5.0)
    again = oracle.spcc_calibrate_rgb(r, g, b, 1.5, detection=(stars, oracle.compute_image_stats(lum).max))
    assert again == res
    # a per-channel gain is undone: factors scale inversely with the channel gains (chromaticity ratios)
    r2, g2, b2 = coloured_field(3, 300, 400, 60, gains=(2.0, 1.0, 0.5))
    res2 = oracle.spcc_calibrate_rgb(r2, g2, b2, 1.5, min_snr=10.0)
    assert res2.r_factor < res.r_factor and res2.b_factor > res.b_factor
    # hand restatement of compute_correction_factors (:385-435) on the same matched set
    good = [s for s in stars if s.snr >= 20.0 and s.peak < float(np.float32(oracle.compute_image_stats(lum).max * 0.9))
            and 10.0 <= s.x < 390.0 and 10.0 <= s.y < 290.0]
    good.sort(key=lambda s: -s.snr)
    good = good[:200]
    assert len(good) == res.stars_total
    sr = sg = sb = sw = sci = 0.0
    n = 0
    wr = oracle.spcc_white_reference_rgb("average_spiral")
    for s in good:
        rad = max(s.fwhm * 1.5, 3.0)
        m = [oracle.aperture_flux_f32(p, s.x, s.y, rad) for p in (r, g, b)]
        if min(m) <= 0.0:
            continue
        n += 1
        nf = min(max(s.flux / max(s.peak, 1e-10), 0.1), 100.0)
        bp = min(max(1.0 / math.sqrt(nf) + min(max(s.fwhm - 3.0, -2.0), 5.0) * 0.1, -0.3), 4.0)
        x = min(max(bp, -0.5), 5.0)
        teff = (10000.0 + (-x) * 20000.0 if x < 0 else 7500.0 + (0.5 - x) * 5000.0 if x < 0.5 else
                5800.0 + (1.0 - x) * 3400.0 if x < 1.0 else 4500.0 + (1.5 - x) * 2600.0 if x < 1.5 else
                3500.0 + (2.5 - x) * 1000.0 if x < 2.5 else 2800.0 + (5.0 - x) * 280.0)
        e = planck_rgb(teff)
        tm, te = sum(m), sum(e)
        wgt = math.sqrt(tm)
        sr += (e[0] / te) / (m[0] / tm) * wgt
        sg += (e[1] / te) / (m[1] / tm) * wgt
        sb += (e[2] / te) / (m[2] / tm) * wgt
        sw += wgt
        sci += bp
    assert n == res.stars_matched
    rf, gf, bf = sr / sw * wr[0], sg / sw * wr[1], sb / sw * wr[2]
    assert res.r_factor == pytest.approx(rf / gf, rel=1e-12) and res.b_factor == pytest.approx(bf / gf, rel=1e-12)
    assert res.avg_color_index == pytest.approx(sci / n, rel=1e-12)


def test_spcc_error_paths(oracle):
    flat = np.full((64, 64), 0.1, np.float32)
    with pytest.raises(ValueError, match=r"Only 0 stars passed quality filters \(need 5\+\)\. Try lowering min_snr\."):
        oracle.spcc_calibrate_rgb(flat, flat, flat, 1.0)
    r, g, b = coloured_field(3, 300, 400, 60)
    with pytest.raises(ValueError, match=r"Only 0 stars cross-matched \(need 3\+\)\. Check WCS solution quality\."):
        oracle.spcc_calibrate_rgb(r, g, b, 0.0)                      # degenerate WCS: match radius 0
    with pytest.raises(ValueError, match=r"Only 0 stars cross-matched"):
        oracle.spcc_calibrate_rgb(r, np.zeros_like(g), b, 1.0)       # zero green flux: no star survives :326
