"""oracle/orc_masked.c held to an independent numpy restatement of star_mask.rs / masked_stretch.rs (tests/masked_restatement.py):
the mask, the stretched plane, the iteration count, the final background and the coverage must be EQUAL -- f32 arithmetic in the
reference's order on both sides.  The GPU path is held to the oracle by tests/test_gpu_masked.py; no test of the reference pins
either file, so this is the second opinion (VERDICT r4 missing 5)."""
import numpy as np
import pytest

import masked_restatement as mr

STARS = [(150.3, 100.7, 4.0), (2.1, 3.2, 3.0), (298.9, 198.5, 6.5), (100.0, 50.0, 1.0), (60.0, 60.0, 31.0),
         (-30.0, -30.0, 5.0), (400.0, 100.0, 5.0), (152.0, 103.0, 2.0), (20.5, 180.25, 29.0), (151.0, 101.0, 4.0),
         (10.0, 10.0, float("nan")), (50.0, 250.0, 1.5), (299.5, 0.0, 30.0)]


def star_field(rng, rows, cols, n_stars, background=0.02, noise=0.002, fwhm=3.5, amp_hi=0.9):
    img = rng.normal(background, noise, (rows, cols))
    sig = fwhm / 2.3548
    for _ in range(n_stars):
        cy, cx, amp = rng.uniform(8, rows - 8), rng.uniform(8, cols - 8), rng.uniform(0.05, amp_hi)
        y0, y1, x0, x1 = max(int(cy) - 12, 0), int(cy) + 13, max(int(cx) - 12, 0), int(cx) + 13
        yy, xx = np.mgrid[y0:min(y1, rows), x0:min(x1, cols)]
        img[y0:y1, x0:x1] += amp * np.exp(-((yy - cy) ** 2 + (xx - cx) ** 2) / (2 * sig * sig))
    return img.clip(1e-5, None).astype(np.float32)


@pytest.mark.parametrize("protect,ceiling", [(False, 0.85), (True, 0.85), (True, 1.0), (True, 0.2)])
@pytest.mark.parametrize("growth,softness", [(2.5, 4.0), (1.0, 0.0), (3.0, 12.5)])
def test_star_mask_equals_the_restatement(oracle, protect, ceiling, growth, softness):
    rng = np.random.default_rng(0)
    img = rng.uniform(0.0, 1.2, (200, 300)).astype(np.float32)
    img[4, 4], img[5, 5], img[6, 6] = np.nan, np.inf, -np.inf
    kw = dict(growth_factor=growth, softness=softness, luminance_protect=protect, luminance_ceiling=ceiling)
    got = oracle.generate_star_mask(img, stars=STARS, **kw)
    mask, n, cov = mr.star_mask_from_stars(img, STARS, **kw)
    assert got.stars_masked == n == 10
    assert np.array_equal(got.mask, mask)
    assert got.coverage_fraction == cov


def test_star_mask_many_stars_and_none(oracle):
    rng = np.random.default_rng(2)
    stars = [(rng.uniform(-5, 530), rng.uniform(-5, 270), rng.uniform(1.0, 12.0)) for _ in range(800)]
    img = rng.uniform(0, 1, (256, 512)).astype(np.float32)
    got = oracle.generate_star_mask(img, stars=stars, luminance_protect=True, luminance_ceiling=0.9)
    mask, n, cov = mr.star_mask_from_stars(img, stars, luminance_protect=True, luminance_ceiling=0.9)
    assert got.stars_masked == n and np.array_equal(got.mask, mask) and got.coverage_fraction == cov
    got = oracle.generate_star_mask(img, stars=[])
    mask, n, cov = mr.star_mask_from_stars(img, [])
    assert n == got.stars_masked == 0 and not mask.any() and cov == got.coverage_fraction == 0.0


@pytest.mark.parametrize("rows,cols", [(160, 240), (333, 517)])
@pytest.mark.parametrize("cfg", [dict(), dict(iterations=3, target_background=0.12, protection_amount=1.0),
                                 dict(iterations=25, target_background=0.4, protection_amount=0.3, convergence_threshold=1e-7),
                                 dict(iterations=0), dict(iterations=40, target_background=0.9999, protection_amount=0.0)])
def test_masked_stretch_equals_the_restatement(oracle, rows, cols, cfg):
    rng = np.random.default_rng(rows + cols)
    img = star_field(rng, rows, cols, 80)
    img[0, 0], img[1, 1], img[2, 2], img[3, 3] = np.nan, -1.0, np.inf, 0.0
    stars = [(rng.uniform(0, cols), rng.uniform(0, rows), rng.uniform(1.5, 8.0)) for _ in range(60)]
    m_or = oracle.generate_star_mask(img, stars=stars, luminance_protect=True)
    mask, n, cov = mr.star_mask_from_stars(img, stars, luminance_protect=True)
    assert np.array_equal(m_or.mask, mask)
    got = oracle.masked_stretch(img, mask=m_or, **cfg)
    out, iters, final_bg, conv = mr.masked_stretch_with_mask(img, mask, **cfg)
    assert (got.iterations_run, got.final_background, got.converged) == (iters, final_bg, conv)
    assert np.array_equal(got.image, out)
    assert got.stars_masked == n and got.mask_coverage == cov


def test_masked_stretch_degenerate_planes(oracle):
    """a constant plane (range < 1e-10 -> zeros, median 0), a plane without a valid pixel, a plane that is all mask"""
    flat = np.full((40, 50), 3.25, np.float32)
    none = np.full((40, 50), -2.0, np.float32)
    none[3, 3] = np.nan
    rng = np.random.default_rng(4)
    busy = rng.uniform(0.01, 0.9, (40, 50)).astype(np.float32)
    zero_mask = np.zeros((40, 50), np.float32)
    full_mask = np.ones((40, 50), np.float32)
    from oracle.pyoracle import StarMaskResult
    for img, mk in ((flat, zero_mask), (none, zero_mask), (busy, full_mask), (busy, zero_mask)):
        got = oracle.masked_stretch(img, mask=StarMaskResult(mk, 0, 0.0))
        out, iters, final_bg, conv = mr.masked_stretch_with_mask(img, mk)
        assert (got.iterations_run, got.final_background, got.converged) == (iters, final_bg, conv)
        assert np.array_equal(got.image, out)


def test_shared_rgb_luminance(oracle):
    rng = np.random.default_rng(9)
    r, g, b = (rng.uniform(0, 1, (64, 80)).astype(np.float32) for _ in range(3))
    r[1, 1], g[2, 2], b[3, 3] = np.nan, np.inf, -np.inf
    assert np.array_equal(oracle.masked_stretch_luminance(r, g, b), mr.luminance(r, g, b))
