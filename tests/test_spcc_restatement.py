"""oracle/orc_spcc.c held to an independent restatement of spcc.rs:86-435 (tests/spcc_restatement.py, plain Python floats, the
cross-match as a real nearest-neighbour loop): same star counts, factors and colour index to 1e-12 relative (the f64 contract of
SURVEY 8c; the two sides call different libm `exp`).  The GPU path is held to the oracle by tests/test_gpu_spcc.py."""
import numpy as np
import pytest

import spcc_restatement as sr
from test_oracle_spcc_cases import coloured_field


def lum_of(r, g, b):
    return (np.float32(0.2126) * r + np.float32(0.7152) * g) + np.float32(0.0722) * b


@pytest.mark.parametrize("seed,shape,n,gains", [(3, (300, 400), 60, (1.0, 1.0, 1.0)), (4, (256, 320), 45, (2.0, 1.0, 0.5)),
                                                 (5, (400, 300), 90, (0.7, 1.1, 1.6))])
@pytest.mark.parametrize("wr", ["average_spiral", "g2v", "photopic", (0.9, 1.0, 1.1)])
def test_spcc_equals_the_restatement(oracle, seed, shape, n, gains, wr):
    r, g, b = coloured_field(seed, shape[0], shape[1], n, gains=gains)
    lum = lum_of(r, g, b)
    stars, _, _ = oracle.detect_stars(lum, 5.0)
    lum_max = oracle.compute_image_stats(lum).max
    for min_snr, max_stars, sat in ((20.0, 200, 0.9), (10.0, 12, 0.9), (20.0, 200, 0.5)):
        try:
            want = sr.spcc_from_detection(r, g, b, stars, lum_max, 1.5, min_snr, max_stars, sat, wr)
        except sr.SpccError as e:
            with pytest.raises(ValueError) as got:
                oracle.spcc_calibrate_rgb(r, g, b, 1.5, min_snr, max_stars, sat, wr, detection=(stars, lum_max))
            assert str(got.value) == str(e)
            continue
        res = oracle.spcc_calibrate_rgb(r, g, b, 1.5, min_snr, max_stars, sat, wr, detection=(stars, lum_max))
        assert (res.stars_matched, res.stars_total) == (want[3], want[4])
        assert res.g_factor == want[1] == 1.0
        assert res.r_factor == pytest.approx(want[0], rel=1e-12) and res.b_factor == pytest.approx(want[2], rel=1e-12)
        assert res.avg_color_index == pytest.approx(want[5], rel=1e-12)


def test_spcc_error_paths_equal_the_restatement(oracle):
    r, g, b = coloured_field(3, 300, 400, 60)
    lum = lum_of(r, g, b)
    stars, _, _ = oracle.detect_stars(lum, 5.0)
    lum_max = oracle.compute_image_stats(lum).max
    for scale, planes, kw in ((0.0, (r, g, b), {}), (1.0, (r, np.zeros_like(g), b), {}), (1.5, (r, g, b), dict(min_snr=1e9))):
        with pytest.raises(sr.SpccError) as want:
            sr.spcc_from_detection(*planes, stars, lum_max, scale, **kw)
        with pytest.raises(ValueError) as got:
            oracle.spcc_calibrate_rgb(*planes, scale, detection=(stars, lum_max), **kw)
        assert str(got.value) == str(want.value)


def test_aperture_and_colour_pieces(oracle):
    rng = np.random.default_rng(0)
    img = rng.uniform(0.0, 1.0, (60, 80)).astype(np.float32)
    for x, y, rad in [(40.3, 30.7, 4.5), (1.0, 2.0, 3.0), (78.9, 58.2, 6.0), (20.0, 20.0, 3.0), (-5.0, 10.0, 3.0), (85.0, 70.0, 3.0)]:
        assert oracle.aperture_flux_f32(img, x, y, rad) == sr.aperture_flux_f32(img, x, y, rad)
    for wr in ("average_spiral", "g2v", "photopic", (0.5, 1.0, 2.0)):
        assert np.allclose(oracle.spcc_white_reference_rgb(wr), sr.white_reference_rgb(wr), rtol=1e-13, atol=0)
