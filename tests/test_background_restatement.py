"""oracle/orc_background.c held to an independent restatement of background.rs (tests/background_restatement.py): the same sample
count, the SAME coefficients, model and corrected plane bit for bit (f32 medians, f64 elimination in the source's order, f64 -> f32
model), the same rms.  The GPU path is held to the oracle by tests/test_gpu_background.py."""
import numpy as np
import pytest

import background_restatement as br


def scene(seed, rows, cols, stars=40, holes=True):
    rng = np.random.default_rng(seed)
    y, x = np.mgrid[0:rows, 0:cols]
    ny, nx = y / rows - 0.5, x / cols - 0.5
    img = 200.0 + 40.0 * ny - 25.0 * nx + 30.0 * ny * nx + 15.0 * nx * nx + rng.normal(0, 1.5, (rows, cols))
    for _ in range(stars):
        cy, cx = rng.integers(5, rows - 5), rng.integers(5, cols - 5)
        img[cy - 3:cy + 4, cx - 3:cx + 4] += rng.uniform(50, 4000)
    img = img.astype(np.float32)
    if holes:
        img[:rows // 6, :cols // 5] = 0.0                              # a padded corner: cells with > 30 % zeros are skipped
        img[rng.random((rows, cols)) < 0.002] = np.nan
        img[rng.random((rows, cols)) < 0.001] = np.inf
        img[rng.random((rows, cols)) < 0.001] = -3.0
    return img


@pytest.mark.parametrize("mode", [0, 1])
@pytest.mark.parametrize("shape,grid,degree,kappa,iters", [((192, 256), 8, 3, 2.5, 3), ((200, 333), 6, 2, 2.0, 2), ((131, 97), 5, 1, 3.0, 1),
                                                           ((300, 300), 10, 4, 2.5, 4), ((256, 256), 12, 5, 50.0, 1)])
def test_extract_background_equals_the_restatement(oracle, shape, grid, degree, kappa, iters, mode):
    img = scene(shape[0] + grid, *shape)
    try:
        model, corrected, n, rms, coeffs = br.extract_background(img, grid, degree, kappa, iters, mode)
    except br.BackgroundError as e:
        with pytest.raises(ValueError) as got:
            oracle.extract_background(img, grid, degree, kappa, iters, mode)
        assert str(got.value) == str(e)
        return
    r = oracle.extract_background(img, grid, degree, kappa, iters, mode)
    nt = (degree + 1) * (degree + 2) // 2
    assert r.sample_count == n
    assert list(r.coeffs[:nt]) == list(coeffs)
    assert np.array_equal(r.model, model)
    assert np.array_equal(r.corrected, corrected, equal_nan=True)
    assert r.rms_residual == rms


def test_error_paths_equal_the_restatement(oracle):
    for img, kw in ((np.ones((16, 16), np.float32), dict(grid_size=8)), (np.zeros((64, 64), np.float32), dict(grid_size=4)),
                    (scene(1, 64, 64, holes=False), dict(grid_size=3, poly_degree=5))):
        with pytest.raises(br.BackgroundError) as want:
            br.extract_background(img, **kw)
        with pytest.raises(ValueError) as got:
            oracle.extract_background(img, **kw)
        assert str(got.value) == str(want.value)
