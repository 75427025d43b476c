"""Pin the background-extraction oracle against the reference's unit tests (background.rs:465-592)
and an independent numpy restatement of the fit.

Reference finding: test_flat_background_extraction (:510-538) asserts |corrected| < 1 for a flat
100.0 image, but apply_correction (:367-369) re-centres on the model's median
(img - bg + model_median), so the code returns ~100.  The oracle follows the CODE."""
import numpy as np
import pytest


def gradient_image(rows=128, cols=128):                           # :544-551
    y = np.arange(rows, dtype=np.float32)[:, None]
    return np.broadcast_to((y / np.float32(rows)) * np.float32(50.0) + np.float32(100.0), (rows, cols)).astype(np.float32)


def test_flat_background(oracle):                                   # :510-538 (see module docstring)
    image = np.full((64, 64), 100.0, np.float32)
    r = oracle.extract_background(image, grid_size=4, poly_degree=1, sigma_clip=3.0, iterations=2)
    assert r.sample_count == 16
    assert np.all(np.abs(r.model[10:-10, 10:-10] - 100.0) < 1e-3)
    assert np.all(np.abs(r.corrected[10:-10, 10:-10] - 100.0) < 1.0)   # code: img - bg + median(model)
    assert r.rms_residual < 1e-3


def test_gradient_removal(oracle):                                  # :540-575
    r = oracle.extract_background(gradient_image(), grid_size=6, poly_degree=1, sigma_clip=3.0, iterations=2)
    inner = r.corrected[10:-10, 10:-10]
    assert float(inner.std()) < 5.0
    assert float(inner.std()) < 0.05                                  # a plane is fitted exactly (up to the ridge)


def test_too_small_and_not_enough_samples(oracle):                  # :127-129, :71-77
    with pytest.raises(ValueError, match="Image too small for grid_size=8"):
        oracle.extract_background(np.ones((16, 16), np.float32), grid_size=8)
    img = np.zeros((64, 64), np.float32)                            # every cell > 30 % non-positive -> no samples
    with pytest.raises(ValueError, match=r"Not enough background samples \(0\) for polynomial degree 3"):
        oracle.extract_background(img, grid_size=4)


def numpy_fit(samples, rows, cols, degree):
    """Independent restatement of fit_polynomial_surface (:251-290) with numpy's solver."""
    ny = samples[:, 0].astype(np.float64) / rows - 0.5
    nx = samples[:, 1].astype(np.float64) / cols - 0.5
    basis = np.stack([ny ** yp * nx ** (t - yp) for t in range(degree + 1) for yp in range(t, -1, -1)], axis=1)
    ata = basis.T @ basis + 1e-8 * np.eye(basis.shape[1])
    return np.linalg.solve(ata, basis.T @ samples[:, 2].astype(np.float64))


@pytest.mark.parametrize("degree", [1, 2, 3, 5])
def test_fit_matches_numpy(oracle, degree):
    rng = np.random.default_rng(degree)
    rows, cols, grid = 192, 256, 8
    y, x = np.mgrid[0:rows, 0:cols]
    ny, nx = y / rows - 0.5, x / cols - 0.5
    truth = 200.0 + 40.0 * ny - 25.0 * nx + 30.0 * ny * nx + 15.0 * nx * nx
    image = (truth + rng.normal(0, 0.5, truth.shape)).astype(np.float32)
    r = oracle.extract_background(image, grid_size=grid, poly_degree=degree, sigma_clip=50.0, iterations=1)
    assert r.sample_count == grid * grid
    ch, cw = rows // grid, cols // grid
    mh, mw = ch // 4, cw // 4
    ih, iw = ch - 2 * mh, cw - 2 * mw
    samples = []
    for gy in range(grid):
        for gx in range(grid):
            y0, x0 = gy * ch + mh, gx * cw + mw
            cell = np.sort(image[y0:y0 + ih, x0:x0 + iw].ravel())
            n = cell.size
            med = (cell[n // 2 - 1] + cell[n // 2]) / np.float32(2.0) if n % 2 == 0 else cell[n // 2]
            samples.append((np.float32(y0 + ih // 2), np.float32(x0 + iw // 2), med))
    want = numpy_fit(np.array(samples, dtype=np.float32), rows, cols, degree)
    n_terms = (degree + 1) * (degree + 2) // 2
    assert np.allclose(r.coeffs[:n_terms], want, rtol=1e-6, atol=1e-6 * np.abs(want).max())
    assert np.all(r.coeffs[n_terms:] == 0.0)
    if degree >= 2:
        assert np.abs(r.model - truth).max() < 1.0
        assert r.rms_residual < 0.5


def test_divide_mode_and_star_rejection(oracle):
    rng = np.random.default_rng(7)
    rows = cols = 256
    y, x = np.mgrid[0:rows, 0:cols]
    vignette = (1.0 - 0.3 * (((y - 128) / 256.0) ** 2 + ((x - 128) / 256.0) ** 2)).astype(np.float32)
    image = (np.float32(500.0) * vignette + rng.normal(0, 1.0, vignette.shape).astype(np.float32)).astype(np.float32)
    image[40:60, 40:60] += 5000.0                                   # a bright blob covering one cell's core
    image[3, 5] = np.nan
    image[9, 9] = np.inf
    r = oracle.extract_background(image, grid_size=8, poly_degree=2, sigma_clip=2.5, iterations=3, mode=1)
    assert r.sample_count < 64                                       # the blob's cell (and clipped corners) dropped
    flat = r.corrected[70:, 70:]
    assert float(flat.std() / flat.mean()) < 0.01                    # vignetting divided out
    assert np.isnan(r.corrected[3, 5]) and np.isinf(r.corrected[9, 9])
