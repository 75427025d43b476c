"""Pin the phase-correlation oracle: its FFT against numpy's (an independent exact DFT), and the
reference's unit tests (phase_correlation.rs:197-240, downsample.rs:52-67, align.rs:187-223).

SIGN CONVENTION.  The reference computes ifft2(A * conj(B)) with A = fft2(reference),
B = fft2(target) (complex.rs:27-33, phase_correlation.rs:124-126).  For target(y, x) =
reference(y - sy, x - sx) that surface peaks at (-sy, -sx) -- checked below against numpy's FFT.
Three reference tests assert the OPPOSITE sign (phase_correlation.rs:205-220 expects dy = +10 for
shift_array(img, 10, -5); align.rs:187-223 likewise); by the code's own arithmetic they cannot
pass, and the reference CI never runs `cargo test`.  As with warp_image, the oracle
follows the CODE: magnitudes and tolerances are taken from those tests, the sign from the code."""
import numpy as np


def make_pattern(rows, cols):                                  # phase_correlation.rs:171-177
    y = np.arange(rows, dtype=np.float32)[:, None]
    x = np.arange(cols, dtype=np.float32)[None, :]
    t3 = ((np.arange(rows)[:, None] * 7 + np.arange(cols)[None, :] * 13).astype(np.float32) * np.float32(0.01))
    return (np.sin(y * np.float32(0.3)) * np.cos(x * np.float32(0.2)) * np.float32(1000.0)
            + np.float32(500.0) + np.sin(t3) * np.float32(200.0)).astype(np.float32)


def shift_array(img, dy, dx):                                  # phase_correlation.rs:179-195
    out = np.zeros_like(img)
    rows, cols = img.shape
    ys, xs = np.arange(rows) - dy, np.arange(cols) - dx
    vy, vx = (ys >= 0) & (ys < rows), (xs >= 0) & (xs < cols)
    out[np.ix_(vy, vx)] = img[np.ix_(ys[vy], xs[vx])]
    return out


def test_fft2d_matches_numpy(oracle):
    rng = np.random.default_rng(0)
    for shape in [(1, 1), (2, 4), (8, 8), (64, 32), (512, 256)]:
        z = rng.standard_normal(shape) + 1j * rng.standard_normal(shape)
        f = oracle.fft2d(z)
        ref = np.fft.fft2(z)
        assert np.max(np.abs(f - ref)) <= 1e-11 * max(1.0, np.max(np.abs(ref)))
        back = oracle.fft2d(f, inverse=True)
        assert np.max(np.abs(back - z)) <= 1e-12


def test_identical_images(oracle):                             # :197-203
    img = make_pattern(128, 128)
    dx, dy, _ = oracle.phase_correlate(img, img)
    assert abs(dx) < 0.5 and abs(dy) < 0.5


def test_sign_convention_is_the_codes(oracle):
    img = np.random.default_rng(0).standard_normal((128, 128))
    tgt = np.roll(img, (5, -3), axis=(0, 1))                    # tgt(y, x) = img(y - 5, x + 3), circular
    x = np.fft.fft2(img) * np.conj(np.fft.fft2(tgt))
    c = np.fft.ifft2(x / np.maximum(np.abs(x), 1e-15)).real
    py, px = np.unravel_index(np.argmax(c), c.shape)
    assert (py, px) == (128 - 5, 3)                             # i.e. (-5, +3): minus the applied shift
    pat = make_pattern(256, 256)
    dx, dy, _ = oracle.phase_correlate(pat, shift_array(pat, 5, -3))
    assert abs(dy + 5.0) < 0.5 and abs(dx - 3.0) < 0.5


def test_known_integer_shift(oracle):                          # :205-220 (tolerance 1.0; sign per the code)
    img = make_pattern(256, 256)
    dx, dy, _ = oracle.phase_correlate(img, shift_array(img, 4, -5))
    assert abs(dx - 5.0) < 1.0 and abs(dy + 4.0) < 1.0


def test_nan_no_panic(oracle):                                 # :222-231
    img = make_pattern(64, 64)
    img[10, 10], img[20, 30], img[5, 5] = np.nan, np.inf, -np.inf
    dx, dy, _ = oracle.phase_correlate(img, img)
    assert np.isfinite(dx) and np.isfinite(dy)


def test_constant_image(oracle):                               # :233-240
    img = np.full((64, 64), 100.0, np.float32)
    assert oracle.phase_correlate(img, img) == (0.0, 0.0, 0.0)


def test_estimate_offset(oracle):                              # align.rs:216-223 (tolerance 1.5; sign per the code)
    ref = make_pattern(256, 256)            # (the 128 px pattern is too periodic: sin(0.3 y) repeats every 21 px)
    dx, dy, _ = oracle.phase_correlate(ref, shift_array(ref, 5, -3))
    assert abs(dy + 5.0) < 1.5 and abs(dx - 3.0) < 1.5


def test_large_image_coarse_to_fine(oracle):
    rng = np.random.default_rng(1)
    base = rng.standard_normal((700, 900)).astype(np.float32)
    from scipy.ndimage import gaussian_filter
    base = gaussian_filter(base, 2.0).astype(np.float32) * 1000
    ref = base[20:620, 30:830]
    tgt = base[20 - 7:620 - 7, 30 + 11:830 + 11]               # tgt(y, x) = ref(y - 7, x + 11)
    dx, dy, conf = oracle.phase_correlate(ref, tgt)
    # coarse pass: minus the shift (the code's sign).  The refine crop is then centred at
    # ref_centre + coarse (phase_correlation.rs:68-72), i.e. moved AWAY from the match, so the crops are
    # offset by twice the shift and the refinement adds another -2x: the code returns -3x the shift.
    assert abs(dy + 21.0) < 2.0 and abs(dx - 33.0) < 2.0 and conf > 5.0


def test_area_downsample(oracle):                              # downsample.rs:52-67
    img = np.arange(64, dtype=np.float32).reshape(8, 8)
    assert np.array_equal(oracle.area_downsample(img, 8, 8), img)
    ds = oracle.area_downsample(img, 4, 4)
    assert ds.shape == (4, 4) and abs(ds[0, 0] - (0 + 1 + 8 + 9) / 4.0) < 1e-6
    img[0, 0] = np.nan
    assert abs(oracle.area_downsample(img, 4, 4)[0, 0] - (1 + 8 + 9) / 3.0) < 1e-6


def test_stack_images_align(oracle):
    ref = make_pattern(96, 128)
    rng = np.random.default_rng(2)
    frames = [ref] + [shift_array(ref, dy, dx) + rng.standard_normal(ref.shape).astype(np.float32)
                      for dy, dx in [(3, -2), (-4, 5), (1, 1)]]
    out, rej, offs = oracle.stack_images_align(frames)
    assert offs[0] == (0, 0)
    for k in (1, 2, 3):                                        # offsets = rounded (dy, dx) of phase_correlate (combine.rs:135-136)
        dx, dy, _ = oracle.phase_correlate(frames[0], frames[k])
        assert offs[k] == (int(np.round(dy)), int(np.round(dx)))
    assert out.shape == ref.shape and np.isfinite(out).all()
