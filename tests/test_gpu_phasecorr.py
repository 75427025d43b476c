"""GPU parity: phase correlation (phase_correlation.rs, downsample.rs) and stack_images(align=true).

The HIP FFT performs the oracle's butterflies in the oracle's order, so the correlation surface is
compared bit for bit; dx / dy follow exactly, the confidence goes through a reduction whose order
differs (1e-9 relative)."""
import numpy as np
import pytest

from test_oracle_phasecorr_cases import make_pattern, shift_array

pytestmark = pytest.mark.gpu


def check(got, ref):
    assert got[0] == ref[0] and got[1] == ref[1], (got, ref)
    assert abs(got[2] - ref[2]) <= 1e-9 * max(1.0, abs(ref[2]))


@pytest.mark.parametrize("shape", [(128, 128), (64, 200), (100, 75), (512, 512), (33, 17), (2, 2)])
def test_correlate_single_surface_bit_exact(ctx, oracle, shape):
    rng = np.random.default_rng(shape[0])
    a = (make_pattern(*shape) + rng.standard_normal(shape).astype(np.float32) * 10)
    b = shift_array(a, 3 % shape[0], -2 % shape[1]) + rng.standard_normal(shape).astype(np.float32)
    a[0, 0] = np.nan
    got = ctx.correlate_single(a, b, want_surface=True)
    ref = oracle.correlate_single(a, b, want_surface=True)
    assert np.array_equal(got[3], ref[3]), f"surface max |d| = {np.abs(got[3] - ref[3]).max()}"
    check(got[:3], ref[:3])


def test_reference_cases(ctx):                                  # phase_correlation.rs:197-240
    img = make_pattern(128, 128)
    dx, dy, _ = ctx.phase_correlate(img, img)
    assert abs(dx) < 0.5 and abs(dy) < 0.5
    big = make_pattern(256, 256)
    dx, dy, _ = ctx.phase_correlate(big, shift_array(big, 4, -5))   # (not the reference's (10, -5): see test_hip_against_the_independent_restatement)
    assert abs(dx - 5.0) < 1.0 and abs(dy + 4.0) < 1.0
    nan = make_pattern(64, 64)
    nan[10, 10], nan[20, 30], nan[5, 5] = np.nan, np.inf, -np.inf
    dx, dy, _ = ctx.phase_correlate(nan, nan)
    assert np.isfinite(dx) and np.isfinite(dy)
    const = np.full((64, 64), 100.0, np.float32)
    assert ctx.phase_correlate(const, const) == (0.0, 0.0, 0.0)


def test_hip_against_the_independent_restatement(ctx):
    """The HIP path held to tests/phasecorr_restatement.py (numpy, written from the Rust, nothing shared with the oracle) to 1e-6 px:
    the reference's three test inputs VERBATIM (phase_correlation.rs:205-220, align.rs:216-223, pair.rs:126-156 -- whose asserted
    values the reference's own code does not produce, tests/test_oracle_phasecorr_cases.py), and aperiodic fields on either side of
    the 512 px switch to the coarse-to-fine driver."""
    import phasecorr_restatement as R
    from scipy.ndimage import gaussian_filter
    cases = []
    for n, (sy, sx) in ((256, (10, -5)), (128, (5, -3)), (128, (6, -4))):
        p = make_pattern(n, n)
        cases.append((p, shift_array(p, sy, sx)))
    base = gaussian_filter(np.random.default_rng(1).standard_normal((1300, 1300)), 2.0).astype(np.float32) * 1000
    for (r, c), (sy, sx) in [((400, 500), (7, -11)), ((512, 512), (3, 4)), ((520, 300), (2, -3)), ((600, 800), (7, -11)), ((1030, 1030), (-9, 13))]:
        cases.append((base[40:40 + r, 50:50 + c].copy(), base[40 - sy:40 - sy + r, 50 - sx:50 - sx + c].copy()))
    for ref, tgt in cases:
        got, want = ctx.phase_correlate(ref, tgt), R.phase_correlate(ref, tgt)
        assert abs(got[0] - want[0]) <= 1e-6 and abs(got[1] - want[1]) <= 1e-6, (ref.shape, got, want)
        assert abs(got[2] - want[2]) <= 1e-6 * max(1.0, abs(want[2])), (ref.shape, got, want)


@pytest.mark.parametrize("dims", [((600, 800), (600, 800)), ((520, 300), (520, 300)), ((700, 900), (650, 1000)),
                                   ((1030, 1030), (1030, 1030))])
def test_phase_correlate_coarse_to_fine(ctx, oracle, dims):
    from scipy.ndimage import gaussian_filter
    rng = np.random.default_rng(5)
    base = gaussian_filter(rng.standard_normal((1200, 1200)), 2.0).astype(np.float32) * 1000
    (r1, c1), (r2, c2) = dims
    ref = base[40:40 + r1, 50:50 + c1].copy()
    tgt = base[40 - 9:40 - 9 + r2, 50 + 13:50 + 13 + c2].copy() + rng.standard_normal((r2, c2)).astype(np.float32)
    tgt[5, 5] = np.nan
    got = ctx.phase_correlate(ref, tgt)
    check(got, oracle.phase_correlate(ref, tgt))


def test_stack_images_align_matches_oracle(ctx, oracle):
    ref = make_pattern(96, 128)
    rng = np.random.default_rng(2)
    frames = [ref] + [shift_array(ref, dy, dx) + rng.standard_normal(ref.shape).astype(np.float32)
                      for dy, dx in [(3, -2), (-4, 5), (1, 1), (0, 0)]]
    frames[2] = np.pad(frames[2], ((0, 7), (0, 3)))              # ragged: cropped top-left to the minimum dims
    res = ctx.stack_images(frames, align=True)
    out, rej, offs = oracle.stack_images_align(frames)
    assert res.offsets == offs
    assert np.array_equal(res.image, out)
    assert res.rejected_pixels == rej and res.frame_count == 5


def test_stack_images_align_device_planes(ctx, oracle):
    import torch
    ref = make_pattern(128, 160)
    rng = np.random.default_rng(3)
    frames = [ref] + [shift_array(ref, dy, dx) + rng.standard_normal(ref.shape).astype(np.float32) * 3
                      for dy, dx in [(2, 2), (-3, 1), (5, -6)]]
    ctx.use_torch_stream()
    res = ctx.stack_images([torch.from_numpy(f).cuda() for f in frames], align=True)
    out, rej, offs = oracle.stack_images_align(frames)
    assert res.offsets == offs and np.array_equal(res.image.cpu().numpy(), out) and res.rejected_pixels == rej
