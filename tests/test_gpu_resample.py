"""GPU parity: bicubic sub-pixel shift and affine warp (align.rs:36-57, affine.rs:663-690,
sampling.rs:51-80) vs the CPU oracle.  f64 weights in the reference's evaluation order ->
bit-exact f32 output is the bar."""
import math

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def make_pattern(rows, cols):                                   # align.rs:160-166
    y = np.arange(rows, dtype=np.float32)[:, None]
    x = np.arange(cols, dtype=np.float32)[None, :]
    t3 = ((np.arange(rows)[:, None] * 7 + np.arange(cols)[None, :] * 13).astype(np.float32) * np.float32(0.01))
    return (np.sin(y * np.float32(0.3)) * np.cos(x * np.float32(0.2)) * np.float32(1000.0)
            + np.float32(500.0) + np.sin(t3) * np.float32(200.0)).astype(np.float32)


@pytest.mark.parametrize("dy,dx", [(0.0, 0.0), (2.0, 3.0), (0.25, -0.75), (-7.3, 5.9), (1e-13, 0.0), (0.5, 0.5),
                                   (63.6, -70.2), (200.0, 0.3)])
def test_shift_matches_oracle(ctx, oracle, dy, dx):
    img = make_pattern(97, 133)
    got = ctx.shift_image_subpixel(img, dy, dx)
    ref = oracle.shift_image_subpixel(img, dy, dx)
    assert np.array_equal(got, ref), f"max |d| = {np.abs(got - ref).max()}"


def test_shift_reference_cases(ctx):                            # align.rs:225-241
    img = make_pattern(64, 64)
    assert np.all(np.abs(img - ctx.shift_image_subpixel(img, 0.0, 0.0)) < 1e-5)
    s = ctx.shift_image_subpixel(img, 2.0, 3.0)
    assert s.shape == (64, 64) and np.isfinite(s[30, 30])


@pytest.mark.parametrize("deg,scale,tx,ty", [(0.0, 1.0, 0.0, 0.0), (0.0, 1.0, 5.0, 3.0), (0.4, 1.0, -6.2, 4.7),
                                              (-2.0, 1.03, 3.3, -1.1), (10.0, 0.9, 20.0, -15.0),
                                              (0.0, 1.0, 1000.0, 1000.0)])
def test_warp_matches_oracle(ctx, oracle, deg, scale, tx, ty):
    rng = np.random.default_rng(3)
    img = (make_pattern(120, 150) + rng.standard_normal((120, 150)).astype(np.float32))
    img[10:12, 20:40] = np.nan                                   # NaNs propagate through the taps identically
    c, s = math.cos(math.radians(deg)) * scale, math.sin(math.radians(deg)) * scale
    t = (c, -s, tx, s, c, ty)
    for out_dims in [(120, 150), (100, 170)]:
        got = ctx.warp_image(img, t, *out_dims)
        ref = oracle.warp_image(img, t, *out_dims)
        assert np.array_equal(got, ref, equal_nan=True), f"max |d| = {np.nanmax(np.abs(got - ref))}"


@pytest.mark.parametrize("t", [(1.0, 0.0, float("nan"), 0.0, 1.0, 2.0), (float("inf"), 0.0, 1.0, 0.0, 1.0, 2.0), (1e200, -1e200, 3.0, 0.0, 1.0, 0.5),
                               (1.0, 0.0, 0.25, 1e-300, 1.0, 1e160), (-0.0, -0.0, -0.0, 0.0, 1.0, 0.0)],
                         ids=["nan tx", "inf a", "huge a and b cancel to nan", "ty beyond 1e150", "negative zeros"])
def test_warp_with_untame_coefficients(ctx, oracle, t):
    """warp_kernel takes the in-bounds test on integer floors; a coordinate that is NaN (or a coefficient that could make one) goes
    through the guarded instance -- the output equals the oracle's either way (mostly zeros: nothing maps inside the source)"""
    img = np.random.default_rng(4).normal(100, 10, (96, 130)).astype(np.float32)
    got = ctx.warp_image(img, t, 96, 130)
    ref = oracle.warp_image(img, t, 96, 130)
    assert np.array_equal(got, ref, equal_nan=True)


def test_warp_reference_cases(ctx):                             # affine.rs:776-832
    img = np.arange(2500, dtype=np.float32).reshape(50, 50)
    w = ctx.warp_image(img, (1, 0, 0, 0, 1, 0), 50, 50)
    assert np.all(np.abs(w[2:48, 2:48] - img[2:48, 2:48]) < 0.5)
    w = ctx.warp_image(np.full((50, 50), 100.0, np.float32), (1, 0, 1000.0, 0, 1, 1000.0), 50, 50)
    assert abs(w[25, 25]) < 1e-10


def test_register_then_stack_device(ctx, oracle):
    """device-resident pipeline: warp every frame onto frame 0's grid, then stack (the bench step)"""
    import torch
    from astroburst_amd import synth
    rows, cols, n = 96, 128, 6
    cat = synth.star_catalog(rows, cols, 40)
    shifts = [(0.0, 0.0), (1.3, -2.1), (-0.6, 0.4), (2.2, 2.9), (-3.1, 1.7), (0.9, -0.2)]
    frames = [synth.make_frame(rows, cols, k, cat=cat, shift=shifts[k], bad_patch_rate=0.0) for k in range(n)]
    ts = [(1.0, 0.0, dx, 0.0, 1.0, dy) for dy, dx in shifts]
    dev = [f.cuda() for f in frames]
    ctx.use_torch_stream()
    warped = [dev[0]] + [ctx.warp_image(dev[k], ts[k], rows, cols) for k in range(1, n)]
    out, rej = ctx.stack_sigma_clip(warped)
    ref_w = [frames[0].numpy()] + [oracle.warp_image(frames[k].numpy(), ts[k], rows, cols) for k in range(1, n)]
    ref, ref_rej = oracle.stack_images(ref_w, order=oracle.ORDER_ASCENDING)
    for a, b in zip(warped, ref_w):
        assert np.array_equal(a.cpu().numpy(), b)
    assert np.array_equal(out.cpu().numpy(), ref) and rej == ref_rej
