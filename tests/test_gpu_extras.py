"""GPU parity for the caller-side helpers (LRGB, unguarded luminance, linked STF, calibrate_channel, master
frames) vs the CPU oracle.  Bar: bit-exact, except the master flat whose mean is a two-level f64 sum on the
GPU and a sequential one in the reference (f32 scale factor within 1 ulp -> planes within 1.2e-7 relative)."""
import numpy as np
import pytest

from astroburst_amd import AstroBurstError, ImageStats

pytestmark = pytest.mark.gpu


def mine(s):
    return ImageStats(s.min, s.max, s.median, s.mad, s.sigma, s.mean, s.valid_count)


def tup(s):
    return (s.min, s.max, s.median, s.mad, s.sigma, s.mean, s.valid_count)


@pytest.mark.parametrize("lw,cw", [(1.0, 1.0), (0.7, 0.6), (0.0, 0.3), (0.5, 0.0)])
def test_apply_lrgb(ctx, oracle, lw, cw):
    import torch
    rng = np.random.default_rng(0)
    l, r, g, b = (rng.uniform(0, 1, (123, 257)).astype(np.float32) for _ in range(4))
    r[:5], g[:5], b[:5] = 0.0, 0.0, 0.0                               # lum_old < 1e-10 branch
    l[7, 7] = np.nan
    want = oracle.apply_lrgb(l, r, g, b, lw, cw)
    r2, g2, b2 = r.copy(), g.copy(), b.copy()
    ctx.apply_lrgb(l, r2, g2, b2, lw, cw)
    dr, dg, db = (torch.from_numpy(x).cuda() for x in (r, g, b))
    ctx.apply_lrgb(torch.from_numpy(l).cuda(), dr, dg, db, lw, cw)
    for got, dev, ref in zip((r2, g2, b2), (dr, dg, db), want):
        assert np.array_equal(got, ref, equal_nan=True) and np.array_equal(dev.cpu().numpy(), ref, equal_nan=True)
    with pytest.raises(AstroBurstError, match=r"L dimensions \(123, 257\) do not match RGB"):
        ctx.apply_lrgb(l, r[:, :-1].copy(), g[:, :-1].copy(), b[:, :-1].copy())


def test_synthesize_luminance_unguarded(ctx, oracle):
    rng = np.random.default_rng(1)
    r, g, b = (rng.uniform(0, 1, (64, 96)).astype(np.float32) for _ in range(3))
    g[3, 3] = np.nan
    got = ctx.synthesize_luminance(r, g, b)
    assert np.array_equal(got, oracle.synthesize_luminance(r, g, b), equal_nan=True) and np.isnan(got[3, 3])


def test_linked_stf_and_calibrate_channel(ctx, oracle):
    rng = np.random.default_rng(2)
    chans = [rng.uniform(0.01, 0.9, (64, 64)).astype(np.float32) * np.float32(k) for k in (1.0, 0.7, 1.4)]
    sts = [oracle.compute_image_stats(c) for c in chans]
    stf, comb = ctx.compute_linked_stf(*[mine(s) for s in sts])
    wstf, wcomb = oracle.compute_linked_stf(*sts)
    assert (stf.shadow, stf.midtone, stf.highlight) == (wstf.shadow, wstf.midtone, wstf.highlight) and tup(comb) == tup(wcomb)
    for img in (chans[0], rng.uniform(0.01, 0.9, (2100, 2000)).astype(np.float32)):
        st = oracle.compute_image_stats(img)
        for factor in (1.5, -2.0, 0.0):
            want, wst = oracle.calibrate_channel(img, factor, st)
            got, gst = ctx.calibrate_channel(img, factor, mine(st))
            assert np.array_equal(got, want) and tup(gst) == tup(wst)


def test_create_master(ctx, oracle):
    import torch
    rng = np.random.default_rng(3)
    frames = [rng.uniform(900, 1100, (200, 311)).astype(np.float32) for _ in range(9)]
    frames[2][3, 4] = np.nan
    frames[5][100, :50] = np.inf
    bias = rng.uniform(95, 105, (200, 311)).astype(np.float32)
    dark = rng.uniform(1, 3, (200, 311)).astype(np.float32)
    assert np.array_equal(ctx.create_master("bias", frames), oracle.create_master("bias", frames))
    assert np.array_equal(ctx.create_master("dark", frames, master_bias=bias), oracle.create_master("dark", frames, master_bias=bias))
    dev = ctx.create_master("dark", [torch.from_numpy(f).cuda() for f in frames], master_bias=torch.from_numpy(bias).cuda())
    assert np.array_equal(dev.cpu().numpy(), oracle.create_master("dark", frames, master_bias=bias))
    for kw in (dict(), dict(master_bias=bias), dict(master_bias=bias, master_dark=dark)):
        want = oracle.create_master("flat", frames, **kw)
        got = ctx.create_master("flat", frames, **kw)
        assert np.allclose(got, want, rtol=1.2e-7, atol=0)
        assert (got == 1.0).sum() >= (want == 1.0).sum() - 0 and abs(float(got.mean()) - 1.0) < 1e-4
    many = [rng.uniform(900, 1100, (40, 52)).astype(np.float32) for _ in range(90)]   # > 64 frames: the wave-per-pixel stack
    many[7][1, 2] = np.nan
    assert np.array_equal(ctx.create_master("bias", many), oracle.create_master("bias", many))
    assert np.array_equal(ctx.create_master("dark", many, master_bias=bias[:40, :52].copy()),
                          oracle.create_master("dark", many, master_bias=bias[:40, :52].copy()))
    lots = [rng.uniform(900, 1100, (12, 20)).astype(np.float32) for _ in range(600)]   # > 512 frames (calibration.rs:297-318 has no limit)
    lots[5][1, 2] = np.nan
    assert np.array_equal(ctx.create_master("bias", lots), oracle.create_master("bias", lots))
    assert np.allclose(ctx.create_master("flat", lots, master_bias=bias[:12, :20].copy()),
                       oracle.create_master("flat", lots, master_bias=bias[:12, :20].copy()), rtol=1.2e-7, atol=0)
    with pytest.raises(AstroBurstError, match="No dark frames provided"):
        ctx.create_master("dark", [])
    with pytest.raises(AstroBurstError, match=r"Dimension mismatch: expected \(200, 311\), got \(200, 310\)"):
        ctx.create_master("bias", [frames[0], frames[1][:, :-1].copy()])
