"""GPU parity for the preview / tile renderers vs the CPU oracle.  Bar: every byte exact."""
import struct

import numpy as np
import pytest

from astroburst_amd import AstroBurstError
from astroburst_amd.core import ImageStats, StfParams

pytestmark = pytest.mark.gpu


def planes(rows, cols, seed=0):
    rng = np.random.default_rng(seed)
    out = [rng.uniform(-0.2, 1.2, (rows, cols)).astype(np.float32) for _ in range(3)]
    out[0][0, 0] = np.nan
    out[1][rows // 2, cols // 3] = np.inf
    out[2][rows - 1, cols - 1] = -np.inf
    return out


def stf3(oracle):
    from oracle import pyoracle as po
    st = [ImageStats(min=-0.2, max=1.2, median=0.4 + 0.05 * c, mad=0.1, sigma=0.15, mean=0.5, valid_count=1000) for c in range(3)]
    stf = [StfParams(0.10, 0.30, 1.0), StfParams(0.0, 0.5, 1.0), StfParams(0.2, 0.2, 0.9)]
    ost = [po.ImageStats(**s.__dict__) for s in st]
    ostf = [po.StfParams(p.shadow, p.midtone, p.highlight) for p in stf]
    return stf, st, ostf, ost


@pytest.mark.parametrize("rows,cols,max_dim", [(90, 130, 200), (90, 130, 50), (513, 257, 100), (301, 999, 333), (64, 64, 1)])
def test_rgb_preview_bytes(ctx, oracle, rows, cols, max_dim):
    import torch
    r, g, b = planes(rows, cols, rows)
    stf, st, ostf, ost = stf3(oracle)
    assert ctx.preview_dims(rows, cols, max_dim) == oracle.preview_dims(rows, cols, max_dim)
    assert np.array_equal(ctx.render_rgb_preview(r, g, b, max_dim), oracle.render_rgb_preview(r, g, b, max_dim))
    want = oracle.render_rgb_preview(r, g, b, max_dim, ostf, ost)
    assert np.array_equal(ctx.render_rgb_preview(r, g, b, max_dim, stf, st), want)
    dev = ctx.render_rgb_preview(*[torch.from_numpy(x).cuda() for x in (r, g, b)], max_dim, stf, st)
    assert dev.is_cuda and np.array_equal(dev.cpu().numpy(), want)


@pytest.mark.parametrize("max_dim", [0, 100, 1000])
def test_ipc_buffer(ctx, oracle, max_dim):
    import torch
    rng = np.random.default_rng(3)
    a = rng.normal(5, 2, (300, 500)).astype(np.float32)
    a[0, 0], a[7, 9] = np.nan, np.inf
    want = oracle.ipc_encode_with_header(a, max_dim)
    assert ctx.ipc_encode_with_header(a, max_dim).tobytes() == want
    assert ctx.ipc_encode_with_header(torch.from_numpy(a).cuda(), max_dim).cpu().numpy().tobytes() == want
    allnan = ctx.ipc_encode_with_header(np.full((5, 7), np.nan, np.float32)).tobytes()
    assert struct.unpack("<IIff", allnan[:16]) == (7, 5, 0.0, 1.0) and allnan == oracle.ipc_encode_with_header(np.full((5, 7), np.nan, np.float32))


def test_reference_cases(ctx):                                        # tiles.rs:487-524
    assert [ctx.tile_compute_num_levels(n, n, 256) for n in (256, 512, 1024)] == [1, 2, 3]
    assert 6 <= ctx.tile_compute_num_levels(14000, 14000, 256) <= 8
    res = ctx.tile_downsample_2x(np.arange(1, 17, dtype=np.float32).reshape(4, 4))
    assert res.shape == (2, 2) and res[0, 0] == 3.5 and res[1, 1] == 13.5
    assert ctx.tile_downsample_2x(np.ones((5, 5), np.float32)).shape == (3, 3)


@pytest.mark.parametrize("shape", [(4, 4), (5, 5), (301, 517), (1, 9), (640, 1)])
def test_downsample_2x(ctx, oracle, shape):
    rng = np.random.default_rng(shape[0])
    a = rng.normal(0, 1e3, shape).astype(np.float32)
    a[rng.random(shape) < 0.05] = np.nan
    a[rng.random(shape) < 0.02] = np.inf
    assert np.array_equal(ctx.tile_downsample_2x(a), oracle.tile_downsample_2x(a))


def test_percentile_bounds(ctx, oracle):
    rng = np.random.default_rng(5)
    a = rng.gamma(2.0, 0.1, (700, 900)).astype(np.float32)
    a[rng.random(a.shape) < 0.1] = 0.0
    a[3, 3] = np.nan
    assert ctx.tile_percentile_bounds(a) == oracle.tile_percentile_bounds(a)
    assert ctx.tile_percentile_bounds(a, 0.25, 0.5) == oracle.tile_percentile_bounds(a, 0.25, 0.5)
    none = np.array([[0.0, -2.0, np.nan, 1e-8]], np.float32)
    assert ctx.tile_percentile_bounds(none) == oracle.tile_percentile_bounds(none) == (-2.0, np.float32(1e-8))


@pytest.mark.parametrize("rows,cols,ts", [(512, 512, 256), (300, 260, 256), (1000, 1500, 256), (257, 129, 64), (100, 90, 33), (200, 200, 256)])
def test_mono_pyramid(ctx, oracle, rows, cols, ts):
    import torch
    rng = np.random.default_rng(rows + ts)
    a = rng.beta(2.0, 5.0, (rows, cols)).astype(np.float32)
    a[rng.random(a.shape) < 0.01] = np.nan
    a[:, :3] = 0.0
    want, wl, wb = oracle.generate_tile_pyramid(a, ts)
    got, gl, gb = ctx.generate_tile_pyramid(a, ts)
    assert gl == wl and gb == wb
    assert np.array_equal(got, want)
    dev, _, _ = ctx.generate_tile_pyramid(torch.from_numpy(a).cuda(), ts)
    assert np.array_equal(dev.cpu().numpy(), want)
    assert ctx.tile_pyramid_layout(rows, cols, ts, 1) == oracle.tile_pyramid_layout(rows, cols, ts, 1)


@pytest.mark.parametrize("rows,cols,ts", [(300, 280, 256), (1030, 700, 256), (130, 257, 31)])
def test_rgb_pyramid(ctx, oracle, rows, cols, ts):
    r, g, b = planes(rows, cols, ts)
    stf, st, ostf, ost = stf3(oracle)
    want, wl = oracle.generate_tile_pyramid_rgb(r, g, b, ts)
    got, gl = ctx.generate_tile_pyramid_rgb(r, g, b, ts)
    assert gl == wl and np.array_equal(got, want)
    want, _ = oracle.generate_tile_pyramid_rgb(r, g, b, ts, ostf, ost)
    got, _ = ctx.generate_tile_pyramid_rgb(r, g, b, ts, stf, st)
    assert np.array_equal(got, want)


def test_full_size_preview_and_pyramid(ctx, oracle):
    """the compose output's sizes: 4096^2 x 3 -> 2048 preview and an 5-level pyramid, device resident"""
    import torch
    g = torch.Generator(device="cuda").manual_seed(9)
    r, gg, b = (torch.rand((4096, 4096), device="cuda", generator=g) * 1.2 - 0.1 for _ in range(3))
    stf, st, ostf, ost = stf3(oracle)
    prev = ctx.render_rgb_preview(r, gg, b, 2048, stf, st)
    want = oracle.render_rgb_preview(r.cpu().numpy(), gg.cpu().numpy(), b.cpu().numpy(), 2048, ostf, ost)
    assert np.array_equal(prev.cpu().numpy(), want)
    tiles, levels = ctx.generate_tile_pyramid_rgb(r, gg, b, 256, stf, st)
    assert len(levels) == 5 and tiles.numel() == (1 + 4 + 16 + 64 + 256) * 256 * 256 * 3
    fine = tiles[levels[4]["offset"]:].reshape(16, 16, 256, 256, 3)
    full = ctx.render_rgb_preview(r, gg, b, 4096, stf, st)           # fits: the 1:1 STF render
    assert torch.equal(fine[3, 5], full[3 * 256:4 * 256, 5 * 256:6 * 256])
    mono, ml, (lo, hi) = ctx.generate_tile_pyramid(gg.clamp(0, 1), 256)
    assert len(ml) == 5 and 0.0 < lo < 0.01 and 0.99 < hi <= 1.0


def test_errors(ctx):
    z = np.zeros((8, 8), np.float32)
    with pytest.raises(AstroBurstError, match="share their dims"):
        ctx.render_rgb_preview(z, z, np.zeros((8, 9), np.float32), 4)
    with pytest.raises(AstroBurstError, match="max_dim must be > 0"):
        ctx.render_rgb_preview(z, z, z, 0)
    with pytest.raises(AstroBurstError, match="bad arguments"):
        ctx.generate_tile_pyramid(z, 0)
    with pytest.raises(AstroBurstError, match="image stats"):
        ctx.render_rgb_preview(z, z, z, 4, [StfParams()] * 3, None)
