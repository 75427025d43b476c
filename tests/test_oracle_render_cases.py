"""Pin the preview / tile-render oracle against the reference's unit tests (infra/render/tiles.rs:483-565,
infra/ipc.rs:150-230) and against independent numpy restatements of the same loops."""
import struct

import numpy as np


def test_compute_num_levels(oracle):                                 # tiles.rs:487-494
    assert oracle.tile_compute_num_levels(256, 256, 256) == 1
    assert oracle.tile_compute_num_levels(512, 512, 256) == 2
    assert oracle.tile_compute_num_levels(1024, 1024, 256) == 3
    assert 6 <= oracle.tile_compute_num_levels(14000, 14000, 256) <= 8


def test_downsample_2x_identity_dim(oracle):                         # :496-501
    assert oracle.tile_downsample_2x(np.ones((4, 4), np.float32)).shape == (2, 2)


def test_downsample_2x_values(oracle):                               # :503-517
    res = oracle.tile_downsample_2x(np.arange(1, 17, dtype=np.float32).reshape(4, 4))
    assert res.shape == (2, 2) and abs(res[0, 0] - 3.5) < 1e-4 and abs(res[1, 1] - 13.5) < 1e-4


def test_downsample_2x_non_divisible(oracle):                        # :519-524
    res = oracle.tile_downsample_2x(np.ones((5, 5), np.float32))
    assert res.shape == (3, 3) and np.all(res == 1.0)


def test_downsample_2x_skips_non_finite(oracle):                     # :58-63
    a = np.array([[1.0, np.nan], [np.inf, 4.0]], np.float32)
    assert oracle.tile_downsample_2x(a)[0, 0] == 2.5
    assert oracle.tile_downsample_2x(np.full((2, 2), np.nan, np.float32))[0, 0] == 0.0


def test_generate_tile_pyramid(oracle):                              # :526-564
    data = (np.arange(512 * 512, dtype=np.float32) / np.float32(512.0 * 512.0)).reshape(512, 512)
    tiles, levels, (gmin, gmax) = oracle.generate_tile_pyramid(data, 256)
    assert len(levels) == 2
    assert (levels[0]["cols"], levels[0]["rows"]) == (1, 1) and (levels[1]["cols"], levels[1]["rows"]) == (2, 2)
    assert (levels[1]["width"], levels[1]["height"]) == (512, 512) and levels[0]["scale_factor"] == 0.5
    assert tiles.size == 5 * 256 * 256
    # independent restatement of render_tile on the fine level
    valid = np.sort(data[data > 1e-7].ravel())
    n = valid.size
    assert gmax == valid[min(int(n * 0.999), n - 1)] and gmin == valid[min(int(n * 0.001), n - 1)]
    inv = np.float32(255.0) / np.float32(max(np.float32(gmax) - np.float32(gmin), np.float32(1e-10)))
    t = (data[256:, 256:] - np.float32(gmin)) * inv
    want = np.clip(np.sign(t) * np.floor(np.abs(t) + np.float32(0.5)), 0, 255).astype(np.uint8)
    got = tiles[levels[1]["offset"] + 3 * 65536:levels[1]["offset"] + 4 * 65536].reshape(256, 256)
    assert np.array_equal(got, want)


def test_tile_padding_and_nan(oracle):
    a = np.linspace(0.01, 1.0, 300 * 260, dtype=np.float32).reshape(300, 260)
    a[5, 5] = np.nan
    tiles, levels, _ = oracle.generate_tile_pyramid(a, 256)
    assert len(levels) == 2 and (levels[1]["cols"], levels[1]["rows"]) == (2, 2)
    t = tiles[levels[1]["offset"]:].reshape(4, 256, 256)
    assert t[0][5, 5] == 0 and np.all(t[1][:, 4:] == 0) and np.all(t[2][44:, :] == 0) and t[3][43, 3] == 255


def test_percentile_bounds_without_valid_pixels(oracle):             # :156-159
    lo, hi = oracle.tile_percentile_bounds(np.array([[0.0, -2.0, np.nan, 1e-8]], np.float32))
    assert (lo, hi) == (-2.0, np.float32(1e-8))


def test_encode_roundtrip(oracle):                                   # ipc.rs:154-174
    arr = np.arange(64 * 64, dtype=np.float32).reshape(64, 64)
    buf = oracle.ipc_encode_with_header(arr)
    w, h, mn, mx = struct.unpack("<IIff", buf[:16])
    assert (w, h) == (64, 64) and len(buf) - 16 == 64 * 64 * 4
    px = np.frombuffer(buf[16:], "<f4")
    assert abs(px[0]) < 1e-6 and abs(px[-1] - 4095.0) < 1e-6 and (mn, mx) == (0.0, 4095.0)


def test_header_layout(oracle):                                      # :176-188
    arr = np.add.outer(np.arange(100), np.arange(200)).astype(np.float32) + 1.0
    w, h, _, _ = struct.unpack("<IIff", oracle.ipc_encode_with_header(arr)[:16])
    assert (w, h) == (200, 100)


def test_encode_with_header(oracle):                                 # :190-195
    assert len(oracle.ipc_encode_with_header(np.ones((10, 10), np.float32))) == 16 + 400


def test_nan_handling(oracle):                                       # :197-
    raw = np.ones((4, 4), np.float32)
    raw[0, 0], raw[0, 1] = np.nan, np.inf
    buf = oracle.ipc_encode_with_header(raw)
    px = np.frombuffer(buf[16:], "<f4")
    assert px[0] == 0.0 and px[1] == 0.0 and px[2] == 1.0
    assert struct.unpack("<ff", buf[8:16]) == (1.0, 1.0)              # min / max over the finite inputs only
    allnan = oracle.ipc_encode_with_header(np.full((2, 2), np.nan, np.float32))
    assert struct.unpack("<ff", allnan[8:16]) == (0.0, 1.0)           # :56-57


def test_encode_downsampled(oracle):                                 # :105-148
    rng = np.random.default_rng(0)
    a = rng.normal(5, 2, (300, 500)).astype(np.float32)
    a[0, 0] = np.nan
    buf = oracle.ipc_encode_with_header(a, 100)
    w, h, mn, mx = struct.unpack("<IIff", buf[:16])
    assert (w, h) == (100, 60)
    sy = np.minimum(np.arange(60) * (300 / 60), 299).astype(int)
    sx = np.minimum(np.arange(100) * (500 / 100), 499).astype(int)
    want = np.nan_to_num(a[np.ix_(sy, sx)], nan=0.0)
    assert np.array_equal(np.frombuffer(buf[16:], "<f4").reshape(60, 100), want)
    assert (mn, mx) == (want.min(), want.max())
    assert oracle.ipc_encode_with_header(a, 500) == oracle.ipc_encode_with_header(a)   # fits: the full encoder


def test_render_rgb_preview(oracle):                                 # helpers.rs:204-322
    from oracle.pyoracle import ImageStats, StfParams
    rng = np.random.default_rng(1)
    r, g, b = (rng.uniform(-0.2, 1.2, (90, 130)).astype(np.float32) for _ in range(3))
    r[0, 0] = np.nan
    full = oracle.render_rgb_preview(r, g, b, 200)
    assert full.shape == (90, 130, 3) and full[0, 0, 0] == 0
    want = (np.clip(g, 0, 1) * np.float32(255.0)).astype(np.uint8)    # truncation, rgb.rs:30
    assert np.array_equal(full[:, :, 1], want)
    small = oracle.render_rgb_preview(r, g, b, 50)
    assert small.shape == (35, 50, 3)
    sy = np.minimum(np.arange(35) * (90 / 35), 89).astype(int)
    sx = np.minimum(np.arange(50) * (130 / 50), 129).astype(int)
    assert np.array_equal(small[:, :, 2], (np.clip(b, 0, 1) * np.float32(255.0)).astype(np.uint8)[np.ix_(sy, sx)])
    st = [ImageStats(min=-0.2, max=1.2, median=0.5, mad=0.1, sigma=0.15, mean=0.5, valid_count=r.size)] * 3
    stf = [StfParams(0.1, 0.3, 1.0), StfParams(0.0, 0.5, 1.0), StfParams(0.2, 0.2, 0.9)]
    with_stf = oracle.render_rgb_preview(r, g, b, 50, stf, st)
    for c, ch in enumerate((r, g, b)):
        assert np.array_equal(with_stf[:, :, c], oracle.apply_stf(ch, stf[c], st[c])[np.ix_(sy, sx)])


def test_rgb_tiles(oracle):                                          # tiles.rs:257-341
    from oracle.pyoracle import ImageStats, StfParams
    rng = np.random.default_rng(2)
    r, g, b = (rng.uniform(-0.2, 1.2, (300, 280)).astype(np.float32) for _ in range(3))
    tiles, levels = oracle.generate_tile_pyramid_rgb(r, g, b, 256)
    assert len(levels) == 2 and tiles.size == 5 * 256 * 256 * 3
    fine = tiles[levels[1]["offset"]:].reshape(4, 256, 256, 3)
    x = np.clip(g[:256, :256], 0, 1) * np.float32(255.0)
    assert np.array_equal(fine[0][:, :, 1], np.floor(x + np.float32(0.5)).astype(np.uint8))    # rounded, :290
    assert np.all(fine[1][:, 24:, :] == 0) and np.all(fine[2][44:, :, :] == 0)
    coarse = tiles[:256 * 256 * 3].reshape(256, 256, 3)
    half = oracle.tile_downsample_2x(b)
    assert np.array_equal(coarse[:150, :140, 2], np.floor(np.clip(half, 0, 1) * np.float32(255.0) + np.float32(0.5)).astype(np.uint8))
    st = [ImageStats(min=-0.2, max=1.2, median=0.5, mad=0.1, sigma=0.15, mean=0.5, valid_count=r.size)] * 3
    stf = [StfParams(0.1, 0.3, 1.0)] * 3
    tiles_stf, _ = oracle.generate_tile_pyramid_rgb(r, g, b, 256, stf, st)
    fine = tiles_stf[levels[1]["offset"]:].reshape(4, 256, 256, 3)
    assert np.array_equal(fine[0][:, :, 0], oracle.apply_stf(r, stf[0], st[0])[:256, :256])
