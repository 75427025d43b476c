"""HIP parity of the multi-GPU entry points and of the device-side statistics chain, on ONE MI355X.

A one-rank communicator (ncclCommInitRank with nranks = 1) runs the exact N > 1 code path of csrc/sharded.hip and
csrc/comm.hip -- real ncclAllReduce / ncclBroadcast calls on the context's stream -- so everything but the wire is
covered here; the partition / exchange protocol for N > 1 is covered on CPU ranks in tests/test_distributed_cpu.py.
"""
import numpy as np
import pytest
import torch

from astroburst_amd import synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def comm(ctx):
    import astroburst_amd as ab
    c = ab.Comm(ctx, ab.Comm.unique_id(), 1, 0)
    assert (c.rank, c.size) == (0, 1)
    yield c
    c.close()


def _frames(n, rows, cols):
    host = synth.make_stack(n, rows, cols)
    return host, [f.cuda() for f in host]


def test_comm_allreduce_is_rccl(ctx, comm):
    t = torch.arange(1000, dtype=torch.float64, device="cuda")
    before = comm.collectives_issued
    comm.allreduce(t, "sum")
    u = torch.arange(7, dtype=torch.int32, device="cuda")
    comm.allreduce(u, "max")
    torch.cuda.synchronize()
    assert comm.collectives_issued == before + 2
    assert torch.equal(t.cpu(), torch.arange(1000, dtype=torch.float64))
    assert torch.equal(u.cpu(), torch.arange(7, dtype=torch.int32))


def test_frame_sharded_stack_through_the_library(ctx, comm, oracle):
    """ab_stack_sigma_clip_sharded: partial -> all-reduce(sum f64, count u32) -> divide == the two-level oracle"""
    host, dev = _frames(24, 97, 160)
    out = torch.empty((97, 160), device="cuda")
    _, rej = ctx.stack_sigma_clip_sharded(comm, dev, out, want_rejected=True)
    s, c, want_rej = oracle.stack_partial([f.numpy() for f in host])
    want = np.where(c > 0, (s / np.maximum(c, 1)).astype(np.float32), np.float32(0))
    assert np.array_equal(out.cpu().numpy(), want)
    assert rej == want_rej
    # without a communicator the entry point is a world of one: same result, no RCCL call
    out2 = torch.empty_like(out)
    ctx.stack_sigma_clip_sharded(None, dev, out2)
    assert torch.equal(out, out2)


@pytest.mark.parametrize("world", [1, 2, 3, 8])
def test_row_bands_reassemble_the_single_level_stack(ctx, oracle, world):
    """ab_stack_sigma_clip_rows band by band == the reference's stack_images over the whole frames, bit for bit"""
    rows, cols = 50, 96
    host, dev = _frames(16, rows, cols)
    want, want_rej = oracle.stack_images([f.numpy() for f in host])
    full = torch.empty((rows, cols), device="cuda")
    tot = 0
    for r in range(world):
        row0, nrows = ctx.shard_rows(rows, world, r)
        band, rej = ctx.stack_sigma_clip_rows(dev, row0, nrows)
        full[row0:row0 + nrows] = band
        tot += rej
    assert np.array_equal(full.cpu().numpy(), want, equal_nan=True)
    assert tot == want_rej


def test_rowband_entry_and_allgather(ctx, comm, oracle):
    host, dev = _frames(8, 33, 70)
    band = torch.empty((33, 70), device="cuda")
    _, rej = ctx.stack_sigma_clip_rowband(comm, dev, band)
    want, want_rej = oracle.stack_images([f.numpy() for f in host])
    assert np.array_equal(band.cpu().numpy(), want, equal_nan=True) and rej == want_rej
    full = torch.zeros((33, 70), device="cuda")
    ctx.allgather_rows(comm, band, full)
    assert torch.equal(full, band)


def test_warp_rows_equal_the_rows_of_a_full_warp(ctx):
    g = torch.Generator().manual_seed(5)
    img = torch.rand((120, 150), generator=g).cuda()
    t = (0.9998, -0.012, 3.25, 0.011, 1.0003, -2.5)
    full = ctx.warp_image(img, t, 120, 150)
    for row0, nrows in ((0, 40), (40, 41), (81, 39), (17, 1)):
        band = torch.empty((nrows, 150), device="cuda")
        ctx.warp_image_rows(img, t, 120, row0, band)
        assert torch.equal(band, full[row0:row0 + nrows]), (row0, nrows)


def test_register_frames_sharded_equals_register_frames(ctx, comm):
    y, x, flux = synth.star_catalog(320, 384, 220, seed=5)
    cat = (y, x, flux * 30.0)
    ref = synth.make_frame(320, 384, 0, cat=cat, bad_patch_rate=0.0, cosmic_rate=0.0).cuda()
    tg = [synth.make_frame(320, 384, k, cat=cat, shift=(1.5 * k, -0.75 * k), bad_patch_rate=0.0, cosmic_rate=0.0).cuda() for k in (1, 2, 3)]
    a = ctx.register_frames(ref, tg)
    b = ctx.register_frames_sharded(comm, ref, tg)
    assert [r.transform for r in a] == [r.transform for r in b]
    assert [(r.method, r.inliers, r.matched_stars) for r in a] == [(r.method, r.inliers, r.matched_stars) for r in b]


def _stats_image(rows, cols, seed=3):
    rng = np.random.default_rng(seed)
    img = (1000 + 30 * rng.standard_normal((rows, cols))).astype(np.float32)
    img[rng.random(img.shape) < 0.001] = np.nan
    img[:5] = 0
    img[100:110, 200:260] += 20000
    return img


def _check_stats(got, want):
    assert got.valid_count == want.valid_count
    for k in ("min", "max", "median", "mad", "sigma"):
        assert getattr(got, k) == getattr(want, k), k     # from integer histograms / order statistics only: exact
    assert abs(got.mean - want.mean) <= 1e-12 * max(abs(want.mean), 1e-300)


@pytest.mark.parametrize("shape", [(2100, 2000), (600, 700), (1, 9), (2003, 2001)])
def test_auto_stretch_preview_chain(ctx, oracle, shape):
    """cmd/common.rs:18-22 as one device chain: statistics (both paths), auto_stf and the u8 plane equal the oracle's"""
    img = _stats_image(*shape) if shape[0] > 200 else np.linspace(1, 2, shape[0] * shape[1], dtype=np.float32).reshape(shape)
    want = oracle.compute_image_stats(img)
    wp = oracle.auto_stf(want)
    u8, st, p = ctx.auto_stretch_preview(torch.from_numpy(img).cuda())
    _check_stats(st, want)
    got_p = ctx.auto_stf(st)                                  # the host's auto_stf on the device's statistics ...
    assert (p.shadow, p.midtone, p.highlight) == (got_p.shadow, got_p.midtone, got_p.highlight)   # ... equals the device's
    assert (p.shadow, p.midtone, p.highlight) == (wp.shadow, wp.midtone, wp.highlight)   # (auto_stf never reads the mean)
    assert np.array_equal(u8.cpu().numpy(), oracle.apply_stf(img, oracle.auto_stf(want), want))


def test_stats_of_an_image_without_valid_pixels(ctx):
    for shape in ((50, 60), (2100, 2000)):
        z = torch.zeros(shape, device="cuda")
        z[0, 0] = float("nan")
        u8, st, p = ctx.auto_stretch_preview(z)
        assert (st.min, st.max, st.median, st.mad, st.sigma, st.mean, st.valid_count) == (0, 0, 0, 0, 0, 0, 0)
        assert (p.shadow, p.midtone, p.highlight) == (0.0, 0.5, 1.0)          # stf.rs:14-20
        assert int(u8.max()) == 0


def test_sharded_statistics_equal_the_whole_image(ctx, comm, oracle):
    """ab_compute_image_stats_sharded on a one-rank communicator (all-reduces really issued) == compute_image_stats"""
    for shape in ((2100, 2000), (600, 700)):
        img = _stats_image(*shape)
        d = torch.from_numpy(img).cuda()
        before = comm.collectives_issued
        got = ctx.compute_image_stats_sharded(comm, d, shape[0])
        assert comm.collectives_issued > before
        _check_stats(got, oracle.compute_image_stats(img))
        _check_stats(got, ctx.compute_image_stats(d))


def test_band_statistics_protocol_on_the_device(ctx, oracle):
    """the row-band protocol with the kernels: bands of one image, joined through the histogram entry point, give the
    whole image's value histogram bin for bin (what the in-stream all-reduce sums)"""
    img = _stats_image(2100, 2000)
    gmin, gmax = float(np.nanmin(np.where(img > 1e-7, img, np.nan))), float(np.nanmax(img))
    whole, s, c = ctx.stats_value_hist(torch.from_numpy(img).cuda(), gmin, gmax)
    acc, cs, cc = np.zeros(65536, np.uint64), 0.0, 0
    for r in range(3):
        row0, nrows = ctx.shard_rows(2100, 3, r)
        h, s_, c_ = ctx.stats_value_hist(torch.from_numpy(img[row0:row0 + nrows]).cuda(), gmin, gmax)
        acc += h
        cs += s_
        cc += c_
    assert np.array_equal(acc, whole) and cc == c and abs(cs - s) <= 1e-12 * abs(s)


def test_progress_and_cancel(ctx):
    """background.rs:55-116: four stage ticks with the reference's stage strings; a cancel request stops at the next boundary"""
    from astroburst_amd import AstroBurstError, _lib
    g = torch.Generator().manual_seed(1)
    img = (100 + torch.rand((256, 256), generator=g)).cuda()
    seen = []
    ctx.set_progress_cb(lambda stage, cur, tot: seen.append((stage, cur, tot)))
    try:
        ctx.extract_background(img)
        assert seen == [("sampling background", 1, 4), ("fitting polynomial surface", 2, 4), ("generating model", 3, 4),
                        ("applying correction", 4, 4)]
        ctx.request_cancel()
        with pytest.raises(AstroBurstError) as e:
            ctx.extract_background(img)
        assert e.value.code == _lib.AB_ERR_CANCELLED and "cancelled" in e.value.message.lower()
    finally:
        ctx.clear_cancel()
        ctx.set_progress_cb(None)
    ctx.extract_background(img)   # usable again
