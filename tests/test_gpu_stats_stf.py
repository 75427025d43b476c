"""GPU parity: compute_image_stats (stats.rs) and the STF stretch (stf.rs) vs the CPU oracle.

Bar: integer results (valid_count, every histogram bin, u8 STF output) bit-exact; min/max/median/
mad/sigma exact (same f64 scalar code on identical integer histograms / exact order statistics);
mean within 1e-12 relative (f64 summation order is unspecified in the reference, stats.rs:252-257)."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(params=["resident", "chain"])
def hist_engine(request):
    """The histogram path has two engines in the library: one kernel that holds the plane in the register file
    (csrc/stats_resident.hpp, the default where it fits) and the five-pass chain (AB_STATS_CHAIN=1; also what sharded, very
    large and fetch-less calls use).  The library reads the variable per call."""
    old = os.environ.get("AB_STATS_CHAIN")
    os.environ["AB_STATS_CHAIN"] = "1" if request.param == "chain" else "0"
    yield request.param
    if old is None:
        os.environ.pop("AB_STATS_CHAIN", None)
    else:
        os.environ["AB_STATS_CHAIN"] = old


def sky_image(rng, rows, cols, pad=True):
    img = (1000.0 + 30.0 * rng.standard_normal((rows, cols))).astype(np.float32)
    stars = rng.random((rows, cols)) < 1e-3
    img[stars] += rng.pareto(2.5, stars.sum()).astype(np.float32) * 2000.0
    if pad:
        img[:8, :] = 0.0
        img[:, :5] = 1e-8                                        # below PADDING_THRESHOLD
        img[20:24, 30:60] = np.nan
        img[40, 41] = np.inf
        img[41, 41] = -5.0
    return img


def check_stats(got, ref, exact_mean=False):
    assert got.valid_count == ref.valid_count
    assert got.min == ref.min and got.max == ref.max
    assert got.median == ref.median, (got.median, ref.median)
    assert got.mad == ref.mad and got.sigma == ref.sigma
    assert abs(got.mean - ref.mean) <= 1e-12 * abs(ref.mean)


@pytest.mark.parametrize("shape", [(4, 4), (100, 100), (257, 311), (1200, 1000)])
def test_stats_exact_path(ctx, oracle, shape):                  # stats.rs:43-73
    rng = np.random.default_rng(shape[0])
    img = sky_image(rng, *shape, pad=shape[0] > 50)
    check_stats(ctx.compute_image_stats(img), oracle.compute_image_stats(img))


def test_stats_exact_even_and_odd_counts(ctx, oracle):          # median.rs:27-63 even-n averaging
    for n in (1, 2, 3, 4, 5, 1000, 1001):
        img = np.linspace(1.0, 50.0, n, dtype=np.float32).reshape(1, n)
        check_stats(ctx.compute_image_stats(img), oracle.compute_image_stats(img))


def test_stats_all_invalid(ctx, oracle):
    img = np.zeros((32, 32), np.float32)
    img[3, 3] = np.nan
    got = ctx.compute_image_stats(img)
    assert got.valid_count == 0 and got.min == 0.0 and got.sigma == 0.0    # ImageStats::default()


@pytest.mark.parametrize("shape", [(2048, 2051), (2100, 2300)])
def test_stats_hist_path(ctx, oracle, shape, hist_engine):      # stats.rs:75-210 (> 4 000 000 px)
    rng = np.random.default_rng(9)
    img = sky_image(rng, *shape)
    ref = oracle.compute_image_stats(img)
    check_stats(ctx.compute_image_stats(img), ref)
    # pass-2 histogram bin for bin
    h, s, c = ctx.stats_value_hist(img, ref.min, ref.max)
    rh, rs, rc = oracle.stats_value_hist(img, ref.min, ref.max)
    assert np.array_equal(h, rh) and c == rc and abs(s - rs) <= 1e-12 * abs(rs)
    # known-range variant (stats.rs:25-41), including the fall-backs
    check_stats(ctx.compute_image_stats_with_known_range(img, 900.0, 1500.0),
                oracle.compute_image_stats_with_known_range(img, 900.0, 1500.0))
    check_stats(ctx.compute_image_stats_with_known_range(img, float("nan"), 1.0),
                oracle.compute_image_stats_with_known_range(img, float("nan"), 1.0))


def test_stats_hist_path_constant_and_tiny_range(ctx, oracle, hist_engine):
    img = np.full((2048, 2049), 7.25, np.float32)
    check_stats(ctx.compute_image_stats(img), oracle.compute_image_stats(img))
    img[0, 0] = np.float32(7.2500005)
    check_stats(ctx.compute_image_stats(img), oracle.compute_image_stats(img))


def test_stats_hist_path_shapes_of_the_resident_kernel(ctx, oracle, hist_engine):
    """What the one-kernel engine has to get right beyond the sky image: a pixel count that is not a multiple of four or of a
    workgroup's 65 536 pixels, exactly 256 full workgroups (4096 x 4096), a plane with no valid pixel, a plane whose valid
    pixels all sit in one workgroup, two-valued data (the MAD's rank falls on a bin edge), heavy tails."""
    rng = np.random.default_rng(31)
    cases = {
        "odd count": sky_image(rng, 2049, 2051),
        "4096 x 4096": sky_image(rng, 4096, 4096),
        "no valid pixel": np.zeros((2048, 2050), np.float32),
        "valid pixels in one workgroup": np.zeros((2048, 2050), np.float32),
        "two values": np.where(rng.random((2048, 2050)) < 0.5, np.float32(3.0), np.float32(5.0)).astype(np.float32),
        "heavy tails": (rng.standard_cauchy((2048, 2050)) * 10.0 + 500.0).astype(np.float32),
        "ramp": np.linspace(1.0, 2.0, 2048 * 2050, dtype=np.float32).reshape(2048, 2050),
    }
    cases["valid pixels in one workgroup"][700, :1500] = (100.0 + rng.standard_normal(1500)).astype(np.float32)
    for name, img in cases.items():
        got, ref = ctx.compute_image_stats(img), oracle.compute_image_stats(img)
        try:
            check_stats(got, ref)
        except AssertionError as e:
            raise AssertionError(f"{name} [{hist_engine}]: {got} != {ref}") from e


@pytest.mark.parametrize("shape", [(2048, 2051), (2049, 2051), (4096, 4096)])
def test_auto_stretch_preview_hist_path(ctx, oracle, shape, hist_engine):   # cmd/common.rs:18-22 on > 4 000 000 px
    import torch
    rng = np.random.default_rng(shape[1])
    img = sky_image(rng, *shape)
    st = oracle.compute_image_stats(img)
    p = oracle.auto_stf(st)
    want = oracle.apply_stf(img, p, st)
    u8, gst, gp = ctx.auto_stretch_preview(torch.from_numpy(img).cuda())
    check_stats(gst, st)
    assert (gp.shadow, gp.midtone, gp.highlight) == (p.shadow, p.midtone, p.highlight)
    assert np.array_equal(u8.cpu().numpy(), want)
    # the fetch-less form (always the chain) leaves the same bytes
    u8b, _, _ = ctx.auto_stretch_preview(torch.from_numpy(img).cuda(), fetch=False)
    ctx.synchronize()
    assert np.array_equal(u8b.cpu().numpy(), want)


@pytest.mark.parametrize("bins", [512, 65536, 100000])
def test_build_histogram(ctx, oracle, bins):                    # stats.rs:378-421
    rng = np.random.default_rng(4)
    img = sky_image(rng, 300, 400)
    st = oracle.compute_image_stats(img)
    assert np.array_equal(ctx.build_histogram(img, bins, st.min, st.max), oracle.build_histogram(img, bins, st.min, st.max))
    assert ctx.build_histogram(img, bins, 5.0, 5.0).sum() == 0   # range < 1e-10


# ---- STF ---------------------------------------------------------------------------------------------
def test_stf_matches_oracle(ctx, oracle):
    rng = np.random.default_rng(8)
    img = sky_image(rng, 301, 403)
    st = oracle.compute_image_stats(img)
    gst = ctx.compute_image_stats(img)
    p = oracle.auto_stf(st)
    gp = ctx.auto_stf(gst)
    assert (gp.shadow, gp.midtone, gp.highlight) == (p.shadow, p.midtone, p.highlight)
    assert np.array_equal(ctx.apply_stf(img, gp, gst), oracle.apply_stf(img, p, st))
    assert np.array_equal(ctx.apply_stf_f32(img, gp, gst), oracle.apply_stf_f32(img, p, st))
    for params in [oracle.StfParams(0.0, 0.5, 1.0), oracle.StfParams(0.1, 0.02, 0.9), oracle.StfParams(0.5, 0.5, 0.5)]:
        assert np.array_equal(ctx.apply_stf(img, params, gst), oracle.apply_stf(img, params, st))
        assert np.array_equal(ctx.apply_stf_f32(img, params, gst), oracle.apply_stf_f32(img, params, st))


def test_stf_reference_cases(ctx):                              # stf.rs:214-262
    data = (np.arange(1, 17, dtype=np.float32) * 100.0).reshape(4, 4)
    st = ctx.compute_image_stats(data)
    import astroburst_amd as ab
    buf = ctx.apply_stf(data, ab.StfParams(0.0, 0.5, 1.0), st).ravel()
    assert buf[0] == 0 and buf[15] == 255
    raw = np.zeros(16, np.float32)
    raw[8], raw[9] = 0.5, 1.0
    d2 = raw.reshape(4, 4)
    buf = ctx.apply_stf(d2, ab.StfParams(0.0, 0.5, 1.0), ctx.compute_image_stats(d2)).ravel()
    assert np.all(buf[:8] == 0)


def test_stf_inplace_device(ctx, oracle):                       # stf.rs:147-155
    import torch
    rng = np.random.default_rng(2)
    img = sky_image(rng, 128, 256)
    st = oracle.compute_image_stats(img)
    p = oracle.auto_stf(st)
    d = torch.from_numpy(img).cuda()
    ctx.use_torch_stream()
    gst = ctx.compute_image_stats(d)
    ctx.apply_stf_f32(d, ctx.auto_stf(gst), gst, out=d)
    assert np.array_equal(d.cpu().numpy(), oracle.apply_stf_f32(img, p, st))


def test_stats_hist_path_from_two_threads(oracle):
    """The one-kernel engine needs the whole chip; two contexts calling at once must not wait for each other's workgroups (the
    library lets one of them through and gives the other the chain).  Results stay exact either way."""
    import threading
    import astroburst_amd as ab
    rng = np.random.default_rng(77)
    imgs = [sky_image(rng, 2048, 2100 + 8 * i) for i in range(2)]
    refs = [oracle.compute_image_stats(im) for im in imgs]
    ctxs = [ab.Context(0) for _ in imgs]
    out, errs = [[] for _ in imgs], []

    def work(i):
        try:
            for _ in range(20):
                out[i].append(ctxs[i].compute_image_stats(imgs[i]))
        except Exception as e:      # noqa: BLE001
            errs.append(e)

    ths = [threading.Thread(target=work, args=(i,)) for i in range(len(imgs))]
    [t.start() for t in ths]
    [t.join() for t in ths]
    [c.close() for c in ctxs]
    assert not errs, errs
    for i, ref in enumerate(refs):
        assert len(out[i]) == 20
        for got in out[i]:
            check_stats(got, ref)


@pytest.mark.parametrize("n", [4, 1020, 4096, 4100, 1 << 22, (1 << 22) + 12])
def test_bench_copy_probe_copies(ctx, n):
    """bench.py's streaming probe is a real copy (whole 16 KiB workgroup steps and the ragged end)"""
    import torch
    a = torch.arange(n, dtype=torch.float32, device="cuda")
    b = torch.full((n + 8,), -1.0, dtype=torch.float32, device="cuda")
    ctx.bench_copy(a, b[:n])
    ctx.synchronize()
    assert torch.equal(b[:n], a) and bool((b[n:] == -1.0).all())


def test_stats_resident_abort_falls_back_to_the_chain(ctx, oracle, dev_build):
    """A grid barrier of the one-kernel engine that cannot complete raises an abort flag; the host then runs the chain.  The test
    hook raises the flag before the launch: statistics and stretched bytes must still be the oracle's."""
    import torch
    rng = np.random.default_rng(5)
    img = sky_image(rng, 2048, 2052)
    st = oracle.compute_image_stats(img)
    want = oracle.apply_stf(img, oracle.auto_stf(st), st)
    old = {k: os.environ.get(k) for k in ("AB_STATS_FORCE_ABORT", "AB_STATS_CHAIN")}
    os.environ["AB_STATS_FORCE_ABORT"] = "1"
    os.environ["AB_STATS_CHAIN"] = "0"
    try:
        check_stats(ctx.compute_image_stats(img), st)
        u8, gst, _ = ctx.auto_stretch_preview(torch.from_numpy(img).cuda())
        check_stats(gst, st)
        assert np.array_equal(u8.cpu().numpy(), want)
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


@pytest.mark.parametrize("barrier", [1, 2, 3, 4, 5, 6, 7])
def test_stats_resident_timeout_after_arrival_at_any_barrier(ctx, oracle, barrier, dev_build):
    """A workgroup that times out at a barrier AFTER publishing its arrival there lets its peers pass.  At the LAST barrier of a
    launch they then run to the end without it, and with the completion marker written by workgroup 0 alone (round 3) the host took
    a preview with an unwritten tile for complete.  The hook makes the last workgroup do exactly that at its `barrier`-th barrier
    (a launch has five to seven; a hook beyond the last one does nothing): statistics and EVERY stretched byte must be the oracle's.
    The output buffer is poisoned first, so that a tile nobody wrote cannot pass by luck."""
    import torch
    rng = np.random.default_rng(50 + barrier)
    img = sky_image(rng, 2048, 2052)
    st = oracle.compute_image_stats(img)
    want = oracle.apply_stf(img, oracle.auto_stf(st), st)
    old = {k: os.environ.get(k) for k in ("AB_STATS_FORCE_ABORT", "AB_STATS_CHAIN")}
    os.environ["AB_STATS_FORCE_ABORT"] = f"b{barrier}"
    os.environ["AB_STATS_CHAIN"] = "0"
    try:
        dev = torch.from_numpy(img).cuda()
        out = torch.full(img.shape, 0x5a, dtype=torch.uint8, device="cuda")
        u8, gst, _ = ctx.auto_stretch_preview(dev, out=out)
        check_stats(gst, st)
        assert np.array_equal(u8.cpu().numpy(), want)
        os.environ["AB_STATS_FORCE_ABORT"] = "0"               # and the context recovers: the next launch is a clean resident one
        out.fill_(0x5a)
        u8, gst, _ = ctx.auto_stretch_preview(dev, out=out)
        check_stats(gst, st)
        assert np.array_equal(u8.cpu().numpy(), want)
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


def test_stats_resident_soak_alternating_images_under_load(oracle):
    """The resident engine's grid barrier carries data between workgroups of different XCDs through device-scope stores and
    loads at addresses that every launch reuses.  500 previews on ONE context, alternating two images whose statistics differ
    (a stale slab row or partial from the launch before would be the OTHER image's), while a second context keeps the chip busy
    with warps on its own stream: every result must equal the chain's."""
    import threading
    import torch
    import astroburst_amd as ab
    rng = np.random.default_rng(123)
    imgs = [sky_image(rng, 2048, 2052), (sky_image(rng, 2048, 2052, pad=False) * np.float32(1.7) + np.float32(333.0))]
    old = os.environ.get("AB_STATS_CHAIN")
    main, load = ab.Context(0), ab.Context(0)
    try:
        devs = [torch.from_numpy(im).cuda() for im in imgs]
        os.environ["AB_STATS_CHAIN"] = "1"
        want = []
        for d in devs:
            u8, st, _ = main.auto_stretch_preview(d)
            want.append((u8.cpu().numpy().copy(), st))
        assert want[0][1].median != want[1][1].median
        os.environ["AB_STATS_CHAIN"] = "0"
        stop, errs = threading.Event(), []

        def churn():
            try:
                src = torch.from_numpy(imgs[0]).cuda()
                dst = torch.empty_like(src)
                while not stop.is_set():
                    load.warp_image(src, (1.0, 0.0005, 0.3, -0.0005, 1.0, -0.2), src.shape[0], src.shape[1], out=dst)
                    load.synchronize()
            except Exception as e:      # noqa: BLE001
                errs.append(e)

        th = threading.Thread(target=churn)
        th.start()
        try:
            out = torch.empty(imgs[0].shape, dtype=torch.uint8, device="cuda")
            for it in range(500):
                k = it & 1
                u8, st, _ = main.auto_stretch_preview(devs[k], out=out)
                check_stats(st, want[k][1], exact_mean=True)
                assert st.mean == want[k][1].mean
                if it % 25 == 0 or it >= 490:
                    assert np.array_equal(u8.cpu().numpy(), want[k][0]), it
        finally:
            stop.set()
            th.join()
        assert not errs, errs
    finally:
        main.close()
        load.close()
        if old is None:
            os.environ.pop("AB_STATS_CHAIN", None)
        else:
            os.environ["AB_STATS_CHAIN"] = old
