"""GPU parity: star detection / segmentation (star_detection.rs) and affine registration (affine.rs).

Bar: the segmentation is integer work -- star count, order, npix and the tile background statistics are
exact; centroids / fluxes / shape moments are f64 sums whose order differs from the reference's BFS
order (1e-12 relative); the RANSAC geometry is the same host code on the same star lists, so the
transform is compared at 1e-9."""
import math

import numpy as np
import pytest

from test_oracle_detect_cases import make_test_image

pytestmark = pytest.mark.gpu


def compare_stars(got, ref):
    assert len(got) == len(ref)
    for g, r in zip(got, ref):
        assert g.npix == r.npix
        for f in ("x", "y", "flux", "fwhm", "peak", "snr"):
            a, b = getattr(g, f), getattr(r, f)
            assert abs(a - b) <= 1e-10 * max(1.0, abs(b)), (f, a, b)
        assert abs(g.eccentricity - r.eccentricity) <= 1e-6


def test_reference_cases(ctx):                                      # star_detection.rs:289-328
    stars, med, sig = ctx.detect_stars(make_test_image(300, 300), 5.0)
    assert len(stars) >= 3 and sig > 0.0 and stars[0].flux >= stars[1].flux
    assert abs(stars[0].x - 150.0) < 2.0 and abs(stars[0].y - 200.0) < 2.0
    assert ctx.detect_stars(np.full((100, 100), 50.0, np.float32), 5.0)[0] == []
    med, sig = ctx.estimate_background(np.full((200, 200), 100.0, np.float32), 64)
    assert abs(med - 100.0) < 1.0 and sig < 1.0
    assert ctx.detect_stars(np.ones((2, 50), np.float32), 5.0) == ([], 0.0, 1.0)


@pytest.mark.parametrize("shape,tile", [((300, 300), 37), ((200, 513), 64), ((1000, 700), 125), ((64, 64), 16), ((257, 300), 256)])
def test_estimate_background_exact(ctx, oracle, shape, tile):
    rng = np.random.default_rng(shape[0] + tile)
    img = rng.normal(500, 20, shape).astype(np.float32)
    img[rng.random(shape) < 0.01] *= 30
    img[:10, :] = 0.0
    img[20:23, 30:90] = np.nan
    img[50:60, 50:60] = 1e-8
    assert ctx.estimate_background(img, tile) == oracle.estimate_background(img, tile)


@pytest.mark.parametrize("rows,cols", [(500, 640), (333, 257), (301, 515)])   # 4 | pixels: 16-byte threshold pass; odd: scalar pass
@pytest.mark.parametrize("seed,sigma", [(1, 5.0), (2, 3.5), (3, 8.0)])
def test_detect_stars_matches_oracle(ctx, oracle, seed, sigma, rows, cols):
    from astroburst_amd import synth
    y, x, flux = synth.star_catalog(rows, cols, 250, seed=seed)
    img = synth.make_frame(rows, cols, seed, cat=(y, x, flux * 20.0), bad_patch_rate=1e-5).numpy()
    img[:, 0] += 5000.0                                              # a bright border column: growth may enter the border,
    img[0, 100:140] += 5000.0                                        # but border-only components are never seeded
    got, gm, gs = ctx.detect_stars(img, sigma)
    ref, rm, rs = oracle.detect_stars(img, sigma)
    assert (gm, gs) == (rm, rs)
    assert len(ref) > 20
    compare_stars(got, ref)


def test_detect_with_a_large_bright_region(ctx, oracle):
    """40 % of the frame above threshold in one slab: the threshold pass fills and flushes its LDS list several times per
    block, the slab itself is one component far beyond 5000 pixels (dropped), the stars elsewhere survive"""
    from astroburst_amd import synth
    rows, cols = 512, 640
    y, x, flux = synth.star_catalog(rows, cols, 120, seed=8)
    img = synth.make_frame(rows, cols, 4, cat=(y, x, flux * 20.0), bad_patch_rate=0.0).numpy()
    img[:200, :] += 3000.0
    got, gm, gs = ctx.detect_stars(img, 5.0)
    ref, rm, rs = oracle.detect_stars(img, 5.0)
    assert (gm, gs) == (rm, rs) and len(ref) > 10
    compare_stars(got, ref)


def test_detect_touching_blobs_and_size_limits(ctx, oracle):
    img = np.full((200, 200), 100.0, np.float32) + np.random.default_rng(0).normal(0, 1, (200, 200)).astype(np.float32)
    img[50:52, 50] += 500.0                                          # 2 px: below the 3 px minimum
    img[80:160, 20:100] += 500.0                                     # 6400 px: above the 5000 px maximum
    img[20, 20] += 500.0; img[21, 21] += 500.0; img[22, 20] += 500.0  # diagonal links: one 8-connected component
    img[100:104, 150:154] += 800.0
    img[104, 154] += 800.0                                           # corner-touching pixel joins the 4x4 block
    got, _, _ = ctx.detect_stars(img, 5.0)
    ref, _, _ = oracle.detect_stars(img, 5.0)
    compare_stars(got, ref)
    assert sorted(s.npix for s in got) == [3, 17]


def test_component_shapes_through_every_moments_path(ctx, oracle):
    """comp_moments (round 4) walks a component's box as a lane patch of 8 / 16 / 32 / 64 columns, keeps boxes of up to eight trips
    in registers and walks larger ones eight trips at a time, four components per wave: one frame holding every shape -- a wide
    thin bar (> 64 columns: several column blocks), a tall one (more than eight trips of an 8-column patch), a 44 x 44 block
    (44 trips of a 64-column patch) and a 70 x 70 one, an L whose box is mostly empty, a ring (the box's centre is not a member),
    neighbours whose boxes overlap, and ordinary stars in between"""
    from astroburst_amd import synth
    rows, cols = 600, 720
    y, x, flux = synth.star_catalog(rows, cols, 150, seed=21)
    img = synth.make_frame(rows, cols, 2, cat=(y, x, flux * 20.0), bad_patch_rate=0.0, cosmic_rate=0.0).numpy()
    ramp = np.linspace(400.0, 900.0, 200, dtype=np.float32)
    img[40:43, 100:300] += ramp                                       # 200 x 3: wide, flux gradient along it
    img[100:260, 30:34] += ramp[:160, None]                           # 4 x 160: tall
    img[300:344, 400:444] += 600.0 + 3.0 * np.arange(44, dtype=np.float32)[None, :]   # 44 x 44: about the largest box whose FWHM passes (:162)
    img[380:450, 400:470] += 600.0                                    # 70 x 70 = 4900 px: measured, then dropped by its FWHM
    img[450:520, 100:104] += 700.0                                    # an L: 70 x 4 ...
    img[516:520, 100:190] += 700.0                                    # ... and 4 x 90
    yy, xx = np.mgrid[0:41, 0:41]
    ring = (np.hypot(yy - 20, xx - 20) > 14) & (np.hypot(yy - 20, xx - 20) < 19)
    img[200:241, 500:541][ring] += 800.0
    img[420:426, 600:606] += 500.0                                    # two 6 x 6 blocks one pixel apart on the diagonal:
    img[427:433, 607:613] += 500.0                                    # separate components with overlapping neighbourhoods
    for sigma in (5.0, 3.5):
        got, gm, gs = ctx.detect_stars(img, sigma)
        ref, rm, rs = oracle.detect_stars(img, sigma)
        assert (gm, gs) == (rm, rs) and len(ref) > 40
        compare_stars(got, ref)
    assert sum(s.npix > 400 for s in got) >= 2   # the block and the ring are stars; the bars and the 70 x 70 block fail the FWHM test
    # ... and the same frame through the grouped chain of a registration batch (comp_moments_many): every target of a batch of
    # five equals the pairwise call (the group path's records travel through pinned memory, 72 or 8 bytes each)
    import torch
    tgts = [torch.from_numpy(np.roll(img, (k, -k), axis=(0, 1)).copy()).cuda() for k in range(1, 6)]
    ref_t = torch.from_numpy(img).cuda()
    batch = ctx.register_frames(ref_t, tgts, num_threads=8)
    for t, b in zip(tgts, batch):
        one = ctx.align_channel_affine(ref_t, t, num_threads=8)
        assert (b.method, b.inliers, b.matched_stars) == (one.method, one.inliers, one.matched_stars) and b.transform == one.transform


def test_normalize_for_detection(ctx, oracle):
    rng = np.random.default_rng(2)
    img = rng.normal(1000, 30, (300, 400)).astype(np.float32)
    img[5, 5], img[6, 6] = np.nan, np.inf
    assert np.array_equal(ctx.normalize_for_detection(img), oracle.normalize_for_detection(img), equal_nan=True)
    small = np.arange(50, dtype=np.float32).reshape(5, 10)
    assert np.array_equal(ctx.normalize_for_detection(small), small)


@pytest.mark.parametrize("shape", [(300, 300), (300, 400), (1024, 1024), (2048, 2048)])
@pytest.mark.parametrize("kind", ["sky", "signed", "ties", "wide", "holes", "flat", "few"])
def test_normalize_percentiles_on_the_gpu(ctx, oracle, shape, kind):
    """The 1 % / 99.9 % order statistics of the subsample come from a one-workgroup radix select (registers for <= 102 400
    samples: 300 x 300 and 2048 x 2048 here; a buffer above that: 300 x 400, 1024 x 1024).  Same normalised frame as the
    oracle's sorted subsample, bit for bit, for negative values, ties, thirty binades, NaN / inf holes, a flat frame
    (range < 1e-15: returned unchanged) and a frame with fewer than 100 finite samples (unchanged)."""
    import zlib
    rng = np.random.default_rng(zlib.crc32(repr((shape, kind)).encode()))
    n = shape[0] * shape[1]
    if kind == "sky":
        img = rng.normal(1000, 30, n)
    elif kind == "signed":
        img = rng.normal(0, 5, n)
        img[::7] = -0.0
        img[3::11] = 0.0
    elif kind == "ties":
        img = np.round(rng.normal(100, 3, n))
    elif kind == "wide":
        img = 10.0 ** rng.uniform(-20, 20, n) * rng.choice([-1.0, 1.0], n)
    elif kind == "holes":
        img = rng.normal(50, 10, n)
        img[rng.random(n) < 0.3] = np.nan
        img[rng.random(n) < 0.01] = np.inf
        img[rng.random(n) < 0.01] = -np.inf
    elif kind == "flat":
        img = np.full(n, 7.25)
    else:
        img = np.full(n, np.nan)
        img[rng.choice(n, 60, replace=False)] = rng.normal(5, 1, 60)
    img = img.astype(np.float32).reshape(shape)
    with np.errstate(all="ignore"):
        want = oracle.normalize_for_detection(img)
    assert np.array_equal(ctx.normalize_for_detection(img), want, equal_nan=True)


def test_affine_from_stars_matches_oracle(ctx, oracle):
    rng = np.random.default_rng(1)
    ref = np.column_stack([rng.uniform(20, 1180, 90), rng.uniform(20, 980, 90)])
    ang = math.radians(-0.8)
    c, s = math.cos(ang), math.sin(ang)
    tgt = np.column_stack([c * ref[:, 0] - s * ref[:, 1] - 21.0, s * ref[:, 0] + c * ref[:, 1] + 6.5]) + rng.normal(0, 0.1, ref.shape)
    tgt = tgt[rng.permutation(90)][:70]                               # 20 stars missing in the target
    for nt in (1, 4, 8, 13):
        g = ctx.affine_from_stars(ref, tgt, 1000, 1200, num_threads=nt)
        r = oracle.affine_from_stars(ref, tgt, 1000, 1200, num_threads=nt)
        assert g is not None and r is not None
        assert g.method == r.method and g.matched_stars == r.matched_stars and g.inliers == r.inliers
        assert np.allclose(g.transform, r.transform, rtol=0, atol=1e-9) and abs(g.residual_px - r.residual_px) < 1e-9
    assert ctx.affine_from_stars(ref[:3], tgt[:3], 1000, 1200) is None


def test_align_channel_affine_end_to_end(ctx, oracle):
    from astroburst_amd import synth
    rows, cols = 600, 800
    y, x, flux = synth.star_catalog(rows, cols, 400, seed=5)
    cat = (y, x, flux * 30.0)
    ref = synth.make_frame(rows, cols, 0, cat=cat, bad_patch_rate=0.0, cosmic_rate=0.0).numpy()
    tgt = synth.make_frame(rows, cols, 1, cat=cat, shift=(3.0, -2.0), bad_patch_rate=0.0, cosmic_rate=0.0).numpy()
    g = ctx.align_channel_affine(ref, tgt, num_threads=8)
    r = oracle.align_channel_affine(ref, tgt, num_threads=8)
    assert g.method == r.method and g.method in ("affine", "rigid")
    assert g.matched_stars == r.matched_stars and g.inliers == r.inliers
    assert np.allclose(g.transform, r.transform, rtol=0, atol=1e-8)
    # and the estimated transform registers the frame: warp the target with it and compare star positions
    assert abs(g.transform[2] + 2.0) < 0.3 and abs(g.transform[5] - 3.0) < 0.3


def test_align_channel_affine_falls_back(ctx, oracle):
    rng = np.random.default_rng(9)
    from scipy.ndimage import gaussian_filter
    base = gaussian_filter(rng.standard_normal((500, 500)), 3.0).astype(np.float32) * 100 + 1000   # no stars at all
    ref, tgt = base[10:410, 10:410].copy(), base[14:414, 7:407].copy()
    g, r = ctx.align_channel_affine(ref, tgt), oracle.align_channel_affine(ref, tgt)
    assert g.method == r.method and g.method in ("phase_correlation", "identity")
    assert g.transform == r.transform


def test_register_frames_equals_pairwise_calls(ctx, oracle):
    """ab_register_frames shares the reference's detection / triangle table across targets: results must be
    identical to one align_channel_affine call per target, and match the oracle pair by pair."""
    import torch
    from astroburst_amd import synth
    rows, cols = 640, 768
    y, x, flux = synth.star_catalog(rows, cols, 500, seed=11)
    cat = (y, x, flux * 30.0)
    ref = synth.make_frame(rows, cols, 0, cat=cat, bad_patch_rate=0.0, cosmic_rate=0.0)
    shifts = [(3.0, -2.0), (-4.5, 6.25), (0.0, 0.0), (7.75, 1.5)]
    tgts = [synth.make_frame(rows, cols, k + 1, cat=cat, shift=s, bad_patch_rate=0.0, cosmic_rate=0.0) for k, s in enumerate(shifts)]
    flat = torch.full((rows, cols), 1000.0)                          # no stars: falls through to phase correlation / identity
    batch = ctx.register_frames(ref.cuda(), [t.cuda() for t in tgts] + [flat.cuda()], num_threads=8)
    assert len(batch) == 5
    for t, b, s in zip(tgts + [flat], batch, shifts + [None]):
        single = ctx.align_channel_affine(ref.numpy(), t.numpy(), num_threads=8)
        assert (b.method, b.matched_stars, b.inliers, b.transform, b.residual_px) == \
               (single.method, single.matched_stars, single.inliers, single.transform, single.residual_px)
        want = oracle.align_channel_affine(ref.numpy(), t.numpy(), num_threads=8)
        assert b.method == want.method and b.matched_stars == want.matched_stars and b.inliers == want.inliers
        assert np.allclose(b.transform, want.transform, rtol=0, atol=1e-8)
        if s is not None:
            assert b.method in ("affine", "rigid")
            assert abs(b.transform[2] - s[1]) < 0.3 and abs(b.transform[5] - s[0]) < 0.3   # output -> source mapping
    assert ctx.register_frames(ref.cuda(), []) == []


def test_align_pairs_affine_equals_estimate_then_warp(ctx, oracle):
    """ab_align_pairs_affine = align_pair(.., Affine) per target (pair.rs:41-77): the transforms of register_frames and
    warp_image(target, transform) planes, bit for bit."""
    import torch
    from astroburst_amd import synth
    rows, cols = 512, 640
    y, x, flux = synth.star_catalog(rows, cols, 400, seed=13)
    cat = (y, x, flux * 30.0)
    ref = synth.make_frame(rows, cols, 0, cat=cat, bad_patch_rate=0.0, cosmic_rate=0.0).cuda()
    tgts = [synth.make_frame(rows, cols, k + 1, cat=cat, shift=s, bad_patch_rate=0.0, cosmic_rate=0.0).cuda()
            for k, s in enumerate([(2.5, -1.0), (-3.0, 4.0), (0.5, 0.25)])]
    outs = [torch.empty_like(t) for t in tgts]
    res = ctx.align_pairs_affine(ref, tgts, outs, num_threads=8)
    again = ctx.register_frames(ref, tgts, num_threads=8)
    for r, a, t, o in zip(res, again, tgts, outs):
        assert (r.method, r.transform, r.inliers) == (a.method, a.transform, a.inliers)
        want = oracle.warp_image(t.cpu().numpy(), r.transform, rows, cols)
        assert np.array_equal(o.cpu().numpy(), want)


@pytest.mark.parametrize("n_targets", [2, 7])
def test_align_pairs_affine_takes_host_frames(ctx, oracle, n_targets):
    """The application holds its frames on the host (calibration.rs:306-315): ab_align_pairs_affine uploads on_device = 0 frames on
    its own stream and registers each as it lands.  Transforms and warped planes equal the device-resident call's bit for bit, for
    pinned tensors, pageable numpy arrays, a mix of both with device frames, and a host reference; 2 targets take the
    frame-by-frame path, 7 the grouped one (two groups, the second short)."""
    import torch
    from astroburst_amd import synth
    rows, cols = 448, 576
    y, x, flux = synth.star_catalog(rows, cols, 350, seed=21)
    cat = (y, x, flux * 30.0)
    ref = synth.make_frame(rows, cols, 0, cat=cat, bad_patch_rate=0.0, cosmic_rate=0.0)
    shifts = [(2.5, -1.0), (-3.0, 4.0), (0.5, 0.25), (6.0, 2.0), (-1.25, -2.5), (0.0, 3.5), (4.0, -4.0)][:n_targets]
    tgts = [synth.make_frame(rows, cols, k + 1, cat=cat, shift=s, bad_patch_rate=0.0, cosmic_rate=0.0) for k, s in enumerate(shifts)]
    if n_targets > 2:
        tgts[2] = torch.full((rows, cols), 1000.0)   # no stars: the phase-correlation fallback reads the staged copy as well
    dev = [t.cuda() for t in tgts]
    want_out = [torch.empty_like(t) for t in dev]
    want = ctx.align_pairs_affine(ref.cuda(), dev, want_out, num_threads=8)
    o0 = oracle.align_channel_affine(ref.numpy(), tgts[0].numpy(), num_threads=8)
    assert want[0].method == o0.method and np.allclose(want[0].transform, o0.transform, rtol=0, atol=1e-8)
    variants = {
        "pinned": (ref.cuda(), [t.pin_memory() for t in tgts]),
        "pageable numpy": (ref.cuda(), [t.numpy() for t in tgts]),
        "mixed": (ref.cuda(), [t.pin_memory() if k % 2 else d for k, (t, d) in enumerate(zip(tgts, dev))]),
        "host reference": (ref.numpy(), [t.pin_memory() for t in tgts]),
    }
    for name, (r, ts) in variants.items():
        for rep in range(2):   # the second call reuses the staging area and the events
            outs = [torch.full_like(d, -1.0) for d in dev]
            got = ctx.align_pairs_affine(r, ts, outs, num_threads=8)
            for g, w, o, wo in zip(got, want, outs, want_out):
                assert (g.method, g.transform, g.inliers, g.matched_stars) == (w.method, w.transform, w.inliers, w.matched_stars), name
                assert torch.equal(o.isnan(), wo.isnan()) and torch.equal(o.nan_to_num(), wo.nan_to_num()), name


def test_host_frames_cancel_errors_and_degenerate_sets(ctx, oracle):
    """The host-frame path's edges: a cancel requested before the call stops it with the reference's error and leaves nothing in
    flight (the staging copies read the caller's memory: the call drains them before it returns) and the context usable; planes too
    small for background tiles take the percentile path after the uploads; a host reference with no target; mismatched dims."""
    import torch
    from astroburst_amd import AstroBurstError, _lib, synth
    rows, cols = 320, 384
    y, x, flux = synth.star_catalog(rows, cols, 220, seed=5)
    cat = (y, x, flux * 30.0)
    ref = synth.make_frame(rows, cols, 0, cat=cat, bad_patch_rate=0.0, cosmic_rate=0.0)
    tgts = [synth.make_frame(rows, cols, k + 1, cat=cat, shift=(1.5 * k, -0.75 * k), bad_patch_rate=0.0, cosmic_rate=0.0).pin_memory() for k in range(5)]
    outs = [torch.empty((rows, cols), device="cuda") for _ in tgts]
    want = ctx.align_pairs_affine(ref.cuda(), [t.cuda() for t in tgts], [torch.empty_like(o) for o in outs], num_threads=8)
    ctx.request_cancel()
    try:
        with pytest.raises(AstroBurstError) as e:
            ctx.align_pairs_affine(ref.cuda(), tgts, outs, num_threads=8)
        assert e.value.code == _lib.AB_ERR_CANCELLED
    finally:
        ctx.clear_cancel()
    for t in tgts:
        t.zero_()            # nothing may still be reading these ...
    for t, k in zip(tgts, range(5)):
        t.copy_(synth.make_frame(rows, cols, k + 1, cat=cat, shift=(1.5 * k, -0.75 * k), bad_patch_rate=0.0, cosmic_rate=0.0))
    got = ctx.align_pairs_affine(ref.cuda(), tgts, outs, num_threads=8)   # ... and the context works as before
    assert [(g.method, g.transform) for g in got] == [(w.method, w.transform) for w in want]
    assert ctx.align_pairs_affine(ref.numpy(), [], [], num_threads=8) == []
    tiny_ref = np.arange(4, dtype=np.float32).reshape(2, 2)
    tiny = [np.ones((2, 2), np.float32) * k for k in range(1, 6)]
    touts = [torch.empty((2, 2), device="cuda") for _ in tiny]
    res = ctx.align_pairs_affine(torch.from_numpy(tiny_ref).cuda(), tiny, touts, num_threads=8)
    dres = ctx.align_pairs_affine(torch.from_numpy(tiny_ref).cuda(), [torch.from_numpy(t).cuda() for t in tiny], [torch.empty_like(o) for o in touts], num_threads=8)
    assert [(r.method, r.transform) for r in res] == [(r.method, r.transform) for r in dres]
    with pytest.raises(AstroBurstError, match="differ from the reference's dims"):
        ctx.align_pairs_affine(ref.cuda(), [np.zeros((rows, cols + 1), np.float32)], [torch.empty((rows, cols + 1), device="cuda")], num_threads=8)


def test_fed_pipeline_equals_upfront_percentiles(tmp_path, dev_build):
    """Round 4: frames arriving from the host go through a pipeline whose percentiles run chunk by chunk on the device
    (ab_bg_pipeline_begin_fed); AB_PIPE_FED=1 sends device-resident frames the same way.  The default order (all frames'
    percentiles up front, a host join, then the tiles) and the fed one give the same transforms, bit for bit."""
    import os
    import subprocess
    import sys
    from astroburst_amd import synth
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    y, x, flux = synth.star_catalog(768, 768, 400, seed=5)
    cat = (y, x, flux * 25.0)
    frames = [synth.make_frame(768, 768, k, cat=cat, device="cuda", shift=(0.7 * k, -1.1 * k)).cpu().numpy() for k in range(7)]
    np.save(tmp_path / "frames.npy", np.stack(frames))
    code = (
        "import sys, json\n"
        "sys.path.insert(0, %r)\n"
        "import numpy as np, torch\n"
        "import astroburst_amd as ab\n"
        "ctx = ab.Context(0)\n"
        "f = [torch.from_numpy(a).cuda() for a in np.load(%r)]\n"
        "res = ctx.register_frames(f[0], f[1:], 8)\n"
        "print('RESULT', json.dumps([[float(v).hex() for v in r.transform] + [str(r.method), int(r.inliers)] for r in res]))\n"
    ) % (root, str(tmp_path / "frames.npy"))
    out = []
    for fed in (False, True):
        env = dict(os.environ)
        env.pop("AB_PIPE_FED", None)
        if fed:
            env["AB_PIPE_FED"] = "1"
        r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
        line = [ln for ln in r.stdout.splitlines() if ln.startswith("RESULT")]
        assert r.returncode == 0 and line, r.stdout[-2000:] + r.stderr[-3000:]
        out.append(line[0])
    assert out[0] == out[1]
    assert out[0].count("affine") == 6


def test_chained_detection_gives_the_same_transform(tmp_path, dev_build):
    """AB_DETECT_CHAIN=1 (read once per process): percentiles -> tiles -> background -> threshold -> labels enqueued back to back
    with the parameters travelling through device memory.  Same registration result as the default path, bit for bit.  (The two
    child processes load the SAME frames from disk: rendering them again would not do, the star renderer accumulates with
    atomics and its frames differ in the last bit from process to process.)"""
    import os
    import subprocess
    import sys
    from astroburst_amd import synth
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    y, x, flux = synth.star_catalog(1024, 1024, 400, seed=3)
    cat = (y, x, flux * 25.0)
    np.save(tmp_path / "ref.npy", synth.make_frame(1024, 1024, 0, cat=cat, device="cuda").cpu().numpy())
    np.save(tmp_path / "tgt.npy", synth.make_frame(1024, 1024, 1, cat=cat, device="cuda", shift=(3.3, -5.1)).cpu().numpy())
    code = (
        "import sys, json\n"
        "sys.path.insert(0, %r)\n"
        "import numpy as np\n"
        "import astroburst_amd as ab\n"
        "ctx = ab.Context(0)\n"
        "r = ctx.align_channel_affine(np.load(%r), np.load(%r), 8)\n"
        "print('RESULT', json.dumps([float(v).hex() for v in r.transform] + [str(r.method), int(r.inliers)]))\n"
    ) % (root, str(tmp_path / "ref.npy"), str(tmp_path / "tgt.npy"))
    out = []
    for chain in (False, True):
        env = dict(os.environ)
        env.pop("AB_DETECT_CHAIN", None)
        if chain:
            env["AB_DETECT_CHAIN"] = "1"
        r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
        line = [ln for ln in r.stdout.splitlines() if ln.startswith("RESULT")]
        assert r.returncode == 0 and line, r.stdout[-2000:] + r.stderr[-3000:]
        out.append(line[0])
    assert out[0] == out[1]
    assert "affine" in out[0]


def _crowded_frames(kind, rows, cols, shifts, seed):
    """Frames whose brightest several hundred components do NOT become the matcher's stars:
    "spikes": 700 components of three pixels with all their flux in the middle one (FWHM < 0.5: star_detection.rs:160-163 drops them),
    "clusters": 130 crosses of five 3-pixel components whose centroids sit 2.4 - 2.8 px from the middle one (the 3 px dedup, :217-248, keeps
    one in five).  Behind them ~260 ordinary, fainter stars.  The device-side selection of the brightest 480 candidates
    (comp_select_many_kernel) cannot yield 120 stars from either: the frame goes through the full path again."""
    from astroburst_amd import synth
    rng = np.random.default_rng(seed)
    y, x, flux = synth.star_catalog(rows, cols, 260, seed=seed)
    cat = (y, x, flux * 12.0)
    ny, nx = (rows - 120) // 12, (cols - 120) // 12
    cells = rng.permutation(ny * nx)[:700]
    cy, cx = 20 + 12 * (cells // nx), 20 + 12 * (cells % nx)
    amp = rng.uniform(30000.0, 60000.0, cells.size)
    frames = []
    for k, (dy, dx) in enumerate(shifts):
        f = synth.make_frame(rows, cols, k, cat=cat, shift=(float(dy), float(dx)), bad_patch_rate=0.0, cosmic_rate=0.0).numpy().copy()
        oy, ox = int(round(dy)), int(round(dx))
        # a saturated 40 x 40 patch: more than 0.1 % of the pixels, so normalize_for_detection's 99.9th percentile (affine.rs:24-53)
        # sits on it and nothing else is clamped -- the components below keep their distinct fluxes
        f[rows - 70 + oy:rows - 30 + oy, cols - 75 + ox:cols - 35 + ox] = 65000.0
        if kind == "spikes":
            for j in range(cells.size):
                yy, xx = cy[j] + oy, cx[j] + ox
                f[yy, xx] += amp[j]
                f[yy, xx - 1] += 0.004 * amp[j]
                f[yy, xx + 1] += 0.004 * amp[j]
        else:
            # five 3-pixel components: the middle one and four whose centroids sit 2.4 - 2.8 px from its centroid, one empty pixel apart
            parts = (([(0, 0), (0, 1), (1, 0)], 1.0), ([(0, -2), (1, -2), (1, -3)], 0.99), ([(0, 3), (1, 3), (2, 3)], 0.98),
                     ([(-2, 0), (-2, 1), (-2, 2)], 0.97), ([(3, -1), (3, 0), (3, 1)], 0.96))
            for j in range(130):
                yy, xx = cy[j] + oy, cx[j] + ox
                for pix, a in parts:
                    for (qy, qx) in pix:
                        f[yy + qy, xx + qx] += a * amp[j]
        frames.append(np.ascontiguousarray(f))
    return frames


@pytest.mark.parametrize("kind", ["spikes", "clusters"])
def test_registration_when_the_brightest_components_are_not_stars(ctx, oracle, kind):
    """VERDICT r4 item 1a's gate: the grouped registration sends the brightest 480 candidates per frame; when those do not hold 120
    stars the whole list decides (the full path).  Five targets (one group of four + one) must equal the frame-by-frame calls --
    which always take every record -- exactly, and the oracle pair by pair."""
    import torch
    rows, cols = 900, 1100
    shifts = [(0, 0), (3, -2), (-4, 6), (0, 0), (7, 1), (-2, -5)]
    frames = _crowded_frames(kind, rows, cols, shifts, seed=31 if kind == "spikes" else 32)
    ref, tgts = frames[0], frames[1:]
    # the premise: the full star list of the reference frame is long and its brightest 480 candidates hold fewer than 120 stars
    stars, _, _ = ctx.detect_stars(ctx.normalize_for_detection(ref), 3.5)
    assert len(stars) >= 120
    batch = ctx.register_frames(torch.from_numpy(ref).cuda(), [torch.from_numpy(t).cuda() for t in tgts], num_threads=8)
    for t, b in zip(tgts, batch):
        single = ctx.align_channel_affine(ref, t, num_threads=8)
        assert (b.method, b.matched_stars, b.inliers, b.transform, b.residual_px) == \
               (single.method, single.matched_stars, single.inliers, single.transform, single.residual_px)
    want = oracle.align_channel_affine(ref, tgts[0], num_threads=8)
    assert batch[0].method == want.method and batch[0].matched_stars == want.matched_stars and batch[0].inliers == want.inliers
    assert np.allclose(batch[0].transform, want.transform, rtol=0, atol=1e-8)


def _same_registration(a, b):
    return (a.method, a.matched_stars, a.inliers, a.transform, a.residual_px) == (b.method, b.matched_stars, b.inliers, b.transform, b.residual_px)


def _holds_to_the_oracle(oracle, ref, tgts, got, atol=1e-8):
    """every target's grouped result against oracle.align_channel_affine of the pair (detection -> matching -> RANSAC -> fit)"""
    ref_h = ref.cpu().numpy()
    for t, g in zip(tgts, got):
        want = oracle.align_channel_affine(ref_h, t.cpu().numpy(), num_threads=8)
        assert g.method == want.method and g.matched_stars == want.matched_stars and g.inliers == want.inliers, (g, want)
        assert np.allclose(g.transform, want.transform, rtol=0, atol=atol)


def _grouped_case(shape):
    from astroburst_amd import synth
    rows, cols = shape
    y, x, flux = synth.star_catalog(rows, cols, int(900 * rows * cols / 1e6) + 150, seed=rows)
    cat = (y, x, flux * 30.0)
    ref = synth.make_frame(rows, cols, 0, cat=cat, bad_patch_rate=0.0, cosmic_rate=1e-4).cuda()
    tgts = [synth.make_frame(rows, cols, k + 1, cat=cat, shift=(1.25 * k - 4.0, 3.0 - 0.8 * k), bad_patch_rate=1e-6, cosmic_rate=1e-4).cuda()
            for k in range(9)]
    return ref, tgts


@pytest.mark.parametrize("shape", [(512, 640), (600, 800), (257, 1000), (1100, 2048), (300, 1003), (515, 777)])
def test_grouped_registration_equals_the_oracle(ctx, oracle, shape):
    """How a GROUP of frames is detected (tile-local union-find in LDS + a border pass, at any width: the mask's rows are padded to
    whole words and a row's ragged last quad is loaded float by float -- 1000, 1003 and 777 are not multiples of 32, the last two
    not of 4 either, so their rows start at every dword alignment; one record per tile-local component; the brightest 480 candidates
    selected on the device; moments for those only) against the ORACLE's pairwise align_channel_affine: nine targets = two groups of
    four + one, every transform, star count and inlier count.  (Round 5 held this path to its own superseded forms only.)"""
    ref, tgts = _grouped_case(shape)
    before = ctx.fallback_counts()
    new = ctx.register_frames(ref, tgts, num_threads=8)
    _holds_to_the_oracle(oracle, ref, tgts, new)
    assert sum(a.method in ("affine", "rigid") for a in new) >= 7
    after = ctx.fallback_counts()
    assert after["frames_redone"] == before["frames_redone"], "an ordinary star field must not need the full path"


@pytest.mark.parametrize("shape", [(512, 640), (257, 1000), (515, 777)])
def test_grouped_registration_equals_the_superseded_forms(ctx, ctx_r4_detect, ctx_midjoin, ctx_pixel_list, ctx_pixelwise, shape):
    """DEVELOPER build only (the release library compiles these switches out): round 4's forms (AB_LABEL_LEGACY=1
    AB_DETECT_FULL_RECORDS=1), the mid-join chain, the pixel-list chain and the pixel-by-pixel tile unions give IDENTICAL transforms,
    star counts and inliers.  A self-comparison, kept as a bisecting aid: the evidence is the oracle test above."""
    ref, tgts = _grouped_case(shape)
    new = ctx.register_frames(ref, tgts, num_threads=8)
    old = ctx_r4_detect.register_frames(ref, tgts, num_threads=8)
    mid = ctx_midjoin.register_frames(ref, tgts, num_threads=8)      # (the selection with a host join after the root numbering)
    plist = ctx_pixel_list.register_frames(ref, tgts, num_threads=8)   # (the chain over pixel lists instead of tile-component records)
    pixw = ctx_pixelwise.register_frames(ref, tgts, num_threads=8)     # (tile-local unions pixel by pixel instead of run by run)
    for a, *others in zip(new, old, mid, plist, pixw):
        for b in others:
            assert _same_registration(a, b)


def _big_case(rows, cols, n_targets, seed, gradient=0.0, crowd_in=()):
    """frames whose short side is >= 2048 px (256-px background tiles: the labelling takes the tile pass's candidate lists)"""
    import torch
    from astroburst_amd import synth
    y, x, flux = synth.star_catalog(rows, cols, int(500 * rows * cols / 2048 ** 2), seed=seed)
    ky, kx = np.meshgrid(256.0 * np.arange(1, rows // 256 + 1, 2) - 0.5, 256.0 * np.arange(1, cols // 256 + 1, 3) - 0.5, indexing="ij")   # stars on tile corners
    keep = (ky.ravel() < rows - 8) & (kx.ravel() < cols - 8)
    y = torch.cat([y, torch.from_numpy(ky.ravel()[keep]).to(y.dtype)])
    x = torch.cat([x, torch.from_numpy(kx.ravel()[keep]).to(x.dtype)])
    flux = torch.cat([flux, torch.full((int(keep.sum()),), float(flux.max()) * 0.5, dtype=flux.dtype)])
    cat = (y, x, flux * 30.0)
    ref = synth.make_frame(rows, cols, 0, cat=cat, bad_patch_rate=0.0, cosmic_rate=1e-4)
    tgts = [synth.make_frame(rows, cols, k + 1, cat=cat, shift=(1.25 * k - 3.0, 2.0 - 0.8 * k), bad_patch_rate=1e-6, cosmic_rate=1e-4) for k in range(n_targets)]
    if gradient:                                                     # vignetting: the bright side's tiles sit above the frame's threshold
        ramp = (1.0 + gradient * torch.linspace(0.0, 1.0, cols)).to(ref.dtype)[None, :]
        ref = ref * ramp
        tgts = [t * ramp for t in tgts]
    for k in crowd_in:                                               # > 512 three-pixel components inside one 256 x 256 tile
        for yy in range(300, 420, 4):
            for xx in range(560, 700, 4):
                tgts[k][yy, xx] += 9000.0
                tgts[k][yy, xx + 1] += 7000.0
                tgts[k][yy + 1, xx] += 7000.0
    return ref.cuda(), [t.cuda() for t in tgts]


@pytest.mark.parametrize("rows,cols,gradient", [(2048, 2304, 0.0), (2100, 2500, 0.0), (2048, 2560, 0.03), (2048, 2560, 0.6)])
def test_labelling_from_the_tile_pass_candidate_lists(ctx, oracle, rows, cols, gradient):
    """Round 6: on frames with 256-px background tiles the tile pass leaves, per whole tile, the list of pixels above a
    conservative cut, and the labelling pass works from those lists instead of reading the frame again (label_bgtile_body).  Whole and
    partial tiles (2100 x 2500: the last row and column of tiles are read from the frame), stars on tile corners (four tile-local
    components folded into one), cosmic rays and NaN patches, and a vignetted field whose bright tiles lie above the frame's threshold
    (their lists are not supersets: read from the frame) -- every transform, star count and inlier count equals the ORACLE's."""
    ref, tgts = _big_case(rows, cols, 5, seed=rows + cols, gradient=gradient)
    before = ctx.fallback_counts()
    got = ctx.register_frames(ref, tgts, num_threads=8)
    after = ctx.fallback_counts()
    _holds_to_the_oracle(oracle, ref, tgts, got)
    assert gradient > 0.1 or sum(a.method in ("affine", "rigid") for a in got) >= 4
    # (a gradient of a few sigma puts whole tiles at the threshold: thousands of one-pixel components per tile overflow the record
    # slots -- of these tiles as of round 5's 32 x 128 ones -- and the frame is redone through the full path: slower, same result)
    assert gradient > 0.0 or after["frames_redone"] == before["frames_redone"]
    dense = after["label_tiles_dense"] - before["label_tiles_dense"]
    tiles = ((rows + 255) // 256) * ((cols + 255) // 256)
    partial = tiles - (rows // 256) * (cols // 256)
    if gradient == 0.0:
        assert dense == 6 * partial, (dense, partial)               # reference + 5 targets: only the partial tiles are read from the frame
    elif after["frames_redone"] == before["frames_redone"]:
        assert 6 * partial < dense < 6 * tiles, (dense, tiles)      # some, not all: the bright side


def test_crowded_background_tile_is_redone_in_full(ctx, oracle):
    """more than 512 components in one 256 x 256 tile: the tile raises the frame's overflow flag, the host redoes that frame through the
    full path and counts it; the result equals the oracle's"""
    ref, tgts = _big_case(2048, 2304, 5, seed=99, crowd_in=(1, 4))
    before = ctx.fallback_counts()
    got = ctx.register_frames(ref, tgts, num_threads=8)
    after = ctx.fallback_counts()
    _holds_to_the_oracle(oracle, ref, tgts, got)
    assert after["tile_slots"] - before["tile_slots"] == 2 and after["frames_redone"] - before["frames_redone"] == 2, (before, after)


def test_crowded_tiles_and_stars_on_tile_corners(ctx, oracle):
    """The records form of the tile labelling (one record per tile-local component, 64 slots per 32 x 128 tile): (i) stars centred on
    tile CORNERS are four tile-local components whose records comp_merge folds into one; (ii) a patch of 3-pixel components on a
    4-pixel lattice puts ~250 components into one tile -- more than its slots: the frame's overflow flag is raised and the host
    redoes it through the full path (the context COUNTS that: ab_ctx_fallback_counts).  Transforms, star counts and inliers must equal
    the oracle's; on a developer build also round 4's forms and the pixel-list chain."""
    import torch
    from astroburst_amd import synth
    rows, cols = 512, 640
    y, x, flux = synth.star_catalog(rows, cols, 420, seed=77)
    ky, kx = np.meshgrid(32.0 * np.arange(2, 14, 3) - 0.5, 128.0 * np.arange(1, 5) - 0.5, indexing="ij")     # tile corners
    y = torch.cat([y, torch.from_numpy(ky.ravel()).to(y.dtype)])
    x = torch.cat([x, torch.from_numpy(kx.ravel()).to(x.dtype)])
    flux = torch.cat([flux, torch.full((ky.size,), float(flux.max()) * 0.8, dtype=flux.dtype)])
    cat = (y, x, flux * 30.0)
    ref = synth.make_frame(rows, cols, 0, cat=cat, bad_patch_rate=0.0, cosmic_rate=0.0)
    tgts = [synth.make_frame(rows, cols, k + 1, cat=cat, shift=(2.0 * k - 5.0, 1.5 * k - 3.0), bad_patch_rate=0.0, cosmic_rate=0.0) for k in range(6)]
    for t in (tgts[1], tgts[4]):                                     # the crowded patch, in two frames of different groups
        for yy in range(68, 92, 4):
            for xx in range(260, 380, 4):
                t[yy, xx] += 9000.0
                t[yy, xx + 1] += 7000.0
                t[yy + 1, xx] += 7000.0
    ref, tgts = ref.cuda(), [t.cuda() for t in tgts]
    before = ctx.fallback_counts()
    new = ctx.register_frames(ref, tgts, num_threads=8)
    after = ctx.fallback_counts()
    _holds_to_the_oracle(oracle, ref, tgts, new)
    assert sum(a.method in ("affine", "rigid") for a in new) >= 5
    assert after["tile_slots"] - before["tile_slots"] == 2 and after["frames_redone"] - before["frames_redone"] == 2, (before, after)
    import astroburst_amd as ab
    if ab.is_dev_build():
        from conftest import _ctx_under
        for env in ({"AB_LABEL_LEGACY": "1", "AB_DETECT_FULL_RECORDS": "1"}, {"AB_DETECT_NO_RECS": "1"}):
            c = _ctx_under(env)
            try:
                for a, b in zip(new, c.register_frames(ref, tgts, num_threads=8)):
                    assert _same_registration(a, b), env
            finally:
                c.close()


def test_ctx_trim_releases_and_the_context_keeps_working(oracle):
    """ab_ctx_trim (the host-frame staging area -- 63 x 8192^2 = 17 GB -- stayed pinned to the context until it was destroyed):
    workspaces, scratch and staging of the context and its workers go back to the allocator; the next calls allocate again and
    give the same answers (registration of host frames, a stack, a phase correlation: every user of a kept workspace must notice
    that its buffer is new)."""
    import torch
    import astroburst_amd as ab
    from astroburst_amd import synth
    rows, cols = 512, 640
    y, x, flux = synth.star_catalog(rows, cols, 400, seed=13)
    cat = (y, x, flux * 30.0)
    ref = synth.make_frame(rows, cols, 0, cat=cat, bad_patch_rate=0.0, cosmic_rate=0.0)
    tgts = [synth.make_frame(rows, cols, k + 1, cat=cat, shift=s, bad_patch_rate=0.0, cosmic_rate=0.0).pin_memory()
            for k, s in enumerate([(2.5, -1.0), (-3.0, 4.0), (0.5, 0.25), (6.0, 2.0), (-1.25, -2.5)])]
    c = ab.Context(0)
    try:
        def run():
            outs = [torch.empty((rows, cols), device="cuda") for _ in tgts]
            res = c.align_pairs_affine(ref, tgts, outs, num_threads=8)
            st = c.stack_images([ref.cuda()] + outs, align=True)
            big = torch.cat([o for o in outs] * 3, dim=0)[:2100]                    # > 4 000 000 px: the statistics' histogram path
            u8, stats, stf = c.auto_stretch_preview(big.contiguous())
            c.synchronize()
            return ([(r.method, r.transform, r.inliers) for r in res], [o.cpu().numpy() for o in outs], st.image.cpu().numpy(),
                    (st.offsets, stats.median, stats.mad, stats.valid_count, stf.midtone, int(u8.sum().item())))
        # (one run and a trim BEFORE the baseline: the HIP runtime allocates a per-queue scratch arena -- ~270 MB for comp_moments' 528
        # bytes per lane -- when a stream first launches a kernel that spills; it belongs to the stream, not to the context's
        # workspaces, and stays until the stream is destroyed)
        run()
        c.trim()
        torch.cuda.synchronize()
        free0 = torch.cuda.mem_get_info()[0]
        first = run()
        used = free0 - torch.cuda.mem_get_info()[0]
        c.trim()
        after = free0 - torch.cuda.mem_get_info()[0]
        assert used > 8 * rows * cols * 4 and after < used // 4, (used, after)     # the bulk went back
        for rep in range(2):
            again = run()
            assert again[0] == first[0] and again[3] == first[3]
            assert all(np.array_equal(a, b) for a, b in zip(again[1], first[1])) and np.array_equal(again[2], first[2], equal_nan=True)
            c.trim()
    finally:
        c.close()
