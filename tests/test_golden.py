"""Frozen fixtures (tests/golden/*.npz, written by tools/gen_golden.py): the reference's own unit-test inputs plus small
adversarial inputs, with the oracle's outputs at the commit that generated them.

* CPU (`-m "not gpu"`): today's oracle must still produce the frozen numbers -- a later edit of oracle/*.c that moves a
  result is caught even if the HIP path moves with it; the reference's asserted tolerances are re-checked on the frozen
  outputs themselves.
* GPU (`-m gpu`): libastroburst_hip.so, through the C ABI, must produce the same frozen numbers.
"""
import os

import numpy as np
import pytest

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
ADV_N = (2, 3, 5, 8, 16, 33, 64)
ADV_CFG = ((3.0, 3.0, 5), (2.0, 2.5, 2), (1.5, 1.5, 8))
REF_CASES = ("clean", "outlier", "cosmic", "single")


def load(name):
    return np.load(os.path.join(GOLD, f"{name}.npz"))


def same(a, b):
    return np.array_equal(np.asarray(a), np.asarray(b), equal_nan=True)


def frames_of(g, n):
    px = g[f"adv{n}_in"]
    return [np.ascontiguousarray(px[f].reshape(12, 32)) for f in range(n)]


# ---- the frozen outputs satisfy the reference's own assertions (combine.rs:199-284, median.rs:99-145) -----------------
def test_frozen_outputs_meet_the_reference_assertions():
    g = load("combine")
    assert abs(float(g["clean_out"][0]) - 10.04) < 0.1 and int(g["clean_rej"][0]) == 0          # combine.rs:199-205
    assert float(g["outlier_out"][0]) < 15.0 and int(g["outlier_rej"][0]) > 0                    # :207-213
    assert abs(float(g["cosmic_out"][0]) - 100.0) < 1.0 and int(g["cosmic_rej"][0]) >= 1          # :215-221
    assert float(g["single_out"][0]) == 42.0 and int(g["single_rej"][0]) == 0                     # :231-237
    assert abs(g["identical_out"][0, 0]) < 1e-4 and abs(g["identical_out"][1, 1] - 50.0) < 1e-4   # :239-257
    assert abs(g["reject_out"][2, 2] - 100.0) < 1.0 and int(g["reject_rej"][0]) > 0               # :259-284
    m = load("median")
    assert abs(m["odd_out"][0] - 3.0) < 1e-6 and abs(m["even_out"][0] - 2.5) < 1e-6               # median.rs:99-113
    assert abs(m["f32_out"][0] - 3.0) < 1e-6 and abs(m["mad_out"][0] - 1.0) < 1e-6                # :121-131
    assert 40.0 < m["clipped_outliers"][0] < 60.0 and m["clipped_outliers"][1] < 500.0            # sigma_clip.rs:40-47


# ---- CPU: the oracle against its own frozen outputs ---------------------------------------------------------------------
def test_oracle_combine_frozen(oracle):
    g = load("combine")
    for name in REF_CASES:
        sl, sh, it = g[f"{name}_cfg"]
        m, r = oracle.sigma_clip_combine(g[f"{name}_in"], float(sl), float(sh), int(it))
        assert np.float32(m) == g[f"{name}_out"][0] and r == int(g[f"{name}_rej"][0]), name
    for n in ADV_N:
        fr = frames_of(g, n)
        for (sl, sh, it) in ADV_CFG:
            img, rej = oracle.stack_images(fr, sl, sh, it)
            tag = f"adv{n}_{sl}_{sh}_{it}"
            assert same(img, g[f"{tag}_out"]) and rej == int(g[f"{tag}_rej"][0]), tag
    s, c, rj = oracle.stack_partial(frames_of(g, 16), 3.0, 3.0, 5)
    assert same(s, g["partial16_sum"]) and same(c.astype(np.int64), g["partial16_cnt"]) and rj == int(g["partial16_rej"][0])


def test_oracle_median_frozen(oracle):
    m = load("median")
    assert oracle.exact_median_mut(m["odd_in"]) == m["odd_out"][0]
    assert oracle.exact_median_mut(m["even_in"]) == m["even_out"][0]
    assert np.float32(oracle.median_f32_mut(m["f32_in"])) == m["f32_out"][0]
    assert np.float32(oracle.exact_mad_mut(m["mad_in"], 3.0)) == m["mad_out"][0]
    assert oracle.exact_median_mut(m["rand_in"]) == m["rand_median_odd"][0]
    assert oracle.exact_median_mut(m["rand_in"][:1000]) == m["rand_median_even"][0]
    assert same(np.asarray(oracle.sigma_clipped_stats(m["rand_in"], 3.0, 2)), m["clipped_rand"])


def stats_row(s):
    return np.asarray([s.min, s.max, s.median, s.mad, s.sigma, s.mean, float(s.valid_count)])


def big_image():
    big = (1000.0 + 30.0 * np.random.default_rng(3).standard_normal((2100, 2000))).astype(np.float32)
    big[np.random.default_rng(4).random(big.shape) < 0.001] = np.nan
    big[:5] = 0.0
    big[100:110, 200:260] += 20000.0
    return big


def test_oracle_stats_stf_frozen(oracle):
    g = load("stats_stf")
    img = g["img"]
    st = oracle.compute_image_stats(img)
    assert same(stats_row(st), g["stats_exact"])
    assert same(stats_row(oracle.compute_image_stats(img, path="hist")), g["stats_hist"])
    p = oracle.auto_stf(st)
    assert same([p.shadow, p.midtone, p.highlight], g["auto_stf"])
    assert same(oracle.apply_stf(img, p, st), g["apply_u8"]) and same(oracle.apply_stf_f32(img, p, st), g["apply_f32"])
    for r, want in zip(g["stf_stats_rows"], g["stf_rows_out"]):
        q = oracle.auto_stf(oracle.ImageStats(r[0], r[1], r[2], r[3], r[4], r[5], int(r[6])))
        assert same([q.shadow, q.midtone, q.highlight], want)
    big = big_image()
    assert int(big.view(np.uint32).astype(np.uint64).sum()) == int(g["big_input_sum_u32"][0]), "numpy's generator stream changed"
    assert same(stats_row(oracle.compute_image_stats(big)), g["big_stats"])


def star_rows(stars):
    return np.asarray([[s.x, s.y, s.flux, s.fwhm, s.eccentricity, s.peak, s.snr, float(s.npix)] for s in stars])


def test_oracle_detect_resample_frozen(oracle):
    g = load("detect")
    stars, m, s = oracle.detect_stars(g["img"], 5.0)
    assert same([m, s], g["bg"]) and same(star_rows(stars), g["stars"])
    assert same(list(oracle.estimate_background(g["img"], 32)), g["estimate_background_32"])
    assert len(g["stars"]) >= 10                                   # the fixture really holds a star field
    r = load("resample")
    assert same(oracle.shift_image_subpixel(r["src"], 1.25, -2.5), r["shift"])
    assert same(oracle.warp_image(r["src"], tuple(r["transform"]), 37, 41), r["warp"])
    assert same([oracle.bicubic_sample(r["src"], 37, 41, y, x) for y, x in r["bicubic_pts"]], r["bicubic"])


# ---- round 6: a6 / a12 / a13 / a18 / f2 -- the rows whose only pins were restatements by the oracle's own author ---------------------
METHODS = ("affine", "rigid", "phase_correlation", "identity")


def affine_row(a):
    return list(a.transform), [a.matched_stars, a.inliers, METHODS.index(a.method)], a.residual_px


def check_affine(api, exact):
    g = load("affine")
    rows, cols = (int(v) for v in g["stars_dims"])
    for nt in (1, 8):
        t, c, r = affine_row(api.affine_from_stars(g["stars_ref_xy"], g["stars_tgt_xy"], rows, cols, num_threads=nt))
        assert c == list(g[f"stars_t{nt}_counts"]), nt
        if exact:
            assert same(t, g[f"stars_t{nt}_transform"]) and r == g[f"stars_t{nt}_residual"][0], nt
        else:   # (the library's host geometry is C++ with the oracle's operation order; libm / contraction free: held to 1e-9 like the parity tests)
            assert np.allclose(t, g[f"stars_t{nt}_transform"], rtol=0, atol=1e-9) and abs(r - g[f"stars_t{nt}_residual"][0]) < 1e-9, nt
    t, c, r = affine_row(api.align_channel_affine(g["pair_ref"], g["pair_tgt"], num_threads=8))
    assert c == list(g["pair_counts"]) and c[2] == 0 and c[1] >= 20            # an affine fit on dozens of inliers
    assert abs(g["pair_transform"][2] - (-1.75)) < 0.5 and abs(g["pair_transform"][5] - 2.5) < 0.5   # (the frozen answer IS the generating shift)
    if exact:
        assert same(t, g["pair_transform"]) and r == g["pair_residual"][0]
    else:
        assert np.allclose(t, g["pair_transform"], rtol=0, atol=1e-8) and abs(r - g["pair_residual"][0]) < 1e-8
    return g, t


def test_oracle_affine_frozen(oracle):
    g, t = check_affine(oracle, exact=True)
    w = oracle.warp_image(g["pair_tgt"], tuple(t), 256, 320)
    assert int(w.view(np.uint32).astype(np.uint64).sum()) == int(g["pair_warped_checksum"][0])


def check_background(api, mode_of):
    g = load("background")
    for tag in ("sub_g4_d2", "div_g6_d1"):
        grid, deg, kappa, it, mode = g[f"{tag}_cfg"]
        b = api.extract_background(g["img"], grid_size=int(grid), poly_degree=int(deg), sigma_clip=float(kappa), iterations=int(it), mode=mode_of(int(mode)))
        assert b.sample_count == int(g[f"{tag}_scalars"][0]), tag
        n = len(np.asarray(b.coeffs))
        assert same(np.asarray(b.coeffs, np.float64), g[f"{tag}_coeffs"][:n]) and not g[f"{tag}_coeffs"][n:].any(), tag
        assert same(b.model, g[f"{tag}_model"]) and same(b.corrected, g[f"{tag}_corrected"]), tag
        assert abs(b.rms_residual - g[f"{tag}_scalars"][1]) <= 1e-12 * g[f"{tag}_scalars"][1], tag


def test_oracle_background_frozen(oracle):
    check_background(oracle, lambda m: m)


def check_masked(api):
    g = load("masked")
    mk = api.generate_star_mask(g["img"], stars=[tuple(s) for s in g["stars_xyf"]], luminance_protect=True, luminance_ceiling=0.85)
    assert same(mk.mask, g["mask"]) and [float(mk.stars_masked), mk.coverage_fraction] == list(g["mask_scalars"])
    assert 0 < mk.stars_masked < len(g["stars_xyf"])                 # some of the list is outside the plane / the FWHM range
    for tag, cfg in (("default", dict()), ("hard", dict(iterations=25, target_background=0.4, protection_amount=0.3, convergence_threshold=1e-7))):
        ms = api.masked_stretch(g["img"], mask=mk, **cfg)
        img = ms.image.cpu().numpy() if hasattr(ms.image, "cpu") else ms.image
        assert same(img, g[f"{tag}_image"]), tag
        assert [float(ms.iterations_run), ms.final_background, float(ms.converged), float(ms.stars_masked), ms.mask_coverage] == list(g[f"{tag}_scalars"]), tag


def test_oracle_masked_frozen(oracle):
    check_masked(oracle)


def check_spcc(api, oracle_mod, rtol):
    g = load("spcc")
    stars = [oracle_mod.DetectedStar(x=r[0], y=r[1], flux=r[2], fwhm=r[3], eccentricity=r[4], peak=r[5], snr=r[6], npix=int(r[7])) for r in g["stars"]]
    for white in ("average_spiral", "g2v"):
        res = api.spcc_calibrate_rgb(g["r"], g["g"], g["b"], 1.2, detection=(stars, float(g["lum_max"][0])), min_snr=15.0, max_stars=150,
                                     saturation_limit=0.95, white_reference=white)
        assert [res.stars_matched, res.stars_total] == list(g[f"{white}_counts"]) and res.stars_matched >= 20, white
        got = [res.r_factor, res.g_factor, res.b_factor, res.avg_color_index]
        assert np.allclose(got, g[f"{white}_factors"], rtol=rtol, atol=0), white
    assert g["average_spiral_factors"][0] < 1.0 < g["average_spiral_factors"][2]   # the field was rendered with gains (1.4, 1.0, 0.7)


def test_oracle_spcc_frozen(oracle):
    check_spcc(oracle, oracle, 0.0)


def batch_frames(g, n):
    return [np.ascontiguousarray(g[f"adv{n}_in"][f].reshape(12, 32)) for f in range(n)]


def test_oracle_batch_frozen(oracle):
    g = load("batch")
    for n in (5, 16, 33):
        for (sl, sh, it) in ((2.5, 3.0, 5), (1.0, 1.0, 2)):
            img, rej = oracle.sigma_clipped_mean_stack(batch_frames(g, n), sl, sh, it)
            assert same(img, g[f"adv{n}_{sl}_{sh}_{it}_out"]) and same(np.asarray(rej, np.int64), g[f"adv{n}_{sl}_{sh}_{it}_rej"]), (n, sl, sh, it)
    lights = [np.ascontiguousarray(f) for f in g["lights"]]
    for normalize in (True, False):
        img, rej, mean, std = oracle.run_batch_channel(lights, g["bias"], None, g["flat"], normalize=normalize)
        k = f"channel_norm{int(normalize)}"
        assert same(img, g[f"{k}_out"]) and same(np.asarray(rej, np.int64), g[f"{k}_rej"]) and same([mean, std], g[f"{k}_stats"]), k


def test_generator_reproduces_the_committed_fixtures():
    """tools/gen_golden.py --check: the committed files are what the committed generator + oracle produce"""
    import subprocess
    import sys
    root = os.path.dirname(GOLD.rstrip("/")).rsplit("/tests", 1)[0]
    res = subprocess.run([sys.executable, os.path.join(root, "tools", "gen_golden.py"), "--check"], capture_output=True, text=True)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-2000:]


# ---- GPU: libastroburst_hip.so against the same frozen numbers ------------------------------------------------------------
@pytest.mark.gpu
def test_hip_combine_frozen(ctx):
    g = load("combine")
    for name in REF_CASES:
        sl, sh, it = g[f"{name}_cfg"]
        m, r = ctx.sigma_clip_combine(g[f"{name}_in"], float(sl), float(sh), int(it))
        assert np.float32(m) == g[f"{name}_out"][0] and r == int(g[f"{name}_rej"][0]), name
    for n in ADV_N:
        fr = frames_of(g, n)
        for (sl, sh, it) in ADV_CFG:
            img, rej = ctx.stack_sigma_clip(fr, sl, sh, it)
            tag = f"adv{n}_{sl}_{sh}_{it}"
            assert same(img, g[f"{tag}_out"]) and rej == int(g[f"{tag}_rej"][0]), tag
    import torch
    s, c, rj = ctx.stack_partial([torch.from_numpy(f).cuda() for f in frames_of(g, 16)], 3.0, 3.0, 5)
    assert same(s.cpu().numpy(), g["partial16_sum"]) and same(c.cpu().numpy().astype(np.int64), g["partial16_cnt"])
    assert rj == int(g["partial16_rej"][0])


@pytest.mark.gpu
def test_hip_stats_stf_frozen(ctx):
    import torch
    g = load("stats_stf")
    img = g["img"]
    st = ctx.compute_image_stats(img)
    want = g["stats_exact"]
    assert same(stats_row(st)[:5], want[:5]) and st.valid_count == int(want[6]) and abs(st.mean - want[5]) <= 1e-12 * abs(want[5])
    p = ctx.auto_stf(st)
    assert same([p.shadow, p.midtone, p.highlight], g["auto_stf"])
    assert same(ctx.apply_stf(img, p, st), g["apply_u8"]) and same(ctx.apply_stf_f32(img, p, st), g["apply_f32"])
    for r, w in zip(g["stf_stats_rows"], g["stf_rows_out"]):
        from astroburst_amd import ImageStats
        q = ctx.auto_stf(ImageStats(r[0], r[1], r[2], r[3], r[4], r[5], int(r[6])))
        assert same([q.shadow, q.midtone, q.highlight], w)
    big = big_image()
    u8, sb, pb = ctx.auto_stretch_preview(torch.from_numpy(big).cuda())     # the histogram path, one device chain
    want = g["big_stats"]
    assert same(stats_row(sb)[:5], want[:5]) and sb.valid_count == int(want[6]) and abs(sb.mean - want[5]) <= 1e-12 * abs(want[5])
    assert same([pb.shadow, pb.midtone, pb.highlight], g["big_auto_stf"])
    assert same(np.bincount(u8.cpu().numpy().ravel(), minlength=256), g["big_u8_hist"])


@pytest.mark.gpu
def test_hip_detect_resample_frozen(ctx):
    g = load("detect")
    stars, m, s = ctx.detect_stars(g["img"], 5.0)[:3]
    assert same([m, s], g["bg"])
    got, want = star_rows(stars), g["stars"]
    assert got.shape == want.shape
    assert same(got[:, 7], want[:, 7])                              # npix: the segmentation, exact and in the same (flux) order
    assert np.allclose(got[:, :7], want[:, :7], rtol=1e-10, atol=0)  # f64 moments: raster vs BFS summation order
    assert same(list(ctx.estimate_background(g["img"], 32)), g["estimate_background_32"])
    r = load("resample")
    assert same(ctx.shift_image_subpixel(r["src"], 1.25, -2.5), r["shift"])
    assert same(ctx.warp_image(r["src"], tuple(r["transform"]), 37, 41), r["warp"])


@pytest.mark.gpu
def test_hip_affine_frozen(ctx):
    g, t = check_affine(ctx, exact=False)
    w = ctx.warp_image(g["pair_tgt"], tuple(g["pair_transform"]), 256, 320)      # (the FROZEN transform: the warp is bit-exact given its input)
    assert int(np.asarray(w).view(np.uint32).astype(np.uint64).sum()) == int(g["pair_warped_checksum"][0])


@pytest.mark.gpu
def test_hip_background_frozen(ctx):
    check_background(ctx, lambda m: ("subtract", "divide")[m])


@pytest.mark.gpu
def test_hip_masked_frozen(ctx):
    check_masked(ctx)


@pytest.mark.gpu
def test_hip_spcc_frozen(ctx, oracle):
    check_spcc(ctx, oracle, 0.0)          # (aperture sums in the oracle's raster order, the colour maths the same host f64 ops: bit for bit)


@pytest.mark.gpu
def test_hip_batch_frozen(ctx):
    from astroburst_amd.core import BatchStackConfig
    g = load("batch")
    for n in (5, 16, 33):
        for (sl, sh, it) in ((2.5, 3.0, 5), (1.0, 1.0, 2)):
            img, rej = ctx.sigma_clipped_mean_stack(batch_frames(g, n), BatchStackConfig(sl, sh, it))
            assert same(img, g[f"adv{n}_{sl}_{sh}_{it}_out"]) and same(np.asarray(rej, np.int64), g[f"adv{n}_{sl}_{sh}_{it}_rej"]), (n, sl, sh, it)
    lights = [np.ascontiguousarray(f) for f in g["lights"]]
    for normalize in (True, False):
        img, rej, mean, std = ctx.run_batch_channel(lights, g["bias"], None, g["flat"], BatchStackConfig(normalize_before_stack=normalize))
        k = f"channel_norm{int(normalize)}"
        assert same(img, g[f"{k}_out"]) and same(np.asarray(rej, np.int64), g[f"{k}_rej"]), k
        assert abs(mean - g[f"{k}_stats"][0]) <= 1e-12 * abs(g[f"{k}_stats"][0]) and abs(std - g[f"{k}_stats"][1]) <= 1e-10 * abs(g[f"{k}_stats"][1]), k
