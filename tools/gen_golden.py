#!/usr/bin/env python3
"""Freeze the oracle: writes tests/golden/*.npz = inputs + the oracle's outputs at the commit that generated them.

    python tools/gen_golden.py            # (re)generate every fixture
    python tools/gen_golden.py --check    # regenerate in memory and compare with the committed files (CI drift check)

The reference (Rust) cannot be built here and ships no golden vectors -- only behavioural #[test]s -- so the fixtures
hold (a) the INPUTS of the reference's own unit tests on the hot path, verbatim (file:line in each key's comment below),
with the oracle's outputs for them, and (b) small adversarial inputs (NaN / inf / ties / subnormals / ragged counts).
tests/test_golden.py compares BOTH the oracle (CPU, -m "not gpu") and libastroburst_hip.so (-m gpu) with these frozen
numbers, so a later edit of oracle/*.c that changes a result is caught even if the HIP path changes with it.

Fixtures are data only (inputs / expected outputs); nothing of the reference's source text is stored.
"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
GOLD = os.path.join(ROOT, "tests", "golden")


def adversarial_pixels(rng, n_frames, n_px):
    """n_px pixels x n_frames samples: sky-like values with outliers, ties, non-finite samples, tiny and huge values"""
    v = (1000.0 + 30.0 * rng.standard_normal((n_frames, n_px))).astype(np.float32)
    k = n_px // 8
    v[:, :k] = np.round(v[:, :k] / 8.0) * 8.0                       # heavy ties
    hot = rng.random((n_frames, n_px)) < 0.02
    v[hot] *= rng.uniform(5, 50, hot.sum()).astype(np.float32)      # cosmic rays
    v[rng.random((n_frames, n_px)) < 0.01] = np.nan
    v[rng.random((n_frames, n_px)) < 0.003] = np.inf
    v[rng.random((n_frames, n_px)) < 0.003] = -np.inf
    v[:, k:k + 8] = np.float32(7.0)                                  # all samples equal (MAD = 0 -> sigma floor 1e-10)
    v[:, k + 8:k + 16] = np.nan                                      # no finite sample at all
    v[: n_frames // 2, k + 16:k + 24] = np.nan                       # half the frames missing
    v[:, k + 24:k + 32] *= np.float32(1e-38)                         # subnormal range
    v[:, k + 32:k + 40] *= np.float32(1e30)
    if n_frames > 1:
        v[1:, k + 40:k + 48] = np.nan                                # exactly one finite sample
    return v


def build():
    from oracle import pyoracle as o
    out = {}

    # ---- a1 / a2  core/stacking/combine.rs ---------------------------------------------------------------
    ref_cases = {   # the reference's own test inputs (combine.rs:199-237): (values, sigma_low, sigma_high, max_iter)
        "clean": ([10.0, 10.1, 9.9, 10.0, 10.2], 3.0, 3.0, 5),              # combine.rs:199-205
        "outlier": ([10.0, 10.1, 9.9, 10.0, 500.0], 3.0, 3.0, 5),           # combine.rs:207-213
        "cosmic": ([100.0, 100.2, 99.8, 100.1, 100.0, 5000.0, 99.9], 2.0, 2.0, 5),   # combine.rs:215-221
        "single": ([42.0], 3.0, 3.0, 5),                                      # combine.rs:231-237
    }
    d = {}
    for name, (vals, sl, sh, it) in ref_cases.items():
        m, r = o.sigma_clip_combine(vals, sl, sh, it)
        d[f"{name}_in"] = np.asarray(vals, np.float32)
        d[f"{name}_cfg"] = np.asarray([sl, sh, it], np.float64)
        d[f"{name}_out"] = np.asarray([m], np.float32)
        d[f"{name}_rej"] = np.asarray([r], np.int64)
    rng = np.random.default_rng(20260929)
    for n in (2, 3, 5, 8, 16, 33, 64):
        px = adversarial_pixels(rng, n, 384)
        frames = [px[f].reshape(12, 32) for f in range(n)]
        for (sl, sh, it) in ((3.0, 3.0, 5), (2.0, 2.5, 2), (1.5, 1.5, 8)):
            img, rej = o.stack_images(frames, sl, sh, it)
            tag = f"adv{n}_{sl}_{sh}_{it}"
            d[f"{tag}_out"] = img
            d[f"{tag}_rej"] = np.asarray([rej], np.int64)
        d[f"adv{n}_in"] = px
    img = (np.arange(16, dtype=np.float32) * 10.0).reshape(4, 4)               # combine.rs:239-257
    d["identical_in"] = img
    d["identical_out"] = o.stack_images([img, img, img])[0]
    clean = np.full((4, 4), 100.0, np.float32)                                 # combine.rs:259-284
    noisy = clean.copy()
    noisy[2, 2] = 50000.0
    res, rej = o.stack_images([clean, clean, clean, noisy, clean], 3.0, 3.0, 5)
    d["reject_out"], d["reject_rej"] = res, np.asarray([rej], np.int64)
    # the two-level (frame-sharded) estimator's partials on the same adversarial pixels
    px = d["adv16_in"]
    s, c, rj = o.stack_partial([px[f].reshape(12, 32) for f in range(16)], 3.0, 3.0, 5)
    d["partial16_sum"], d["partial16_cnt"], d["partial16_rej"] = s, c.astype(np.int64), np.asarray([rj], np.int64)
    out["combine"] = d

    # ---- math/median.rs:99-145 -----------------------------------------------------------------------------
    d = {}
    med_cases = {"odd": [5.0, 1.0, 3.0, 2.0, 4.0], "even": [1.0, 2.0, 3.0, 4.0], "f32": [5.0, 1.0, 3.0, 2.0, 4.0],
                 "mad": [1.0, 2.0, 3.0, 4.0, 5.0]}
    for k, v in med_cases.items():
        d[f"{k}_in"] = np.asarray(v, np.float32)
    d["odd_out"] = np.asarray([o.exact_median_mut(med_cases["odd"])])
    d["even_out"] = np.asarray([o.exact_median_mut(med_cases["even"])])
    d["f32_out"] = np.asarray([o.median_f32_mut(med_cases["f32"])], np.float32)
    d["mad_out"] = np.asarray([o.exact_mad_mut(med_cases["mad"], 3.0)], np.float32)
    vals = (50.0 + 10.0 * rng.standard_normal(1001)).astype(np.float32)
    d["rand_in"] = vals
    d["rand_median_odd"] = np.asarray([o.exact_median_mut(vals)])
    d["rand_median_even"] = np.asarray([o.exact_median_mut(vals[:1000])])
    m, s_ = o.sigma_clipped_stats(np.concatenate([np.arange(1, 101, dtype=np.float32), [np.float32(100000.0)]]), 3.0, 3)   # sigma_clip.rs:40-47
    d["clipped_outliers"] = np.asarray([m, s_])
    m, s_ = o.sigma_clipped_stats(vals, 3.0, 2)
    d["clipped_rand"] = np.asarray([m, s_])
    out["median"] = d

    # ---- a9-a11  core/imaging/stats.rs, stf.rs:161-262 ----------------------------------------------------------------
    d = {}
    img = (1000.0 + 30.0 * rng.standard_normal((40, 50))).astype(np.float32)
    img[:3] = 0.0
    img[5, 5:9] = np.nan
    img[6, 6] = np.inf
    img[20:22, 20:24] += 20000.0
    d["img"] = img
    st = o.compute_image_stats(img)                      # exact path (<= 4 000 000 px)
    sth = o.compute_image_stats(img, path="hist")        # the histogram path forced onto the same pixels
    as_row = lambda s: np.asarray([s.min, s.max, s.median, s.mad, s.sigma, s.mean, float(s.valid_count)])
    d["stats_exact"], d["stats_hist"] = as_row(st), as_row(sth)
    p = o.auto_stf(st)
    d["auto_stf"] = np.asarray([p.shadow, p.midtone, p.highlight])
    d["apply_u8"] = o.apply_stf(img, p, st)
    d["apply_f32"] = o.apply_stf_f32(img, p, st)
    # stf.rs:161-262: auto_stf on hand-built statistics (min, max, median, mad, sigma, mean, n)
    rows = [(0.0, 1.0, 0.1, 0.01, 0.015, 0.1, 1000), (0.0, 65535.0, 1200.0, 20.0, 29.652, 1300.0, 1 << 20),
            (5.0, 5.0, 5.0, 0.0, 1e-30, 5.0, 10), (0.0, 1.0, 0.5, 0.1, 0.14826, 0.5, 100), (0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0)]
    d["stf_stats_rows"] = np.asarray(rows, np.float64)
    got = []
    for r in rows:
        q = o.auto_stf(o.ImageStats(r[0], r[1], r[2], r[3], r[4], r[5], int(r[6])))
        got.append([q.shadow, q.midtone, q.highlight])
    d["stf_rows_out"] = np.asarray(got)
    d["mtf"] = np.asarray([[o.mtf(x, m) for x in (0.0, 0.1, 0.25, 0.5, 0.9, 1.0)] for m in (0.1, 0.25, 0.5, 0.75)])
    # checksum-pinned large case: inputs come from the seeded generator, only the outputs are frozen
    big = (1000.0 + 30.0 * np.random.default_rng(3).standard_normal((2100, 2000))).astype(np.float32)
    big[np.random.default_rng(4).random(big.shape) < 0.001] = np.nan
    big[:5] = 0.0
    big[100:110, 200:260] += 20000.0
    d["big_input_sum_u32"] = np.asarray([int(big.view(np.uint32).astype(np.uint64).sum())], np.uint64)   # generator drift guard
    sb = o.compute_image_stats(big)
    d["big_stats"] = as_row(sb)
    pb = o.auto_stf(sb)
    d["big_auto_stf"] = np.asarray([pb.shadow, pb.midtone, pb.highlight])
    u8 = o.apply_stf(big, pb, sb)
    d["big_u8_hist"] = np.bincount(u8.ravel(), minlength=256).astype(np.int64)
    out["stats_stf"] = d

    # ---- a7  core/analysis/star_detection.rs:289-328 ---------------------------------------------------------
    d = {}
    import torch
    from astroburst_amd import synth
    y, x, flux = synth.star_catalog(200, 240, 40, seed=11)
    frame = synth.make_frame(200, 240, 0, cat=(y, x, flux * 40.0), bad_patch_rate=0.0, cosmic_rate=0.0).numpy()
    d["img"] = frame
    stars, bg_m, bg_s = o.detect_stars(frame, 5.0)
    d["bg"] = np.asarray([bg_m, bg_s])
    d["stars"] = np.asarray([[s.x, s.y, s.flux, s.fwhm, s.eccentricity, s.peak, s.snr, float(s.npix)] for s in stars])
    em, es = o.estimate_background(frame, 32)
    d["estimate_background_32"] = np.asarray([em, es])
    flat = np.full((64, 64), 100.0, np.float32)                                 # star_detection.rs: no stars on a flat field
    s0, m0, g0 = o.detect_stars(flat, 5.0)
    d["flat_count"] = np.asarray([len(s0)], np.int64)
    d["flat_bg"] = np.asarray([m0, g0])
    out["detect"] = d

    # ---- a3-a5  sampling.rs:86-142, affine.rs warp -----------------------------------------------------------------
    d = {}
    src = rng.random((37, 41)).astype(np.float32)
    d["src"] = src
    d["shift"] = o.shift_image_subpixel(src, 1.25, -2.5)
    t = (0.9998, -0.012, 1.25, 0.011, 1.0003, -0.5)
    d["transform"] = np.asarray(t)
    d["warp"] = o.warp_image(src, t, 37, 41)
    pts = [(3.5, 4.25), (0.0, 0.0), (36.0, 40.0), (-0.4, 12.3), (17.49, 40.3)]
    d["bicubic_pts"] = np.asarray(pts)
    d["bicubic"] = np.asarray([o.bicubic_sample(src, 37, 41, yy, xx) for yy, xx in pts])
    d["bilinear"] = np.asarray([o.bilinear_sample(src, 37, 41, yy, xx) for yy, xx in pts])
    out["resample"] = d

    # ==== round 6: rows whose only pins were same-author restatements get FROZEN arrays too, so that an edit
    # which moves oracle/*.c and the kernels together no longer goes unseen ===================================================

    # ---- a6  core/alignment/affine.rs:129-212 (+ :400-642 the fits / RANSAC) ---------------------------------------------------
    d = {}
    import math
    r6 = np.random.default_rng(606)
    ref_xy = np.column_stack([r6.uniform(20, 1180, 90), r6.uniform(20, 980, 90)])
    ang = math.radians(-0.8)
    ca, sa = math.cos(ang), math.sin(ang)
    tgt_xy = np.column_stack([ca * ref_xy[:, 0] - sa * ref_xy[:, 1] - 21.0, sa * ref_xy[:, 0] + ca * ref_xy[:, 1] + 6.5]) + r6.normal(0, 0.1, ref_xy.shape)
    tgt_xy = tgt_xy[r6.permutation(90)][:70]                         # 20 stars missing in the target, order shuffled
    d["stars_ref_xy"], d["stars_tgt_xy"] = ref_xy, tgt_xy
    d["stars_dims"] = np.asarray([1000, 1200], np.int64)
    for nt in (1, 8):
        a = o.affine_from_stars(ref_xy, tgt_xy, 1000, 1200, num_threads=nt)
        d[f"stars_t{nt}_transform"] = np.asarray(a.transform)
        d[f"stars_t{nt}_counts"] = np.asarray([a.matched_stars, a.inliers, o.AFFINE_METHODS.index(a.method)], np.int64)
        d[f"stars_t{nt}_residual"] = np.asarray([a.residual_px])
    y, x, flux = synth.star_catalog(256, 320, 160, seed=66)
    cat = (y, x, flux * 30.0)
    pair_ref = synth.make_frame(256, 320, 0, cat=cat, bad_patch_rate=0.0, cosmic_rate=0.0).numpy()
    pair_tgt = synth.make_frame(256, 320, 1, cat=cat, shift=(2.5, -1.75), bad_patch_rate=0.0, cosmic_rate=0.0).numpy()
    d["pair_ref"], d["pair_tgt"] = pair_ref, pair_tgt
    a = o.align_channel_affine(pair_ref, pair_tgt, num_threads=8)
    d["pair_transform"] = np.asarray(a.transform)
    d["pair_counts"] = np.asarray([a.matched_stars, a.inliers, o.AFFINE_METHODS.index(a.method)], np.int64)
    d["pair_residual"] = np.asarray([a.residual_px])
    d["pair_warped_checksum"] = np.asarray([int(o.warp_image(pair_tgt, a.transform, 256, 320).view(np.uint32).astype(np.uint64).sum())], np.uint64)
    out["affine"] = d

    # ---- a12  core/imaging/background.rs:118-290 ---------------------------------------------------------------------------------
    d = {}
    yy, xx = np.mgrid[0:96, 0:128]
    ny, nx = yy / 96 - 0.5, xx / 128 - 0.5
    sky = 300.0 + 80.0 * ny - 40.0 * nx + 60.0 * ny * nx + 35.0 * nx * nx + r6.normal(0, 3.0, (96, 128))
    for _ in range(12):
        cy, cx, amp, sg = r6.uniform(0, 96), r6.uniform(0, 128), r6.uniform(200, 20000), r6.uniform(1.0, 3.0)
        sky += amp * np.exp(-((yy - cy) ** 2 + (xx - cx) ** 2) / (2 * sg * sg))
    sky = sky.astype(np.float32)
    sky[5, 7:30] = np.nan
    sky[48, 40] = np.inf
    sky[30, 64] = -np.inf
    d["img"] = sky
    for tag, kw in (("sub_g4_d2", dict(grid_size=4, poly_degree=2, sigma_clip=2.5, iterations=3, mode=0)),
                    ("div_g6_d1", dict(grid_size=6, poly_degree=1, sigma_clip=3.0, iterations=2, mode=1))):
        b = o.extract_background(sky, **kw)
        d[f"{tag}_cfg"] = np.asarray([kw["grid_size"], kw["poly_degree"], kw["sigma_clip"], kw["iterations"], kw["mode"]], np.float64)
        d[f"{tag}_coeffs"] = np.asarray(b.coeffs, np.float64)
        d[f"{tag}_model"], d[f"{tag}_corrected"] = b.model, b.corrected
        d[f"{tag}_scalars"] = np.asarray([float(b.sample_count), b.rms_residual])
    out["background"] = d

    # ---- a13  core/imaging/star_mask.rs:38-138, masked_stretch.rs:60-118 -----------------------------------------------------
    d = {}
    field = r6.normal(0.02, 0.002, (120, 160))
    sig = 3.5 / 2.3548
    yy, xx = np.mgrid[0:120, 0:160]
    for _ in range(25):
        cy, cx, amp = r6.uniform(8, 112), r6.uniform(8, 152), r6.uniform(0.05, 0.9)
        field += amp * np.exp(-((yy - cy) ** 2 + (xx - cx) ** 2) / (2 * sig * sig))
    field = field.clip(1e-5, None).astype(np.float32)
    field[0, 0], field[1, 1], field[2, 2] = np.nan, -1.0, np.inf
    mstars = np.asarray([(r6.uniform(0, 160), r6.uniform(0, 120), r6.uniform(1.5, 8.0)) for _ in range(20)]
                        + [(-30.0, -30.0, 5.0), (400.0, 100.0, 5.0), (60.0, 60.0, 31.0), (100.0, 50.0, 1.0)])   # outside / FWHM out of range
    d["img"], d["stars_xyf"] = field, mstars
    mk = o.generate_star_mask(field, stars=[tuple(s) for s in mstars], luminance_protect=True, luminance_ceiling=0.85)
    d["mask"] = mk.mask
    d["mask_scalars"] = np.asarray([float(mk.stars_masked), mk.coverage_fraction])
    for tag, cfg in (("default", dict()), ("hard", dict(iterations=25, target_background=0.4, protection_amount=0.3, convergence_threshold=1e-7))):
        ms = o.masked_stretch(field, mask=mk, **cfg)
        d[f"{tag}_image"] = ms.image
        d[f"{tag}_scalars"] = np.asarray([float(ms.iterations_run), ms.final_background, float(ms.converged), float(ms.stars_masked), ms.mask_coverage])
    out["masked"] = d

    # ---- a18  core/astrometry/spcc.rs:328-435 (after the detection: apertures, cross-match on the synthetic catalogue, factors) ---
    d = {}
    planes = [np.full((160, 200), 0.02, np.float64) for _ in range(3)]
    sig = 3.2 / 2.3548
    for _ in range(45):
        cy, cx, amp = r6.uniform(15, 145), r6.uniform(15, 185), r6.uniform(0.05, 0.6)
        col = r6.uniform(0.7, 1.3, 3)
        y0, x0 = int(cy) - 12, int(cx) - 12
        yy, xx = np.mgrid[y0:y0 + 25, x0:x0 + 25]
        psf = amp * np.exp(-((yy - cy) ** 2 + (xx - cx) ** 2) / (2 * sig * sig))
        for c, gain in enumerate((1.4, 1.0, 0.7)):
            planes[c][y0:y0 + 25, x0:x0 + 25] += psf * col[c] * gain
    rgb = [(p + r6.normal(0, 0.0008, p.shape)).astype(np.float32) for p in planes]
    lum = (np.float32(0.2126) * rgb[0] + np.float32(0.7152) * rgb[1]) + np.float32(0.0722) * rgb[2]
    sp_stars, _, _ = o.detect_stars(lum, 5.0)
    lum_max = o.compute_image_stats(lum).max
    d["r"], d["g"], d["b"] = rgb
    d["stars"] = np.asarray([[s.x, s.y, s.flux, s.fwhm, s.eccentricity, s.peak, s.snr, float(s.npix)] for s in sp_stars])
    d["lum_max"] = np.asarray([lum_max])
    for white in ("average_spiral", "g2v"):
        res = o.spcc_calibrate_rgb(*rgb, 1.2, detection=(sp_stars, lum_max), min_snr=15.0, max_stars=150, saturation_limit=0.95, white_reference=white)
        d[f"{white}_factors"] = np.asarray([res.r_factor, res.g_factor, res.b_factor, res.avg_color_index])
        d[f"{white}_counts"] = np.asarray([res.stars_matched, res.stars_total], np.int64)
    out["spcc"] = d

    # ---- f2  core/imaging/calibration_pipeline.rs:317-378 (median / MAD every iteration, strict <, per-frame rejection counts) -----
    d = {}
    for n in (5, 16, 33):
        px = adversarial_pixels(r6, n, 384)
        d[f"adv{n}_in"] = px
        frames = [np.ascontiguousarray(px[f].reshape(12, 32)) for f in range(n)]
        for (sl, sh, it) in ((2.5, 3.0, 5), (1.0, 1.0, 2)):
            img, rej = o.sigma_clipped_mean_stack(frames, sl, sh, it)
            d[f"adv{n}_{sl}_{sh}_{it}_out"] = img
            d[f"adv{n}_{sl}_{sh}_{it}_rej"] = np.asarray(rej, np.int64)
    lights = [r6.normal(400 + 3 * k, 12, (24, 40)).astype(np.float32) for k in range(9)]
    lights[2][r6.random((24, 40)) < 0.05] += 500.0
    lights[1][5, 5] = np.nan
    bias = r6.normal(100, 2, (24, 40)).astype(np.float32)
    flat = r6.normal(1.0, 0.05, (24, 40)).astype(np.float32)
    flat[3, 3] = 0.0
    d["lights"], d["bias"], d["flat"] = np.stack(lights), bias, flat
    for normalize in (True, False):
        img, rej, mean, std = o.run_batch_channel(lights, bias, None, flat, normalize=normalize)
        d[f"channel_norm{int(normalize)}_out"] = img
        d[f"channel_norm{int(normalize)}_rej"] = np.asarray(rej, np.int64)
        d[f"channel_norm{int(normalize)}_stats"] = np.asarray([mean, std])
    out["batch"] = d
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--check", action="store_true")
    args = ap.parse_args()
    data = build()
    os.makedirs(GOLD, exist_ok=True)
    bad = 0
    for name, d in data.items():
        path = os.path.join(GOLD, f"{name}.npz")
        if args.check:
            old = np.load(path)
            for k, v in d.items():
                if k not in old or not np.array_equal(np.asarray(old[k]), np.asarray(v), equal_nan=True):
                    print(f"DRIFT {name}:{k}")
                    bad += 1
        else:
            np.savez_compressed(path, **d)
            print(f"wrote {path}: {len(d)} arrays, {os.path.getsize(path)} bytes")
    if args.check:
        print("golden fixtures match the oracle" if not bad else f"{bad} arrays differ")
        sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
