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
