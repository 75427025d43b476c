"""The star matcher / RANSAC / transform fit held to an independent restatement (tests/affine_restatement.py,
plain Python from affine.rs) -- the oracle's C and the library's host code are two restatements by one author.
Everything here is f64 arithmetic in the reference's order, so the comparison is EXACT: same method, same inlier count, same six
coefficients bit for bit."""
import ctypes
import math

import numpy as np
import pytest

import affine_restatement as ar


def star_field(rng, n, rows, cols):
    return np.column_stack([rng.uniform(20, cols - 20, n), rng.uniform(20, rows - 20, n)])


def moved(ref, ang_deg, tx, ty, shear=0.0, scale=1.0):
    a = math.radians(ang_deg)
    c, s = math.cos(a) * scale, math.sin(a) * scale
    return np.column_stack([c * ref[:, 0] - (s - shear) * ref[:, 1] + tx, s * ref[:, 0] + c * ref[:, 1] + ty])


def cases():
    rng = np.random.default_rng(11)
    out = []
    ref = star_field(rng, 80, 1000, 1200)
    out.append(("rigid 1.3 deg, 80 stars, permuted", ref, (moved(ref, 1.3, 14.5, -9.25) + rng.normal(0, 0.05, ref.shape))[rng.permutation(80)], 1000, 1200))
    ref = star_field(rng, 40, 800, 900)
    out.append(("shear + scale, 40 stars", ref, moved(ref, -2.0, -30.0, 22.0, shear=0.01, scale=1.02) + rng.normal(0, 0.1, ref.shape), 800, 900))
    ref = star_field(rng, 30, 600, 600)
    tgt = moved(ref, 0.4, 3.0, 4.0) + rng.normal(0, 0.05, ref.shape)
    tgt[:8] = star_field(rng, 8, 600, 600)  # eight stars that match nothing
    out.append(("30 stars, 8 of them strangers", ref, tgt, 600, 600))
    ref = star_field(rng, 12, 400, 500)
    out.append(("12 stars, translation", ref, ref + np.array([5.25, -7.5]), 400, 500))
    ref = star_field(rng, 25, 500, 500)
    out.append(("unrelated fields", ref, star_field(rng, 25, 500, 500), 500, 500))
    out.append(("three stars", ref[:3], ref[:3], 500, 500))
    ref = star_field(rng, 60, 900, 900)
    out.append(("rotation beyond the sanity limit", ref, moved(ref - 450.0, 40.0, 450.0, 450.0), 900, 900))
    return out


CASES = cases()


def same(res, want):
    if want is None:
        assert res is None
        return
    assert res is not None
    assert res.method == want["method"] and res.inliers == want["inliers"] and res.matched_stars == want["matched_stars"]
    assert tuple(res.transform) == tuple(want["transform"]), (res.transform, want["transform"])
    assert res.residual_px == want["residual_px"]


@pytest.mark.parametrize("name,ref,tgt,rows,cols", CASES, ids=[c[0] for c in CASES])
@pytest.mark.parametrize("threads", [1, 3, 8])
def test_oracle_equals_the_restatement(oracle, name, ref, tgt, rows, cols, threads):
    if threads != 8 and len(ref) > 40:
        pytest.skip("the large fields run once")
    same(oracle.affine_from_stars(ref, tgt, rows, cols, num_threads=threads), ar.affine_from_stars(ref, tgt, rows, cols, num_threads=threads))


@pytest.mark.parametrize("name,ref,tgt,rows,cols", CASES, ids=[c[0] for c in CASES])
def test_library_host_matcher_equals_the_restatement(name, ref, tgt, rows, cols):
    """ab_affine_from_stars is host code of the product library (no device needed): the same comparison through the C ABI"""
    from astroburst_amd import _lib
    from astroburst_amd.core import AFFINE_METHODS
    L = _lib.lib()
    r = np.ascontiguousarray(ref, np.float64)
    t = np.ascontiguousarray(tgt, np.float64)
    res, found = _lib.AffineAlignResultC(), ctypes.c_int(0)
    rc = L.ab_affine_from_stars(r.ctypes.data_as(ctypes.POINTER(ctypes.c_double)), r.shape[0], t.ctypes.data_as(ctypes.POINTER(ctypes.c_double)),
                                t.shape[0], rows, cols, 8, ctypes.byref(res), ctypes.byref(found))
    assert rc == _lib.AB_OK
    want = ar.affine_from_stars(ref, tgt, rows, cols, num_threads=8)
    if want is None:
        assert not found.value
        return
    assert found.value
    assert AFFINE_METHODS[res.method] == want["method"] and res.inliers == want["inliers"] and res.matched_stars == want["matched_stars"]
    assert tuple(res.transform) == tuple(want["transform"])
    assert res.residual_px == want["residual_px"]


def test_fits_equal_the_restatement(oracle):
    rng = np.random.default_rng(5)
    for n in (2, 3, 4, 17, 60):
        ref = star_field(rng, n, 700, 700)
        tgt = moved(ref, 0.9, -4.0, 6.0, shear=0.003) + rng.normal(0, 0.2, ref.shape)
        m = [tuple(map(float, row)) for row in np.column_stack([ref, tgt])]
        assert oracle.fit_rigid(m) == ar.fit_rigid(m)
        got, want = oracle.fit_affine(m), ar.fit_affine(m)
        assert got == want if want is not None else got is None
    # a degenerate (collinear) sample has no affine solution: |det| < 1e-12
    line = [(float(i), 2.0 * i, float(i) + 1.0, 2.0 * i - 3.0) for i in range(5)]
    assert ar.fit_affine(line) is None and oracle.fit_affine(line) is None


@pytest.mark.gpu
def test_gpu_vote_path_equals_the_restatement(ctx, oracle):
    """align_channel_affine on the device (GPU triangle build / bucket / vote, host RANSAC) against the restatement fed with the
    device's own star lists: the matcher's product path, not only its host twin"""
    import torch
    from astroburst_amd import synth
    y, x, flux = synth.star_catalog(512, 640, 260, seed=9)
    cat = (y, x, flux * 30.0)
    ref_f = synth.make_frame(512, 640, 0, cat=cat, bad_patch_rate=0.0, cosmic_rate=0.0)
    tgt_f = synth.make_frame(512, 640, 1, cat=cat, shift=(3.5, -2.25), bad_patch_rate=0.0, cosmic_rate=0.0)
    res = ctx.align_channel_affine(ref_f.cuda(), tgt_f.cuda(), num_threads=8)
    rn, tn = oracle.normalize_for_detection(ref_f.numpy()), oracle.normalize_for_detection(tgt_f.numpy())
    rs = [(s.x, s.y) for s in ctx.detect_stars(torch.from_numpy(rn).cuda(), 3.5)[0]]
    ts = [(s.x, s.y) for s in ctx.detect_stars(torch.from_numpy(tn).cuda(), 3.5)[0]]
    want = ar.affine_from_stars(rs, ts, 512, 640, num_threads=8)
    assert want is not None and res.method == want["method"] and res.inliers == want["inliers"]
    assert tuple(res.transform) == tuple(want["transform"])
