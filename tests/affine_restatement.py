"""An INDEPENDENT restatement of the reference's star matcher and transform fit, in plain Python floats (IEEE f64, one rounding
per operation, no fused multiply-add), written from the Rust and from nothing else:

    core/alignment/affine.rs:8-22     the constants
    core/alignment/affine.rs:74-97    AffineTransform::map, rotation_deg, scale_x, scale_y
    core/alignment/affine.rs:157-204  the driver after star detection (affine RANSAC, then rigid, else "no star solution")
    core/alignment/affine.rs:207-241  check_transform_sanity
    core/alignment/affine.rs:279-317  build_triangles
    core/alignment/affine.rs:319-383  match_triangles
    core/alignment/affine.rs:385-398  sort_triangle_vertices
    core/alignment/affine.rs:400-517  ransac_affine (xorshift streams, 20-attempt sampling, per-thread best, first-best reduce)
    core/alignment/affine.rs:519-587  fit_affine, solve_3x3_ls, solve_3x3 (adjugate inverse)
    core/alignment/affine.rs:589-640  fit_rigid
    core/alignment/affine.rs:642-660  compute_residual, dist

It shares no code with oracle/orc_affine.c nor with csrc/affine.hip (VERDICT r3 "weak" 3: those two are restatements by one
author; this is the third, in another language, that both are held to).  Two things the Rust leaves open are taken as
PARAMETERS of this file, because they are choices and not arithmetic:

  * the vote pairs come out of a std HashMap and are sorted by votes only (affine.rs:351-360): pairs with equal votes are in a
    random order per process.  `tie_break` orders them; the oracle's documented pin is (ref index, target index) ascending.
  * rayon's thread count decides how the 2000 RANSAC iterations are cut into xorshift streams (affine.rs:411-413) and
    `reduce_with` folds the per-thread bests in thread order keeping the earlier one on ties (:475).  `num_threads` is the
    caller's, as in the oracle and the C ABI.
"""
import math

import numpy as np

MAX_STARS = 120
TRIANGLE_TOLERANCE = 0.02
MIN_MATCHES_AFFINE = 6
MIN_MATCHES_RIGID = 4
RANSAC_ITERATIONS = 2000
RANSAC_INLIER_PX = 3.0
MIN_TRIANGLE_SIDE = 15.0
MIN_VOTES = 1
MIN_INLIER_RATIO = 0.20
MAX_RESIDUAL_PX = 5.0
MAX_OFFSET_FRACTION = 0.40
MAX_ROTATION_DEG = 30.0
MIN_SCALE = 0.70
MAX_SCALE = 1.40
U64 = (1 << 64) - 1


def dist(a, b):                                                   # :656-660  (powi(2) is x * x)
    dx, dy = a[0] - b[0], a[1] - b[1]
    return math.sqrt(dx * dx + dy * dy)


def tmap(t, x, y):                                                # :74-80, left to right
    a, b, tx, c, d, ty = t
    return (a * x + b * y + tx, c * x + d * y + ty)


def build_triangles(stars):                                       # :279-317
    n = len(stars)
    if n < 3:
        return []
    limit = min(n, 60)
    tris = []
    for i in range(limit):
        for j in range(i + 1, limit):
            for k in range(j + 1, limit):
                sides = sorted([dist(stars[i], stars[j]), dist(stars[j], stars[k]), dist(stars[i], stars[k])])
                if sides[0] < MIN_TRIANGLE_SIDE:
                    continue
                tris.append(((i, j, k), sides[1] / sides[0], sides[2] / sides[0]))
    return tris


def sort_triangle_vertices(stars, idx):                           # :385-398: by the opposite side, stable
    i, j, k = idx
    verts = [(i, dist(stars[j], stars[k])), (j, dist(stars[i], stars[k])), (k, dist(stars[i], stars[j]))]
    verts.sort(key=lambda v: v[1])
    return [v[0] for v in verts]


def match_triangles(ref_stars, tgt_stars, ref_tris, tgt_tris, tie_break=lambda pair: pair):   # :319-383
    votes = {}
    tgt_sorted = [sort_triangle_vertices(tgt_stars, tt[0]) for tt in tgt_tris]
    # (the two `> TRIANGLE_TOLERANCE` tests of every pair as numpy f64 array operations: the same subtractions and compares,
    # 34 220^2 of them for 60 + 60 stars)
    t_mid, t_long = np.array([tt[1] for tt in tgt_tris]), np.array([tt[2] for tt in tgt_tris])
    for rt in ref_tris:
        hits = np.nonzero(~(np.abs(rt[1] - t_mid) > TRIANGLE_TOLERANCE) & ~(np.abs(rt[2] - t_long) > TRIANGLE_TOLERANCE))[0]
        if hits.size == 0:
            continue
        ref_sorted = sort_triangle_vertices(ref_stars, rt[0])
        for h in hits:
            ts = tgt_sorted[h]
            for p in range(3):
                key = (ref_sorted[p], ts[p])
                votes[key] = votes.get(key, 0) + 1
    pairs = sorted(votes.items(), key=lambda kv: (-kv[1], tie_break(kv[0])))
    used_ref, used_tgt, matches = set(), set(), []
    for (ri, ti), v in pairs:
        if v < MIN_VOTES:
            break
        if ri in used_ref or ti in used_tgt:
            continue
        used_ref.add(ri)
        used_tgt.add(ti)
        matches.append((ref_stars[ri][0], ref_stars[ri][1], tgt_stars[ti][0], tgt_stars[ti][1]))
    return matches


def solve_3x3(a, b):                                              # :552-587
    det = (a[0][0] * (a[1][1] * a[2][2] - a[1][2] * a[2][1]) - a[0][1] * (a[1][0] * a[2][2] - a[1][2] * a[2][0])
           + a[0][2] * (a[1][0] * a[2][1] - a[1][1] * a[2][0]))
    if abs(det) < 1e-12:
        return None
    inv_det = 1.0 / det
    inv = [[(a[1][1] * a[2][2] - a[1][2] * a[2][1]) * inv_det, (a[0][2] * a[2][1] - a[0][1] * a[2][2]) * inv_det,
            (a[0][1] * a[1][2] - a[0][2] * a[1][1]) * inv_det],
           [(a[1][2] * a[2][0] - a[1][0] * a[2][2]) * inv_det, (a[0][0] * a[2][2] - a[0][2] * a[2][0]) * inv_det,
            (a[0][2] * a[1][0] - a[0][0] * a[1][2]) * inv_det],
           [(a[1][0] * a[2][1] - a[1][1] * a[2][0]) * inv_det, (a[0][1] * a[2][0] - a[0][0] * a[2][1]) * inv_det,
            (a[0][0] * a[1][1] - a[0][1] * a[1][0]) * inv_det]]
    return [inv[r][0] * b[0] + inv[r][1] * b[1] + inv[r][2] * b[2] for r in range(3)]


def solve_3x3_ls(matches, solve_x):                               # :532-550
    ata = [[0.0] * 3 for _ in range(3)]
    atb = [0.0] * 3
    for rx, ry, tx, ty in matches:
        target = tx if solve_x else ty
        row = (rx, ry, 1.0)
        for i in range(3):
            for j in range(3):
                ata[i][j] += row[i] * row[j]
            atb[i] += row[i] * target
    return solve_3x3(ata, atb)


def fit_affine(matches):                                          # :519-530
    if len(matches) < 3:
        return None
    x = solve_3x3_ls(matches, True)
    if x is None:
        return None
    y = solve_3x3_ls(matches, False)
    if y is None:
        return None
    return (x[0], x[1], x[2], y[0], y[1], y[2])


def fit_rigid(matches):                                           # :589-640
    n = len(matches)
    if n < 2:
        return None
    rcx = rcy = tcx = tcy = 0.0
    for rx, ry, tx, ty in matches:
        rcx += rx
        rcy += ry
        tcx += tx
        tcy += ty
    nf = float(n)
    rcx /= nf
    rcy /= nf
    tcx /= nf
    tcy /= nf
    num = den = 0.0
    for rx, ry, tx, ty in matches:
        drx, dry, dtx, dty = rx - rcx, ry - rcy, tx - tcx, ty - tcy
        num += drx * dty - dry * dtx
        den += drx * dtx + dry * dty
    theta = math.atan2(num, den)
    cos_t, sin_t = math.cos(theta), math.sin(theta)
    tx = tcx - cos_t * rcx + sin_t * rcy
    ty = tcy - sin_t * rcx - cos_t * rcy
    return (cos_t, -sin_t, tx, sin_t, cos_t, ty)


def compute_residual(matches, t):                                 # :642-654
    if not matches:
        return 0.0
    s = 0.0
    for rx, ry, tx, ty in matches:
        px, py = tmap(t, rx, ry)
        ex, ey = px - tx, py - ty
        s += math.sqrt(ex * ex + ey * ey)
    return s / float(len(matches))


def ransac_affine(matches, method, num_threads):                  # :400-517; method "affine" | "rigid"
    n = len(matches)
    min_sample = 3 if method == "affine" else 2
    if n < min_sample:
        return None
    num_threads = max(num_threads, 1)
    chunk = (RANSAC_ITERATIONS + num_threads - 1) // num_threads
    fit = fit_affine if method == "affine" else fit_rigid
    best = None  # (inliers, transform, mask): reduce_with keeps the EARLIER thread on ties (b.0 > a.0 replaces)
    for thread_id in range(num_threads):
        state = (0xDEADBEEFCAFEBABE + thread_id * 0x9E3779B97F4A7C15) & U64
        local = (0, (1.0, 0.0, 0.0, 0.0, 1.0, 0.0), [False] * n)
        for _ in range(chunk):
            sample, attempts = [], 0
            while len(sample) < min_sample and attempts < 20:
                state ^= (state << 13) & U64
                state ^= state >> 7
                state ^= (state << 17) & U64
                idx = state % n
                if idx not in sample:
                    sample.append(idx)
                attempts += 1
            if len(sample) < min_sample:
                continue
            t = fit([matches[i] for i in sample])
            if t is None:
                continue
            mask, count = [False] * n, 0
            for i, (rx, ry, tx, ty) in enumerate(matches):
                px, py = tmap(t, rx, ry)
                ex, ey = px - tx, py - ty
                if math.sqrt(ex * ex + ey * ey) < RANSAC_INLIER_PX:
                    count += 1
                    mask[i] = True
            if count > local[0]:
                local = (count, t, mask)
        if best is None or local[0] > best[0]:
            best = local
    best_inliers, best_t, best_mask = best
    if best_inliers < MIN_MATCHES_RIGID:
        return None
    if best_inliers / float(n) < MIN_INLIER_RATIO:
        return None
    inl = [m for m, keep in zip(matches, best_mask) if keep]
    refined = fit(inl)
    if refined is None:
        refined = best_t
    residual = compute_residual(inl, refined)
    if residual > MAX_RESIDUAL_PX:
        return None
    return dict(transform=refined, matched_stars=n, inliers=best_inliers, residual_px=residual, method=method)


def check_transform_sanity(res, rows, cols):                      # :207-241 -> True when accepted
    a, b, tx, c, d, ty = res["transform"]
    if abs(tx) > cols * MAX_OFFSET_FRACTION or abs(ty) > rows * MAX_OFFSET_FRACTION:
        return False
    if abs(math.degrees(math.atan2(c, a))) > MAX_ROTATION_DEG:
        return False
    sx, sy = math.sqrt(a * a + c * c), math.sqrt(b * b + d * d)
    return not (sx < MIN_SCALE or sx > MAX_SCALE or sy < MIN_SCALE or sy > MAX_SCALE)


def affine_from_stars(ref_stars, tgt_stars, rows, cols, num_threads=8, tie_break=lambda pair: pair):
    """affine.rs:157-204 from the two star lists (brightest first) on: the accepted result, or None where the reference falls
    back to phase correlation."""
    ref_stars = [tuple(map(float, s)) for s in ref_stars][:MAX_STARS]
    tgt_stars = [tuple(map(float, s)) for s in tgt_stars][:MAX_STARS]
    if len(ref_stars) < MIN_MATCHES_RIGID or len(tgt_stars) < MIN_MATCHES_RIGID:
        return None
    ref_tris, tgt_tris = build_triangles(ref_stars), build_triangles(tgt_stars)
    if not ref_tris or not tgt_tris:
        return None
    matches = match_triangles(ref_stars, tgt_stars, ref_tris, tgt_tris, tie_break)
    if len(matches) < MIN_MATCHES_RIGID:
        return None
    if len(matches) >= MIN_MATCHES_AFFINE:
        res = ransac_affine(matches, "affine", num_threads)
        if res is not None and check_transform_sanity(res, rows, cols):
            return res
    res = ransac_affine(matches, "rigid", num_threads)
    if res is not None and check_transform_sanity(res, rows, cols):
        return res
    return None
