"""An independent restatement of the star mask and the masked stretch in numpy.

Written from core/imaging/star_mask.rs:46-138 and core/imaging/masked_stretch.rs:60-118,195-254 alone -- nothing here is shared with
oracle/orc_masked.c or csrc/masked_stretch.hip, which are two restatements by one author and which no test of the reference pins
(VERDICT r4 missing 5).  f32 where the Rust is f32, f64 where it is f64, the operations in the source's order (numpy's float32
arithmetic is IEEE single precision with no fused multiply-add, as rustc's).  TEST INFRASTRUCTURE, like everything under oracle/.
"""
import math

import numpy as np

F = np.float32


def _as_usize(x: float) -> int:
    """Rust `f64 as usize`: saturating, NaN -> 0"""
    if x != x or x <= 0.0:
        return 0
    return int(min(x, 2.0 ** 63))


def star_mask_from_stars(image, stars, growth_factor=2.5, softness=4.0, min_fwhm=1.5, max_fwhm=30.0, luminance_protect=False,
                         luminance_ceiling=0.85):
    """generate_star_mask_from_detection (star_mask.rs:46-138); stars = [(x, y, fwhm)] -> (mask, stars_masked, coverage)"""
    image = np.asarray(image, F)
    h, w = image.shape
    valid = [(x, y, fw) for (x, y, fw) in stars if fw >= min_fwhm and fw <= max_fwhm]           # :53-57 (a NaN fwhm fails both)
    mask = np.zeros((h, w), F)
    for (sx, sy, fw) in valid:
        radius = fw * growth_factor                                                              # :64-65
        soft_radius = radius + softness
        y_min = _as_usize(max(math.floor(sy - soft_radius), 0.0))                                # :67-70
        y_max = min(_as_usize(math.ceil(sy + soft_radius)), max(h - 1, 0))
        x_min = _as_usize(max(math.floor(sx - soft_radius), 0.0))
        x_max = min(_as_usize(math.ceil(sx + soft_radius)), max(w - 1, 0))
        if y_min > y_max or x_min > x_max:
            continue
        r2_inner = radius * radius                                                               # :72-74
        r2_outer = soft_radius * soft_radius
        fade_range = max(r2_outer - r2_inner, 1e-10)
        py = np.arange(y_min, y_max + 1, dtype=np.float64)[:, None]
        px = np.arange(x_min, x_max + 1, dtype=np.float64)[None, :]
        dx, dy = px - sx, py - sy                                                                # :79-81
        d2 = dx * dx + dy * dy
        with np.errstate(invalid="ignore", over="ignore"):
            inner = d2 <= r2_inner
            ring = ~inner & (d2 <= r2_outer)
            t = ((d2 - r2_inner) / fade_range).astype(F)                                         # :86 (`as f32`)
            smooth = (t * t) * (F(3.0) - F(2.0) * t)                                             # :87, f32
            val = np.where(inner, F(1.0), F(1.0) - smooth).astype(F)
        hit = inner | ring                                                                       # else `continue` (:90)
        box = mask[y_min:y_max + 1, x_min:x_max + 1]
        upd = hit & (val > box)                                                                  # :101-106
        box[upd] = val[upd]
    if luminance_protect:                                                                        # :109-127
        ceiling = F(luminance_ceiling)
        inv_range = F(1.0) / (F(1.0) - ceiling) if ceiling < F(1.0) else F(1.0)
        with np.errstate(invalid="ignore"):
            touch = (image > ceiling) & (mask < F(1.0))
            excess = np.clip((image - ceiling) * inv_range, F(0.0), F(1.0)).astype(F)
            smooth = (excess * excess) * (F(3.0) - F(2.0) * excess)
            upd = touch & (smooth > mask)
        mask[upd] = smooth[upd]
    total = float(h * w)
    coverage = float(np.count_nonzero(mask > F(0.01))) / total if total else float("nan")      # :129-130
    return mask, len(valid), coverage


def _valid_min_max(image):
    """ImageStats.min / .max as masked_stretch's normalize_to_01 uses them (stats.rs:10-13,43-75: over finite pixels > 1e-7;
    ImageStats::default() = zeros when there is none)"""
    with np.errstate(invalid="ignore"):
        ok = np.isfinite(image) & (image > F(1e-7))
    if not ok.any():
        return 0.0, 0.0
    v = image[ok]
    return float(v.min()), float(v.max())


def normalize_to_01(image):                                                                      # masked_stretch.rs:195-213
    image = np.asarray(image, F)
    mn, mx = _valid_min_max(image)
    rng = F(mx - mn)                                                                             # `(stats.max - stats.min) as f32`
    if rng < F(1e-10):
        return np.zeros(image.shape, F)
    dmin = F(mn)
    inv = F(1.0) / rng
    with np.errstate(invalid="ignore", over="ignore"):
        out = np.clip((image - dmin) * inv, F(0.0), F(1.0)).astype(F)
        bad = ~np.isfinite(image) | (image <= F(0.0))
    out[bad] = F(0.0)
    return out


def masked_median(working, mask) -> float:                                                       # :215-233
    with np.errstate(invalid="ignore"):
        sel = (mask < F(0.5)) & np.isfinite(working) & (working > F(0.0))
    vals = working[sel]
    if vals.size == 0:
        return 0.0
    return float(np.partition(vals, vals.size // 2)[vals.size // 2])


def mtf_balance(median: float, target: float) -> float:                                          # :235-241
    denom = 2.0 * target * median - target - median
    if abs(denom) < 1e-15:
        return 0.5
    return min(max(median * (target - 1.0) / denom, 0.0001), 0.9999)


def apply_mtf(data, m):                                                                          # :243-259, m: f32
    m = F(m)
    with np.errstate(invalid="ignore", divide="ignore", over="ignore"):
        denom = (F(2.0) * m - F(1.0)) * data - m
        y = np.clip(((m - F(1.0)) * data) / denom, F(0.0), F(1.0)).astype(F)
        out = np.where(np.abs(denom) < F(1e-10), data, y).astype(F)
    out[data >= F(1.0)] = F(1.0)
    out[data <= F(0.0)] = F(0.0)
    return out


def masked_stretch_with_mask(image, mask, iterations=10, target_background=0.25, protection_amount=0.85, convergence_threshold=1e-5):
    """masked_stretch_with_mask (masked_stretch.rs:60-118) -> (image, iterations_run, final_background, converged)"""
    working = normalize_to_01(image)
    mask = np.asarray(mask, F)
    protection = F(protection_amount)
    prev_bg = masked_median(working, mask)
    iterations_run, converged = 0, False
    for it in range(iterations):
        iterations_run = it + 1
        bg = masked_median(working, mask)
        at_target = abs(bg - target_background) < convergence_threshold
        stagnated = it > 0 and abs(bg - prev_bg) < convergence_threshold * 0.1
        if at_target:
            converged = True
            break
        if stagnated:
            break
        midtone = mtf_balance(bg, target_background)
        unmasked = apply_mtf(working, F(midtone))
        blend = mask * protection                                                                # :96-98, f32
        working = (working * blend + unmasked * (F(1.0) - blend)).astype(F)
        prev_bg = bg
    final_bg = masked_median(working, mask)
    with np.errstate(invalid="ignore"):
        working = np.clip(working, F(0.0), F(1.0)).astype(F)                                     # clamp_inplace (:261-263)
    return working, iterations_run, final_bg, converged


def luminance(r, g, b):                                                                          # compute_luminance (:120-153)
    rn, gn, bn = (np.where(np.isfinite(x), x, F(0.0)).astype(F) for x in (np.asarray(r, F), np.asarray(g, F), np.asarray(b, F)))
    return ((F(0.2126) * rn + F(0.7152) * gn) + F(0.0722) * bn).astype(F)
