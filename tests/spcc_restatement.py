"""An independent restatement of SPCC after detection in plain Python floats.

Written from core/astrometry/spcc.rs:86-183 (filters, order, truncation, error paths), :198-279 (Teff(bp_rp), Planck colours, white
references, the synthetic catalogue), :285-339 (the cross-match -- done here as the real O(n^2) nearest-neighbour loop on world
coordinates of a linear WCS, NOT assumed to be the identity as oracle/orc_spcc.c does), :341-383 (aperture photometry) and
:385-435 (weighted chromaticity ratios).  Nothing is shared with oracle/orc_spcc.c or csrc/spcc.hip; no test of the reference
pins this file.  TEST INFRASTRUCTURE.
"""
import math

import numpy as np


def bp_rp_to_teff(bp_rp):                                             # :198-213
    x = min(max(bp_rp, -0.5), 5.0)
    if x < 0.0:
        return 10000.0 + (-x) * 20000.0
    if x < 0.5:
        return 7500.0 + (0.5 - x) * 5000.0
    if x < 1.0:
        return 5800.0 + (1.0 - x) * 3400.0
    if x < 1.5:
        return 4500.0 + (1.5 - x) * 2600.0
    if x < 2.5:
        return 3500.0 + (2.5 - x) * 1000.0
    return 2800.0 + (5.0 - x) * 280.0


def planck_intensity(teff, wavelength_nm):                            # :228-243
    lam = wavelength_nm * 1e-9
    h, c, k = 6.626e-34, 2.998e8, 1.381e-23
    exponent = h * c / (lam * k * teff)
    if exponent > 500.0:
        return 0.0
    lam5 = lam * lam * lam * lam * lam                                 # powi(5)
    return (2.0 * h * c * c / lam5) / (math.exp(exponent) - 1.0)


def planck_rgb(teff):                                                 # :215-226
    r, g, b = planck_intensity(teff, 640.0), planck_intensity(teff, 530.0), planck_intensity(teff, 460.0)
    m = max(max(r, g), b)
    if m < 1e-30:
        return 1.0, 1.0, 1.0
    return r / m, g / m, b / m


def white_reference_rgb(wr):                                          # :245-255
    if wr == "g2v":
        return planck_rgb(5778.0)
    if wr == "average_spiral":
        r, g, b = planck_rgb(5500.0)
        return r * 0.98, g * 1.0, b * 1.02
    if wr == "photopic":
        return 1.0, 1.0, 1.0
    return tuple(float(v) for v in wr)


def estimate_bp_rp_from_flux(star):                                   # :275-279
    norm_flux = min(max(star.flux / max(star.peak, 1e-10), 0.1), 100.0)
    fwhm_factor = min(max(star.fwhm - 3.0, -2.0), 5.0) * 0.1
    return min(max(1.0 / math.sqrt(norm_flux) + fwhm_factor, -0.3), 4.0)


def aperture_flux_f32(image, x, y, radius):                           # :341-383
    h, w = image.shape
    r2 = radius * radius
    inner_r2, outer = (radius * 1.2) ** 2, radius * 1.8
    outer_r2 = outer * outer

    def usize(v):
        return 0 if (v != v or v <= 0.0) else int(v)
    y_min, y_max = usize(max(math.floor(y - outer), 0.0)), min(usize(math.ceil(y + outer)), max(h - 1, 0))
    x_min, x_max = usize(max(math.floor(x - outer), 0.0)), min(usize(math.ceil(x + outer)), max(w - 1, 0))
    flux = bg_sum = 0.0
    bg_count = 0
    for py in range(y_min, y_max + 1):
        row = image[py]
        dy = float(py) - y
        for px in range(x_min, x_max + 1):
            dx = float(px) - x
            d2 = dx * dx + dy * dy
            v = float(row[px])
            if d2 <= r2:
                flux += v
            elif inner_r2 <= d2 <= outer_r2:
                bg_sum += v
                bg_count += 1
    if bg_count > 0:
        flux -= (bg_sum / bg_count) * (math.pi * r2)
    return max(flux, 0.0)


def compute_correction_factors(matched, wr):                          # :385-435; matched = [(bp_rp, r, g, b)]
    sr = sg = sb = sw = sci = 0.0
    for bp_rp, mr_, mg_, mb_ in matched:
        er_, eg_, eb_ = planck_rgb(bp_rp_to_teff(bp_rp))
        tm, te = mr_ + mg_ + mb_, er_ + eg_ + eb_
        if tm < 1e-10 or te < 1e-10:
            continue
        weight = math.sqrt(tm)
        mr, mg, mb = mr_ / tm, mg_ / tm, mb_ / tm
        er, eg, eb = er_ / te, eg_ / te, eb_ / te
        if mr > 1e-6:
            sr += (er / mr) * weight
        if mg > 1e-6:
            sg += (eg / mg) * weight
        if mb > 1e-6:
            sb += (eb / mb) * weight
        sw += weight
        sci += bp_rp
    if sw < 1e-10 or not matched:
        return 1.0, 1.0, 1.0, 0.0
    rf, gf, bf = sr / sw * wr[0], sg / sw * wr[1], sb / sw * wr[2]
    if gf > 1e-10:
        rf, bf, gf = rf / gf, bf / gf, 1.0
    return rf, gf, bf, sci / len(matched)


class SpccError(ValueError):
    pass


def spcc_from_detection(r, g, b, stars, lum_max, pixel_scale_arcsec, min_snr=20.0, max_stars=200, saturation_limit=0.90,
                        white_reference="average_spiral", ra0=83.8, dec0=-5.4):
    """spcc_calibrate_rgb after detect_stars / compute_image_stats (:86-183) -> (r, g, b factors, matched, total, avg colour index)"""
    h, w = r.shape
    sat_limit = float(np.float32(lum_max * saturation_limit))          # :89 `as f32`, compared `as f64` (:96)
    x_hi, y_hi = float((w - 10) % 2 ** 64), float((h - 10) % 2 ** 64)  # usize arithmetic (wraps below 10 in a release build)
    good = [s for s in stars if s.snr >= min_snr and s.peak < sat_limit and s.x >= 10.0 and s.y >= 10.0 and s.x < x_hi and s.y < y_hi]
    good.sort(key=lambda s: -s.snr)                                    # stable, descending snr (:104)
    good = good[:max_stars]
    if len(good) < 5:
        raise SpccError(f"Only {len(good)} stars passed quality filters (need 5+). Try lowering min_snr.")
    # a linear stand-in for the TAN solution: north up, east left, `pixel_scale_arcsec` per pixel around (ra0, dec0)
    s_deg = pixel_scale_arcsec / 3600.0
    cosd = math.cos(math.radians(dec0))
    world = [(ra0 - (s.x - w / 2.0) * s_deg / cosd, dec0 - (s.y - h / 2.0) * s_deg) for s in good]
    catalog = [(ra, dec, estimate_bp_rp_from_flux(s)) for (ra, dec), s in zip(world, good)]   # generate_synthetic_catalog (:257-273)
    match_radius = (pixel_scale_arcsec * 3.0) / 3600.0                 # :294-295
    match_r2 = match_radius * match_radius
    matched = []
    for s, (ra, dec) in zip(good, world):
        best_dist, best = float("inf"), None
        for (cra, cdec, cbp) in catalog:
            dra = ra - cra
            if dra > 180.0:
                dra -= 360.0
            elif dra < -180.0:
                dra += 360.0
            dra = dra * math.cos(math.radians(dec))
            ddec = dec - cdec
            d2 = dra * dra + ddec * ddec
            if d2 < match_r2 and d2 < best_dist:
                best_dist, best = d2, cbp
        if best is None:
            continue
        radius = max(s.fwhm * 1.5, 3.0)
        rf, gf, bf = (aperture_flux_f32(p, s.x, s.y, radius) for p in (r, g, b))
        if rf > 0.0 and gf > 0.0 and bf > 0.0:
            matched.append((best, rf, gf, bf))
    if len(matched) < 3:
        raise SpccError(f"Only {len(matched)} stars cross-matched (need 3+). Check WCS solution quality.")
    rf, gf, bf, ci = compute_correction_factors(matched, white_reference_rgb(white_reference))
    return rf, gf, bf, len(matched), len(good), ci
