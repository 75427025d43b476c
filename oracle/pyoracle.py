"""ctypes binding of the CPU ORACLE (test infrastructure, NOT product code).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module.  It loads oracle/liboracle.so (the plain-C restatement of the
reference's Rust hot path, see oracle/ab_oracle.h) and exposes numpy-friendly
wrappers named after the reference functions they restate.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from dataclasses import dataclass

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "liboracle.so")

ORDER_SELECT = 0
ORDER_ASCENDING = 1


def build(force: bool = False) -> str:
    """Compile oracle/*.c into liboracle.so (gcc, see oracle/Makefile)."""
    if force or not os.path.exists(_LIB_PATH):
        subprocess.check_call(["make", "-C", _HERE, "-s"] + (["-B"] if force else []))
    return _LIB_PATH


class _Stats(C.Structure):
    _fields_ = [("min", C.c_double), ("max", C.c_double), ("median", C.c_double), ("mad", C.c_double),
                ("sigma", C.c_double), ("mean", C.c_double), ("valid_count", C.c_uint64)]


class _Stf(C.Structure):
    _fields_ = [("shadow", C.c_double), ("midtone", C.c_double), ("highlight", C.c_double)]


@dataclass
class ImageStats:  # types/image.rs:2-10
    min: float = 0.0
    max: float = 0.0
    median: float = 0.0
    mad: float = 0.0
    sigma: float = 0.0
    mean: float = 0.0
    valid_count: int = 0


@dataclass
class StfParams:  # types/image.rs:36-40
    shadow: float = 0.0
    midtone: float = 0.5
    highlight: float = 1.0


_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    build()
    try:
        L = C.CDLL(_LIB_PATH)
    except OSError:
        build(force=True)
        L = C.CDLL(_LIB_PATH)
    fp = C.POINTER(C.c_float)
    L.orc_f32_cmp.restype = C.c_int
    L.orc_f32_cmp.argtypes = [C.c_float, C.c_float]
    L.orc_select_nth_f32.argtypes = [fp, C.c_size_t, C.c_size_t]
    L.orc_exact_median_mut.restype = C.c_double
    L.orc_exact_median_mut.argtypes = [fp, C.c_size_t]
    L.orc_median_f32_mut.restype = C.c_float
    L.orc_median_f32_mut.argtypes = [fp, C.c_size_t]
    L.orc_exact_mad_mut.restype = C.c_float
    L.orc_exact_mad_mut.argtypes = [fp, C.c_size_t, C.c_float]
    L.orc_sigma_clipped_stats.argtypes = [fp, C.POINTER(C.c_size_t), C.c_float, C.c_size_t,
                                          C.POINTER(C.c_double), C.POINTER(C.c_double)]
    L.orc_sigma_clip_combine.restype = C.c_float
    L.orc_sigma_clip_combine.argtypes = [fp, C.c_size_t, C.c_float, C.c_float, C.c_size_t, C.c_int,
                                         C.POINTER(C.c_uint32)]
    L.orc_stack_images_noalign.restype = C.c_int
    L.orc_stack_images_noalign.argtypes = [C.POINTER(fp), C.POINTER(C.c_int64), C.POINTER(C.c_int64),
                                           C.c_size_t, C.c_float, C.c_float, C.c_size_t, C.c_int, C.c_int,
                                           fp, C.POINTER(C.c_uint64), C.POINTER(C.c_int64),
                                           C.POINTER(C.c_int64)]
    L.orc_stack_partial_noalign.argtypes = [C.POINTER(fp), C.c_size_t, C.c_int64, C.c_float, C.c_float,
                                            C.c_size_t, C.c_int, C.POINTER(C.c_double),
                                            C.POINTER(C.c_uint32), C.POINTER(C.c_uint64)]
    L.orc_catmull_rom.restype = C.c_double
    L.orc_catmull_rom.argtypes = [C.c_double]
    L.orc_clamp_index.restype = C.c_size_t
    L.orc_clamp_index.argtypes = [C.c_int64, C.c_size_t]
    for name in ("orc_nearest_sample", "orc_bilinear_sample", "orc_bicubic_sample"):
        f = getattr(L, name)
        f.restype = C.c_float
        f.argtypes = [fp, C.c_size_t, C.c_size_t, C.c_double, C.c_double]
    L.orc_shift_image_subpixel.argtypes = [fp, C.c_size_t, C.c_size_t, C.c_double, C.c_double, C.c_int, fp]
    L.orc_warp_image.argtypes = [fp, C.c_size_t, C.c_size_t, C.POINTER(C.c_double), C.c_size_t, C.c_size_t,
                                 C.c_int, fp]
    sp = C.POINTER(_Stats)
    for name in ("orc_compute_image_stats", "orc_compute_image_stats_exact", "orc_compute_image_stats_hist"):
        getattr(L, name).argtypes = [fp, C.c_size_t, sp]
    L.orc_compute_image_stats_with_known_range.argtypes = [fp, C.c_size_t, C.c_double, C.c_double, sp]
    L.orc_build_histogram.restype = C.c_int
    L.orc_build_histogram.argtypes = [fp, C.c_size_t, C.c_size_t, C.c_double, C.c_double,
                                      C.POINTER(C.c_uint32)]
    L.orc_stats_value_hist.argtypes = [fp, C.c_size_t, C.c_double, C.c_double, C.POINTER(C.c_uint64),
                                       C.POINTER(C.c_double), C.POINTER(C.c_uint64)]
    L.orc_auto_stf.argtypes = [sp, C.c_double, C.c_double, C.POINTER(_Stf)]
    L.orc_mtf.restype = C.c_double
    L.orc_mtf.argtypes = [C.c_double, C.c_double]
    L.orc_mtf_balance.restype = C.c_double
    L.orc_mtf_balance.argtypes = [C.c_double, C.c_double]
    L.orc_apply_stf_u8.argtypes = [fp, C.c_size_t, C.POINTER(_Stf), sp, C.c_int, C.POINTER(C.c_uint8)]
    L.orc_apply_stf_f32.argtypes = [fp, C.c_size_t, C.POINTER(_Stf), sp, C.c_int, fp]
    L.orc_max_threads.restype = C.c_int
    L.orc_apply_scnr_inplace.argtypes = [fp, fp, fp, C.c_size_t, C.c_int, C.c_float, C.c_int]
    L.orc_blend_channels.argtypes = [C.POINTER(fp), C.c_size_t, C.POINTER(C.c_double), C.c_size_t, C.c_size_t,
                                     fp, fp, fp]
    L.orc_spline_lut_from_points.argtypes = [C.POINTER(C.c_double), C.c_size_t, fp]
    L.orc_apply_curve.argtypes = [fp, C.c_size_t, fp, fp]
    L.orc_apply_levels.argtypes = [fp, C.c_size_t, C.c_double, C.c_double, C.c_double, fp]
    L.orc_arcsinh_stretch_with_stats.argtypes = [fp, C.c_size_t, C.c_float, C.c_float, C.c_float, C.c_float, fp]
    L.orc_luminance.argtypes = [fp, fp, fp, C.c_size_t, fp]
    L.orc_scale.argtypes = [fp, C.c_size_t, C.c_float, fp]
    L.orc_calibrate_image.argtypes = [fp, fp, fp, fp, C.c_float, C.c_size_t, fp]
    L.orc_median_combine.argtypes = [C.POINTER(fp), C.c_size_t, C.c_size_t, fp]
    _lib = L
    return L


# ---- phase correlation (orc_phasecorr.c) ---------------------------------------------------------------
def _pc_protos():
    L = lib()
    if getattr(L, "_pc_ready", False):
        return L
    fp = C.POINTER(C.c_float)
    dp = C.POINTER(C.c_double)
    L.orc_phase_correlate.argtypes = [fp, C.c_size_t, C.c_size_t, fp, C.c_size_t, C.c_size_t, dp, dp, dp]
    L.orc_correlate_single.argtypes = [fp, fp, C.c_size_t, C.c_size_t, dp, dp, dp, dp]
    L.orc_area_downsample.argtypes = [fp, C.c_size_t, C.c_size_t, C.c_size_t, C.c_size_t, fp]
    L.orc_fft2d.argtypes = [dp, C.c_size_t, C.c_size_t, C.c_int]
    L.orc_stack_images_align.restype = C.c_int
    L.orc_stack_images_align.argtypes = [C.POINTER(fp), C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.c_size_t,
                                         C.c_float, C.c_float, C.c_size_t, C.c_int, C.c_int, fp,
                                         C.POINTER(C.c_uint64), C.POINTER(C.c_int32)]
    L._pc_ready = True
    return L


def phase_correlate(reference, target):
    r, t = _f32(reference), _f32(target)
    dx, dy, cf = C.c_double(), C.c_double(), C.c_double()
    _pc_protos().orc_phase_correlate(_fp(r), r.shape[0], r.shape[1], _fp(t), t.shape[0], t.shape[1], C.byref(dx),
                                     C.byref(dy), C.byref(cf))
    return dx.value, dy.value, cf.value


def correlate_single(a, b, want_surface=False):
    a, b = _f32(a), _f32(b)
    fr = 1 << max(0, (a.shape[0] - 1).bit_length())
    fc = 1 << max(0, (a.shape[1] - 1).bit_length())
    surf = np.zeros((fr, fc), np.float64)
    dx, dy, cf = C.c_double(), C.c_double(), C.c_double()
    _pc_protos().orc_correlate_single(_fp(a), _fp(b), a.shape[0], a.shape[1], C.byref(dx), C.byref(dy), C.byref(cf),
                                      surf.ctypes.data_as(C.POINTER(C.c_double)))
    return (dx.value, dy.value, cf.value, surf) if want_surface else (dx.value, dy.value, cf.value)


def area_downsample(image, out_rows, out_cols) -> np.ndarray:
    im = _f32(image)
    out = np.zeros((out_rows, out_cols), np.float32)
    _pc_protos().orc_area_downsample(_fp(im), im.shape[0], im.shape[1], out_rows, out_cols, _fp(out))
    return out


def fft2d(buf, inverse=False) -> np.ndarray:
    z = np.ascontiguousarray(buf, dtype=np.complex128).copy()
    _pc_protos().orc_fft2d(z.view(np.float64).ctypes.data_as(C.POINTER(C.c_double)), z.shape[0], z.shape[1],
                           1 if inverse else 0)
    return z


def stack_images_align(images, sigma_low=3.0, sigma_high=3.0, max_iterations=5, order=ORDER_ASCENDING, threads=0):
    """combine.rs:94-193 with align=true.  Returns (image, rejected_pixels, offsets[(dy, dx)])."""
    imgs = [_f32(im) for im in images]
    n = len(imgs)
    if n == 0:
        raise ValueError("No images to stack")
    rows = (C.c_int64 * n)(*[im.shape[0] for im in imgs])
    cols = (C.c_int64 * n)(*[im.shape[1] for im in imgs])
    ptrs = (C.POINTER(C.c_float) * n)(*[_fp(im) for im in imgs])
    out = np.zeros((min(im.shape[0] for im in imgs), min(im.shape[1] for im in imgs)), np.float32)
    rej = C.c_uint64(0)
    offs = (C.c_int32 * (2 * n))()
    _pc_protos().orc_stack_images_align(ptrs, rows, cols, n, sigma_low, sigma_high, max_iterations, order, threads,
                                        _fp(out), C.byref(rej), offs)
    return out, int(rej.value), [(int(offs[2 * i]), int(offs[2 * i + 1])) for i in range(n)]


# ---- star detection / affine registration (orc_detect.c, orc_affine.c) -----------------------------------
class _Star(C.Structure):
    _fields_ = [("x", C.c_double), ("y", C.c_double), ("flux", C.c_double), ("fwhm", C.c_double),
                ("eccentricity", C.c_double), ("peak", C.c_double), ("snr", C.c_double), ("npix", C.c_uint64),
                ("order", C.c_uint64)]


class _Affine(C.Structure):
    _fields_ = [("t", C.c_double * 6), ("matched_stars", C.c_uint64), ("inliers", C.c_uint64),
                ("residual_px", C.c_double), ("method", C.c_int32)]


@dataclass
class DetectedStar:  # star_detection.rs:10-20
    x: float
    y: float
    flux: float
    fwhm: float
    eccentricity: float
    peak: float
    npix: int
    snr: float


@dataclass
class AffineAlignResult:  # affine.rs:82-89
    transform: tuple
    matched_stars: int
    inliers: int
    residual_px: float
    method: str


AFFINE_METHODS = ("affine", "rigid", "phase_correlation", "identity")


def _det_protos():
    L = lib()
    if getattr(L, "_det_ready", False):
        return L
    fp, dp = C.POINTER(C.c_float), C.POINTER(C.c_double)
    L.orc_estimate_background.argtypes = [fp, C.c_size_t, C.c_size_t, C.c_size_t, dp, dp]
    L.orc_detect_stars.restype = C.c_size_t
    L.orc_detect_stars.argtypes = [fp, C.c_size_t, C.c_size_t, C.c_double, C.POINTER(_Star), C.c_size_t,
                                   C.POINTER(C.c_size_t), dp, dp]
    L.orc_normalize_for_detection.restype = C.c_int
    L.orc_normalize_for_detection.argtypes = [fp, C.c_size_t, fp]
    L.orc_align_channel_affine.argtypes = [fp, fp, C.c_size_t, C.c_size_t, C.c_int, C.POINTER(_Affine)]
    L.orc_affine_from_stars.restype = C.c_int
    L.orc_affine_from_stars.argtypes = [dp, C.c_size_t, dp, C.c_size_t, C.c_size_t, C.c_size_t, C.c_int,
                                        C.POINTER(_Affine)]
    L.orc_fit_rigid.restype = C.c_int
    L.orc_fit_rigid.argtypes = [dp, C.c_size_t, dp]
    L.orc_fit_affine.restype = C.c_int
    L.orc_fit_affine.argtypes = [dp, C.c_size_t, dp]
    L._det_ready = True
    return L


def estimate_background(image, tile_size):
    im = _f32(image)
    m, s = C.c_double(), C.c_double()
    _det_protos().orc_estimate_background(_fp(im), im.shape[0], im.shape[1], tile_size, C.byref(m), C.byref(s))
    return m.value, s.value


def detect_stars(image, sigma_threshold, max_stars=100000):
    """detect_stars (star_detection.rs:86-258) -> (stars, background_median, background_sigma)"""
    im = _f32(image)
    buf = (_Star * max_stars)()
    tot = C.c_size_t(0)
    m, s = C.c_double(), C.c_double()
    n = _det_protos().orc_detect_stars(_fp(im), im.shape[0], im.shape[1], sigma_threshold, buf, max_stars,
                                       C.byref(tot), C.byref(m), C.byref(s))
    stars = [DetectedStar(b.x, b.y, b.flux, b.fwhm, b.eccentricity, b.peak, int(b.npix), b.snr) for b in buf[:n]]
    return stars, m.value, s.value


def normalize_for_detection(image) -> np.ndarray:
    im = _f32(image)
    out = np.zeros_like(im)
    _det_protos().orc_normalize_for_detection(_fp(im), im.size, _fp(out))
    return out


def _affine_out(a: _Affine) -> AffineAlignResult:
    return AffineAlignResult(tuple(a.t), int(a.matched_stars), int(a.inliers), a.residual_px, AFFINE_METHODS[a.method])


def align_channel_affine(reference, target, num_threads=8) -> AffineAlignResult:
    r, t = _f32(reference), _f32(target)
    _pc_protos()
    a = _Affine()
    _det_protos().orc_align_channel_affine(_fp(r), _fp(t), r.shape[0], r.shape[1], num_threads, C.byref(a))
    return _affine_out(a)


def affine_from_stars(ref_xy, tgt_xy, rows, cols, num_threads=8):
    r = np.ascontiguousarray(np.asarray(ref_xy, np.float64).reshape(-1, 2))
    t = np.ascontiguousarray(np.asarray(tgt_xy, np.float64).reshape(-1, 2))
    a = _Affine()
    ok = _det_protos().orc_affine_from_stars(r.ctypes.data_as(C.POINTER(C.c_double)), r.shape[0],
                                             t.ctypes.data_as(C.POINTER(C.c_double)), t.shape[0], rows, cols,
                                             num_threads, C.byref(a))
    return _affine_out(a) if ok else None


def fit_rigid(matches):
    m = np.ascontiguousarray(np.asarray(matches, np.float64).reshape(-1, 4))
    t = (C.c_double * 6)()
    ok = _det_protos().orc_fit_rigid(m.ctypes.data_as(C.POINTER(C.c_double)), m.shape[0], t)
    return tuple(t) if ok else None


def fit_affine(matches):
    m = np.ascontiguousarray(np.asarray(matches, np.float64).reshape(-1, 4))
    t = (C.c_double * 6)()
    ok = _det_protos().orc_fit_affine(m.ctypes.data_as(C.POINTER(C.c_double)), m.shape[0], t)
    return tuple(t) if ok else None


# ---- colour / tone / calibration maps (orc_color.c) ------------------------------------------------
def apply_scnr(r, g, b, method="average", amount=1.0, preserve_luminance=False):
    """returns new (r, g, b) after apply_scnr_inplace (scnr.rs:18-53)"""
    r, g, b = _f32(r).copy(), _f32(g).copy(), _f32(b).copy()
    if r.shape != g.shape or g.shape != b.shape:
        return r, g, b
    lib().orc_apply_scnr_inplace(_fp(r), _fp(g), _fp(b), r.size, 0 if method in ("average", 0) else 1, amount,
                                 1 if preserve_luminance else 0)
    return r, g, b


def blend_channels(channels, weights, rows, cols):
    chans = [_f32(c) for c in channels]
    n = len(chans)
    ptrs = (C.POINTER(C.c_float) * n)(*[_fp(c) for c in chans])
    w = np.ascontiguousarray(np.asarray([[float(x) for x in ww] for ww in weights], np.float64).reshape(-1, 4))
    outs = [np.zeros((rows, cols), np.float32) for _ in range(3)]
    lib().orc_blend_channels(ptrs, n, w.ctypes.data_as(C.POINTER(C.c_double)), w.shape[0], rows * cols,
                             _fp(outs[0]), _fp(outs[1]), _fp(outs[2]))
    return tuple(outs)


def spline_lut_from_points(points) -> np.ndarray:
    pts = np.ascontiguousarray(np.asarray(points, np.float64).reshape(-1, 2))
    lut = np.zeros(4096, np.float32)
    lib().orc_spline_lut_from_points(pts.ctypes.data_as(C.POINTER(C.c_double)), pts.shape[0], _fp(lut))
    return lut


def apply_curve(image, lut) -> np.ndarray:
    im = _f32(image)
    out = np.zeros_like(im)
    lut = _f32(lut)
    lib().orc_apply_curve(_fp(im), im.size, _fp(lut), _fp(out))
    return out


def apply_levels(image, black=0.0, gamma=1.0, white=1.0) -> np.ndarray:
    im = _f32(image)
    out = np.zeros_like(im)
    lib().orc_apply_levels(_fp(im), im.size, black, gamma, white, _fp(out))
    return out


def arcsinh_stretch_with_stats(image, dmin, dmax, factor, gamma=1.0) -> np.ndarray:
    im = _f32(image)
    out = np.zeros_like(im)
    lib().orc_arcsinh_stretch_with_stats(_fp(im), im.size, dmin, dmax, factor, gamma, _fp(out))
    return out


def luminance(r, g, b) -> np.ndarray:
    r, g, b = _f32(r), _f32(g), _f32(b)
    out = np.zeros_like(r)
    lib().orc_luminance(_fp(r), _fp(g), _fp(b), r.size, _fp(out))
    return out


def scale(image, factor) -> np.ndarray:
    im = _f32(image)
    out = np.zeros_like(im)
    lib().orc_scale(_fp(im), im.size, factor, _fp(out))
    return out


def calibrate_image(raw, master_bias=None, master_dark=None, master_flat=None, dark_exposure_ratio=1.0) -> np.ndarray:
    raw = _f32(raw)
    opt = [None if x is None else _f32(x) for x in (master_bias, master_dark, master_flat)]
    out = np.zeros_like(raw)
    lib().orc_calibrate_image(_fp(raw), *[None if x is None else _fp(x) for x in opt], dark_exposure_ratio, raw.size,
                              _fp(out))
    return out


def median_combine(frames) -> np.ndarray:
    fr = [_f32(f) for f in frames]
    ptrs = (C.POINTER(C.c_float) * len(fr))(*[_fp(f) for f in fr])
    out = np.zeros_like(fr[0])
    lib().orc_median_combine(ptrs, len(fr), fr[0].size, _fp(out))
    return out


@dataclass
class BackgroundResult:  # background.rs:35-42 (+ fitted coefficients)
    model: np.ndarray
    corrected: np.ndarray
    sample_count: int
    rms_residual: float
    coeffs: np.ndarray


_BG_ERRORS = {1: "Image too small for grid_size={grid}",
              2: "Not enough background samples ({n}) for polynomial degree {degree}",
              3: "Failed to solve polynomial fit: Singular matrix in polynomial fit"}


def extract_background(image, grid_size=8, poly_degree=3, sigma_clip=2.5, iterations=3, mode=0) -> BackgroundResult:
    """background.rs:55-116; raises ValueError with the reference's message on its Err paths."""
    L = lib()
    L.orc_extract_background.restype = C.c_int
    L.orc_extract_background.argtypes = [C.POINTER(C.c_float), C.c_size_t, C.c_size_t, C.c_size_t, C.c_size_t, C.c_float,
                                         C.c_size_t, C.c_int, C.POINTER(C.c_float), C.POINTER(C.c_float),
                                         C.POINTER(C.c_size_t), C.POINTER(C.c_double), C.POINTER(C.c_double)]
    im = _f32(image)
    model, corr = np.zeros_like(im), np.zeros_like(im)
    ns, rms = C.c_size_t(0), C.c_double(0.0)
    coeffs = np.zeros(21, dtype=np.float64)
    rc = L.orc_extract_background(_fp(im), im.shape[0], im.shape[1], grid_size, poly_degree, sigma_clip, iterations, mode,
                                  _fp(model), _fp(corr), C.byref(ns), C.byref(rms), coeffs.ctypes.data_as(C.POINTER(C.c_double)))
    if rc != 0:
        raise ValueError(_BG_ERRORS[rc].format(grid=grid_size, n=ns.value, degree=poly_degree))
    return BackgroundResult(model, corr, int(ns.value), float(rms.value), coeffs)


@dataclass
class StarMaskResult:  # star_mask.rs:32-37
    mask: np.ndarray
    stars_masked: int
    coverage_fraction: float


@dataclass
class MaskedStretchResult:  # masked_stretch.rs:34-42
    image: np.ndarray
    iterations_run: int
    final_background: float
    stars_masked: int
    mask_coverage: float
    converged: bool


def generate_star_mask(image, growth_factor=2.5, softness=4.0, detection_sigma=5.0, min_fwhm=1.5, max_fwhm=30.0,
                       luminance_protect=False, luminance_ceiling=0.85, stars=None) -> StarMaskResult:
    """star_mask.rs:38-138; stars = [(x, y, fwhm)] or DetectedStar list selects generate_star_mask_from_detection."""
    L = lib()
    dp = C.POINTER(C.c_double)
    L.orc_star_mask_from_stars.restype = C.c_size_t
    L.orc_star_mask_from_stars.argtypes = [C.POINTER(C.c_float), C.c_size_t, C.c_size_t, dp, dp, dp, C.c_size_t, C.c_double,
                                           C.c_double, C.c_double, C.c_double, C.c_int, C.c_double, C.POINTER(C.c_float), dp]
    L.orc_generate_star_mask.restype = C.c_size_t
    L.orc_generate_star_mask.argtypes = [C.POINTER(C.c_float), C.c_size_t, C.c_size_t, C.c_double, C.c_double, C.c_double,
                                         C.c_double, C.c_double, C.c_int, C.c_double, C.POINTER(C.c_float), dp]
    im = _f32(image)
    mask = np.zeros_like(im)
    cov = C.c_double(0.0)
    if stars is None:
        n = L.orc_generate_star_mask(_fp(im), im.shape[0], im.shape[1], growth_factor, softness, detection_sigma, min_fwhm,
                                     max_fwhm, int(bool(luminance_protect)), luminance_ceiling, _fp(mask), C.byref(cov))
    else:
        xyz = np.array([(s.x, s.y, s.fwhm) if hasattr(s, "fwhm") else tuple(s) for s in stars], np.float64).reshape(-1, 3)
        xs, ys, fw = (np.ascontiguousarray(xyz[:, k]) for k in range(3))
        n = L.orc_star_mask_from_stars(_fp(im), im.shape[0], im.shape[1], xs.ctypes.data_as(dp), ys.ctypes.data_as(dp),
                                       fw.ctypes.data_as(dp), len(xs), growth_factor, softness, min_fwhm, max_fwhm,
                                       int(bool(luminance_protect)), luminance_ceiling, _fp(mask), C.byref(cov))
    return StarMaskResult(mask, int(n), float(cov.value))


def masked_stretch(image, iterations=10, target_background=0.25, mask_growth=2.5, mask_softness=4.0, luminance_protect=True,
                   luminance_ceiling=0.85, protection_amount=0.85, convergence_threshold=1e-5, mask=None) -> MaskedStretchResult:
    """masked_stretch (masked_stretch.rs:44-58) / masked_stretch_with_mask (:60-118) when mask is given."""
    L = lib()
    L.orc_masked_stretch_with_mask.restype = None
    L.orc_masked_stretch_with_mask.argtypes = [C.POINTER(C.c_float), C.POINTER(C.c_float), C.c_size_t, C.c_size_t, C.c_double,
                                               C.c_double, C.c_double, C.POINTER(C.c_float), C.POINTER(C.c_size_t),
                                               C.POINTER(C.c_double), C.POINTER(C.c_int)]
    im = _f32(image)
    if mask is None:
        mask = generate_star_mask(im, growth_factor=mask_growth, softness=mask_softness, luminance_protect=luminance_protect,
                                  luminance_ceiling=luminance_ceiling)
    mk = _f32(mask.mask)
    out = np.zeros_like(im)
    it, fb, cv = C.c_size_t(0), C.c_double(0.0), C.c_int(0)
    L.orc_masked_stretch_with_mask(_fp(im), _fp(mk), im.size, iterations, target_background, protection_amount,
                                   convergence_threshold, _fp(out), C.byref(it), C.byref(fb), C.byref(cv))
    return MaskedStretchResult(out, int(it.value), float(fb.value), mask.stars_masked, mask.coverage_fraction, bool(cv.value))


def masked_stretch_luminance(r, g, b) -> np.ndarray:
    """compute_luminance (masked_stretch.rs:120-153): non-finite -> 0, 0.2126 r + 0.7152 g + 0.0722 b in f32."""
    rn, gn, bn = (np.where(np.isfinite(x), x, np.float32(0)).astype(np.float32) for x in (_f32(r), _f32(g), _f32(b)))
    return (np.float32(0.2126) * rn + np.float32(0.7152) * gn) + np.float32(0.0722) * bn


def masked_stretch_rgb_shared(r, g, b, **cfg):
    """masked_stretch_rgb_shared (masked_stretch.rs:155-193)."""
    lum = masked_stretch_luminance(r, g, b)
    mask = generate_star_mask(lum, growth_factor=cfg.get("mask_growth", 2.5), softness=cfg.get("mask_softness", 4.0),
                              luminance_protect=cfg.get("luminance_protect", True),
                              luminance_ceiling=cfg.get("luminance_ceiling", 0.85))
    return tuple(masked_stretch(x, mask=mask, **cfg) for x in (r, g, b)) + (mask,)


class _RgbConfig(C.Structure):  # orc_rgb_config
    _fields_ = [("white_balance", C.c_int32), ("wb_manual", C.c_double * 3), ("auto_stretch", C.c_int32),
                ("linked_stf", C.c_int32), ("has_stf", C.c_int32 * 3), ("stf", _Stf * 3), ("align", C.c_int32),
                ("align_method", C.c_int32), ("has_scnr", C.c_int32), ("scnr_method", C.c_int32),
                ("scnr_amount", C.c_float), ("scnr_preserve", C.c_int32), ("num_threads", C.c_int32)]


class _RgbResult(C.Structure):  # orc_rgb_result
    _fields_ = [("rows", C.c_uint64), ("cols", C.c_uint64), ("stf", _Stf * 3), ("chan_stats", (C.c_double * 4) * 3),
                ("offset_g", C.c_double * 2), ("offset_b", C.c_double * 2), ("scnr_applied", C.c_int32),
                ("resampled", C.c_int32), ("stats_wb", _Stats * 3)]


@dataclass
class ProcessedRgb:  # rgb.rs:18-40
    r: np.ndarray
    g: np.ndarray
    b: np.ndarray
    rows: int
    cols: int
    stf: tuple
    channel_stats: tuple
    offset_g: tuple
    offset_b: tuple
    scnr_applied: bool
    resampled: bool
    pre_stretch: tuple
    stats_wb: tuple


def resample_image(image, target_rows: int, target_cols: int) -> np.ndarray:
    """resample.rs:25-61"""
    L = lib()
    L.orc_resample_image.restype = C.c_int
    L.orc_resample_image.argtypes = [C.POINTER(C.c_float), C.c_size_t, C.c_size_t, C.c_size_t, C.c_size_t, C.POINTER(C.c_float)]
    im = _f32(image)
    out = np.zeros((target_rows, target_cols), np.float32)
    if L.orc_resample_image(_fp(im), im.shape[0], im.shape[1], target_rows, target_cols, _fp(out)) != 0:
        raise ValueError("Target dimensions must be > 0")
    return out


def select_wb_reference(sr: ImageStats, sg: ImageStats, sb: ImageStats):
    """white_balance.rs:3-20"""
    L = lib()
    L.orc_select_wb_reference.argtypes = [C.POINTER(_Stats)] * 3 + [C.POINTER(C.c_double)]
    out = (C.c_double * 3)()
    L.orc_select_wb_reference(*[C.byref(_stats_in(s)) for s in (sr, sg, sb)], out)
    return tuple(out)


def compose_apply_stf(image, params: StfParams, stats: ImageStats) -> np.ndarray:
    """the compose-local apply_stf_inplace (rgb.rs:191-207)"""
    L = lib()
    L.orc_compose_apply_stf_inplace.argtypes = [C.POINTER(C.c_float), C.c_size_t, C.POINTER(_Stf), C.POINTER(_Stats)]
    out = _f32(image).copy()
    p = _Stf(params.shadow, params.midtone, params.highlight)
    L.orc_compose_apply_stf_inplace(_fp(out), out.size, C.byref(p), C.byref(_stats_in(stats)))
    return out


def process_rgb(r, g, b, white_balance="auto", auto_stretch=True, stf=(None, None, None), linked_stf=False, align=True,
                align_method="phase_correlation", scnr=None, num_threads=8) -> ProcessedRgb:
    """process_rgb (rgb.rs:209-323); raises ValueError with the reference's message on its Err paths."""
    L = _pc_protos()
    _det_protos()
    fp = C.POINTER(C.c_float)
    L.orc_process_rgb.restype = C.c_int
    L.orc_process_rgb.argtypes = [fp, C.c_size_t, C.c_size_t] * 3 + [C.POINTER(_RgbConfig)] + [fp] * 6 + [
        C.POINTER(_RgbResult), C.c_char_p, C.c_size_t]
    chans = [None if x is None else _f32(x) for x in (r, g, b)]
    present = [x for x in chans if x is not None]
    rows = max((x.shape[0] for x in present), default=0)
    cols = max((x.shape[1] for x in present), default=0)
    cfg = _RgbConfig()
    if isinstance(white_balance, str):
        cfg.white_balance = {"auto": 0, "none": 2}[white_balance]
    else:
        cfg.white_balance = 1
        cfg.wb_manual[:] = [float(v) for v in white_balance]
    cfg.auto_stretch, cfg.linked_stf = int(bool(auto_stretch)), int(bool(linked_stf))
    for c, p in enumerate(stf):
        if p is not None:
            cfg.has_stf[c] = 1
            cfg.stf[c] = _Stf(p.shadow, p.midtone, p.highlight)
    cfg.align = int(bool(align))
    cfg.align_method = {"phase_correlation": 0, "affine": 1}[align_method]
    if scnr is not None:
        cfg.has_scnr = 1
        cfg.scnr_method = 0 if scnr.get("method", "average") in ("average", "AverageNeutral", 0) else 1
        cfg.scnr_amount = scnr.get("amount", 1.0)
        cfg.scnr_preserve = int(bool(scnr.get("preserve_luminance", False)))
    cfg.num_threads = num_threads
    outs = [np.zeros((rows, cols), np.float32) for _ in range(6)]
    res = _RgbResult()
    err = C.create_string_buffer(512)
    args = []
    for x in chans:
        args += [None, 0, 0] if x is None else [_fp(x), x.shape[0], x.shape[1]]
    rc = L.orc_process_rgb(*args, C.byref(cfg), *[_fp(o) for o in outs], C.byref(res), err, 512)
    if rc != 0:
        raise ValueError(err.value.decode())
    return ProcessedRgb(outs[0], outs[1], outs[2], int(res.rows), int(res.cols),
                        tuple(StfParams(p.shadow, p.midtone, p.highlight) for p in res.stf),
                        tuple(tuple(row) for row in res.chan_stats), tuple(res.offset_g), tuple(res.offset_b),
                        bool(res.scnr_applied), bool(res.resampled), tuple(outs[3:]),
                        tuple(_stats_out(s) for s in res.stats_wb))


class _SpccConfig(C.Structure):  # orc_spcc_config
    _fields_ = [("min_snr", C.c_double), ("max_stars", C.c_uint64), ("saturation_limit", C.c_double),
                ("white_reference", C.c_int32), ("custom", C.c_double * 3)]


class _SpccResult(C.Structure):  # orc_spcc_result
    _fields_ = [("r_factor", C.c_double), ("g_factor", C.c_double), ("b_factor", C.c_double),
                ("stars_matched", C.c_uint64), ("stars_total", C.c_uint64), ("avg_color_index", C.c_double)]


@dataclass
class SpccResult:  # spcc.rs:45-56
    r_factor: float
    g_factor: float
    b_factor: float
    stars_matched: int
    stars_total: int
    avg_color_index: float


_WHITE_REFS = {"average_spiral": 0, "g2v": 1, "photopic": 2}


def _spcc_cfg(min_snr, max_stars, saturation_limit, white_reference) -> _SpccConfig:
    cfg = _SpccConfig(min_snr, max_stars, saturation_limit, 0, (C.c_double * 3)(0, 0, 0))
    if isinstance(white_reference, str):
        cfg.white_reference = _WHITE_REFS[white_reference]
    else:
        cfg.white_reference = 3
        cfg.custom[:] = [float(v) for v in white_reference]
    return cfg


def spcc_white_reference_rgb(white_reference="average_spiral"):
    """white_reference_rgb (spcc.rs:245-255)"""
    cfg = _spcc_cfg(20.0, 200, 0.9, white_reference)
    out = (C.c_double * 3)()
    L = lib()
    L.orc_spcc_white_reference_rgb.argtypes = [C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double)]
    L.orc_spcc_white_reference_rgb(cfg.white_reference, cfg.custom, out)
    return tuple(out)


def aperture_flux_f32(image, x: float, y: float, radius: float) -> float:
    """aperture_flux_f32 (spcc.rs:341-383)"""
    L = lib()
    L.orc_aperture_flux_f32.restype = C.c_double
    L.orc_aperture_flux_f32.argtypes = [C.POINTER(C.c_float), C.c_size_t, C.c_size_t, C.c_double, C.c_double, C.c_double]
    im = _f32(image)
    return float(L.orc_aperture_flux_f32(_fp(im), im.shape[0], im.shape[1], x, y, radius))


def spcc_calibrate_rgb(r, g, b, pixel_scale_arcsec: float, min_snr=20.0, max_stars=200, saturation_limit=0.90,
                       white_reference="average_spiral", detection=None) -> SpccResult:
    """spcc_calibrate_rgb (spcc.rs:73-183); detection=(stars, lum_max) skips detect_stars (:90-183 only).
    Raises ValueError with the reference's message on its Err paths."""
    L = _det_protos()
    fp = C.POINTER(C.c_float)
    L.orc_spcc_calibrate_rgb.restype = C.c_int
    L.orc_spcc_calibrate_rgb.argtypes = [fp, fp, fp, C.c_size_t, C.c_size_t, C.c_double, C.POINTER(_SpccConfig),
                                         C.POINTER(_SpccResult)]
    L.orc_spcc_from_detection.restype = C.c_int
    L.orc_spcc_from_detection.argtypes = [fp, fp, fp, C.c_size_t, C.c_size_t, C.POINTER(_Star), C.c_size_t, C.c_double,
                                          C.c_double, C.POINTER(_SpccConfig), C.POINTER(_SpccResult)]
    rr, gg, bb = _f32(r), _f32(g), _f32(b)
    cfg = _spcc_cfg(min_snr, max_stars, saturation_limit, white_reference)
    res = _SpccResult()
    if detection is None:
        rc = L.orc_spcc_calibrate_rgb(_fp(rr), _fp(gg), _fp(bb), rr.shape[0], rr.shape[1], pixel_scale_arcsec, C.byref(cfg),
                                      C.byref(res))
    else:
        stars, lum_max = detection
        buf = (_Star * max(len(stars), 1))()
        for i, (d, s) in enumerate(zip(buf, stars)):
            d.x, d.y, d.flux, d.fwhm, d.eccentricity, d.peak, d.snr, d.npix, d.order = (
                s.x, s.y, s.flux, s.fwhm, s.eccentricity, s.peak, s.snr, s.npix, i)
        rc = L.orc_spcc_from_detection(_fp(rr), _fp(gg), _fp(bb), rr.shape[0], rr.shape[1], buf, len(stars), lum_max,
                                       pixel_scale_arcsec, C.byref(cfg), C.byref(res))
    if rc == 1:
        raise ValueError(f"Only {res.stars_total} stars passed quality filters (need 5+). Try lowering min_snr.")
    if rc == 2:
        raise ValueError(f"Only {res.stars_matched} stars cross-matched (need 3+). Check WCS solution quality.")
    return SpccResult(res.r_factor, res.g_factor, res.b_factor, int(res.stars_matched), int(res.stars_total),
                      res.avg_color_index)


def apply_lrgb(l, r, g, b, lightness_weight=1.0, chrominance_weight=1.0):
    """apply_lrgb (lrgb.rs:4-45) -> new (r, g, b); ValueError on mismatched dims."""
    ll, rr, gg, bb = _f32(l), _f32(r).copy(), _f32(g).copy(), _f32(b).copy()
    if not (rr.shape == gg.shape == bb.shape == ll.shape):
        raise ValueError(f"L dimensions {ll.shape} do not match RGB (R: {rr.shape}, G: {gg.shape}, B: {bb.shape})")
    L = lib()
    fp = C.POINTER(C.c_float)
    L.orc_apply_lrgb.argtypes = [fp, fp, fp, fp, C.c_size_t, C.c_float, C.c_float]
    L.orc_apply_lrgb(_fp(ll), _fp(rr), _fp(gg), _fp(bb), ll.size, lightness_weight, chrominance_weight)
    return rr, gg, bb


def synthesize_luminance(r, g, b) -> np.ndarray:
    """lrgb.rs:47-64"""
    rr, gg, bb = _f32(r), _f32(g), _f32(b)
    out = np.zeros_like(rr)
    L = lib()
    fp = C.POINTER(C.c_float)
    L.orc_synthesize_luminance.argtypes = [fp, fp, fp, C.c_size_t, fp]
    L.orc_synthesize_luminance(_fp(rr), _fp(gg), _fp(bb), rr.size, _fp(out))
    return out


def compute_linked_stf(sr: ImageStats, sg: ImageStats, sb: ImageStats, target_bg=0.25, shadow_k=-2.8):
    """cmd/helpers.rs:185-202 -> (StfParams, combined ImageStats)"""
    L = lib()
    L.orc_compute_linked_stf.argtypes = [C.POINTER(_Stats)] * 3 + [C.c_double, C.c_double, C.POINTER(_Stf), C.POINTER(_Stats)]
    p, c = _Stf(), _Stats()
    L.orc_compute_linked_stf(*[C.byref(_stats_in(s)) for s in (sr, sg, sb)], target_bg, shadow_k, C.byref(p), C.byref(c))
    return StfParams(p.shadow, p.midtone, p.highlight), _stats_out(c)


def calibrate_channel(orig, factor: float, orig_stats: ImageStats):
    """cmd/compose/color.rs:21-49 -> (scaled plane, ImageStats)"""
    im = _f32(orig)
    out = np.zeros_like(im)
    st = _Stats()
    L = lib()
    fp = C.POINTER(C.c_float)
    L.orc_calibrate_channel.argtypes = [fp, C.c_size_t, C.c_float, C.POINTER(_Stats), fp, C.POINTER(_Stats)]
    L.orc_calibrate_channel(_fp(im), im.size, factor, C.byref(_stats_in(orig_stats)), _fp(out), C.byref(st))
    return out, _stats_out(st)


def create_master(kind: str, frames, master_bias=None, master_dark=None) -> np.ndarray:
    """create_master_bias / _dark / _flat on in-memory frames (calibration.rs:127-255)"""
    if len(frames) == 0:
        raise ValueError(f"No {kind} frames provided")
    fr = [_f32(f) for f in frames]
    for f in fr[1:]:
        if f.shape != fr[0].shape:
            raise ValueError(f"Dimension mismatch: expected {fr[0].shape}, got {f.shape}")
    fp = C.POINTER(C.c_float)
    ptrs = (fp * len(fr))(*[_fp(f) for f in fr])
    out = np.zeros_like(fr[0])
    mb = None if master_bias is None else _f32(master_bias)
    md = None if master_dark is None else _f32(master_dark)
    L = lib()
    L.orc_create_master.argtypes = [C.c_int, C.POINTER(fp), C.c_size_t, C.c_size_t, fp, fp, fp]
    L.orc_create_master({"bias": 0, "dark": 1, "flat": 2}[kind], ptrs, len(fr), fr[0].size,
                        None if mb is None else _fp(mb), None if md is None else _fp(md), _fp(out))
    return out


def fits_decode_pixels(data, bitpix: int, bscale=1.0, bzero=0.0) -> np.ndarray:
    """decode_pixels (infra/fits/reader.rs:42-101) -> flat f32 array (empty for an unknown BITPIX)"""
    buf = np.frombuffer(data, dtype=np.uint8) if isinstance(data, (bytes, bytearray, memoryview)) else np.ascontiguousarray(data, np.uint8)
    bpp = {8: 1, 16: 2, 32: 4, -32: 4, -64: 8}.get(bitpix, 0)
    out = np.zeros(buf.size // bpp if bpp else 0, np.float32)
    L = lib()
    L.orc_fits_decode_pixels.restype = C.c_size_t
    L.orc_fits_decode_pixels.argtypes = [C.c_void_p, C.c_size_t, C.c_int64, C.c_double, C.c_double, C.POINTER(C.c_float)]
    n = L.orc_fits_decode_pixels(C.c_void_p(buf.ctypes.data), buf.size, bitpix, bscale, bzero, _fp(out) if out.size else None)
    return out[:n]


def fits_compute_bzero_bscale(image):
    """compute_bzero_bscale (writer.rs:143-159) -> (bzero, bscale)"""
    im = _f32(image)
    bz, bs = C.c_double(), C.c_double()
    L = lib()
    L.orc_fits_compute_bzero_bscale.argtypes = [C.POINTER(C.c_float), C.c_size_t, C.POINTER(C.c_double), C.POINTER(C.c_double)]
    L.orc_fits_compute_bzero_bscale(_fp(im), im.size, C.byref(bz), C.byref(bs))
    return bz.value, bs.value


def fits_encode_pixels(image, bitpix: int, bzero=0.0, bscale=1.0) -> np.ndarray:
    """write_f32 / i16 / f64_slice_as_be (writer.rs:82-135) -> uint8 data unit"""
    im = _f32(image)
    out = np.zeros(im.size * {-32: 4, 16: 2, -64: 8}[bitpix], np.uint8)
    L = lib()
    L.orc_fits_encode_pixels.restype = C.c_size_t
    L.orc_fits_encode_pixels.argtypes = [C.POINTER(C.c_float), C.c_size_t, C.c_int32, C.c_double, C.c_double, C.c_void_p]
    L.orc_fits_encode_pixels(_fp(im), im.size, bitpix, bzero, bscale, C.c_void_p(out.ctypes.data))
    return out


def _f32(a) -> np.ndarray:
    return np.ascontiguousarray(a, dtype=np.float32)


def _fp(a: np.ndarray):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def max_threads() -> int:
    return int(lib().orc_max_threads())


# ---- math/median.rs -------------------------------------------------------
def f32_cmp(a: float, b: float) -> int:
    return int(lib().orc_f32_cmp(a, b))


def select_nth(values, k: int) -> np.ndarray:
    v = _f32(values).copy()
    lib().orc_select_nth_f32(_fp(v), v.size, k)
    return v


def exact_median_mut(values) -> float:
    v = _f32(values).copy()
    return float(lib().orc_exact_median_mut(_fp(v), v.size))


def median_f32_mut(values) -> float:
    v = _f32(values).copy()
    return float(lib().orc_median_f32_mut(_fp(v), v.size))


def exact_mad_mut(values, median: float) -> float:
    v = _f32(values).copy()
    return float(lib().orc_exact_mad_mut(_fp(v), v.size, median))


def sigma_clipped_stats(values, kappa: float, iterations: int):
    v = _f32(values).copy()
    if v.size == 0:
        v = np.zeros(1, np.float32)
        n = C.c_size_t(0)
    else:
        n = C.c_size_t(v.size)
    med, sig = C.c_double(), C.c_double()
    lib().orc_sigma_clipped_stats(_fp(v), C.byref(n), kappa, iterations, C.byref(med), C.byref(sig))
    return med.value, sig.value


# ---- core/stacking/combine.rs --------------------------------------------
def sigma_clip_combine(values, sigma_low=3.0, sigma_high=3.0, max_iter=5, order=ORDER_ASCENDING):
    v = _f32(values).copy()
    n = v.size
    if n == 0:
        v = np.zeros(1, np.float32)
    rej = C.c_uint32(0)
    r = lib().orc_sigma_clip_combine(_fp(v), n, sigma_low, sigma_high, max_iter, order, C.byref(rej))
    return np.float32(r), int(rej.value)


def stack_images(images, sigma_low=3.0, sigma_high=3.0, max_iterations=5, order=ORDER_ASCENDING, threads=0):
    """combine.rs:94-193 with align=false.  Returns (image, rejected_pixels)."""
    imgs = [_f32(im) for im in images]
    n = len(imgs)
    if n == 0:
        raise ValueError("No images to stack")
    rows = (C.c_int64 * n)(*[im.shape[0] for im in imgs])
    cols = (C.c_int64 * n)(*[im.shape[1] for im in imgs])
    ptrs = (C.POINTER(C.c_float) * n)(*[_fp(im) for im in imgs])
    r = min(im.shape[0] for im in imgs)
    c = min(im.shape[1] for im in imgs)
    out = np.zeros((r, c), np.float32)
    rej = C.c_uint64(0)
    orow, ocol = C.c_int64(0), C.c_int64(0)
    rc = lib().orc_stack_images_noalign(ptrs, rows, cols, n, sigma_low, sigma_high, max_iterations, order,
                                        threads, _fp(out), C.byref(rej), C.byref(orow), C.byref(ocol))
    if rc != 0:
        raise ValueError("No images to stack")
    return out, int(rej.value)


def stack_partial(images, sigma_low=3.0, sigma_high=3.0, max_iterations=5, threads=0):
    imgs = [_f32(im) for im in images]
    n = len(imgs)
    npix = imgs[0].size
    ptrs = (C.POINTER(C.c_float) * n)(*[_fp(im) for im in imgs])
    s = np.zeros(imgs[0].shape, np.float64)
    cnt = np.zeros(imgs[0].shape, np.uint32)
    rej = C.c_uint64(0)
    lib().orc_stack_partial_noalign(ptrs, n, npix, sigma_low, sigma_high, max_iterations, threads,
                                    s.ctypes.data_as(C.POINTER(C.c_double)),
                                    cnt.ctypes.data_as(C.POINTER(C.c_uint32)), C.byref(rej))
    return s, cnt, int(rej.value)


# ---- sampling / warp --------------------------------------------------------
def catmull_rom(t: float) -> float:
    return float(lib().orc_catmull_rom(t))


def clamp_index(idx: int, length: int) -> int:
    return int(lib().orc_clamp_index(idx, length))


def _sample(fn, data, rows, cols, y, x):
    d = _f32(data).ravel()
    if d.size == 0:
        return float(fn(None, rows, cols, y, x))
    return float(fn(_fp(d), rows, cols, y, x))


def nearest_sample(data, rows, cols, y, x):
    return _sample(lib().orc_nearest_sample, data, rows, cols, y, x)


def bilinear_sample(data, rows, cols, y, x):
    return _sample(lib().orc_bilinear_sample, data, rows, cols, y, x)


def bicubic_sample(data, rows, cols, y, x):
    return _sample(lib().orc_bicubic_sample, data, rows, cols, y, x)


def shift_image_subpixel(image, dy: float, dx: float, threads=0) -> np.ndarray:
    im = _f32(image)
    out = np.zeros_like(im)
    lib().orc_shift_image_subpixel(_fp(im), im.shape[0], im.shape[1], dy, dx, threads, _fp(out))
    return out


def warp_image(image, transform, out_rows: int, out_cols: int, threads=0) -> np.ndarray:
    """transform = (a, b, tx, c, d, ty), maps output (x, y) -> source (sx, sy)."""
    im = _f32(image)
    t = (C.c_double * 6)(*[float(v) for v in transform])
    out = np.zeros((out_rows, out_cols), np.float32)
    lib().orc_warp_image(_fp(im), im.shape[0], im.shape[1], t, out_rows, out_cols, threads, _fp(out))
    return out


# ---- stats / stf --------------------------------------------------------------
def _stats_out(s: _Stats) -> ImageStats:
    return ImageStats(s.min, s.max, s.median, s.mad, s.sigma, s.mean, int(s.valid_count))


def _stats_in(st: ImageStats) -> _Stats:
    return _Stats(st.min, st.max, st.median, st.mad, st.sigma, st.mean, st.valid_count)


def compute_image_stats(image, path: str = "auto") -> ImageStats:
    im = _f32(image).ravel()
    s = _Stats()
    fn = {"auto": lib().orc_compute_image_stats, "exact": lib().orc_compute_image_stats_exact,
          "hist": lib().orc_compute_image_stats_hist}[path]
    fn(_fp(im), im.size, C.byref(s))
    return _stats_out(s)


def compute_image_stats_with_known_range(image, known_min, known_max) -> ImageStats:
    im = _f32(image).ravel()
    s = _Stats()
    lib().orc_compute_image_stats_with_known_range(_fp(im), im.size, known_min, known_max, C.byref(s))
    return _stats_out(s)


def build_histogram(image, bins: int, dmin: float, dmax: float) -> np.ndarray:
    im = _f32(image).ravel()
    out = np.zeros(bins, np.uint32)
    lib().orc_build_histogram(_fp(im), im.size, bins, dmin, dmax, out.ctypes.data_as(C.POINTER(C.c_uint32)))
    return out


def stats_value_hist(image, gmin: float, gmax: float):
    im = _f32(image).ravel()
    h = np.zeros(65536, np.uint64)
    s, c = C.c_double(), C.c_uint64()
    lib().orc_stats_value_hist(_fp(im), im.size, gmin, gmax, h.ctypes.data_as(C.POINTER(C.c_uint64)),
                               C.byref(s), C.byref(c))
    return h, s.value, int(c.value)


def auto_stf(stats: ImageStats, target_bg=0.25, shadow_k=-2.8) -> StfParams:
    s = _stats_in(stats)
    p = _Stf()
    lib().orc_auto_stf(C.byref(s), target_bg, shadow_k, C.byref(p))
    return StfParams(p.shadow, p.midtone, p.highlight)


def mtf(x: float, m: float) -> float:
    return float(lib().orc_mtf(x, m))


def mtf_balance(m: float, t: float) -> float:
    return float(lib().orc_mtf_balance(m, t))


def apply_stf(image, params: StfParams, stats: ImageStats, threads=0) -> np.ndarray:
    im = _f32(image)
    out = np.zeros(im.shape, np.uint8)
    p = _Stf(params.shadow, params.midtone, params.highlight)
    s = _stats_in(stats)
    lib().orc_apply_stf_u8(_fp(im), im.size, C.byref(p), C.byref(s), threads,
                           out.ctypes.data_as(C.POINTER(C.c_uint8)))
    return out


def apply_stf_f32(image, params: StfParams, stats: ImageStats, threads=0) -> np.ndarray:
    im = _f32(image)
    out = np.zeros(im.shape, np.float32)
    p = _Stf(params.shadow, params.midtone, params.highlight)
    s = _stats_in(stats)
    lib().orc_apply_stf_f32(_fp(im), im.size, C.byref(p), C.byref(s), threads, _fp(out))
    return out


# ---- subframe scoring (orc_subframe.c, core/analysis/subframe.rs) ------------------------------------------
class _SubCfg(C.Structure):
    _fields_ = [("fwhm_weight", C.c_double), ("eccentricity_weight", C.c_double), ("snr_weight", C.c_double),
                ("noise_weight", C.c_double), ("max_fwhm", C.c_double), ("max_eccentricity", C.c_double),
                ("min_snr", C.c_double), ("min_stars", C.c_uint64)]


class _SubMetrics(C.Structure):
    _fields_ = [("star_count", C.c_uint64), ("median_fwhm", C.c_double), ("median_eccentricity", C.c_double),
                ("median_snr", C.c_double), ("background_median", C.c_double), ("background_sigma", C.c_double),
                ("noise_ratio", C.c_double), ("weight", C.c_double), ("accepted", C.c_int32)]

    def as_dict(self):
        d = {k: getattr(self, k) for k, _ in self._fields_}
        d["star_count"], d["accepted"] = int(d["star_count"]), bool(d["accepted"])
        return d


SUBFRAME_DEFAULTS = dict(fwhm_weight=1.0, eccentricity_weight=0.5, snr_weight=1.0, noise_weight=0.3, max_fwhm=8.0,
                         max_eccentricity=0.7, min_snr=5.0, min_stars=5)                    # subframe.rs:36-49


def _sub_cfg(**kw):
    d = dict(SUBFRAME_DEFAULTS)
    d.update(kw)
    return _SubCfg(*[d[k] for k, _ in _SubCfg._fields_])


def subframe_compute_weight(fwhm, ecc, snr, noise, **cfg) -> float:
    L = lib()
    L.orc_subframe_compute_weight.restype = C.c_double
    L.orc_subframe_compute_weight.argtypes = [C.c_double] * 4 + [C.POINTER(_SubCfg)]
    return L.orc_subframe_compute_weight(fwhm, ecc, snr, noise, C.byref(_sub_cfg(**cfg)))


def analyze_subframe(image, **cfg) -> dict:
    im = _f32(image)
    L = lib()
    L.orc_analyze_subframe.argtypes = [C.POINTER(C.c_float), C.c_size_t, C.c_size_t, C.POINTER(_SubCfg), C.POINTER(_SubMetrics)]
    out = _SubMetrics()
    L.orc_analyze_subframe(_fp(im), im.shape[0], im.shape[1], C.byref(_sub_cfg(**cfg)), C.byref(out))
    return out.as_dict()


def subframe_from_detection(stars, bg_median, bg_sigma, **cfg) -> dict:
    """the metrics from a detect_stars() result; stars = [(fwhm, eccentricity, snr)]"""
    L = lib()
    L.orc_subframe_from_detection.argtypes = [C.POINTER(_Star), C.c_size_t, C.c_double, C.c_double, C.POINTER(_SubCfg),
                                              C.POINTER(_SubMetrics)]
    buf = (_Star * max(len(stars), 1))()
    for b, (f, e, s) in zip(buf, stars):
        b.fwhm, b.eccentricity, b.snr = f, e, s
    out = _SubMetrics()
    L.orc_subframe_from_detection(buf, len(stars), bg_median, bg_sigma, C.byref(_sub_cfg(**cfg)), C.byref(out))
    return out.as_dict()


def subframe_normalize_weights(weights):
    L = lib()
    L.orc_subframe_normalize_weights.argtypes = [C.POINTER(_SubMetrics), C.c_size_t]
    buf = (_SubMetrics * max(len(weights), 1))()
    for b, w in zip(buf, weights):
        b.weight = w
    L.orc_subframe_normalize_weights(buf, len(weights))
    return [b.weight for b in buf[:len(weights)]]


# ---- preview / tile renderers up to the PNG encoder (orc_render.c) -----------------------------------------
class _TileLevel(C.Structure):
    _fields_ = [("level", C.c_uint64), ("width", C.c_uint64), ("height", C.c_uint64), ("cols", C.c_uint64), ("rows", C.c_uint64),
                ("scale_factor", C.c_double), ("offset", C.c_uint64)]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


def _stf3(stf, stats):
    if stf is None:
        return None, None
    return ((_Stf * 3)(*[_Stf(p.shadow, p.midtone, p.highlight) for p in stf]), (_Stats * 3)(*[_stats_in(s) for s in stats]))


def preview_dims(rows, cols, max_dim):
    ph, pw = C.c_size_t(), C.c_size_t()
    L = lib()
    L.orc_preview_dims.argtypes = [C.c_size_t] * 3 + [C.POINTER(C.c_size_t)] * 2
    L.orc_preview_dims(rows, cols, max_dim, C.byref(ph), C.byref(pw))
    return ph.value, pw.value


def render_rgb_preview(r, g, b, max_dim, stf=None, stats=None) -> np.ndarray:
    r, g, b = _f32(r), _f32(g), _f32(b)
    ph, pw = preview_dims(r.shape[0], r.shape[1], max_dim)
    out = np.zeros((ph, pw, 3), np.uint8)
    p, s = _stf3(stf, stats)
    L = lib()
    L.orc_render_rgb_preview.argtypes = [C.POINTER(C.c_float)] * 3 + [C.c_size_t] * 3 + [C.c_void_p, C.c_void_p, C.c_void_p]
    L.orc_render_rgb_preview(_fp(r), _fp(g), _fp(b), r.shape[0], r.shape[1], max_dim, C.cast(p, C.c_void_p) if p else None,
                             C.cast(s, C.c_void_p) if s else None, out.ctypes.data)
    return out


def ipc_encode_with_header(arr, max_dim=0) -> bytes:
    a = _f32(arr)
    out = np.zeros(16 + 4 * a.size, np.uint8)
    L = lib()
    L.orc_ipc_encode_with_header.restype = C.c_size_t
    L.orc_ipc_encode_with_header.argtypes = [C.POINTER(C.c_float), C.c_size_t, C.c_size_t, C.c_size_t, C.c_void_p]
    n = L.orc_ipc_encode_with_header(_fp(a), a.shape[0], a.shape[1], max_dim, out.ctypes.data)
    return out[:n].tobytes()


def tile_compute_num_levels(width, height, tile_size) -> int:
    L = lib()
    L.orc_tile_compute_num_levels.restype = C.c_size_t
    L.orc_tile_compute_num_levels.argtypes = [C.c_size_t] * 3
    return L.orc_tile_compute_num_levels(width, height, tile_size)


def tile_downsample_2x(arr) -> np.ndarray:
    a = _f32(arr)
    out = np.zeros(((a.shape[0] + 1) // 2, (a.shape[1] + 1) // 2), np.float32)
    L = lib()
    L.orc_tile_downsample_2x.argtypes = [C.POINTER(C.c_float), C.c_size_t, C.c_size_t, C.POINTER(C.c_float)]
    L.orc_tile_downsample_2x(_fp(a), a.shape[0], a.shape[1], _fp(out))
    return out


def tile_percentile_bounds(arr, low_pct=0.001, high_pct=0.999):
    a = _f32(arr)
    lo, hi = C.c_float(), C.c_float()
    L = lib()
    L.orc_tile_percentile_bounds.argtypes = [C.POINTER(C.c_float), C.c_size_t, C.c_double, C.c_double, C.POINTER(C.c_float), C.POINTER(C.c_float)]
    L.orc_tile_percentile_bounds(_fp(a), a.size, low_pct, high_pct, C.byref(lo), C.byref(hi))
    return lo.value, hi.value


def tile_pyramid_layout(rows, cols, tile_size, channels):
    lv, n = (_TileLevel * 64)(), C.c_size_t()
    L = lib()
    L.orc_tile_pyramid_layout.restype = C.c_size_t
    L.orc_tile_pyramid_layout.argtypes = [C.c_size_t] * 4 + [C.POINTER(_TileLevel), C.POINTER(C.c_size_t)]
    total = L.orc_tile_pyramid_layout(rows, cols, tile_size, channels, lv, C.byref(n))
    return [l.as_dict() for l in lv[:n.value]], total


def generate_tile_pyramid(normalized, tile_size):
    """-> (packed tile bytes, levels, (global_min, global_max))"""
    a = _f32(normalized)
    levels, total = tile_pyramid_layout(a.shape[0], a.shape[1], tile_size, 1)
    tiles = np.zeros(total, np.uint8)
    lv, n, lo, hi = (_TileLevel * 64)(), C.c_size_t(), C.c_float(), C.c_float()
    L = lib()
    L.orc_generate_tile_pyramid.argtypes = [C.POINTER(C.c_float), C.c_size_t, C.c_size_t, C.c_size_t, C.c_void_p, C.POINTER(_TileLevel),
                                            C.POINTER(C.c_size_t), C.POINTER(C.c_float), C.POINTER(C.c_float)]
    L.orc_generate_tile_pyramid(_fp(a), a.shape[0], a.shape[1], tile_size, tiles.ctypes.data, lv, C.byref(n), C.byref(lo), C.byref(hi))
    return tiles, levels, (lo.value, hi.value)


def generate_tile_pyramid_rgb(r, g, b, tile_size, stf=None, stats=None):
    r, g, b = _f32(r), _f32(g), _f32(b)
    levels, total = tile_pyramid_layout(r.shape[0], r.shape[1], tile_size, 3)
    tiles = np.zeros(total, np.uint8)
    lv, n = (_TileLevel * 64)(), C.c_size_t()
    p, s = _stf3(stf, stats)
    L = lib()
    L.orc_generate_tile_pyramid_rgb.argtypes = [C.POINTER(C.c_float)] * 3 + [C.c_size_t] * 3 + [C.c_void_p] * 3 + [C.POINTER(_TileLevel),
                                                                                                             C.POINTER(C.c_size_t)]
    L.orc_generate_tile_pyramid_rgb(_fp(r), _fp(g), _fp(b), r.shape[0], r.shape[1], tile_size, C.cast(p, C.c_void_p) if p else None,
                                    C.cast(s, C.c_void_p) if s else None, tiles.ctypes.data, lv, C.byref(n))
    return tiles, levels


# ---- batch calibration pipeline (orc_batch.c, core/imaging/calibration_pipeline.rs) -------------------------
def _opt(a):
    if a is None:
        return None, 0, None
    x = _f32(a)
    return _fp(x), x.size, x


def calibrate_light(light, bias=None, dark=None, flat=None) -> np.ndarray:
    im = _f32(light)
    out = np.zeros_like(im)
    (bp, bl, _b), (dp, dl, _d), (fp_, fl, _f) = _opt(bias), _opt(dark), _opt(flat)
    L = lib()
    F = C.POINTER(C.c_float)
    L.orc_calibrate_light.argtypes = [F, C.c_size_t, F, C.c_size_t, F, C.c_size_t, F, C.c_size_t, F]
    L.orc_calibrate_light(_fp(im), im.size, bp, bl, dp, dl, fp_, fl, _fp(out))
    return out


def normalize_frames(frames):
    out = []
    L = lib()
    L.orc_normalize_frame.argtypes = [C.POINTER(C.c_float), C.c_size_t, C.POINTER(C.c_float)]
    for fr in frames:
        im = _f32(fr)
        o = np.zeros_like(im)
        L.orc_normalize_frame(_fp(im), im.size, _fp(o))
        out.append(o)
    return out


def _frame_ptrs(frames):
    ims = [_f32(f) for f in frames]
    return ims, (C.POINTER(C.c_float) * len(ims))(*[_fp(i) for i in ims])


def sigma_clipped_mean_stack(frames, sigma_low=2.5, sigma_high=3.0, max_iterations=5):
    """-> (stacked, rejection_counts)"""
    ims, ptrs = _frame_ptrs(frames)
    out = np.zeros_like(ims[0])
    rej = (C.c_uint64 * len(ims))()
    L = lib()
    L.orc_sigma_clipped_mean_stack.argtypes = [C.c_void_p, C.c_size_t, C.c_size_t, C.c_float, C.c_float, C.c_size_t, C.POINTER(C.c_float),
                                               C.POINTER(C.c_uint64)]
    L.orc_sigma_clipped_mean_stack(ptrs, len(ims), ims[0].size, sigma_low, sigma_high, max_iterations, _fp(out), rej)
    return out, list(rej)


def run_batch_channel(lights, bias=None, dark=None, flat=None, sigma_low=2.5, sigma_high=3.0, max_iterations=5, normalize=True):
    """-> (master, rejection_counts, mean, stddev)"""
    ims, ptrs = _frame_ptrs(lights)
    out = np.zeros_like(ims[0])
    rej = (C.c_uint64 * len(ims))()
    (bp, bl, _b), (dp, dl, _d), (fp_, fl, _f) = _opt(bias), _opt(dark), _opt(flat)
    mean, std = C.c_double(), C.c_double()
    L = lib()
    F = C.POINTER(C.c_float)
    L.orc_run_batch_channel.argtypes = [C.c_void_p, C.c_size_t, C.c_size_t, F, C.c_size_t, F, C.c_size_t, F, C.c_size_t, C.c_float, C.c_float,
                                        C.c_size_t, C.c_int, F, C.POINTER(C.c_uint64), C.POINTER(C.c_double), C.POINTER(C.c_double)]
    L.orc_run_batch_channel(ptrs, len(ims), ims[0].size, bp, bl, dp, dl, fp_, fl, sigma_low, sigma_high, max_iterations, int(normalize),
                            _fp(out), rej, C.byref(mean), C.byref(std))
    return out, list(rej), mean.value, std.value


def compose_rgb_from_masters(r, g, b, l=None) -> np.ndarray:
    r, g, b = _f32(r), _f32(g), _f32(b)
    h = min(r.shape[0], g.shape[0], b.shape[0])
    w = min(r.shape[1], g.shape[1], b.shape[1])
    out = np.zeros((h, w, 3), np.float32)
    (lp, _n, lk) = _opt(l)
    oh, ow = C.c_size_t(), C.c_size_t()
    L = lib()
    F = C.POINTER(C.c_float)
    L.orc_compose_rgb_from_masters.argtypes = [F, C.c_size_t, C.c_size_t] * 4 + [F, C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]
    L.orc_compose_rgb_from_masters(_fp(r), r.shape[0], r.shape[1], _fp(g), g.shape[0], g.shape[1], _fp(b), b.shape[0], b.shape[1], lp,
                                   lk.shape[0] if lk is not None else 0, lk.shape[1] if lk is not None else 0, _fp(out),
                                   C.byref(oh), C.byref(ow))
    assert (oh.value, ow.value) == (h, w)
    return out
