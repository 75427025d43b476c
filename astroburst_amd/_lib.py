"""ctypes binding of libastroburst_hip.so (the C ABI in include/astroburst_hip.h).

The library is the product; this module only loads it and declares prototypes.  There is NO
CPU fallback: if the shared object is missing or no gfx950 device is present, calls raise.
"""
from __future__ import annotations

import ctypes as C
import os
import re
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(_HERE)
LIB_PATH = os.environ.get("AB_LIB_PATH") or os.path.join(_HERE, "libastroburst_hip.so")  # AB_LIB_PATH: A/B-test a variant build
HEADER_PATH = os.path.join(ROOT, "include", "astroburst_hip.h")
CSRC = os.path.join(_HERE, "csrc")


class AstroBurstError(RuntimeError):
    def __init__(self, code: int, message: str):
        super().__init__(f"[ab_status {code}] {message}")
        self.code = code
        self.message = message


AB_OK, AB_ERR_INVALID, AB_ERR_HIP, AB_ERR_NO_DEVICE, AB_ERR_UNSUPPORTED, AB_ERR_NOMEM, AB_ERR_COMM, AB_ERR_CANCELLED = range(8)
AB_DT_I32, AB_DT_U32, AB_DT_I64, AB_DT_U64, AB_DT_F32, AB_DT_F64 = range(6)
AB_RED_SUM, AB_RED_MAX, AB_RED_MIN = range(3)
AB_COMM_ID_BYTES = 128
# void (*ab_progress_cb)(const char *stage, uint64_t current, uint64_t total, void *user)
PROGRESS_CB = C.CFUNCTYPE(None, C.c_char_p, C.c_uint64, C.c_uint64, C.c_void_p)


class Plane(C.Structure):
    _fields_ = [("data", C.c_void_p), ("rows", C.c_int64), ("cols", C.c_int64), ("on_device", C.c_int32)]


class StackConfig(C.Structure):  # types/stacking.rs:3-20
    _fields_ = [("sigma_low", C.c_float), ("sigma_high", C.c_float), ("max_iterations", C.c_uint32),
                ("align", C.c_int32)]


class ImageStatsC(C.Structure):  # types/image.rs:2-10
    _fields_ = [("min", C.c_double), ("max", C.c_double), ("median", C.c_double), ("mad", C.c_double),
                ("sigma", C.c_double), ("mean", C.c_double), ("valid_count", C.c_uint64)]


class StfParamsC(C.Structure):  # types/image.rs:36-40
    _fields_ = [("shadow", C.c_double), ("midtone", C.c_double), ("highlight", C.c_double)]


class AutoStfConfigC(C.Structure):  # types/image.rs:52-65
    _fields_ = [("target_bg", C.c_double), ("shadow_k", C.c_double)]


class DetectedStarC(C.Structure):  # star_detection.rs:10-20
    _fields_ = [("x", C.c_double), ("y", C.c_double), ("flux", C.c_double), ("fwhm", C.c_double),
                ("eccentricity", C.c_double), ("peak", C.c_double), ("snr", C.c_double), ("npix", C.c_uint64)]


class AffineAlignResultC(C.Structure):  # affine.rs:82-89
    _fields_ = [("transform", C.c_double * 6), ("matched_stars", C.c_uint64), ("inliers", C.c_uint64),
                ("residual_px", C.c_double), ("method", C.c_int32)]


class PhaseCorrelationResultC(C.Structure):  # phase_correlation.rs:15-20
    _fields_ = [("dx", C.c_double), ("dy", C.c_double), ("confidence", C.c_double)]


class ScnrConfigC(C.Structure):  # types/image.rs:82-100
    _fields_ = [("method", C.c_int32), ("amount", C.c_float), ("preserve_luminance", C.c_int32)]


class BlendWeightC(C.Structure):  # channel_blend.rs:5-11
    _fields_ = [("channel_idx", C.c_uint64), ("r_weight", C.c_double), ("g_weight", C.c_double),
                ("b_weight", C.c_double)]


class LevelsParamsC(C.Structure):  # curves.rs:4-9
    _fields_ = [("black", C.c_double), ("gamma", C.c_double), ("white", C.c_double)]


class BackgroundConfigC(C.Structure):  # background.rs:14-33
    _fields_ = [("grid_size", C.c_size_t), ("poly_degree", C.c_size_t), ("sigma_clip", C.c_float),
                ("iterations", C.c_size_t), ("mode", C.c_int)]


class BackgroundInfoC(C.Structure):  # scalars of BackgroundResult, background.rs:35-42
    _fields_ = [("sample_count", C.c_size_t), ("rms_residual", C.c_double), ("coeffs", C.c_double * 21)]


class StarMaskConfigC(C.Structure):  # star_mask.rs:6-30
    _fields_ = [("growth_factor", C.c_double), ("softness", C.c_double), ("detection_sigma", C.c_double),
                ("min_fwhm", C.c_double), ("max_fwhm", C.c_double), ("luminance_protect", C.c_int),
                ("luminance_ceiling", C.c_double)]


class StarMaskInfoC(C.Structure):  # star_mask.rs:32-37
    _fields_ = [("stars_masked", C.c_size_t), ("coverage_fraction", C.c_double)]


class MaskedStretchConfigC(C.Structure):  # masked_stretch.rs:7-32
    _fields_ = [("iterations", C.c_size_t), ("target_background", C.c_double), ("mask_growth", C.c_double),
                ("mask_softness", C.c_double), ("luminance_protect", C.c_int), ("luminance_ceiling", C.c_double),
                ("protection_amount", C.c_double), ("convergence_threshold", C.c_double)]


class MaskedStretchResultC(C.Structure):  # masked_stretch.rs:34-42
    _fields_ = [("iterations_run", C.c_size_t), ("final_background", C.c_double), ("stars_masked", C.c_size_t),
                ("mask_coverage", C.c_double), ("converged", C.c_int)]


class RgbComposeConfigC(C.Structure):  # types/compose.rs:47-75
    _fields_ = [("white_balance", C.c_int32), ("wb_manual", C.c_double * 3), ("auto_stretch", C.c_int32),
                ("linked_stf", C.c_int32), ("has_stf", C.c_int32 * 3), ("stf", StfParamsC * 3), ("align", C.c_int32),
                ("align_method", C.c_int32), ("has_scnr", C.c_int32), ("scnr", ScnrConfigC), ("num_threads", C.c_int32)]


class ProcessedRgbInfoC(C.Structure):  # rgb.rs:18-40
    _fields_ = [("rows", C.c_uint64), ("cols", C.c_uint64), ("stf", StfParamsC * 3), ("chan_stats", (C.c_double * 4) * 3),
                ("offset_g", C.c_double * 2), ("offset_b", C.c_double * 2), ("scnr_applied", C.c_int32),
                ("resampled", C.c_int32), ("stats_wb", ImageStatsC * 3)]


class SpccConfigC(C.Structure):  # spcc.rs:9-28
    _fields_ = [("min_snr", C.c_double), ("max_stars", C.c_uint64), ("saturation_limit", C.c_double),
                ("white_reference", C.c_int32), ("custom", C.c_double * 3)]


class SpccResultC(C.Structure):  # spcc.rs:45-56
    _fields_ = [("r_factor", C.c_double), ("g_factor", C.c_double), ("b_factor", C.c_double),
                ("stars_matched", C.c_uint64), ("stars_total", C.c_uint64), ("avg_color_index", C.c_double)]


class BatchStackConfigC(C.Structure):  # calibration_pipeline.rs:20-37
    _fields_ = [("sigma_low", C.c_float), ("sigma_high", C.c_float), ("max_iterations", C.c_uint64),
                ("normalize_before_stack", C.c_int32)]


class CalibrationMastersC(C.Structure):  # calibration_pipeline.rs:6-11 (NULL = None)
    _fields_ = [("bias", C.POINTER(Plane)), ("dark", C.POINTER(Plane)), ("flat", C.POINTER(Plane))]


class BatchChannelStatsC(C.Structure):  # calibration_pipeline.rs:65-72
    _fields_ = [("lights_input", C.c_uint64), ("mean", C.c_double), ("stddev", C.c_double)]


class BatchChannelInputC(C.Structure):  # calibration_pipeline.rs:13-17
    _fields_ = [("label", C.c_char_p), ("lights", C.POINTER(Plane)), ("n_lights", C.c_size_t),
                ("rejection_counts", C.POINTER(C.c_uint64))]


class TileLevelC(C.Structure):  # infra/render/tiles.rs:21-29 (+ offset into the packed tile buffer)
    _fields_ = [("level", C.c_uint64), ("width", C.c_uint64), ("height", C.c_uint64), ("cols", C.c_uint64),
                ("rows", C.c_uint64), ("scale_factor", C.c_double), ("offset", C.c_uint64)]


MAX_TILE_LEVELS = 32


class SubframeWeightConfigC(C.Structure):  # subframe.rs:24-49
    _fields_ = [("fwhm_weight", C.c_double), ("eccentricity_weight", C.c_double), ("snr_weight", C.c_double),
                ("noise_weight", C.c_double), ("max_fwhm", C.c_double), ("max_eccentricity", C.c_double),
                ("min_snr", C.c_double), ("min_stars", C.c_uint64)]


class SubframeMetricsC(C.Structure):  # subframe.rs:9-22
    _fields_ = [("star_count", C.c_uint64), ("median_fwhm", C.c_double), ("median_eccentricity", C.c_double),
                ("median_snr", C.c_double), ("background_median", C.c_double), ("background_sigma", C.c_double),
                ("noise_ratio", C.c_double), ("weight", C.c_double), ("accepted", C.c_int32)]


def build(force: bool = False, verbose: bool = False) -> str:
    """hipcc --offload-arch=gfx950 every csrc/*.hip into astroburst_amd/libastroburst_hip.so."""
    cmd = ["make", "-C", CSRC, "-j8"] + (["-B"] if force else [])
    res = subprocess.run(cmd, capture_output=True, text=True)
    if verbose or res.returncode != 0:
        print(res.stdout[-4000:])
        print(res.stderr[-4000:])
    if res.returncode != 0:
        raise RuntimeError("building libastroburst_hip.so failed")
    return LIB_PATH


def declared_symbols() -> list[str]:
    """Every entry point include/astroburst_hip.h declares."""
    text = open(HEADER_PATH).read()
    return sorted(set(re.findall(r"AB_API[^;(]*?\b(ab_[a-z0-9_]+)\s*\(", text)))


_lib = None


def version() -> str:
    return lib().ab_version().decode()


def is_dev_build() -> bool:
    """True when the loaded library was built with -DAB_DEV_ABLATION (`make -C astroburst_amd/csrc dev`, selected through AB_LIB_PATH):
    only then are the developer switches of ab_dev_env() (superseded forms, sweep knobs, fault injection) live."""
    return lib().ab_version().decode().endswith("+dev")


def lib() -> C.CDLL:
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise AstroBurstError(AB_ERR_NO_DEVICE,
                              f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'`"
                              " -- there is no CPU fallback")
    L = C.CDLL(LIB_PATH)
    vp, pp = C.c_void_p, C.POINTER(Plane)
    L.ab_ctx_create.argtypes = [C.c_int, C.POINTER(vp)]
    L.ab_ctx_destroy.argtypes = [vp]
    L.ab_ctx_destroy.restype = None
    L.ab_last_error.argtypes = [vp]
    L.ab_last_error.restype = C.c_char_p
    L.ab_version.restype = C.c_char_p
    L.ab_ctx_set_stream.argtypes = [vp, vp]
    L.ab_ctx_reset_stream.argtypes = [vp]
    L.ab_ctx_get_stream.argtypes = [vp]
    L.ab_ctx_get_stream.restype = vp
    L.ab_ctx_synchronize.argtypes = [vp]
    L.ab_device_alloc.argtypes = [vp, C.c_size_t, C.POINTER(vp)]
    L.ab_device_free.argtypes = [vp, vp]
    L.ab_upload.argtypes = [vp, vp, vp, C.c_size_t]
    L.ab_download.argtypes = [vp, vp, vp, C.c_size_t]
    L.ab_device_info.argtypes = [vp, C.c_char_p, C.c_size_t, C.POINTER(C.c_int), C.POINTER(C.c_uint64)]
    L.ab_stack_sigma_clip.argtypes = [vp, pp, C.c_size_t, C.POINTER(StackConfig), pp, C.POINTER(C.c_uint64)]
    L.ab_stack_last_rejected.argtypes = [vp, C.POINTER(C.c_uint64)]
    L.ab_stack_last_kernel_ms.argtypes = [vp, C.POINTER(C.c_float)]
    L.ab_stack_images.argtypes = [vp, pp, C.c_size_t, C.POINTER(StackConfig), pp, C.POINTER(C.c_int32),
                                  C.POINTER(C.c_uint64)]
    L.ab_stack_sigma_clip_partial.argtypes = [vp, pp, C.c_size_t, C.POINTER(StackConfig), C.c_int64, C.c_int64,
                                              vp, vp, C.POINTER(C.c_uint64)]
    L.ab_stack_finalize_partial.argtypes = [vp, vp, vp, C.c_int64, vp]
    L.ab_shift_image_subpixel.argtypes = [vp, pp, C.c_double, C.c_double, pp]
    L.ab_warp_image.argtypes = [vp, pp, C.POINTER(C.c_double), pp]
    L.ab_compute_image_stats.argtypes = [vp, pp, C.POINTER(ImageStatsC)]
    L.ab_compute_image_stats_with_known_range.argtypes = [vp, pp, C.c_double, C.c_double, C.POINTER(ImageStatsC)]
    L.ab_build_histogram.argtypes = [vp, pp, C.c_size_t, C.c_double, C.c_double, vp]
    L.ab_stats_value_hist.argtypes = [vp, pp, C.c_double, C.c_double, vp, C.POINTER(C.c_double),
                                      C.POINTER(C.c_uint64)]
    L.ab_auto_stf.argtypes = [C.POINTER(ImageStatsC), C.POINTER(AutoStfConfigC), C.POINTER(StfParamsC)]
    L.ab_apply_stf_u8.argtypes = [vp, pp, C.POINTER(StfParamsC), C.POINTER(ImageStatsC), vp, C.c_int32]
    L.ab_apply_stf_f32.argtypes = [vp, pp, C.POINTER(StfParamsC), C.POINTER(ImageStatsC), pp]
    L.ab_bench_copy.argtypes = [vp, vp, vp, C.c_size_t]
    L.ab_estimate_background.argtypes = [vp, pp, C.c_int64, C.POINTER(C.c_double), C.POINTER(C.c_double)]
    L.ab_background_tile_stats.argtypes = [vp, pp, C.c_int64, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_int32), C.c_size_t,
                                           C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]
    L.ab_detect_stars.argtypes = [vp, pp, C.c_double, C.POINTER(DetectedStarC), C.c_size_t, C.POINTER(C.c_size_t),
                                  C.POINTER(C.c_size_t), C.POINTER(C.c_double), C.POINTER(C.c_double)]
    L.ab_normalize_for_detection.argtypes = [vp, pp, pp]
    L.ab_align_channel_affine.argtypes = [vp, pp, pp, C.c_int, C.POINTER(AffineAlignResultC)]
    L.ab_affine_from_stars.argtypes = [C.POINTER(C.c_double), C.c_size_t, C.POINTER(C.c_double), C.c_size_t, C.c_int64,
                                       C.c_int64, C.c_int, C.POINTER(AffineAlignResultC), C.POINTER(C.c_int)]
    L.ab_phase_correlate.argtypes = [vp, pp, pp, C.POINTER(PhaseCorrelationResultC)]
    L.ab_correlate_single.argtypes = [vp, pp, pp, C.POINTER(PhaseCorrelationResultC), vp]
    L.ab_apply_scnr_inplace.argtypes = [vp, pp, pp, pp, C.POINTER(ScnrConfigC)]
    L.ab_blend_channels.argtypes = [vp, pp, C.c_size_t, C.POINTER(BlendWeightC), C.c_size_t, pp, pp, pp]
    L.ab_spline_lut_from_points.argtypes = [C.POINTER(C.c_double), C.c_size_t, C.POINTER(C.c_float)]
    L.ab_apply_curve.argtypes = [vp, pp, C.POINTER(C.c_float), pp]
    L.ab_apply_levels.argtypes = [vp, pp, C.POINTER(LevelsParamsC), pp]
    L.ab_arcsinh_stretch_with_stats.argtypes = [vp, pp, C.c_float, C.c_float, C.c_float, C.c_float, pp]
    L.ab_luminance.argtypes = [vp, pp, pp, pp, pp]
    L.ab_scale.argtypes = [vp, pp, C.c_float, pp]
    L.ab_calibrate_image.argtypes = [vp, pp, pp, pp, pp, C.c_float, pp]
    L.ab_median_combine.argtypes = [vp, pp, C.c_size_t, pp]
    L.ab_generate_star_mask.argtypes = [vp, pp, C.POINTER(StarMaskConfigC), pp, C.POINTER(StarMaskInfoC)]
    L.ab_generate_star_mask_from_stars.argtypes = [vp, pp, C.POINTER(DetectedStarC), C.c_size_t, C.POINTER(StarMaskConfigC),
                                                   pp, C.POINTER(StarMaskInfoC)]
    L.ab_masked_stretch.argtypes = [vp, pp, C.POINTER(MaskedStretchConfigC), pp, C.POINTER(MaskedStretchResultC)]
    L.ab_masked_stretch_with_mask.argtypes = [vp, pp, pp, C.POINTER(StarMaskInfoC), C.POINTER(MaskedStretchConfigC), pp,
                                              C.POINTER(MaskedStretchResultC)]
    L.ab_masked_stretch_rgb_shared.argtypes = [vp, pp, pp, pp, C.POINTER(MaskedStretchConfigC), pp, pp, pp,
                                               C.POINTER(MaskedStretchResultC), C.POINTER(StarMaskInfoC)]
    L.ab_resample_image.argtypes = [vp, pp, pp]
    L.ab_select_wb_reference.argtypes = [C.POINTER(ImageStatsC)] * 3 + [C.POINTER(C.c_double)]
    L.ab_process_rgb.argtypes = [vp, pp, pp, pp, C.POINTER(RgbComposeConfigC), pp, pp, pp, pp, pp, pp,
                                 C.POINTER(ProcessedRgbInfoC)]
    L.ab_spcc_calibrate_rgb.argtypes = [vp, pp, pp, pp, C.c_double, C.POINTER(SpccConfigC), C.POINTER(SpccResultC)]
    L.ab_spcc_from_detection.argtypes = [vp, pp, pp, pp, C.POINTER(DetectedStarC), C.c_size_t, C.c_double, C.c_double,
                                         C.POINTER(SpccConfigC), C.POINTER(SpccResultC)]
    L.ab_spcc_white_reference_rgb.argtypes = [C.c_int32, C.POINTER(C.c_double), C.POINTER(C.c_double)]
    L.ab_apply_lrgb.argtypes = [vp, pp, pp, pp, pp, C.c_float, C.c_float]
    L.ab_synthesize_luminance.argtypes = [vp, pp, pp, pp, pp]
    L.ab_compute_linked_stf.argtypes = [C.POINTER(ImageStatsC)] * 3 + [C.POINTER(AutoStfConfigC), C.POINTER(StfParamsC),
                                                                       C.POINTER(ImageStatsC)]
    L.ab_calibrate_channel.argtypes = [vp, pp, C.c_float, C.POINTER(ImageStatsC), pp, C.POINTER(ImageStatsC)]
    L.ab_create_master.argtypes = [vp, C.c_int32, pp, C.c_size_t, pp, pp, pp]
    L.ab_register_frames.argtypes = [vp, pp, pp, C.c_size_t, C.c_int, C.POINTER(AffineAlignResultC)]
    L.ab_align_pairs_affine.argtypes = [vp, pp, pp, C.c_size_t, C.c_int, C.POINTER(AffineAlignResultC), pp]
    L.ab_fits_decode_pixels.argtypes = [vp, vp, C.c_size_t, C.c_int32, C.c_int64, C.c_double, C.c_double, pp]
    L.ab_fits_compute_bzero_bscale.argtypes = [vp, pp, C.POINTER(C.c_double), C.POINTER(C.c_double)]
    L.ab_fits_encode_pixels.argtypes = [vp, pp, C.c_int32, C.c_double, C.c_double, vp, C.c_int32]
    L.ab_stack_sigma_clip_raw.argtypes = [vp, C.POINTER(vp), C.c_size_t, C.c_int64, C.c_double, C.c_double, C.POINTER(StackConfig), pp,
                                          C.POINTER(C.c_uint64)]
    L.ab_subframe_weight_config_default.argtypes = [C.POINTER(SubframeWeightConfigC)]
    L.ab_subframe_weight_config_default.restype = None
    L.ab_analyze_subframe.argtypes = [vp, pp, C.POINTER(SubframeWeightConfigC), C.POINTER(SubframeMetricsC)]
    L.ab_analyze_subframes.argtypes = [vp, pp, C.c_size_t, C.POINTER(SubframeWeightConfigC), C.POINTER(SubframeMetricsC)]
    L.ab_normalize_subframe_weights.argtypes = [C.POINTER(SubframeMetricsC), C.c_size_t]
    L.ab_normalize_subframe_weights.restype = None
    i64p = C.POINTER(C.c_int64)
    u64p = C.POINTER(C.c_uint64)
    L.ab_batch_stack_config_default.argtypes = [C.POINTER(BatchStackConfigC)]
    L.ab_batch_stack_config_default.restype = None
    L.ab_calibrate_light.argtypes = [vp, pp, C.POINTER(CalibrationMastersC), pp]
    L.ab_normalize_frames.argtypes = [vp, pp, C.c_size_t, pp]
    L.ab_sigma_clipped_mean_stack.argtypes = [vp, pp, C.c_size_t, C.POINTER(BatchStackConfigC), pp, u64p]
    L.ab_run_batch_channel.argtypes = [vp, pp, C.c_size_t, C.POINTER(CalibrationMastersC), C.POINTER(BatchStackConfigC), pp, u64p,
                                       C.POINTER(BatchChannelStatsC)]
    L.ab_compose_rgb_from_masters.argtypes = [vp, pp, pp, pp, pp, vp, C.c_int32, i64p, i64p]
    L.ab_run_batch_pipeline.argtypes = [vp, C.POINTER(BatchChannelInputC), C.c_size_t, C.POINTER(CalibrationMastersC),
                                        C.POINTER(BatchStackConfigC), pp, C.POINTER(BatchChannelStatsC), vp, C.c_int32, i64p, i64p]
    L.ab_preview_dims.argtypes = [C.c_int64, C.c_int64, C.c_int64, i64p, i64p]
    L.ab_render_rgb_preview.argtypes = [vp, pp, pp, pp, C.c_int64, C.POINTER(StfParamsC), C.POINTER(ImageStatsC), vp, C.c_int32]
    L.ab_ipc_encode_with_header.argtypes = [vp, pp, C.c_int64, vp, C.c_int32, C.POINTER(C.c_size_t)]
    L.ab_tile_compute_num_levels.argtypes = [C.c_int64, C.c_int64, C.c_int64]
    L.ab_tile_pyramid_layout.argtypes = [C.c_int64, C.c_int64, C.c_int64, C.c_int32, C.POINTER(TileLevelC), C.POINTER(C.c_int32),
                                         C.POINTER(C.c_size_t)]
    L.ab_tile_downsample_2x.argtypes = [vp, pp, pp]
    L.ab_tile_percentile_bounds.argtypes = [vp, pp, C.c_double, C.c_double, C.POINTER(C.c_float), C.POINTER(C.c_float)]
    L.ab_generate_tile_pyramid.argtypes = [vp, pp, C.c_int64, vp, C.c_int32, C.POINTER(TileLevelC), C.POINTER(C.c_int32),
                                           C.POINTER(C.c_float), C.POINTER(C.c_float)]
    L.ab_generate_tile_pyramid_rgb.argtypes = [vp, pp, pp, pp, C.c_int64, C.POINTER(StfParamsC), C.POINTER(ImageStatsC), vp, C.c_int32,
                                               C.POINTER(TileLevelC), C.POINTER(C.c_int32)]
    L.ab_extract_background.argtypes = [vp, pp, C.POINTER(BackgroundConfigC), pp, pp, C.POINTER(BackgroundInfoC)]
    # progress / cancel
    L.ab_ctx_set_progress_cb.argtypes = [vp, PROGRESS_CB, vp]
    L.ab_ctx_request_cancel.argtypes = [vp]
    L.ab_ctx_clear_cancel.argtypes = [vp]
    # (e) multi-GPU
    u8p = C.POINTER(C.c_uint8)
    L.ab_comm_get_unique_id.argtypes = [u8p]
    L.ab_comm_init_rank.argtypes = [vp, u8p, C.c_int, C.c_int, C.POINTER(vp)]
    L.ab_comm_init_all.argtypes = [C.POINTER(vp), C.c_int, C.POINTER(vp)]
    L.ab_comm_init_rank_host.argtypes = [vp, C.c_char_p, C.c_int, C.c_int, C.POINTER(vp)]
    L.ab_comm_is_host.argtypes = [vp]
    L.ab_comm_agree.argtypes = [vp, vp, C.c_int]
    L.ab_comm_abort.argtypes = [vp]
    L.ab_comm_set_timeout_ms.argtypes = [vp, C.c_int64]
    L.ab_comm_destroy.argtypes = [vp]
    L.ab_comm_destroy.restype = None
    L.ab_comm_rank.argtypes = [vp]
    L.ab_comm_size.argtypes = [vp]
    L.ab_comm_collectives_issued.argtypes = [vp]
    L.ab_comm_collectives_issued.restype = C.c_uint64
    L.ab_comm_group_start.argtypes = []
    L.ab_comm_group_end.argtypes = []
    L.ab_comm_allreduce.argtypes = [vp, vp, vp, C.c_size_t, C.c_int, C.c_int]
    L.ab_comm_allgather.argtypes = [vp, vp, vp, vp, C.c_size_t]
    L.ab_comm_broadcast.argtypes = [vp, vp, vp, C.c_size_t, C.c_int]
    L.ab_ctx_trim.argtypes = [vp]
    L.ab_ctx_fallback_counts.argtypes = [vp, C.POINTER(C.c_uint64), C.c_size_t, C.c_int]
    L.ab_shard_rows.argtypes = [C.c_int64, C.c_int, C.c_int, i64p, i64p]
    L.ab_shard_frames.argtypes = [C.c_size_t, C.c_int, C.c_int, C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]
    L.ab_stack_sigma_clip_rows.argtypes = [vp, pp, C.c_size_t, C.POINTER(StackConfig), C.c_int64, pp, u64p]
    L.ab_stack_sigma_clip_rowband.argtypes = [vp, vp, pp, C.c_size_t, C.POINTER(StackConfig), pp, u64p]
    L.ab_stack_sigma_clip_sharded.argtypes = [vp, vp, pp, C.c_size_t, C.POINTER(StackConfig), pp, u64p]
    L.ab_stack_sharded_last_ms.argtypes = [vp, C.POINTER(C.c_float), C.POINTER(C.c_float)]
    L.ab_allgather_rows.argtypes = [vp, vp, pp, pp]
    L.ab_register_frames_sharded.argtypes = [vp, vp, pp, pp, C.c_size_t, C.c_int, C.POINTER(AffineAlignResultC)]
    L.ab_align_pairs_affine_rowband.argtypes = [vp, vp, pp, pp, C.POINTER(C.c_int64), C.c_size_t, C.c_int, C.c_int64, C.POINTER(AffineAlignResultC), pp]
    L.ab_compute_image_stats_sharded.argtypes = [vp, vp, pp, C.c_int64, C.POINTER(ImageStatsC)]
    L.ab_warp_image_rows.argtypes = [vp, pp, C.POINTER(C.c_double), C.c_int64, C.c_int64, pp]
    L.ab_warp_image_rows_from_band.argtypes = [vp, pp, C.c_int64, C.c_int64, C.POINTER(C.c_double), C.c_int64, C.c_int64, pp]
    L.ab_warp_source_rows.argtypes = [C.POINTER(C.c_double), C.c_int64, C.c_int64, C.c_int64, C.c_int64, C.c_int64, i64p, i64p]
    L.ab_shard_source_rows.argtypes = [C.POINTER(C.c_double), C.c_size_t, C.c_int64, C.c_int64, C.c_int64, C.c_int64, C.c_int, C.c_int,
                                       i64p, i64p]
    L.ab_auto_stretch_preview.argtypes = [vp, vp, pp, C.c_int64, C.POINTER(AutoStfConfigC), vp, C.POINTER(ImageStatsC),
                                          C.POINTER(StfParamsC)]
    for name in declared_symbols():
        fn = getattr(L, name)  # AttributeError here = header / library drift
        if fn.restype is C.c_int and name not in ("ab_last_error", "ab_version", "ab_ctx_get_stream", "ab_comm_collectives_issued"):
            fn.restype = C.c_int
    _lib = L
    return L
