"""Host-side mirror of the reference's `core::*` functions for the hot path, over the C ABI.

Names, argument meaning and error behaviour follow the reference (src-tauri/src/core/...), so the
parity tests read like the reference's own unit tests.  Planes may be numpy arrays (host: staged
through HBM by the library) or torch CUDA tensors (device: used in place on torch's current
stream).  Every function here ends in a HIP kernel of libastroburst_hip.so; nothing is computed
in Python or on the CPU.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass

import numpy as np

from . import _lib
from ._lib import AstroBurstError, AutoStfConfigC, ImageStatsC, Plane, StackConfig, StfParamsC

try:  # torch is plumbing only (device memory, streams, torch.distributed)
    import torch
except Exception:  # pragma: no cover
    torch = None


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
class StfParams:  # types/image.rs:36-50
    shadow: float = 0.0
    midtone: float = 0.5
    highlight: float = 1.0


WHITE_REFERENCES = {"average_spiral": 0, "g2v": 1, "photopic": 2}


@dataclass
class SpccResult:  # spcc.rs:45-56
    r_factor: float
    g_factor: float
    b_factor: float
    stars_matched: int
    stars_total: int
    avg_color_index: float
    white_ref_name: str
    catalog_name: str = "Built-in Bp-Rp"
    is_synthetic_catalog: bool = True


@dataclass
class ProcessedRgb:  # rgb.rs:18-40
    r: object
    g: object
    b: object
    rows: int
    cols: int
    stf: tuple            # (StfParams r, g, b)
    channel_stats: tuple  # 3 x (min, max, median, mean) before white balance
    offset_g: tuple
    offset_b: tuple
    scnr_applied: bool
    resampled: bool
    pre_stretch: tuple    # (r, g, b) white-balanced, pre-STF planes (or Nones)
    stats_wb: tuple       # 3 x ImageStats after white balance


@dataclass
class StarMaskResult:  # star_mask.rs:32-37
    mask: object
    stars_masked: int
    coverage_fraction: float


@dataclass
class MaskedStretchResult:  # masked_stretch.rs:34-42
    image: object
    iterations_run: int
    final_background: float
    stars_masked: int
    mask_coverage: float
    converged: bool


@dataclass
class BackgroundResult:  # background.rs:35-42
    model: object
    corrected: object
    sample_count: int
    rms_residual: float
    coeffs: np.ndarray


@dataclass
class StackResult:  # types/stacking.rs:22-28
    image: object
    frame_count: int
    rejected_pixels: int
    offsets: list


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


@dataclass
class BatchStackConfig:  # calibration_pipeline.rs:20-37
    sigma_low: float = 2.5
    sigma_high: float = 3.0
    max_iterations: int = 5
    normalize_before_stack: bool = True

    def _c(self):
        return _lib.BatchStackConfigC(self.sigma_low, self.sigma_high, self.max_iterations, int(self.normalize_before_stack))


@dataclass
class BatchChannelStats:  # calibration_pipeline.rs:65-72
    label: str
    lights_input: int
    lights_after_rejection: list
    mean: float
    stddev: float


@dataclass
class BatchPipelineResult:  # calibration_pipeline.rs:51-56
    master_channels: list          # [(label, master)]
    rgb: object                    # (h, w, 3) f32 or None
    channels: list                 # [BatchChannelStats]
    darks_combined: int = 0
    flats_combined: int = 0
    bias_combined: int = 0


@dataclass
class SubframeWeightConfig:  # subframe.rs:24-49
    fwhm_weight: float = 1.0
    eccentricity_weight: float = 0.5
    snr_weight: float = 1.0
    noise_weight: float = 0.3
    max_fwhm: float = 8.0
    max_eccentricity: float = 0.7
    min_snr: float = 5.0
    min_stars: int = 5

    def _c(self):
        return _lib.SubframeWeightConfigC(self.fwhm_weight, self.eccentricity_weight, self.snr_weight, self.noise_weight,
                                          self.max_fwhm, self.max_eccentricity, self.min_snr, self.min_stars)


@dataclass
class SubframeMetrics:  # subframe.rs:9-22 (file_path stays with the caller)
    star_count: int
    median_fwhm: float
    median_eccentricity: float
    median_snr: float
    background_median: float
    background_sigma: float
    noise_ratio: float
    weight: float
    accepted: bool


AFFINE_METHODS = ("affine", "rigid", "phase_correlation", "identity")


def _is_torch(x) -> bool:
    return torch is not None and isinstance(x, torch.Tensor)


class Comm:
    """One RCCL rank bound to a Context's GPU (include/astroburst_hip.h section (e)).  Multi-process: rank 0 calls
    Comm.unique_id(), the host carries the 128 bytes to the other ranks, every rank constructs Comm(ctx, id, nranks, rank)."""

    @staticmethod
    def unique_id() -> bytes:
        buf = (C.c_uint8 * _lib.AB_COMM_ID_BYTES)()
        rc = _lib.lib().ab_comm_get_unique_id(buf)
        if rc != _lib.AB_OK:
            raise AstroBurstError(rc, "ab_comm_get_unique_id failed (librccl.so.1 not loadable?)")
        return bytes(buf)

    def __init__(self, ctx: "Context", uid: bytes, nranks: int, rank: int):
        assert len(uid) == _lib.AB_COMM_ID_BYTES
        self._ctx = ctx
        self._L = ctx._L
        h = C.c_void_p()
        buf = (C.c_uint8 * _lib.AB_COMM_ID_BYTES).from_buffer_copy(uid)
        ctx._check(self._L.ab_comm_init_rank(ctx._h, buf, nranks, rank, C.byref(h)))
        self._h = h

    @classmethod
    def host(cls, ctx: "Context", name: str, nranks: int, rank: int) -> "Comm":
        """ab_comm_init_rank_host: host-staged collectives through the shared segment /abcomm_<name> (ranks may share a GPU)."""
        self = cls.__new__(cls)
        self._ctx = ctx
        self._L = ctx._L
        h = C.c_void_p()
        ctx._check(self._L.ab_comm_init_rank_host(ctx._h, name.encode(), nranks, rank, C.byref(h)))
        self._h = h
        return self

    @property
    def is_host(self) -> bool:
        return bool(self._L.ab_comm_is_host(self._h))

    def agree(self, local_status: int = 0) -> int:
        """ab_comm_agree: AB_OK only if every rank passed AB_OK (raises otherwise)"""
        self._ctx.use_torch_stream()
        rc = self._L.ab_comm_agree(self._ctx._h, self._h, int(local_status))
        self._ctx._check(rc)
        return rc

    def abort(self):
        self._L.ab_comm_abort(self._h)

    def set_timeout_ms(self, ms: int):
        self._ctx._check(self._L.ab_comm_set_timeout_ms(self._h, int(ms)))

    @property
    def rank(self) -> int:
        return self._L.ab_comm_rank(self._h)

    @property
    def size(self) -> int:
        return self._L.ab_comm_size(self._h)

    @property
    def collectives_issued(self) -> int:
        return int(self._L.ab_comm_collectives_issued(self._h))

    _DT = {"torch.int32": _lib.AB_DT_I32, "torch.int64": _lib.AB_DT_I64, "torch.float32": _lib.AB_DT_F32, "torch.float64": _lib.AB_DT_F64,
           "torch.uint32": _lib.AB_DT_U32, "torch.uint64": _lib.AB_DT_U64}

    def allreduce(self, t, op: str = "sum"):
        """in place, on the context's stream"""
        assert t.is_cuda and t.is_contiguous()
        self._ctx.use_torch_stream()
        ops = {"sum": _lib.AB_RED_SUM, "max": _lib.AB_RED_MAX, "min": _lib.AB_RED_MIN}
        self._ctx._check(self._L.ab_comm_allreduce(self._ctx._h, self._h, C.c_void_p(t.data_ptr()), t.numel(), self._DT[str(t.dtype)], ops[op]))
        return t

    def close(self):
        if getattr(self, "_h", None):
            self._L.ab_comm_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class PlaneList:
    """The ab_plane descriptors of a list of planes, built ONCE (Context.planes).  A host in Rust or C holds its `ab_plane` array ready;
    this module rebuilds one per call -- 2 us of Python per plane, 0.4 ms of an idle GPU for the bench step's 190 planes -- so a caller
    that passes the same frames again and again (bench.py) marshals them once and passes the list."""

    def __init__(self, ctx, frames):
        self.frames = list(frames)
        self.keep = []
        self.n = len(self.frames)
        self.array = (Plane * max(self.n, 1))(*[ctx._plane(f, self.keep) for f in self.frames])
        self.device = any(_is_torch(f) and f.is_cuda for f in self.frames)
        self.rows = min((p.rows for p in self.array[:self.n]), default=0)
        self.cols = min((p.cols for p in self.array[:self.n]), default=0)

    def __len__(self):
        return self.n

    def __getitem__(self, i):
        return self.frames[i]


class Context:
    """One ab_ctx: a device, a stream, a scratch arena.  Not thread-safe; one per caller thread."""

    def __init__(self, device: int = 0):
        self._L = _lib.lib()
        h = C.c_void_p()
        rc = self._L.ab_ctx_create(device, C.byref(h))
        if rc != _lib.AB_OK:
            raise AstroBurstError(rc, "ab_ctx_create failed: no gfx950 (MI355X) device visible"
                                      if rc == _lib.AB_ERR_NO_DEVICE else "ab_ctx_create failed")
        self._h = h
        self.device = device
        self._stream = None   # the torch stream the context currently launches on (None: its own)

    def close(self):
        if getattr(self, "_h", None):
            self._L.ab_ctx_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc: int):
        if rc != _lib.AB_OK:
            raise AstroBurstError(rc, self._L.ab_last_error(self._h).decode())

    def use_torch_stream(self):
        """Launch on torch's current stream: calls are then ordered with the torch kernels that produce their inputs and
        consume their outputs, and torch.cuda.Event timing brackets them.  Done automatically whenever a torch tensor is
        passed as a plane (a device plane handed over by torch is only meaningful in torch's stream order)."""
        s = torch.cuda.current_stream().cuda_stream
        if s != self._stream:
            self._check(self._L.ab_ctx_set_stream(self._h, C.c_void_p(s)))
            self._stream = s

    def use_own_stream(self):
        """Back to the context's private stream (host planes; C callers that manage ordering themselves)."""
        self._check(self._L.ab_ctx_reset_stream(self._h))
        self._stream = None

    def synchronize(self):
        self._check(self._L.ab_ctx_synchronize(self._h))

    def trim(self):
        """release the device memory the context has grown (workspaces, scratch, host-frame staging); it stays usable"""
        self._check(self._L.ab_ctx_trim(self._h))

    FALLBACK_KINDS = ("frames_redone", "tile_slots", "component_table", "selection_short", "selection_cut", "tiles_declined",
                      "stats_chain", "stack_general_pixels", "label_tiles_dense")   # ab_fallback_kind, include/astroburst_hip.h

    def fallback_counts(self, reset: bool = False) -> dict:
        """How often a fast path handed over to its exact fallback (same results, more time) since the context was created or last
        reset -- frame workers included (ab_ctx_fallback_counts)."""
        out = (C.c_uint64 * len(self.FALLBACK_KINDS))()
        self._check(self._L.ab_ctx_fallback_counts(self._h, out, len(self.FALLBACK_KINDS), 1 if reset else 0))
        return dict(zip(self.FALLBACK_KINDS, (int(v) for v in out)))

    def device_info(self):
        name = C.create_string_buffer(256)
        cu, mem = C.c_int(), C.c_uint64()
        self._check(self._L.ab_device_info(self._h, name, 256, C.byref(cu), C.byref(mem)))
        return name.value.decode(), cu.value, mem.value

    # ---- plane marshalling -----------------------------------------------------------------
    def planes(self, frames) -> "PlaneList":
        """the frames' ab_plane descriptors, marshalled once; every call that takes a list of planes takes the result too"""
        return frames if isinstance(frames, PlaneList) else PlaneList(self, frames)

    def _plane_array(self, frames, keep):
        """(ctypes array of ab_plane, count) of a list of planes or of a PlaneList"""
        if isinstance(frames, PlaneList):
            if frames.device:
                self.use_torch_stream()   # (what _plane does once per call for device planes)
            return frames.array, frames.n
        return (Plane * max(len(frames), 1))(*[self._plane(f, keep) for f in frames]), len(frames)

    def _plane(self, x, keep):
        if _is_torch(x) and not x.is_cuda:  # a host plane held by torch (pinned memory keeps the library's uploads asynchronous)
            assert x.dtype == torch.float32 and x.dim() == 2 and x.is_contiguous(), "host planes are contiguous 2-D float32 tensors"
            keep.append(x)
            return Plane(C.c_void_p(x.data_ptr()), x.shape[0], x.shape[1], 0)
        if _is_torch(x):
            assert x.is_cuda and x.dtype == torch.float32 and x.dim() == 2 and x.is_contiguous(), \
                "device planes are contiguous 2-D float32 CUDA tensors"
            if not keep or not _is_torch(keep[-1]):  # once per call: torch.cuda.current_stream() costs microseconds, a stack has 64 planes
                self.use_torch_stream()
            keep.append(x)
            return Plane(C.c_void_p(x.data_ptr()), x.shape[0], x.shape[1], 1)
        a = np.ascontiguousarray(x, dtype=np.float32)
        assert a.ndim == 2, "planes are 2-D (rows, cols)"
        keep.append(a)
        return Plane(C.c_void_p(a.ctypes.data), a.shape[0], a.shape[1], 0)

    def _new_like(self, ref, rows, cols):
        if _is_torch(ref):
            return torch.empty((rows, cols), dtype=torch.float32, device=ref.device)
        return np.empty((rows, cols), np.float32)

    # ---- core/stacking/combine.rs ------------------------------------------------------------
    def stack_sigma_clip(self, frames, sigma_low=3.0, sigma_high=3.0, max_iterations=5, out=None, want_rejected=True):
        """Per-pixel sigma_clip_combine over equal-or-larger frames (combine.rs:14-92,160-182).
        want_rejected=False keeps the call asynchronous (fetch the count with last_rejected())."""
        if len(frames) == 0:
            raise AstroBurstError(_lib.AB_ERR_INVALID, "No images to stack")
        keep = []
        planes, n = self._plane_array(frames, keep)
        if isinstance(frames, PlaneList):
            rows, cols = frames.rows, frames.cols
        else:
            rows = min(p.rows for p in planes)
            cols = min(p.cols for p in planes)
        if out is None:
            out = self._new_like(frames[0], rows, cols)
        po = self._plane(out, keep) if _is_torch(out) else Plane(C.c_void_p(out.ctypes.data), rows, cols, 0)
        cfg = StackConfig(sigma_low, sigma_high, max_iterations, 0)
        rej = C.c_uint64(0)
        self._check(self._L.ab_stack_sigma_clip(self._h, planes, n, C.byref(cfg), C.byref(po),
                                                C.byref(rej) if want_rejected else None))
        return out, (int(rej.value) if want_rejected else None)

    def last_rejected(self) -> int:
        rej = C.c_uint64(0)
        self._check(self._L.ab_stack_last_rejected(self._h, C.byref(rej)))
        return int(rej.value)

    def stack_last_kernel_ms(self) -> float:
        """duration of the last stack launch's kernels (HIP events recorded by the library on the launch stream)"""
        ms = C.c_float()
        self._check(self._L.ab_stack_last_kernel_ms(self._h, C.byref(ms)))
        return ms.value

    def stack_images(self, images, sigma_low=3.0, sigma_high=3.0, max_iterations=5, align=True) -> StackResult:
        """stack_images(&[Array2<f32>], &StackConfig) (combine.rs:94-193)."""
        n = len(images)
        if n == 0:
            raise AstroBurstError(_lib.AB_ERR_INVALID, "No images to stack")
        keep = []
        planes = (Plane * n)(*[self._plane(f, keep) for f in images])
        rows = min(p.rows for p in planes)
        cols = min(p.cols for p in planes)
        out = self._new_like(images[0], rows, cols)
        po = self._plane(out, keep) if _is_torch(out) else Plane(C.c_void_p(out.ctypes.data), rows, cols, 0)
        cfg = StackConfig(sigma_low, sigma_high, max_iterations, 1 if align else 0)
        rej = C.c_uint64(0)
        offs = (C.c_int32 * (2 * n))()
        self._check(self._L.ab_stack_images(self._h, planes, n, C.byref(cfg), C.byref(po), offs, C.byref(rej)))
        offsets = [(int(offs[2 * i]), int(offs[2 * i + 1])) for i in range(n)]
        return StackResult(out, n, int(rej.value), offsets)

    def sigma_clip_combine(self, values, sigma_low=3.0, sigma_high=3.0, max_iter=5):
        """sigma_clip_combine(values, lo, hi, max_iter) -> (f32, u32) (combine.rs:14-92), run on
        the GPU as a one-pixel stack of len(values) frames."""
        vals = np.asarray(values, dtype=np.float32).ravel()
        if vals.size == 0:
            return np.float32(0.0), 0  # combine.rs:21-23 (no frames: nothing to launch)
        frames = [np.full((1, 1), v, np.float32) for v in vals]
        out, rej = self.stack_sigma_clip(frames, sigma_low, sigma_high, max_iter)
        return np.float32(out[0, 0]), rej

    def stack_partial(self, frames, sigma_low=3.0, sigma_high=3.0, max_iterations=5):
        """Frame-sharded partial: per-pixel (f64 sum, u32 count) of this shard's survivors."""
        assert all(_is_torch(f) for f in frames), "partial stacking takes device frames"
        keep = []
        planes = (Plane * len(frames))(*[self._plane(f, keep) for f in frames])
        rows = min(p.rows for p in planes)
        cols = min(p.cols for p in planes)
        dev = frames[0].device
        s = torch.empty((rows, cols), dtype=torch.float64, device=dev)
        cnt = torch.empty((rows, cols), dtype=torch.int32, device=dev)
        cfg = StackConfig(sigma_low, sigma_high, max_iterations, 0)
        rej = C.c_uint64(0)
        self._check(self._L.ab_stack_sigma_clip_partial(self._h, planes, len(frames), C.byref(cfg), rows, cols,
                                                        C.c_void_p(s.data_ptr()), C.c_void_p(cnt.data_ptr()),
                                                        C.byref(rej)))
        return s, cnt, int(rej.value)

    def stack_partial_into(self, frames, s, cnt, sigma_low=3.0, sigma_high=3.0, max_iterations=5):
        """stack_partial into caller-owned device buffers, fully asynchronous (no rejected read-back)."""
        keep = []
        planes = (Plane * len(frames))(*[self._plane(f, keep) for f in frames])
        cfg = StackConfig(sigma_low, sigma_high, max_iterations, 0)
        self._check(self._L.ab_stack_sigma_clip_partial(self._h, planes, len(frames), C.byref(cfg), s.shape[0],
                                                        s.shape[1], C.c_void_p(s.data_ptr()),
                                                        C.c_void_p(cnt.data_ptr()), None))
        return s, cnt, None

    def stack_finalize_partial_into(self, s, cnt, out):
        self._check(self._L.ab_stack_finalize_partial(self._h, C.c_void_p(s.data_ptr()), C.c_void_p(cnt.data_ptr()),
                                                      s.numel(), C.c_void_p(out.data_ptr())))
        return out

    def stack_finalize_partial(self, s, cnt):
        out = torch.empty(s.shape, dtype=torch.float32, device=s.device)
        self._check(self._L.ab_stack_finalize_partial(self._h, C.c_void_p(s.data_ptr()), C.c_void_p(cnt.data_ptr()),
                                                      s.numel(), C.c_void_p(out.data_ptr())))
        return out

    # ---- progress / cancel (infra/progress.rs:39-74) ------------------------------------------
    def set_progress_cb(self, fn):
        """fn(stage: str, current: int, total: int) at the reference's stage boundaries; None removes it."""
        if fn is None:
            self._progress = None
            self._check(self._L.ab_ctx_set_progress_cb(self._h, _lib.PROGRESS_CB(0), None))
            return
        self._progress = _lib.PROGRESS_CB(lambda stage, cur, tot, _u: fn(stage.decode(), int(cur), int(tot)))
        self._check(self._L.ab_ctx_set_progress_cb(self._h, self._progress, None))

    def request_cancel(self):
        self._check(self._L.ab_ctx_request_cancel(self._h))

    def clear_cancel(self):
        self._check(self._L.ab_ctx_clear_cancel(self._h))

    # ---- (e) multi-GPU: row bands, frame shards (csrc/sharded.hip) ---------------
    def shard_rows(self, rows: int, nranks: int, rank: int):
        r0, nr = C.c_int64(), C.c_int64()
        self._check(self._L.ab_shard_rows(rows, nranks, rank, C.byref(r0), C.byref(nr)))
        return int(r0.value), int(nr.value)

    def shard_frames(self, n_frames: int, nranks: int, rank: int):
        f0, nf = C.c_size_t(), C.c_size_t()
        self._check(self._L.ab_shard_frames(n_frames, nranks, rank, C.byref(f0), C.byref(nf)))
        return int(f0.value), int(nf.value)

    def stack_sigma_clip_rows(self, frames, row0: int, nrows: int, sigma_low=3.0, sigma_high=3.0, max_iterations=5, out=None,
                              want_rejected=True):
        """rows [row0, row0 + nrows) of the stack of ALL frames (the exact single-level estimator on a band)."""
        keep = []
        planes = (Plane * len(frames))(*[self._plane(f, keep) for f in frames])
        cols = min(p.cols for p in planes)
        if out is None:
            out = torch.empty((nrows, cols), dtype=torch.float32, device=frames[0].device)
        po = Plane(C.c_void_p(out.data_ptr()), nrows, cols, 1)
        cfg = StackConfig(sigma_low, sigma_high, max_iterations, 0)
        rej = C.c_uint64(0)
        self._check(self._L.ab_stack_sigma_clip_rows(self._h, planes, len(frames), C.byref(cfg), row0, C.byref(po),
                                                     C.byref(rej) if want_rejected else None))
        return out, (int(rej.value) if want_rejected else None)

    def stack_sigma_clip_rowband(self, comm, frames, out_band, sigma_low=3.0, sigma_high=3.0, max_iterations=5, want_rejected=True):
        """this rank's band of the exact stack + StackResult.rejected_pixels summed over the ranks"""
        keep = []
        planes = (Plane * len(frames))(*[self._plane(f, keep) for f in frames])
        po = Plane(C.c_void_p(out_band.data_ptr()), out_band.shape[0], out_band.shape[1], 1)
        cfg = StackConfig(sigma_low, sigma_high, max_iterations, 0)
        rej = C.c_uint64(0)
        self._check(self._L.ab_stack_sigma_clip_rowband(self._h, comm._h if comm else None, planes, len(frames), C.byref(cfg), C.byref(po),
                                                        C.byref(rej) if want_rejected else None))
        return out_band, (int(rej.value) if want_rejected else None)

    def stack_sigma_clip_sharded(self, comm, local_frames, out, sigma_low=3.0, sigma_high=3.0, max_iterations=5, want_rejected=False):
        """frame-sharded two-level stack: partial over this rank's frames -> RCCL all-reduce(sum, count) -> divide"""
        keep = []
        planes = (Plane * len(local_frames))(*[self._plane(f, keep) for f in local_frames])
        po = self._plane(out, keep)
        cfg = StackConfig(sigma_low, sigma_high, max_iterations, 0)
        rej = C.c_uint64(0)
        self._check(self._L.ab_stack_sigma_clip_sharded(self._h, comm._h if comm else None, planes, len(local_frames), C.byref(cfg),
                                                        C.byref(po), C.byref(rej) if want_rejected else None))
        return out, (int(rej.value) if want_rejected else None)

    def stack_sharded_last_ms(self):
        """(span of the partial stacks on the context's stream, time inside the all-reduces + divisions on the comm stream) of the last
        stack_sigma_clip_sharded; blocks until that call's work is done"""
        a, b = C.c_float(0.0), C.c_float(0.0)
        self._check(self._L.ab_stack_sharded_last_ms(self._h, C.byref(a), C.byref(b)))
        return float(a.value), float(b.value)

    def allgather_rows(self, comm, band, full):
        keep = []
        pb = Plane(C.c_void_p(band.data_ptr() if band.numel() else 0), band.shape[0], band.shape[1], 1)
        pf = self._plane(full, keep)
        self._check(self._L.ab_allgather_rows(self._h, comm._h if comm else None, C.byref(pb), C.byref(pf)))
        return full

    def register_frames_sharded(self, comm, reference, targets, num_threads: int = 8):
        """align_channel_affine(reference, t) for every t, target i estimated on rank i mod size; all results everywhere"""
        keep = []
        pr = self._plane(reference, keep)
        planes = (Plane * max(len(targets), 1))(*[self._plane(t, keep) for t in targets])
        res = (_lib.AffineAlignResultC * max(len(targets), 1))()
        self._check(self._L.ab_register_frames_sharded(self._h, comm._h if comm else None, C.byref(pr), planes, len(targets), num_threads, res))
        return [AffineAlignResult(tuple(r.transform), int(r.matched_stars), int(r.inliers), r.residual_px, AFFINE_METHODS[r.method])
                for r in res[:len(targets)]]

    def align_pairs_affine_rowband(self, comm, reference, targets, out_bands, row0: int, target_row0=None, num_threads: int = 8):
        """the row-band scheme's registration as one call: this rank's rows [row0, row0 + band rows) of every registered frame;
        targets[i] = rows [target_row0[i], ...) of target i (whole for i mod size == rank); own frames are warped as they are fitted"""
        keep = []
        pr = self._plane(reference, keep)
        n = len(targets)
        planes = (Plane * max(n, 1))()
        for i, t in enumerate(targets):
            if t.numel() == 0:
                planes[i] = Plane(C.c_void_p(0), 0, reference.shape[1], 1)
            else:
                planes[i] = self._plane(t, keep)
        outs = (Plane * max(n, 1))()
        for i, o in enumerate(out_bands):
            outs[i] = Plane(C.c_void_p(o.data_ptr() if o.numel() else 0), o.shape[0], o.shape[1], 1)
        t0 = (C.c_int64 * max(n, 1))(*[int(v) for v in (target_row0 if target_row0 is not None else [0] * n)]) if n else None
        res = (_lib.AffineAlignResultC * max(n, 1))()
        self._check(self._L.ab_align_pairs_affine_rowband(self._h, comm._h if comm else None, C.byref(pr), planes, t0, n, num_threads, row0, res, outs))
        return [AffineAlignResult(tuple(r.transform), int(r.matched_stars), int(r.inliers), r.residual_px, AFFINE_METHODS[r.method]) for r in res[:n]]

    def compute_image_stats_sharded(self, comm, band, total_rows: int) -> ImageStats:
        keep = []
        pb = self._plane(band, keep)
        s = ImageStatsC()
        self._check(self._L.ab_compute_image_stats_sharded(self._h, comm._h if comm else None, C.byref(pb), total_rows, C.byref(s)))
        return self._stats_out(s)

    def warp_image_rows(self, image, transform, out_rows: int, row0: int, out_band):
        keep = []
        pi = self._plane(image, keep)
        po = self._plane(out_band, keep)
        t = (C.c_double * 6)(*[float(v) for v in transform])
        self._check(self._L.ab_warp_image_rows(self._h, C.byref(pi), t, out_rows, row0, C.byref(po)))
        return out_band

    def warp_image_rows_from_band(self, src_band, src_row0: int, src_rows: int, transform, out_rows: int, row0: int, out_band):
        """rows [row0, ..) of warp_image with the source given as rows [src_row0, src_row0 + len(src_band)) of a src_rows-row frame"""
        keep = []
        pi = self._plane(src_band, keep)
        po = self._plane(out_band, keep)
        t = (C.c_double * 6)(*[float(v) for v in transform])
        self._check(self._L.ab_warp_image_rows_from_band(self._h, C.byref(pi), src_row0, src_rows, t, out_rows, row0, C.byref(po)))
        return out_band

    def warp_source_rows(self, transform, src_rows: int, src_cols: int, out_cols: int, row0: int, nrows: int):
        """(first source row, count) that output rows [row0, row0 + nrows) of warp_image read"""
        t = (C.c_double * 6)(*[float(v) for v in transform])
        s0, sn = C.c_int64(), C.c_int64()
        self._check(self._L.ab_warp_source_rows(t, src_rows, src_cols, out_cols, row0, nrows, C.byref(s0), C.byref(sn)))
        return s0.value, sn.value

    def shard_source_rows(self, transforms, src_rows: int, src_cols: int, out_rows: int, out_cols: int, nranks: int, rank: int):
        """(first source row, count): this rank's rows of every target + the halo the transforms need"""
        flat = [float(v) for tr in transforms for v in tr]
        t = (C.c_double * max(len(flat), 1))(*flat)
        s0, sn = C.c_int64(), C.c_int64()
        self._check(self._L.ab_shard_source_rows(t, len(transforms), src_rows, src_cols, out_rows, out_cols, nranks, rank, C.byref(s0), C.byref(sn)))
        return s0.value, sn.value

    def auto_stretch_preview(self, image, out=None, comm=None, total_rows: int = 0, target_bg=0.25, shadow_k=-2.8, fetch=True):
        """auto_stretch_preview (cmd/common.rs:18-22): compute_image_stats -> auto_stf -> apply_stf (u8) as one asynchronous device
        chain.  -> (u8 plane, ImageStats | None, StfParams | None); fetch=False leaves the call fully asynchronous."""
        keep = []
        pi = self._plane(image, keep)
        if out is None:
            out = torch.empty((pi.rows, pi.cols), dtype=torch.uint8, device=image.device)
        cfg = AutoStfConfigC(target_bg, shadow_k)
        s, p = ImageStatsC(), StfParamsC()
        self._check(self._L.ab_auto_stretch_preview(self._h, comm._h if comm else None, C.byref(pi), total_rows, C.byref(cfg),
                                                    C.c_void_p(out.data_ptr()), C.byref(s) if fetch else None, C.byref(p) if fetch else None))
        if not fetch:
            return out, None, None
        return out, self._stats_out(s), StfParams(p.shadow, p.midtone, p.highlight)

    # ---- core/stacking/align.rs, core/alignment/affine.rs -------------------------------------
    def shift_image_subpixel(self, image, dy: float, dx: float, out=None):
        keep = []
        pi = self._plane(image, keep)
        if out is None:
            out = self._new_like(image, pi.rows, pi.cols)
        po = self._plane(out, keep) if _is_torch(out) else Plane(C.c_void_p(out.ctypes.data), pi.rows, pi.cols, 0)
        self._check(self._L.ab_shift_image_subpixel(self._h, C.byref(pi), dy, dx, C.byref(po)))
        return out

    def warp_image(self, image, transform, out_rows: int, out_cols: int, out=None):
        """transform = (a, b, tx, c, d, ty): output (x, y) -> source (sx, sy) (affine.rs:74-80)."""
        keep = []
        pi = self._plane(image, keep)
        if out is None:
            out = self._new_like(image, out_rows, out_cols)
        po = self._plane(out, keep) if _is_torch(out) else Plane(C.c_void_p(out.ctypes.data), out_rows, out_cols, 0)
        t = (C.c_double * 6)(*[float(v) for v in transform])
        self._check(self._L.ab_warp_image(self._h, C.byref(pi), t, C.byref(po)))
        return out

    # ---- core/imaging/stats.rs ------------------------------------------------------------------
    @staticmethod
    def _stats_out(s: ImageStatsC) -> ImageStats:
        return ImageStats(s.min, s.max, s.median, s.mad, s.sigma, s.mean, int(s.valid_count))

    @staticmethod
    def _stats_in(st: ImageStats) -> ImageStatsC:
        return ImageStatsC(st.min, st.max, st.median, st.mad, st.sigma, st.mean, st.valid_count)

    def compute_image_stats(self, image) -> ImageStats:
        keep = []
        pi = self._plane(image, keep)
        s = ImageStatsC()
        self._check(self._L.ab_compute_image_stats(self._h, C.byref(pi), C.byref(s)))
        return self._stats_out(s)

    def compute_image_stats_with_known_range(self, image, known_min: float, known_max: float) -> ImageStats:
        keep = []
        pi = self._plane(image, keep)
        s = ImageStatsC()
        self._check(self._L.ab_compute_image_stats_with_known_range(self._h, C.byref(pi), known_min, known_max,
                                                                    C.byref(s)))
        return self._stats_out(s)

    def build_histogram(self, image, bins: int, dmin: float, dmax: float) -> np.ndarray:
        keep = []
        pi = self._plane(image, keep)
        out = np.zeros(bins, np.uint32)
        self._check(self._L.ab_build_histogram(self._h, C.byref(pi), bins, dmin, dmax, C.c_void_p(out.ctypes.data)))
        return out

    def stats_value_hist(self, image, gmin: float, gmax: float):
        keep = []
        pi = self._plane(image, keep)
        h = np.zeros(65536, np.uint64)
        s, c = C.c_double(), C.c_uint64()
        self._check(self._L.ab_stats_value_hist(self._h, C.byref(pi), gmin, gmax, C.c_void_p(h.ctypes.data),
                                                C.byref(s), C.byref(c)))
        return h, s.value, int(c.value)

    # ---- core/imaging/stf.rs ---------------------------------------------------------------------
    def auto_stf(self, stats: ImageStats, target_bg=0.25, shadow_k=-2.8) -> StfParams:
        s = self._stats_in(stats)
        cfg = AutoStfConfigC(target_bg, shadow_k)
        p = StfParamsC()
        rc = self._L.ab_auto_stf(C.byref(s), C.byref(cfg), C.byref(p))
        if rc != _lib.AB_OK:
            raise AstroBurstError(rc, "ab_auto_stf: null argument")
        return StfParams(p.shadow, p.midtone, p.highlight)

    def apply_stf(self, image, params: StfParams, stats: ImageStats, out=None):
        """apply_stf -> u8 (stf.rs:89-102)."""
        keep = []
        pi = self._plane(image, keep)
        p = StfParamsC(params.shadow, params.midtone, params.highlight)
        s = self._stats_in(stats)
        if _is_torch(image):
            if out is None:
                out = torch.empty((pi.rows, pi.cols), dtype=torch.uint8, device=image.device)
            self._check(self._L.ab_apply_stf_u8(self._h, C.byref(pi), C.byref(p), C.byref(s),
                                                C.c_void_p(out.data_ptr()), 1))
        else:
            if out is None:
                out = np.empty((pi.rows, pi.cols), np.uint8)
            self._check(self._L.ab_apply_stf_u8(self._h, C.byref(pi), C.byref(p), C.byref(s),
                                                C.c_void_p(out.ctypes.data), 0))
        return out

    def apply_stf_f32(self, image, params: StfParams, stats: ImageStats, out=None):
        """apply_stf_f32 (stf.rs:104-120); pass out=image for apply_stf_inplace (:147-155)."""
        keep = []
        pi = self._plane(image, keep)
        if out is None:
            out = self._new_like(image, pi.rows, pi.cols)
        po = self._plane(out, keep) if _is_torch(out) else Plane(C.c_void_p(out.ctypes.data), pi.rows, pi.cols, 0)
        p = StfParamsC(params.shadow, params.midtone, params.highlight)
        s = self._stats_in(stats)
        self._check(self._L.ab_apply_stf_f32(self._h, C.byref(pi), C.byref(p), C.byref(s), C.byref(po)))
        return out

    # ---- core/imaging/calibration_pipeline.rs -------------------------------------------
    def _masters(self, bias, dark, flat, keep):
        m = _lib.CalibrationMastersC()
        for name, x in (("bias", bias), ("dark", dark), ("flat", flat)):
            if x is not None:
                pl = self._plane(x, keep)
                keep.append(pl)
                setattr(m, name, C.pointer(pl))
        return m

    def _planes(self, xs, keep):
        return (Plane * max(len(xs), 1))(*[self._plane(x, keep) for x in xs])

    def calibrate_light(self, light, bias=None, dark=None, flat=None, out=None):
        """calibrate_light (calibration_pipeline.rs:74-118)"""
        keep = []
        pi = self._plane(light, keep)
        if out is None:
            out = self._new_like(light, pi.rows, pi.cols)
        po = self._plane(out, keep)
        m = self._masters(bias, dark, flat, keep)
        self._check(self._L.ab_calibrate_light(self._h, C.byref(pi), C.byref(m), C.byref(po)))
        return out

    def normalize_frames(self, frames):
        """normalize_frames (:309-319) -> new frames"""
        keep = []
        outs = [self._new_like(f, f.shape[0], f.shape[1]) for f in frames]
        self._check(self._L.ab_normalize_frames(self._h, self._planes(frames, keep), len(frames), self._planes(outs, keep)))
        return outs

    def sigma_clipped_mean_stack(self, frames, config: "BatchStackConfig | None" = None):
        """sigma_clipped_mean_stack (:321-378) -> (stacked, rejection_counts)"""
        keep = []
        out = self._new_like(frames[0], frames[0].shape[0], frames[0].shape[1]) if len(frames) else None
        rej = (C.c_uint64 * max(len(frames), 1))()
        cfg = (config or BatchStackConfig())._c()
        po = self._plane(out, keep) if out is not None else Plane()
        self._check(self._L.ab_sigma_clipped_mean_stack(self._h, self._planes(frames, keep), len(frames), C.byref(cfg), C.byref(po), rej))
        return out, list(rej[:len(frames)])

    def run_batch_channel(self, lights, bias=None, dark=None, flat=None, config: "BatchStackConfig | None" = None):
        """one channel of run_batch_pipeline (:157-190), fused -> (master, rejection_counts, mean, stddev)"""
        keep = []
        out = self._new_like(lights[0], lights[0].shape[0], lights[0].shape[1])
        rej = (C.c_uint64 * max(len(lights), 1))()
        st = _lib.BatchChannelStatsC()
        cfg = (config or BatchStackConfig())._c()
        m = self._masters(bias, dark, flat, keep)
        po = self._plane(out, keep)
        self._check(self._L.ab_run_batch_channel(self._h, self._planes(lights, keep), len(lights), C.byref(m), C.byref(cfg), C.byref(po), rej,
                                                 C.byref(st)))
        return out, list(rej[:len(lights)]), st.mean, st.stddev

    def compose_rgb_from_masters(self, r, g, b, l=None):
        """compose_rgb_from_masters (:201-267) -> (h, w, 3) f32"""
        keep = []
        pr, pg, pb = (self._plane(x, keep) for x in (r, g, b))
        pl = self._plane(l, keep) if l is not None else None
        h, w = min(pr.rows, pg.rows, pb.rows), min(pr.cols, pg.cols, pb.cols)
        if _is_torch(r):
            out = torch.empty((h, w, 3), dtype=torch.float32, device=r.device)
            ptr, dev = C.c_void_p(out.data_ptr()), 1
        else:
            out = np.empty((h, w, 3), np.float32)
            ptr, dev = C.c_void_p(out.ctypes.data), 0
        oh, ow = C.c_int64(), C.c_int64()
        self._check(self._L.ab_compose_rgb_from_masters(self._h, C.byref(pr), C.byref(pg), C.byref(pb), C.byref(pl) if pl is not None else None,
                                                        ptr, dev, C.byref(oh), C.byref(ow)))
        assert (oh.value, ow.value) == (h, w)
        return out

    def run_batch_pipeline(self, channels, bias=None, dark=None, flat=None, config: "BatchStackConfig | None" = None):
        """run_batch_pipeline (:120-199); channels = [(label, [lights])] -> BatchPipelineResult"""
        keep = []
        n = len(channels)
        cin = (_lib.BatchChannelInputC * max(n, 1))()
        masters, rejs = [], []
        for c, (label, lights) in enumerate(channels):
            cin[c].label = label.encode()
            planes = self._planes(lights, keep)
            keep.append(planes)
            cin[c].lights = planes if len(lights) else None
            cin[c].n_lights = len(lights)
            rej = (C.c_uint64 * max(len(lights), 1))()
            rejs.append(rej)
            cin[c].rejection_counts = rej
            masters.append(self._new_like(lights[0], lights[0].shape[0], lights[0].shape[1]) if len(lights) else np.zeros((1, 1), np.float32))
        pout = self._planes(masters, keep)
        st = (_lib.BatchChannelStatsC * max(n, 1))()
        cfg = (config or BatchStackConfig())._c()
        m = self._masters(bias, dark, flat, keep)
        find = {}
        for c, (label, _) in enumerate(channels):
            find.setdefault(label.upper(), c)
        rgb, ptr, dev = None, None, 0
        if n and all(k in find for k in "RGB"):
            h = min(masters[find[k]].shape[0] for k in "RGB")
            w = min(masters[find[k]].shape[1] for k in "RGB")
            if _is_torch(masters[0]):
                rgb = torch.empty((h, w, 3), dtype=torch.float32, device=masters[0].device)
                ptr, dev = C.c_void_p(rgb.data_ptr()), 1
            else:
                rgb = np.empty((h, w, 3), np.float32)
                ptr, dev = C.c_void_p(rgb.ctypes.data), 0
        oh, ow = C.c_int64(), C.c_int64()
        self._check(self._L.ab_run_batch_pipeline(self._h, cin, n, C.byref(m), C.byref(cfg), pout, st, ptr, dev, C.byref(oh), C.byref(ow)))
        stats = [BatchChannelStats(channels[c][0], int(st[c].lights_input), list(rejs[c][:len(channels[c][1])]), st[c].mean, st[c].stddev)
                 for c in range(n)]
        return BatchPipelineResult([(channels[c][0], masters[c]) for c in range(n)], rgb, stats, int(dark is not None), int(flat is not None),
                                   int(bias is not None))

    # ---- preview / tile renderers up to the PNG encoder --------------------------------
    def _stf3(self, stf, stats):
        if stf is None:
            return None, None
        return ((StfParamsC * 3)(*[StfParamsC(p.shadow, p.midtone, p.highlight) for p in stf]),
                (_lib.ImageStatsC * 3)(*[self._stats_in(s) for s in stats]) if stats is not None else None)

    def _bytes_out(self, like, shape):
        """u8 output next to the input: a CUDA tensor for device planes, a numpy array for host planes -> (array, ptr, on_device)"""
        if _is_torch(like):
            out = torch.empty(shape, dtype=torch.uint8, device=like.device)
            return out, C.c_void_p(out.data_ptr()), 1
        out = np.empty(shape, np.uint8)
        return out, C.c_void_p(out.ctypes.data), 0

    def preview_dims(self, rows: int, cols: int, max_dim: int):
        ph, pw = C.c_int64(), C.c_int64()
        if self._L.ab_preview_dims(rows, cols, max_dim, C.byref(ph), C.byref(pw)) != _lib.AB_OK:
            raise AstroBurstError(_lib.AB_ERR_INVALID, "ab_preview_dims: bad arguments")
        return ph.value, pw.value

    def render_rgb_preview(self, r, g, b, max_dim: int, stf=None, stats=None):
        """render_rgb_preview / render_rgb (stf None) or render_rgb_preview_with_stf (stf, stats = 3 each)
        (cmd/helpers.rs:204-322) -> (ph, pw, 3) u8"""
        keep = []
        pr, pg, pb = (self._plane(x, keep) for x in (r, g, b))
        ph, pw = self.preview_dims(pr.rows, pr.cols, max_dim) if max_dim > 0 else (1, 1)
        out, ptr, dev = self._bytes_out(r, (ph, pw, 3))
        p, s = self._stf3(stf, stats)
        self._check(self._L.ab_render_rgb_preview(self._h, C.byref(pr), C.byref(pg), C.byref(pb), max_dim, p, s, ptr, dev))
        return out

    def ipc_encode_with_header(self, image, max_dim: int = 0):
        """encode_with_header / encode_with_header_downsampled (infra/ipc.rs:93-148) -> u8 buffer (header + f32 LE pixels)"""
        keep = []
        pi = self._plane(image, keep)
        full = max_dim <= 0 or (pi.rows <= max_dim and pi.cols <= max_dim)
        ph, pw = (pi.rows, pi.cols) if full else self.preview_dims(pi.rows, pi.cols, max_dim)
        out, ptr, dev = self._bytes_out(image, (16 + 4 * ph * pw,))
        n = C.c_size_t(0)
        self._check(self._L.ab_ipc_encode_with_header(self._h, C.byref(pi), max_dim, ptr, dev, C.byref(n)))
        assert n.value == out.shape[0]
        return out

    def tile_compute_num_levels(self, width: int, height: int, tile_size: int) -> int:
        return self._L.ab_tile_compute_num_levels(width, height, tile_size)

    def tile_pyramid_layout(self, rows: int, cols: int, tile_size: int, channels: int):
        lv, n, total = (_lib.TileLevelC * _lib.MAX_TILE_LEVELS)(), C.c_int32(), C.c_size_t()
        if self._L.ab_tile_pyramid_layout(rows, cols, tile_size, channels, lv, C.byref(n), C.byref(total)) != _lib.AB_OK:
            raise AstroBurstError(_lib.AB_ERR_INVALID, "ab_tile_pyramid_layout: bad arguments")
        return [{k: getattr(l, k) for k, _ in l._fields_} for l in lv[:n.value]], total.value

    def tile_downsample_2x(self, image):
        keep = []
        pi = self._plane(image, keep)
        out = self._new_like(image, (pi.rows + 1) // 2, (pi.cols + 1) // 2)
        po = self._plane(out, keep) if _is_torch(out) else Plane(C.c_void_p(out.ctypes.data), out.shape[0], out.shape[1], 0)
        self._check(self._L.ab_tile_downsample_2x(self._h, C.byref(pi), C.byref(po)))
        return out

    def tile_percentile_bounds(self, image, low_pct: float = 0.001, high_pct: float = 0.999):
        keep = []
        pi = self._plane(image, keep)
        lo, hi = C.c_float(), C.c_float()
        self._check(self._L.ab_tile_percentile_bounds(self._h, C.byref(pi), low_pct, high_pct, C.byref(lo), C.byref(hi)))
        return lo.value, hi.value

    def generate_tile_pyramid(self, normalized, tile_size: int = 256):
        """generate_tile_pyramid (infra/render/tiles.rs:180-255) up to the encoder -> (packed tiles u8, levels, (gmin, gmax))"""
        keep = []
        pi = self._plane(normalized, keep)
        levels, total = self.tile_pyramid_layout(pi.rows, pi.cols, tile_size, 1)
        out, ptr, dev = self._bytes_out(normalized, (total,))
        lv, n, lo, hi = (_lib.TileLevelC * _lib.MAX_TILE_LEVELS)(), C.c_int32(), C.c_float(), C.c_float()
        self._check(self._L.ab_generate_tile_pyramid(self._h, C.byref(pi), tile_size, ptr, dev, lv, C.byref(n), C.byref(lo), C.byref(hi)))
        return out, levels, (lo.value, hi.value)

    def generate_tile_pyramid_rgb(self, r, g, b, tile_size: int = 256, stf=None, stats=None):
        """generate_tile_pyramid_rgb / _rgb_stf (tiles.rs:363-481) up to the encoder -> (packed tiles u8, levels)"""
        keep = []
        pr, pg, pb = (self._plane(x, keep) for x in (r, g, b))
        levels, total = self.tile_pyramid_layout(pr.rows, pr.cols, tile_size, 3)
        out, ptr, dev = self._bytes_out(r, (total,))
        lv, n = (_lib.TileLevelC * _lib.MAX_TILE_LEVELS)(), C.c_int32()
        p, s = self._stf3(stf, stats)
        self._check(self._L.ab_generate_tile_pyramid_rgb(self._h, C.byref(pr), C.byref(pg), C.byref(pb), tile_size, p, s, ptr, dev, lv,
                                                         C.byref(n)))
        return out, levels

    # ---- core/analysis/star_detection.rs, core/alignment/affine.rs ------------------------------------
    def estimate_background(self, image, tile_size: int):
        keep = []
        pi = self._plane(image, keep)
        m, s = C.c_double(), C.c_double()
        self._check(self._L.ab_estimate_background(self._h, C.byref(pi), tile_size, C.byref(m), C.byref(s)))
        return m.value, s.value

    def background_tile_stats(self, image, tile_size: int):
        """per-tile sigma_clipped_stats of estimate_background's tiling -> (median[nty, ntx], sigma[nty, ntx], valid[nty, ntx])"""
        keep = []
        pi = self._plane(image, keep)
        step = max(int(tile_size), 16)
        nty, ntx = -(-pi.rows // step), -(-pi.cols // step)
        med, sig = np.zeros(nty * ntx), np.zeros(nty * ntx)
        val = np.zeros(nty * ntx, np.int32)
        n, nx = C.c_size_t(), C.c_size_t()
        self._check(self._L.ab_background_tile_stats(self._h, C.byref(pi), tile_size, med.ctypes.data_as(C.POINTER(C.c_double)),
                                                     sig.ctypes.data_as(C.POINTER(C.c_double)), val.ctypes.data_as(C.POINTER(C.c_int32)),
                                                     nty * ntx, C.byref(n), C.byref(nx)))
        assert (n.value, nx.value) == (nty * ntx, ntx)
        return med.reshape(nty, ntx), sig.reshape(nty, ntx), val.reshape(nty, ntx).astype(bool)

    def detect_stars(self, image, sigma_threshold: float, max_stars: int = 100000):
        """detect_stars(image, sigma) -> (stars, background_median, background_sigma) (star_detection.rs:86-258)"""
        keep = []
        pi = self._plane(image, keep)
        buf = (_lib.DetectedStarC * max_stars)()
        n, tot = C.c_size_t(0), C.c_size_t(0)
        m, s = C.c_double(), C.c_double()
        self._check(self._L.ab_detect_stars(self._h, C.byref(pi), sigma_threshold, buf, max_stars, C.byref(n),
                                            C.byref(tot), C.byref(m), C.byref(s)))
        stars = [DetectedStar(b.x, b.y, b.flux, b.fwhm, b.eccentricity, b.peak, int(b.npix), b.snr)
                 for b in buf[:n.value]]
        return stars, m.value, s.value

    def normalize_for_detection(self, image, out=None):
        return self._unary(self._L.ab_normalize_for_detection, image, out)

    def align_channel_affine(self, reference, target, num_threads: int = 8) -> "AffineAlignResult":
        """align_channel_affine(reference, target) (affine.rs:129-212)"""
        keep = []
        pr, pt = self._plane(reference, keep), self._plane(target, keep)
        res = _lib.AffineAlignResultC()
        self._check(self._L.ab_align_channel_affine(self._h, C.byref(pr), C.byref(pt), num_threads, C.byref(res)))
        return AffineAlignResult(tuple(res.transform), int(res.matched_stars), int(res.inliers), res.residual_px,
                                 AFFINE_METHODS[res.method])

    def register_frames(self, reference, targets, num_threads: int = 8):
        """align_channel_affine(reference, t) for every t in targets, sharing the reference's detection
        (affine.rs:129-212) -> [AffineAlignResult]"""
        keep = []
        pr = self._plane(reference, keep)
        planes = (Plane * max(len(targets), 1))(*[self._plane(t, keep) for t in targets])
        res = (_lib.AffineAlignResultC * max(len(targets), 1))()
        self._check(self._L.ab_register_frames(self._h, C.byref(pr), planes, len(targets), num_threads, res))
        return [AffineAlignResult(tuple(r.transform), int(r.matched_stars), int(r.inliers), r.residual_px, AFFINE_METHODS[r.method])
                for r in res[:len(targets)]]

    def align_pairs_affine(self, reference, targets, outs, num_threads: int = 8):
        """align_pair(reference, t, AlignMethod::Affine) for every t (pair.rs:41-77): estimate + warp_image into outs[i]
        (device planes).  The reference and the targets may be HOST planes (numpy arrays / CPU tensors, pinned for asynchronous
        copies): the library uploads them on its own stream and registers every frame as it lands.  -> [AffineAlignResult]"""
        keep = []
        pr = self._plane(reference, keep)
        planes, n = self._plane_array(targets, keep)
        pouts, n_out = self._plane_array(outs, keep)
        assert n == n_out, "one output plane per target"
        res = (_lib.AffineAlignResultC * max(len(targets), 1))()
        self._check(self._L.ab_align_pairs_affine(self._h, C.byref(pr), planes, n, num_threads, res, pouts))
        return [AffineAlignResult(tuple(r.transform), int(r.matched_stars), int(r.inliers), r.residual_px, AFFINE_METHODS[r.method])
                for r in res[:len(targets)]]

    def analyze_subframes(self, images, config: "SubframeWeightConfig | None" = None, normalize: bool = False):
        """analyze_subframe(image, _, config) for every image (subframe.rs:51-121), frame-parallel -> [SubframeMetrics];
        normalize=True also applies normalize_weights (:148-159)."""
        keep = []
        n = len(images)
        planes = (Plane * max(n, 1))(*[self._plane(im, keep) for im in images])
        res = (_lib.SubframeMetricsC * max(n, 1))()
        cfg = (config or SubframeWeightConfig())._c()
        self._check(self._L.ab_analyze_subframes(self._h, planes, n, C.byref(cfg), res))
        if normalize:
            self._L.ab_normalize_subframe_weights(res, n)
        return [SubframeMetrics(int(r.star_count), r.median_fwhm, r.median_eccentricity, r.median_snr, r.background_median,
                                r.background_sigma, r.noise_ratio, r.weight, bool(r.accepted)) for r in res[:n]]

    def analyze_subframe(self, image, config: "SubframeWeightConfig | None" = None):
        return self.analyze_subframes([image], config)[0]

    def affine_from_stars(self, ref_xy, tgt_xy, rows, cols, num_threads: int = 8):
        r = np.ascontiguousarray(np.asarray(ref_xy, np.float64).reshape(-1, 2))
        t = np.ascontiguousarray(np.asarray(tgt_xy, np.float64).reshape(-1, 2))
        res, found = _lib.AffineAlignResultC(), C.c_int(0)
        rc = self._L.ab_affine_from_stars(r.ctypes.data_as(C.POINTER(C.c_double)), r.shape[0],
                                          t.ctypes.data_as(C.POINTER(C.c_double)), t.shape[0], rows, cols,
                                          num_threads, C.byref(res), C.byref(found))
        if rc != _lib.AB_OK:
            raise AstroBurstError(rc, "ab_affine_from_stars: bad arguments")
        if not found.value:
            return None
        return AffineAlignResult(tuple(res.transform), int(res.matched_stars), int(res.inliers), res.residual_px,
                                 AFFINE_METHODS[res.method])

    # ---- core/alignment/phase_correlation.rs ----------------------------------------------------------
    def phase_correlate(self, reference, target):
        """phase_correlate(reference, target) -> (dx, dy, confidence) (phase_correlation.rs:22-89)."""
        keep = []
        pr, pt = self._plane(reference, keep), self._plane(target, keep)
        res = _lib.PhaseCorrelationResultC()
        self._check(self._L.ab_phase_correlate(self._h, C.byref(pr), C.byref(pt), C.byref(res)))
        return res.dx, res.dy, res.confidence

    def correlate_single(self, a, b, want_surface=False):
        """correlate_single (phase_correlation.rs:105-141), dims <= 512; optionally the correlation surface."""
        keep = []
        pa, pb = self._plane(a, keep), self._plane(b, keep)
        res = _lib.PhaseCorrelationResultC()
        surf = None
        if want_surface:
            fr = 1 << max(0, (pa.rows - 1).bit_length())
            fc = 1 << max(0, (pa.cols - 1).bit_length())
            surf = np.zeros((fr, fc), np.float64)
        self._check(self._L.ab_correlate_single(self._h, C.byref(pa), C.byref(pb), C.byref(res),
                                                C.c_void_p(surf.ctypes.data) if surf is not None else None))
        return (res.dx, res.dy, res.confidence, surf) if want_surface else (res.dx, res.dy, res.confidence)

    # ---- colour / tone / calibration maps -----------------------------------------------------------
    def _out_plane(self, out, keep, rows, cols):
        return self._plane(out, keep) if _is_torch(out) else Plane(C.c_void_p(out.ctypes.data), rows, cols, 0)

    def _unary(self, fn, image, out, *mid):
        keep = []
        pi = self._plane(image, keep)
        if out is None:
            out = self._new_like(image, pi.rows, pi.cols)
        po = self._out_plane(out, keep, pi.rows, pi.cols)
        self._check(fn(self._h, C.byref(pi), *mid, C.byref(po)))
        return out

    def apply_scnr_inplace(self, r, g, b, method="average", amount=1.0, preserve_luminance=False):
        """apply_scnr_inplace(&mut r, &mut g, &mut b, &ScnrConfig) (scnr.rs:18-53); mutates r, g, b."""
        keep = []
        planes = []
        for x in (r, g, b):
            if _is_torch(x):
                planes.append(self._plane(x, keep))
            else:
                assert isinstance(x, np.ndarray) and x.dtype == np.float32 and x.flags.c_contiguous, \
                    "in-place SCNR needs contiguous float32 arrays"
                planes.append(Plane(C.c_void_p(x.ctypes.data), x.shape[0], x.shape[1], 0))
        cfg = _lib.ScnrConfigC(0 if method in ("average", "AverageNeutral", 0) else 1, amount,
                               1 if preserve_luminance else 0)
        self._check(self._L.ab_apply_scnr_inplace(self._h, C.byref(planes[0]), C.byref(planes[1]),
                                                  C.byref(planes[2]), C.byref(cfg)))

    def blend_channels(self, channels, weights, rows, cols):
        """blend_channels(&[&Array2], &[BlendWeight], rows, cols) -> (R, G, B) (channel_blend.rs:13-70).
        weights: iterable of (channel_idx, r_weight, g_weight, b_weight)."""
        keep = []
        chans = (Plane * len(channels))(*[self._plane(c, keep) for c in channels])
        ws = (_lib.BlendWeightC * max(1, len(weights)))(*[_lib.BlendWeightC(int(w[0]), w[1], w[2], w[3])
                                                           for w in weights])
        outs = [self._new_like(channels[0], rows, cols) for _ in range(3)]
        pos = [self._out_plane(o, keep, rows, cols) for o in outs]
        self._check(self._L.ab_blend_channels(self._h, chans, len(channels), ws, len(weights), C.byref(pos[0]),
                                              C.byref(pos[1]), C.byref(pos[2])))
        return tuple(outs)

    def spline_lut_from_points(self, points) -> np.ndarray:
        """SplineLut::from_points (curves.rs:69-95) -> 4096 f32 (host scalar maths in the library)."""
        pts = np.ascontiguousarray(np.asarray(points, dtype=np.float64).reshape(-1, 2))
        lut = np.zeros(4096, np.float32)
        rc = self._L.ab_spline_lut_from_points(pts.ctypes.data_as(C.POINTER(C.c_double)), pts.shape[0],
                                               lut.ctypes.data_as(C.POINTER(C.c_float)))
        if rc != _lib.AB_OK:
            raise AstroBurstError(rc, "ab_spline_lut_from_points: bad arguments")
        return lut

    def apply_curve(self, image, lut, out=None):
        lut = np.ascontiguousarray(lut, dtype=np.float32)
        assert lut.size == 4096
        return self._unary(self._L.ab_apply_curve, image, out, lut.ctypes.data_as(C.POINTER(C.c_float)))

    def apply_levels(self, image, black=0.0, gamma=1.0, white=1.0, out=None):
        p = _lib.LevelsParamsC(black, gamma, white)
        return self._unary(self._L.ab_apply_levels, image, out, C.byref(p))

    def arcsinh_stretch_with_stats(self, image, dmin, dmax, factor, gamma=1.0, out=None):
        return self._unary(self._L.ab_arcsinh_stretch_with_stats, image, out, C.c_float(dmin), C.c_float(dmax),
                           C.c_float(factor), C.c_float(gamma))

    def scale(self, image, factor, out=None):
        return self._unary(self._L.ab_scale, image, out, C.c_float(factor))

    def luminance(self, r, g, b, out=None):
        keep = []
        pr, pg, pb = (self._plane(x, keep) for x in (r, g, b))
        if out is None:
            out = self._new_like(r, pr.rows, pr.cols)
        po = self._out_plane(out, keep, pr.rows, pr.cols)
        self._check(self._L.ab_luminance(self._h, C.byref(pr), C.byref(pg), C.byref(pb), C.byref(po)))
        return out

    def calibrate_image(self, raw, master_bias=None, master_dark=None, master_flat=None, dark_exposure_ratio=1.0,
                        out=None):
        """calibrate_image(raw, &CalibrationConfig) (calibration.rs:47-82)."""
        keep = []
        praw = self._plane(raw, keep)
        opt = [C.byref(self._plane(x, keep)) if x is not None else None for x in (master_bias, master_dark, master_flat)]
        if out is None:
            out = self._new_like(raw, praw.rows, praw.cols)
        po = self._out_plane(out, keep, praw.rows, praw.cols)
        self._check(self._L.ab_calibrate_image(self._h, C.byref(praw), opt[0], opt[1], opt[2],
                                               C.c_float(dark_exposure_ratio), C.byref(po)))
        return out

    def median_combine(self, frames, out=None):
        """median_combine_row_major (calibration.rs:84-125): per-pixel upper median of the finite samples."""
        keep = []
        planes = (Plane * len(frames))(*[self._plane(f, keep) for f in frames])
        rows, cols = planes[0].rows, planes[0].cols
        if out is None:
            out = self._new_like(frames[0], rows, cols)
        po = self._out_plane(out, keep, rows, cols)
        self._check(self._L.ab_median_combine(self._h, planes, len(frames), C.byref(po)))
        return out

    # ---- a12 background extraction ------------------------------------------------------------------
    def extract_background(self, image, grid_size=8, poly_degree=3, sigma_clip=2.5, iterations=3, mode="subtract",
                           want_model=True):
        """extract_background (background.rs:55-116) -> BackgroundResult(model, corrected, sample_count, rms_residual).

        mode: "subtract" | "divide" (BackgroundMode).  Raises AstroBurstError with the reference's message when the
        image is too small for the grid, too few samples survive, or the fit is singular."""
        keep = []
        pi = self._plane(image, keep)
        rows, cols = pi.rows, pi.cols
        corrected = self._new_like(image, rows, cols)
        pc = self._out_plane(corrected, keep, rows, cols)
        model, pm = None, None
        if want_model:
            model = self._new_like(image, rows, cols)
            pm = C.byref(self._out_plane(model, keep, rows, cols))
        cfg = _lib.BackgroundConfigC(grid_size, poly_degree, sigma_clip, iterations,
                                     {"subtract": 0, "divide": 1}[mode] if isinstance(mode, str) else int(mode))
        info = _lib.BackgroundInfoC()
        self._check(self._L.ab_extract_background(self._h, C.byref(pi), C.byref(cfg), pm, C.byref(pc), C.byref(info)))
        return BackgroundResult(model, corrected, int(info.sample_count), float(info.rms_residual),
                                np.array(info.coeffs[:], dtype=np.float64))

    # ---- a17 RGB composition -------------------------------------------------------------------------
    def resample_image(self, image, target_rows: int, target_cols: int, out=None):
        """resample_image(image, target_rows, target_cols) (resample.rs:25-61)"""
        keep = []
        pi = self._plane(image, keep)
        if out is None:
            out = self._new_like(image, target_rows, target_cols)
        po = self._out_plane(out, keep, target_rows, target_cols)
        self._check(self._L.ab_resample_image(self._h, C.byref(pi), C.byref(po)))
        return out

    def select_wb_reference(self, sr: ImageStats, sg: ImageStats, sb: ImageStats):
        """select_wb_reference (white_balance.rs:3-20) -> (wb_r, wb_g, wb_b)"""
        out = (C.c_double * 3)()
        rc = self._L.ab_select_wb_reference(*[C.byref(self._stats_in(s)) for s in (sr, sg, sb)], out)
        if rc != _lib.AB_OK:
            raise AstroBurstError(rc, "select_wb_reference: null argument")
        return tuple(out)

    def process_rgb(self, r, g, b, white_balance="auto", auto_stretch=True, stf=(None, None, None), linked_stf=False,
                    align=True, align_method="phase_correlation", scnr=None, num_threads=8, want_pre_stretch=True
                    ) -> ProcessedRgb:
        """process_rgb(r?, g?, b?, &RgbComposeConfig) (rgb.rs:209-323).  Absent channels are None.
        white_balance: "auto" | "none" | (r, g, b) manual multipliers; stf: per-channel StfParams or None (used when
        auto_stretch is False); scnr: None or dict(method=, amount=, preserve_luminance=)."""
        keep = []
        chans = [None if x is None else self._plane(x, keep) for x in (r, g, b)]
        present = [p for p in chans if p is not None]
        first = next((x for x in (r, g, b) if x is not None), None)
        rows = max((p.rows for p in present), default=0)
        cols = max((p.cols for p in present), default=0)
        cfg = _lib.RgbComposeConfigC()
        if isinstance(white_balance, str):
            cfg.white_balance = {"auto": 0, "none": 2}[white_balance]
        else:
            cfg.white_balance = 1
            cfg.wb_manual[:] = [float(v) for v in white_balance]
        cfg.auto_stretch, cfg.linked_stf = int(bool(auto_stretch)), int(bool(linked_stf))
        for c, p in enumerate(stf):
            if p is not None:
                cfg.has_stf[c] = 1
                cfg.stf[c] = StfParamsC(p.shadow, p.midtone, p.highlight)
        cfg.align = int(bool(align))
        cfg.align_method = {"phase_correlation": 0, "affine": 1}[align_method]
        if scnr is not None:
            cfg.has_scnr = 1
            m = scnr.get("method", "average")
            cfg.scnr = _lib.ScnrConfigC(0 if m in ("average", "AverageNeutral", 0) else 1, scnr.get("amount", 1.0),
                                        int(bool(scnr.get("preserve_luminance", False))))
        cfg.num_threads = num_threads
        if first is None:
            first = np.empty((0, 0), np.float32)
        outs = [self._new_like(first, rows, cols) for _ in range(3)]
        pos = [self._out_plane(o, keep, rows, cols) for o in outs]
        pres = [self._new_like(first, rows, cols) if want_pre_stretch else None for _ in range(3)]
        pps = [None if p is None else C.byref(self._out_plane(p, keep, rows, cols)) for p in pres]
        info = _lib.ProcessedRgbInfoC()
        self._check(self._L.ab_process_rgb(self._h, *[None if p is None else C.byref(p) for p in chans], C.byref(cfg),
                                           *[C.byref(p) for p in pos], *pps, C.byref(info)))
        return ProcessedRgb(outs[0], outs[1], outs[2], int(info.rows), int(info.cols),
                            tuple(StfParams(p.shadow, p.midtone, p.highlight) for p in info.stf),
                            tuple(tuple(row) for row in info.chan_stats), tuple(info.offset_g), tuple(info.offset_b),
                            bool(info.scnr_applied), bool(info.resampled), tuple(pres),
                            tuple(self._stats_out(s) for s in info.stats_wb))

    # ---- FITS pixel codecs ----------------------------------------------------------
    _BPP = {8: 1, 16: 2, 32: 4, -32: 4, -64: 8}

    def fits_decode_pixels(self, data, rows: int, cols: int, bitpix: int, bscale=1.0, bzero=0.0, out=None):
        """decode_pixels (reader.rs:42-101): data = the big-endian data unit (bytes / numpy uint8 on the host, or a torch
        uint8 CUDA tensor) -> f32 plane."""
        if bitpix not in self._BPP:
            raise AstroBurstError(_lib.AB_ERR_INVALID, f"unsupported BITPIX {bitpix}")
        keep = []
        if _is_torch(data):
            assert data.is_cuda and data.dtype == torch.uint8 and data.is_contiguous()
            self.use_torch_stream()
            ptr, nbytes, on_dev = C.c_void_p(data.data_ptr()), data.numel(), 1
            keep.append(data)
            if out is None:
                out = torch.empty((rows, cols), dtype=torch.float32, device=data.device)
        else:
            buf = np.frombuffer(data, dtype=np.uint8) if isinstance(data, (bytes, bytearray, memoryview)) else np.ascontiguousarray(data, np.uint8)
            keep.append(buf)
            ptr, nbytes, on_dev = C.c_void_p(buf.ctypes.data), buf.size, 0
            if out is None:
                out = np.empty((rows, cols), np.float32)
        po = self._out_plane(out, keep, rows, cols)
        self._check(self._L.ab_fits_decode_pixels(self._h, ptr, nbytes, on_dev, bitpix, bscale, bzero, C.byref(po)))
        return out

    def fits_compute_bzero_bscale(self, image):
        """compute_bzero_bscale (writer.rs:143-159) -> (bzero, bscale)"""
        keep = []
        pi = self._plane(image, keep)
        bz, bs = C.c_double(), C.c_double()
        self._check(self._L.ab_fits_compute_bzero_bscale(self._h, C.byref(pi), C.byref(bz), C.byref(bs)))
        return bz.value, bs.value

    def fits_encode_pixels(self, image, bitpix: int, bzero=0.0, bscale=1.0) -> np.ndarray:
        """write_*_slice_as_be (writer.rs:82-135) -> the big-endian data unit as a host uint8 array"""
        keep = []
        pi = self._plane(image, keep)
        if bitpix not in (-32, 16, -64):
            raise AstroBurstError(_lib.AB_ERR_INVALID, f"the writer supports BITPIX -32, 16 and -64 (got {bitpix})")
        out = np.empty(pi.rows * pi.cols * self._BPP[bitpix], np.uint8)
        self._check(self._L.ab_fits_encode_pixels(self._h, C.byref(pi), bitpix, bzero, bscale, C.c_void_p(out.ctypes.data), 0))
        return out

    def stack_sigma_clip_raw(self, raw_planes, rows: int, cols: int, bitpix: int, bscale=1.0, bzero=0.0, sigma_low=3.0, sigma_high=3.0,
                             max_iterations=5, out=None, want_rejected=True):
        """ab_stack_sigma_clip over raw big-endian data units (torch uint8 CUDA tensors): decode fused into the stack."""
        need = rows * cols * (abs(int(bitpix)) // 8)
        for t in raw_planes:
            assert _is_torch(t) and t.is_cuda and t.dtype == torch.uint8 and t.is_contiguous()
            if bitpix in (-32, 16) and t.numel() < need:   # a truncated data unit would make the kernel read past the tensor
                raise AstroBurstError(_lib.AB_ERR_INVALID, f"raw plane holds {t.numel()} bytes, {rows}x{cols} BITPIX {bitpix} needs {need}")
        self.use_torch_stream()
        ptrs = (C.c_void_p * len(raw_planes))(*[t.data_ptr() for t in raw_planes])
        if out is None:
            out = torch.empty((rows, cols), dtype=torch.float32, device=raw_planes[0].device)
        keep = []
        po = self._out_plane(out, keep, rows, cols)
        cfg = StackConfig(sigma_low, sigma_high, max_iterations, 0)
        rej = C.c_uint64(0)
        self._check(self._L.ab_stack_sigma_clip_raw(self._h, ptrs, len(raw_planes), bitpix, bscale, bzero, C.byref(cfg), C.byref(po),
                                                    C.byref(rej) if want_rejected else None))
        return (out, int(rej.value)) if want_rejected else out

    # ---- caller-side helpers of a17 / a20 ----------------------------------------------------------------
    def apply_lrgb(self, l, r, g, b, lightness_weight=1.0, chrominance_weight=1.0):
        """apply_lrgb(l, &mut r, &mut g, &mut b, lw, cw) (lrgb.rs:4-45); mutates r, g, b."""
        keep = []
        pl = self._plane(l, keep)
        pr, pg, pb = (self._out_plane(x, keep, x.shape[0], x.shape[1]) for x in (r, g, b))
        self._check(self._L.ab_apply_lrgb(self._h, C.byref(pl), C.byref(pr), C.byref(pg), C.byref(pb), lightness_weight,
                                          chrominance_weight))

    def synthesize_luminance(self, r, g, b, out=None):
        """synthesize_luminance (lrgb.rs:47-64 / spcc.rs:185-196): no finite guard."""
        keep = []
        pr, pg, pb = (self._plane(x, keep) for x in (r, g, b))
        if out is None:
            out = self._new_like(r, pr.rows, pr.cols)
        po = self._out_plane(out, keep, pr.rows, pr.cols)
        self._check(self._L.ab_synthesize_luminance(self._h, C.byref(pr), C.byref(pg), C.byref(pb), C.byref(po)))
        return out

    def compute_linked_stf(self, sr: ImageStats, sg: ImageStats, sb: ImageStats, target_bg=0.25, shadow_k=-2.8):
        """compute_linked_stf_with_stats (cmd/helpers.rs:185-202) -> (StfParams, combined ImageStats)"""
        cfg = AutoStfConfigC(target_bg, shadow_k)
        stf, comb = StfParamsC(), ImageStatsC()
        rc = self._L.ab_compute_linked_stf(*[C.byref(self._stats_in(s)) for s in (sr, sg, sb)], C.byref(cfg), C.byref(stf),
                                           C.byref(comb))
        if rc != _lib.AB_OK:
            raise AstroBurstError(rc, "compute_linked_stf: null argument")
        return StfParams(stf.shadow, stf.midtone, stf.highlight), self._stats_out(comb)

    def calibrate_channel(self, orig, factor: float, orig_stats: ImageStats):
        """calibrate_channel (cmd/compose/color.rs:21-49) -> (scaled plane, ImageStats)"""
        keep = []
        pi = self._plane(orig, keep)
        out = self._new_like(orig, pi.rows, pi.cols)
        po = self._out_plane(out, keep, pi.rows, pi.cols)
        st = ImageStatsC()
        self._check(self._L.ab_calibrate_channel(self._h, C.byref(pi), factor, C.byref(self._stats_in(orig_stats)),
                                                 C.byref(po), C.byref(st)))
        return out, self._stats_out(st)

    def create_master(self, kind: str, frames, master_bias=None, master_dark=None):
        """create_master_bias / _dark / _flat on in-memory frames (calibration.rs:127-255); kind in bias|dark|flat."""
        keep = []
        k = {"bias": 0, "dark": 1, "flat": 2}[kind]
        if len(frames) == 0:
            self._check(self._L.ab_create_master(self._h, k, None, 0, None, None, C.byref(Plane(None, 0, 0, 0))))
        planes = (Plane * len(frames))(*[self._plane(f, keep) for f in frames])
        rows, cols = planes[0].rows, planes[0].cols
        out = self._new_like(frames[0], rows, cols)
        po = self._out_plane(out, keep, rows, cols)
        pb = None if master_bias is None else C.byref(self._plane(master_bias, keep))
        pd = None if master_dark is None else C.byref(self._plane(master_dark, keep))
        self._check(self._L.ab_create_master(self._h, k, planes, len(frames), pb, pd, C.byref(po)))
        return out

    # ---- a18 spectrophotometric colour calibration -------------------------------------------------------
    @staticmethod
    def _spcc_cfg(min_snr, max_stars, saturation_limit, white_reference):
        cfg = _lib.SpccConfigC(min_snr, max_stars, saturation_limit, 0, (C.c_double * 3)(0, 0, 0))
        if isinstance(white_reference, str):
            cfg.white_reference = WHITE_REFERENCES[white_reference]
            name = {"average_spiral": "Average Spiral Galaxy", "g2v": "G2V (Solar)", "photopic": "Photopic (Human Eye)"}[white_reference]
        else:
            cfg.white_reference = 3
            cfg.custom[:] = [float(v) for v in white_reference]
            name = "Custom ({:.2f},{:.2f},{:.2f})".format(*cfg.custom)
        return cfg, name

    def spcc_calibrate_rgb(self, r, g, b, pixel_scale_arcsec: float, min_snr=20.0, max_stars=200, saturation_limit=0.90,
                           white_reference="average_spiral", detection=None) -> SpccResult:
        """spcc_calibrate_rgb (spcc.rs:73-183), built-in Bp-Rp catalogue; pixel_scale_arcsec = WCS pixel scale.
        detection=(stars, lum_max) runs the part after detect_stars on a given detection (:90-183)."""
        keep = []
        ps = [self._plane(x, keep) for x in (r, g, b)]
        cfg, name = self._spcc_cfg(min_snr, max_stars, saturation_limit, white_reference)
        res = _lib.SpccResultC()
        if detection is None:
            rc = self._L.ab_spcc_calibrate_rgb(self._h, *[C.byref(p) for p in ps], pixel_scale_arcsec, C.byref(cfg), C.byref(res))
        else:
            stars, lum_max = detection
            buf = (_lib.DetectedStarC * max(len(stars), 1))()
            for d, s in zip(buf, stars):
                d.x, d.y, d.flux, d.fwhm, d.eccentricity, d.peak, d.snr, d.npix = (s.x, s.y, s.flux, s.fwhm, s.eccentricity,
                                                                                   s.peak, s.snr, s.npix)
            rc = self._L.ab_spcc_from_detection(self._h, *[C.byref(p) for p in ps], buf, len(stars), lum_max,
                                                pixel_scale_arcsec, C.byref(cfg), C.byref(res))
        self._check(rc)
        return SpccResult(res.r_factor, res.g_factor, res.b_factor, int(res.stars_matched), int(res.stars_total),
                          res.avg_color_index, name)

    def spcc_white_reference_rgb(self, white_reference="average_spiral"):
        cfg, _ = self._spcc_cfg(20.0, 200, 0.9, white_reference)
        out = (C.c_double * 3)()
        rc = self._L.ab_spcc_white_reference_rgb(cfg.white_reference, cfg.custom, out)
        if rc != _lib.AB_OK:
            raise AstroBurstError(rc, "spcc_white_reference_rgb: invalid argument")
        return tuple(out)

    # ---- a13 star mask + masked stretch ---------------------------------------------------------------
    @staticmethod
    def _mask_cfg(growth_factor, softness, detection_sigma, min_fwhm, max_fwhm, luminance_protect, luminance_ceiling):
        return _lib.StarMaskConfigC(growth_factor, softness, detection_sigma, min_fwhm, max_fwhm, int(bool(luminance_protect)),
                                    luminance_ceiling)

    def generate_star_mask(self, image, growth_factor=2.5, softness=4.0, detection_sigma=5.0, min_fwhm=1.5, max_fwhm=30.0,
                           luminance_protect=False, luminance_ceiling=0.85, stars=None) -> StarMaskResult:
        """generate_star_mask (star_mask.rs:38-44); with stars=[DetectedStar | (x, y, fwhm)] it is
        generate_star_mask_from_detection (:46-138) on that detection."""
        keep = []
        pi = self._plane(image, keep)
        mask = self._new_like(image, pi.rows, pi.cols)
        pm = self._out_plane(mask, keep, pi.rows, pi.cols)
        cfg = self._mask_cfg(growth_factor, softness, detection_sigma, min_fwhm, max_fwhm, luminance_protect, luminance_ceiling)
        info = _lib.StarMaskInfoC()
        if stars is None:
            self._check(self._L.ab_generate_star_mask(self._h, C.byref(pi), C.byref(cfg), C.byref(pm), C.byref(info)))
        else:
            buf = (_lib.DetectedStarC * max(len(stars), 1))()
            for b, s in zip(buf, stars):
                b.x, b.y, b.fwhm = (s.x, s.y, s.fwhm) if hasattr(s, "fwhm") else s
            self._check(self._L.ab_generate_star_mask_from_stars(self._h, C.byref(pi), buf, len(stars), C.byref(cfg),
                                                                 C.byref(pm), C.byref(info)))
        return StarMaskResult(mask, int(info.stars_masked), float(info.coverage_fraction))

    @staticmethod
    def _ms_cfg(iterations, target_background, mask_growth, mask_softness, luminance_protect, luminance_ceiling,
                protection_amount, convergence_threshold):
        return _lib.MaskedStretchConfigC(iterations, target_background, mask_growth, mask_softness, int(bool(luminance_protect)),
                                         luminance_ceiling, protection_amount, convergence_threshold)

    @staticmethod
    def _ms_result(image, r):
        return MaskedStretchResult(image, int(r.iterations_run), float(r.final_background), int(r.stars_masked),
                                   float(r.mask_coverage), bool(r.converged))

    def masked_stretch(self, image, iterations=10, target_background=0.25, mask_growth=2.5, mask_softness=4.0,
                       luminance_protect=True, luminance_ceiling=0.85, protection_amount=0.85, convergence_threshold=1e-5,
                       mask: "StarMaskResult | None" = None) -> MaskedStretchResult:
        """masked_stretch (masked_stretch.rs:44-58); with mask=StarMaskResult it is masked_stretch_with_mask (:60-118)."""
        keep = []
        pi = self._plane(image, keep)
        out = self._new_like(image, pi.rows, pi.cols)
        po = self._out_plane(out, keep, pi.rows, pi.cols)
        cfg = self._ms_cfg(iterations, target_background, mask_growth, mask_softness, luminance_protect, luminance_ceiling,
                           protection_amount, convergence_threshold)
        res = _lib.MaskedStretchResultC()
        if mask is None:
            self._check(self._L.ab_masked_stretch(self._h, C.byref(pi), C.byref(cfg), C.byref(po), C.byref(res)))
        else:
            pm = self._plane(mask.mask, keep)
            mi = _lib.StarMaskInfoC(mask.stars_masked, mask.coverage_fraction)
            self._check(self._L.ab_masked_stretch_with_mask(self._h, C.byref(pi), C.byref(pm), C.byref(mi), C.byref(cfg),
                                                            C.byref(po), C.byref(res)))
        return self._ms_result(out, res)

    def masked_stretch_rgb_shared(self, r, g, b, iterations=10, target_background=0.25, mask_growth=2.5, mask_softness=4.0,
                                  luminance_protect=True, luminance_ceiling=0.85, protection_amount=0.85,
                                  convergence_threshold=1e-5):
        """masked_stretch_rgb_shared (masked_stretch.rs:155-193) -> (res_r, res_g, res_b, shared StarMask scalars)."""
        keep = []
        ps = [self._plane(x, keep) for x in (r, g, b)]
        outs = [self._new_like(x, p.rows, p.cols) for x, p in zip((r, g, b), ps)]
        pos = [self._out_plane(o, keep, p.rows, p.cols) for o, p in zip(outs, ps)]
        cfg = self._ms_cfg(iterations, target_background, mask_growth, mask_softness, luminance_protect, luminance_ceiling,
                           protection_amount, convergence_threshold)
        res = (_lib.MaskedStretchResultC * 3)()
        shared = _lib.StarMaskInfoC()
        self._check(self._L.ab_masked_stretch_rgb_shared(self._h, *[C.byref(p) for p in ps], C.byref(cfg),
                                                         *[C.byref(p) for p in pos], res, C.byref(shared)))
        return (*[self._ms_result(o, x) for o, x in zip(outs, res)],
                StarMaskResult(None, int(shared.stars_masked), float(shared.coverage_fraction)))

    # ---- bench support -----------------------------------------------------------------------------
    def bench_copy(self, src, dst):
        self._check(self._L.ab_bench_copy(self._h, C.c_void_p(src.data_ptr()), C.c_void_p(dst.data_ptr()), src.numel()))
