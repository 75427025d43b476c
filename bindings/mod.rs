//! Safe wrappers over `sys` (bindings/sys.rs, generated from include/astroburst_hip.h) for the AstroBurst backend:
//! drop-ins for the `core::*` functions the Tauri commands call.  Place this directory at `src-tauri/src/hip/`.
//!
//! NOT compiled in the repository that ships it (that image has no Rust toolchain); `sys.rs` is checked field by field
//! against the C header and this file is checked to wrap every entry point of the header (tests/test_abi_cpu.py); the rest is
//! reviewed source.  Every function keeps the argument meaning, the result type and the error strings of the `core::*`
//! function it replaces (cited per function) and takes the calling thread's `&Hip` first, so a command body changes by one
//! path (`core::imaging::curves::apply_curve(..)` -> `hip::apply_curve(h, ..)`); INTEGRATION.md holds the command hunks.
//!
//! Host or device: every plane argument is `&impl PlaneSrc` -- an `Array2<f32>` (staged through the library's pinned buffer,
//! one PCIe crossing each way) or a `DevicePlane` already in HBM -- and every `*_into` variant writes an `impl PlaneDst`.
//! `device::DEVICE_CACHE` mirrors the pinned keys of GLOBAL_IMAGE_CACHE so that consecutive commands keep their planes in HBM.
//!
//! Threading: one `Hip` per blocking thread (`blocking_cmd!`, cmd/common.rs:345-352) -- an `ab_ctx` owns a stream and
//! scratch and is not shared.  The library never unwinds into Rust (every entry point is a C++ function-try-block) and
//! never aborts; failures are status codes + `ab_last_error`.
#![allow(dead_code)]
pub mod device;
pub mod sys;

use anyhow::{anyhow, bail, Result};
use ndarray::{Array2, Array3};
use std::cell::RefCell;
use std::ffi::{c_void, CStr, CString};
use std::os::raw::c_char;

pub use device::{DevicePlane, PlaneDst, PlaneSrc, DEVICE_CACHE};

use crate::core::alignment::affine::{AffineAlignMethod, AffineAlignResult, AffineTransform};
use crate::core::alignment::pair::AlignPairResult;
use crate::core::alignment::phase_correlation::PhaseCorrelationResult;
use crate::core::analysis::star_detection::{DetectedStar, DetectionResult};
use crate::core::analysis::subframe::{SubframeMetrics, SubframeWeightConfig};
use crate::core::astrometry::spcc::{SpccConfig, SpccResult, WhiteReference};
use crate::core::compose::channel_blend::BlendWeight;
use crate::core::compose::rgb::ProcessedRgb;
use crate::core::imaging::background::{BackgroundConfig, BackgroundResult};
use crate::core::imaging::calibration_pipeline::{
    BatchChannelStats, BatchPipelineConfig, BatchPipelineResult, BatchPipelineStats, BatchStackConfig, CalibrationMasters, ChannelInput,
};
use crate::core::imaging::curves::LevelsParams;
use crate::core::imaging::masked_stretch::{MaskedStretchConfig, MaskedStretchResult, MaskedStretchRgbResult};
use crate::core::imaging::star_mask::{StarMaskConfig, StarMaskResult};
use crate::core::stacking::calibration::CalibrationConfig;
use crate::infra::progress::ProgressHandle;
use crate::types::compose::{AlignMethod, ChannelStats, DimensionHarmonize, RgbComposeConfig, WhiteBalance};
use crate::types::image::{AutoStfConfig, ImageStats, ScnrConfig, ScnrMethod, StfParams};
use crate::types::stacking::{StackConfig, StackResult};

pub struct Hip {
    pub(crate) ctx: *mut sys::ab_ctx,
}
unsafe impl Send for Hip {}
// `&Hip` crosses into the scoped rank threads of `on_all_gpus`; each rank only ever touches its OWN context there
unsafe impl Sync for Hip {}

thread_local! { static HIP: RefCell<Option<Hip>> = RefCell::new(None); }

/// the calling thread's context on device 0 (created on first use)
pub fn with_hip<T>(f: impl FnOnce(&Hip) -> Result<T>) -> Result<T> {
    HIP.with(|slot| {
        let mut slot = slot.borrow_mut();
        if slot.is_none() {
            *slot = Some(Hip::new(0)?);
        }
        f(slot.as_ref().unwrap())
    })
}

impl Hip {
    pub fn new(device: i32) -> Result<Self> {
        let mut ctx = std::ptr::null_mut();
        match unsafe { sys::ab_ctx_create(device, &mut ctx) } {
            sys::AB_OK => Ok(Self { ctx }),
            sys::AB_ERR_NO_DEVICE => bail!("no gfx950 (MI355X) device"),
            rc => bail!("ab_ctx_create failed ({rc})"),
        }
    }

    pub(crate) fn check(&self, rc: i32) -> Result<()> {
        if rc == sys::AB_OK {
            return Ok(());
        }
        let msg = unsafe { CStr::from_ptr(sys::ab_last_error(self.ctx)) }.to_string_lossy().into_owned();
        if rc == sys::AB_ERR_CANCELLED {
            return Err(crate::types::error::AppError::Cancelled.into()); // background.rs:80-82
        }
        Err(anyhow!(msg)) // the reference's own strings, e.g. "No images to stack"
    }

    /// (name, compute units, HBM bytes)
    pub fn device_info(&self) -> Result<(String, i32, u64)> {
        let mut name = [0 as c_char; 128];
        let (mut cus, mut hbm) = (0i32, 0u64);
        self.check(unsafe { sys::ab_device_info(self.ctx, name.as_mut_ptr(), name.len(), &mut cus, &mut hbm) })?;
        Ok((unsafe { CStr::from_ptr(name.as_ptr()) }.to_string_lossy().into_owned(), cus, hbm))
    }

    /// release the device memory the context has grown for its calls (workspaces, scratch, the staging area of host frames);
    /// the context stays usable -- for a long-lived command thread after a large batch
    pub fn trim(&self) -> Result<()> {
        self.check(unsafe { sys::ab_ctx_trim(self.ctx) })
    }

    /// how often a fast path handed over to its exact fallback (same results, more time) since the context was created or last
    /// reset, indexed by the `sys::AB_FB_*` constants (frames redone in full, crowded tiles, short / close selections, declined
    /// background tiles, aborted resident statistics) -- for the host's log: a session whose registration suddenly takes twice as
    /// long shows up here, not in the pixels
    pub fn fallback_counts(&self, reset: bool) -> Result<[u64; sys::AB_FB_COUNT as usize]> {
        let mut out = [0u64; sys::AB_FB_COUNT as usize];
        self.check(unsafe { sys::ab_ctx_fallback_counts(self.ctx, out.as_mut_ptr(), out.len(), reset as i32) })?;
        Ok(out)
    }

    pub fn synchronize(&self) -> Result<()> {
        self.check(unsafe { sys::ab_ctx_synchronize(self.ctx) })
    }
    /// run on a HIP stream the host owns (e.g. one shared with another library); `reset_stream` returns to the context's own
    pub fn set_stream(&self, hip_stream: *mut c_void) -> Result<()> {
        self.check(unsafe { sys::ab_ctx_set_stream(self.ctx, hip_stream) })
    }
    pub fn reset_stream(&self) -> Result<()> {
        self.check(unsafe { sys::ab_ctx_reset_stream(self.ctx) })
    }
    pub fn stream(&self) -> *mut c_void {
        unsafe { sys::ab_ctx_get_stream(self.ctx) }
    }
    /// duration of the stack kernels of the last stack call on this context (HIP events around the launches)
    pub fn stack_last_kernel_ms(&self) -> Result<f32> {
        let mut ms = 0f32;
        self.check(unsafe { sys::ab_stack_last_kernel_ms(self.ctx, &mut ms) })?;
        Ok(ms)
    }

    /// Forward the library's stage ticks to a `ProgressHandle` (infra/progress.rs:39-74) for the duration of `f`, and the
    /// handle's cancel flag to the library: the reference polls `p.is_cancelled()` between stages (background.rs:80-91); here
    /// every tick -- the library ticks at those same stage boundaries -- forwards a pending cancel, and the library's own
    /// check at the next boundary returns AB_ERR_CANCELLED.
    pub fn with_progress<T>(&self, progress: Option<&ProgressHandle>, f: impl FnOnce() -> Result<T>) -> Result<T> {
        struct User<'a> {
            ctx: *mut sys::ab_ctx,
            p: &'a ProgressHandle,
        }
        unsafe extern "C" fn tick(stage: *const c_char, _cur: u64, total: u64, user: *mut c_void) {
            let u = &*(user as *const User);
            if u.p.is_cancelled() {
                sys::ab_ctx_request_cancel(u.ctx);
            }
            u.p.set_total(total);
            u.p.tick_with_stage(&CStr::from_ptr(stage).to_string_lossy());
        }
        let user = progress.map(|p| User { ctx: self.ctx, p });
        if let Some(u) = user.as_ref() {
            if u.p.is_cancelled() {
                unsafe { sys::ab_ctx_request_cancel(self.ctx) };
            }
            unsafe { sys::ab_ctx_set_progress_cb(self.ctx, Some(tick), u as *const User as *mut c_void) };
        }
        let r = f();
        unsafe {
            sys::ab_ctx_set_progress_cb(self.ctx, None, std::ptr::null_mut());
            sys::ab_ctx_clear_cancel(self.ctx);
        }
        r
    }
}

impl Drop for Hip {
    fn drop(&mut self) {
        unsafe { sys::ab_ctx_destroy(self.ctx) }
    }
}

/// "astroburst-hip x.y (gfx950)"
pub fn version() -> String {
    unsafe { CStr::from_ptr(sys::ab_version()) }.to_string_lossy().into_owned()
}
/// BatchStackConfig::default() / SubframeWeightConfig::default() as the LIBRARY holds them (calibration_pipeline.rs:27-37,
/// subframe.rs:36-49): a start-up assertion that both sides agree costs nothing and catches a drifted default
pub fn library_defaults_match() -> bool {
    let mut b: sys::ab_batch_stack_config = unsafe { std::mem::zeroed() };
    let mut w: sys::ab_subframe_weight_config = unsafe { std::mem::zeroed() };
    unsafe {
        sys::ab_batch_stack_config_default(&mut b);
        sys::ab_subframe_weight_config_default(&mut w);
    }
    let (db, dw) = (BatchStackConfig::default(), SubframeWeightConfig::default());
    b.sigma_low == db.sigma_low && b.sigma_high == db.sigma_high && b.max_iterations == db.max_iterations as u64 && (b.normalize_before_stack != 0) == db.normalize_before_stack
        && w.fwhm_weight == dw.fwhm_weight && w.eccentricity_weight == dw.eccentricity_weight && w.snr_weight == dw.snr_weight && w.noise_weight == dw.noise_weight
        && w.max_fwhm == dw.max_fwhm && w.max_eccentricity == dw.max_eccentricity && w.min_snr == dw.min_snr && w.min_stars == dw.min_stars as u64
}

// ---- small conversions -------------------------------------------------------------------------------------------------------
fn stats_to_sys(st: &ImageStats) -> sys::ab_image_stats {
    sys::ab_image_stats { min: st.min, max: st.max, median: st.median, mad: st.mad, sigma: st.sigma, mean: st.mean, valid_count: st.valid_count }
}
fn stats_from_sys(st: &sys::ab_image_stats) -> ImageStats {
    ImageStats { min: st.min, max: st.max, median: st.median, mad: st.mad, sigma: st.sigma, mean: st.mean, valid_count: st.valid_count }
}
fn stf_to_sys(p: &StfParams) -> sys::ab_stf_params {
    sys::ab_stf_params { shadow: p.shadow, midtone: p.midtone, highlight: p.highlight }
}
fn stf_from_sys(p: &sys::ab_stf_params) -> StfParams {
    StfParams { shadow: p.shadow, midtone: p.midtone, highlight: p.highlight }
}
fn stack_cfg(c: &StackConfig) -> sys::ab_stack_config {
    sys::ab_stack_config { sigma_low: c.sigma_low, sigma_high: c.sigma_high, max_iterations: c.max_iterations as u32, align: c.align as i32 }
}
fn scnr_to_sys(c: &ScnrConfig) -> sys::ab_scnr_config {
    sys::ab_scnr_config { method: matches!(c.method, ScnrMethod::MaximumNeutral) as i32, amount: c.amount, preserve_luminance: c.preserve_luminance as i32 }
}
fn star_from_sys(s: &sys::ab_detected_star) -> DetectedStar {
    DetectedStar { x: s.x, y: s.y, flux: s.flux, fwhm: s.fwhm, eccentricity: s.eccentricity, peak: s.peak, npix: s.npix as usize, snr: s.snr }
}
fn star_to_sys(s: &DetectedStar) -> sys::ab_detected_star {
    sys::ab_detected_star { x: s.x, y: s.y, flux: s.flux, fwhm: s.fwhm, eccentricity: s.eccentricity, peak: s.peak, snr: s.snr, npix: s.npix as u64 }
}
fn align_from_sys(r: &sys::ab_affine_align_result) -> AffineAlignResult {
    let t = r.transform;
    AffineAlignResult {
        transform: AffineTransform { a: t[0], b: t[1], tx: t[2], c: t[3], d: t[4], ty: t[5] },
        matched_stars: r.matched_stars as usize,
        inliers: r.inliers as usize,
        residual_px: r.residual_px,
        method: match r.method {
            0 => AffineAlignMethod::Affine,
            1 => AffineAlignMethod::Rigid,
            2 => AffineAlignMethod::PhaseCorrelation,
            _ => AffineAlignMethod::Identity,
        },
    }
}
fn num_threads() -> i32 {
    rayon::current_num_threads() as i32 // the RANSAC draws are partitioned and seeded per rayon worker (affine.rs:410-416)
}
fn opt_plane<P: PlaneSrc>(p: Option<&P>) -> Option<sys::ab_plane> {
    p.map(|x| x.ab())
}
fn opt_ptr(p: &Option<sys::ab_plane>) -> *const sys::ab_plane {
    p.as_ref().map_or(std::ptr::null(), |x| x as *const _)
}
fn masked_cfg(c: &MaskedStretchConfig) -> sys::ab_masked_stretch_config {
    sys::ab_masked_stretch_config {
        iterations: c.iterations,
        target_background: c.target_background,
        mask_growth: c.mask_growth,
        mask_softness: c.mask_softness,
        luminance_protect: c.luminance_protect as i32,
        luminance_ceiling: c.luminance_ceiling,
        protection_amount: c.protection_amount,
        convergence_threshold: c.convergence_threshold,
    }
}
fn masked_res(image: Array2<f32>, r: &sys::ab_masked_stretch_result) -> MaskedStretchResult {
    MaskedStretchResult {
        image,
        iterations_run: r.iterations_run,
        final_background: r.final_background,
        stars_masked: r.stars_masked,
        mask_coverage: r.mask_coverage,
        converged: r.converged != 0,
    }
}
fn mask_cfg(c: &StarMaskConfig) -> sys::ab_star_mask_config {
    sys::ab_star_mask_config {
        growth_factor: c.growth_factor,
        softness: c.softness,
        detection_sigma: c.detection_sigma,
        min_fwhm: c.min_fwhm,
        max_fwhm: c.max_fwhm,
        luminance_protect: c.luminance_protect as i32,
        luminance_ceiling: c.luminance_ceiling,
    }
}

// ---- a1 / a2  core/stacking/combine.rs -----------------------------------------------------------------------------------------
/// drop-in for core::stacking::combine::stack_images (combine.rs:94-193)
pub fn stack_images(hip: &Hip, images: &[Array2<f32>], config: &StackConfig) -> Result<StackResult> {
    if images.is_empty() {
        bail!("No images to stack"); // combine.rs:98-100
    }
    let rows = images.iter().map(|i| i.nrows()).min().unwrap();
    let cols = images.iter().map(|i| i.ncols()).min().unwrap();
    let planes: Vec<_> = images.iter().map(|i| i.ab()).collect();
    let mut out = Array2::<f32>::zeros((rows, cols));
    let mut po = out.ab_mut();
    let mut offs = vec![0i32; 2 * images.len()];
    let mut rejected = 0u64;
    hip.check(unsafe { sys::ab_stack_images(hip.ctx, planes.as_ptr(), planes.len(), &stack_cfg(config), &mut po, offs.as_mut_ptr(), &mut rejected) })?;
    Ok(StackResult { image: out, frame_count: images.len(), rejected_pixels: rejected, offsets: offs.chunks(2).map(|c| (c[0], c[1])).collect() })
}

/// the per-pixel loop of stack_images alone (combine.rs:160-182; `sigma_clip_combine` :14-92 per pixel) on frames that are
/// already registered, host or device, into a host or device plane; returns StackResult.rejected_pixels
pub fn stack_sigma_clip_into<P: PlaneSrc>(hip: &Hip, frames: &[P], config: &StackConfig, out: &mut impl PlaneDst) -> Result<u64> {
    let planes: Vec<_> = frames.iter().map(|f| f.ab()).collect();
    let mut rejected = 0u64;
    hip.check(unsafe { sys::ab_stack_sigma_clip(hip.ctx, planes.as_ptr(), planes.len(), &stack_cfg(config), &mut out.ab_mut(), &mut rejected) })?;
    Ok(rejected)
}
/// rejected_pixels of the last stack on this context when the call itself was made without waiting for it
pub fn stack_last_rejected(hip: &Hip) -> Result<u64> {
    let mut r = 0u64;
    hip.check(unsafe { sys::ab_stack_last_rejected(hip.ctx, &mut r) })?;
    Ok(r)
}
/// two-level estimator for more frames than one GPU holds: per-shard survivor sums and counts, then one divide
pub fn stack_sigma_clip_partial(hip: &Hip, frames: &[DevicePlane], config: &StackConfig, rows: usize, cols: usize, sum_dev: *mut f64, cnt_dev: *mut u32) -> Result<u64> {
    let planes: Vec<_> = frames.iter().map(|f| f.ab()).collect();
    let mut rejected = 0u64;
    hip.check(unsafe { sys::ab_stack_sigma_clip_partial(hip.ctx, planes.as_ptr(), planes.len(), &stack_cfg(config), rows as i64, cols as i64, sum_dev, cnt_dev, &mut rejected) })?;
    Ok(rejected)
}
pub fn stack_finalize_partial(hip: &Hip, sum_dev: *const f64, cnt_dev: *const u32, out: &mut DevicePlane) -> Result<()> {
    let (r, c) = out.dims();
    hip.check(unsafe { sys::ab_stack_finalize_partial(hip.ctx, sum_dev, cnt_dev, (r * c) as i64, out.ab_mut().data) })
}
/// stack_images' per-pixel loop fed straight from FITS data units in HBM (decode_pixels fused into the gather, reader.rs:36-83)
pub fn stack_sigma_clip_raw(hip: &Hip, raw_planes_dev: &[*const c_void], bitpix: i64, bscale: f64, bzero: f64, config: &StackConfig, out: &mut DevicePlane) -> Result<u64> {
    let mut rejected = 0u64;
    hip.check(unsafe { sys::ab_stack_sigma_clip_raw(hip.ctx, raw_planes_dev.as_ptr(), raw_planes_dev.len(), bitpix, bscale, bzero, &stack_cfg(config), &mut out.ab_mut(), &mut rejected) })?;
    Ok(rejected)
}

// ---- a3 / a4 / a5  core/stacking/align.rs, core/alignment/affine.rs --------------------------------------------------------------
/// drop-in for core::stacking::align::shift_image_subpixel (align.rs:36-70)
pub fn shift_image_subpixel(hip: &Hip, image: &impl PlaneSrc, dy: f64, dx: f64) -> Result<Array2<f32>> {
    let mut out = Array2::<f32>::zeros(image.dims());
    hip.check(unsafe { sys::ab_shift_image_subpixel(hip.ctx, &image.ab(), dy, dx, &mut out.ab_mut()) })?;
    Ok(out)
}
/// drop-in for core::alignment::affine::warp_image (affine.rs:663-690)
pub fn warp_image(hip: &Hip, image: &impl PlaneSrc, t: &AffineTransform, out_rows: usize, out_cols: usize) -> Result<Array2<f32>> {
    let mut out = Array2::<f32>::zeros((out_rows, out_cols));
    warp_image_into(hip, image, t, &mut out)?;
    Ok(out)
}
pub fn warp_image_into(hip: &Hip, image: &impl PlaneSrc, t: &AffineTransform, out: &mut impl PlaneDst) -> Result<()> {
    let m = [t.a, t.b, t.tx, t.c, t.d, t.ty];
    hip.check(unsafe { sys::ab_warp_image(hip.ctx, &image.ab(), m.as_ptr(), &mut out.ab_mut()) })
}
/// rows [row0, row0 + band.rows) of warp_image's output (a GPU's row band of a registered frame)
pub fn warp_image_rows(hip: &Hip, image: &impl PlaneSrc, t: &AffineTransform, out_rows: usize, row0: usize, band: &mut impl PlaneDst) -> Result<()> {
    let m = [t.a, t.b, t.tx, t.c, t.d, t.ty];
    hip.check(unsafe { sys::ab_warp_image_rows(hip.ctx, &image.ab(), m.as_ptr(), out_rows as i64, row0 as i64, &mut band.ab_mut()) })
}

/// the same with the source given as rows [src_row0, src_row0 + src_band.rows) of a `src_rows`-row frame: a GPU of the row-band
/// scheme holds its rows of every target + the halo `shard_source_rows` names, not the frame set
pub fn warp_image_rows_from_band(hip: &Hip, src_band: &impl PlaneSrc, src_row0: usize, src_rows: usize, t: &AffineTransform, out_rows: usize,
                                 row0: usize, band: &mut impl PlaneDst) -> Result<()> {
    let m = [t.a, t.b, t.tx, t.c, t.d, t.ty];
    hip.check(unsafe {
        sys::ab_warp_image_rows_from_band(hip.ctx, &src_band.ab(), src_row0 as i64, src_rows as i64, m.as_ptr(), out_rows as i64, row0 as i64, &mut band.ab_mut())
    })
}
/// source rows (first, count) that output rows [row0, row0 + nrows) of warp_image read (affine.rs:663-690, sampling.rs:48-80)
/// (context-free: a non-zero status -- negative dimensions, a band outside the output -- has no message to fetch, so it is named here;
/// it must not become "(0, 0): no source rows needed")
pub fn warp_source_rows(t: &AffineTransform, src_rows: usize, src_cols: usize, out_cols: usize, row0: usize, nrows: usize) -> Result<(usize, usize)> {
    let m = [t.a, t.b, t.tx, t.c, t.d, t.ty];
    let (mut s0, mut sn) = (0i64, 0i64);
    let rc = unsafe { sys::ab_warp_source_rows(m.as_ptr(), src_rows as i64, src_cols as i64, out_cols as i64, row0 as i64, nrows as i64, &mut s0, &mut sn) };
    if rc != 0 {
        bail!("warp_source_rows: invalid arguments (status {rc}: src {src_rows}x{src_cols}, out cols {out_cols}, rows {row0}+{nrows})");
    }
    Ok((s0 as usize, sn as usize))
}

// ---- a8  core/alignment/phase_correlation.rs --------------------------------------------------------------------------------------
/// drop-in for phase_correlate (phase_correlation.rs:22-89)
pub fn phase_correlate(hip: &Hip, reference: &impl PlaneSrc, target: &impl PlaneSrc) -> Result<PhaseCorrelationResult> {
    let mut r = sys::ab_phase_correlation_result { dx: 0.0, dy: 0.0, confidence: 0.0 };
    hip.check(unsafe { sys::ab_phase_correlate(hip.ctx, &reference.ab(), &target.ab(), &mut r) })?;
    Ok(PhaseCorrelationResult { dx: r.dx, dy: r.dy, confidence: r.confidence })
}

// ---- a7  core/analysis/star_detection.rs -----------------------------------------------------------------------------------------
/// drop-in for estimate_background (star_detection.rs:32-84)
pub fn estimate_background(hip: &Hip, image: &impl PlaneSrc, tile_size: usize) -> Result<(f64, f64)> {
    let (mut med, mut sig) = (0.0, 1.0);
    hip.check(unsafe { sys::ab_estimate_background(hip.ctx, &image.ab(), tile_size as i64, &mut med, &mut sig) })?;
    Ok((med, sig))
}
/// drop-in for detect_stars (star_detection.rs:86-258)
pub fn detect_stars(hip: &Hip, image: &impl PlaneSrc, sigma_threshold: f64) -> Result<DetectionResult> {
    let (rows, cols) = image.dims();
    let mut cap = 4096usize;
    loop {
        let mut buf: Vec<sys::ab_detected_star> = Vec::with_capacity(cap);
        let (mut count, mut total) = (0usize, 0usize);
        let (mut bg_median, mut bg_sigma) = (0.0, 1.0);
        hip.check(unsafe { sys::ab_detect_stars(hip.ctx, &image.ab(), sigma_threshold, buf.as_mut_ptr(), cap, &mut count, &mut total, &mut bg_median, &mut bg_sigma) })?;
        if total > cap {
            cap = total; // a crowded field: once more with room for every star
            continue;
        }
        unsafe { buf.set_len(count) };
        return Ok(DetectionResult {
            stars: buf.iter().map(star_from_sys).collect(),
            background_median: bg_median,
            background_sigma: bg_sigma,
            threshold_sigma: sigma_threshold,
            image_width: cols,
            image_height: rows,
        });
    }
}

// ---- a6  core/alignment/affine.rs, pair.rs ------------------------------------------------------------------------------------------
/// drop-in for normalize_for_detection (affine.rs:24-53)
pub fn normalize_for_detection(hip: &Hip, image: &impl PlaneSrc) -> Result<Array2<f32>> {
    let mut out = Array2::<f32>::zeros(image.dims());
    hip.check(unsafe { sys::ab_normalize_for_detection(hip.ctx, &image.ab(), &mut out.ab_mut()) })?;
    Ok(out)
}
/// drop-in for align_channel_affine (affine.rs:129-212)
pub fn align_channel_affine(hip: &Hip, reference: &impl PlaneSrc, target: &impl PlaneSrc) -> Result<AffineAlignResult> {
    let mut r: sys::ab_affine_align_result = unsafe { std::mem::zeroed() };
    hip.check(unsafe { sys::ab_align_channel_affine(hip.ctx, &reference.ab(), &target.ab(), num_threads(), &mut r) })?;
    Ok(align_from_sys(&r))
}
/// align_channel_affine(reference, targets[i]) for every i in one call: the loop of align_channels_cmd (cmd/compose/blend.rs:226-232)
/// and of compose::rgb::align_channels (rgb.rs:165-189); the reference frame is detected once
pub fn register_frames<P: PlaneSrc>(hip: &Hip, reference: &impl PlaneSrc, targets: &[P]) -> Result<Vec<AffineAlignResult>> {
    let planes: Vec<_> = targets.iter().map(|t| t.ab()).collect();
    let mut out: Vec<sys::ab_affine_align_result> = vec![unsafe { std::mem::zeroed() }; targets.len()];
    hip.check(unsafe { sys::ab_register_frames(hip.ctx, &reference.ab(), planes.as_ptr(), planes.len(), num_threads(), out.as_mut_ptr()) })?;
    Ok(out.iter().map(align_from_sys).collect())
}
/// align_pair(reference, targets[i], Affine) for every i (pair.rs:41-77): estimate + warp into HBM.  The reference and the targets
/// may be host arrays (`&Array2<f32>` as GLOBAL_IMAGE_CACHE holds them) or DevicePlanes: host frames are uploaded by the library
/// on its own stream and registered as they land (one call for the whole set -- do not chunk it)
pub fn align_pairs_affine<P: PlaneSrc>(hip: &Hip, reference: &impl PlaneSrc, targets: &[P], aligned: &mut [DevicePlane]) -> Result<Vec<AffineAlignResult>> {
    let planes: Vec<_> = targets.iter().map(|t| t.ab()).collect();
    let mut outs: Vec<_> = aligned.iter_mut().map(|a| a.ab_mut()).collect();
    let mut res: Vec<sys::ab_affine_align_result> = vec![unsafe { std::mem::zeroed() }; targets.len()];
    hip.check(unsafe { sys::ab_align_pairs_affine(hip.ctx, &reference.ab(), planes.as_ptr(), planes.len(), num_threads(), res.as_mut_ptr(), outs.as_mut_ptr()) })?;
    Ok(res.iter().map(align_from_sys).collect())
}
/// the star-list half of align_channel_affine on given centroids (host arithmetic inside the library)
pub fn affine_from_stars(ref_xy: &[(f64, f64)], tgt_xy: &[(f64, f64)], rows: usize, cols: usize) -> Option<AffineAlignResult> {
    let r: Vec<f64> = ref_xy.iter().flat_map(|p| [p.0, p.1]).collect();
    let t: Vec<f64> = tgt_xy.iter().flat_map(|p| [p.0, p.1]).collect();
    let mut out: sys::ab_affine_align_result = unsafe { std::mem::zeroed() };
    let mut found = 0i32;
    let rc = unsafe { sys::ab_affine_from_stars(r.as_ptr(), ref_xy.len(), t.as_ptr(), tgt_xy.len(), rows as i64, cols as i64, num_threads(), &mut out, &mut found) };
    (rc == sys::AB_OK && found != 0).then(|| align_from_sys(&out))
}
/// drop-in for core::alignment::pair::align_pair (pair.rs:41-77)
pub fn align_pair(hip: &Hip, reference: &impl PlaneSrc, target: &impl PlaneSrc, method: AlignMethod, rows: usize, cols: usize) -> Result<AlignPairResult> {
    match method {
        AlignMethod::PhaseCorrelation => {
            let pc = phase_correlate(hip, reference, target)?;
            Ok(AlignPairResult {
                aligned: shift_image_subpixel(hip, target, pc.dy, pc.dx)?,
                offset: (pc.dy, pc.dx),
                confidence: pc.confidence,
                method_used: "phase_correlation".into(),
                matched_stars: 0,
                inliers: 0,
                residual_px: 0.0,
            })
        }
        AlignMethod::Affine => {
            let r = align_channel_affine(hip, reference, target)?;
            Ok(AlignPairResult {
                aligned: warp_image(hip, target, &r.transform, rows, cols)?,
                offset: (r.transform.ty, r.transform.tx),
                confidence: if r.inliers > 0 { 1.0 } else { 0.0 },
                method_used: r.method.to_string(),
                matched_stars: r.matched_stars,
                inliers: r.inliers,
                residual_px: r.residual_px,
            })
        }
    }
}

// ---- a9 / a10 / a11  core/imaging/stats.rs, stf.rs ----------------------------------------------------------------------------------
/// drop-in for compute_image_stats (stats.rs:15-23)
pub fn compute_image_stats(hip: &Hip, data: &impl PlaneSrc) -> Result<ImageStats> {
    let mut st: sys::ab_image_stats = unsafe { std::mem::zeroed() };
    hip.check(unsafe { sys::ab_compute_image_stats(hip.ctx, &data.ab(), &mut st) })?;
    Ok(stats_from_sys(&st))
}
/// drop-in for compute_image_stats_with_known_range (stats.rs:25-83)
pub fn compute_image_stats_with_known_range(hip: &Hip, data: &impl PlaneSrc, known_min: f64, known_max: f64) -> Result<ImageStats> {
    let mut st: sys::ab_image_stats = unsafe { std::mem::zeroed() };
    hip.check(unsafe { sys::ab_compute_image_stats_with_known_range(hip.ctx, &data.ab(), known_min, known_max, &mut st) })?;
    Ok(stats_from_sys(&st))
}
/// drop-in for build_histogram (stats.rs:212-240)
pub fn build_histogram(hip: &Hip, data: &impl PlaneSrc, bins: usize, dmin: f64, dmax: f64) -> Result<Vec<u32>> {
    let mut out = vec![0u32; bins];
    hip.check(unsafe { sys::ab_build_histogram(hip.ctx, &data.ab(), bins, dmin, dmax, out.as_mut_ptr()) })?;
    Ok(out)
}
/// the 65 536-bin value histogram + sum + count behind the large-image statistics (stats.rs:85-210), for callers that merge shards
pub fn stats_value_hist(hip: &Hip, data: &impl PlaneSrc, gmin: f64, gmax: f64) -> Result<(Vec<u64>, f64, u64)> {
    let mut hist = vec![0u64; 65536];
    let (mut sum, mut cnt) = (0.0f64, 0u64);
    hip.check(unsafe { sys::ab_stats_value_hist(hip.ctx, &data.ab(), gmin, gmax, hist.as_mut_ptr(), &mut sum, &mut cnt) })?;
    Ok((hist, sum, cnt))
}
/// drop-in for auto_stf (stf.rs:13-47): host arithmetic inside the library
pub fn auto_stf(stats: &ImageStats, config: &AutoStfConfig) -> StfParams {
    let cfg = sys::ab_auto_stf_config { target_bg: config.target_bg, shadow_k: config.shadow_k };
    let mut p = sys::ab_stf_params { shadow: 0.0, midtone: 0.5, highlight: 1.0 };
    unsafe { sys::ab_auto_stf(&stats_to_sys(stats), &cfg, &mut p) };
    stf_from_sys(&p)
}
/// drop-in for apply_stf (stf.rs:89-102)
pub fn apply_stf(hip: &Hip, data: &impl PlaneSrc, p: &StfParams, st: &ImageStats) -> Result<Vec<u8>> {
    let (r, c) = data.dims();
    let mut out = vec![0u8; r * c];
    hip.check(unsafe { sys::ab_apply_stf_u8(hip.ctx, &data.ab(), &stf_to_sys(p), &stats_to_sys(st), out.as_mut_ptr(), 0) })?;
    Ok(out)
}
/// drop-in for apply_stf_f32 (stf.rs:104-118)
pub fn apply_stf_f32(hip: &Hip, data: &impl PlaneSrc, p: &StfParams, st: &ImageStats) -> Result<Array2<f32>> {
    let mut out = Array2::<f32>::zeros(data.dims());
    apply_stf_f32_into(hip, data, p, st, &mut out)?;
    Ok(out)
}
pub fn apply_stf_f32_into(hip: &Hip, data: &impl PlaneSrc, p: &StfParams, st: &ImageStats, out: &mut impl PlaneDst) -> Result<()> {
    hip.check(unsafe { sys::ab_apply_stf_f32(hip.ctx, &data.ab(), &stf_to_sys(p), &stats_to_sys(st), &mut out.ab_mut()) })
}
/// drop-in for cmd::common::auto_stretch_preview (cmd/common.rs:18-22): stats -> auto_stf -> apply_stf as one device chain.
/// A host plane is uploaded once and never read back; only the u8 preview and 80 bytes of scalars return.
pub fn auto_stretch_preview(hip: &Hip, arr: &impl PlaneSrc) -> Result<(Vec<u8>, ImageStats, StfParams)> {
    let (rows, cols) = arr.dims();
    let src = arr.ab();
    let staged = if src.on_device == 0 { Some(DevicePlane::alloc(rows, cols)?) } else { None };
    if let Some(d) = staged.as_ref() {
        hip.check(unsafe { sys::ab_upload(hip.ctx, d.as_ptr() as *mut c_void, src.data as *const c_void, rows * cols * 4) })?;
    }
    let p = staged.as_ref().map_or(src, |d| d.ab());
    let mut du8 = std::ptr::null_mut();
    hip.check(unsafe { sys::ab_device_alloc(hip.ctx, rows * cols, &mut du8) })?;
    let r = (|| {
        let mut st: sys::ab_image_stats = unsafe { std::mem::zeroed() };
        let mut stf = sys::ab_stf_params { shadow: 0.0, midtone: 0.5, highlight: 1.0 };
        hip.check(unsafe { sys::ab_auto_stretch_preview(hip.ctx, std::ptr::null_mut(), &p, 0, std::ptr::null(), du8 as *mut u8, &mut st, &mut stf) })?;
        let mut out = vec![0u8; rows * cols];
        hip.check(unsafe { sys::ab_download(hip.ctx, out.as_mut_ptr() as *mut c_void, du8, rows * cols) })?;
        Ok((out, stats_from_sys(&st), stf_from_sys(&stf)))
    })();
    unsafe { sys::ab_device_free(hip.ctx, du8) };
    r
}

// ---- a14  core/imaging/scnr.rs ---------------------------------------------------------------------------------------------------------
/// drop-in for apply_scnr_inplace (scnr.rs:18-53); mismatched dims or amount < 1e-7: a silent no-op, as in the reference
pub fn apply_scnr_inplace(hip: &Hip, r: &mut impl PlaneDst, g: &mut impl PlaneDst, b: &mut impl PlaneDst, config: &ScnrConfig) -> Result<()> {
    hip.check(unsafe { sys::ab_apply_scnr_inplace(hip.ctx, &mut r.ab_mut(), &mut g.ab_mut(), &mut b.ab_mut(), &scnr_to_sys(config)) })
}

// ---- a16  core/compose/channel_blend.rs ----------------------------------------------------------------------------------------------------
fn blend_weights(w: &[BlendWeight]) -> Vec<sys::ab_blend_weight> {
    w.iter().map(|x| sys::ab_blend_weight { channel_idx: x.channel_idx as u64, r_weight: x.r_weight, g_weight: x.g_weight, b_weight: x.b_weight }).collect()
}
/// drop-in for blend_channels (channel_blend.rs:13-70)
pub fn blend_channels<P: PlaneSrc>(hip: &Hip, channels: &[P], weights: &[BlendWeight], rows: usize, cols: usize) -> Result<(Array2<f32>, Array2<f32>, Array2<f32>)> {
    let (mut r, mut g, mut b) = (Array2::<f32>::zeros((rows, cols)), Array2::<f32>::zeros((rows, cols)), Array2::<f32>::zeros((rows, cols)));
    blend_channels_into(hip, channels, weights, &mut r, &mut g, &mut b)?;
    Ok((r, g, b))
}
pub fn blend_channels_into<P: PlaneSrc>(hip: &Hip, channels: &[P], weights: &[BlendWeight], r: &mut impl PlaneDst, g: &mut impl PlaneDst, b: &mut impl PlaneDst) -> Result<()> {
    let planes: Vec<_> = channels.iter().map(|c| c.ab()).collect();
    let w = blend_weights(weights);
    hip.check(unsafe { sys::ab_blend_channels(hip.ctx, planes.as_ptr(), planes.len(), w.as_ptr(), w.len(), &mut r.ab_mut(), &mut g.ab_mut(), &mut b.ab_mut()) })
}

// ---- a15  core/imaging/curves.rs -------------------------------------------------------------------------------------------------------------
/// SplineLut (curves.rs:64-184): the 4096-entry table of SplineLut::from_points (Fritsch-Carlson monotone cubic)
pub struct SplineLut {
    lut: [f32; 4096],
}
impl SplineLut {
    pub fn from_points(points: &[(f64, f64)]) -> Self {
        let flat: Vec<f64> = points.iter().flat_map(|p| [p.0, p.1]).collect();
        let mut lut = [0f32; 4096];
        unsafe { sys::ab_spline_lut_from_points(flat.as_ptr(), points.len(), lut.as_mut_ptr()) };
        Self { lut }
    }
}
/// drop-in for apply_curve (curves.rs:186-197)
pub fn apply_curve(hip: &Hip, data: &impl PlaneSrc, lut: &SplineLut) -> Result<Array2<f32>> {
    let mut out = Array2::<f32>::zeros(data.dims());
    apply_curve_into(hip, data, lut, &mut out)?;
    Ok(out)
}
pub fn apply_curve_into(hip: &Hip, data: &impl PlaneSrc, lut: &SplineLut, out: &mut impl PlaneDst) -> Result<()> {
    hip.check(unsafe { sys::ab_apply_curve(hip.ctx, &data.ab(), lut.lut.as_ptr(), &mut out.ab_mut()) })
}
/// drop-in for apply_levels (curves.rs:31-52)
pub fn apply_levels(hip: &Hip, data: &impl PlaneSrc, p: &LevelsParams) -> Result<Array2<f32>> {
    let mut out = Array2::<f32>::zeros(data.dims());
    apply_levels_into(hip, data, p, &mut out)?;
    Ok(out)
}
pub fn apply_levels_into(hip: &Hip, data: &impl PlaneSrc, p: &LevelsParams, out: &mut impl PlaneDst) -> Result<()> {
    let lp = sys::ab_levels_params { black: p.black, gamma: p.gamma, white: p.white };
    hip.check(unsafe { sys::ab_apply_levels(hip.ctx, &data.ab(), &lp, &mut out.ab_mut()) })
}
/// drop-ins for apply_levels_rgb / apply_curve_rgb (curves.rs:54-62,199-207): three launches on one stream
pub fn apply_levels_rgb(hip: &Hip, r: &impl PlaneSrc, g: &impl PlaneSrc, b: &impl PlaneSrc, lr: &LevelsParams, lg: &LevelsParams, lb: &LevelsParams) -> Result<(Array2<f32>, Array2<f32>, Array2<f32>)> {
    Ok((apply_levels(hip, r, lr)?, apply_levels(hip, g, lg)?, apply_levels(hip, b, lb)?))
}
pub fn apply_curve_rgb(hip: &Hip, r: &impl PlaneSrc, g: &impl PlaneSrc, b: &impl PlaneSrc, lr: &SplineLut, lg: &SplineLut, lb: &SplineLut) -> Result<(Array2<f32>, Array2<f32>, Array2<f32>)> {
    Ok((apply_curve(hip, r, lr)?, apply_curve(hip, g, lg)?, apply_curve(hip, b, lb)?))
}

// ---- a19  core/imaging/stretch.rs ---------------------------------------------------------------------------------------------------------------
/// drop-in for arcsinh_stretch_with_stats (stretch.rs:10-45)
pub fn arcsinh_stretch_with_stats(hip: &Hip, data: &impl PlaneSrc, dmin: f32, dmax: f32, factor: f32, gamma: f32) -> Result<Array2<f32>> {
    let mut out = Array2::<f32>::zeros(data.dims());
    hip.check(unsafe { sys::ab_arcsinh_stretch_with_stats(hip.ctx, &data.ab(), dmin, dmax, factor, gamma, &mut out.ab_mut()) })?;
    Ok(out)
}
/// drop-in for arcsinh_stretch (stretch.rs:5-8)
pub fn arcsinh_stretch(hip: &Hip, data: &impl PlaneSrc, factor: f32) -> Result<Array2<f32>> {
    let st = compute_image_stats(hip, data)?;
    arcsinh_stretch_with_stats(hip, data, st.min as f32, st.max as f32, factor, 1.0)
}
/// compute_luminance (masked_stretch.rs:128-154: the finite-guarded Rec.709 luminance)
pub fn compute_luminance(hip: &Hip, r: &impl PlaneSrc, g: &impl PlaneSrc, b: &impl PlaneSrc) -> Result<Array2<f32>> {
    let mut out = Array2::<f32>::zeros(r.dims());
    hip.check(unsafe { sys::ab_luminance(hip.ctx, &r.ab(), &g.ab(), &b.ab(), &mut out.ab_mut()) })?;
    Ok(out)
}
/// `orig * factor` of calibrate_channel (cmd/compose/color.rs:28-40)
pub fn scale_into(hip: &Hip, data: &impl PlaneSrc, factor: f32, out: &mut impl PlaneDst) -> Result<()> {
    hip.check(unsafe { sys::ab_scale(hip.ctx, &data.ab(), factor, &mut out.ab_mut()) })
}

// ---- a20  core/stacking/calibration.rs ------------------------------------------------------------------------------------------------------------
/// drop-in for calibrate_image (calibration.rs:47-82)
pub fn calibrate_image(hip: &Hip, raw: &impl PlaneSrc, config: &CalibrationConfig) -> Result<Array2<f32>> {
    let (bias, dark, flat) = (opt_plane(config.master_bias.as_ref()), opt_plane(config.master_dark.as_ref()), opt_plane(config.master_flat.as_ref()));
    let mut out = Array2::<f32>::zeros(raw.dims());
    hip.check(unsafe { sys::ab_calibrate_image(hip.ctx, &raw.ab(), opt_ptr(&bias), opt_ptr(&dark), opt_ptr(&flat), config.dark_exposure_ratio, &mut out.ab_mut()) })?;
    Ok(out)
}
/// drop-in for median_combine_row_major on in-memory frames (calibration.rs:84-125)
pub fn median_combine<P: PlaneSrc>(hip: &Hip, frames: &[P]) -> Result<Array2<f32>> {
    if frames.is_empty() {
        bail!("No images to stack");
    }
    let planes: Vec<_> = frames.iter().map(|f| f.ab()).collect();
    let mut out = Array2::<f32>::zeros(frames[0].dims());
    hip.check(unsafe { sys::ab_median_combine(hip.ctx, planes.as_ptr(), planes.len(), &mut out.ab_mut()) })?;
    Ok(out)
}
pub enum MasterKind {
    Bias = 0,
    Dark = 1,
    Flat = 2,
}
/// create_master_bias / _dark / _flat after their loader (calibration.rs:127-255): the frames are already in memory
pub fn create_master<P: PlaneSrc>(hip: &Hip, kind: MasterKind, frames: &[P], master_bias: Option<&Array2<f32>>, master_dark: Option<&Array2<f32>>) -> Result<Array2<f32>> {
    let planes: Vec<_> = frames.iter().map(|f| f.ab()).collect();
    let (bias, dark) = (opt_plane(master_bias), opt_plane(master_dark));
    let dims = frames.first().map_or((0, 0), |f| f.dims());
    let mut out = Array2::<f32>::zeros(dims);
    hip.check(unsafe { sys::ab_create_master(hip.ctx, kind as i32, planes.as_ptr(), planes.len(), opt_ptr(&bias), opt_ptr(&dark), &mut out.ab_mut()) })?;
    Ok(out)
}

// ---- a12  core/imaging/background.rs -----------------------------------------------------------------------------------------------------------------
/// drop-in for extract_background (background.rs:55-116), progress and cancel included
pub fn extract_background(hip: &Hip, image: &impl PlaneSrc, config: &BackgroundConfig, progress: Option<&ProgressHandle>) -> Result<BackgroundResult> {
    let start = std::time::Instant::now();
    let (rows, cols) = image.dims();
    let (mut model, mut corrected) = (Array2::<f32>::zeros((rows, cols)), Array2::<f32>::zeros((rows, cols)));
    let cfg = sys::ab_background_config { grid_size: config.grid_size, poly_degree: config.poly_degree, sigma_clip: config.sigma_clip, iterations: config.iterations, mode: config.mode as i32 };
    let mut info: sys::ab_background_info = unsafe { std::mem::zeroed() };
    let (mut pm, mut pc) = (model.ab_mut(), corrected.ab_mut());
    hip.with_progress(progress, || hip.check(unsafe { sys::ab_extract_background(hip.ctx, &image.ab(), &cfg, &mut pm, &mut pc, &mut info) }))?;
    if let Some(p) = progress {
        p.emit_complete(); // background.rs:105-107
    }
    Ok(BackgroundResult { model, corrected, sample_count: info.sample_count, rms_residual: info.rms_residual, elapsed_ms: start.elapsed().as_millis() as u64 })
}

// ---- a13  core/imaging/star_mask.rs, masked_stretch.rs -------------------------------------------------------------------------------------------------
/// drop-in for generate_star_mask (star_mask.rs:38-44)
pub fn generate_star_mask(hip: &Hip, image: &impl PlaneSrc, config: &StarMaskConfig) -> Result<StarMaskResult, String> {
    let mut mask = Array2::<f32>::zeros(image.dims());
    let mut info = sys::ab_star_mask_info { stars_masked: 0, coverage_fraction: 0.0 };
    hip.check(unsafe { sys::ab_generate_star_mask(hip.ctx, &image.ab(), &mask_cfg(config), &mut mask.ab_mut(), &mut info) }).map_err(|e| e.to_string())?;
    Ok(StarMaskResult { mask, stars_masked: info.stars_masked, coverage_fraction: info.coverage_fraction })
}
/// drop-in for generate_star_mask_from_detection (star_mask.rs:46-138)
pub fn generate_star_mask_from_detection(hip: &Hip, image: &impl PlaneSrc, detection: &DetectionResult, config: &StarMaskConfig) -> Result<StarMaskResult, String> {
    let stars: Vec<_> = detection.stars.iter().map(star_to_sys).collect();
    let mut mask = Array2::<f32>::zeros(image.dims());
    let mut info = sys::ab_star_mask_info { stars_masked: 0, coverage_fraction: 0.0 };
    hip.check(unsafe { sys::ab_generate_star_mask_from_stars(hip.ctx, &image.ab(), stars.as_ptr(), stars.len(), &mask_cfg(config), &mut mask.ab_mut(), &mut info) })
        .map_err(|e| e.to_string())?;
    Ok(StarMaskResult { mask, stars_masked: info.stars_masked, coverage_fraction: info.coverage_fraction })
}
/// drop-in for masked_stretch (masked_stretch.rs:44-58)
pub fn masked_stretch(hip: &Hip, image: &impl PlaneSrc, config: &MaskedStretchConfig) -> Result<MaskedStretchResult, String> {
    let mut out = Array2::<f32>::zeros(image.dims());
    let mut res: sys::ab_masked_stretch_result = unsafe { std::mem::zeroed() };
    hip.check(unsafe { sys::ab_masked_stretch(hip.ctx, &image.ab(), &masked_cfg(config), &mut out.ab_mut(), &mut res) }).map_err(|e| e.to_string())?;
    Ok(masked_res(out, &res))
}
/// drop-in for masked_stretch_with_mask (masked_stretch.rs:60-118)
pub fn masked_stretch_with_mask(hip: &Hip, image: &impl PlaneSrc, mask_result: &StarMaskResult, config: &MaskedStretchConfig) -> Result<MaskedStretchResult, String> {
    let mut out = Array2::<f32>::zeros(image.dims());
    let mut res: sys::ab_masked_stretch_result = unsafe { std::mem::zeroed() };
    let info = sys::ab_star_mask_info { stars_masked: mask_result.stars_masked, coverage_fraction: mask_result.coverage_fraction };
    hip.check(unsafe { sys::ab_masked_stretch_with_mask(hip.ctx, &image.ab(), &mask_result.mask.ab(), &info, &masked_cfg(config), &mut out.ab_mut(), &mut res) })
        .map_err(|e| e.to_string())?;
    Ok(masked_res(out, &res))
}
/// drop-in for masked_stretch_rgb_shared (masked_stretch.rs:157-193)
pub fn masked_stretch_rgb_shared(hip: &Hip, r: &impl PlaneSrc, g: &impl PlaneSrc, b: &impl PlaneSrc, config: &MaskedStretchConfig) -> Result<MaskedStretchRgbResult, String> {
    let d = r.dims();
    let (mut or, mut og, mut ob) = (Array2::<f32>::zeros(d), Array2::<f32>::zeros(d), Array2::<f32>::zeros(d));
    let (res3, shared) = masked_stretch_rgb_shared_into(hip, r, g, b, config, &mut or, &mut og, &mut ob).map_err(|e| e.to_string())?;
    Ok(MaskedStretchRgbResult {
        r: masked_res(or, &res3[0]),
        g: masked_res(og, &res3[1]),
        b: masked_res(ob, &res3[2]),
        shared_mask_coverage: shared.coverage_fraction,
        shared_stars_masked: shared.stars_masked,
    })
}
pub fn masked_stretch_rgb_shared_into(
    hip: &Hip, r: &impl PlaneSrc, g: &impl PlaneSrc, b: &impl PlaneSrc, config: &MaskedStretchConfig,
    out_r: &mut impl PlaneDst, out_g: &mut impl PlaneDst, out_b: &mut impl PlaneDst,
) -> Result<([sys::ab_masked_stretch_result; 3], sys::ab_star_mask_info)> {
    let mut res3: [sys::ab_masked_stretch_result; 3] = unsafe { std::mem::zeroed() };
    let mut shared = sys::ab_star_mask_info { stars_masked: 0, coverage_fraction: 0.0 };
    hip.check(unsafe {
        sys::ab_masked_stretch_rgb_shared(hip.ctx, &r.ab(), &g.ab(), &b.ab(), &masked_cfg(config), &mut out_r.ab_mut(), &mut out_g.ab_mut(), &mut out_b.ab_mut(), res3.as_mut_ptr(), &mut shared)
    })?;
    Ok((res3, shared))
}

// ---- a17  core/compose/rgb.rs, white_balance.rs, lrgb.rs, core/imaging/resample.rs ---------------------------------------------------------------------
/// drop-in for resample_image (resample.rs:25-61)
pub fn resample_image(hip: &Hip, image: &impl PlaneSrc, target_rows: usize, target_cols: usize) -> Result<Array2<f32>> {
    let mut out = Array2::<f32>::zeros((target_rows, target_cols));
    hip.check(unsafe { sys::ab_resample_image(hip.ctx, &image.ab(), &mut out.ab_mut()) })?;
    Ok(out)
}
/// drop-in for select_wb_reference (white_balance.rs:3-20)
pub fn select_wb_reference(sr: &ImageStats, sg: &ImageStats, sb: &ImageStats) -> (f64, f64, f64) {
    let mut m = [1.0f64; 3];
    unsafe { sys::ab_select_wb_reference(&stats_to_sys(sr), &stats_to_sys(sg), &stats_to_sys(sb), m.as_mut_ptr()) };
    (m[0], m[1], m[2])
}
/// drop-in for process_rgb (rgb.rs:209-323)
pub fn process_rgb(hip: &Hip, r: Option<&Array2<f32>>, g: Option<&Array2<f32>>, b: Option<&Array2<f32>>, config: &RgbComposeConfig) -> Result<ProcessedRgb> {
    let count = [r, g, b].iter().filter(|c| c.is_some()).count();
    if count < 2 {
        bail!("Need at least 2 channels for RGB compose (got {})", count); // rgb.rs:217
    }
    let rows = [r, g, b].iter().flatten().map(|a| a.nrows()).max().unwrap();
    let cols = [r, g, b].iter().flatten().map(|a| a.ncols()).max().unwrap();
    let (pr, pg, pb) = (opt_plane(r), opt_plane(g), opt_plane(b));
    let mut cfg: sys::ab_rgb_compose_config = unsafe { std::mem::zeroed() };
    match config.white_balance {
        WhiteBalance::Auto => cfg.white_balance = 0,
        WhiteBalance::Manual(x, y, z) => {
            cfg.white_balance = 1;
            cfg.wb_manual = [x, y, z];
        }
        WhiteBalance::None => cfg.white_balance = 2,
    }
    cfg.auto_stretch = config.auto_stretch as i32;
    cfg.linked_stf = config.linked_stf as i32;
    for (i, s) in [&config.stf_r, &config.stf_g, &config.stf_b].iter().enumerate() {
        if let Some(p) = s {
            cfg.has_stf[i] = 1;
            cfg.stf[i] = stf_to_sys(p);
        }
    }
    cfg.align = config.align as i32;
    cfg.align_method = matches!(config.align_method, AlignMethod::Affine) as i32;
    if let Some(s) = config.scnr.as_ref() {
        cfg.has_scnr = 1;
        cfg.scnr = scnr_to_sys(s);
    }
    cfg.num_threads = num_threads();
    let z = || Array2::<f32>::zeros((rows, cols));
    let (mut or, mut og, mut ob, mut qr, mut qg, mut qb) = (z(), z(), z(), z(), z(), z());
    let mut info: sys::ab_processed_rgb_info = unsafe { std::mem::zeroed() };
    hip.check(unsafe {
        sys::ab_process_rgb(hip.ctx, opt_ptr(&pr), opt_ptr(&pg), opt_ptr(&pb), &cfg, &mut or.ab_mut(), &mut og.ab_mut(), &mut ob.ab_mut(), &mut qr.ab_mut(), &mut qg.ab_mut(), &mut qb.ab_mut(), &mut info)
    })?;
    let cs = |i: usize| ChannelStats { min: info.chan_stats[i][0], max: info.chan_stats[i][1], median: info.chan_stats[i][2], mean: info.chan_stats[i][3] };
    let dims = |a: Option<&Array2<f32>>| a.map(|x| [x.nrows(), x.ncols()]);
    Ok(ProcessedRgb {
        r: or,
        g: og,
        b: ob,
        rows: info.rows as usize,
        cols: info.cols as usize,
        stf_r: stf_from_sys(&info.stf[0]),
        stf_g: stf_from_sys(&info.stf[1]),
        stf_b: stf_from_sys(&info.stf[2]),
        stats_r: cs(0),
        stats_g: cs(1),
        stats_b: cs(2),
        offset_g: (info.offset_g[0], info.offset_g[1]),
        offset_b: (info.offset_b[0], info.offset_b[1]),
        scnr_applied: info.scnr_applied != 0,
        dimension_info: (info.resampled != 0).then(|| DimensionHarmonize { original_r: dims(r), original_g: dims(g), original_b: dims(b), target: [rows, cols], resampled: true }),
        pre_stretch_r: Some(qr),
        pre_stretch_g: Some(qg),
        pre_stretch_b: Some(qb),
        stats_wb_r: Some(stats_from_sys(&info.stats_wb[0])),
        stats_wb_g: Some(stats_from_sys(&info.stats_wb[1])),
        stats_wb_b: Some(stats_from_sys(&info.stats_wb[2])),
    })
}
/// drop-in for apply_lrgb (lrgb.rs:4-45)
pub fn apply_lrgb(hip: &Hip, l: &impl PlaneSrc, r: &mut impl PlaneDst, g: &mut impl PlaneDst, b: &mut impl PlaneDst, lightness_weight: f32, chrominance_weight: f32) -> Result<()> {
    hip.check(unsafe { sys::ab_apply_lrgb(hip.ctx, &l.ab(), &mut r.ab_mut(), &mut g.ab_mut(), &mut b.ab_mut(), lightness_weight, chrominance_weight) })
}
/// drop-in for synthesize_luminance (lrgb.rs:47-64)
pub fn synthesize_luminance(hip: &Hip, r: &impl PlaneSrc, g: &impl PlaneSrc, b: &impl PlaneSrc) -> Result<Array2<f32>> {
    let mut out = Array2::<f32>::zeros(r.dims());
    hip.check(unsafe { sys::ab_synthesize_luminance(hip.ctx, &r.ab(), &g.ab(), &b.ab(), &mut out.ab_mut()) })?;
    Ok(out)
}
/// drop-in for helpers::compute_linked_stf_with_stats (cmd/helpers.rs:185-202)
pub fn compute_linked_stf_with_stats(sr: &ImageStats, sg: &ImageStats, sb: &ImageStats, config: &AutoStfConfig) -> (StfParams, ImageStats) {
    let cfg = sys::ab_auto_stf_config { target_bg: config.target_bg, shadow_k: config.shadow_k };
    let mut p = sys::ab_stf_params { shadow: 0.0, midtone: 0.5, highlight: 1.0 };
    let mut c: sys::ab_image_stats = unsafe { std::mem::zeroed() };
    unsafe { sys::ab_compute_linked_stf(&stats_to_sys(sr), &stats_to_sys(sg), &stats_to_sys(sb), &cfg, &mut p, &mut c) };
    (stf_from_sys(&p), stats_from_sys(&c))
}
/// drop-in for calibrate_channel (cmd/compose/color.rs:21-49): the scaled plane and its statistics
pub fn calibrate_channel(hip: &Hip, orig: &impl PlaneSrc, factor: f32, orig_stats: &ImageStats) -> Result<(Array2<f32>, ImageStats)> {
    let mut out = Array2::<f32>::zeros(orig.dims());
    let st = calibrate_channel_into(hip, orig, factor, orig_stats, &mut out)?;
    Ok((out, st))
}
pub fn calibrate_channel_into(hip: &Hip, orig: &impl PlaneSrc, factor: f32, orig_stats: &ImageStats, out: &mut impl PlaneDst) -> Result<ImageStats> {
    let mut st: sys::ab_image_stats = unsafe { std::mem::zeroed() };
    hip.check(unsafe { sys::ab_calibrate_channel(hip.ctx, &orig.ab(), factor, &stats_to_sys(orig_stats), &mut out.ab_mut(), &mut st) })?;
    Ok(stats_from_sys(&st))
}

// ---- a18  core/astrometry/spcc.rs ---------------------------------------------------------------------------------------------------------------------------
fn spcc_cfg(config: &SpccConfig) -> sys::ab_spcc_config {
    let (kind, custom) = match config.white_reference {
        WhiteReference::AverageSpiral => (0, [0.0; 3]),
        WhiteReference::G2V => (1, [0.0; 3]),
        WhiteReference::Photopic => (2, [0.0; 3]),
        WhiteReference::Custom(r, g, b) => (3, [r, g, b]),
    };
    sys::ab_spcc_config { min_snr: config.min_snr, max_stars: config.max_stars as u64, saturation_limit: config.saturation_limit, white_reference: kind, custom }
}
fn white_ref_name(w: &WhiteReference) -> String {
    // spcc.rs:157-162
    match w {
        WhiteReference::AverageSpiral => "Average Spiral Galaxy".into(),
        WhiteReference::G2V => "G2V (Sun-like)".into(),
        WhiteReference::Photopic => "Photopic".into(),
        WhiteReference::Custom(..) => "Custom".into(),
    }
}
/// drop-in for spcc_calibrate_rgb with the built-in Bp-Rp catalogue (spcc.rs:73-183).  The header enters that path only
/// through its WCS's pixel scale (include/astroburst_hip.h, a18).
pub fn spcc_calibrate_rgb(hip: &Hip, r: &impl PlaneSrc, g: &impl PlaneSrc, b: &impl PlaneSrc, header: &crate::types::header::HduHeader, config: &SpccConfig) -> Result<SpccResult, String> {
    let wcs = crate::core::astrometry::wcs::WcsTransform::from_header(header).map_err(|e| format!("WCS not available: {}. Run Plate Solve first.", e))?; // spcc.rs:80-81
    let mut res: sys::ab_spcc_result = unsafe { std::mem::zeroed() };
    hip.check(unsafe { sys::ab_spcc_calibrate_rgb(hip.ctx, &r.ab(), &g.ab(), &b.ab(), wcs.pixel_scale_arcsec(), &spcc_cfg(config), &mut res) }).map_err(|e| e.to_string())?;
    Ok(spcc_result(&res, config))
}
/// the part of spcc_calibrate_rgb after detection (spcc.rs:90-183) on a detection the caller already has
pub fn spcc_from_detection(hip: &Hip, r: &impl PlaneSrc, g: &impl PlaneSrc, b: &impl PlaneSrc, detection: &DetectionResult, lum_max: f64, pixel_scale_arcsec: f64, config: &SpccConfig) -> Result<SpccResult, String> {
    let stars: Vec<_> = detection.stars.iter().map(star_to_sys).collect();
    let mut res: sys::ab_spcc_result = unsafe { std::mem::zeroed() };
    hip.check(unsafe { sys::ab_spcc_from_detection(hip.ctx, &r.ab(), &g.ab(), &b.ab(), stars.as_ptr(), stars.len(), lum_max, pixel_scale_arcsec, &spcc_cfg(config), &mut res) })
        .map_err(|e| e.to_string())?;
    Ok(spcc_result(&res, config))
}
fn spcc_result(res: &sys::ab_spcc_result, config: &SpccConfig) -> SpccResult {
    SpccResult {
        r_factor: res.r_factor,
        g_factor: res.g_factor,
        b_factor: res.b_factor,
        stars_matched: res.stars_matched as usize,
        stars_total: res.stars_total as usize,
        avg_color_index: res.avg_color_index,
        white_ref_name: white_ref_name(&config.white_reference),
        catalog_name: "Built-in Bp-Rp (synthetic)".into(), // spcc.rs:164-166
        is_synthetic_catalog: true,
    }
}
/// white_reference_rgb (spcc.rs:245-255)
pub fn white_reference_rgb(w: &WhiteReference) -> (f64, f64, f64) {
    let c = spcc_cfg(&SpccConfig { white_reference: w.clone(), ..SpccConfig::default() });
    let mut out = [1.0f64; 3];
    unsafe { sys::ab_spcc_white_reference_rgb(c.white_reference, c.custom.as_ptr(), out.as_mut_ptr()) };
    (out[0], out[1], out[2])
}

// ---- f1  infra/fits/reader.rs, writer.rs ---------------------------------------------------------------------------------------------------------------------------
/// decode_pixels (reader.rs:41-110): a FITS data unit (big-endian, any BITPIX) -> f32 plane, host or device bytes
pub fn fits_decode_pixels(hip: &Hip, data: &[u8], bitpix: i64, bscale: f64, bzero: f64, out: &mut impl PlaneDst) -> Result<()> {
    hip.check(unsafe { sys::ab_fits_decode_pixels(hip.ctx, data.as_ptr() as *const c_void, data.len(), 0, bitpix, bscale, bzero, &mut out.ab_mut()) })
}
/// compute_bzero_bscale (writer.rs:150-186)
pub fn fits_compute_bzero_bscale(hip: &Hip, image: &impl PlaneSrc) -> Result<(f64, f64)> {
    let (mut bzero, mut bscale) = (0.0, 1.0);
    hip.check(unsafe { sys::ab_fits_compute_bzero_bscale(hip.ctx, &image.ab(), &mut bzero, &mut bscale) })?;
    Ok((bzero, bscale))
}
/// the pixel encoder of write_fits_mono (writer.rs:188-240): f32 plane -> big-endian data unit bytes
pub fn fits_encode_pixels(hip: &Hip, image: &impl PlaneSrc, bitpix: i32, bzero: f64, bscale: f64) -> Result<Vec<u8>> {
    let (r, c) = image.dims();
    let mut out = vec![0u8; r * c * (bitpix.unsigned_abs() as usize / 8)];
    hip.check(unsafe { sys::ab_fits_encode_pixels(hip.ctx, &image.ab(), bitpix, bzero, bscale, out.as_mut_ptr() as *mut c_void, 0) })?;
    Ok(out)
}

// ---- f2  core/imaging/calibration_pipeline.rs ----------------------------------------------------------------------------------------------------------------------------
fn batch_cfg(c: &BatchStackConfig) -> sys::ab_batch_stack_config {
    sys::ab_batch_stack_config { sigma_low: c.sigma_low, sigma_high: c.sigma_high, max_iterations: c.max_iterations as u64, normalize_before_stack: c.normalize_before_stack as i32 }
}
struct Masters {
    planes: [Option<sys::ab_plane>; 3],
}
impl Masters {
    fn new(m: &CalibrationMasters) -> Self {
        Self { planes: [opt_plane(m.bias.as_ref()), opt_plane(m.dark.as_ref()), opt_plane(m.flat.as_ref())] }
    }
    fn sys(&self) -> sys::ab_calibration_masters {
        sys::ab_calibration_masters { bias: opt_ptr(&self.planes[0]), dark: opt_ptr(&self.planes[1]), flat: opt_ptr(&self.planes[2]) }
    }
}
/// drop-in for calibrate_light (calibration_pipeline.rs:74-118)
pub fn calibrate_light(hip: &Hip, light: &impl PlaneSrc, masters: &CalibrationMasters) -> Result<Array2<f32>> {
    let m = Masters::new(masters);
    let mut out = Array2::<f32>::zeros(light.dims());
    hip.check(unsafe { sys::ab_calibrate_light(hip.ctx, &light.ab(), &m.sys(), &mut out.ab_mut()) })?;
    Ok(out)
}
/// drop-in for normalize_frames (calibration_pipeline.rs:309-319)
pub fn normalize_frames(hip: &Hip, frames: &[Array2<f32>]) -> Result<Vec<Array2<f32>>> {
    let planes: Vec<_> = frames.iter().map(|f| f.ab()).collect();
    let mut outs: Vec<Array2<f32>> = frames.iter().map(|f| Array2::zeros(f.dim())).collect();
    let mut po: Vec<_> = outs.iter_mut().map(|o| o.ab_mut()).collect();
    hip.check(unsafe { sys::ab_normalize_frames(hip.ctx, planes.as_ptr(), planes.len(), po.as_mut_ptr()) })?;
    Ok(outs)
}
/// drop-in for sigma_clipped_mean_stack (calibration_pipeline.rs:321-378): (master, per-frame rejection counts)
pub fn sigma_clipped_mean_stack(hip: &Hip, frames: &[Array2<f32>], config: &BatchStackConfig) -> Result<(Array2<f32>, Vec<usize>)> {
    let planes: Vec<_> = frames.iter().map(|f| f.ab()).collect();
    let mut out = Array2::<f32>::zeros(frames.first().map_or((0, 0), |f| f.dim()));
    let mut rej = vec![0u64; frames.len()];
    hip.check(unsafe { sys::ab_sigma_clipped_mean_stack(hip.ctx, planes.as_ptr(), planes.len(), &batch_cfg(config), &mut out.ab_mut(), rej.as_mut_ptr()) })?;
    Ok((out, rej.into_iter().map(|x| x as usize).collect()))
}
/// one channel of run_batch_pipeline (calibration_pipeline.rs:157-190), fused: no calibrated or normalised frame is written
pub fn run_batch_channel(hip: &Hip, lights: &[Array2<f32>], masters: &CalibrationMasters, config: &BatchStackConfig, label: &str) -> Result<(Array2<f32>, BatchChannelStats)> {
    let planes: Vec<_> = lights.iter().map(|f| f.ab()).collect();
    let m = Masters::new(masters);
    let mut out = Array2::<f32>::zeros(lights.first().map_or((0, 0), |f| f.dim()));
    let mut rej = vec![0u64; lights.len()];
    let mut st: sys::ab_batch_channel_stats = unsafe { std::mem::zeroed() };
    hip.check(unsafe { sys::ab_run_batch_channel(hip.ctx, planes.as_ptr(), planes.len(), &m.sys(), &batch_cfg(config), &mut out.ab_mut(), rej.as_mut_ptr(), &mut st) })?;
    Ok((out, BatchChannelStats { label: label.to_string(), lights_input: st.lights_input as usize, lights_after_rejection: rej.into_iter().map(|x| x as usize).collect(), mean: st.mean, stddev: st.stddev }))
}
/// compose_rgb_from_masters (calibration_pipeline.rs:201-267)
pub fn compose_rgb_from_masters(hip: &Hip, r: &Array2<f32>, g: &Array2<f32>, b: &Array2<f32>, l: Option<&Array2<f32>>) -> Result<Array3<f32>> {
    let lp = opt_plane(l);
    let (mut rows, mut cols) = (0i64, 0i64);
    hip.check(unsafe { sys::ab_compose_rgb_from_masters(hip.ctx, &r.ab(), &g.ab(), &b.ab(), opt_ptr(&lp), std::ptr::null_mut(), 0, &mut rows, &mut cols) })?;
    let mut out = Array3::<f32>::zeros((rows as usize, cols as usize, 3));
    hip.check(unsafe { sys::ab_compose_rgb_from_masters(hip.ctx, &r.ab(), &g.ab(), &b.ab(), opt_ptr(&lp), out.as_mut_ptr(), 0, &mut rows, &mut cols) })?;
    Ok(out)
}
/// drop-in for run_batch_pipeline (calibration_pipeline.rs:120-199)
pub fn run_batch_pipeline(hip: &Hip, channels: Vec<ChannelInput>, masters: &CalibrationMasters, config: &BatchPipelineConfig) -> Result<BatchPipelineResult, String> {
    if channels.is_empty() {
        return Err("No channels provided".into()); // :125-127
    }
    let labels: Vec<CString> = channels.iter().map(|c| CString::new(c.label.as_str()).unwrap_or_default()).collect();
    let planes: Vec<Vec<sys::ab_plane>> = channels.iter().map(|c| c.lights.iter().map(|l| l.ab()).collect()).collect();
    let mut rej: Vec<Vec<u64>> = channels.iter().map(|c| vec![0u64; c.lights.len()]).collect();
    let inputs: Vec<sys::ab_batch_channel_input> = (0..channels.len())
        .map(|i| sys::ab_batch_channel_input { label: labels[i].as_ptr(), lights: planes[i].as_ptr(), n_lights: planes[i].len(), rejection_counts: rej[i].as_mut_ptr() })
        .collect();
    let m = Masters::new(masters);
    let mut outs: Vec<Array2<f32>> = channels.iter().map(|c| Array2::zeros(c.lights.first().map_or((0, 0), |l| l.dim()))).collect();
    let mut po: Vec<_> = outs.iter_mut().map(|o| o.ab_mut()).collect();
    let mut stats: Vec<sys::ab_batch_channel_stats> = vec![unsafe { std::mem::zeroed() }; channels.len()];
    let rmin = outs.iter().map(|o| o.nrows()).min().unwrap();
    let cmin = outs.iter().map(|o| o.ncols()).min().unwrap();
    let mut rgb = vec![0f32; rmin * cmin * 3];
    let (mut rr, mut rc) = (0i64, 0i64);
    hip.check(unsafe { sys::ab_run_batch_pipeline(hip.ctx, inputs.as_ptr(), inputs.len(), &m.sys(), &batch_cfg(&config.stack), po.as_mut_ptr(), stats.as_mut_ptr(), rgb.as_mut_ptr(), 0, &mut rr, &mut rc) })
        .map_err(|e| e.to_string())?;
    let rgb = (rr > 0).then(|| {
        rgb.truncate(rr as usize * rc as usize * 3);
        Array3::from_shape_vec((rr as usize, rc as usize, 3), rgb).unwrap()
    });
    let chan_stats = (0..channels.len())
        .map(|i| BatchChannelStats {
            label: channels[i].label.clone(),
            lights_input: stats[i].lights_input as usize,
            lights_after_rejection: rej[i].iter().map(|&x| x as usize).collect(),
            mean: stats[i].mean,
            stddev: stats[i].stddev,
        })
        .collect();
    Ok(BatchPipelineResult {
        master_channels: channels.iter().map(|c| c.label.clone()).zip(outs).collect(),
        rgb,
        stats: BatchPipelineStats {
            darks_combined: masters.dark.is_some() as usize, // :192-194
            flats_combined: masters.flat.is_some() as usize,
            bias_combined: masters.bias.is_some() as usize,
            channels: chan_stats,
        },
    })
}

// ---- f3  core/analysis/subframe.rs ----------------------------------------------------------------------------------------------------------------------------------------
fn subframe_cfg(c: &SubframeWeightConfig) -> sys::ab_subframe_weight_config {
    sys::ab_subframe_weight_config {
        fwhm_weight: c.fwhm_weight,
        eccentricity_weight: c.eccentricity_weight,
        snr_weight: c.snr_weight,
        noise_weight: c.noise_weight,
        max_fwhm: c.max_fwhm,
        max_eccentricity: c.max_eccentricity,
        min_snr: c.min_snr,
        min_stars: c.min_stars as u64,
    }
}
fn subframe_metrics(m: &sys::ab_subframe_metrics, file_path: &str) -> SubframeMetrics {
    SubframeMetrics {
        file_path: file_path.to_string(),
        file_name: file_path.split(&['/', '\\'][..]).last().unwrap_or(file_path).to_string(), // subframe.rs:56-60
        star_count: m.star_count as usize,
        median_fwhm: m.median_fwhm,
        median_eccentricity: m.median_eccentricity,
        median_snr: m.median_snr,
        background_median: m.background_median,
        background_sigma: m.background_sigma,
        noise_ratio: m.noise_ratio,
        weight: m.weight,
        accepted: m.accepted != 0,
    }
}
/// drop-in for analyze_subframe (subframe.rs:51-136)
pub fn analyze_subframe(hip: &Hip, image: &impl PlaneSrc, file_path: &str, config: &SubframeWeightConfig) -> Result<SubframeMetrics> {
    let mut m: sys::ab_subframe_metrics = unsafe { std::mem::zeroed() };
    hip.check(unsafe { sys::ab_analyze_subframe(hip.ctx, &image.ab(), &subframe_cfg(config), &mut m) })?;
    Ok(subframe_metrics(&m, file_path))
}
/// analyze_subframe over a batch + normalize_weights (subframe.rs:138-159), the body of analyze_subframes_cmd
pub fn analyze_subframes<P: PlaneSrc>(hip: &Hip, images: &[P], file_paths: &[String], config: &SubframeWeightConfig) -> Result<Vec<SubframeMetrics>> {
    let planes: Vec<_> = images.iter().map(|i| i.ab()).collect();
    let mut ms: Vec<sys::ab_subframe_metrics> = vec![unsafe { std::mem::zeroed() }; images.len()];
    hip.check(unsafe { sys::ab_analyze_subframes(hip.ctx, planes.as_ptr(), planes.len(), &subframe_cfg(config), ms.as_mut_ptr()) })?;
    unsafe { sys::ab_normalize_subframe_weights(ms.as_mut_ptr(), ms.len()) };
    Ok(ms.iter().zip(file_paths).map(|(m, p)| subframe_metrics(m, p)).collect())
}

// ---- f4  cmd/helpers.rs render_rgb_preview, infra/ipc.rs, infra/tiles.rs -----------------------------------------------------------------------------------------------------
/// preview dims of render_rgb_preview (cmd/helpers.rs:204-230)
pub fn preview_dims(rows: usize, cols: usize, max_dim: usize) -> (usize, usize) {
    let (mut r, mut c) = (0i64, 0i64);
    unsafe { sys::ab_preview_dims(rows as i64, cols as i64, max_dim as i64, &mut r, &mut c) };
    (r as usize, c as usize)
}
/// the pixel half of render_rgb_preview(_with_stf): box-downsampled interleaved RGB u8, `stf`/`stats` = None for planes already in [0, 1]
pub fn render_rgb_preview(hip: &Hip, r: &impl PlaneSrc, g: &impl PlaneSrc, b: &impl PlaneSrc, max_dim: usize, stf: Option<(&[StfParams; 3], &[ImageStats; 3])>) -> Result<(Vec<u8>, usize, usize)> {
    let (rows, cols) = r.dims();
    let (pr, pc) = preview_dims(rows, cols, max_dim);
    let mut out = vec![0u8; pr * pc * 3];
    let (p, s): (Vec<_>, Vec<_>) = stf.map_or((vec![], vec![]), |(p, s)| (p.iter().map(stf_to_sys).collect(), s.iter().map(stats_to_sys).collect()));
    let (pp, sp) = if p.is_empty() { (std::ptr::null(), std::ptr::null()) } else { (p.as_ptr(), s.as_ptr()) };
    hip.check(unsafe { sys::ab_render_rgb_preview(hip.ctx, &r.ab(), &g.ab(), &b.ab(), max_dim as i64, pp, sp, out.as_mut_ptr(), 0) })?;
    Ok((out, pr, pc))
}
/// infra/ipc.rs encode_with_header: the f32 preview payload the frontend reads
pub fn ipc_encode_with_header(hip: &Hip, image: &impl PlaneSrc, max_dim: usize) -> Result<Vec<u8>> {
    let (rows, cols) = image.dims();
    let (pr, pc) = preview_dims(rows, cols, max_dim);
    let mut out = vec![0u8; 64 + pr * pc * 4];
    let mut len = 0usize;
    hip.check(unsafe { sys::ab_ipc_encode_with_header(hip.ctx, &image.ab(), max_dim as i64, out.as_mut_ptr() as *mut c_void, 0, &mut len) })?;
    out.truncate(len);
    Ok(out)
}
/// compute_num_levels (tiles.rs:137-147)
pub fn tile_compute_num_levels(width: usize, height: usize, tile_size: usize) -> usize {
    unsafe { sys::ab_tile_compute_num_levels(width as i64, height as i64, tile_size as i64) as usize }
}
/// downsample_2x (tiles.rs:41-70)
pub fn tile_downsample_2x(hip: &Hip, image: &impl PlaneSrc) -> Result<Array2<f32>> {
    let (r, c) = image.dims();
    let mut out = Array2::<f32>::zeros(((r + 1) / 2, (c + 1) / 2));
    hip.check(unsafe { sys::ab_tile_downsample_2x(hip.ctx, &image.ab(), &mut out.ab_mut()) })?;
    Ok(out)
}
/// percentile bounds of the tile normalisation (tiles.rs:72-100)
pub fn tile_percentile_bounds(hip: &Hip, image: &impl PlaneSrc, low_pct: f64, high_pct: f64) -> Result<(f32, f32)> {
    let (mut lo, mut hi) = (0f32, 1f32);
    hip.check(unsafe { sys::ab_tile_percentile_bounds(hip.ctx, &image.ab(), low_pct, high_pct, &mut lo, &mut hi) })?;
    Ok((lo, hi))
}
pub struct TilePyramid {
    pub levels: Vec<sys::ab_tile_level>,
    pub tiles: Vec<u8>,
    pub global_min: f32,
    pub global_max: f32,
}
fn pyramid_layout(rows: usize, cols: usize, tile_size: usize, channels: i32) -> (Vec<sys::ab_tile_level>, i32, usize) {
    let mut levels: Vec<sys::ab_tile_level> = vec![unsafe { std::mem::zeroed() }; 32];
    let (mut n, mut bytes) = (0i32, 0usize);
    unsafe { sys::ab_tile_pyramid_layout(rows as i64, cols as i64, tile_size as i64, channels, levels.as_mut_ptr(), &mut n, &mut bytes) };
    (levels, n, bytes)
}
/// generate_tile_pyramid (tiles.rs:149-230): every level's u8 tiles in one buffer + the level table
pub fn generate_tile_pyramid(hip: &Hip, normalized: &impl PlaneSrc, tile_size: usize) -> Result<TilePyramid> {
    let (rows, cols) = normalized.dims();
    let (mut levels, mut n, bytes) = pyramid_layout(rows, cols, tile_size, 1);
    let mut tiles = vec![0u8; bytes];
    let (mut gmin, mut gmax) = (0f32, 1f32);
    hip.check(unsafe { sys::ab_generate_tile_pyramid(hip.ctx, &normalized.ab(), tile_size as i64, tiles.as_mut_ptr(), 0, levels.as_mut_ptr(), &mut n, &mut gmin, &mut gmax) })?;
    levels.truncate(n as usize);
    Ok(TilePyramid { levels, tiles, global_min: gmin, global_max: gmax })
}
/// the RGB pyramid of the composite viewer (tiles.rs:232-300)
pub fn generate_tile_pyramid_rgb(hip: &Hip, r: &impl PlaneSrc, g: &impl PlaneSrc, b: &impl PlaneSrc, tile_size: usize, stf: &[StfParams; 3], stats: &[ImageStats; 3]) -> Result<TilePyramid> {
    let (rows, cols) = r.dims();
    let (mut levels, mut n, bytes) = pyramid_layout(rows, cols, tile_size, 3);
    let mut tiles = vec![0u8; bytes];
    let p: Vec<_> = stf.iter().map(stf_to_sys).collect();
    let s: Vec<_> = stats.iter().map(stats_to_sys).collect();
    hip.check(unsafe { sys::ab_generate_tile_pyramid_rgb(hip.ctx, &r.ab(), &g.ab(), &b.ab(), tile_size as i64, p.as_ptr(), s.as_ptr(), tiles.as_mut_ptr(), 0, levels.as_mut_ptr(), &mut n) })?;
    levels.truncate(n as usize);
    Ok(TilePyramid { levels, tiles, global_min: 0.0, global_max: 1.0 })
}

// ---- several GPUs of one node (include/astroburst_hip.h section (e)) -----------------------------------------------------------
pub struct Comm(pub *mut sys::ab_comm);
unsafe impl Send for Comm {}
impl Comm {
    pub fn rank(&self) -> i32 {
        unsafe { sys::ab_comm_rank(self.0) }
    }
    pub fn size(&self) -> i32 {
        unsafe { sys::ab_comm_size(self.0) }
    }
    pub fn is_host_transport(&self) -> bool {
        unsafe { sys::ab_comm_is_host(self.0) != 0 }
    }
    pub fn collectives_issued(&self) -> u64 {
        unsafe { sys::ab_comm_collectives_issued(self.0) }
    }
    /// every later collective on this communicator fails with AB_ERR_COMM instead of waiting for a rank that is gone
    pub fn abort(&self) {
        unsafe { sys::ab_comm_abort(self.0) };
    }
    pub fn set_timeout_ms(&self, ms: i64) {
        unsafe { sys::ab_comm_set_timeout_ms(self.0, ms) };
    }
    /// one process per GPU (the RCCL transport): rank 0 makes the id and hands it to the others out of band
    pub fn unique_id() -> Result<[u8; sys::AB_COMM_ID_BYTES]> {
        let mut id = [0u8; sys::AB_COMM_ID_BYTES];
        match unsafe { sys::ab_comm_get_unique_id(id.as_mut_ptr()) } {
            sys::AB_OK => Ok(id),
            rc => bail!("ab_comm_get_unique_id failed ({rc})"),
        }
    }
    pub fn init_rank(hip: &Hip, id: &[u8; sys::AB_COMM_ID_BYTES], nranks: i32, rank: i32) -> Result<Self> {
        let mut c = std::ptr::null_mut();
        hip.check(unsafe { sys::ab_comm_init_rank(hip.ctx, id.as_ptr(), nranks, rank, &mut c) })?;
        Ok(Self(c))
    }
    /// host-staged transport (a POSIX shared-memory segment `name`): several ranks on ONE GPU, or a node without xGMI
    pub fn init_rank_host(hip: &Hip, name: &str, nranks: i32, rank: i32) -> Result<Self> {
        let n = CString::new(name)?;
        let mut c = std::ptr::null_mut();
        hip.check(unsafe { sys::ab_comm_init_rank_host(hip.ctx, n.as_ptr(), nranks, rank, &mut c) })?;
        Ok(Self(c))
    }
    /// agree on a status before a data collective: every rank returns the same Ok / Err (a failed rank fails all, none hangs)
    pub fn agree(&self, hip: &Hip, local_ok: bool) -> Result<()> {
        hip.check(unsafe { sys::ab_comm_agree(hip.ctx, self.0, if local_ok { sys::AB_OK } else { sys::AB_ERR_INVALID }) })
    }
    pub fn allreduce_f64_sum(&self, hip: &Hip, buf_dev: *mut f64, count: usize) -> Result<()> {
        hip.check(unsafe { sys::ab_comm_allreduce(hip.ctx, self.0, buf_dev as *mut c_void, count, sys::AB_DT_F64, sys::AB_RED_SUM) })
    }
    pub fn allgather(&self, hip: &Hip, send_dev: *const c_void, recv_dev: *mut c_void, bytes_per_rank: usize) -> Result<()> {
        hip.check(unsafe { sys::ab_comm_allgather(hip.ctx, self.0, send_dev, recv_dev, bytes_per_rank) })
    }
    pub fn broadcast(&self, hip: &Hip, buf_dev: *mut c_void, bytes: usize, root: i32) -> Result<()> {
        hip.check(unsafe { sys::ab_comm_broadcast(hip.ctx, self.0, buf_dev, bytes, root) })
    }
    /// several collectives as one RCCL group (no-ops on the host transport)
    pub fn group<T>(f: impl FnOnce() -> Result<T>) -> Result<T> {
        unsafe { sys::ab_comm_group_start() };
        let r = f();
        unsafe { sys::ab_comm_group_end() };
        r
    }
}
impl Drop for Comm {
    fn drop(&mut self) {
        unsafe { sys::ab_comm_destroy(self.0) }
    }
}
/// rows [row0, row0 + nrows) of a `rows`-row image that belong to `rank`
pub fn shard_rows(rows: usize, nranks: i32, rank: i32) -> Result<(usize, usize)> {
    let (mut r0, mut n) = (0i64, 0i64);
    let rc = unsafe { sys::ab_shard_rows(rows as i64, nranks, rank, &mut r0, &mut n) };
    if rc != 0 {
        bail!("shard_rows: invalid arguments (status {rc}: {rows} rows, rank {rank} of {nranks})");
    }
    Ok((r0 as usize, n as usize))
}
/// what `rank` must hold of every target frame to warp its band with `transforms`: its rows + the halo (first row, count)
pub fn shard_source_rows(transforms: &[AffineTransform], src_rows: usize, src_cols: usize, out_rows: usize, out_cols: usize, nranks: i32, rank: i32) -> Result<(usize, usize)> {
    let flat: Vec<f64> = transforms.iter().flat_map(|t| [t.a, t.b, t.tx, t.c, t.d, t.ty]).collect();
    let (mut s0, mut sn) = (0i64, 0i64);
    let rc = unsafe {
        sys::ab_shard_source_rows(flat.as_ptr(), transforms.len(), src_rows as i64, src_cols as i64, out_rows as i64, out_cols as i64, nranks, rank, &mut s0, &mut sn)
    };
    if rc != 0 {
        // (AB_ERR_INVALID from ab_shard_rows or the dimension checks: an empty band would only fail later, in ab_warp_image_rows_from_band)
        bail!("shard_source_rows: invalid arguments (status {rc}: src {src_rows}x{src_cols}, out {out_rows}x{out_cols}, rank {rank} of {nranks})");
    }
    Ok((s0 as usize, sn as usize))
}
/// frames [f0, f0 + nf) of an n-frame stack that belong to `rank`
pub fn shard_frames(n_frames: usize, nranks: i32, rank: i32) -> Result<(usize, usize)> {
    let (mut f0, mut nf) = (0usize, 0usize);
    let rc = unsafe { sys::ab_shard_frames(n_frames, nranks, rank, &mut f0, &mut nf) };
    if rc != 0 {
        bail!("shard_frames: invalid arguments (status {rc}: {n_frames} frames, rank {rank} of {nranks})");
    }
    Ok((f0, nf))
}

/// One context + one RCCL rank per GPU, each driven by its own thread (what `handleStackAll`'s concurrent commands already
/// are on the host side).  `f(rank, hip, comm)` runs on every rank; the sharded entry points inside enqueue their collectives
/// on the rank's stream.
pub fn on_all_gpus<T: Send>(devices: &[i32], f: impl Fn(usize, &Hip, &Comm) -> Result<T> + Sync) -> Result<Vec<T>> {
    let hips: Vec<Hip> = devices.iter().map(|&d| Hip::new(d)).collect::<Result<_>>()?;
    let ctxs: Vec<*mut sys::ab_ctx> = hips.iter().map(|h| h.ctx).collect();
    let mut raw = vec![std::ptr::null_mut(); devices.len()];
    hips[0].check(unsafe { sys::ab_comm_init_all(ctxs.as_ptr(), ctxs.len() as i32, raw.as_mut_ptr()) })?;
    let comms: Vec<Comm> = raw.into_iter().map(Comm).collect(); // destroyed on drop, after the scope below has joined every rank
    let results = std::thread::scope(|s| {
        let handles: Vec<_> = hips
            .iter()
            .zip(comms.iter())
            .enumerate()
            .map(|(rank, (hip, comm))| {
                let f = &f;
                let comm = SendRef(comm);
                s.spawn(move || {
                    let c = comm; // the whole wrapper moves into the thread (edition-2021 closures capture fields otherwise)
                    f(rank, hip, c.0)
                })
            })
            .collect();
        handles.into_iter().map(|h| h.join().expect("rank thread")).collect::<Vec<_>>()
    });
    results.into_iter().collect()
}
struct SendRef<'a>(&'a Comm);
unsafe impl<'a> Send for SendRef<'a> {}

/// rows [row0, ..) of the per-pixel loop on this GPU alone (no communicator)
pub fn stack_sigma_clip_rows(hip: &Hip, planes_dev: &[DevicePlane], config: &StackConfig, row0: usize, out_band: &mut DevicePlane) -> Result<u64> {
    let planes: Vec<_> = planes_dev.iter().map(|p| p.ab()).collect();
    let mut rejected = 0u64;
    hip.check(unsafe { sys::ab_stack_sigma_clip_rows(hip.ctx, planes.as_ptr(), planes.len(), &stack_cfg(config), row0 as i64, &mut out_band.ab_mut(), &mut rejected) })?;
    Ok(rejected)
}
/// stack_images over frames already resident on the GPUs, the per-pixel loop split by ROWS (exact: equals the single-GPU
/// and the reference result bit for bit).  `planes_dev` = this rank's device copies of all n frames.
pub fn stack_rowband(hip: &Hip, comm: &Comm, planes_dev: &[DevicePlane], config: &StackConfig, out_band: &mut DevicePlane) -> Result<u64> {
    let planes: Vec<_> = planes_dev.iter().map(|p| p.ab()).collect();
    let mut rejected = 0u64;
    hip.check(unsafe { sys::ab_stack_sigma_clip_rowband(hip.ctx, comm.0, planes.as_ptr(), planes.len(), &stack_cfg(config), &mut out_band.ab_mut(), &mut rejected) })?;
    Ok(rejected)
}
/// the frames sharded over the ranks: per-GPU partial + all-reduce(sum f64, count u32) + divide (the two-level estimator)
pub fn stack_sharded(hip: &Hip, comm: &Comm, local_planes: &[DevicePlane], config: &StackConfig, out: &mut DevicePlane) -> Result<u64> {
    let planes: Vec<_> = local_planes.iter().map(|p| p.ab()).collect();
    let mut rejected = 0u64;
    hip.check(unsafe { sys::ab_stack_sigma_clip_sharded(hip.ctx, comm.0, planes.as_ptr(), planes.len(), &stack_cfg(config), &mut out.ab_mut(), &mut rejected) })?;
    Ok(rejected)
}
/// (partial stacks' span on the context's stream, time inside the all-reduces + divisions on the comm stream) of the last `stack_sharded`, ms
pub fn stack_sharded_last_ms(hip: &Hip) -> Result<(f32, f32)> {
    let (mut stack_ms, mut comm_ms) = (0f32, 0f32);
    hip.check(unsafe { sys::ab_stack_sharded_last_ms(hip.ctx, &mut stack_ms, &mut comm_ms) })?;
    Ok((stack_ms, comm_ms))
}
pub fn allgather_rows(hip: &Hip, comm: &Comm, band: &DevicePlane, full: &mut DevicePlane) -> Result<()> {
    hip.check(unsafe { sys::ab_allgather_rows(hip.ctx, comm.0, &band.ab(), &mut full.ab_mut()) })
}
/// register_frames with the targets spread over the ranks and the 80-byte results exchanged
pub fn register_frames_sharded(hip: &Hip, comm: &Comm, reference: &DevicePlane, targets: &[DevicePlane]) -> Result<Vec<AffineAlignResult>> {
    let planes: Vec<_> = targets.iter().map(|t| t.ab()).collect();
    let mut out: Vec<sys::ab_affine_align_result> = vec![unsafe { std::mem::zeroed() }; targets.len()];
    hip.check(unsafe { sys::ab_register_frames_sharded(hip.ctx, comm.0, &reference.ab(), planes.as_ptr(), planes.len(), num_threads(), out.as_mut_ptr()) })?;
    Ok(out.iter().map(align_from_sys).collect())
}
/// the row-band scheme's registration as one call: this rank's rows `[row0, row0 + bands[i].rows)` of every registered frame.  `targets[i]` holds
/// rows `[target_row0[i], ..)` of target i -- whole for the targets this rank estimates (i mod size == rank); own frames are warped as they are
/// fitted, the estimates are exchanged, the other frames are warped from their bands
pub fn align_pairs_affine_rowband(hip: &Hip, comm: &Comm, reference: &DevicePlane, targets: &[DevicePlane], target_row0: &[usize], row0: usize,
                                  bands: &mut [DevicePlane]) -> Result<Vec<AffineAlignResult>> {
    if targets.len() != target_row0.len() || targets.len() != bands.len() {
        bail!("align_pairs_affine_rowband: {} targets, {} first rows, {} bands", targets.len(), target_row0.len(), bands.len());
    }
    let planes: Vec<_> = targets.iter().map(|t| t.ab()).collect();
    let first: Vec<i64> = target_row0.iter().map(|&r| r as i64).collect();
    let mut outs: Vec<_> = bands.iter_mut().map(|b| b.ab_mut()).collect();
    let mut out: Vec<sys::ab_affine_align_result> = vec![unsafe { std::mem::zeroed() }; targets.len()];
    hip.check(unsafe {
        sys::ab_align_pairs_affine_rowband(hip.ctx, comm.0, &reference.ab(), planes.as_ptr(), first.as_ptr(), planes.len(), num_threads(), row0 as i64,
                                           out.as_mut_ptr(), outs.as_mut_ptr())
    })?;
    Ok(out.iter().map(align_from_sys).collect())
}
/// compute_image_stats of an image of which this rank holds a row band (histograms all-reduced in stream)
pub fn compute_image_stats_sharded(hip: &Hip, comm: &Comm, band: &DevicePlane, total_rows: usize) -> Result<ImageStats> {
    let mut st: sys::ab_image_stats = unsafe { std::mem::zeroed() };
    hip.check(unsafe { sys::ab_compute_image_stats_sharded(hip.ctx, comm.0, &band.ab(), total_rows as i64, &mut st) })?;
    Ok(stats_from_sys(&st))
}
