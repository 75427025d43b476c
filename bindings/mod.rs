//! Safe wrappers over `sys` (bindings/sys.rs, generated from include/astroburst_hip.h) for the AstroBurst backend:
//! drop-ins for the `core::*` functions the Tauri commands call.  Place this directory at `src-tauri/src/hip/`.
//!
//! NOT compiled in the repository that ships it (that image has no Rust toolchain); `sys.rs` is checked field by field
//! against the C header (tests/test_abi_cpu.py), this file is reviewed source.  Every function keeps the signature and
//! the error strings of the `core::*` function it replaces (cited per function), so a command body changes by one path.
//!
//! Threading: one `Hip` per blocking thread (`blocking_cmd!`, cmd/common.rs:345-352) -- an `ab_ctx` owns a stream and
//! scratch and is not shared.  The library never unwinds into Rust (every entry point is a C++ function-try-block) and
//! never aborts; failures are status codes + `ab_last_error`.
#![allow(dead_code)]
pub mod sys;

use anyhow::{anyhow, bail, Result};
use ndarray::{Array2, ArrayView2};
use std::cell::RefCell;
use std::ffi::{c_void, CStr};
use std::os::raw::c_char;

use crate::infra::progress::ProgressHandle;
use crate::types::image::{AutoStfConfig, ImageStats, StfParams};
use crate::types::stacking::{StackConfig, StackResult};

pub struct Hip {
    ctx: *mut sys::ab_ctx,
}
unsafe impl Send for Hip {}

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

    fn check(&self, rc: i32) -> Result<()> {
        if rc == sys::AB_OK {
            return Ok(());
        }
        let msg = unsafe { CStr::from_ptr(sys::ab_last_error(self.ctx)) }.to_string_lossy().into_owned();
        if rc == sys::AB_ERR_CANCELLED {
            return Err(crate::types::error::AppError::Cancelled.into()); // background.rs:80-82
        }
        Err(anyhow!(msg)) // the reference's own strings, e.g. "No images to stack"
    }

    /// Forward the library's stage ticks to a `ProgressHandle` (infra/progress.rs:39-74) for the duration of `f`, and
    /// its cancel flag to the library.  `extract_background` ticks the reference's four stages with their strings.
    pub fn with_progress<T>(&self, progress: Option<&ProgressHandle>, f: impl FnOnce() -> Result<T>) -> Result<T> {
        unsafe extern "C" fn tick(stage: *const c_char, _cur: u64, total: u64, user: *mut c_void) {
            let p = &*(user as *const ProgressHandle);
            p.set_total(total);
            p.tick_with_stage(&CStr::from_ptr(stage).to_string_lossy());
        }
        if let Some(p) = progress {
            if p.is_cancelled() {
                unsafe { sys::ab_ctx_request_cancel(self.ctx) };
            }
            unsafe { sys::ab_ctx_set_progress_cb(self.ctx, Some(tick), p as *const _ as *mut c_void) };
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

fn plane(a: &ArrayView2<f32>) -> sys::ab_plane {
    let s = a.as_slice().expect("contiguous"); // the precondition of combine.rs:154
    sys::ab_plane { data: s.as_ptr(), rows: a.nrows() as i64, cols: a.ncols() as i64, on_device: 0 }
}

fn plane_mut(a: &mut Array2<f32>) -> sys::ab_plane_mut {
    let (rows, cols) = a.dim();
    sys::ab_plane_mut { data: a.as_mut_ptr(), rows: rows as i64, cols: cols as i64, on_device: 0 }
}

fn stats_to_sys(st: &ImageStats) -> sys::ab_image_stats {
    sys::ab_image_stats { min: st.min, max: st.max, median: st.median, mad: st.mad, sigma: st.sigma, mean: st.mean, valid_count: st.valid_count }
}

fn stats_from_sys(st: &sys::ab_image_stats) -> ImageStats {
    ImageStats { min: st.min, max: st.max, median: st.median, mad: st.mad, sigma: st.sigma, mean: st.mean, valid_count: st.valid_count }
}

/// drop-in for core::stacking::combine::stack_images (combine.rs:94-193)
pub fn stack_images(hip: &Hip, images: &[Array2<f32>], config: &StackConfig) -> Result<StackResult> {
    if images.is_empty() {
        bail!("No images to stack"); // combine.rs:98-100
    }
    let rows = images.iter().map(|i| i.nrows()).min().unwrap();
    let cols = images.iter().map(|i| i.ncols()).min().unwrap();
    let planes: Vec<_> = images.iter().map(|i| plane(&i.view())).collect();
    let mut out = Array2::<f32>::zeros((rows, cols));
    let mut po = plane_mut(&mut out);
    let cfg = sys::ab_stack_config {
        sigma_low: config.sigma_low,
        sigma_high: config.sigma_high,
        max_iterations: config.max_iterations as u32,
        align: config.align as i32,
    };
    let mut offs = vec![0i32; 2 * images.len()];
    let mut rejected = 0u64;
    hip.check(unsafe { sys::ab_stack_images(hip.ctx, planes.as_ptr(), planes.len(), &cfg, &mut po, offs.as_mut_ptr(), &mut rejected) })?;
    Ok(StackResult { image: out, frame_count: images.len(), rejected_pixels: rejected, offsets: offs.chunks(2).map(|c| (c[0], c[1])).collect() })
}

/// drop-in for core::imaging::stats::compute_image_stats (stats.rs:15-23)
pub fn compute_image_stats(hip: &Hip, data: &Array2<f32>) -> Result<ImageStats> {
    let mut st: sys::ab_image_stats = unsafe { std::mem::zeroed() };
    hip.check(unsafe { sys::ab_compute_image_stats(hip.ctx, &plane(&data.view()), &mut st) })?;
    Ok(stats_from_sys(&st))
}

/// drop-in for core::imaging::stf::auto_stf (stf.rs:13-47): host arithmetic inside the library
pub fn auto_stf(stats: &ImageStats, config: &AutoStfConfig) -> StfParams {
    let cfg = sys::ab_auto_stf_config { target_bg: config.target_bg, shadow_k: config.shadow_k };
    let mut p = sys::ab_stf_params { shadow: 0.0, midtone: 0.5, highlight: 1.0 };
    unsafe { sys::ab_auto_stf(&stats_to_sys(stats), &cfg, &mut p) };
    StfParams { shadow: p.shadow, midtone: p.midtone, highlight: p.highlight }
}

/// drop-in for core::imaging::stf::apply_stf (stf.rs:89-102)
pub fn apply_stf(hip: &Hip, data: &Array2<f32>, p: &StfParams, st: &ImageStats) -> Result<Vec<u8>> {
    let mut out = vec![0u8; data.len()];
    let sp = sys::ab_stf_params { shadow: p.shadow, midtone: p.midtone, highlight: p.highlight };
    hip.check(unsafe { sys::ab_apply_stf_u8(hip.ctx, &plane(&data.view()), &sp, &stats_to_sys(st), out.as_mut_ptr(), 0) })?;
    Ok(out)
}

/// drop-in for cmd::common::auto_stretch_preview (cmd/common.rs:18-22): stats -> auto_stf -> apply_stf as one device chain.
/// The plane is uploaded once and never read back; only the u8 preview and 80 bytes of scalars return.
pub fn auto_stretch_preview(hip: &Hip, arr: &Array2<f32>) -> Result<(Vec<u8>, ImageStats, StfParams)> {
    let (rows, cols) = arr.dim();
    let bytes = rows * cols * 4;
    let (mut dimg, mut du8) = (std::ptr::null_mut(), std::ptr::null_mut());
    hip.check(unsafe { sys::ab_device_alloc(hip.ctx, bytes, &mut dimg) })?;
    let r = (|| {
        hip.check(unsafe { sys::ab_device_alloc(hip.ctx, rows * cols, &mut du8) })?;
        hip.check(unsafe { sys::ab_upload(hip.ctx, dimg, arr.as_ptr() as *const c_void, bytes) })?;
        let p = sys::ab_plane { data: dimg as *const f32, rows: rows as i64, cols: cols as i64, on_device: 1 };
        let mut st: sys::ab_image_stats = unsafe { std::mem::zeroed() };
        let mut stf = sys::ab_stf_params { shadow: 0.0, midtone: 0.5, highlight: 1.0 };
        hip.check(unsafe { sys::ab_auto_stretch_preview(hip.ctx, std::ptr::null_mut(), &p, 0, std::ptr::null(), du8 as *mut u8, &mut st, &mut stf) })?;
        let mut out = vec![0u8; rows * cols];
        hip.check(unsafe { sys::ab_download(hip.ctx, out.as_mut_ptr() as *mut c_void, du8, rows * cols) })?;
        Ok((out, stats_from_sys(&st), StfParams { shadow: stf.shadow, midtone: stf.midtone, highlight: stf.highlight }))
    })();
    unsafe {
        sys::ab_device_free(hip.ctx, dimg);
        sys::ab_device_free(hip.ctx, du8);
    }
    r
}

/// drop-in for core::alignment::affine::warp_image (affine.rs:663-690)
pub fn warp_image(hip: &Hip, image: &Array2<f32>, t: &crate::core::alignment::affine::AffineTransform, out_rows: usize, out_cols: usize) -> Result<Array2<f32>> {
    let mut out = Array2::<f32>::zeros((out_rows, out_cols));
    let mut po = plane_mut(&mut out);
    let m = [t.a, t.b, t.tx, t.c, t.d, t.ty];
    hip.check(unsafe { sys::ab_warp_image(hip.ctx, &plane(&image.view()), m.as_ptr(), &mut po) })?;
    Ok(out)
}

/// drop-in for core::imaging::background::extract_background (background.rs:55-116), progress and cancel included
pub fn extract_background(
    hip: &Hip,
    image: &Array2<f32>,
    config: &crate::core::imaging::background::BackgroundConfig,
    progress: Option<&ProgressHandle>,
) -> Result<crate::core::imaging::background::BackgroundResult> {
    let start = std::time::Instant::now();
    let (rows, cols) = image.dim();
    let (mut model, mut corrected) = (Array2::<f32>::zeros((rows, cols)), Array2::<f32>::zeros((rows, cols)));
    let cfg = sys::ab_background_config {
        grid_size: config.grid_size,
        poly_degree: config.poly_degree,
        sigma_clip: config.sigma_clip,
        iterations: config.iterations,
        mode: config.mode as i32,
    };
    let mut info: sys::ab_background_info = unsafe { std::mem::zeroed() };
    let (mut pm, mut pc) = (plane_mut(&mut model), plane_mut(&mut corrected));
    hip.with_progress(progress, || hip.check(unsafe { sys::ab_extract_background(hip.ctx, &plane(&image.view()), &cfg, &mut pm, &mut pc, &mut info) }))?;
    if let Some(p) = progress {
        p.emit_complete(); // background.rs:105-107
    }
    Ok(crate::core::imaging::background::BackgroundResult {
        model,
        corrected,
        sample_count: info.sample_count,
        rms_residual: info.rms_residual,
        elapsed_ms: start.elapsed().as_millis() as u64,
    })
}

// ---- several GPUs of one node (include/astroburst_hip.h section (e)) -----------------------------------------------------------
/// One context + one RCCL rank per GPU, each driven by its own thread (what `handleStackAll`'s concurrent commands already
/// are on the host side).  `f(rank, hip, comm)` runs on every rank; the sharded entry points inside enqueue their collectives
/// on the rank's stream.
pub fn on_all_gpus<T: Send>(devices: &[i32], f: impl Fn(usize, &Hip, *mut sys::ab_comm) -> Result<T> + Sync) -> Result<Vec<T>> {
    let hips: Vec<Hip> = devices.iter().map(|&d| Hip::new(d)).collect::<Result<_>>()?;
    let ctxs: Vec<*mut sys::ab_ctx> = hips.iter().map(|h| h.ctx).collect();
    let mut comms = vec![std::ptr::null_mut(); devices.len()];
    hips[0].check(unsafe { sys::ab_comm_init_all(ctxs.as_ptr(), ctxs.len() as i32, comms.as_mut_ptr()) })?;
    struct SendPtr(*mut sys::ab_comm);
    unsafe impl Send for SendPtr {}
    let results = std::thread::scope(|s| {
        let handles: Vec<_> = hips
            .iter()
            .zip(comms.iter().map(|&c| SendPtr(c)))
            .enumerate()
            .map(|(rank, (hip, comm))| {
                let f = &f;
                s.spawn(move || f(rank, hip, comm.0))
            })
            .collect();
        handles.into_iter().map(|h| h.join().expect("rank thread")).collect::<Vec<_>>()
    });
    for c in comms {
        unsafe { sys::ab_comm_destroy(c) };
    }
    results.into_iter().collect()
}

/// stack_images over frames already resident on the GPUs, the per-pixel loop split by ROWS (exact: equals the single-GPU
/// and the reference result bit for bit).  `planes_dev[rank]` = that rank's device copies of all n frames.
pub fn stack_rowband(hip: &Hip, comm: *mut sys::ab_comm, planes_dev: &[sys::ab_plane], config: &StackConfig, out_band_dev: &mut sys::ab_plane_mut) -> Result<u64> {
    let cfg = sys::ab_stack_config { sigma_low: config.sigma_low, sigma_high: config.sigma_high, max_iterations: config.max_iterations as u32, align: 0 };
    let mut rejected = 0u64;
    hip.check(unsafe { sys::ab_stack_sigma_clip_rowband(hip.ctx, comm, planes_dev.as_ptr(), planes_dev.len(), &cfg, out_band_dev, &mut rejected) })?;
    Ok(rejected)
}
