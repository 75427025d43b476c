//! `stack` (src-tauri/src/cmd/stacking/combine.rs:76-135) re-pointed at libastroburst_hip.so.
//!
//! The command's signature, its progress events, its JSON keys and its error strings are the reference's; the two compute
//! calls -- `stack_from_paths` (core/stacking/calibration.rs:297-318: load + stack_images) and `compute_image_stats` -- go
//! through `crate::hip`.  Drop this body in place of the original one.  (Reviewed source: not compiled in the repository that
//! ships it.)
use serde_json::json;

use crate::cmd::common::{blocking_cmd, render_asinh_and_save, resolve_output_dir};
use crate::core::stacking::calibration::load_fits_image;
use crate::hip;
use crate::infra::progress::ProgressHandle;
use crate::types::constants::*;
use crate::types::stacking::StackConfig;

#[tauri::command]
pub async fn stack(
    app: tauri::AppHandle,
    paths: Vec<String>,
    output_dir: String,
    sigma_low: Option<f32>,
    sigma_high: Option<f32>,
    max_iterations: Option<usize>,
    align: Option<bool>,
    name: Option<String>,
) -> Result<serde_json::Value, String> {
    let frame_count = paths.len() as u64;
    let progress = ProgressHandle::new(&app, EVENT_STACK_PROGRESS, frame_count + 2);
    let progress_clone = progress.clone();

    blocking_cmd!({
        resolve_output_dir(&output_dir)?;

        let config = StackConfig {
            sigma_low: sigma_low.unwrap_or(3.0),
            sigma_high: sigma_high.unwrap_or(3.0),
            max_iterations: max_iterations.unwrap_or(5),
            align: align.unwrap_or(true),
        };

        // stack_from_paths (calibration.rs:297-318): the loader stays, the stack moves to the GPU
        let frames = paths
            .iter()
            .map(|p| load_fits_image(p).map(|r| r.data))
            .collect::<anyhow::Result<Vec<_>>>()?;
        let (result, stats) = hip::with_hip(|h| {
            let result = hip::stack_images(h, &frames, &config)?;
            let stats = hip::compute_image_stats(h, &result.image)?;
            Ok((result, stats))
        })?;

        progress_clone.tick_with_stage(STAGE_RENDER);

        let stem = name.as_deref().unwrap_or("stacked");
        let (png_path, fits_path) = render_asinh_and_save(&result.image, &output_dir, stem, true)?;
        let (rows, cols) = result.image.dim();

        progress_clone.tick_with_stage(STAGE_SAVE);
        progress_clone.emit_complete();

        Ok(json!({
            RES_PNG_PATH: png_path,
            RES_FITS_PATH: fits_path,
            RES_DIMENSIONS: [cols, rows],
            RES_FRAME_COUNT: result.frame_count,
            RES_REJECTED_PIXELS: result.rejected_pixels,
            RES_OFFSETS: result.offsets.iter().map(|(dy, dx)| json!({RES_DY: dy, RES_DX: dx})).collect::<Vec<_>>(),
            RES_STATS: {
                RES_MIN: stats.min,
                RES_MAX: stats.max,
                RES_MEAN: stats.mean,
                RES_SIGMA: stats.sigma,
            },
        }))
    })
}
