//! Device-resident planes and the keyed cache that lets consecutive commands hand them to each other in HBM.
//!
//! The reference passes planes from command to command through `GLOBAL_IMAGE_CACHE` (infra/cache.rs:306-310): `blend_channels_cmd`
//! inserts `__composite_r/g/b` (+ `__composite_orig_*`), `calibrate_and_scnr_cmd` / `apply_tone_composite_cmd` /
//! `masked_stretch_composite_cmd` read them back (cmd/helpers.rs:87-147; keys types/constants.rs:189-195).  With host arrays
//! on both sides every command would cross PCIe twice (a 4096 x 4096 plane is 67 MB: ~1.3 ms each way at the 50 GB/s a pinned
//! Gen5 x16 copy sustains -- longer than any kernel that touches it).  `DEVICE_CACHE` mirrors the pinned keys of that cache with
//! `DevicePlane`s: a command that produced a composite leaves it in HBM under the same key, the next command picks it up there,
//! and the host copy the rest of the application expects is materialised lazily (`host()`), once, when something asks for it.
//!
//! (Reviewed source: this image has no Rust toolchain.  The C side of every call is exercised by the GPU tests through ctypes.)
use anyhow::{anyhow, Result};
use ndarray::{Array2, ArrayView2};
use std::collections::HashMap;
use std::ffi::c_void;
use std::sync::{Arc, Mutex, OnceLock, RwLock};

use super::sys;
use super::Hip;
use crate::types::image::ImageStats;

/// What a wrapper reads: a host array (staged through the library's pinned buffer) or a plane already in HBM.
pub trait PlaneSrc {
    fn ab(&self) -> sys::ab_plane;
    fn dims(&self) -> (usize, usize);
}
/// What a wrapper writes.
pub trait PlaneDst {
    fn ab_mut(&mut self) -> sys::ab_plane_mut;
}

impl PlaneSrc for Array2<f32> {
    fn ab(&self) -> sys::ab_plane {
        let s = self.as_slice().expect("contiguous"); // the precondition of combine.rs:154
        sys::ab_plane { data: s.as_ptr(), rows: self.nrows() as i64, cols: self.ncols() as i64, on_device: 0 }
    }
    fn dims(&self) -> (usize, usize) {
        self.dim()
    }
}
impl<'a> PlaneSrc for ArrayView2<'a, f32> {
    fn ab(&self) -> sys::ab_plane {
        let s = self.as_slice().expect("contiguous");
        sys::ab_plane { data: s.as_ptr(), rows: self.nrows() as i64, cols: self.ncols() as i64, on_device: 0 }
    }
    fn dims(&self) -> (usize, usize) {
        self.dim()
    }
}
impl<T: PlaneSrc + ?Sized> PlaneSrc for &T {
    fn ab(&self) -> sys::ab_plane {
        (**self).ab()
    }
    fn dims(&self) -> (usize, usize) {
        (**self).dims()
    }
}
impl PlaneDst for Array2<f32> {
    fn ab_mut(&mut self) -> sys::ab_plane_mut {
        let (rows, cols) = self.dim();
        sys::ab_plane_mut { data: self.as_mut_ptr(), rows: rows as i64, cols: cols as i64, on_device: 0 }
    }
}

/// A rows x cols f32 plane in HBM.  Device memory belongs to the device, not to the context that allocated it: it is released
/// through a process-wide allocator context, so a plane may outlive the thread (and thread-local `Hip`) that produced it.
pub struct DevicePlane {
    ptr: *mut c_void,
    rows: usize,
    cols: usize,
}
unsafe impl Send for DevicePlane {}
unsafe impl Sync for DevicePlane {} // read-only once published through the cache

fn allocator() -> &'static Mutex<Hip> {
    static A: OnceLock<Mutex<Hip>> = OnceLock::new();
    A.get_or_init(|| Mutex::new(Hip::new(0).expect("no gfx950 (MI355X) device")))
}

impl DevicePlane {
    pub fn alloc(rows: usize, cols: usize) -> Result<Self> {
        let a = allocator().lock().unwrap();
        let mut ptr = std::ptr::null_mut();
        a.check(unsafe { sys::ab_device_alloc(a.ctx, rows * cols * 4, &mut ptr) })?;
        Ok(Self { ptr, rows, cols })
    }
    /// upload (one pinned-staged copy on the calling context's stream)
    pub fn from_host(hip: &Hip, a: &Array2<f32>) -> Result<Self> {
        let p = Self::alloc(a.nrows(), a.ncols())?;
        let s = a.as_slice().ok_or_else(|| anyhow!("non-contiguous array"))?;
        hip.check(unsafe { sys::ab_upload(hip.ctx, p.ptr, s.as_ptr() as *const c_void, s.len() * 4) })?;
        Ok(p)
    }
    pub fn to_host(&self, hip: &Hip) -> Result<Array2<f32>> {
        let mut out = Array2::<f32>::zeros((self.rows, self.cols));
        hip.check(unsafe { sys::ab_download(hip.ctx, out.as_mut_ptr() as *mut c_void, self.ptr, self.rows * self.cols * 4) })?;
        Ok(out)
    }
    pub fn as_ptr(&self) -> *const f32 {
        self.ptr as *const f32
    }
}
impl Drop for DevicePlane {
    fn drop(&mut self) {
        let a = allocator().lock().unwrap();
        unsafe { sys::ab_device_free(a.ctx, self.ptr) };
    }
}
impl PlaneSrc for DevicePlane {
    fn ab(&self) -> sys::ab_plane {
        sys::ab_plane { data: self.ptr as *const f32, rows: self.rows as i64, cols: self.cols as i64, on_device: 1 }
    }
    fn dims(&self) -> (usize, usize) {
        (self.rows, self.cols)
    }
}
impl PlaneDst for DevicePlane {
    fn ab_mut(&mut self) -> sys::ab_plane_mut {
        sys::ab_plane_mut { data: self.ptr as *mut f32, rows: self.rows as i64, cols: self.cols as i64, on_device: 1 }
    }
}

/// One cached composite: the plane in HBM, its statistics (the reference caches them with the array, infra/cache.rs:20-40), and
/// the host copy once somebody has asked for it.
pub struct DeviceEntry {
    pub plane: DevicePlane,
    pub stats: ImageStats,
    host: OnceLock<Arc<Array2<f32>>>,
}
impl DeviceEntry {
    pub fn host(&self, hip: &Hip) -> Result<Arc<Array2<f32>>> {
        if let Some(h) = self.host.get() {
            return Ok(Arc::clone(h));
        }
        let a = Arc::new(self.plane.to_host(hip)?);
        Ok(Arc::clone(self.host.get_or_init(|| a)))
    }
}

/// The device-side twin of GLOBAL_IMAGE_CACHE for its PINNED keys (`__composite*`, `__wizard_ch_*`, `__star_mask`:
/// infra/cache.rs:90-92 -- the entries the LRU never evicts, i.e. exactly the ones commands hand to each other).
pub struct DeviceCache {
    map: RwLock<HashMap<String, Arc<DeviceEntry>>>,
}
impl DeviceCache {
    pub fn is_pinned(key: &str) -> bool {
        key.starts_with("__composite") || key.starts_with("__wizard_ch_") || key == "__star_mask"
    }
    pub fn insert(&self, key: &str, plane: DevicePlane, stats: ImageStats) -> Arc<DeviceEntry> {
        let e = Arc::new(DeviceEntry { plane, stats, host: OnceLock::new() });
        self.map.write().unwrap().insert(key.to_string(), Arc::clone(&e));
        e
    }
    /// the same entry under a second key (insert_composite_and_orig, cmd/helpers.rs:127-147: `__composite_orig_r` and
    /// `__composite_r` share one Arc there too)
    pub fn alias(&self, key: &str, e: &Arc<DeviceEntry>) {
        self.map.write().unwrap().insert(key.to_string(), Arc::clone(e));
    }
    pub fn get(&self, key: &str) -> Option<Arc<DeviceEntry>> {
        self.map.read().unwrap().get(key).cloned()
    }
    pub fn remove(&self, key: &str) {
        self.map.write().unwrap().remove(key);
    }
    pub fn clear(&self) {
        self.map.write().unwrap().clear();
    }
}
pub static DEVICE_CACHE: std::sync::LazyLock<DeviceCache> = std::sync::LazyLock::new(|| DeviceCache { map: RwLock::new(HashMap::new()) });

/// cmd/helpers.rs:127-147 on the device cache
pub fn insert_composite_and_orig(r: DevicePlane, g: DevicePlane, b: DevicePlane, sr: ImageStats, sg: ImageStats, sb: ImageStats) {
    use crate::types::constants::*;
    for (key, orig, p, s) in [(COMPOSITE_KEY_R, COMPOSITE_ORIG_R, r, sr), (COMPOSITE_KEY_G, COMPOSITE_ORIG_G, g, sg), (COMPOSITE_KEY_B, COMPOSITE_ORIG_B, b, sb)] {
        let e = DEVICE_CACHE.insert(orig, p, s);
        DEVICE_CACHE.alias(key, &e);
    }
}
/// cmd/helpers.rs:113-125
pub fn insert_composite_rgb(r: DevicePlane, g: DevicePlane, b: DevicePlane, sr: ImageStats, sg: ImageStats, sb: ImageStats) {
    use crate::types::constants::*;
    DEVICE_CACHE.insert(COMPOSITE_KEY_R, r, sr);
    DEVICE_CACHE.insert(COMPOSITE_KEY_G, g, sg);
    DEVICE_CACHE.insert(COMPOSITE_KEY_B, b, sb);
}
/// cmd/helpers.rs:87-98, with the reference's error strings
pub fn load_composite_rgb() -> Result<(Arc<DeviceEntry>, Arc<DeviceEntry>, Arc<DeviceEntry>)> {
    use crate::types::constants::*;
    let r = DEVICE_CACHE.get(COMPOSITE_KEY_R).ok_or_else(|| anyhow!("Composite R not in cache"))?;
    let g = DEVICE_CACHE.get(COMPOSITE_KEY_G).ok_or_else(|| anyhow!("Composite G not in cache"))?;
    let b = DEVICE_CACHE.get(COMPOSITE_KEY_B).ok_or_else(|| anyhow!("Composite B not in cache"))?;
    Ok((r, g, b))
}
/// cmd/helpers.rs:100-111
pub fn load_orig_or_composite() -> Result<(Arc<DeviceEntry>, Arc<DeviceEntry>, Arc<DeviceEntry>)> {
    use crate::types::constants::*;
    let pick = |orig: &str, key: &str, name: &str| DEVICE_CACHE.get(orig).or_else(|| DEVICE_CACHE.get(key)).ok_or_else(|| anyhow!("Composite {} not in cache", name));
    Ok((pick(COMPOSITE_ORIG_R, COMPOSITE_KEY_R, "R")?, pick(COMPOSITE_ORIG_G, COMPOSITE_KEY_G, "G")?, pick(COMPOSITE_ORIG_B, COMPOSITE_KEY_B, "B")?))
}
