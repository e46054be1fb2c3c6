//! Safe wrappers over `libce_metrics_hip.so` shaped like the call sites they replace in codec-eval:
//!
//! * [`HipMetrics::calculate_metrics`]  — `EvalSession::calculate_metrics` (src/eval/session.rs:437-497)
//! * [`HipMetrics::evaluate_grid`]      — the `codec x quality` sweep of `evaluate_image` as ONE device batch
//! * [`HipSsim2Reference`]              — `Ssimulacra2Reference::{new, compare}` (crates/codec-iter/src/eval.rs:138-149)
//!   plus `compare_many`, the whole quality loop of `run_eval` in one launch
//! * [`HipSsim2`]                       — `GpuSsim2::{new, compute}` (crates/codec-iter/src/gpu.rs:40-116)
//!
//! One call in flight per context (`&mut self`, like `GpuSsim2::compute`); any number of contexts per device.
pub mod sys;

use std::ffi::CStr;
use std::ptr;

/// The two error kinds the metric path produces in codec-eval (`Error::DimensionMismatch`,
/// `Error::MetricCalculation`, src/error.rs) — convert with `From` at the call site.
#[derive(Debug, thiserror::Error)]
pub enum HipError {
    #[error("Dimension mismatch: expected {expected:?}, got {actual:?}")]
    DimensionMismatch { expected: (usize, usize), actual: (usize, usize) },
    #[error("Metric calculation failed: {metric}: {reason}")]
    MetricCalculation { metric: String, reason: String },
}

/// `MetricConfig` (src/metrics/mod.rs:46-63) as the ABI's mask + flags.
#[derive(Clone, Copy, Debug, Default)]
pub struct Metrics {
    pub dssim: bool,
    pub ssimulacra2: bool,
    pub butteraugli: bool,
    pub psnr: bool,
    pub xyb_roundtrip: bool,
}

impl Metrics {
    fn mask(&self) -> u32 {
        (self.dssim as u32) * sys::CE_METRIC_DSSIM
            | (self.ssimulacra2 as u32) * sys::CE_METRIC_SSIMULACRA2
            | (self.butteraugli as u32) * sys::CE_METRIC_BUTTERAUGLI
            | (self.psnr as u32) * sys::CE_METRIC_PSNR
    }
    fn flags(&self) -> u32 {
        if self.xyb_roundtrip { sys::CE_FLAG_XYB_ROUNDTRIP } else { 0 }
    }
}

/// `MetricResult` (src/metrics/mod.rs:140-149).
#[derive(Clone, Copy, Debug, Default)]
pub struct Scores {
    pub dssim: Option<f64>,
    pub ssimulacra2: Option<f64>,
    pub butteraugli: Option<f64>,
    pub psnr: Option<f64>,
}

impl From<sys::ce_scores> for Scores {
    fn from(s: sys::ce_scores) -> Self {
        let pick = |bit: u32, v: f64| if s.valid & bit != 0 { Some(v) } else { None };
        Scores {
            dssim: pick(sys::CE_METRIC_DSSIM, s.dssim),
            ssimulacra2: pick(sys::CE_METRIC_SSIMULACRA2, s.ssimulacra2),
            butteraugli: pick(sys::CE_METRIC_BUTTERAUGLI, s.butteraugli),
            psnr: pick(sys::CE_METRIC_PSNR, s.psnr),
        }
    }
}

/// A device context (`GpuSsim2::new` / `Drop`, gpu.rs:40-80,118-133).
pub struct HipMetrics {
    ctx: *mut sys::ce_ctx,
}

unsafe impl Send for HipMetrics {}

impl HipMetrics {
    pub fn new(device: i32) -> Result<Self, HipError> {
        // HIP multiplexes streams onto GPU_MAX_HW_QUEUES hardware queues (4 by default); a sweep keeps more streams than
        // that busy.  Read once, when the HIP runtime initialises - so this only helps before the first HIP call.
        if std::env::var_os("GPU_MAX_HW_QUEUES").is_none() {
            std::env::set_var("GPU_MAX_HW_QUEUES", "16");
        }
        let mut ctx = ptr::null_mut();
        let rc = unsafe { sys::ce_ctx_create(device, &mut ctx) };
        if rc != sys::CE_OK {
            return Err(HipError::MetricCalculation { metric: "hip".into(), reason: last_error(ptr::null()) });
        }
        Ok(Self { ctx })
    }

    pub fn device_count() -> i32 {
        unsafe { sys::ce_device_count() }
    }

    fn check(&self, rc: i32, w: u32, h: u32, test_len: usize) -> Result<(), HipError> {
        match rc {
            sys::CE_OK => Ok(()),
            sys::CE_ERR_DIM_MISMATCH => Err(HipError::DimensionMismatch {
                expected: (w as usize, h as usize),
                actual: (if h > 0 { test_len / 3 / h as usize } else { 0 }, h as usize),
            }),
            _ => Err(HipError::MetricCalculation { metric: "hip".into(), reason: last_error(self.ctx) }),
        }
    }

    /// `calculate_metrics(&self, reference, test, width, height)` — one pair, host buffers.
    pub fn calculate_metrics(&mut self, reference: &[u8], test: &[u8], width: u32, height: u32, m: Metrics)
                             -> Result<Scores, HipError> {
        let mut s = sys::ce_scores::default();
        let rc = unsafe {
            sys::ce_eval_pair(self.ctx, reference.as_ptr(), reference.len(), test.as_ptr(), test.len(), width, height,
                              m.mask(), m.flags(), sys::CE_DEFAULT_INTENSITY_TARGET, &mut s)
        };
        self.check(rc, width, height, test.len())?;
        Ok(s.into())
    }

    /// The whole `(codec, quality)` grid of `evaluate_image` (session.rs:375-376) in one call: decode every cell
    /// first, then pass `(reference, decoded, width, height)` per cell.  Cells that share a reference slice share
    /// one device slot.  Per-cell failures come back as `Err` in their position.
    pub fn evaluate_grid(&mut self, cells: &[(&[u8], &[u8], u32, u32)], m: Metrics) -> Result<Vec<Result<Scores, HipError>>, HipError> {
        let descs: Vec<sys::ce_pair_desc> = cells.iter().map(|(r, t, w, h)| sys::ce_pair_desc {
            reference: r.as_ptr(), reference_len: r.len(), test: t.as_ptr(), test_len: t.len(), width: *w, height: *h,
        }).collect();
        let mut out = vec![sys::ce_scores::default(); cells.len()];
        let rc = unsafe {
            sys::ce_eval_batch(self.ctx, descs.len(), descs.as_ptr(), m.mask(), m.flags(), sys::CE_DEFAULT_INTENSITY_TARGET,
                               out.as_mut_ptr())
        };
        if rc != sys::CE_OK {
            return Err(HipError::MetricCalculation { metric: "hip".into(), reason: last_error(self.ctx) });
        }
        Ok(out.iter().zip(cells).map(|(s, (_, t, w, h))| self.check(s.status, *w, *h, t.len()).map(|_| (*s).into())).collect())
    }

    /// `xyb_roundtrip(rgb, width, height)` (src/metrics/xyb.rs:225-253), u8-exact.
    pub fn xyb_roundtrip(&mut self, rgb: &[u8], width: usize, height: usize) -> Result<Vec<u8>, HipError> {
        let mut out = vec![0u8; rgb.len()];
        let rc = unsafe { sys::ce_xyb_roundtrip(self.ctx, rgb.as_ptr(), rgb.len(), width, height, out.as_mut_ptr()) };
        self.check(rc, width as u32, height as u32, rgb.len())?;
        Ok(out)
    }
}

/// Page-locked host bytes (`ce_host_alloc`): a decoder that writes its RGB8 output here lets `evaluate_grid` copy it with
/// the DMA engines straight from this memory, overlapped with the kernels of the previous chunk, instead of staging it
/// through the library's ring with host threads.  Derefs to `[u8]`.
pub struct PinnedBytes {
    p: *mut u8,
    len: usize,
}

unsafe impl Send for PinnedBytes {}

impl HipMetrics {
    pub fn pinned(&self, len: usize) -> Result<PinnedBytes, HipError> {
        let mut p: *mut std::os::raw::c_void = ptr::null_mut();
        let rc = unsafe { sys::ce_host_alloc(self.ctx, len, &mut p) };
        if rc != sys::CE_OK {
            return Err(HipError::MetricCalculation { metric: "hip".into(), reason: last_error(self.ctx) });
        }
        unsafe { ptr::write_bytes(p as *mut u8, 0, len) };
        Ok(PinnedBytes { p: p as *mut u8, len })
    }
}

impl std::ops::Deref for PinnedBytes {
    type Target = [u8];
    fn deref(&self) -> &[u8] {
        unsafe { std::slice::from_raw_parts(self.p, self.len) }
    }
}

impl std::ops::DerefMut for PinnedBytes {
    fn deref_mut(&mut self) -> &mut [u8] {
        unsafe { std::slice::from_raw_parts_mut(self.p, self.len) }
    }
}

impl Drop for PinnedBytes {
    fn drop(&mut self) {
        unsafe { sys::ce_host_free(ptr::null_mut(), self.p as *mut std::os::raw::c_void) }; // valid with or without its context
    }
}

impl Drop for HipMetrics {
    fn drop(&mut self) {
        unsafe { sys::ce_ctx_destroy(self.ctx) } // synchronises its streams first (the order gpu.rs:118-133 spells out)
    }
}

/// `Ssimulacra2Reference::{new, compare}`: the source image stays on the device with its reference-side state.
/// Must be dropped before the `HipMetrics` it was created from.
pub struct HipSsim2Reference<'a> {
    owner: &'a HipMetrics,
    handle: *mut sys::ce_ref,
    width: u32,
    height: u32,
}

impl<'a> HipSsim2Reference<'a> {
    pub fn new(owner: &'a HipMetrics, rgb: &[u8], width: u32, height: u32) -> Result<Self, HipError> {
        let mut handle = ptr::null_mut();
        let rc = unsafe { sys::ce_ref_create(owner.ctx, rgb.as_ptr(), rgb.len(), width, height, 0, &mut handle) };
        owner.check(rc, width, height, rgb.len())?;
        Ok(Self { owner, handle, width, height })
    }

    pub fn compare(&mut self, distorted: &[u8]) -> Result<f64, HipError> {
        let mut s = sys::ce_scores::default();
        let rc = unsafe {
            sys::ce_ref_compare(self.handle, distorted.as_ptr(), distorted.len(), sys::CE_METRIC_SSIMULACRA2,
                                sys::CE_DEFAULT_INTENSITY_TARGET, &mut s)
        };
        self.owner.check(rc, self.width, self.height, distorted.len())?;
        Ok(s.ssimulacra2)
    }

    /// The quality loop `for q in quality_levels { reference.compare(decoded[q]) }` (eval.rs:83-89) as one launch.
    pub fn compare_many(&mut self, distorted: &[&[u8]]) -> Result<Vec<f64>, HipError> {
        let ptrs: Vec<*const u8> = distorted.iter().map(|d| d.as_ptr()).collect();
        let lens: Vec<usize> = distorted.iter().map(|d| d.len()).collect();
        let mut out = vec![sys::ce_scores::default(); distorted.len()];
        let rc = unsafe {
            sys::ce_ref_compare_many(self.handle, ptrs.as_ptr(), lens.as_ptr(), distorted.len() as u32, sys::CE_METRIC_SSIMULACRA2,
                                     sys::CE_DEFAULT_INTENSITY_TARGET, out.as_mut_ptr())
        };
        self.owner.check(rc, self.width, self.height, 0)?;
        out.iter().zip(&lens).map(|(s, l)| self.owner.check(s.status, self.width, self.height, *l).map(|_| s.ssimulacra2)).collect()
    }
}

impl HipSsim2Reference<'_> {
    /// All metrics against the resident reference (the handle keeps every metric's reference-side state: XYB roundtrip,
    /// SSIMULACRA2 XYB pyramid, DSSIM img / mu / blur(img^2) pyramid, Butteraugli PsychoImage).  `scores[i].status`
    /// reports per-item failures.
    pub fn compare_many_metrics(&mut self, distorted: &[&[u8]], mask: u32, intensity_target: f32) -> Result<Vec<sys::ce_scores>, HipError> {
        let ptrs: Vec<*const u8> = distorted.iter().map(|d| d.as_ptr()).collect();
        let lens: Vec<usize> = distorted.iter().map(|d| d.len()).collect();
        let mut out = vec![sys::ce_scores::default(); distorted.len()];
        let rc = unsafe {
            sys::ce_ref_compare_many(self.handle, ptrs.as_ptr(), lens.as_ptr(), distorted.len() as u32, mask, intensity_target, out.as_mut_ptr())
        };
        self.owner.check(rc, self.width, self.height, 0)?;
        Ok(out)
    }

    /// Compares so far that had to (re)build the reference side of (SSIMULACRA2, DSSIM, Butteraugli).
    pub fn reference_builds(&self) -> [u32; 3] {
        let mut b = [0u32; 3];
        unsafe { sys::ce_ref_stats(self.handle, b.as_mut_ptr()) };
        b
    }
}

impl Drop for HipSsim2Reference<'_> {
    fn drop(&mut self) {
        unsafe { sys::ce_ref_destroy(self.handle) }
    }
}

/// `GpuSsim2` look-alike for codec-iter's `Ssim2Backend` (eval.rs:56-92): `new(w, h)`, `compute(&mut self, ref, dis)`.
pub struct HipSsim2 {
    metrics: HipMetrics,
    width: u32,
    height: u32,
}

impl HipSsim2 {
    pub fn new(width: u32, height: u32) -> Result<Self, HipError> {
        Ok(Self { metrics: HipMetrics::new(0)?, width, height })
    }

    pub fn compute(&mut self, reference: &[u8], distorted: &[u8]) -> Result<f64, HipError> {
        let expected = self.width as usize * self.height as usize * 3;
        if reference.len() != expected || distorted.len() != expected {
            return Err(HipError::MetricCalculation {
                metric: "SSIMULACRA2".into(),
                reason: format!("Image size mismatch: expected {} bytes ({}x{}x3), got ref={} dis={}", expected, self.width,
                                self.height, reference.len(), distorted.len()),
            });
        }
        let m = Metrics { ssimulacra2: true, ..Metrics::default() };
        Ok(self.metrics.calculate_metrics(reference, distorted, self.width, self.height, m)?.ssimulacra2.unwrap_or(f64::NAN))
    }

    pub fn dimensions(&self) -> (u32, u32) {
        (self.width, self.height)
    }
}

fn last_error(ctx: *const sys::ce_ctx) -> String {
    let p = unsafe { sys::ce_last_error(ctx) };
    if p.is_null() { String::new() } else { unsafe { CStr::from_ptr(p) }.to_string_lossy().into_owned() }
}
