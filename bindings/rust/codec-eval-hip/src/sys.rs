//! Raw declarations of `include/ce_metrics.h`.  One `pub fn` per exported symbol, same order as the header.
#![allow(non_camel_case_types)]
use std::os::raw::{c_char, c_double, c_float, c_int, c_void};

#[repr(C)]
pub struct ce_ctx {
    _private: [u8; 0],
}
#[repr(C)]
pub struct ce_batch {
    _private: [u8; 0],
}
#[repr(C)]
pub struct ce_ref {
    _private: [u8; 0],
}
#[repr(C)]
pub struct ce_lut {
    _private: [u8; 0],
}

pub const CE_OK: c_int = 0;
pub const CE_ERR_DIM_MISMATCH: c_int = 1;
pub const CE_ERR_BAD_LENGTH: c_int = 2;
pub const CE_ERR_TOO_SMALL: c_int = 3;
pub const CE_ERR_BACKEND: c_int = 4;
pub const CE_ERR_INVALID_ARG: c_int = 5;

pub const CE_METRIC_DSSIM: u32 = 1 << 0;
pub const CE_METRIC_SSIMULACRA2: u32 = 1 << 1;
pub const CE_METRIC_BUTTERAUGLI: u32 = 1 << 2;
pub const CE_METRIC_PSNR: u32 = 1 << 3;
pub const CE_FLAG_XYB_ROUNDTRIP: u32 = 1 << 0;
pub const CE_DEFAULT_INTENSITY_TARGET: c_float = 80.0;

pub const CE_PIXEL_RGB8: c_int = 0;
pub const CE_PIXEL_RGBA8: c_int = 1;
pub const CE_PIXEL_RGB16_10BIT: c_int = 2;
pub const CE_PIXEL_RGBA16_10BIT: c_int = 3;

/// `ce_scores` (40 bytes): a score is meaningful iff its bit is set in `valid`.
#[repr(C)]
#[derive(Clone, Copy, Debug, Default)]
pub struct ce_scores {
    pub dssim: c_double,
    pub ssimulacra2: c_double,
    pub butteraugli: c_double,
    pub psnr: c_double,
    pub valid: u32,
    pub status: i32,
}

/// `ce_pair_desc` (40 bytes): one item of the (image x codec x quality) grid, host pointers.
#[repr(C)]
#[derive(Clone, Copy, Debug)]
pub struct ce_pair_desc {
    pub reference: *const u8,
    pub reference_len: usize,
    pub test: *const u8,
    pub test_len: usize,
    pub width: u32,
    pub height: u32,
}

extern "C" {
    pub fn ce_version() -> *const c_char;
    pub fn ce_device_count() -> c_int;
    pub fn ce_ctx_create(device: c_int, out: *mut *mut ce_ctx) -> c_int;
    pub fn ce_ctx_create_on_stream(device: c_int, hip_stream: *mut c_void, out: *mut *mut ce_ctx) -> c_int;
    pub fn ce_ctx_destroy(ctx: *mut ce_ctx);
    pub fn ce_ctx_synchronize(ctx: *mut ce_ctx) -> c_int;
    pub fn ce_ctx_stream(ctx: *mut ce_ctx) -> *mut c_void;
    pub fn ce_last_error(ctx: *const ce_ctx) -> *const c_char;
    pub fn ce_calculate_psnr(ctx: *mut ce_ctx, reference: *const u8, reference_len: usize, test: *const u8, test_len: usize,
                             width: usize, height: usize, out: *mut c_double) -> c_int;
    pub fn ce_calculate_ssimulacra2(ctx: *mut ce_ctx, reference: *const u8, reference_len: usize, test: *const u8, test_len: usize,
                                    width: usize, height: usize, out: *mut c_double) -> c_int;
    pub fn ce_calculate_dssim(ctx: *mut ce_ctx, reference: *const u8, reference_len: usize, test: *const u8, test_len: usize,
                              width: usize, height: usize, out: *mut c_double) -> c_int;
    pub fn ce_calculate_butteraugli(ctx: *mut ce_ctx, reference: *const u8, reference_len: usize, test: *const u8, test_len: usize,
                                    width: usize, height: usize, intensity_target: c_float, out: *mut c_double) -> c_int;
    pub fn ce_xyb_roundtrip(ctx: *mut ce_ctx, rgb: *const u8, rgb_len: usize, width: usize, height: usize, out: *mut u8) -> c_int;
    pub fn ce_rgb8_to_dssim_image(ctx: *mut ce_ctx, rgb: *const u8, rgb_len: usize, width: usize, height: usize,
                                  out_rgba_f32: *mut c_float) -> c_int;
    pub fn ce_eval_pair(ctx: *mut ce_ctx, reference: *const u8, reference_len: usize, test: *const u8, test_len: usize,
                        width: u32, height: u32, metric_mask: u32, flags: u32, intensity_target: c_float,
                        out: *mut ce_scores) -> c_int;
    pub fn ce_eval_batch(ctx: *mut ce_ctx, n: usize, pairs: *const ce_pair_desc, metric_mask: u32, flags: u32,
                         intensity_target: c_float, out: *mut ce_scores) -> c_int;
    pub fn ce_estimate_batch_bytes(width: u32, height: u32, n_refs: u32, n_pairs: u32, metric_mask: u32) -> usize;
    pub fn ce_ctx_memory_info(ctx: *mut ce_ctx, free_bytes: *mut usize, total_bytes: *mut usize) -> c_int;
    pub fn ce_host_alloc(ctx: *mut ce_ctx, bytes: usize, out: *mut *mut c_void) -> c_int;
    pub fn ce_host_free(ctx: *mut ce_ctx, p: *mut c_void) -> c_int;
    pub fn ce_eval_batch_lut(ctx: *mut ce_ctx, n: usize, pairs: *const ce_pair_desc, test_luts: *const *const ce_lut, metric_mask: u32,
                             flags: u32, intensity_target: c_float, out: *mut ce_scores) -> c_int;
    pub fn ce_batch_create(ctx: *mut ce_ctx, width: u32, height: u32, max_refs: u32, max_pairs: u32, out: *mut *mut ce_batch) -> c_int;
    pub fn ce_batch_destroy(b: *mut ce_batch);
    pub fn ce_batch_set_reference(b: *mut ce_batch, ref_index: u32, rgb: *const u8, len: usize) -> c_int;
    pub fn ce_batch_set_test(b: *mut ce_batch, pair_index: u32, ref_index: u32, rgb: *const u8, len: usize) -> c_int;
    pub fn ce_batch_set_reference_fmt(b: *mut ce_batch, ref_index: u32, pixels: *const c_void, len: usize, format: c_int) -> c_int;
    pub fn ce_batch_set_test_fmt(b: *mut ce_batch, pair_index: u32, ref_index: u32, pixels: *const c_void, len: usize,
                                 format: c_int) -> c_int;
    pub fn ce_lut_create(ctx: *mut ce_ctx, table: *const u8, table_len: usize, out: *mut *mut ce_lut) -> c_int;
    pub fn ce_lut_destroy(lut: *mut ce_lut);
    pub fn ce_batch_set_reference_lut(b: *mut ce_batch, ref_index: u32, pixels: *const c_void, len: usize, format: c_int,
                                      lut: *const ce_lut) -> c_int;
    pub fn ce_batch_set_test_lut(b: *mut ce_batch, pair_index: u32, ref_index: u32, pixels: *const c_void, len: usize, format: c_int,
                                 lut: *const ce_lut) -> c_int;
    pub fn ce_batch_reference_slab(b: *mut ce_batch) -> *mut c_void;
    pub fn ce_batch_test_slab(b: *mut ce_batch) -> *mut c_void;
    pub fn ce_batch_bind_pair(b: *mut ce_batch, pair_index: u32, ref_index: u32) -> c_int;
    pub fn ce_batch_run(b: *mut ce_batch, n_pairs: u32, metric_mask: u32, flags: u32, intensity_target: c_float,
                        out: *mut ce_scores) -> c_int;
    pub fn ce_batch_launch(b: *mut ce_batch, n_pairs: u32, metric_mask: u32, flags: u32, intensity_target: c_float) -> c_int;
    pub fn ce_batch_collect(b: *mut ce_batch, n_pairs: u32, out: *mut ce_scores) -> c_int;
    pub fn ce_batch_butteraugli_pnorm3(b: *mut ce_batch, n_pairs: u32, out: *mut c_double) -> c_int;
    pub fn ce_ref_create(ctx: *mut ce_ctx, reference: *const u8, reference_len: usize, width: u32, height: u32, flags: u32,
                         out: *mut *mut ce_ref) -> c_int;
    pub fn ce_ref_compare(r: *mut ce_ref, test: *const u8, test_len: usize, metric_mask: u32, intensity_target: c_float,
                          out: *mut ce_scores) -> c_int;
    pub fn ce_ref_compare_many(r: *mut ce_ref, tests: *const *const u8, test_lens: *const usize, n_tests: u32, metric_mask: u32,
                               intensity_target: c_float, out: *mut ce_scores) -> c_int;
    pub fn ce_ref_stats(r: *const ce_ref, builds: *mut u32) -> c_int;
    pub fn ce_ref_destroy(r: *mut ce_ref);
    pub fn ce_prof_enable(ctx: *mut ce_ctx, on: c_int) -> c_int;
    pub fn ce_prof_filter(ctx: *mut ce_ctx, substring: *const c_char) -> c_int;
    pub fn ce_prof_reset(ctx: *mut ce_ctx) -> c_int;
    pub fn ce_prof_count(ctx: *mut ce_ctx) -> c_int;
    pub fn ce_prof_get(ctx: *mut ce_ctx, index: c_int, name: *mut *const c_char, launches: *mut u64, total_ms: *mut c_double) -> c_int;
    pub fn ce_timer_start(ctx: *mut ce_ctx) -> c_int;
    pub fn ce_timer_stop(ctx: *mut ce_ctx, elapsed_ms: *mut c_double) -> c_int;
    pub fn ce_debug_ssim2_planes(b: *mut ce_batch, scale: c_int, which: c_int, channel: c_int, out: *mut c_float, out_floats: usize,
                                 w_out: *mut u32, h_out: *mut u32) -> c_int;
    pub fn ce_debug_ssim2_limit_scales(b: *mut ce_batch, max_scales: c_int) -> c_int;
    pub fn ce_debug_ssim2_averages(b: *mut ce_batch, pair_index: u32, avg: *mut c_double, n_scales: *mut c_int) -> c_int;
    pub fn ce_debug_ssim2_occupancy(which: c_int) -> c_int;
    pub fn ce_debug_cbrt_sweep(ctx: *mut ce_ctx, first_bits: u32, count: u64, mismatches: *mut u64, slow_path: *mut u64) -> c_int;
    pub fn ce_debug_div_sweep(ctx: *mut ce_ctx, seed: u64, count: u64, mismatches: *mut u64) -> c_int;
    pub fn ce_debug_calibrate_traffic(ctx: *mut ce_ctx, bytes: usize) -> c_int;
}
