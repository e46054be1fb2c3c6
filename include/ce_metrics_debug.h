/*
 * ce_metrics_debug.h - test hooks of libce_metrics_hip.so.
 *
 * NOT part of the drop-in boundary (include/ce_metrics.h): nothing here replaces a reference item.  The parity tests use
 * these to compare intermediate planes with the oracle, to sweep hand-expanded arithmetic against the compiler's on the
 * device, and to calibrate the rocprofv3 traffic counters.
 */
#ifndef CE_METRICS_DEBUG_H
#define CE_METRICS_DEBUG_H

#include "ce_metrics.h"

#ifdef __cplusplus
extern "C" {
#endif

/* ---- stage outputs of the SSIMULACRA2 pipeline (plane-level parity) --- */
/* Copies the device planes of pair 0 of the last ce_batch_run at `scale` to host.
 * which: 0 = linear RGB ref, 1 = linear RGB test, 2 = XYB ref, 3 = XYB test (3*h*w floats,
 * planar), 4 = the 5 row-blurred planes of channel `channel` (5*h*w floats). */
int ce_debug_ssim2_planes(ce_batch *b, int scale, int which, int channel, float *out, size_t out_floats,
                          uint32_t *w_out, uint32_t *h_out);
/* stop the pyramid after max_scales levels (so the level-0 XYB / row-blur planes survive the run) */
int ce_debug_ssim2_limit_scales(ce_batch *b, int max_scales);
/* avg[scale][c][6] of pair `pair_index` from the last run (ssim l1,l4, artifact l1,l4, detail l1,l4) */
int ce_debug_ssim2_averages(ce_batch *b, uint32_t pair_index, double *avg /* [6][3][6] */, int *n_scales);
/* The XYB front end's cube root has a division-free fast form that falls back to the reference form
 * (msun cbrtf: two f64 Halley steps) near f32 rounding boundaries.  This runs the f32 bit patterns
 * [first_bits, first_bits + count) through both on the device and reports how many results differ
 * (must be 0) and how many inputs took the fallback. */
/* resident workgroups per CU that the HIP runtime reports for the SSIMULACRA2 row pass (0) / column pass (1) */
int ce_debug_ssim2_occupancy(int which);
int ce_debug_cbrt_sweep(ce_ctx *ctx, uint32_t first_bits, uint64_t count, uint64_t *mismatches, uint64_t *slow_path);
/* Butteraugli's Malta pre-scaling forms two quotients by one denominator with a hand-expanded division that refines the
 * reciprocal once (butteraugli.hip: div2_shared_rcp).  This runs `count` pseudo-random operand triples through it and
 * through operator/ on the device and reports how many quotients differ (must be 0). */
int ce_debug_div_sweep(ce_ctx *ctx, uint64_t seed, uint64_t count, uint64_t *mismatches);
/* Known-byte-count streams for calibrating the rocprofv3 traffic counters: reads `bytes` of a scratch buffer with 1, 4
 * and 16 bytes per lane and writes it with 4 and 16 (kernels k_calib_read<W> / k_calib_write<W>), so a PMC pass can
 * measure FETCH_SIZE's / WRITE_SIZE's correction factor per access width (profiles/make_traffic.py). */
int ce_debug_calibrate_traffic(ce_ctx *ctx, size_t bytes);

#ifdef __cplusplus
}
#endif
#endif
