/*
 * ce_oracle.h — CPU oracle for the codec-eval perceptual-metric hot path.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, bench.py's
 * cpu_baseline leg and __graft_entry__.smoke() may load it.  The shipped
 * library (libce_metrics_hip.so) never links, loads or calls anything here.
 *
 * It restates, in plain scalar C, the arithmetic behind the reference's
 *   src/metrics/mod.rs:312-331          calculate_psnr
 *   src/metrics/xyb.rs:33-253           xyb_roundtrip
 *   src/metrics/dssim.rs:78-114         srgb_to_linear / rgb8_to_dssim_image
 *   src/metrics/ssimulacra2.rs:59-100   calculate_ssimulacra2  -> fast-ssim2 0.8.0
 *   src/metrics/dssim.rs:40-71          calculate_dssim        -> dssim-core 3.4.0
 *   src/metrics/butteraugli.rs:45-136   calculate_butteraugli  -> butteraugli 0.9.0
 * (paths relative to /root/reference).
 *
 * Pinning status (see DESIGN.md "Oracle"):
 *   PSNR, XYB roundtrip, sRGB->linear staging: in-tree algorithms, pinned by the
 *     reference's own tests (mod.rs:368-383, xyb.rs:259-301) and the 2^24-colour
 *     histogram table in xyb.rs:15-24.
 *   SSIMULACRA2 / DSSIM / Butteraugli: the arithmetic lives in crates that are NOT
 *     under /root/reference (Cargo.lock:410,356,132) and cannot be built here
 *     (no cargo/rustc, no network).  They are restated from the published
 *     algorithms and checked only against the inequality tests the reference
 *     holds at those call sites.  PARITY UNPINNED for these three.
 */
#ifndef CE_ORACLE_H
#define CE_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* status codes mirror include/ce_metrics.h (kept numerically identical, but this
 * header is deliberately self-contained). */
enum {
    CEO_OK = 0,
    CEO_DIM_MISMATCH = 1, /* Error::DimensionMismatch   (src/error.rs:31-38) */
    CEO_BAD_LENGTH = 2,   /* Error::MetricCalculation "Invalid image size" */
    CEO_TOO_SMALL = 3,    /* image below 8x8 */
    CEO_BACKEND = 4
};

/* ---- sensitivity switches (tests/golden/sensitivity.py ONLY) ---------------------------------------------------
 * The SSIMULACRA2 / DSSIM / Butteraugli restatements make choices the absent crates could make differently
 * (DESIGN.md §2 lists them).  Each switch flips ONE such choice so that its effect on the score can be measured;
 * 0 (the default) is the restatement every test and the device are held to.  Process-global, not thread safe. */
enum ceo_variant_key {
    CEO_V_SSIM2_SRGB_F32_POWF = 0, /* sRGB->linear table by f32 powf instead of f64 pow rounded once */
    CEO_V_SSIM2_HOST_CBRTF = 1,    /* host libm cbrtf instead of the msun two-Halley-steps form */
    CEO_V_SSIM2_IIR_NO_FMA = 2,    /* recursion with separate multiply and add (blur_mode 1 only) */
    CEO_V_DSSIM_LAB_NO_FMA = 3,    /* RGB -> XYZ / Lab affine steps with separate multiply and add */
    CEO_V_DSSIM_F32_FINAL = 4,     /* per-scale scores, weighting and 1/ssim - 1 in f32, widened at the end */
    CEO_V_BA_MALTA_F32 = 5,        /* Malta asymmetry term in f32 instead of f64 */
    CEO_V_BA_LIBM_LOG2 = 6,        /* libm log2f instead of the lineage's FastLog2f in Gamma() */
    CEO_V_SSIM2_F32_POOL = 7,      /* per-pixel map terms summed in f32 per row before the f64 pool */
    CEO_V_BA_L2_EARLY = 8,         /* the HF / MF L2 terms join block_diff_ac between the Malta bands (uhf, L2asym(hf), hf, L2(mf), mf)
                                      instead of after all three: the same in-place accumulations in another order */
    CEO_V_BA_BLUR_FMA = 9,         /* the taps of the long separable blurs (sigma 1.56 .. 7.16) accumulate with fused multiply-add */
    CEO_V_COUNT = 10
};
void ceo_set_variant(int key, int value);
int ceo_get_variant(int key);

/* ---- PSNR: src/metrics/mod.rs:312-331 ---------------------------------- */
int ceo_psnr(const uint8_t *ref, size_t ref_len, const uint8_t *test, size_t test_len,
             size_t width, size_t height, double *out);
/* exact integer sum of squared differences (the quantity the f64 loop holds) */
uint64_t ceo_sse_u8(const uint8_t *a, const uint8_t *b, size_t n);

/* ---- sRGB -> linear, src/metrics/dssim.rs:78-85 ------------------------- */
float ceo_srgb_u8_to_linear(uint8_t v);
/* rgb8_to_dssim_image (dssim.rs:102-114): out is w*h*4 floats RGBA, a = 1.0 */
void ceo_rgb8_to_dssim_image(const uint8_t *rgb, size_t npix, float *rgba_out);

/* ---- XYB roundtrip: src/metrics/xyb.rs:225-253 -------------------------- */
/* the pinned cube root of the roundtrip (glibc 2.35's cbrtf restated; psnr_xyb.c) and the host's, n values each */
float ceo_cbrtf_pinned(float x);
void ceo_cbrtf_compare(const float *x, size_t n, float *pinned, float *host);
int ceo_xyb_roundtrip(const uint8_t *rgb, size_t len, size_t width, size_t height, uint8_t *out);

/* ---- SSIMULACRA2: ssimulacra2.rs:59-100 -> fast-ssim2 ------------------- */
/* blur_mode: 0 = 9-tap FIR (the exact impulse response of the recursive
 * Gaussian) ; 1 = f32 recursive (IIR) form of the libjxl
 * lineage.  Mode 0 is the parity oracle; mode 1 exists to measure the f32
 * noise floor of the recursive form against it. */
int ceo_ssimulacra2(const uint8_t *ref, size_t ref_len, const uint8_t *test, size_t test_len,
                    size_t width, size_t height, int blur_mode, double *out);
/* same, also returning the per-scale averages: avg[scale][c][6] =
 * {ssim_l1, ssim_l4, artifact_l1, artifact_l4, detail_l1, detail_l4}; n_scales out */
int ceo_ssimulacra2_detail(const uint8_t *ref, const uint8_t *test, size_t width, size_t height,
                           int blur_mode, double *avg /* [6][3][6] */, int *n_scales, double *score);
/* score from the averages (Msssim::score) */
double ceo_ssimulacra2_score(const double *avg /* [n_scales][3][6] */, int n_scales);
/* building blocks exposed for plane-level parity tests */
void ceo_ssim2_srgb_lut(float lut[256]);
void ceo_ssim2_blur_taps(float taps[5], double taps_f64[5]);
void ceo_ssim2_linear_planar(const uint8_t *rgb, size_t npix, float *planes /* 3*npix */);
void ceo_ssim2_downscale(const float *in, size_t w, size_t h, float *out /* 3*ceil(w/2)*ceil(h/2) */);
void ceo_ssim2_xyb_positive(const float *lin, size_t npix, float *xyb /* 3*npix */);
void ceo_ssim2_blur_plane(const float *in, size_t w, size_t h, int blur_mode, float *out);

/* ---- DSSIM: dssim.rs:40-71 -> dssim-core -------------------------------- */
int ceo_dssim_rgb8(const uint8_t *ref, size_t ref_len, const uint8_t *test, size_t test_len,
                   size_t width, size_t height, double *out);
/* the reference's own signature: linear RGBA f32 in (dssim.rs:40) */
int ceo_dssim_rgbaf(const float *ref_rgba, size_t rw, size_t rh, const float *test_rgba, size_t tw,
                    size_t th, double *out);
int ceo_dssim_detail(const uint8_t *ref, const uint8_t *test, size_t width, size_t height,
                     double *scale_scores /* [5] */, int *n_scales, double *out);

/* ---- Butteraugli: butteraugli.rs:45-136 -> butteraugli crate ------------ */
int ceo_butteraugli(const uint8_t *ref, size_t ref_len, const uint8_t *test, size_t test_len,
                    size_t width, size_t height, float intensity_target, double *score,
                    double *pnorm3);

#ifdef __cplusplus
}
#endif
#endif
