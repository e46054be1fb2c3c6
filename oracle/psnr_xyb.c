/*
 * ORACLE (test infrastructure only — see ce_oracle.h).
 *
 * PSNR, sRGB->linear staging and the XYB roundtrip: the three pieces of the hot
 * path whose arithmetic is in the reference tree itself.  Each function follows
 * the cited Rust lines operation by operation.  Compile with -ffp-contract=off:
 * rustc never contracts a*b+c into an fma unless the source says mul_add, and
 * none of the cited code does.
 *
 * libm notes: Rust's f32::powf / f32::cbrt / f32::round lower to the platform
 * libm's powf / cbrtf / roundf on linux-gnu.  powf and roundf are called from
 * the host libm here (the device builds its sRGB tables from the same host
 * powf, ce_tables.cpp, so both sides follow the host).  cbrtf is PINNED: its
 * input is a continuous f32, so no table can stand in for it, and glibc
 * changed the routine (2.35: frexp / quadratic seed / one f64 Halley step;
 * later releases ship a correctly rounded one).  Oracle and device both
 * execute the glibc-2.35 algorithm restated with IEEE basic operations
 * (ceo_cbrtf_pinned below, cbrtf_glibc_pos in codec-eval_amd/csrc/xyb.hip);
 * that algorithm reproduces the reference's 2^24-colour known-answer table
 * (xyb.rs:15-24).  tests/test_oracle_pinning.py reports how the host's own
 * cbrtf compares.
 */
#include "ce_oracle.h"

#include <math.h>

/* sensitivity switches (ce_oracle.h): all 0 unless tests/golden/sensitivity.py sets one */
int ceo_variant[CEO_V_COUNT];
void ceo_set_variant(int key, int value) { if (key >= 0 && key < CEO_V_COUNT) ceo_variant[key] = value; }
int ceo_get_variant(int key) { return key >= 0 && key < CEO_V_COUNT ? ceo_variant[key] : 0; }

/* ---------------------------------------------------------------- PSNR ---- */

/* src/metrics/mod.rs:316-322 — the loop sums (r-t)^2 in f64.  Every partial sum
 * is an integer < 2^53, so the f64 sum is exact and equals this u64 sum. */
uint64_t ceo_sse_u8(const uint8_t *a, const uint8_t *b, size_t n)
{
    uint64_t s = 0;
    for (size_t i = 0; i < n; i++) {
        int d = (int)a[i] - (int)b[i];
        s += (uint64_t)(d * d);
    }
    return s;
}

/* src/metrics/mod.rs:312-331.  The reference asserts (panics) on bad lengths
 * (:313-314); the oracle reports them as status codes instead. */
int ceo_psnr(const uint8_t *ref, size_t ref_len, const uint8_t *test, size_t test_len,
             size_t width, size_t height, double *out)
{
    if (ref_len != test_len) return CEO_DIM_MISMATCH; /* :313 */
    if (ref_len != width * height * 3) return CEO_BAD_LENGTH; /* :314 */
    double mse_sum = 0.0;                                /* :316 */
    double pixel_count = (double)(width * height * 3);   /* :317 */
    for (size_t i = 0; i < ref_len; i++) {               /* :319-322 */
        double diff = (double)ref[i] - (double)test[i];
        mse_sum += diff * diff;
    }
    double mse = mse_sum / pixel_count;                  /* :324 */
    if (mse == 0.0)                                      /* :326-327 */
        *out = INFINITY;
    else
        *out = 10.0 * log10(255.0 * 255.0 / mse);        /* :329 */
    return CEO_OK;
}

/* ------------------------------------------------------- sRGB -> linear --- */

/* src/metrics/dssim.rs:78-85 (twin: src/eval/helpers.rs:60-67) */
float ceo_srgb_u8_to_linear(uint8_t v)
{
    float s = (float)v / 255.0f;
    if (s <= 0.04045f)
        return s / 12.92f;
    return powf((s + 0.055f) / 1.055f, 2.4f);
}

/* src/metrics/dssim.rs:102-114 */
void ceo_rgb8_to_dssim_image(const uint8_t *rgb, size_t npix, float *rgba_out)
{
    for (size_t i = 0; i < npix; i++) {
        rgba_out[4 * i + 0] = ceo_srgb_u8_to_linear(rgb[3 * i + 0]);
        rgba_out[4 * i + 1] = ceo_srgb_u8_to_linear(rgb[3 * i + 1]);
        rgba_out[4 * i + 2] = ceo_srgb_u8_to_linear(rgb[3 * i + 2]);
        rgba_out[4 * i + 3] = 1.0f;
    }
}

/* -------------------------------------------------------- XYB roundtrip --- */

/* src/metrics/xyb.rs:33-56 — constants, written as the same decimal literals so
 * the f32 roundings are identical. */
static const float XYB_OPSIN_ABSORBANCE_MATRIX[9] = {
    0.30f, 0.622f, 0.078f,
    0.23f, 0.692f, 0.078f,
    0.24342269f, 0.20476744f, 0.55180987f,
};
static const float XYB_OPSIN_ABSORBANCE_BIAS[3] = {0.0037930733f, 0.0037930733f, 0.0037930733f};
static const float XYB_NEG_OPSIN_ABSORBANCE_BIAS_CBRT[3] = {-0.15595412f, -0.15595412f, -0.15595412f};
static const float INV_OPSIN_MATRIX[9] = {
    11.031567f, -9.866944f, -0.164623f,
    -3.254147f, 4.41877f, -0.164623f,
    -3.658851f, 2.712923f, 1.945928f,
};

/* xyb.rs:60-66 */
static float srgb_to_linear_f32(float v)
{
    if (v <= 0.04045f) return v / 12.92f;
    return powf((v + 0.055f) / 1.055f, 2.4f);
}
/* xyb.rs:70-76 */
static float linear_to_srgb_f32(float v)
{
    if (v <= 0.0031308f) return v * 12.92f;
    return 1.055f * powf(v, 1.0f / 2.4f) - 0.055f;
}
/* xyb.rs:80-82 */
static float srgb_u8_to_linear(uint8_t v) { return srgb_to_linear_f32((float)v / 255.0f); }
/* xyb.rs:86-88 — f32::clamp, then *255, f32::round (half away from zero), `as u8` saturates */
static uint8_t linear_to_srgb_u8(float v)
{
    float c = v < 0.0f ? 0.0f : (v > 1.0f ? 1.0f : v);
    float r = roundf(linear_to_srgb_f32(c) * 255.0f);
    if (!(r > 0.0f)) return 0;
    if (r > 255.0f) return 255;
    return (uint8_t)r;
}
/* glibc 2.35 sysdeps/ieee754/flt-32/s_cbrtf.c for positive normal x, IEEE basic
 * operations only (the pinned cbrtf, see the header). */
float ceo_cbrtf_pinned(float x)
{
    if (!(x > 0.0f) || !isfinite(x) || x < 1.17549435e-38f) return cbrtf(x); /* never reached from the opsin values */
    union { float f; uint32_t u; } b, m, o;
    b.f = x;
    const int xe = (int)(b.u >> 23) - 126;                 /* frexpf exponent */
    m.u = (b.u & 0x007fffffu) | 0x3f000000u;               /* mantissa in [0.5, 1) */
    const float xm = m.f;
    const float u = (float)(0.492659620528969547 + (0.697570460207922770 - 0.191502161678719066 * (double)xm) * (double)xm);
    const float t2 = u * u * u;
    const int r = xe % 3;
    const double factor = r == -2 ? 1.0 / 1.5874010519681994748
                        : r == -1 ? 1.0 / 1.2599210498948731648
                        : r == 0  ? 1.0
                        : r == 1  ? 1.2599210498948731648
                                  : 1.5874010519681994748;
    const float ym = (float)((double)u * ((double)t2 + 2.0 * (double)xm) / (2.0 * (double)t2 + (double)xm) * factor);
    o.f = ym;
    o.u += (uint32_t)(xe / 3) << 23;                       /* ldexpf(ym, xe / 3) */
    return o.f;
}
void ceo_cbrtf_compare(const float *x, size_t n, float *pinned, float *host)
{
    for (size_t i = 0; i < n; i++) {
        pinned[i] = ceo_cbrtf_pinned(x[i]);
        host[i] = cbrtf(x[i]);
    }
}
/* xyb.rs:92-94 */
static float mixed_cbrt(float v) { return v < 0.0f ? -ceo_cbrtf_pinned(-v) : ceo_cbrtf_pinned(v); }
/* xyb.rs:98-100 — powi(3) expands to (v*v)*v */
static float mixed_cube(float v)
{
    if (v < 0.0f) { float n = -v; return -((n * n) * n); }
    return (v * v) * v;
}

/* xyb.rs:104-129 */
static void linear_rgb_to_xyb(float r, float g, float b, float *x, float *y, float *bo)
{
    const float *m = XYB_OPSIN_ABSORBANCE_MATRIX;
    const float *bias = XYB_OPSIN_ABSORBANCE_BIAS;
    float opsin_r = m[0] * r + m[1] * g + m[2] * b + bias[0];
    float opsin_g = m[3] * r + m[4] * g + m[5] * b + bias[1];
    float opsin_b = m[6] * r + m[7] * g + m[8] * b + bias[2];
    float cbrt_r = mixed_cbrt(opsin_r);
    float cbrt_g = mixed_cbrt(opsin_g);
    float cbrt_b = mixed_cbrt(opsin_b);
    const float *nb = XYB_NEG_OPSIN_ABSORBANCE_BIAS_CBRT;
    cbrt_r = cbrt_r + nb[0];
    cbrt_g = cbrt_g + nb[1];
    cbrt_b = cbrt_b + nb[2];
    *x = 0.5f * (cbrt_r - cbrt_g);
    *y = 0.5f * (cbrt_r + cbrt_g);
    *bo = cbrt_b;
}

/* xyb.rs:133-164 */
static void xyb_to_linear_rgb(float x, float y, float b, float *r, float *g, float *bo)
{
    const float *nb = XYB_NEG_OPSIN_ABSORBANCE_BIAS_CBRT;
    float cbrt_r = y + x;
    float cbrt_g = y - x;
    float cbrt_b = b;
    cbrt_r = cbrt_r - nb[0];
    cbrt_g = cbrt_g - nb[1];
    cbrt_b = cbrt_b - nb[2];
    float opsin_r = mixed_cube(cbrt_r);
    float opsin_g = mixed_cube(cbrt_g);
    float opsin_b = mixed_cube(cbrt_b);
    const float *bias = XYB_OPSIN_ABSORBANCE_BIAS;
    opsin_r = opsin_r - bias[0];
    opsin_g = opsin_g - bias[1];
    opsin_b = opsin_b - bias[2];
    const float *inv = INV_OPSIN_MATRIX;
    *r = inv[0] * opsin_r + inv[1] * opsin_g + inv[2] * opsin_b;
    *g = inv[3] * opsin_r + inv[4] * opsin_g + inv[5] * opsin_b;
    *bo = inv[6] * opsin_r + inv[7] * opsin_g + inv[8] * opsin_b;
}

/* xyb.rs:185-190 */
#define X_MIN (-0.016f)
#define X_MAX (0.029f)
#define Y_MIN (0.0f)
#define Y_MAX (0.846f)
#define B_MIN (0.0f)
#define B_MAX (0.846f)

/* xyb.rs:194-199 */
static float quantize_to_u8(float value, float min, float max)
{
    float range = max - min;
    float normalized = (value - min) / range;
    float q = roundf(normalized * 255.0f);
    q = q < 0.0f ? 0.0f : (q > 255.0f ? 255.0f : q);
    float quantized = q / 255.0f;
    return quantized * range + min;
}

/* xyb.rs:225-253.  The reference asserts on the length (:227). */
int ceo_xyb_roundtrip(const uint8_t *rgb, size_t len, size_t width, size_t height, uint8_t *out)
{
    size_t num_pixels = width * height;
    if (len != num_pixels * 3) return CEO_BAD_LENGTH;
    for (size_t i = 0; i < num_pixels; i++) {
        uint8_t r = rgb[i * 3], g = rgb[i * 3 + 1], b = rgb[i * 3 + 2];
        float x, y, bx;
        /* srgb_to_xyb, xyb.rs:167-172 */
        linear_rgb_to_xyb(srgb_u8_to_linear(r), srgb_u8_to_linear(g), srgb_u8_to_linear(b), &x, &y, &bx);
        float xq = quantize_to_u8(x, X_MIN, X_MAX);
        float yq = quantize_to_u8(y, Y_MIN, Y_MAX);
        float bq = quantize_to_u8(bx, B_MIN, B_MAX);
        /* xyb_to_srgb, xyb.rs:175-182 */
        float lr, lg, lb;
        xyb_to_linear_rgb(xq, yq, bq, &lr, &lg, &lb);
        out[i * 3] = linear_to_srgb_u8(lr);
        out[i * 3 + 1] = linear_to_srgb_u8(lg);
        out[i * 3 + 2] = linear_to_srgb_u8(lb);
    }
    return CEO_OK;
}
