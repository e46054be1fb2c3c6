/*
 * ORACLE (test infrastructure only — see ce_oracle.h).  PARITY UNPINNED.
 *
 * SSIMULACRA2 as called at /root/reference/src/metrics/ssimulacra2.rs:96
 * (`fast_ssim2::compute_ssimulacra2`) and crates/codec-iter/src/eval.rs:142,87
 * (`Ssimulacra2Reference::{new,compare}`).  fast-ssim2 0.8.0 (Cargo.lock:410-421)
 * is NOT in the reference tree and cannot be built here, so this file restates
 * the published algorithm it implements — SSIMULACRA 2.1, libjxl
 * tools/ssimulacra2.cc -> `ssimulacra2` crate -> fast-ssim2 ("identical
 * results", ssimulacra2.rs:16-18) — following SURVEY.md Appendix A.1 step by
 * step.  It is checked against the only results the reference pins at this
 * boundary: the inequalities of ssimulacra2.rs:153-182 and helpers.rs:337-375.
 *
 * Numeric types follow the lineage: f32 planes, f64 pooled sums.
 * Compile with -ffp-contract=off; every fused multiply-add below is an
 * explicit fmaf()/fma() where the lineage writes mul_add.
 */
#include "ce_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

extern int ceo_variant[CEO_V_COUNT]; /* sensitivity switches, all 0 by default (ce_oracle.h) */
#define NUM_SCALES 6
#define BLUR_RADIUS 5 /* round(3.2795*1.5 + 0.2546) */

/* ---- A.1 step 1: sRGB u8 -> linear f32 -------------------------------------
 * Standard piecewise 2.4 curve evaluated per u8 code in f64 and rounded once to
 * f32 (a 256-entry table; what a u8-input implementation holds). */
void ceo_ssim2_srgb_lut(float lut[256])
{
    for (int i = 0; i < 256; i++) {
        double v = (double)i / 255.0;
        double l = v <= 0.04045 ? v / 12.92 : pow((v + 0.055) / 1.055, 2.4);
        lut[i] = (float)l;
        if (ceo_variant[CEO_V_SSIM2_SRGB_F32_POWF]) {
            float vf = (float)i / 255.0f;
            lut[i] = vf <= 0.04045f ? vf / 12.92f : powf((vf + 0.055f) / 1.055f, 2.4f);
        }
    }
}

void ceo_ssim2_linear_planar(const uint8_t *rgb, size_t npix, float *planes)
{
    float lut[256];
    ceo_ssim2_srgb_lut(lut);
    for (size_t i = 0; i < npix; i++)
        for (int c = 0; c < 3; c++) planes[(size_t)c * npix + i] = lut[rgb[3 * i + c]];
}

/* ---- A.1 step 2: 2x2 box average of LINEAR RGB, edge-clamped, ceil sizes ---- */
void ceo_ssim2_downscale(const float *in, size_t w, size_t h, float *out)
{
    size_t ow = (w + 1) / 2, oh = (h + 1) / 2;
    for (int c = 0; c < 3; c++) {
        const float *ip = in + (size_t)c * w * h;
        float *op = out + (size_t)c * ow * oh;
        for (size_t oy = 0; oy < oh; oy++)
            for (size_t ox = 0; ox < ow; ox++) {
                float sum = 0.0f;
                for (size_t iy = 0; iy < 2; iy++)
                    for (size_t ix = 0; ix < 2; ix++) {
                        size_t x = ox * 2 + ix, y = oy * 2 + iy;
                        if (x > w - 1) x = w - 1;
                        if (y > h - 1) y = h - 1;
                        sum += ip[y * w + x];
                    }
                op[oy * ow + ox] = sum * 0.25f;
            }
    }
}

/* ---- A.1 step 3: linear RGB -> XYB, then make_positive_xyb -------------------
 * cbrt: the FreeBSD msun cbrtf (bit-trick seed + two f64 Newton steps, rounded
 * once to f32) — deterministic IEEE arithmetic only, so the identical sequence
 * on the device gives identical bits. */
static float ssim2_cbrtf(float x)
{
    union { float f; uint32_t u; } t, fx;
    fx.f = x;
    uint32_t hx = fx.u & 0x7fffffffu;
    if (hx == 0) return x;
    if (ceo_variant[CEO_V_SSIM2_HOST_CBRTF]) return cbrtf(x);
    if (hx < 0x00800000u) { /* subnormal */
        t.u = 0x4b800000u;
        t.f *= x;
        t.u = (t.u & 0x7fffffffu) / 3 + 642849266u;
    } else {
        t.u = hx / 3 + 709958130u;
    }
    double T = t.f, r;
    r = T * T * T;
    T = T * ((double)x + x + r) / (x + r + r);
    r = T * T * T;
    T = T * ((double)x + x + r) / (x + r + r);
    return (float)T;
}

#define K_M00 0.30f
#define K_M02 0.078f
#define K_M01 (1.0f - K_M02 - K_M00)
#define K_M10 0.23f
#define K_M12 0.078f
#define K_M11 (1.0f - K_M12 - K_M10)
#define K_M20 0.24342268924547819f
#define K_M21 0.20476744424496821f
#define K_M22 (1.0f - K_M20 - K_M21)
#define K_B0 0.0037930732552754493f

void ceo_ssim2_xyb_positive(const float *lin, size_t npix, float *xyb)
{
    const float cbrt_bias = ssim2_cbrtf(K_B0);
    const float m01 = K_M01, m11 = K_M11, m22 = K_M22;
    for (size_t i = 0; i < npix; i++) {
        float r = lin[i], g = lin[npix + i], b = lin[2 * npix + i];
        float m0 = fmaf(K_M00, r, fmaf(m01, g, fmaf(K_M02, b, K_B0)));
        float m1 = fmaf(K_M10, r, fmaf(m11, g, fmaf(K_M12, b, K_B0)));
        float m2 = fmaf(K_M20, r, fmaf(K_M21, g, fmaf(m22, b, K_B0)));
        m0 = ssim2_cbrtf(m0 < 0.0f ? 0.0f : m0) - cbrt_bias;
        m1 = ssim2_cbrtf(m1 < 0.0f ? 0.0f : m1) - cbrt_bias;
        m2 = ssim2_cbrtf(m2 < 0.0f ? 0.0f : m2) - cbrt_bias;
        float X = 0.5f * (m0 - m1);
        float Y = 0.5f * (m0 + m1);
        float B = m2;
        /* make_positive_xyb */
        B = (B - Y) + 0.55f;
        X = fmaf(X, 14.0f, 0.42f);
        Y = Y + 0.01f;
        xyb[i] = X;
        xyb[npix + i] = Y;
        xyb[2 * npix + i] = B;
    }
}

/* ---- A.1 step 4 + §9: the sigma = 1.5 recursive Gaussian -------------------- */
typedef struct {
    double beta[3], omega[3], n2[3], d1[3];
    double taps[BLUR_RADIUS]; /* h[0..4]; h[5] == 0 */
} rg_coeffs;

static void inv3x3(double *m)
{
    double a = m[0], b = m[1], c = m[2], d = m[3], e = m[4], f = m[5], g = m[6], h = m[7], i = m[8];
    double det = a * (e * i - f * h) - b * (d * i - f * g) + c * (d * h - e * g);
    double id = 1.0 / det;
    m[0] = (e * i - f * h) * id; m[1] = (c * h - b * i) * id; m[2] = (b * f - c * e) * id;
    m[3] = (f * g - d * i) * id; m[4] = (a * i - c * g) * id; m[5] = (c * d - a * f) * id;
    m[6] = (d * h - e * g) * id; m[7] = (b * g - a * h) * id; m[8] = (a * e - b * d) * id;
}

/* libjxl CreateRecursiveGaussian (Charalampidis 2016, truncated cosine, k = 1,3,5) */
static void rg_create(double sigma, rg_coeffs *rg)
{
    const double radius = round(3.2795 * sigma + 0.2546);
    const double pi_div_2r = M_PI / (2.0 * radius);
    const double omega[3] = {pi_div_2r, 3.0 * pi_div_2r, 5.0 * pi_div_2r};
    const double p1 = +1.0 / tan(0.5 * omega[0]);
    const double p3 = -1.0 / tan(0.5 * omega[1]);
    const double p5 = +1.0 / tan(0.5 * omega[2]);
    const double r1 = +p1 * p1 / sin(omega[0]);
    const double r3 = -p3 * p3 / sin(omega[1]);
    const double r5 = +p5 * p5 / sin(omega[2]);
    const double neg_half_sigma2 = -0.5 * sigma * sigma;
    const double recip_radius = 1.0 / radius;
    double rho[3];
    for (int i = 0; i < 3; i++) rho[i] = exp(neg_half_sigma2 * omega[i] * omega[i]) * recip_radius;
    const double D13 = p1 * r3 - r1 * p3;
    const double D35 = p3 * r5 - r3 * p5;
    const double D51 = p5 * r1 - r5 * p1;
    const double recip_d13 = 1.0 / D13;
    const double zeta15 = D35 * recip_d13;
    const double zeta35 = D51 * recip_d13;
    double A[9] = {p1, p3, p5, r1, r3, r5, zeta15, zeta35, 1.0};
    inv3x3(A);
    const double gamma[3] = {1.0, radius * radius - sigma * sigma,
                             zeta15 * rho[0] + zeta35 * rho[1] + rho[2]};
    for (int i = 0; i < 3; i++) {
        rg->beta[i] = A[3 * i] * gamma[0] + A[3 * i + 1] * gamma[1] + A[3 * i + 2] * gamma[2];
        rg->omega[i] = omega[i];
        rg->n2[i] = -rg->beta[i] * cos(omega[i] * (radius + 1.0));
        rg->d1[i] = -2.0 * cos(omega[i]);
    }
    /* exact impulse response: h[m] = sum_k beta_k cos(omega_k m), |m| <= N; h[N] = 0 */
    for (int m = 0; m < BLUR_RADIUS; m++) {
        double s = 0.0;
        for (int k = 0; k < 3; k++) s += rg->beta[k] * cos(omega[k] * m);
        rg->taps[m] = s;
    }
}

void ceo_ssim2_blur_taps(float taps[5], double taps_f64[5])
{
    rg_coeffs rg;
    rg_create(1.5, &rg);
    for (int m = 0; m < BLUR_RADIUS; m++) {
        taps[m] = (float)rg.taps[m];
        if (taps_f64) taps_f64[m] = rg.taps[m];
    }
}

/* one 1-D pass of the f32 recursive form over `n` samples with stride `stride`
 * (zero outside the line; the three second-order sections k = 1,3,5) */
static void rg_line_iir(const rg_coeffs *rg, const float *in, float *out, ptrdiff_t n, ptrdiff_t stride)
{
    const ptrdiff_t N = BLUR_RADIUS;
    const float mul_in[3] = {(float)rg->n2[0], (float)rg->n2[1], (float)rg->n2[2]};
    const float mul_prev[3] = {(float)-rg->d1[0], (float)-rg->d1[1], (float)-rg->d1[2]};
    float prev[3] = {0, 0, 0}, prev2[3] = {0, 0, 0};
    for (ptrdiff_t i = -N + 1; i < n; i++) {
        ptrdiff_t left = i - N - 1, right = i + N - 1;
        float lv = left >= 0 ? in[left * stride] : 0.0f;
        float rv = right < n ? in[right * stride] : 0.0f;
        float sum = lv + rv;
        float o[3];
        for (int k = 0; k < 3; k++) {
            float v = sum * mul_in[k];
            if (ceo_variant[CEO_V_SSIM2_IIR_NO_FMA]) {
                v = v - prev2[k];
                prev2[k] = prev[k];
                float t = mul_prev[k] * prev[k];
                v = t + v;
                prev[k] = v;
                o[k] = v;
                continue;
            }
            v = fmaf(-1.0f, prev2[k], v);
            prev2[k] = prev[k];
            v = fmaf(mul_prev[k], prev[k], v);
            prev[k] = v;
            o[k] = v;
        }
        if (i >= 0) out[i * stride] = o[0] + o[1] + o[2];
    }
}

/* one 1-D pass of the 9-tap FIR form, zero padding, f32 taps and f32 accumulate
 * in the fixed order  c, (+1,-1), (+2,-2), (+3,-3), (+4,-4)  — the order the
 * device kernel uses. */
static void rg_line_fir(const float taps[5], const float *in, float *out, ptrdiff_t n, ptrdiff_t stride)
{
    for (ptrdiff_t i = 0; i < n; i++) {
        float acc = taps[0] * in[i * stride];
        for (ptrdiff_t k = 1; k < BLUR_RADIUS; k++) {
            float a = i - k >= 0 ? in[(i - k) * stride] : 0.0f;
            float b = i + k < n ? in[(i + k) * stride] : 0.0f;
            acc = fmaf(taps[k], a + b, acc);
        }
        out[i * stride] = acc;
    }
}

/* horizontal pass then vertical pass (A.1 step 4) */
void ceo_ssim2_blur_plane(const float *in, size_t w, size_t h, int blur_mode, float *out)
{
    rg_coeffs rg;
    rg_create(1.5, &rg);
    float taps[5];
    for (int m = 0; m < 5; m++) taps[m] = (float)rg.taps[m];
    float *tmp = (float *)malloc(sizeof(float) * w * h);
    for (size_t y = 0; y < h; y++) {
        if (blur_mode == 1)
            rg_line_iir(&rg, in + y * w, tmp + y * w, (ptrdiff_t)w, 1);
        else
            rg_line_fir(taps, in + y * w, tmp + y * w, (ptrdiff_t)w, 1);
    }
    for (size_t x = 0; x < w; x++) {
        if (blur_mode == 1)
            rg_line_iir(&rg, tmp + x, out + x, (ptrdiff_t)h, (ptrdiff_t)w);
        else
            rg_line_fir(taps, tmp + x, out + x, (ptrdiff_t)h, (ptrdiff_t)w);
    }
    free(tmp);
}

/* ---- A.1 step 5: SSIM map pooled to (mean, 4-norm) per channel ------------- */
static void ssim_map(size_t w, size_t h, const float *m1, const float *m2, const float *s11,
                     const float *s22, const float *s12, double *avg /* stride 6: [0],[1] */)
{
    const float C2 = 0.0009f;
    const size_t n = w * h;
    const double one_per_pixels = 1.0 / (double)n;
    for (int c = 0; c < 3; c++) {
        double sum1[2] = {0.0, 0.0};
        const size_t o = (size_t)c * n;
        for (size_t i = 0; i < n; i++) {
            float mu1 = m1[o + i], mu2 = m2[o + i];
            float mu11 = mu1 * mu1, mu22 = mu2 * mu2, mu12 = mu1 * mu2;
            float mu_diff = mu1 - mu2;
            float num_m = fmaf(mu_diff, -mu_diff, 1.0f);
            float num_s = fmaf(2.0f, s12[o + i] - mu12, C2);
            float denom_s = (s11[o + i] - mu11) + (s22[o + i] - mu22) + C2;
            double d = 1.0 - (double)((num_m * num_s) / denom_s);
            if (ceo_variant[CEO_V_SSIM2_F32_POOL]) d = (double)(1.0f - (num_m * num_s) / denom_s);
            if (!(d > 0.0)) d = 0.0;
            sum1[0] += d;
            double d2 = d * d;
            sum1[1] += d2 * d2;
        }
        avg[c * 6 + 0] = one_per_pixels * sum1[0];
        avg[c * 6 + 1] = sqrt(sqrt(one_per_pixels * sum1[1]));
    }
}

/* ---- A.1 step 6: edge-difference maps pooled likewise ---------------------- */
static void edge_diff_map(size_t w, size_t h, const float *img1, const float *mu1, const float *img2,
                          const float *mu2, double *avg /* stride 6: [2..5] */)
{
    const size_t n = w * h;
    const double one_per_pixels = 1.0 / (double)n;
    for (int c = 0; c < 3; c++) {
        double sum1[4] = {0, 0, 0, 0};
        const size_t o = (size_t)c * n;
        for (size_t i = 0; i < n; i++) {
            double d1 = (1.0 + (double)fabsf(img2[o + i] - mu2[o + i])) /
                            (1.0 + (double)fabsf(img1[o + i] - mu1[o + i])) - 1.0;
            if (ceo_variant[CEO_V_SSIM2_F32_POOL])
                d1 = (double)((1.0f + fabsf(img2[o + i] - mu2[o + i])) / (1.0f + fabsf(img1[o + i] - mu1[o + i])) - 1.0f);
            double artifact = d1 > 0.0 ? d1 : 0.0;
            double detail_lost = -d1 > 0.0 ? -d1 : 0.0;
            sum1[0] += artifact;
            double a2 = artifact * artifact;
            sum1[1] += a2 * a2;
            sum1[2] += detail_lost;
            double l2 = detail_lost * detail_lost;
            sum1[3] += l2 * l2;
        }
        avg[c * 6 + 2] = one_per_pixels * sum1[0];
        avg[c * 6 + 3] = sqrt(sqrt(one_per_pixels * sum1[1]));
        avg[c * 6 + 4] = one_per_pixels * sum1[2];
        avg[c * 6 + 5] = sqrt(sqrt(one_per_pixels * sum1[3]));
    }
}

/* ---- A.1 steps 7-8: Msssim::score ------------------------------------------ */
static const double WEIGHT[108] = {
    0.0, 0.0007376606707406586, 0.0, 0.0, 0.0007793481682867309, 0.0, 0.0, 0.0004371155730107379, 0.0, 1.1041726426657346, 0.00066284834129271, 0.00015231632783718752,
    0.0, 0.0016406437456599754, 0.0, 1.8422455520539298, 11.441172603757666, 0.0, 0.0007989109436015163, 0.000176816438078653, 0.0, 1.8787594979546387, 10.94906990605142, 0.0,
    0.0007289346991508072, 0.9677937080626833, 0.0, 0.00014003424285435884, 0.9981766977854967, 0.00031949755934435053, 0.0004550992113792063, 0.0, 0.0, 0.0013648766163243398, 0.0, 0.0,
    0.0, 0.0, 0.0, 7.466890328078848, 0.0, 17.445833984131262, 0.0006235601634041466, 0.0, 0.0, 6.683678146179332, 0.00037724407979611296, 1.027889937768264,
    225.20515300849274, 0.0, 0.0, 19.213238186143016, 0.0011401524586618361, 0.001237755635509985, 176.39317598450694, 0.0, 0.0, 24.43300999870476, 0.28520802612117757, 0.0004485436923833408,
    0.0, 0.0, 0.0, 34.77906344483772, 44.835625328877896, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0,
    0.0, 0.0008680556573291698, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0005313191874358747, 0.0, 0.00016533814161379112, 0.0, 0.0,
    0.0, 0.0, 0.0, 0.0004179171803251336, 0.0017290828234722833, 0.0, 0.0020827005846636437, 0.0, 0.0, 8.826982764996862, 23.19243343998926, 0.0,
    95.1080498811086, 0.9863978034400682, 0.9834382792465353, 0.0012286405048278493, 171.2667255897307, 0.9807858872435379, 0.0, 0.0, 0.0, 0.0005130064588990679, 0.0, 0.00010854057858411537,
};

/* avg layout: [scale][c][6] = ssim{l1,l4}, artifact{l1,l4}, detail{l1,l4}.
 * The weight cursor advances only over the scales that exist (the loop is
 * `for c { for scale in &self.scales { for n in 0..2 {..} } }`), so an image
 * with fewer than six scales uses a contiguous prefix pattern of the table,
 * exactly as the lineage does. */
double ceo_ssimulacra2_score(const double *avg, int n_scales)
{
    double ssim = 0.0;
    size_t i = 0;
    for (int c = 0; c < 3; c++)
        for (int s = 0; s < n_scales; s++) {
            const double *a = avg + ((size_t)s * 3 + c) * 6;
            for (int n = 0; n < 2; n++) {
                ssim = fma(WEIGHT[i++], fabs(a[0 + n]), ssim);
                ssim = fma(WEIGHT[i++], fabs(a[2 + n]), ssim);
                ssim = fma(WEIGHT[i++], fabs(a[4 + n]), ssim);
            }
        }
    ssim *= 0.9562382616834844;
    ssim = fma(6.248496625763138e-5 * ssim * ssim, ssim,
               fma(2.326765642916932, ssim, -0.020884521182843837 * ssim * ssim));
    if (ssim > 0.0)
        ssim = fma(pow(ssim, 0.6276336467831387), -10.0, 100.0);
    else
        ssim = 100.0;
    return ssim;
}

int ceo_ssimulacra2_detail(const uint8_t *ref, const uint8_t *test, size_t width, size_t height,
                           int blur_mode, double *avg, int *n_scales_out, double *score)
{
    if (width < 8 || height < 8) return CEO_TOO_SMALL;
    size_t w = width, h = height, n = w * h;
    float *lin1 = (float *)malloc(sizeof(float) * 3 * n);
    float *lin2 = (float *)malloc(sizeof(float) * 3 * n);
    float *tmp = (float *)malloc(sizeof(float) * 3 * n);
    float *x1 = (float *)malloc(sizeof(float) * 3 * n);
    float *x2 = (float *)malloc(sizeof(float) * 3 * n);
    float *mul = (float *)malloc(sizeof(float) * 3 * n);
    float *s11 = (float *)malloc(sizeof(float) * 3 * n);
    float *s22 = (float *)malloc(sizeof(float) * 3 * n);
    float *s12 = (float *)malloc(sizeof(float) * 3 * n);
    float *mu1 = (float *)malloc(sizeof(float) * 3 * n);
    float *mu2 = (float *)malloc(sizeof(float) * 3 * n);
    ceo_ssim2_linear_planar(ref, n, lin1);
    ceo_ssim2_linear_planar(test, n, lin2);
    int ns = 0;
    for (int scale = 0; scale < NUM_SCALES; scale++) {
        if (w < 8 || h < 8) break;
        if (scale > 0) {
            ceo_ssim2_downscale(lin1, w, h, tmp);
            size_t ow = (w + 1) / 2, oh = (h + 1) / 2;
            memcpy(lin1, tmp, sizeof(float) * 3 * ow * oh);
            ceo_ssim2_downscale(lin2, w, h, tmp);
            memcpy(lin2, tmp, sizeof(float) * 3 * ow * oh);
            w = ow;
            h = oh;
            n = w * h;
        }
        ceo_ssim2_xyb_positive(lin1, n, x1);
        ceo_ssim2_xyb_positive(lin2, n, x2);
        for (int c = 0; c < 3; c++) {
            const size_t o = (size_t)c * n;
            for (size_t i = 0; i < n; i++) mul[o + i] = x1[o + i] * x1[o + i];
            ceo_ssim2_blur_plane(mul + o, w, h, blur_mode, s11 + o);
            for (size_t i = 0; i < n; i++) mul[o + i] = x2[o + i] * x2[o + i];
            ceo_ssim2_blur_plane(mul + o, w, h, blur_mode, s22 + o);
            for (size_t i = 0; i < n; i++) mul[o + i] = x1[o + i] * x2[o + i];
            ceo_ssim2_blur_plane(mul + o, w, h, blur_mode, s12 + o);
            ceo_ssim2_blur_plane(x1 + o, w, h, blur_mode, mu1 + o);
            ceo_ssim2_blur_plane(x2 + o, w, h, blur_mode, mu2 + o);
        }
        double *a = avg + (size_t)scale * 18;
        ssim_map(w, h, mu1, mu2, s11, s22, s12, a);
        edge_diff_map(w, h, x1, mu1, x2, mu2, a);
        ns++;
    }
    free(lin1); free(lin2); free(tmp); free(x1); free(x2); free(mul);
    free(s11); free(s22); free(s12); free(mu1); free(mu2);
    if (n_scales_out) *n_scales_out = ns;
    *score = ceo_ssimulacra2_score(avg, ns);
    return CEO_OK;
}

/* src/metrics/ssimulacra2.rs:59-100 — validation order and error kinds */
int ceo_ssimulacra2(const uint8_t *ref, size_t ref_len, const uint8_t *test, size_t test_len,
                    size_t width, size_t height, int blur_mode, double *out)
{
    if (ref_len != test_len) return CEO_DIM_MISMATCH;          /* :65-70 */
    if (ref_len != width * height * 3) return CEO_BAD_LENGTH;  /* :72-82 */
    double avg[NUM_SCALES * 18];
    int ns;
    return ceo_ssimulacra2_detail(ref, test, width, height, blur_mode, avg, &ns, out); /* :96 */
}
