/*
 * ORACLE (test infrastructure only — see ce_oracle.h).  PARITY UNPINNED.
 *
 * DSSIM as called at /root/reference/src/metrics/dssim.rs:52-70:
 *   Dssim::new(); create_image(reference); create_image(test); compare(..) -> f64
 * dssim-core 3.4.0 (Cargo.lock:356-365) is NOT in the reference tree and cannot
 * be built here; this file restates its published algorithm (SURVEY.md Appendix
 * A.2): multi-scale SSIM over a normalised L*a*b*-like space, 3x3 blur applied
 * twice, chroma pre-blur, mean-absolute-deviation pooling, 1/ssim - 1.
 * Checked only against the inequalities of dssim.rs:180-250 and
 * helpers.rs:337-383.
 *
 * Input convention is the reference's: linear-light RGBA f32 with a = 1.0
 * (dssim.rs:102-114 feeds it).  Alpha is always 1.0 on this path
 * (session.rs:98-117 strips alpha), so no background blending happens.
 * Compile with -ffp-contract=off.
 */
#include "ce_oracle.h"

#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

#define DSSIM_MAX_SCALES 5
static const double DEFAULT_WEIGHTS[DSSIM_MAX_SCALES] = {0.028, 0.197, 0.322, 0.298, 0.155};

/* the fixed 3x3 kernel, applied twice per "blur" */
static const float KERNEL[9] = {
    0.095332f, 0.118095f, 0.095332f,
    0.118095f, 0.146293f, 0.118095f,
    0.095332f, 0.118095f, 0.095332f,
};

/* one 3x3 pass with edge replication; summation order: corners, edges, centre */
static void blur_pass(const float *src, float *dst, size_t w, size_t h)
{
    for (size_t y = 0; y < h; y++) {
        const float *prev = src + (y > 0 ? y - 1 : 0) * w;
        const float *curr = src + y * w;
        const float *next = src + (y + 1 < h ? y + 1 : y) * w;
        for (size_t x = 0; x < w; x++) {
            size_t c0 = x > 0 ? x - 1 : 0, c1 = x, c2 = x + 1 < w ? x + 1 : w - 1;
            dst[y * w + x] = (prev[c0] + prev[c2] + next[c0] + next[c2]) * KERNEL[0] +
                             (prev[c1] + curr[c0] + curr[c2] + next[c1]) * KERNEL[1] +
                             curr[c1] * KERNEL[4];
        }
    }
}

static void blur2(const float *src, float *dst, float *tmp, size_t w, size_t h)
{
    blur_pass(src, tmp, w, h);
    blur_pass(tmp, dst, w, h);
}

/* cube root by a quadratic seed + two Halley steps, f32 */
static float cbrt_poly(float x)
{
    float y = (-0.5f * x + 1.51f) * x + 0.2f;
    float y3 = y * y * y;
    y = y * (y3 + 2.0f * x) / (2.0f * y3 + x);
    y3 = y * y * y;
    y = y * (y3 + 2.0f * x) / (2.0f * y3 + x);
    return y;
}

#define D65X 0.9505f
#define D65Y 1.0f
#define D65Z 1.089f
#define LAB_EPSILON (216.0f / 24389.0f)
#define LAB_K (24389.0f / (27.0f * 116.0f))

extern int ceo_variant[CEO_V_COUNT]; /* sensitivity switches, all 0 by default (ce_oracle.h) */

/* linear RGB -> normalised (L, a, b), each in ~[0,1] */
static void rgb_to_lab(float r, float g, float b, float *L, float *A, float *B)
{
    float fx = fmaf(b, 0.1805f / D65X, fmaf(g, 0.3576f / D65X, r * (0.4124f / D65X)));
    float fy = fmaf(b, 0.0722f / D65Y, fmaf(g, 0.7152f / D65Y, r * (0.2126f / D65Y)));
    float fz = fmaf(b, 0.9505f / D65Z, fmaf(g, 0.1192f / D65Z, r * (0.0193f / D65Z)));
    if (ceo_variant[CEO_V_DSSIM_LAB_NO_FMA]) {
        fx = (r * (0.4124f / D65X) + g * (0.3576f / D65X)) + b * (0.1805f / D65X);
        fy = (r * (0.2126f / D65Y) + g * (0.7152f / D65Y)) + b * (0.0722f / D65Y);
        fz = (r * (0.0193f / D65Z) + g * (0.1192f / D65Z)) + b * (0.9505f / D65Z);
    }
    float X = fx > LAB_EPSILON ? cbrt_poly(fx) - 16.0f / 116.0f : LAB_K * fx;
    float Y = fy > LAB_EPSILON ? cbrt_poly(fy) - 16.0f / 116.0f : LAB_K * fy;
    float Z = fz > LAB_EPSILON ? cbrt_poly(fz) - 16.0f / 116.0f : LAB_K * fz;
    *L = Y * 1.05f;
    *A = fmaf(500.0f / 220.0f, X - Y, 86.2f / 220.0f);
    *B = fmaf(200.0f / 220.0f, Y - Z, 107.9f / 220.0f);
    if (ceo_variant[CEO_V_DSSIM_LAB_NO_FMA]) {
        *A = (500.0f / 220.0f) * (X - Y) + 86.2f / 220.0f;
        *B = (200.0f / 220.0f) * (Y - Z) + 107.9f / 220.0f;
    }
}

typedef struct {
    size_t w, h;
    float *img[3];   /* L, a(pre-blurred), b(pre-blurred) */
    float *mu[3];    /* blur(img) */
    float *sq[3];    /* blur(img*img) */
} dssim_scale;

typedef struct {
    int n;
    dssim_scale s[DSSIM_MAX_SCALES];
} dssim_image;

static void free_image(dssim_image *im)
{
    for (int i = 0; i < im->n; i++)
        for (int c = 0; c < 3; c++) {
            free(im->s[i].img[c]);
            free(im->s[i].mu[c]);
            free(im->s[i].sq[c]);
        }
}

/* Dssim::create_image: per scale  LAB planes -> chroma pre-blur -> mu, blur(img^2);
 * next scale = 2x2 average of linear RGB with floor(w/2) x floor(h/2), stops
 * when the image to be halved is under 8 px on a side. */
static void create_image(const float *rgb_lin /* 3 planar */, size_t w, size_t h, dssim_image *out)
{
    float *cur = (float *)malloc(sizeof(float) * 3 * w * h);
    memcpy(cur, rgb_lin, sizeof(float) * 3 * w * h);
    out->n = 0;
    for (int scale = 0; scale < DSSIM_MAX_SCALES; scale++) {
        const size_t n = w * h;
        dssim_scale *s = &out->s[out->n++];
        s->w = w;
        s->h = h;
        float *tmp = (float *)malloc(sizeof(float) * n);
        float *tmp2 = (float *)malloc(sizeof(float) * n);
        for (int c = 0; c < 3; c++) {
            s->img[c] = (float *)malloc(sizeof(float) * n);
            s->mu[c] = (float *)malloc(sizeof(float) * n);
            s->sq[c] = (float *)malloc(sizeof(float) * n);
        }
        for (size_t i = 0; i < n; i++)
            rgb_to_lab(cur[i], cur[n + i], cur[2 * n + i], &s->img[0][i], &s->img[1][i], &s->img[2][i]);
        for (int c = 0; c < 3; c++) {
            if (c > 0) { /* chroma pre-blur, in place */
                blur2(s->img[c], tmp2, tmp, w, h);
                memcpy(s->img[c], tmp2, sizeof(float) * n);
            }
            blur2(s->img[c], s->mu[c], tmp, w, h);
            for (size_t i = 0; i < n; i++) tmp2[i] = s->img[c][i] * s->img[c][i];
            blur2(tmp2, s->sq[c], tmp, w, h);
        }
        free(tmp);
        free(tmp2);
        if (scale + 1 >= DSSIM_MAX_SCALES) break;
        if (w < 8 || h < 8) break; /* Downsample::downsample returns None */
        size_t hw = w / 2, hh = h / 2;
        float *nxt = (float *)malloc(sizeof(float) * 3 * hw * hh);
        for (int c = 0; c < 3; c++)
            for (size_t y = 0; y < hh; y++)
                for (size_t x = 0; x < hw; x++) {
                    const float *p = cur + (size_t)c * n;
                    float a = p[(2 * y) * w + 2 * x], b = p[(2 * y) * w + 2 * x + 1];
                    float cc = p[(2 * y + 1) * w + 2 * x], d = p[(2 * y + 1) * w + 2 * x + 1];
                    nxt[(size_t)c * hw * hh + y * hw + x] = (a + b + cc + d) * 0.25f;
                }
        free(cur);
        cur = nxt;
        w = hw;
        h = hh;
    }
    free(cur);
}

/* Dssim::compare + compare_scale */
static double compare(const dssim_image *o, const dssim_image *m, double *scale_scores, int *n_scales)
{
    const float c1 = 0.01f * 0.01f, c2 = 0.03f * 0.03f;
    int ns = o->n < m->n ? o->n : m->n;
    double ssim_sum = 0.0, weight_sum = 0.0;
    for (int k = 0; k < ns; k++) {
        const dssim_scale *a = &o->s[k], *b = &m->s[k];
        const size_t w = a->w, h = a->h, n = w * h;
        float *i12[3];
        float *tmp = (float *)malloc(sizeof(float) * n);
        float *mul = (float *)malloc(sizeof(float) * n);
        for (int c = 0; c < 3; c++) {
            i12[c] = (float *)malloc(sizeof(float) * n);
            for (size_t i = 0; i < n; i++) mul[i] = a->img[c][i] * b->img[c][i];
            blur2(mul, i12[c], tmp, w, h);
        }
        float *map = tmp;
        const float third = 1.0f / 3.0f;
        for (size_t i = 0; i < n; i++) {
            /* per-channel products, then the LAB triple collapses to its mean */
            float mu1mu1[3], mu1mu2[3], mu2mu2[3], s1[3], s2[3], s12[3];
            for (int c = 0; c < 3; c++) {
                float u1 = a->mu[c][i], u2 = b->mu[c][i];
                mu1mu1[c] = u1 * u1;
                mu1mu2[c] = u1 * u2;
                mu2mu2[c] = u2 * u2;
                s1[c] = a->sq[c][i] - mu1mu1[c];
                s2[c] = b->sq[c][i] - mu2mu2[c];
                s12[c] = i12[c][i] - mu1mu2[c];
            }
#define AVG3(v) (((v)[0] + (v)[1] + (v)[2]) * third)
            float mu1_sq = AVG3(mu1mu1), mu2_sq = AVG3(mu2mu2), mu1_mu2 = AVG3(mu1mu2);
            float sigma1_sq = AVG3(s1), sigma2_sq = AVG3(s2), sigma12 = AVG3(s12);
#undef AVG3
            map[i] = (2.0f * mu1_mu2 + c1) * (2.0f * sigma12 + c2) /
                     ((mu1_sq + mu2_sq + c1) * (sigma1_sq + sigma2_sq + c2));
        }
        double sum = 0.0;
        for (size_t i = 0; i < n; i++) sum += (double)map[i];
        double len = (double)n;
        double avg = sum / len;
        if (!(avg > 0.0)) avg = 0.0;
        avg = pow(avg, pow(0.5, (double)k));
        double dev = 0.0;
        for (size_t i = 0; i < n; i++) dev += fabs(avg - (double)map[i]);
        double score = 1.0 - dev / len;
        if (scale_scores) scale_scores[k] = score;
        ssim_sum += score * DEFAULT_WEIGHTS[k];
        weight_sum += DEFAULT_WEIGHTS[k];
        for (int c = 0; c < 3; c++) free(i12[c]);
        free(tmp);
        free(mul);
    }
    if (n_scales) *n_scales = ns;
    double ssim = ssim_sum / weight_sum;
    if (ceo_variant[CEO_V_DSSIM_F32_FINAL]) { /* the crate's Val as f32: weighting and to_dssim in f32, widened by f64::from */
        float s = (float)ssim_sum / (float)weight_sum;
        if (!(s > FLT_EPSILON)) s = FLT_EPSILON;
        return (double)(1.0f / s - 1.0f);
    }
    if (!(ssim > DBL_EPSILON)) ssim = DBL_EPSILON;
    return 1.0 / ssim - 1.0; /* to_dssim */
}

static int dssim_planar(const float *p1, const float *p2, size_t w, size_t h, double *scale_scores,
                        int *n_scales, double *out)
{
    dssim_image a, b;
    create_image(p1, w, h, &a);
    create_image(p2, w, h, &b);
    *out = compare(&a, &b, scale_scores, n_scales);
    free_image(&a);
    free_image(&b);
    return CEO_OK;
}

/* src/metrics/dssim.rs:40-71 — RGBA<f32> linear in, DimensionMismatch on w/h */
int ceo_dssim_rgbaf(const float *ref_rgba, size_t rw, size_t rh, const float *test_rgba, size_t tw,
                    size_t th, double *out)
{
    if (rw != tw || rh != th) return CEO_DIM_MISMATCH; /* :45-50 */
    if (rw == 0 || rh == 0) return CEO_BACKEND;        /* create_image -> None, :54-58 */
    size_t n = rw * rh;
    float *p1 = (float *)malloc(sizeof(float) * 3 * n), *p2 = (float *)malloc(sizeof(float) * 3 * n);
    for (size_t i = 0; i < n; i++)
        for (int c = 0; c < 3; c++) {
            p1[(size_t)c * n + i] = ref_rgba[4 * i + c];
            p2[(size_t)c * n + i] = test_rgba[4 * i + c];
        }
    int rc = dssim_planar(p1, p2, rw, rh, NULL, NULL, out);
    free(p1);
    free(p2);
    return rc;
}

int ceo_dssim_detail(const uint8_t *ref, const uint8_t *test, size_t width, size_t height,
                     double *scale_scores, int *n_scales, double *out)
{
    size_t n = width * height;
    if (n == 0) return CEO_BACKEND;
    float *p1 = (float *)malloc(sizeof(float) * 3 * n), *p2 = (float *)malloc(sizeof(float) * 3 * n);
    for (size_t i = 0; i < n; i++)
        for (int c = 0; c < 3; c++) { /* session.rs:468-470 via dssim.rs:102-114 */
            p1[(size_t)c * n + i] = ceo_srgb_u8_to_linear(ref[3 * i + c]);
            p2[(size_t)c * n + i] = ceo_srgb_u8_to_linear(test[3 * i + c]);
        }
    int rc = dssim_planar(p1, p2, width, height, scale_scores, n_scales, out);
    free(p1);
    free(p2);
    return rc;
}

/* the shape session.rs:467-476 composes: RGB8 in, both sides through
 * rgb8_to_dssim_image, then calculate_dssim */
int ceo_dssim_rgb8(const uint8_t *ref, size_t ref_len, const uint8_t *test, size_t test_len,
                   size_t width, size_t height, double *out)
{
    if (ref_len != test_len) return CEO_DIM_MISMATCH;
    if (ref_len != width * height * 3) return CEO_BAD_LENGTH;
    return ceo_dssim_detail(ref, test, width, height, NULL, NULL, out);
}
