/*
 * ORACLE (test infrastructure only — see ce_oracle.h).  PARITY UNPINNED.
 *
 * Butteraugli as called at /root/reference/src/metrics/butteraugli.rs:72-80,127-135
 * (`butteraugli::butteraugli(img1, img2, &params)` -> `result.score`) and by the
 * codec-compare bins (`compute_butteraugli(..).score`, full_comparison.rs:157-161).
 * The butteraugli crate 0.9.0 (Cargo.lock:132-143) is NOT in the reference tree and
 * cannot be built here; this file restates the published algorithm it ports — libjxl
 * lib/jxl/butteraugli/butteraugli.cc — following SURVEY.md Appendix A.3:
 *
 *   sRGB u8 -> linear [0,1] (x intensity_target inside OpsinDynamicsImage)
 *   OpsinDynamicsImage   blur sigma 1.2 -> sensitivity from the blurred image -> XYB
 *   SeparateFrequencies  LF (sigma 7.156) / MF (3.225) / HF (1.564) / UHF, range shaping
 *   Malta line filters on UHF, HF, MF; asymmetric L2 on HF, L2 on MF, LF
 *   Mask                 from HF+UHF of both images (blur 2.7, fuzzy erosion)
 *   CombineChannelsToDiffmap; the same on 2x-subsampled images, added supersampled
 *   score = max(diffmap); p-norm = mean of the 3-, 6-, 12-norms.
 *
 * Checked only against the inequalities of butteraugli.rs:168-207 and helpers.rs:347.
 * f32 planes throughout, like the lineage.  Compile with -ffp-contract=off.
 */
#include "ce_oracle.h"

#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef struct {
    size_t w, h;
    float *p;
} img;

static img img_new(size_t w, size_t h)
{
    img r = {w, h, (float *)calloc((w * h) != 0 ? w * h : 1, sizeof(float))};
    return r;
}
static void img_free(img *a)
{
    free(a->p);
    a->p = NULL;
}

/* ---------------------------------------------------------------- blur ---- */

/* ComputeKernel: half-width max(1, floor(2.25 sigma)), un-normalised exp */
static int compute_kernel(float sigma, float *kernel /* >= 2*diff+1 */)
{
    const float m = 2.25f;
    const double scaler = -1.0 / (2.0 * (double)sigma * (double)sigma);
    int diff = (int)(m * fabsf(sigma));
    if (diff < 1) diff = 1;
    for (int i = -diff; i <= diff; i++) kernel[i + diff] = (float)exp(scaler * i * i);
    return 2 * diff + 1;
}

extern int ceo_variant[CEO_V_COUNT]; /* sensitivity switches, all 0 by default (ce_oracle.h) */

static size_t mirror(ptrdiff_t x, ptrdiff_t n)
{
    while (x < 0 || x >= n) {
        if (x < 0)
            x = -x - 1;
        else
            x = 2 * n - 1 - x;
    }
    return (size_t)x;
}

/* one 1-D pass along x with re-normalisation at the borders (ConvolutionWithTranspose
 * without the transpose: the caller passes strides) */
static void conv_line_renorm(const float *in, ptrdiff_t n, ptrdiff_t stride, const float *kernel, int len,
                             float *out, ptrdiff_t ostride)
{
    const ptrdiff_t off = len / 2;
    float wsum = 0.0f;
    for (int j = 0; j < len; j++) wsum += kernel[j];
    const float scale_no_border = 1.0f / wsum;
    const int fused = ceo_variant[CEO_V_BA_BLUR_FMA]; /* same taps, same order, one rounding per tap instead of two */
    for (ptrdiff_t x = 0; x < n; x++) {
        ptrdiff_t lo = x - off < 0 ? 0 : x - off;
        ptrdiff_t hi = x + off > n - 1 ? n - 1 : x + off;
        float sum = 0.0f;
        if (lo == x - off && hi == x + off) {
            if (fused)
                for (ptrdiff_t j = lo; j <= hi; j++) sum = fmaf(in[j * stride], kernel[j - x + off], sum);
            else
                for (ptrdiff_t j = lo; j <= hi; j++) sum += in[j * stride] * kernel[j - x + off];
            out[x * ostride] = sum * scale_no_border;
        } else {
            float weight = 0.0f;
            for (ptrdiff_t j = lo; j <= hi; j++) weight += kernel[j - x + off];
            const float scale = 1.0f / weight;
            if (fused)
                for (ptrdiff_t j = lo; j <= hi; j++) sum = fmaf(in[j * stride], kernel[j - x + off], sum);
            else
                for (ptrdiff_t j = lo; j <= hi; j++) sum += in[j * stride] * kernel[j - x + off];
            out[x * ostride] = sum * scale;
        }
    }
}

/* Blur(): 5-tap kernels take the mirrored, normalised separable path; longer kernels the
 * border-renormalised one */
static void blur(const img *in, float sigma, img *out)
{
    float kernel[64];
    const int len = compute_kernel(sigma, kernel);
    const size_t w = in->w, h = in->h;
    float *tmp = (float *)malloc(sizeof(float) * w * h);
    if (len == 5) {
        float sw = 0.0f;
        for (int j = 0; j < 5; j++) sw += kernel[j];
        const float scale = 1.0f / sw;
        const float w0 = kernel[2] * scale, w1 = kernel[3] * scale, w2 = kernel[4] * scale;
        for (size_t y = 0; y < h; y++)
            for (size_t x = 0; x < w; x++) {
                const float *r = in->p + y * w;
                ptrdiff_t X = (ptrdiff_t)x, W = (ptrdiff_t)w;
                tmp[y * w + x] = r[x] * w0 + (r[mirror(X - 1, W)] + r[mirror(X + 1, W)]) * w1 +
                                 (r[mirror(X - 2, W)] + r[mirror(X + 2, W)]) * w2;
            }
        for (size_t y = 0; y < h; y++)
            for (size_t x = 0; x < w; x++) {
                ptrdiff_t Y = (ptrdiff_t)y, H = (ptrdiff_t)h;
                out->p[y * w + x] = tmp[y * w + x] * w0 +
                                    (tmp[mirror(Y - 1, H) * w + x] + tmp[mirror(Y + 1, H) * w + x]) * w1 +
                                    (tmp[mirror(Y - 2, H) * w + x] + tmp[mirror(Y + 2, H) * w + x]) * w2;
            }
    } else {
        for (size_t y = 0; y < h; y++)
            conv_line_renorm(in->p + y * w, (ptrdiff_t)w, 1, kernel, len, tmp + y * w, 1);
        for (size_t x = 0; x < w; x++)
            conv_line_renorm(tmp + x, (ptrdiff_t)h, (ptrdiff_t)w, kernel, len, out->p + x, (ptrdiff_t)w);
    }
    free(tmp);
}

/* ------------------------------------------------------ opsin dynamics ---- */

static void opsin_absorbance(float in0, float in1, float in2, float *o0, float *o1, float *o2)
{
    const float mixi0 = 0.29956550340058319f, mixi1 = 0.63373087833825936f, mixi2 = 0.077705617820981968f,
                mixi3 = 1.7557483643287353f;
    const float mixi4 = 0.22158691104574774f, mixi5 = 0.69391388044116142f, mixi6 = 0.0987313588422f,
                mixi7 = 1.7557483643287353f;
    const float mixi8 = 0.02f, mixi9 = 0.02f, mixi10 = 0.20480129041026129f, mixi11 = 12.226454707163354f;
    *o0 = fmaf(mixi0, in0, fmaf(mixi1, in1, fmaf(mixi2, in2, mixi3)));
    *o1 = fmaf(mixi4, in0, fmaf(mixi5, in1, fmaf(mixi6, in2, mixi7)));
    *o2 = fmaf(mixi8, in0, fmaf(mixi9, in1, fmaf(mixi10, in2, mixi11)));
}

/* The lineage's Gamma() takes the logarithm with FastLog2f (libjxl lib/jxl/base/fast_math-inl.h): range
 * reduction of the mantissa to [-1/3, 1/3] by integer arithmetic on the float's bits, then a (2,2) rational
 * polynomial of log2(1 + t), Horner form with fused multiply-adds, one IEEE division (|error| < 3e-6).  A libm
 * log2f here would make the result depend on the C library: on a flat image the reference's X = c0 - c1 is a
 * difference of two nearly equal values and one ulp of the logarithm moves the score by up to 4e-3 relative.
 * Restated with IEEE basic operations only, so host and device agree bit for bit. */
static float fast_log2f(float x)
{
    const float p0 = -1.8503833400518310E-06f, p1 = 1.4287160470083755E+00f, p2 = 7.4245873327820566E-01f;
    const float q0 = 9.9032814277590719E-01f, q1 = 1.0096718572241148E+00f, q2 = 1.7409343003366853E-01f;
    int32_t xb;
    memcpy(&xb, &x, 4);
    const int32_t eb = xb - 0x3f2aaaab;                               /* 0x3f2aaaab = 2/3 */
    const int32_t es = eb >> 23;                                      /* arithmetic shift: floor(log2(x / (2/3))) */
    const int32_t mb = xb - (int32_t)((uint32_t)es << 23);
    float m;
    memcpy(&m, &mb, 4);
    const float t = m - 1.0f;
    const float yp = fmaf(fmaf(p2, t, p1), t, p0);
    const float yq = fmaf(fmaf(q2, t, q1), t, q0);
    return yp / yq + (float)es;
}

extern int ceo_variant[CEO_V_COUNT]; /* sensitivity switches, all 0 by default (ce_oracle.h) */

static float gamma_f(float v)
{
    const float kRetMul = 19.245013259874995f * 0.693147180559945f;
    const float kRetAdd = -23.16046239805755f;
    if (v < 0.0f) v = 0.0f;
    const float biased = v + 9.9710635769299145f;
    const float lg = ceo_variant[CEO_V_BA_LIBM_LOG2] ? log2f(biased) : fast_log2f(biased);
    return fmaf(kRetMul, lg, kRetAdd);
}

/* rgb: 3 planes of linear [0,1]; out: 3 planes XYB */
static void opsin_dynamics_image(const img rgb[3], float intensity_target, img xyb[3])
{
    const size_t w = rgb[0].w, h = rgb[0].h, n = w * h;
    img blurred[3];
    for (int c = 0; c < 3; c++) {
        blurred[c] = img_new(w, h);
        blur(&rgb[c], 1.2f, &blurred[c]);
    }
    const float mn = 1e-4f;
    for (size_t i = 0; i < n; i++) {
        float p0, p1, p2;
        opsin_absorbance(blurred[0].p[i] * intensity_target, blurred[1].p[i] * intensity_target,
                         blurred[2].p[i] * intensity_target, &p0, &p1, &p2);
        p0 = p0 > mn ? p0 : mn;
        p1 = p1 > mn ? p1 : mn;
        p2 = p2 > mn ? p2 : mn;
        float s0 = gamma_f(p0) / p0, s1 = gamma_f(p1) / p1, s2 = gamma_f(p2) / p2;
        s0 = s0 > mn ? s0 : mn;
        s1 = s1 > mn ? s1 : mn;
        s2 = s2 > mn ? s2 : mn;
        float c0, c1, c2;
        opsin_absorbance(rgb[0].p[i] * intensity_target, rgb[1].p[i] * intensity_target,
                         rgb[2].p[i] * intensity_target, &c0, &c1, &c2);
        c0 *= s0;
        c1 *= s1;
        c2 *= s2;
        const float min01 = 1.7557483643287353f, min2 = 12.226454707163354f;
        c0 = c0 > min01 ? c0 : min01;
        c1 = c1 > min01 ? c1 : min01;
        c2 = c2 > min2 ? c2 : min2;
        xyb[0].p[i] = c0 - c1;
        xyb[1].p[i] = c0 + c1;
        xyb[2].p[i] = c2;
    }
    for (int c = 0; c < 3; c++) img_free(&blurred[c]);
}

/* ------------------------------------------------- frequency separation ---- */

typedef struct {
    img uhf[2], hf[2], mf[3], lf[3];
} psycho;

static void psycho_free(psycho *p)
{
    for (int i = 0; i < 2; i++) {
        img_free(&p->uhf[i]);
        img_free(&p->hf[i]);
    }
    for (int i = 0; i < 3; i++) {
        img_free(&p->mf[i]);
        img_free(&p->lf[i]);
    }
}

static float remove_range_around_zero(float w, float x) { return x > w ? x - w : (x < -w ? x + w : 0.0f); }
static float amplify_range_around_zero(float w, float x) { return x > w ? x + w : (x < -w ? x - w : x + x); }
static float maximum_clamp(float v, float maxval)
{
    const float kMul = 0.724216145665f;
    if (v >= maxval) return fmaf(v - maxval, kMul, maxval);
    if (v < -maxval) return fmaf(v + maxval, kMul, -maxval);
    return v;
}

static void separate_frequencies(const img xyb[3], psycho *ps)
{
    const size_t w = xyb[0].w, h = xyb[0].h, n = w * h;
    const float kSigmaLf = 7.15593339443f, kSigmaHf = 3.22489901262f, kSigmaUhf = 1.56416327805f;
    for (int i = 0; i < 3; i++) {
        ps->lf[i] = img_new(w, h);
        ps->mf[i] = img_new(w, h);
    }
    for (int i = 0; i < 2; i++) {
        ps->hf[i] = img_new(w, h);
        ps->uhf[i] = img_new(w, h);
    }
    for (int i = 0; i < 3; i++) {
        blur(&xyb[i], kSigmaLf, &ps->lf[i]);
        for (size_t k = 0; k < n; k++) ps->mf[i].p[k] = xyb[i].p[k] - ps->lf[i].p[k];
        if (i == 2) {
            img t = img_new(w, h);
            blur(&ps->mf[i], kSigmaHf, &t);
            memcpy(ps->mf[i].p, t.p, sizeof(float) * n);
            img_free(&t);
            break;
        }
        memcpy(ps->hf[i].p, ps->mf[i].p, sizeof(float) * n);
        img t = img_new(w, h);
        blur(&ps->mf[i], kSigmaHf, &t);
        memcpy(ps->mf[i].p, t.p, sizeof(float) * n);
        img_free(&t);
        const float kRemoveMfRange = 0.29f, kAddMfRange = 0.1f;
        for (size_t k = 0; k < n; k++) {
            float mf = ps->mf[i].p[k];
            float hf = ps->hf[i].p[k] - mf;
            mf = i == 0 ? remove_range_around_zero(kRemoveMfRange, mf) : amplify_range_around_zero(kAddMfRange, mf);
            ps->mf[i].p[k] = mf;
            ps->hf[i].p[k] = hf;
        }
    }
    /* SuppressXByY(hf[1], &hf[0]) */
    {
        const float suppress = 46.0f, s = 0.653020556257f, one_minus_s = 1.0f - 0.653020556257f;
        for (size_t k = 0; k < n; k++) {
            const float vx = ps->hf[0].p[k], vy = ps->hf[1].p[k];
            const float scaler = fmaf(suppress / fmaf(vy, vy, suppress), one_minus_s, s);
            ps->hf[0].p[k] = scaler * vx;
        }
    }
    for (int i = 0; i < 2; i++) {
        memcpy(ps->uhf[i].p, ps->hf[i].p, sizeof(float) * n);
        img t = img_new(w, h);
        blur(&ps->hf[i], kSigmaUhf, &t);
        memcpy(ps->hf[i].p, t.p, sizeof(float) * n);
        img_free(&t);
        const float kRemoveHfRange = 1.5f, kAddHfRange = 0.132f, kRemoveUhfRange = 0.04f;
        const float kMaxclampHf = 28.4691806922f, kMaxclampUhf = 5.19175294647f;
        const float kMulYHf = 2.155f, kMulYUhf = 2.69313763794f;
        for (size_t k = 0; k < n; k++) {
            float hf = ps->hf[i].p[k];
            if (i == 0) {
                float uhf = ps->uhf[i].p[k] - hf;
                hf = remove_range_around_zero(kRemoveHfRange, hf);
                uhf = remove_range_around_zero(kRemoveUhfRange, uhf);
                ps->hf[i].p[k] = hf;
                ps->uhf[i].p[k] = uhf;
            } else {
                hf = maximum_clamp(hf, kMaxclampHf);
                float uhf = ps->uhf[i].p[k] - hf;
                uhf = maximum_clamp(uhf, kMaxclampUhf);
                uhf *= kMulYUhf;
                ps->uhf[i].p[k] = uhf;
                hf *= kMulYHf;
                hf = amplify_range_around_zero(kAddHfRange, hf);
                ps->hf[i].p[k] = hf;
            }
        }
    }
    /* XybLowFreqToVals, in place on lf */
    {
        const float xmul = 33.832837186260f, ymul = 14.458268100570f, bmul = 49.87984651440f,
                    y_to_b_mul = -0.362267051518f;
        for (size_t k = 0; k < n; k++) {
            const float x = ps->lf[0].p[k], y = ps->lf[1].p[k], b_arg = ps->lf[2].p[k];
            const float b = fmaf(y_to_b_mul, y, b_arg);
            ps->lf[2].p[k] = b * bmul;
            ps->lf[0].p[k] = x * xmul;
            ps->lf[1].p[k] = y * ymul;
        }
    }
}

/* -------------------------------------------------------------- Malta ---- */

/* zero outside the image (PaddedMaltaUnit) */
static float px(const img *a, ptrdiff_t x, ptrdiff_t y)
{
    if (x < 0 || y < 0 || x >= (ptrdiff_t)a->w || y >= (ptrdiff_t)a->h) return 0.0f;
    return a->p[(size_t)y * a->w + (size_t)x];
}

/* the 16 line patterns as (dx,dy) lists; 0,0 (the centre) is included where the
 * lineage's sum includes it */
typedef struct {
    int n;
    signed char d[9][2];
} malta_line;

static const malta_line MALTA_HF[16] = {
    {9, {{-4, 0}, {-3, 0}, {-2, 0}, {-1, 0}, {0, 0}, {1, 0}, {2, 0}, {3, 0}, {4, 0}}},
    {9, {{0, -4}, {0, -3}, {0, -2}, {0, -1}, {0, 0}, {0, 1}, {0, 2}, {0, 3}, {0, 4}}},
    {7, {{-3, -3}, {-2, -2}, {-1, -1}, {0, 0}, {1, 1}, {2, 2}, {3, 3}}},
    {7, {{3, -3}, {2, -2}, {1, -1}, {0, 0}, {-1, 1}, {-2, 2}, {-3, 3}}},
    {9, {{1, -4}, {1, -3}, {1, -2}, {0, -1}, {0, 0}, {0, 1}, {-1, 2}, {-1, 3}, {-1, 4}}},
    {9, {{-1, -4}, {-1, -3}, {-1, -2}, {0, -1}, {0, 0}, {0, 1}, {1, 2}, {1, 3}, {1, 4}}},
    {9, {{-4, -1}, {-3, -1}, {-2, -1}, {-1, 0}, {0, 0}, {1, 0}, {2, 1}, {3, 1}, {4, 1}}},
    {9, {{-4, 1}, {-3, 1}, {-2, 1}, {-1, 0}, {0, 0}, {1, 0}, {2, -1}, {3, -1}, {4, -1}}},
    {7, {{-2, -3}, {-1, -2}, {-1, -1}, {0, 0}, {1, 1}, {1, 2}, {2, 3}}},
    {7, {{2, -3}, {1, -2}, {1, -1}, {0, 0}, {-1, 1}, {-1, 2}, {-2, 3}}},
    {7, {{-3, -2}, {-2, -1}, {-1, -1}, {0, 0}, {1, 1}, {2, 1}, {3, 2}}},
    {7, {{3, -2}, {2, -1}, {1, -1}, {0, 0}, {-1, 1}, {-2, 1}, {-3, 2}}},
    {8, {{-4, 2}, {-3, 2}, {-2, 1}, {-1, 1}, {0, 0}, {1, 0}, {2, -1}, {3, -1}}},
    {8, {{-4, -2}, {-3, -2}, {-2, -1}, {-1, -1}, {0, 0}, {1, 0}, {2, 1}, {3, 1}}},
    {8, {{-2, -4}, {-2, -3}, {-1, -2}, {-1, -1}, {0, 0}, {0, 1}, {1, 2}, {1, 3}}},
    {8, {{2, -4}, {2, -3}, {1, -2}, {1, -1}, {0, 0}, {0, 1}, {-1, 2}, {-1, 3}}},
};

static const malta_line MALTA_LF[16] = {
    {5, {{-4, 0}, {-2, 0}, {0, 0}, {2, 0}, {4, 0}}},
    {5, {{0, -4}, {0, -2}, {0, 0}, {0, 2}, {0, 4}}},
    {5, {{-3, -3}, {-2, -2}, {0, 0}, {2, 2}, {3, 3}}},
    {5, {{3, -3}, {2, -2}, {0, 0}, {-2, 2}, {-3, 3}}},
    {5, {{1, -4}, {1, -2}, {0, 0}, {-1, 2}, {-1, 4}}},
    {5, {{-1, -4}, {-1, -2}, {0, 0}, {1, 2}, {1, 4}}},
    {5, {{-4, -1}, {-2, -1}, {0, 0}, {2, 1}, {4, 1}}},
    {5, {{-4, 1}, {-2, 1}, {0, 0}, {2, -1}, {4, -1}}},
    {5, {{-2, -3}, {-1, -2}, {0, 0}, {1, 2}, {2, 3}}},
    {5, {{2, -3}, {1, -2}, {0, 0}, {-1, 2}, {-2, 3}}},
    {5, {{-3, -2}, {-2, -1}, {0, 0}, {2, 1}, {3, 2}}},
    {5, {{3, -2}, {2, -1}, {0, 0}, {-2, 1}, {-3, 2}}},
    {5, {{-4, 2}, {-2, 1}, {0, 0}, {2, -1}, {4, -2}}},
    {5, {{-4, -2}, {-2, -1}, {0, 0}, {2, 1}, {4, 2}}},
    {5, {{-2, -4}, {-1, -2}, {0, 0}, {1, 2}, {2, 4}}},
    {5, {{2, -4}, {1, -2}, {0, 0}, {-1, 2}, {-2, 4}}},
};

static float malta_unit(const img *d, ptrdiff_t x, ptrdiff_t y, const malta_line *lines)
{
    float ret = 0.0f;
    for (int k = 0; k < 16; k++) {
        float sum = 0.0f;
        for (int j = 0; j < lines[k].n; j++) sum += px(d, x + lines[k].d[j][0], y + lines[k].d[j][1]);
        ret = fmaf(sum, sum, ret);
    }
    return ret;
}

/* MaltaDiffMap / MaltaDiffMapLF: lum0 is the original, lum1 the distorted */
static void malta_diff_map(const img *lum0, const img *lum1, double w_0gt1, double w_0lt1, double norm1, int lf,
                           img *block_diff_ac)
{
    const size_t w = lum0->w, h = lum0->h, n = w * h;
    const double len = 3.75;
    const double mulli = lf ? 0.611612573796 : 0.39905817637;
    const float kWeight0 = 0.5f, kWeight1 = 0.33f;
    const double w_pre0gt1 = mulli * sqrt(kWeight0 * w_0gt1) / (len * 2 + 1);
    const double w_pre0lt1 = mulli * sqrt(kWeight1 * w_0lt1) / (len * 2 + 1);
    const float norm2_0gt1 = (float)(w_pre0gt1 * norm1);
    const float norm2_0lt1 = (float)(w_pre0lt1 * norm1);
    img diffs = img_new(w, h);
    for (size_t k = 0; k < n; k++) {
        const float v0 = lum0->p[k], v1 = lum1->p[k];
        const float absval = 0.5f * (fabsf(v0) + fabsf(v1));
        const float diff = v0 - v1;
        const float scaler = norm2_0gt1 / ((float)norm1 + absval);
        float r = scaler * diff;
        const float scaler2 = norm2_0lt1 / ((float)norm1 + absval);
        const double fabs0 = fabs((double)v0);
        const double too_small = 0.55 * fabs0, too_big = 1.05 * fabs0;
        if (ceo_variant[CEO_V_BA_MALTA_F32]) { /* the same four branches with f32 arithmetic */
            const float fa = fabsf(v0), ts = 0.55f * fa, tb = 1.05f * fa;
            if (v0 < 0) {
                if (v1 > -ts) r = r - scaler2 * (v1 + ts);
                else if (v1 < -tb) r = r + scaler2 * (-v1 - tb);
            } else {
                if (v1 < ts) r = r + scaler2 * (ts - v1);
                else if (v1 > tb) r = r - scaler2 * (v1 - tb);
            }
            diffs.p[k] = r;
            continue;
        }
        if (v0 < 0) {
            if (v1 > -too_small) {
                double impact = scaler2 * (v1 + too_small);
                r = (float)(r - impact);
            } else if (v1 < -too_big) {
                double impact = scaler2 * (-v1 - too_big);
                r = (float)(r + impact);
            }
        } else {
            if (v1 < too_small) {
                double impact = scaler2 * (too_small - v1);
                r = (float)(r + impact);
            } else if (v1 > too_big) {
                double impact = scaler2 * (v1 - too_big);
                r = (float)(r - impact);
            }
        }
        diffs.p[k] = r;
    }
    const malta_line *lines = lf ? MALTA_LF : MALTA_HF;
    for (size_t y = 0; y < h; y++)
        for (size_t x = 0; x < w; x++)
            block_diff_ac->p[y * w + x] += malta_unit(&diffs, (ptrdiff_t)x, (ptrdiff_t)y, lines);
    img_free(&diffs);
}

/* ------------------------------------------------------------ L2 diffs ---- */

static void l2_diff(const img *i0, const img *i1, float w, img *diffmap, int set)
{
    if (w == 0 && !set) return;
    const size_t n = i0->w * i0->h;
    for (size_t k = 0; k < n; k++) {
        const float diff = i0->p[k] - i1->p[k];
        if (set)
            diffmap->p[k] = (diff * diff) * w;
        else
            diffmap->p[k] = fmaf(diff * diff, w, diffmap->p[k]);
    }
}

static void l2_diff_asymmetric(const img *i0, const img *i1, float w_0gt1, float w_0lt1, img *diffmap)
{
    if (w_0gt1 == 0 && w_0lt1 == 0) return;
    const float vw_0gt1 = w_0gt1 * 0.8f, vw_0lt1 = w_0lt1 * 0.8f;
    const size_t n = i0->w * i0->h;
    for (size_t k = 0; k < n; k++) {
        const float val0 = i0->p[k], val1 = i1->p[k];
        const float diff = val0 - val1;
        float total = fmaf(diff * diff, vw_0gt1, diffmap->p[k]);
        const float fabs0 = fabsf(val0);
        const float too_small = 0.4f * fabs0, too_big = fabs0;
        const float if_neg = val1 > -too_small ? val1 + too_small : (val1 < -too_big ? -val1 - too_big : 0.0f);
        const float if_pos = val1 < too_small ? too_small - val1 : (val1 > too_big ? val1 - too_big : 0.0f);
        const float v = val0 < 0.0f ? if_neg : if_pos;
        total = fmaf(vw_0lt1, v * v, total);
        diffmap->p[k] = total;
    }
}

/* ---------------------------------------------------------------- mask ---- */

static void diff_precompute(const img *in, float mul, float bias_arg, img *out)
{
    const float bias = mul * bias_arg;
    const float sqrt_bias = sqrtf(bias);
    const size_t n = in->w * in->h;
    for (size_t k = 0; k < n; k++) out->p[k] = sqrtf(mul * fabsf(in->p[k]) + bias) - sqrt_bias;
}

static void store_min3(float v, float *min0, float *min1, float *min2)
{
    if (v < *min2) {
        if (v < *min0) {
            *min2 = *min1;
            *min1 = *min0;
            *min0 = v;
        } else if (v < *min1) {
            *min2 = *min1;
            *min1 = v;
        } else {
            *min2 = v;
        }
    }
}

static void fuzzy_erosion(const img *from, img *to)
{
    const ptrdiff_t w = (ptrdiff_t)from->w, h = (ptrdiff_t)from->h, S = 3;
    for (ptrdiff_t y = 0; y < h; y++)
        for (ptrdiff_t x = 0; x < w; x++) {
            float min0 = from->p[y * w + x];
            float min1 = 2 * min0, min2 = min1;
            if (x >= S) {
                store_min3(from->p[y * w + x - S], &min0, &min1, &min2);
                if (y >= S) store_min3(from->p[(y - S) * w + x - S], &min0, &min1, &min2);
                if (y < h - S) store_min3(from->p[(y + S) * w + x - S], &min0, &min1, &min2);
            }
            if (x < w - S) {
                store_min3(from->p[y * w + x + S], &min0, &min1, &min2);
                if (y >= S) store_min3(from->p[(y - S) * w + x + S], &min0, &min1, &min2);
                if (y < h - S) store_min3(from->p[(y + S) * w + x + S], &min0, &min1, &min2);
            }
            if (y >= S) store_min3(from->p[(y - S) * w + x], &min0, &min1, &min2);
            if (y < h - S) store_min3(from->p[(y + S) * w + x], &min0, &min1, &min2);
            to->p[y * w + x] = 0.45f * min0 + 0.3f * min1 + 0.25f * min2;
        }
}

/* MaskPsychoImage + Mask: mask from image 0; diff_ac (the Y plane) gets the mask-difference term */
static void mask_psycho_image(const psycho *pi0, const psycho *pi1, img *mask, img *diff_ac)
{
    const size_t w = pi0->hf[0].w, h = pi0->hf[0].h, n = w * h;
    img mask0 = img_new(w, h), mask1 = img_new(w, h);
    const float muls[3] = {2.5f, 0.4f, 0.4f};
    for (size_t k = 0; k < n; k++) {
        const float xdiff0 = (pi0->uhf[0].p[k] + pi0->hf[0].p[k]) * muls[0];
        const float xdiff1 = (pi1->uhf[0].p[k] + pi1->hf[0].p[k]) * muls[0];
        const float ydiff0 = pi0->uhf[1].p[k] * muls[1] + pi0->hf[1].p[k] * muls[2];
        const float ydiff1 = pi1->uhf[1].p[k] * muls[1] + pi1->hf[1].p[k] * muls[2];
        mask0.p[k] = sqrtf(xdiff0 * xdiff0 + ydiff0 * ydiff0);
        mask1.p[k] = sqrtf(xdiff1 * xdiff1 + ydiff1 * ydiff1);
    }
    const float kMul = 6.19424080439f, kBias = 12.61050594197f, kRadius = 2.7f;
    img diff0 = img_new(w, h), diff1 = img_new(w, h), blurred0 = img_new(w, h), blurred1 = img_new(w, h);
    diff_precompute(&mask0, kMul, kBias, &diff0);
    diff_precompute(&mask1, kMul, kBias, &diff1);
    blur(&diff0, kRadius, &blurred0);
    fuzzy_erosion(&blurred0, &diff0);
    blur(&diff1, kRadius, &blurred1);
    const float kMaskToErrorMul = 10.0f;
    for (size_t k = 0; k < n; k++) {
        mask->p[k] = diff0.p[k];
        const float diff = blurred0.p[k] - blurred1.p[k];
        diff_ac->p[k] += kMaskToErrorMul * diff * diff;
    }
    img_free(&mask0); img_free(&mask1); img_free(&diff0); img_free(&diff1);
    img_free(&blurred0); img_free(&blurred1);
}

static const double kInternalGoodQualityThreshold = 17.83 * 0.790799174;

static double mask_y(double delta)
{
    const double offset = 0.829591754942, scaler = 0.451936922203, mul = 2.5485944793;
    const double c = mul / ((scaler * delta) + offset);
    const double retval = (1.0 / kInternalGoodQualityThreshold) * (1.0 + c);
    return retval * retval;
}
static double mask_dc_y(double delta)
{
    const double offset = 0.20025578522, scaler = 3.87449418804, mul = 0.505054525019;
    const double c = mul / ((scaler * delta) + offset);
    const double retval = (1.0 / kInternalGoodQualityThreshold) * (1.0 + c);
    return retval * retval;
}

/* ------------------------------------------------------------ diffmap ---- */

static void diffmap_psycho(const psycho *pi0, const psycho *pi1, img *diffmap)
{
    const size_t w = pi0->hf[0].w, h = pi0->hf[0].h, n = w * h;
    const float hf_asymmetry = 1.0f, xmul = 1.0f; /* ButteraugliParams::default() */
    img ac[3], dc[3];
    for (int c = 0; c < 3; c++) {
        ac[c] = img_new(w, h);
        dc[c] = img_new(w, h);
    }
    const float wmul[9] = {400.0f, 1.50815703118f, 0.0f, 2150.0f, 10.6195433239f, 16.2176043152f,
                           29.2353797994f, 0.844626970982f, 0.703646627719f};
    /* CEO_V_BA_L2_EARLY: the in-place L2 accumulations of channels X and Y run between the Malta bands (the order the
     * device's fused kernel meets the bands in); default: after all Malta bands, as the lineage calls them */
    const int early = ceo_variant[CEO_V_BA_L2_EARLY];
    const double wUhfMalta = 1.10039032555, norm1Uhf = 71.7800275169;
    malta_diff_map(&pi0->uhf[1], &pi1->uhf[1], wUhfMalta * hf_asymmetry, wUhfMalta / hf_asymmetry, norm1Uhf, 0, &ac[1]);
    const double wUhfMaltaX = 173.5, norm1UhfX = 5.0;
    malta_diff_map(&pi0->uhf[0], &pi1->uhf[0], wUhfMaltaX * hf_asymmetry, wUhfMaltaX / hf_asymmetry, norm1UhfX, 0, &ac[0]);
    if (early)
        for (int c = 0; c < 2; c++) l2_diff_asymmetric(&pi0->hf[c], &pi1->hf[c], wmul[c] * hf_asymmetry, wmul[c] / hf_asymmetry, &ac[c]);
    const double wHfMalta = 18.7237414387, norm1Hf = 4498534.45232;
    malta_diff_map(&pi0->hf[1], &pi1->hf[1], wHfMalta * sqrt(hf_asymmetry), wHfMalta / sqrt(hf_asymmetry), norm1Hf, 1, &ac[1]);
    const double wHfMaltaX = 6923.99476109, norm1HfX = 8051.15833247;
    malta_diff_map(&pi0->hf[0], &pi1->hf[0], wHfMaltaX * sqrt(hf_asymmetry), wHfMaltaX / sqrt(hf_asymmetry), norm1HfX, 1, &ac[0]);
    if (early)
        for (int c = 0; c < 2; c++) l2_diff(&pi0->mf[c], &pi1->mf[c], wmul[3 + c], &ac[c], 0);
    const double wMfMalta = 37.0819870399, norm1Mf = 130262059.556;
    malta_diff_map(&pi0->mf[1], &pi1->mf[1], wMfMalta, wMfMalta, norm1Mf, 1, &ac[1]);
    const double wMfMaltaX = 8246.75321353, norm1MfX = 1009002.70582;
    malta_diff_map(&pi0->mf[0], &pi1->mf[0], wMfMaltaX, wMfMaltaX, norm1MfX, 1, &ac[0]);

    for (int c = 0; c < 3; c++) {
        if (c < 2 && !early) l2_diff_asymmetric(&pi0->hf[c], &pi1->hf[c], wmul[c] * hf_asymmetry, wmul[c] / hf_asymmetry, &ac[c]);
        if (c == 2 || !early) l2_diff(&pi0->mf[c], &pi1->mf[c], wmul[3 + c], &ac[c], 0);
        l2_diff(&pi0->lf[c], &pi1->lf[c], wmul[6 + c], &dc[c], 1);
    }
    img mask = img_new(w, h);
    mask_psycho_image(pi0, pi1, &mask, &ac[1]);
    /* CombineChannelsToDiffmap */
    for (size_t k = 0; k < n; k++) {
        const float val = mask.p[k];
        const float maskval = (float)mask_y(val), dc_maskval = (float)mask_dc_y(val);
        float diff_dc[3], diff_ac[3];
        for (int c = 0; c < 3; c++) {
            diff_dc[c] = dc[c].p[k];
            diff_ac[c] = ac[c].p[k];
        }
        diff_ac[0] *= xmul;
        diff_dc[0] *= xmul;
        const float mc_dc = diff_dc[0] * dc_maskval + diff_dc[1] * dc_maskval + diff_dc[2] * dc_maskval;
        const float mc_ac = diff_ac[0] * maskval + diff_ac[1] * maskval + diff_ac[2] * maskval;
        diffmap->p[k] = sqrtf(mc_dc + mc_ac);
    }
    img_free(&mask);
    for (int c = 0; c < 3; c++) {
        img_free(&ac[c]);
        img_free(&dc[c]);
    }
}

static void subsample2x(const img in[3], img out[3])
{
    const size_t w = in[0].w, h = in[0].h, ow = (w + 1) / 2, oh = (h + 1) / 2;
    for (int c = 0; c < 3; c++) {
        out[c] = img_new(ow, oh);
        for (size_t y = 0; y < h; y++)
            for (size_t x = 0; x < w; x++) out[c].p[(y / 2) * ow + x / 2] += 0.25f * in[c].p[y * w + x];
        if (w & 1)
            for (size_t y = 0; y < oh; y++) out[c].p[y * ow + ow - 1] *= 2.0f;
        if (h & 1)
            for (size_t x = 0; x < ow; x++) out[c].p[(oh - 1) * ow + x] *= 2.0f;
    }
}

/* one resolution level: linear RGB pair -> diffmap */
static void diffmap_level(const img rgb0[3], const img rgb1[3], float intensity_target, img *diffmap)
{
    const size_t w = rgb0[0].w, h = rgb0[0].h;
    img xyb0[3], xyb1[3];
    for (int c = 0; c < 3; c++) {
        xyb0[c] = img_new(w, h);
        xyb1[c] = img_new(w, h);
    }
    opsin_dynamics_image(rgb0, intensity_target, xyb0);
    opsin_dynamics_image(rgb1, intensity_target, xyb1);
    psycho p0, p1;
    separate_frequencies(xyb0, &p0);
    separate_frequencies(xyb1, &p1);
    diffmap_psycho(&p0, &p1, diffmap);
    psycho_free(&p0);
    psycho_free(&p1);
    for (int c = 0; c < 3; c++) {
        img_free(&xyb0[c]);
        img_free(&xyb1[c]);
    }
}

int ceo_butteraugli(const uint8_t *ref, size_t ref_len, const uint8_t *test, size_t test_len, size_t width,
                    size_t height, float intensity_target, double *score, double *pnorm3)
{
    if (ref_len != test_len) return CEO_DIM_MISMATCH;         /* butteraugli.rs:51-56 */
    if (ref_len != width * height * 3) return CEO_BAD_LENGTH; /* :58-68 */
    if (width < 8 || height < 8) return CEO_TOO_SMALL;        /* helpers.rs:89 */
    const size_t w = width, h = height, n = w * h;
    img rgb0[3], rgb1[3];
    for (int c = 0; c < 3; c++) {
        rgb0[c] = img_new(w, h);
        rgb1[c] = img_new(w, h);
    }
    float lut[256];
    for (int i = 0; i < 256; i++) {
        double v = (double)i / 255.0;
        lut[i] = (float)(v <= 0.04045 ? v / 12.92 : pow((v + 0.055) / 1.055, 2.4));
    }
    for (size_t i = 0; i < n; i++)
        for (int c = 0; c < 3; c++) {
            rgb0[c].p[i] = lut[ref[3 * i + c]];
            rgb1[c].p[i] = lut[test[3 * i + c]];
        }
    img diffmap = img_new(w, h);
    diffmap_level(rgb0, rgb1, intensity_target, &diffmap);
    /* the half-resolution pass, added supersampled (AddSupersampled2x, weight 0.5) */
    img s0[3], s1[3];
    subsample2x(rgb0, s0);
    subsample2x(rgb1, s1);
    if (s0[0].w >= 8 && s0[0].h >= 8) {
        img sub = img_new(s0[0].w, s0[0].h);
        diffmap_level(s0, s1, intensity_target, &sub);
        const float kHeuristicMixingValue = 0.3f, wgt = 0.5f;
        for (size_t y = 0; y < h; y++)
            for (size_t x = 0; x < w; x++) {
                float *d = &diffmap.p[y * w + x];
                *d *= 1.0f - kHeuristicMixingValue * wgt;
                *d += wgt * sub.p[(y / 2) * sub.w + x / 2];
            }
        img_free(&sub);
    }
    /* ButteraugliScoreFromDiffmap: max; ComputeDistanceP with p = 3 */
    float mx = 0.0f;
    double sum1[3] = {0, 0, 0};
    for (size_t i = 0; i < n; i++) {
        const float d = diffmap.p[i];
        if (d > mx) mx = d;
        const double dd = d, d3 = dd * dd * dd, d6 = d3 * d3;
        sum1[0] += d3;
        sum1[1] += d6;
        sum1[2] += d6 * d6;
    }
    *score = (double)mx;
    if (pnorm3) {
        const double one_per_pixels = 1.0 / (double)n;
        double v = pow(one_per_pixels * sum1[0], 1.0 / 3.0) + pow(one_per_pixels * sum1[1], 1.0 / 6.0) +
                   pow(one_per_pixels * sum1[2], 1.0 / 12.0);
        *pnorm3 = v / 3.0;
    }
    img_free(&diffmap);
    for (int c = 0; c < 3; c++) {
        img_free(&rgb0[c]); img_free(&rgb1[c]); img_free(&s0[c]); img_free(&s1[c]);
    }
    return CEO_OK;
}
