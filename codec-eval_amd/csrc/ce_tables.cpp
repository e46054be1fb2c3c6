// Host-built constant tables for the device kernels.  Built once per context with the host
// libm, because the reference computes the same quantities with the host libm
// (Rust f32::powf -> powf, f64::powf -> pow).
#include <cmath>
#include <cstring>

#include "ce_internal.h"

// sRGB u8 -> linear, evaluated in f64 per code point and rounded once to f32: the front end
// of SSIMULACRA2.
void ce_build_srgb_lut_f64(float lut[256])
{
    for (int i = 0; i < 256; i++) {
        const double v = (double)i / 255.0;
        lut[i] = (float)(v <= 0.04045 ? v / 12.92 : std::pow((v + 0.055) / 1.055, 2.4));
    }
}

// sRGB u8 -> linear exactly as /root/reference/src/metrics/dssim.rs:78-85 and
// src/metrics/xyb.rs:60-66,80-82 write it: f32 arithmetic, f32 powf(2.4).
void ce_build_srgb_lut_powf(float lut[256])
{
    for (int i = 0; i < 256; i++) {
        const float s = (float)i / 255.0f;
        lut[i] = s <= 0.04045f ? s / 12.92f : powf((s + 0.055f) / 1.055f, 2.4f);
    }
}

// Coefficients of the sigma = 1.5 recursive Gaussian (Charalampidis 2016 truncated-cosine
// form as derived in libjxl's CreateRecursiveGaussian):
// three second-order sections k = 1,3,5 with   out_k[n] = n2_k (in[n-N-1] + in[n+N-1])
//                                                          - d1_k out_k[n-1] - out_k[n-2].
// mul_in = n2, mul_prev = -d1, both rounded to f32.
void ce_ssim2_recursive_gaussian(float mul_in[3], float mul_prev[3])
{
    const double sigma = 1.5;
    const double radius = std::round(3.2795 * sigma + 0.2546);
    const double pi_div_2r = M_PI / (2.0 * radius);
    const double omega[3] = {pi_div_2r, 3.0 * pi_div_2r, 5.0 * pi_div_2r};
    const double p1 = +1.0 / std::tan(0.5 * omega[0]);
    const double p3 = -1.0 / std::tan(0.5 * omega[1]);
    const double p5 = +1.0 / std::tan(0.5 * omega[2]);
    const double r1 = +p1 * p1 / std::sin(omega[0]);
    const double r3 = -p3 * p3 / std::sin(omega[1]);
    const double r5 = +p5 * p5 / std::sin(omega[2]);
    const double neg_half_sigma2 = -0.5 * sigma * sigma;
    const double recip_radius = 1.0 / radius;
    double rho[3];
    for (int i = 0; i < 3; i++) rho[i] = std::exp(neg_half_sigma2 * omega[i] * omega[i]) * recip_radius;
    const double D13 = p1 * r3 - r1 * p3;
    const double D35 = p3 * r5 - r3 * p5;
    const double D51 = p5 * r1 - r5 * p1;
    const double recip_d13 = 1.0 / D13;
    const double zeta15 = D35 * recip_d13;
    const double zeta35 = D51 * recip_d13;
    // beta = A^-1 gamma with A = [[p1 p3 p5][r1 r3 r5][zeta15 zeta35 1]]
    const double a = p1, b = p3, c = p5, d = r1, e = r3, f = r5, g = zeta15, h = zeta35, i = 1.0;
    const double det = a * (e * i - f * h) - b * (d * i - f * g) + c * (d * h - e * g);
    const double id = 1.0 / det;
    const double inv[9] = {(e * i - f * h) * id, (c * h - b * i) * id, (b * f - c * e) * id,
                           (f * g - d * i) * id, (a * i - c * g) * id, (c * d - a * f) * id,
                           (d * h - e * g) * id, (b * g - a * h) * id, (a * e - b * d) * id};
    const double gamma[3] = {1.0, radius * radius - sigma * sigma, zeta15 * rho[0] + zeta35 * rho[1] + rho[2]};
    for (int k = 0; k < 3; k++) {
        const double beta = inv[3 * k] * gamma[0] + inv[3 * k + 1] * gamma[1] + inv[3 * k + 2] * gamma[2];
        mul_in[k] = (float)(-beta * std::cos(omega[k] * (radius + 1.0)));
        mul_prev[k] = (float)(2.0 * std::cos(omega[k]));
    }
}

// linear -> sRGB u8 exactly as /root/reference/src/metrics/xyb.rs:70-76,86-88 for an already
// clamped input c in [0,1]
static int xyb_linear_to_srgb_u8_host(float c)
{
    const float e = c <= 0.0031308f ? c * 12.92f : 1.055f * powf(c, 1.0f / 2.4f) - 0.055f;
    const float r = roundf(e * 255.0f);
    if (!(r > 0.0f)) return 0;
    if (r > 255.0f) return 255;
    return (int)r;
}

// thresh[k] (k = 1..255) = the smallest f32 c in [0,1] whose u8 code is >= k under the host's
// powf; thresh[0] = -inf.  The code is a monotone step function of c, so
// code(c) = #{k : thresh[k] <= c}.  Returns false if monotonicity fails in the +-256 ulp
// neighbourhood of any threshold (then the table cannot represent the host function).
bool ce_build_xyb_srgb_thresholds(float thresh[256])
{
    auto f = [](uint32_t bits) {
        float c;
        memcpy(&c, &bits, 4);
        return xyb_linear_to_srgb_u8_host(c);
    };
    const uint32_t one = 0x3f800000u;
    thresh[0] = -INFINITY;
    bool ok = true;
    for (int k = 1; k <= 255; k++) {
        if (f(one) < k) {
            thresh[k] = INFINITY;
            continue;
        }
        uint32_t lo = 0, hi = one;  // f(lo) < k <= f(hi); non-negative floats order like their bits
        if (f(lo) >= k) {
            hi = 0;
        } else {
            while (hi - lo > 1) {
                const uint32_t mid = lo + (hi - lo) / 2;
                if (f(mid) >= k) hi = mid; else lo = mid;
            }
        }
        memcpy(&thresh[k], &hi, 4);
        const uint32_t a = hi > 256 ? hi - 256 : 0, b = hi + 256 < one ? hi + 256 : one;
        for (uint32_t u = a; u <= b; u++) {
            const bool above = f(u) >= k;
            if (above != (u >= hi)) ok = false;
        }
    }
    return ok;
}
