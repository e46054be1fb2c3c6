// XYB roundtrip on gfx950 — replaces xyb_roundtrip (/root/reference/src/metrics/xyb.rs:225-253):
// sRGB u8 -> linear -> XYB -> quantise each channel to 255 steps -> linear -> sRGB u8, all f32.
//
// u8-exactness against the reference's arithmetic (which calls the HOST libm for powf/cbrtf):
//   * sRGB->linear (xyb.rs:60-66,80-82): only 256 inputs exist -> table built by the host with
//     the host's powf (ce_tables.cpp), read from LDS.
//   * cbrt (xyb.rs:92-94): PINNED to glibc 2.35's cbrtf (frexp / quadratic seed / one Halley step in
//     double / ldexp), restated below with IEEE basic operations only.  The oracle runs the same
//     restatement (ceo_cbrtf_pinned, oracle/psnr_xyb.c), so device == oracle on any host whatever its
//     libm; tests/test_oracle_pinning.py::test_xyb_cbrtf_is_pinned_in_one_place compares the pin with
//     the host's cbrtf, tests/test_gpu_parity.py checks the full 2^24-colour cube against the oracle.
//   * linear->sRGB u8 (xyb.rs:70-76,86-88): round(255*(1.055*powf(c,1/2.4)-0.055)) is a
//     monotone step function of the clamped f32 input, so it is decided by 255 thresholds
//     that the host finds with its own powf (ce_tables.cpp); the device only compares.
//   * everything else is f32 +,-,*,/ and roundf in the reference's order; build with
//     -ffp-contract=off so a*b+c is never fused (rustc does not fuse it either).
#include "ce_internal.h"

namespace {

constexpr int kThreads = 256;

// xyb.rs:33-56
__device__ constexpr float M00 = 0.30f, M01 = 0.622f, M02 = 0.078f;
__device__ constexpr float M10 = 0.23f, M11 = 0.692f, M12 = 0.078f;
__device__ constexpr float M20 = 0.24342269f, M21 = 0.20476744f, M22 = 0.55180987f;
__device__ constexpr float BIAS = 0.0037930733f;
__device__ constexpr float NEG_BIAS_CBRT = -0.15595412f;
__device__ constexpr float I00 = 11.031567f, I01 = -9.866944f, I02 = -0.164623f;
__device__ constexpr float I10 = -3.254147f, I11 = 4.41877f, I12 = -0.164623f;
__device__ constexpr float I20 = -3.658851f, I21 = 2.712923f, I22 = 1.945928f;
// xyb.rs:185-190
__device__ constexpr float X_MIN = -0.016f, X_MAX = 0.029f, Y_MIN = 0.0f, Y_MAX = 0.846f, B_MIN = 0.0f, B_MAX = 0.846f;

// glibc 2.35 sysdeps/ieee754/flt-32/s_cbrtf.c restated for positive normal x
__device__ __forceinline__ float cbrtf_glibc_pos(float x)
{
    const uint32_t bits = __float_as_uint(x);
    const int xe = (int)(bits >> 23) - 126;                                  // frexpf exponent
    const float xm = __uint_as_float((bits & 0x007fffffu) | 0x3f000000u);     // mantissa in [0.5,1)
    const float u = (float)(0.492659620528969547 + (0.697570460207922770 - 0.191502161678719066 * (double)xm) * (double)xm);
    const float t2 = u * u * u;
    const int r = xe % 3;  // C remainder, sign follows xe
    const double factor = r == -2 ? 1.0 / 1.5874010519681994748
                        : r == -1 ? 1.0 / 1.2599210498948731648
                        : r == 0  ? 1.0
                        : r == 1  ? 1.2599210498948731648
                                  : 1.5874010519681994748;
    const float ym = (float)((double)u * ((double)t2 + 2.0 * (double)xm) / (2.0 * (double)t2 + (double)xm) * factor);
    // ldexpf(ym, xe/3): ym is in [0.5,2) and the results stay normal for every opsin value
    return __uint_as_float(__float_as_uint(ym) + ((uint32_t)(xe / 3) << 23));
}

// xyb.rs:92-94 (mixed_cbrt) — zero / subnormal / non-finite inputs cannot occur here
// (opsin >= bias > 0 and <= ~1.004), but keep the sign rule
__device__ __forceinline__ float mixed_cbrt(float v)
{
    if (v < 0.0f) return -cbrtf_glibc_pos(-v);
    return cbrtf_glibc_pos(v);
}

// xyb.rs:98-100
__device__ __forceinline__ float mixed_cube(float v)
{
    if (v < 0.0f) {
        const float n = -v;
        return -((n * n) * n);
    }
    return (v * v) * v;
}

// xyb.rs:194-199
__device__ __forceinline__ float quantize_to_u8(float value, float mn, float mx)
{
    const float range = mx - mn;
    const float normalized = (value - mn) / range;
    float q = roundf(normalized * 255.0f);
    q = q < 0.0f ? 0.0f : (q > 255.0f ? 255.0f : q);
    const float quantized = q / 255.0f;
    return quantized * range + mn;
}

// xyb.rs:86-88 through the host-built thresholds: result = #{k in 1..255 : thresh[k] <= clamp(v)}
__device__ __forceinline__ uint32_t linear_to_srgb_u8(float v, const float *thresh)
{
    const float c = v < 0.0f ? 0.0f : (v > 1.0f ? 1.0f : v);
    uint32_t k = 0;
#pragma unroll
    for (uint32_t step = 128; step > 0; step >>= 1)
        if (k + step <= 255 && thresh[k + step] <= c) k += step;
    return k;
}

__device__ __forceinline__ void roundtrip_pixel(uint32_t r8, uint32_t g8, uint32_t b8, const float *lut,
                                                const float *thresh, uint32_t &ro, uint32_t &go, uint32_t &bo)
{
    const float r = lut[r8], g = lut[g8], b = lut[b8];
    // linear_rgb_to_xyb, xyb.rs:104-129
    const float opsin_r = M00 * r + M01 * g + M02 * b + BIAS;
    const float opsin_g = M10 * r + M11 * g + M12 * b + BIAS;
    const float opsin_b = M20 * r + M21 * g + M22 * b + BIAS;
    const float cbrt_r = mixed_cbrt(opsin_r) + NEG_BIAS_CBRT;
    const float cbrt_g = mixed_cbrt(opsin_g) + NEG_BIAS_CBRT;
    const float cbrt_b = mixed_cbrt(opsin_b) + NEG_BIAS_CBRT;
    const float x = 0.5f * (cbrt_r - cbrt_g);
    const float y = 0.5f * (cbrt_r + cbrt_g);
    // xyb.rs:240-242
    const float xq = quantize_to_u8(x, X_MIN, X_MAX);
    const float yq = quantize_to_u8(y, Y_MIN, Y_MAX);
    const float bq = quantize_to_u8(cbrt_b, B_MIN, B_MAX);
    // xyb_to_linear_rgb, xyb.rs:133-164
    float cr = yq + xq, cg = yq - xq, cb = bq;
    cr = cr - NEG_BIAS_CBRT;
    cg = cg - NEG_BIAS_CBRT;
    cb = cb - NEG_BIAS_CBRT;
    const float o_r = mixed_cube(cr) - BIAS;
    const float o_g = mixed_cube(cg) - BIAS;
    const float o_b = mixed_cube(cb) - BIAS;
    const float lr = I00 * o_r + I01 * o_g + I02 * o_b;
    const float lg = I10 * o_r + I11 * o_g + I12 * o_b;
    const float lb = I20 * o_r + I21 * o_g + I22 * o_b;
    ro = linear_to_srgb_u8(lr, thresh);
    go = linear_to_srgb_u8(lg, thresh);
    bo = linear_to_srgb_u8(lb, thresh);
}

// four pixels (12 bytes = three aligned dwords) per thread
__global__ __launch_bounds__(kThreads) void k_xyb_roundtrip(const uint8_t *__restrict__ in, uint8_t *__restrict__ out,
                                                            const float *__restrict__ lut_g,
                                                            const float *__restrict__ thresh_g, size_t n_pixels)
{
    __shared__ float lut[256];
    __shared__ float thresh[256];
    lut[threadIdx.x] = lut_g[threadIdx.x];
    thresh[threadIdx.x] = thresh_g[threadIdx.x];
    __syncthreads();
    const size_t n_quads = n_pixels / 4;
    const size_t stride = (size_t)gridDim.x * kThreads;
    const uint32_t *in32 = reinterpret_cast<const uint32_t *>(in);
    uint32_t *out32 = reinterpret_cast<uint32_t *>(out);
    for (size_t q = (size_t)blockIdx.x * kThreads + threadIdx.x; q < n_quads; q += stride) {
        const uint32_t w0 = in32[3 * q], w1 = in32[3 * q + 1], w2 = in32[3 * q + 2];
        uint32_t px[12] = {w0 & 255, (w0 >> 8) & 255, (w0 >> 16) & 255, w0 >> 24,
                           w1 & 255, (w1 >> 8) & 255, (w1 >> 16) & 255, w1 >> 24,
                           w2 & 255, (w2 >> 8) & 255, (w2 >> 16) & 255, w2 >> 24};
        uint32_t o[12];
#pragma unroll
        for (int k = 0; k < 4; k++)
            roundtrip_pixel(px[3 * k], px[3 * k + 1], px[3 * k + 2], lut, thresh, o[3 * k], o[3 * k + 1], o[3 * k + 2]);
        out32[3 * q] = o[0] | (o[1] << 8) | (o[2] << 16) | (o[3] << 24);
        out32[3 * q + 1] = o[4] | (o[5] << 8) | (o[6] << 16) | (o[7] << 24);
        out32[3 * q + 2] = o[8] | (o[9] << 8) | (o[10] << 16) | (o[11] << 24);
    }
    // tail pixels (n_pixels % 4), one thread each
    const size_t tail0 = n_quads * 4;
    if (blockIdx.x == 0 && threadIdx.x < n_pixels - tail0) {
        const size_t i = tail0 + threadIdx.x;
        uint32_t ro, go, bo;
        roundtrip_pixel(in[3 * i], in[3 * i + 1], in[3 * i + 2], lut, thresh, ro, go, bo);
        out[3 * i] = (uint8_t)ro;
        out[3 * i + 1] = (uint8_t)go;
        out[3 * i + 2] = (uint8_t)bo;
    }
}

}  // namespace

int ce_launch_xyb_roundtrip(ce_ctx *ctx, const uint8_t *d_in, uint8_t *d_out, size_t n_pixels)
{
    if (n_pixels == 0) return CE_OK;
    size_t quads = n_pixels / 4;
    size_t blocks = (quads + kThreads - 1) / kThreads;
    if (blocks > 256 * 16) blocks = 256 * 16;  // grid-stride beyond 16 blocks per CU
    if (blocks == 0) blocks = 1;
    CE_LAUNCH(ctx, "xyb_roundtrip", k_xyb_roundtrip, dim3((uint32_t)blocks), dim3(kThreads), 0, d_in, d_out,
              ctx->d_lut_powf, ctx->d_xyb_thresh, n_pixels);
    CE_HIP(ctx, hipGetLastError());
    return CE_OK;
}
