// DSSIM on gfx950 — replaces rgb8_to_dssim_image + calculate_dssim
// (/root/reference/src/metrics/dssim.rs:102-114, 40-71 -> dssim_core::Dssim::{create_image, compare}).
//
// Per level (5 levels, each the 2x2 average of the previous in LINEAR RGB, floor sizes):
//   per image slot ("create_image", references once per reference, not once per pair):
//     linear RGB -> normalised L*a*b*  ->  chroma planes pre-blurred  ->  mu = blur(img), sq = blur(img*img)
//   per pair ("compare"):
//     i12 = blur(img1*img2) -> SSIM map over the channel-averaged statistics -> mean -> mean |avg - ssim|
// "blur" is the fixed 3x3 kernel applied twice with edge replication.  Every plane op keeps the
// oracle's f32 operation order (oracle/dssim.c), so planes are bit-identical; sums are f64.
// Build with -ffp-contract=off.
#include <algorithm>

#include "ce_internal.h"

namespace {

constexpr int TPB = 256;

__device__ __forceinline__ uint32_t slot_of(uint32_t z, uint32_t n_refs_used, uint32_t max_refs)
{
    return z < n_refs_used ? z : max_refs + (z - n_refs_used);
}

// ---- level 0: sRGB u8 -> linear f32 planes through the host-powf table (dssim.rs:78-85) ----------
__global__ __launch_bounds__(TPB) void k_dssim_linear_u8(const uint8_t *__restrict__ refs, const uint8_t *__restrict__ tests,
                                                         const float *__restrict__ lut, float *__restrict__ lin, uint32_t w,
                                                         uint32_t h, uint32_t pitch, size_t plane, size_t img_bytes,
                                                         uint32_t n_refs_used, uint32_t max_refs)
{
    __shared__ float s_lut[256];
    s_lut[threadIdx.x] = lut[threadIdx.x];
    __syncthreads();
    const uint32_t z = blockIdx.z, slot = slot_of(z, n_refs_used, max_refs);
    const uint8_t *src = z < n_refs_used ? refs + (size_t)z * img_bytes : tests + (size_t)(z - n_refs_used) * img_bytes;
    const uint32_t x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= w || y >= h) return;
    const uint8_t *px = src + ((size_t)y * w + x) * 3;
    float *dst = lin + (size_t)slot * 3 * plane + (size_t)y * pitch + x;
    dst[0] = s_lut[px[0]];
    dst[plane] = s_lut[px[1]];
    dst[2 * plane] = s_lut[px[2]];
}

// ---- rgb8_to_dssim_image (dssim.rs:102-114): interleaved RGBA f32, a = 1.0 -------------------------
__global__ __launch_bounds__(TPB) void k_rgb8_to_rgba_f32(const uint8_t *__restrict__ rgb, const float *__restrict__ lut,
                                                          float4 *__restrict__ out, size_t n)
{
    __shared__ float s_lut[256];
    s_lut[threadIdx.x] = lut[threadIdx.x];
    __syncthreads();
    for (size_t i = (size_t)blockIdx.x * TPB + threadIdx.x; i < n; i += (size_t)gridDim.x * TPB)
        out[i] = make_float4(s_lut[rgb[3 * i]], s_lut[rgb[3 * i + 1]], s_lut[rgb[3 * i + 2]], 1.0f);
}

// ---- Downsample: (a + b + c + d) * 0.25, floor sizes, odd last row/column dropped --------------------
__global__ __launch_bounds__(TPB) void k_dssim_downsample(const float *__restrict__ in, float *__restrict__ out, uint32_t ipitch,
                                                          size_t iplane, uint32_t ow, uint32_t oh, uint32_t opitch,
                                                          size_t oplane, uint32_t n_refs_used, uint32_t max_refs)
{
    const uint32_t slot = slot_of(blockIdx.z / 3, n_refs_used, max_refs), c = blockIdx.z % 3;
    const uint32_t ox = blockIdx.x * 64 + (threadIdx.x & 63), oy = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (ox >= ow || oy >= oh) return;
    const float *p = in + ((size_t)slot * 3 + c) * iplane;
    const float a = p[(size_t)(2 * oy) * ipitch + 2 * ox], b = p[(size_t)(2 * oy) * ipitch + 2 * ox + 1];
    const float cc = p[(size_t)(2 * oy + 1) * ipitch + 2 * ox], d = p[(size_t)(2 * oy + 1) * ipitch + 2 * ox + 1];
    out[((size_t)slot * 3 + c) * oplane + (size_t)oy * opitch + ox] = (a + b + cc + d) * 0.25f;
}

// ---- linear RGB -> normalised (L, a, b) ------------------------------------------------------------
__device__ __forceinline__ float cbrt_poly(float x)
{
    float y = (-0.5f * x + 1.51f) * x + 0.2f;
    float y3 = y * y * y;
    y = y * (y3 + 2.0f * x) / (2.0f * y3 + x);
    y3 = y * y * y;
    y = y * (y3 + 2.0f * x) / (2.0f * y3 + x);
    return y;
}

__global__ __launch_bounds__(TPB) void k_dssim_lab(const float *__restrict__ lin, float *__restrict__ lab, uint32_t w, uint32_t h,
                                                   uint32_t pitch, size_t plane, uint32_t n_refs_used, uint32_t max_refs)
{
    const uint32_t slot = slot_of(blockIdx.z, n_refs_used, max_refs);
    const uint32_t x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= w || y >= h) return;
    const size_t o = (size_t)slot * 3 * plane + (size_t)y * pitch + x;
    const float r = lin[o], g = lin[o + plane], b = lin[o + 2 * plane];
    const float D65X = 0.9505f, D65Y = 1.0f, D65Z = 1.089f;
    const float EPS = 216.0f / 24389.0f, K = 24389.0f / (27.0f * 116.0f);
    const float fx = __builtin_fmaf(b, 0.1805f / D65X, __builtin_fmaf(g, 0.3576f / D65X, r * (0.4124f / D65X)));
    const float fy = __builtin_fmaf(b, 0.0722f / D65Y, __builtin_fmaf(g, 0.7152f / D65Y, r * (0.2126f / D65Y)));
    const float fz = __builtin_fmaf(b, 0.9505f / D65Z, __builtin_fmaf(g, 0.1192f / D65Z, r * (0.0193f / D65Z)));
    const float X = fx > EPS ? cbrt_poly(fx) - 16.0f / 116.0f : K * fx;
    const float Y = fy > EPS ? cbrt_poly(fy) - 16.0f / 116.0f : K * fy;
    const float Z = fz > EPS ? cbrt_poly(fz) - 16.0f / 116.0f : K * fz;
    lab[o] = Y * 1.05f;
    lab[o + plane] = __builtin_fmaf(500.0f / 220.0f, X - Y, 86.2f / 220.0f);
    lab[o + 2 * plane] = __builtin_fmaf(200.0f / 220.0f, Y - Z, 107.9f / 220.0f);
}

// ---- one 3x3 pass with edge replication over n planes ------------------------------------------------
// MODE 0: in ; MODE 1: in*in ; MODE 2: in*in2 (products formed per tap, i.e. "blur of the product image").
// planes are addressed as base + (zslot * planes_per_slot + first_plane + k) * plane
template <int MODE>
__global__ __launch_bounds__(TPB) void k_blur3x3(const float *__restrict__ in, const float *__restrict__ in2,
                                                 float *__restrict__ out, uint32_t w, uint32_t h, uint32_t pitch, size_t plane,
                                                 uint32_t n_planes, uint32_t in_pps, uint32_t in_first, uint32_t out_pps,
                                                 uint32_t out_first, const uint32_t *__restrict__ in_slot_map,
                                                 const uint32_t *__restrict__ in2_slot_map, uint32_t in2_slot_base,
                                                 uint32_t n_refs_used, uint32_t max_refs, int out_by_z)
{
    const uint32_t z = blockIdx.z / n_planes, k = blockIdx.z % n_planes;
    const uint32_t x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= w || y >= h) return;
    // slot selection: image-slot passes enumerate used slots; pair passes map pair -> (ref slot, test slot)
    uint32_t s_in, s_in2 = 0, s_out;
    if (in_slot_map) {
        s_in = in_slot_map[z];
        s_in2 = in2_slot_base + z;
        s_out = z;
    } else {
        s_in = out_by_z ? z : slot_of(z, n_refs_used, max_refs);
        s_out = s_in;
    }
    (void)in2_slot_map;
    const float *p = in + ((size_t)s_in * in_pps + in_first + k) * plane;
    const float *q = MODE == 2 ? in2 + ((size_t)s_in2 * in_pps + in_first + k) * plane : nullptr;
    const uint32_t y0 = y > 0 ? y - 1 : 0, y2 = y + 1 < h ? y + 1 : y;
    const uint32_t c0 = x > 0 ? x - 1 : 0, c2 = x + 1 < w ? x + 1 : w - 1;
    auto at = [&](uint32_t yy, uint32_t xx) {
        const float v = p[(size_t)yy * pitch + xx];
        if (MODE == 0) return v;
        if (MODE == 1) return v * v;
        return v * q[(size_t)yy * pitch + xx];
    };
    const float K0 = 0.095332f, K1 = 0.118095f, K4 = 0.146293f;
    const float r = (at(y0, c0) + at(y0, c2) + at(y2, c0) + at(y2, c2)) * K0 +
                    (at(y0, x) + at(y, c0) + at(y, c2) + at(y2, x)) * K1 + at(y, x) * K4;
    out[((size_t)s_out * out_pps + out_first + k) * plane + (size_t)y * pitch + x] = r;
}

__device__ __forceinline__ double block_sum(double v, double *s_red)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = v;
    __syncthreads();
    double t = 0.0;
    if (threadIdx.x == 0)
        for (int k = 0; k < TPB / 64; k++) t += s_red[k];
    __syncthreads();
    return t;
}

// ---- compare_scale: SSIM over the channel-averaged statistics; writes the map and the block's sum ----
__global__ __launch_bounds__(TPB) void k_dssim_ssim_map(const float *__restrict__ mu, const float *__restrict__ sq,
                                                        const float *__restrict__ i12, const uint32_t *__restrict__ pair_ref,
                                                        float *__restrict__ map, double *__restrict__ part, uint32_t w, uint32_t h,
                                                        uint32_t pitch, size_t plane, uint32_t max_refs, uint32_t level,
                                                        uint32_t n_levels, uint32_t n_blocks)
{
    __shared__ double s_red[TPB / 64];
    const uint32_t p = blockIdx.z;
    const uint32_t x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
    double val = 0.0;
    if (x < w && y < h) {
        const size_t o = (size_t)y * pitch + x;
        const size_t oa = (size_t)pair_ref[p] * 3 * plane + o, ob = (size_t)(max_refs + p) * 3 * plane + o;
        const float c1 = 0.01f * 0.01f, c2 = 0.03f * 0.03f, third = 1.0f / 3.0f;
        float m11[3], m12[3], m22[3], s1[3], s2[3], s12[3];
#pragma unroll
        for (int c = 0; c < 3; c++) {
            const float u1 = mu[oa + c * plane], u2 = mu[ob + c * plane];
            m11[c] = u1 * u1;
            m12[c] = u1 * u2;
            m22[c] = u2 * u2;
            s1[c] = sq[oa + c * plane] - m11[c];
            s2[c] = sq[ob + c * plane] - m22[c];
            s12[c] = i12[((size_t)p * 3 + c) * plane + o] - m12[c];
        }
#define AVG3(v) (((v)[0] + (v)[1] + (v)[2]) * third)
        const float mu1_sq = AVG3(m11), mu2_sq = AVG3(m22), mu1_mu2 = AVG3(m12);
        const float sigma1_sq = AVG3(s1), sigma2_sq = AVG3(s2), sigma12 = AVG3(s12);
#undef AVG3
        const float ssim = (2.0f * mu1_mu2 + c1) * (2.0f * sigma12 + c2) / ((mu1_sq + mu2_sq + c1) * (sigma1_sq + sigma2_sq + c2));
        map[(size_t)p * plane + o] = ssim;
        val = (double)ssim;
    }
    const double t = block_sum(val, s_red);
    if (threadIdx.x == 0)
        part[(((size_t)p * n_levels + level) * 2 + 0) * n_blocks + blockIdx.y * gridDim.x + blockIdx.x] = t;
}

// ---- avg = max(mean, 0)^(0.5^level): one block per pair reduces the SSIM partial sums in a fixed order ----
__global__ __launch_bounds__(TPB) void k_dssim_avg(const double *__restrict__ part, double *__restrict__ avg_out, uint32_t w,
                                                   uint32_t h, uint32_t level, uint32_t n_levels, uint32_t n_blocks,
                                                   uint32_t used_blocks)
{
    __shared__ double s_red[TPB];
    const uint32_t p = blockIdx.x;
    const double *pp = part + ((size_t)p * n_levels + level) * 2 * n_blocks;
    double v = 0.0;
    for (uint32_t k = threadIdx.x; k < used_blocks; k += TPB) v += pp[k];
    s_red[threadIdx.x] = v;
    __syncthreads();
    for (int off = TPB / 2; off > 0; off >>= 1) {
        if ((int)threadIdx.x < off) s_red[threadIdx.x] += s_red[threadIdx.x + off];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        double avg = s_red[0] / (double)((size_t)w * h);
        if (!(avg > 0.0)) avg = 0.0;
        avg_out[(size_t)p * n_levels + level] = pow(avg, pow(0.5, (double)level));
    }
}

// ---- mean absolute deviation of the SSIM map from avg ------------------------------------------------------
__global__ __launch_bounds__(TPB) void k_dssim_absdev(const float *__restrict__ map, const double *__restrict__ avg_in,
                                                      double *__restrict__ part, uint32_t w, uint32_t h, uint32_t pitch,
                                                      size_t plane, uint32_t level, uint32_t n_levels, uint32_t n_blocks)
{
    __shared__ double s_red[TPB / 64];
    const uint32_t p = blockIdx.z;
    double *pp = part + ((size_t)p * n_levels + level) * 2 * n_blocks;
    const double avg = avg_in[(size_t)p * n_levels + level];
    const uint32_t x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
    double val = 0.0;
    if (x < w && y < h) val = fabs(avg - (double)map[(size_t)p * plane + (size_t)y * pitch + x]);
    const double t = block_sum(val, s_red);
    if (threadIdx.x == 0) pp[n_blocks + blockIdx.y * gridDim.x + blockIdx.x] = t;
}

struct ds_geom {
    uint32_t npix[CE_DSSIM_SCALES];
    uint32_t nblk[CE_DSSIM_SCALES];
};

__global__ void k_dssim_finalize_pairs(const double *__restrict__ part, double *__restrict__ level_scores,
                                       ce_dev_scores *__restrict__ scores, uint32_t n_pairs, uint32_t n_levels,
                                       uint32_t n_blocks, ds_geom g)
{
    const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n_pairs) return;
    const double W[CE_DSSIM_SCALES] = {0.028, 0.197, 0.322, 0.298, 0.155};
    double ssim_sum = 0.0, weight_sum = 0.0;
    for (uint32_t l = 0; l < n_levels; l++) {
        const double *pp = part + (((size_t)p * n_levels + l) * 2 + 1) * n_blocks;
        double dev = 0.0;
        for (uint32_t k = 0; k < g.nblk[l]; k++) dev += pp[k];
        const double score = 1.0 - dev / (double)g.npix[l];
        level_scores[(size_t)p * CE_DSSIM_SCALES + l] = score;
        ssim_sum += score * W[l];
        weight_sum += W[l];
    }
    double ssim = ssim_sum / weight_sum;
    if (!(ssim > 2.220446049250313e-16)) ssim = 2.220446049250313e-16;
    scores[p].dssim = 1.0 / ssim - 1.0;
}

}  // namespace

void ce_dssim_free(ce_batch *b)
{
    for (auto &p : b->ds_lin) hipFree(p), p = nullptr;
    for (auto &p : b->ds_tmp) hipFree(p), p = nullptr;
    hipFree(b->ds_img); hipFree(b->ds_mu); hipFree(b->ds_sq); hipFree(b->ds_i12); hipFree(b->ds_map);
    hipFree(b->ds_part); hipFree(b->ds_level_scores);
    b->ds_img = b->ds_mu = b->ds_sq = b->ds_i12 = b->ds_map = nullptr;
    b->ds_part = b->ds_level_scores = nullptr;
    b->dssim_ready = false;
}

static int dssim_prepare(ce_batch *b)
{
    if (b->dssim_ready) return CE_OK;
    ce_ctx *ctx = b->ctx;
    uint32_t w = b->w, h = b->h;
    int n = 0;
    // make_scales_recursive: a level is halved only while it is at least 8x8
    for (int l = 0; l < CE_DSSIM_SCALES; l++) {
        auto &d = b->ds[l];
        d.w = w;
        d.h = h;
        d.pitch = (w + 31u) & ~31u;
        d.plane = (size_t)d.pitch * h;
        n++;
        if (w < 8 || h < 8) break;
        w /= 2;
        h /= 2;
        if (w == 0 || h == 0) break;
    }
    b->ds_levels = n;
    const size_t slots = (size_t)b->max_refs + b->max_pairs, p0 = b->ds[0].plane;
    CE_HIP(ctx, hipMalloc(&b->ds_lin[0], slots * 3 * p0 * sizeof(float)));
    CE_HIP(ctx, hipMalloc(&b->ds_lin[1], slots * 3 * p0 * sizeof(float)));
    CE_HIP(ctx, hipMalloc(&b->ds_img, slots * 3 * p0 * sizeof(float)));
    CE_HIP(ctx, hipMalloc(&b->ds_mu, slots * 3 * p0 * sizeof(float)));
    CE_HIP(ctx, hipMalloc(&b->ds_sq, slots * 3 * p0 * sizeof(float)));
    CE_HIP(ctx, hipMalloc(&b->ds_tmp[0], slots * 3 * p0 * sizeof(float)));
    CE_HIP(ctx, hipMalloc(&b->ds_tmp[1], slots * 3 * p0 * sizeof(float)));
    CE_HIP(ctx, hipMalloc(&b->ds_i12, (size_t)b->max_pairs * 3 * p0 * sizeof(float)));
    CE_HIP(ctx, hipMalloc(&b->ds_map, (size_t)b->max_pairs * p0 * sizeof(float)));
    b->ds_blocks = ((b->ds[0].w + 63) / 64) * ((b->ds[0].h + 3) / 4);
    CE_HIP(ctx, hipMalloc(&b->ds_part, (size_t)b->max_pairs * CE_DSSIM_SCALES * 2 * b->ds_blocks * sizeof(double)));
    CE_HIP(ctx, hipMalloc(&b->ds_level_scores, (size_t)b->max_pairs * CE_DSSIM_SCALES * sizeof(double)));  // avg, then score
    b->dssim_ready = true;
    return CE_OK;
}

int ce_launch_dssim(ce_batch *b, const uint8_t *d_refs, uint32_t n_refs_used, uint32_t n_pairs)
{
    ce_ctx *ctx = b->ctx;
    int rc = dssim_prepare(b);
    if (rc != CE_OK) return rc;
    const uint32_t n_slots = n_refs_used + n_pairs, mr = b->max_refs;
    ds_geom g{};
    for (int l = 0; l < b->ds_levels; l++) {
        const auto &d = b->ds[l];
        float *lin = b->ds_lin[l & 1];
        const dim3 grid((d.w + 63) / 64, (d.h + 3) / 4, n_slots);
        if (l == 0) {
            CE_LAUNCH(ctx, "dssim_linear_u8", k_dssim_linear_u8, grid, dim3(TPB), 0, d_refs, b->d_tests, ctx->d_lut_powf, lin,
                      d.w, d.h, d.pitch, d.plane, b->img_bytes, n_refs_used, mr);
        } else {
            const auto &pd = b->ds[l - 1];
            CE_LAUNCH(ctx, "dssim_downsample", k_dssim_downsample, dim3(grid.x, grid.y, n_slots * 3), dim3(TPB), 0,
                      b->ds_lin[(l - 1) & 1], lin, pd.pitch, pd.plane, d.w, d.h, d.pitch, d.plane, n_refs_used, mr);
        }
        // create_image: LAB, chroma pre-blur, mu, sq  (planes-per-slot = 3 everywhere)
        CE_LAUNCH(ctx, "dssim_lab", k_dssim_lab, grid, dim3(TPB), 0, lin, b->ds_img, d.w, d.h, d.pitch, d.plane, n_refs_used, mr);
        const dim3 g2(grid.x, grid.y, n_slots * 2), g3(grid.x, grid.y, n_slots * 3);
        const uint32_t *nomap = nullptr;
        // chroma planes 1,2: img -> tmp0 -> img
        CE_LAUNCH(ctx, "dssim_blur3x3", k_blur3x3<0>, g2, dim3(TPB), 0, b->ds_img, (const float *)nullptr, b->ds_tmp[0], d.w, d.h,
                  d.pitch, d.plane, 2u, 3u, 1u, 3u, 1u, nomap, nomap, 0u, n_refs_used, mr, 0);
        CE_LAUNCH(ctx, "dssim_blur3x3", k_blur3x3<0>, g2, dim3(TPB), 0, b->ds_tmp[0], (const float *)nullptr, b->ds_img, d.w, d.h,
                  d.pitch, d.plane, 2u, 3u, 1u, 3u, 1u, nomap, nomap, 0u, n_refs_used, mr, 0);
        // mu = blur(blur(img))
        CE_LAUNCH(ctx, "dssim_blur3x3", k_blur3x3<0>, g3, dim3(TPB), 0, b->ds_img, (const float *)nullptr, b->ds_tmp[0], d.w, d.h,
                  d.pitch, d.plane, 3u, 3u, 0u, 3u, 0u, nomap, nomap, 0u, n_refs_used, mr, 0);
        CE_LAUNCH(ctx, "dssim_blur3x3", k_blur3x3<0>, g3, dim3(TPB), 0, b->ds_tmp[0], (const float *)nullptr, b->ds_mu, d.w, d.h,
                  d.pitch, d.plane, 3u, 3u, 0u, 3u, 0u, nomap, nomap, 0u, n_refs_used, mr, 0);
        // sq = blur(blur(img*img))
        CE_LAUNCH(ctx, "dssim_blur3x3_sq", k_blur3x3<1>, g3, dim3(TPB), 0, b->ds_img, (const float *)nullptr, b->ds_tmp[1], d.w,
                  d.h, d.pitch, d.plane, 3u, 3u, 0u, 3u, 0u, nomap, nomap, 0u, n_refs_used, mr, 0);
        CE_LAUNCH(ctx, "dssim_blur3x3", k_blur3x3<0>, g3, dim3(TPB), 0, b->ds_tmp[1], (const float *)nullptr, b->ds_sq, d.w, d.h,
                  d.pitch, d.plane, 3u, 3u, 0u, 3u, 0u, nomap, nomap, 0u, n_refs_used, mr, 0);
        // compare: i12 = blur(blur(img1*img2)) per pair; pass 1 -> tmp0 (indexed by pair), pass 2 -> i12
        const dim3 gp3(grid.x, grid.y, n_pairs * 3), gp(grid.x, grid.y, n_pairs);
        CE_LAUNCH(ctx, "dssim_blur3x3_mul", k_blur3x3<2>, gp3, dim3(TPB), 0, b->ds_img, b->ds_img, b->ds_tmp[0], d.w, d.h, d.pitch,
                  d.plane, 3u, 3u, 0u, 3u, 0u, (const uint32_t *)b->d_pair_ref, nomap, mr, n_refs_used, mr, 1);
        CE_LAUNCH(ctx, "dssim_blur3x3", k_blur3x3<0>, gp3, dim3(TPB), 0, b->ds_tmp[0], (const float *)nullptr, b->ds_i12, d.w, d.h,
                  d.pitch, d.plane, 3u, 3u, 0u, 3u, 0u, nomap, nomap, 0u, n_refs_used, mr, 1);
        const uint32_t used_blocks = grid.x * grid.y;
        CE_LAUNCH(ctx, "dssim_ssim_map", k_dssim_ssim_map, gp, dim3(TPB), 0, b->ds_mu, b->ds_sq, b->ds_i12, b->d_pair_ref, b->ds_map,
                  b->ds_part, d.w, d.h, d.pitch, d.plane, mr, (uint32_t)l, (uint32_t)b->ds_levels, b->ds_blocks);
        CE_LAUNCH(ctx, "dssim_avg", k_dssim_avg, dim3(n_pairs), dim3(TPB), 0, b->ds_part, b->ds_level_scores, d.w, d.h, (uint32_t)l,
                  (uint32_t)b->ds_levels, b->ds_blocks, used_blocks);
        CE_LAUNCH(ctx, "dssim_absdev", k_dssim_absdev, gp, dim3(TPB), 0, b->ds_map, b->ds_level_scores, b->ds_part, d.w, d.h,
                  d.pitch, d.plane, (uint32_t)l, (uint32_t)b->ds_levels, b->ds_blocks);
        g.npix[l] = d.w * d.h;
        g.nblk[l] = used_blocks;
    }
    CE_LAUNCH(ctx, "dssim_finalize", k_dssim_finalize_pairs, dim3((n_pairs + 63) / 64), dim3(64), 0, b->ds_part,
              b->ds_level_scores, b->d_scores, n_pairs, (uint32_t)b->ds_levels, b->ds_blocks, g);
    CE_HIP(ctx, hipGetLastError());
    return CE_OK;
}

int ce_launch_rgb8_to_dssim_image(ce_ctx *ctx, const uint8_t *d_rgb, float *d_rgba, size_t n_pixels)
{
    if (n_pixels == 0) return CE_OK;
    size_t blocks = std::min<size_t>((n_pixels + TPB - 1) / TPB, 4096);
    CE_LAUNCH(ctx, "rgb8_to_dssim_image", k_rgb8_to_rgba_f32, dim3((uint32_t)blocks), dim3(TPB), 0, d_rgb, ctx->d_lut_powf,
              reinterpret_cast<float4 *>(d_rgba), n_pixels);
    CE_HIP(ctx, hipGetLastError());
    return CE_OK;
}
