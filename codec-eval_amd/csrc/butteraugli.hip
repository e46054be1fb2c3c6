// Butteraugli on gfx950 — replaces butteraugli::butteraugli(..).score behind
// /root/reference/src/metrics/butteraugli.rs:72-80,127-135 (and compute_butteraugli(..).score in the
// codec-compare bins).  Restates libjxl's butteraugli.cc pipeline stage by stage, in the same f32
// operation order as the CPU restatement (oracle/butteraugli.c):
//
//   per image slot and resolution level (2x-subsampled first, then full) — the "PsychoImage", built once per
//   reference, not once per pair:
//     sRGB u8 -> linear (the half-resolution level: 2x2 average, formed in the same kernel) -> blur sigma 1.2 (5-tap,
//     mirrored) -> OpsinDynamicsImage -> XYB                                                          [k_ba_front]
//     -> LF = blur 7.156 ; MF = blur 3.225 of (XYB - LF) ; HF = blur 1.564 of the rest ; UHF = rest, with the range
//        shaping of SeparateFrequencies -> 10 planes; the mask input (DiffPrecompute of HF + UHF) falls out of the HF
//        stage and is blurred with sigma 2.7                                           [k_ba_blur_h<33>, k_ba_blur_v_split]
//   per reference and level: the mask values (fuzzy erosion + the two mask curves)                  [k_ba_mask_vals]
//   per pair and level, ONE kernel:                                                                [k_ba_malta_l2_xy]
//     Malta line filters (UHF, HF, MF; X and Y), asymmetric L2 (HF), L2 (MF), L2 (LF), CombineChannelsToDiffmap; the
//     full-resolution level then forms diffmap = 0.85 * full + 0.5 * upsampled(half) in registers and reduces it
//   score = max ; p-norm = mean of 3-, 6-, 12-norms.                                                    [k_ba_score]
//
// The 33-tap blur is a row kernel + a column kernel; the 15-, 13- and 7-tap blurs run row and column pass in one kernel
// (LDS-tiled, borders re-normalised).  Build with -ffp-contract=off.
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <type_traits>

#include "ce_internal.h"

namespace {

constexpr int TPB = 256;
constexpr int PSY = 10;  // uhf0 uhf1 hf0 hf1 mf0 mf1 mf2 lf0 lf1 lf2
enum { UHF0 = 0, UHF1, HF0, HF1, MF0, MF1, MF2, LF0, LF1, LF2 };

struct geom {
    uint32_t w, h, pitch;
    size_t plane;
};

struct blur_kernel {
    int len;
    float k[40];
    // 1 / (sum of the valid weights) for an output d pixels from the low / high border of a line that is at least
    // `len` long, summed on the host in the order the pixel loop would sum them (low border: taps off-d .. len-1,
    // high border: taps 0 .. off+d)
    float lo[16], hi[16];
};

#define BA_XY                                                           \
    const uint32_t x = blockIdx.x * 64 + (threadIdx.x & 63);            \
    const uint32_t y = blockIdx.y * 4 + (threadIdx.x >> 6);             \
    if (x >= g.w || y >= g.h) return;                                   \
    const size_t o = (size_t)y * g.pitch + x

__device__ __forceinline__ uint32_t slot_of(uint32_t z, uint32_t n_refs_used, uint32_t max_refs)
{
    return z < n_refs_used ? z : max_refs + (z - n_refs_used);
}

// ---- separable blurs ------------------------------------------------------------------------------------
// unit stride `us` planes per unit; planes [first, first+n) of each unit are processed; z = unit * n + k
struct plane_sel {
    uint32_t per_unit, first, n;
};

__device__ __forceinline__ uint32_t mirror(int x, int n)
{
    while (x < 0 || x >= n) x = x < 0 ? -x - 1 : 2 * n - 1 - x;
    return (uint32_t)x;
}

// Long separable blurs, LDS-tiled with a register window: a thread produces 8 consecutive outputs
// from 8 + LEN - 1 inputs held in registers (5 LDS reads per output at LEN = 33 instead of 33), each
// output summing its taps in ascending order exactly like ConvolutionWithTranspose.  Taps outside the
// image are zero in the tile (x + 0.0f == x, so the sum over the valid taps is unchanged) and border
// outputs are scaled by 1 / (sum of their valid weights).
constexpr int BW_OUT = 8;

// A tap of the long blurs: sum + v * k with the lineage's two roundings (bit-identical planes).  -DCE_BA_BLUR_FMA=1 builds
// the taps as ONE fused multiply-add - the oracle's switch "ba_blur_fma" bit for bit - for A/B runs: measured in round 3
// at +3 % of the headline (the five blur kernels 1.86 -> 1.70 ms per Kodak step) for a score that moves by up to 6e-5
// relative on small shapes (profiles/r03_experiments.md section 3): too close to the 1e-4 contract, not adopted.
#ifndef CE_BA_BLUR_FMA
#define CE_BA_BLUR_FMA 0
#endif
__device__ __forceinline__ float blur_tap(float sum, float v, float k)
{
#if CE_BA_BLUR_FMA
    return __builtin_fmaf(v, k, sum);
#else
    return sum + v * k;
#endif
}

template <int LEN>
__device__ __forceinline__ float border_scale(const blur_kernel &bk, int pos, int n, float inv_wsum)
{
    constexpr int off = LEN / 2;
    if (pos - off >= 0 && pos + off <= n - 1) return inv_wsum;
    if (n >= LEN) {
        // one border only: pick the host-built scale with compile-time indices (a dynamic index into the kernel
        // arguments would be a serialised global load per tap - it used to cost the x-border blocks ~100 us each)
        const int d = pos < off ? pos : n - 1 - pos;
        float s = 0.0f;
#pragma unroll
        for (int i = 0; i < off; i++) s = d == i ? (pos < off ? bk.lo[i] : bk.hi[i]) : s;
        return s;
    }
    float weight = 0.0f;
    for (int j = max(pos - off, 0); j <= min(pos + off, n - 1); j++) weight += bk.k[j - pos + off];
    return 1.0f / weight;
}

// horizontal: block = 8 rows x 256 columns.  The tile starts 16 columns left of the block whatever LEN is, so rows
// load as aligned float4 (72 per row); LDS rows are padded one float per 8 so that lanes reading with a stride of
// 8 floats hit distinct banks.  A thread produces 8 consecutive outputs as 4 PAIRS: its 8+LEN-1 inputs are held
// as even-aligned pairs (v[2i], v[2i+1]) and odd-aligned pairs (v[2i+1], v[2i+2]), so every tap of an output pair
// is one v_pk_mul_f32 + one v_pk_add_f32 (two IEEE multiplies / adds: bit-identical to the scalar tap loop, taps
// still summed in ascending order).
typedef float ba_f2 __attribute__((ext_vector_type(2)));
#ifndef CE_BH_TILES
#define CE_BH_TILES 4
#endif
constexpr int BH_TILES = CE_BH_TILES;  // 8-row tiles per block of the row blur

template <int LEN>
__global__ __launch_bounds__(TPB) void k_ba_blur_h(const float *__restrict__ in, float *__restrict__ out, geom g, plane_sel si,
                                                   plane_sel so, blur_kernel bk, float inv_wsum, uint32_t n_refs_used,
                                                   uint32_t max_refs, int by_slot, uint32_t z0)
{
    constexpr int off = LEN / 2, TW = 256, TR = 8, LEFT = 16, RAW = TW + 2 * LEFT, ROWF = RAW + RAW / 8 + 1, SH = LEFT - off;
    static_assert(off <= LEFT, "tile halo");
    __shared__ float tile[TR * ROWF];
    const uint32_t u = blockIdx.z / si.n + z0, k = blockIdx.z % si.n;
    const uint32_t unit = by_slot ? slot_of(u, n_refs_used, max_refs) : u;
    const float *p = in + ((size_t)unit * si.per_unit + si.first + k) * g.plane;
    const int x0 = blockIdx.x * TW;
    // A block walks BH_TILES consecutive 8-row tiles; the next tile's float4 pieces are fetched into registers while
    // the current one is filtered out of LDS.
    constexpr int NF = (TR * (RAW / 4) + TPB - 1) / TPB;
    float4 pf[NF];
    auto fetch = [&](int y0) {
#pragma unroll
        for (int m = 0; m < NF; m++) {
            const int i = m * TPB + (int)threadIdx.x, r = i / (RAW / 4), c = 4 * (i % (RAW / 4)), gx = x0 - LEFT + c, gy = y0 + r;
            float4 v = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
            if (r < TR && gx >= 0 && gx < (int)g.pitch && gy < (int)g.h) {  // 16-byte aligned, whole float4 inside the padded row
                v = *reinterpret_cast<const float4 *>(p + (size_t)gy * g.pitch + gx);
                if (gx + 3 >= (int)g.w) {  // the row's padding is not part of the image
                    if (gx + 0 >= (int)g.w) v.x = 0.0f;
                    if (gx + 1 >= (int)g.w) v.y = 0.0f;
                    if (gx + 2 >= (int)g.w) v.z = 0.0f;
                    v.w = 0.0f;
                }
            }
            pf[m] = v;
        }
    };
    const int r = threadIdx.x >> 5, cx = threadIdx.x & 31, gx0 = x0 + 8 * cx;
    // The border scale of a column is the same for every row: each thread forms ONE column's scale (the select chain of
    // border_scale is ~50 instructions) and the block shares the 256 of them through LDS, instead of every thread forming
    // the eight of its own outputs (~700 of the ~3 300 instructions a thread executed per block: profiles/r03_experiments.md 22).
    __shared__ __attribute__((aligned(16))) float s_scale[TW];
    {
        const int gx = x0 + (int)threadIdx.x;
        s_scale[threadIdx.x] = gx < (int)g.w ? border_scale<LEN>(bk, gx, (int)g.w, inv_wsum) : 0.0f;
    }
    constexpr int NV = BW_OUT + LEN - 1, NP = (NV + 1) / 2;
    const float *row = &tile[r * ROWF + 9 * cx];  // element j of the window sits at j + SH + ((j + SH) >> 3)
    auto at = [&](int j) { return row[(j + SH) + ((j + SH) >> 3)]; };
    const int y_first = blockIdx.y * (TR * BH_TILES);
    fetch(y_first);
#pragma unroll 1
    for (int t = 0; t < BH_TILES; t++) {
        const int y0 = y_first + t * TR;
        if (y0 >= (int)g.h) break;  // block-uniform
        if (t) __syncthreads();     // the previous tile has been read
#pragma unroll
        for (int m = 0; m < NF; m++) {
            const int i = m * TPB + (int)threadIdx.x, rr = i / (RAW / 4), c = 4 * (i % (RAW / 4));
            if (rr < TR) {
                float *tp = &tile[rr * ROWF + c + (c >> 3)];  // c is a multiple of 4: the four floats stay inside one group of 8
                tp[0] = pf[m].x, tp[1] = pf[m].y, tp[2] = pf[m].z, tp[3] = pf[m].w;
            }
        }
        if (t + 1 < BH_TILES && y0 + TR < (int)g.h) fetch(y0 + TR);
        __syncthreads();
        const int gy = y0 + r;
        if (gy >= (int)g.h || gx0 >= (int)g.w) continue;
        ba_f2 pe[NP], po[NP];
#pragma unroll
        for (int i = 0; i < NP; i++) {
            pe[i] = ba_f2{at(2 * i), 2 * i + 1 < NV ? at(2 * i + 1) : 0.0f};
            po[i] = ba_f2{2 * i + 1 < NV ? at(2 * i + 1) : 0.0f, 2 * i + 2 < NV ? at(2 * i + 2) : 0.0f};
        }
        float *dst = out + ((size_t)unit * so.per_unit + so.first + k) * g.plane + (size_t)gy * g.pitch + gx0;
        float res[BW_OUT];
#pragma unroll
        for (int op = 0; op < BW_OUT / 2; op++) {
            ba_f2 sum = {0.0f, 0.0f};
#pragma unroll
            for (int j = 0; j < LEN; j++) {
                const int idx = 2 * op + j;
                const ba_f2 src = (idx & 1) ? po[idx >> 1] : pe[idx >> 1];
#if CE_BA_BLUR_FMA
                sum = __builtin_elementwise_fma(src, ba_f2{bk.k[j], bk.k[j]}, sum);
#else
                const ba_f2 prod = src * ba_f2{bk.k[j], bk.k[j]};
                sum = sum + prod;
#endif
            }
            res[2 * op] = sum.x * s_scale[8 * cx + 2 * op];  // written before the tile's first barrier
            res[2 * op + 1] = sum.y * s_scale[8 * cx + 2 * op + 1];
        }
        if (gx0 + BW_OUT <= (int)g.w) {  // rows are 128-byte aligned and gx0 is a multiple of 8
            *reinterpret_cast<float4 *>(dst) = make_float4(res[0], res[1], res[2], res[3]);
            *reinterpret_cast<float4 *>(dst + 4) = make_float4(res[4], res[5], res[6], res[7]);
        } else {
#pragma unroll
            for (int o = 0; o < BW_OUT; o++)
                if (gx0 + o < (int)g.w) dst[o] = res[o];
        }
    }
}

// ---- OpsinDynamicsImage (pointwise part) ---------------------------------------------------------------
__device__ __forceinline__ void opsin_absorbance(float in0, float in1, float in2, float &o0, float &o1, float &o2)
{
    const float mixi0 = 0.29956550340058319f, mixi1 = 0.63373087833825936f, mixi2 = 0.077705617820981968f, mixi3 = 1.7557483643287353f;
    const float mixi4 = 0.22158691104574774f, mixi5 = 0.69391388044116142f, mixi6 = 0.0987313588422f, mixi7 = 1.7557483643287353f;
    const float mixi8 = 0.02f, mixi9 = 0.02f, mixi10 = 0.20480129041026129f, mixi11 = 12.226454707163354f;
    o0 = __builtin_fmaf(mixi0, in0, __builtin_fmaf(mixi1, in1, __builtin_fmaf(mixi2, in2, mixi3)));
    o1 = __builtin_fmaf(mixi4, in0, __builtin_fmaf(mixi5, in1, __builtin_fmaf(mixi6, in2, mixi7)));
    o2 = __builtin_fmaf(mixi8, in0, __builtin_fmaf(mixi9, in1, __builtin_fmaf(mixi10, in2, mixi11)));
}

// FastLog2f of the lineage (libjxl lib/jxl/base/fast_math-inl.h): mantissa reduced to [-1/3, 1/3] by integer
// arithmetic on the bits, (2,2) rational polynomial of log2(1 + t) in Horner form with fused multiply-adds, one
// IEEE division.  IEEE basic operations only: bit-identical to oracle/butteraugli.c (a libm log2f is not, and on a
// flat image one ulp of the logarithm moves the score by up to 4e-3 relative).
__device__ __forceinline__ float fast_log2f(float x)
{
    const float p0 = -1.8503833400518310E-06f, p1 = 1.4287160470083755E+00f, p2 = 7.4245873327820566E-01f;
    const float q0 = 9.9032814277590719E-01f, q1 = 1.0096718572241148E+00f, q2 = 1.7409343003366853E-01f;
    const int32_t xb = (int32_t)__float_as_uint(x);
    const int32_t es = (xb - 0x3f2aaaab) >> 23;  // arithmetic shift
    const float m = __uint_as_float((uint32_t)xb - ((uint32_t)es << 23));
    const float t = m - 1.0f;
    const float yp = __builtin_fmaf(__builtin_fmaf(p2, t, p1), t, p0);
    const float yq = __builtin_fmaf(__builtin_fmaf(q2, t, q1), t, q0);
    return ce_div_noscale(yp, yq) + (float)es;  // yq in [0.67, 1.35], |yp| < 0.6 (ce_internal.h: the IEEE quotient, 8 instructions)
}

__device__ __forceinline__ float gamma_f(float v)
{
    const float kRetMul = 19.245013259874995f * 0.693147180559945f, kRetAdd = -23.16046239805755f;
    if (v < 0.0f) v = 0.0f;
    const float biased = v + 9.9710635769299145f;
    return __builtin_fmaf(kRetMul, fast_log2f(biased), kRetAdd);
}

// Front end, fused: linear RGB tile (level 0: sRGB u8 through the table; level 1: the subsampled planes) with a
// 2-pixel halo in LDS -> the 5-tap sigma-1.2 blur (mirrored at the image border, both passes in LDS) ->
// OpsinDynamicsImage -> XYB planes.  Only XYB is written (12 B/px); nothing else of this stage touches HBM.
constexpr int FT = 32, FR = FT + 4;
// HALF: the half-resolution level - its linear RGB is the 2x2 average of the full-resolution image's (Subsample2x: the
// four quarter-weighted samples added in row-major order, the last odd row / column doubled), formed here from the u8
// source instead of in a pass of its own; gi = the full-resolution geometry.
template <bool HALF>
__global__ __launch_bounds__(TPB) void k_ba_front(const uint8_t *__restrict__ refs, const uint8_t *__restrict__ tests,
                                                  const float *__restrict__ lut, geom gi,
                                                  float *__restrict__ xyb, geom g, float w0, float w1, float w2,
                                                  float intensity_target, size_t img_bytes, uint32_t n_refs_used, uint32_t max_refs,
                                                  uint32_t z0)
{
    __shared__ float L[3][FR * FR];   // linear, region = tile + 2
    __shared__ float H[3][FR * FT];   // row-blurred: FR rows x FT columns
    __shared__ float s_lut[256];
    s_lut[threadIdx.x] = lut[threadIdx.x];
    const uint32_t z = blockIdx.z + z0, slot = slot_of(z, n_refs_used, max_refs);
    const int w = (int)g.w, h = (int)g.h, x0 = blockIdx.x * FT, y0 = blockIdx.y * FT, gx0 = x0 - 2, gy0 = y0 - 2;
    const uint8_t *src8 = z < n_refs_used ? refs + (size_t)z * img_bytes : tests + (size_t)(z - n_refs_used) * img_bytes;
    const int W8 = HALF ? (int)gi.w : w, H8 = HALF ? (int)gi.h : h;  // the u8 image
    __syncthreads();
    const bool interior = gx0 >= 0 && gy0 >= 0 && gx0 + FR <= w && gy0 + FR <= h;
    // four u8 pixels (12 bytes, any alignment) from four aligned dwords
    auto load12 = [&](const uint8_t *p, uint32_t &v0, uint32_t &v1, uint32_t &v2) {
        const uintptr_t a = reinterpret_cast<uintptr_t>(p);
        const uint32_t *q = reinterpret_cast<const uint32_t *>(a & ~(uintptr_t)3);
        const uint32_t d0 = q[0], d1 = q[1], d2 = q[2], d3 = q[3], sh = (uint32_t)(a & 3);
        v0 = __builtin_amdgcn_alignbyte(d1, d0, sh), v1 = __builtin_amdgcn_alignbyte(d2, d1, sh), v2 = __builtin_amdgcn_alignbyte(d3, d2, sh);
    };
    if (!HALF && interior) {
        // four pixels per task; nine tasks per row of the region
        for (int i = threadIdx.x; i < FR * (FR / 4); i += TPB) {
            const int ly = i / (FR / 4), lx = 4 * (i % (FR / 4));
            uint32_t v0, v1, v2;
            load12(src8 + ((size_t)(gy0 + ly) * w + gx0 + lx) * 3, v0, v1, v2);
            const int o = ly * FR + lx;
            L[0][o] = s_lut[v0 & 255u], L[1][o] = s_lut[(v0 >> 8) & 255u], L[2][o] = s_lut[(v0 >> 16) & 255u];
            L[0][o + 1] = s_lut[v0 >> 24], L[1][o + 1] = s_lut[v1 & 255u], L[2][o + 1] = s_lut[(v1 >> 8) & 255u];
            L[0][o + 2] = s_lut[(v1 >> 16) & 255u], L[1][o + 2] = s_lut[v1 >> 24], L[2][o + 2] = s_lut[v2 & 255u];
            L[0][o + 3] = s_lut[(v2 >> 8) & 255u], L[1][o + 3] = s_lut[(v2 >> 16) & 255u], L[2][o + 3] = s_lut[v2 >> 24];
        }
    } else if (HALF && interior && 2 * (gx0 + FR) <= W8 && 2 * (gy0 + FR) <= H8) {
        // two half-resolution elements (2 x 4 u8 pixels) per task: ((0 + q00) + q01) + q10) + q11, q = 0.25 * linear
        for (int i = threadIdx.x; i < FR * (FR / 2); i += TPB) {
            const int ly = i / (FR / 2), lx = 2 * (i % (FR / 2));
            uint32_t a0, a1, a2, b0, b1, b2;
            const uint8_t *p = src8 + ((size_t)(2 * (gy0 + ly)) * W8 + 2 * (gx0 + lx)) * 3;
            load12(p, a0, a1, a2);
            load12(p + (size_t)W8 * 3, b0, b1, b2);
            const int o = ly * FR + lx;
            // row pixels: p0 = (v0.b0, v0.b1, v0.b2), p1 = (v0.b3, v1.b0, v1.b1), p2 = (v1.b2, v1.b3, v2.b0), p3 = (v2.b1, v2.b2, v2.b3)
#define CE_Q(v) (0.25f * s_lut[(v)])
            L[0][o] = ((0.0f + CE_Q(a0 & 255u)) + CE_Q(a0 >> 24)) + CE_Q(b0 & 255u) + CE_Q(b0 >> 24);
            L[1][o] = ((0.0f + CE_Q((a0 >> 8) & 255u)) + CE_Q(a1 & 255u)) + CE_Q((b0 >> 8) & 255u) + CE_Q(b1 & 255u);
            L[2][o] = ((0.0f + CE_Q((a0 >> 16) & 255u)) + CE_Q((a1 >> 8) & 255u)) + CE_Q((b0 >> 16) & 255u) + CE_Q((b1 >> 8) & 255u);
            L[0][o + 1] = ((0.0f + CE_Q((a1 >> 16) & 255u)) + CE_Q((a2 >> 8) & 255u)) + CE_Q((b1 >> 16) & 255u) + CE_Q((b2 >> 8) & 255u);
            L[1][o + 1] = ((0.0f + CE_Q(a1 >> 24)) + CE_Q((a2 >> 16) & 255u)) + CE_Q(b1 >> 24) + CE_Q((b2 >> 16) & 255u);
            L[2][o + 1] = ((0.0f + CE_Q(a2 & 255u)) + CE_Q(a2 >> 24)) + CE_Q(b2 & 255u) + CE_Q(b2 >> 24);
#undef CE_Q
        }
    } else
    for (int i = threadIdx.x; i < FR * FR; i += TPB) {
        const int lx = i % FR, ly = i / FR, X = gx0 + lx, Y = gy0 + ly;
        if (X >= 0 && X < w && Y >= 0 && Y < h) {
            if (!HALF) {
                const uint8_t *px = src8 + ((size_t)Y * w + X) * 3;
                L[0][i] = s_lut[px[0]];
                L[1][i] = s_lut[px[1]];
                L[2][i] = s_lut[px[2]];
            } else {
                float acc[3] = {0.0f, 0.0f, 0.0f};
#pragma unroll
                for (int dy = 0; dy < 2; dy++)
#pragma unroll
                    for (int dx = 0; dx < 2; dx++) {
                        const int ix = 2 * X + dx, iy = 2 * Y + dy;
                        if (ix < W8 && iy < H8) {
                            const uint8_t *px = src8 + ((size_t)iy * W8 + ix) * 3;
#pragma unroll
                            for (int c = 0; c < 3; c++) acc[c] += 0.25f * s_lut[px[c]];
                        }
                    }
#pragma unroll
                for (int c = 0; c < 3; c++) {
                    float v = acc[c];
                    if ((W8 & 1) && X == w - 1) v *= 2.0f;
                    if ((H8 & 1) && Y == h - 1) v *= 2.0f;
                    L[c][i] = v;
                }
            }
        }
    }
    __syncthreads();
    // IN (block-uniform): the whole 36x36 region lies inside the image, so nothing is mirrored and nothing is masked
    auto stages = [&](auto in_tag) {
        constexpr bool IN = decltype(in_tag)::value;
        // row pass on every in-image row of the region, tile columns only; mirror in GLOBAL coordinates
        for (int i = threadIdx.x; i < FR * FT; i += TPB) {
            const int tx = i % FT, ly = i / FT, X = x0 + tx, Y = gy0 + ly;
            if (IN || (X < w && Y >= 0 && Y < h)) {
                const int c0 = X - gx0;
                const int m1 = IN ? c0 - 1 : (int)mirror(X - 1, w) - gx0, p1 = IN ? c0 + 1 : (int)mirror(X + 1, w) - gx0,
                          m2 = IN ? c0 - 2 : (int)mirror(X - 2, w) - gx0, p2 = IN ? c0 + 2 : (int)mirror(X + 2, w) - gx0;
    #pragma unroll
                for (int c = 0; c < 3; c++) {
                    const float *r = &L[c][ly * FR];
                    H[c][i] = r[c0] * w0 + (r[m1] + r[p1]) * w1 + (r[m2] + r[p2]) * w2;
                }
            }
        }
        __syncthreads();
        for (int i = threadIdx.x; i < FT * FT; i += TPB) {
            const int tx = i % FT, ty = i / FT, X = x0 + tx, Y = y0 + ty;
            if (!IN && (X >= w || Y >= h)) continue;
            const int r0 = Y - gy0;
            const int rm1 = IN ? r0 - 1 : (int)mirror(Y - 1, h) - gy0, rp1 = IN ? r0 + 1 : (int)mirror(Y + 1, h) - gy0,
                      rm2 = IN ? r0 - 2 : (int)mirror(Y - 2, h) - gy0, rp2 = IN ? r0 + 2 : (int)mirror(Y + 2, h) - gy0;
            float bl[3], ln[3];
    #pragma unroll
            for (int c = 0; c < 3; c++) {
                const float *col = &H[c][tx];
                bl[c] = col[r0 * FT] * w0 + (col[rm1 * FT] + col[rp1 * FT]) * w1 + (col[rm2 * FT] + col[rp2 * FT]) * w2;
                ln[c] = L[c][(ty + 2) * FR + tx + 2];
            }
            const float mn = 1e-4f;
            float p0, p1v, p2v;
            opsin_absorbance(bl[0] * intensity_target, bl[1] * intensity_target, bl[2] * intensity_target, p0, p1v, p2v);
            p0 = p0 > mn ? p0 : mn;
            p1v = p1v > mn ? p1v : mn;
            p2v = p2v > mn ? p2v : mn;
            // p >= 1e-4 (clamped above), gamma in [21, ~150]: IEEE quotients without the range-scaling steps (ce_internal.h)
            float s0 = ce_div_noscale(gamma_f(p0), p0), s1 = ce_div_noscale(gamma_f(p1v), p1v), s2 = ce_div_noscale(gamma_f(p2v), p2v);
            s0 = s0 > mn ? s0 : mn;
            s1 = s1 > mn ? s1 : mn;
            s2 = s2 > mn ? s2 : mn;
            float c0, c1, c2;
            opsin_absorbance(ln[0] * intensity_target, ln[1] * intensity_target, ln[2] * intensity_target, c0, c1, c2);
            c0 *= s0;
            c1 *= s1;
            c2 *= s2;
            const float min01 = 1.7557483643287353f, min2 = 12.226454707163354f;
            c0 = c0 > min01 ? c0 : min01;
            c1 = c1 > min01 ? c1 : min01;
            c2 = c2 > min2 ? c2 : min2;
            const size_t o = (size_t)slot * 3 * g.plane + (size_t)Y * g.pitch + X;
            xyb[o] = c0 - c1;
            xyb[o + g.plane] = c0 + c1;
            xyb[o + 2 * g.plane] = c2;
        }
    };
    if (interior)
        stages(std::true_type{});
    else
        stages(std::false_type{});
}

// ---- SeparateFrequencies pointwise stages ------------------------------------------------------------------
__device__ __forceinline__ float remove_range(float w, float x) { return x > w ? x - w : (x < -w ? x + w : 0.0f); }
__device__ __forceinline__ float amplify_range(float w, float x) { return x > w ? x + w : (x < -w ? x - w : x + x); }
__device__ __forceinline__ float maximum_clamp(float v, float maxval)
{
    const float kMul = 0.724216145665f;
    if (v >= maxval) return __builtin_fmaf(v - maxval, kMul, maxval);
    if (v < -maxval) return __builtin_fmaf(v + maxval, kMul, -maxval);
    return v;
}

// ---- column blur of all planes of one image slot + the pointwise stage that consumes them, fused ---------------
// SeparateFrequencies alternates a blur with a pointwise split; as separate kernels every split re-reads the blur's
// output and the band it splits.  Here a block blurs the NP planes of its 64 x 32 tile one after the other (same
// LDS tile, results kept in registers) and then runs the split on registers, reading only the raw band:
//   EPI_LF (3 planes, sigma 7.156): lf = blur(xyb);  mf_raw = xyb - lf -> psy MF;  XybLowFreqToVals(lf) -> psy LF
//   EPI_MF (3 planes, sigma 3.225): mf = blur(mf_raw);  hf_raw = mf_raw - mf (X suppressed by Y) -> psy HF;
//                                   range-shaped mf -> psy MF
//   EPI_HF (2 planes, sigma 1.564): hf = blur(hf_raw);  uhf = hf_raw - hf, clamps / ranges -> psy UHF, HF
// Arithmetic per element is exactly that of k_ba_blur_v followed by the old pointwise kernels.
//
// HV (the MF and HF stages, 15 and 7 taps): the ROW blur runs in the same kernel - the raw band's tile (64 + LEN - 1 rows,
// 64 + 16 columns, zero outside the image) is staged in LDS, every tile row is filtered into the column pass's tile
// (taps in ascending order, border scale of k_ba_blur_h: the same sums), so the row-blurred planes never travel through
// HBM.  A fused stage reads its raw band with a halo, so no stage may overwrite its own input: the raw bands live in
// scratch planes (`aux`): LF stage -> raw MF -> aux_out;  MF stage: aux_in -> MF (psy), raw HF -> aux_out;  HF stage:
// aux_in -> HF, UHF (psy), mask input.  (The 33-tap LF stage keeps its separate row pass: its halo would be half a tile.)
//   EPI_MASK (1 plane per slot, sigma 2.7, HV): the mask input's blur, no pointwise stage: tmp[slot] -> aux_out[slot]
enum { EPI_LF = 0, EPI_MF = 1, EPI_HF = 2, EPI_MASK = 3 };

// MaskPsychoImage's input at one pixel: DiffPrecompute of (UHF + HF) of X and Y (pointwise on the FINAL band values)
__device__ __forceinline__ float mask_pre_one(float uhf0, float hf0, float uhf1, float hf1)
{
    const float muls[3] = {2.5f, 0.4f, 0.4f};
    const float xdiff = (uhf0 + hf0) * muls[0];
    const float ydiff = uhf1 * muls[1] + hf1 * muls[2];
    const float m = sqrtf(xdiff * xdiff + ydiff * ydiff);
    const float kMul = 6.19424080439f, kBias = 12.61050594197f;
    const float bias = kMul * kBias;
    const float sqrt_bias = sqrtf(bias);
    return sqrtf(kMul * fabsf(m) + bias) - sqrt_bias;
}

template <int LEN, int EPI, bool HV, int TR = 64>
__global__ __launch_bounds__(TPB) void k_ba_blur_v_split(const float *__restrict__ tmp, const float *__restrict__ xyb,
                                                         float *__restrict__ psy, geom g, blur_kernel bk, float inv_wsum,
                                                         uint32_t n_refs_used, uint32_t max_refs, uint32_t z0,
                                                         float *__restrict__ mask_in, float *__restrict__ aux_out)
{
    constexpr int NP = EPI == EPI_HF ? 2 : EPI == EPI_MASK ? 1 : 3;
    // 64 columns x 64 rows per block (two 8-row groups per thread): the halo of LEN - 1 rows is read once per 64 rows
    constexpr int off = LEN / 2, TW = 64, PARTS = TR / 32, RAW = TR + LEN - 1;
    static_assert(TR % 32 == 0, "four waves x 8 rows per part");
    // HV: the staged input is 4 or 8 columns wider on each side (>= off, and its rows load as aligned float4)
    constexpr int LEFT = off <= 4 ? 4 : 8, IW = HV ? TW + 2 * LEFT : TW;
    static_assert(!HV || off <= LEFT, "row-pass halo");
    __shared__ __attribute__((aligned(16))) float tile[RAW * TW];
    __shared__ __attribute__((aligned(16))) float in_t[HV ? RAW * IW : 4];
    const uint32_t slot = slot_of(blockIdx.z + z0, n_refs_used, max_refs);
    const int x0 = blockIdx.x * TW, y0 = blockIdx.y * TR;
    const int c = threadIdx.x & 63, wv = threadIdx.x >> 6, gx = x0 + c;
    const bool col_live = gx < (int)g.w;
    float res[NP][PARTS][BW_OUT];
    // The (input) tile of plane q+1 is fetched (aligned float4 pieces of the rows, zero outside the image) into registers
    // while plane q is filtered: NF float4 per thread.
    constexpr int NF = (RAW * (IW / 4) + TPB - 1) / TPB;
    float4 pf[NF];
    auto fetch = [&](int q) {
        // the row-blurred planes (HV: the raw band) of this slot; the mask input is one plane per slot
        const float *p = tmp + (EPI == EPI_MASK ? (size_t)slot : (size_t)slot * 3 + q) * g.plane;
#pragma unroll
        for (int m = 0; m < NF; m++) {
            const int i = m * TPB + (int)threadIdx.x, r = i / (IW / 4), X = x0 - (HV ? LEFT : 0) + 4 * (i % (IW / 4)), Y = y0 - off + r;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (r < RAW && Y >= 0 && Y < (int)g.h && X >= 0 && X < (int)g.pitch) v = *reinterpret_cast<const float4 *>(p + (size_t)Y * g.pitch + X);
            v.x = X < (int)g.w ? v.x : 0.0f;
            v.y = X + 1 < (int)g.w ? v.y : 0.0f;
            v.z = X + 2 < (int)g.w ? v.z : 0.0f;
            v.w = X + 3 < (int)g.w ? v.w : 0.0f;
            pf[m] = v;
        }
    };
    auto stash = [&]() {  // the fetched pieces -> LDS (HV: the input tile, else the column pass's tile)
#pragma unroll
        for (int m = 0; m < NF; m++) {
            const int i = m * TPB + (int)threadIdx.x;
            if (i < RAW * (IW / 4)) *reinterpret_cast<float4 *>((HV ? in_t : tile) + 4 * i) = pf[m];
        }
    };
    // HV row pass: a thread filters four adjacent columns of a row from one aligned window of the input tile (NW float4 LDS
    // reads for four outputs); its column group is the same in every row it visits (TPB is a multiple of TW / 4)
    constexpr int S0 = LEFT - off, NW = (S0 + 3 + LEN + 3) / 4;
    const int rc4 = 4 * ((int)threadIdx.x % (TW / 4));
    // the row pass's border scale of a column: one thread per column forms it, the block shares the 64 through LDS (k_ba_blur_h)
    __shared__ __attribute__((aligned(16))) float s_sx[HV ? TW : 4];
    if (HV && threadIdx.x < TW)
        s_sx[threadIdx.x] = x0 + (int)threadIdx.x < (int)g.w ? border_scale<LEN>(bk, x0 + (int)threadIdx.x, (int)g.w, inv_wsum) : 0.0f;
    float scale_x[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    fetch(0);
    if (HV) {
        stash();
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 4; k++) scale_x[k] = s_sx[rc4 + k];
    }
#pragma unroll
    for (int q = 0; q < NP; q++) {
        if (HV) {
            // row pass of every tile row: taps in ascending order, zero outside the image (x + 0.0f == x), border scale
#pragma unroll 1
            for (int u = threadIdx.x; u < RAW * (TW / 4); u += TPB) {
                const int r = u / (TW / 4);
                const float4 *src = reinterpret_cast<const float4 *>(in_t + r * IW + rc4);
                float wnd[4 * NW];
#pragma unroll
                for (int m = 0; m < NW; m++) {
                    const float4 v = src[m];
                    wnd[4 * m] = v.x, wnd[4 * m + 1] = v.y, wnd[4 * m + 2] = v.z, wnd[4 * m + 3] = v.w;
                }
                float out[4];
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    float sum = 0.0f;
#pragma unroll
                    for (int j = 0; j < LEN; j++) sum = blur_tap(sum, wnd[S0 + k + j], bk.k[j]);
                    out[k] = sum * scale_x[k];  // scale 0 right of the image: the column pass sees zeros there
                }
                *reinterpret_cast<float4 *>(tile + r * TW + rc4) = make_float4(out[0], out[1], out[2], out[3]);
            }
            __syncthreads();  // the tile is complete; the input tile may be overwritten
            if (q + 1 < NP) fetch(q + 1);
        } else {
            if (q) __syncthreads();
            stash();
            if (q + 1 < NP) fetch(q + 1);
            __syncthreads();
        }
#pragma unroll
        for (int part = 0; part < PARTS; part++) {
            const int ly = 32 * part + 8 * wv, gy0 = y0 + ly;
            if (col_live && gy0 < (int)g.h) {
                float v[BW_OUT + LEN - 1];
#pragma unroll
                for (int j = 0; j < BW_OUT + LEN - 1; j++) v[j] = tile[(ly + j) * TW + c];
#pragma unroll
                for (int o = 0; o < BW_OUT; o++) {
                    float sum = 0.0f;
#pragma unroll
                    for (int j = 0; j < LEN; j++) sum = blur_tap(sum, v[o + j], bk.k[j]);
                    res[q][part][o] = sum * border_scale<LEN>(bk, gy0 + o, (int)g.h, inv_wsum);
                }
            }
        }
        if (HV && q + 1 < NP) {
            stash();
            __syncthreads();  // the next input tile is complete; this plane's tile has been read
        }
    }
    if (!col_live) return;
    const size_t pl = g.plane;
#pragma unroll
    for (int part = 0; part < PARTS; part++) {
        const int gy0 = y0 + 32 * part + 8 * wv;
        if (gy0 >= (int)g.h) break;
        float *ps = psy + (size_t)slot * PSY * g.plane + (size_t)gy0 * g.pitch + gx;
#pragma unroll
        for (int o = 0; o < BW_OUT; o++) {
            if (gy0 + o >= (int)g.h) break;
            float *q = ps + (size_t)o * g.pitch;
            // this pixel in the 3-plane scratch sets (raw bands in / out)
            const size_t ao = (size_t)slot * 3 * pl + (size_t)(gy0 + o) * g.pitch + gx;
            if (EPI == EPI_LF) {
                const float *xs = xyb + (size_t)slot * 3 * pl + (size_t)(gy0 + o) * g.pitch + gx;
                const float lx = res[0][part][o], ly = res[1][part][o], lb = res[2][part][o];
                aux_out[ao] = xs[0] - lx;  // raw MF: the MF stage reads it with a halo, so it cannot live where MF is written
                aux_out[ao + pl] = xs[pl] - ly;
                aux_out[ao + 2 * pl] = xs[2 * pl] - lb;
                // XybLowFreqToVals
                const float xmul = 33.832837186260f, ymul = 14.458268100570f, bmul = 49.87984651440f, y_to_b_mul = -0.362267051518f;
                const float bb = __builtin_fmaf(y_to_b_mul, ly, lb);
                q[LF2 * pl] = bb * bmul;
                q[LF0 * pl] = lx * xmul;
                q[LF1 * pl] = ly * ymul;
            } else if (EPI == EPI_MF) {
                const float kRemoveMfRange = 0.29f, kAddMfRange = 0.1f;
                const float mf0 = res[0][part][o], mf1 = res[1][part][o];
                const float hf0 = tmp[ao] - mf0, hf1 = tmp[ao + pl] - mf1;
                q[MF0 * pl] = remove_range(kRemoveMfRange, mf0);
                q[MF1 * pl] = amplify_range(kAddMfRange, mf1);
                q[MF2 * pl] = res[2][part][o];
                // SuppressXByY(hf[1], &hf[0])
                const float suppress = 46.0f, sv = 0.653020556257f, one_minus_s = 1.0f - 0.653020556257f;
                const float scaler = __builtin_fmaf(suppress / __builtin_fmaf(hf1, hf1, suppress), one_minus_s, sv);
                aux_out[ao] = scaler * hf0;  // raw HF, for the HF stage
                aux_out[ao + pl] = hf1;
            } else if (EPI == EPI_MASK) {
                aux_out[(size_t)slot * pl + (size_t)(gy0 + o) * g.pitch + gx] = res[0][part][o];
            } else {
                const float kRemoveHfRange = 1.5f, kAddHfRange = 0.132f, kRemoveUhfRange = 0.04f;
                const float kMaxclampHf = 28.4691806922f, kMaxclampUhf = 5.19175294647f, kMulYHf = 2.155f, kMulYUhf = 2.69313763794f;
                float hf0_out, uhf0_out, hf1_out, uhf1_out;
                {
                    const float hf = res[0][part][o];
                    const float uhf = tmp[ao] - hf;
                    hf0_out = remove_range(kRemoveHfRange, hf);
                    uhf0_out = remove_range(kRemoveUhfRange, uhf);
                    q[HF0 * pl] = hf0_out;
                    q[UHF0 * pl] = uhf0_out;
                }
                {
                    float hf = maximum_clamp(res[1][part][o], kMaxclampHf);
                    float uhf = tmp[ao + pl] - hf;
                    uhf = maximum_clamp(uhf, kMaxclampUhf);
                    uhf *= kMulYUhf;
                    uhf1_out = uhf;
                    q[UHF1 * pl] = uhf;
                    hf *= kMulYHf;
                    hf1_out = amplify_range(kAddHfRange, hf);
                    q[HF1 * pl] = hf1_out;
                }
                // the four band values are final here: the mask input of this pixel (one plane per image slot) costs no
                // extra pass over the PsychoImage
                mask_in[(size_t)slot * pl + (size_t)(gy0 + o) * g.pitch + gx] = mask_pre_one(uhf0_out, hf0_out, uhf1_out, hf1_out);
            }
        }
    }
}

// ---- per pair: Malta ----------------------------------------------------------------------------------------
struct malta_params {
    float norm2_0gt1, norm2_0lt1, norm1;
};

struct mline {
    int n;
    signed char d[9][2];
};

__device__ constexpr mline MALTA_HF[16] = {
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
__device__ constexpr mline MALTA_LF[16] = {
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

// 64 x 32 outputs per block from a zero-padded 72 x 40 LDS tile of (x, y) pairs (1.41x halo)
constexpr int MT = 64, MH = 4, ML = MT + 2 * MH;

// a Malta line whose taps run from the bottom row to the top one (every other line runs top to bottom)
__device__ constexpr bool malta_line_descends(const mline &ln)
{
    for (int j = 1; j < ln.n; j++)
        if (ln.d[j][1] > ln.d[j - 1][1]) return false;
    return ln.d[0][1] != ln.d[ln.n - 1][1];
}
__device__ constexpr bool malta_line_monotone(const mline &ln)
{
    bool up = true, down = true;
    for (int j = 1; j < ln.n; j++) {
        if (ln.d[j][1] < ln.d[j - 1][1]) up = false;
        if (ln.d[j][1] > ln.d[j - 1][1]) down = false;
    }
    return up || down;
}
__device__ constexpr bool malta_tables_monotone()
{
    for (int k = 0; k < 16; k++)
        if (!malta_line_monotone(MALTA_HF[k]) || !malta_line_monotone(MALTA_LF[k])) return false;
    return true;
}
static_assert(malta_tables_monotone(), "the row-streaming Malta unit needs every line's rows to be visited in one direction");

// ---- per pair, fused: the three Malta bands of channels X and Y + every channel's L2 terms -----------------------
// For channel c in {X, Y}: UHF (9-sample lines), HF and MF (5-sample lines).  Per band the two images' 72x72
// regions are read once, the asymmetric pre-scaled difference (MaltaDiffMap's first loop) goes to LDS (zero outside
// the image, as PaddedMaltaUnit does) and the 16 line sums are squared and accumulated.  Then the
// L2DiffAsymmetric (HF), L2Diff (MF) and SetL2Diff (LF) terms of the channel are added and ac[c] / dc[c] are written
// once.  Channel B has no Malta term: only L2Diff (MF) and SetL2Diff (LF).
struct malta_bands {
    malta_params p[2][3];  // [channel][band: uhf, hf, mf]
};

// Two correctly rounded f32 quotients a0 / b and a1 / b with ONE reciprocal.  hipcc expands every IEEE division into
// v_div_scale x2, v_rcp, two fused steps that refine the reciprocal, a product, two fused quotient corrections, v_div_fmas
// and v_div_fixup (11 instructions) and does not share anything between two divisions by the same denominator.  For
// operands whose quotient cannot overflow, underflow or involve a denormal - here b = norm1 + |..| in [5, 2^28] and
// the numerators are positive constants in [2, 4e7] - v_div_scale returns its operand unchanged, v_div_fmas is a plain
// fma and v_div_fixup returns the quotient, so the same arithmetic is 3 shared + 5 per numerator = 13 instructions
// instead of 22, bit for bit (ce_debug_div_sweep checks it against operator/ on the device).
__device__ __forceinline__ void div2_shared_rcp(float a0, float a1, float b, float &q0, float &q1)
{
    const float r = ce_rcp_refined(b);  // ce_internal.h
    q0 = ce_div_refined(a0, b, r);
    q1 = ce_div_refined(a1, b, r);
}

__device__ __forceinline__ float malta_pre_diff(float v0, float v1, const malta_params &mp)
{
    const float absval = 0.5f * (fabsf(v0) + fabsf(v1));
    const float diff = v0 - v1;
    float scaler, scaler2;  // norm2_0gt1 / (norm1 + absval), norm2_0lt1 / (norm1 + absval)
    div2_shared_rcp(mp.norm2_0gt1, mp.norm2_0lt1, mp.norm1 + absval, scaler, scaler2);
    const float r = scaler * diff;
    // The four branches of MaltaDiffMap's asymmetry term, without divergence, in f32 (the lineage forms it in f64: worth
    // 9e-8 of the score, tests/golden/sensitivity.json "ba_malta_f32" - the oracle's switch of that name is this arithmetic
    // bit for bit; round 2 paid ten f64-rate instructions per sample for it).  Mirror v1 for a negative v0:
    //   v0 <  0: v1 > -too_small  <=>  u < too_small   r - scaler2 (v1 + too_small) = r - scaler2 (too_small - u)
    //            v1 < -too_big    <=>  u > too_big     r + scaler2 (-v1 - too_big)  = r + scaler2 (u - too_big)
    //   v0 >= 0: v1 <  too_small  <=>  u < too_small   r + scaler2 (too_small - v1)
    //            v1 >  too_big    <=>  u > too_big     r - scaler2 (v1 - too_big)
    // with u = v0 < 0 ? -v1 : v1.
    const float fabs0 = fabsf(v0), too_small = 0.55f * fabs0, too_big = 1.05f * fabs0;
    const bool neg = v0 < 0;
    const float u = neg ? -v1 : v1;
    const bool lo = u < too_small, hi = u > too_big;
    const float impact = scaler2 * (lo ? too_small - u : u - too_big);
    const float rn = r + ((neg == lo) ? -impact : impact);
    return (lo || hi) ? rn : r;
}

// L2DiffAsymmetric's in-place accumulation for one sample of hf[c] (libjxl / oracle l2_diff_asymmetric)
__device__ __forceinline__ float l2_asym_acc(float total, float val0, float val1, float vw_0gt1, float vw_0lt1)
{
    const float diff = val0 - val1;
    total = __builtin_fmaf(diff * diff, vw_0gt1, total);
    const float fabs0 = fabsf(val0);
    const float too_small = 0.4f * fabs0, too_big = fabs0;
    const float if_neg = val1 > -too_small ? val1 + too_small : (val1 < -too_big ? -val1 - too_big : 0.0f);
    const float if_pos = val1 < too_small ? too_small - val1 : (val1 > too_big ? val1 - too_big : 0.0f);
    const float v = val0 < 0.0f ? if_neg : if_pos;
    return __builtin_fmaf(vw_0lt1, v * v, total);
}

// Channels X and Y are PACKED: the LDS tile holds (x-diff, y-diff) pairs and a thread forms the 16 line sums of MaltaUnit
// for two horizontally adjacent centres and both channels at once on v_pk_add_f32 / v_pk_fma_f32 (two IEEE operations
// each, so every channel's sums are exactly those of a scalar loop).  The 9 x 10 window of pairs is STREAMED row by
// row (five 16-byte LDS reads per row) into 32 running line sums instead of being held whole (180 registers, two waves
// per SIMD, VALU busy 59 %): a line's taps are added in the lineage's order - fourteen of the sixteen lines run top
// to bottom, so their running sum simply follows the rows; lines 7 and 12 run bottom to top, so their taps are parked
// until the row of their FIRST tap arrives and are then added in order.  Bit-identical sums, ~150 registers, and with
// a 64 x 32 tile (72 x 40 pairs + 64 x 32 running sums = 39 KB) three to four blocks per CU.
template <bool LF>
__device__ __forceinline__ void malta_rows_xy(const ba_f2 *__restrict__ base, ba_f2 (&acc)[2])
{
    ba_f2 sum[2][16];
    ba_f2 parked[2][16][9];  // only the entries of the two bottom-up lines are ever touched (all indices are static)
#pragma unroll
    for (int r = 0; r < 9; r++) {
        ba_f2 v[10];
        const float4 *row = reinterpret_cast<const float4 *>(base + r * ML);
#pragma unroll
        for (int q = 0; q < 5; q++) {
            const float4 t = row[q];
            v[2 * q] = ba_f2{t.x, t.y};
            v[2 * q + 1] = ba_f2{t.z, t.w};
        }
        const int dy = r - 4;
#pragma unroll
        for (int o = 0; o < 2; o++) {
#pragma unroll
            for (int k = 0; k < 16; k++) {
                const mline &ln = LF ? MALTA_LF[k] : MALTA_HF[k];
                if (!malta_line_descends(ln)) {
#pragma unroll
                    for (int j = 0; j < 9; j++)
                        if (j < ln.n && ln.d[j][1] == dy) {
                            const ba_f2 t = v[4 + o + ln.d[j][0]];
                            sum[o][k] = j == 0 ? t : sum[o][k] + t;  // 0 + t == t: the lineage's sum starts at zero
                        }
                } else if (dy == ln.d[0][1]) {  // the row of the line's first tap: everything below it is parked
                    ba_f2 acc_k = {0.0f, 0.0f};
#pragma unroll
                    for (int j = 0; j < 9; j++)
                        if (j < ln.n) {
                            const ba_f2 t = ln.d[j][1] == dy ? v[4 + o + ln.d[j][0]] : parked[o][k][j];
                            acc_k = j == 0 ? t : acc_k + t;
                        }
                    sum[o][k] = acc_k;
                } else {
#pragma unroll
                    for (int j = 0; j < 9; j++)
                        if (j < ln.n && ln.d[j][1] == dy) parked[o][k][j] = v[4 + o + ln.d[j][0]];
                }
            }
        }
    }
#pragma unroll
    for (int o = 0; o < 2; o++) {
        ba_f2 ret = {0.0f, 0.0f};
#pragma unroll
        for (int k = 0; k < 16; k++) ret = __builtin_elementwise_fma(sum[o][k], sum[o][k], ret);
        acc[o] = acc[o] + ret;
    }
}

// MR rows of outputs per block, NT threads (32 threads per output row pair-column, NT / 32 rows per step):
//   <32, 256>: 72 x 40 tile (1.41x halo), 39 KB LDS, four blocks per CU
//   <64, 512>: 72 x 72 tile (1.27x halo: 10 % fewer pre-scalings), 73 KB LDS, two blocks of eight waves per CU
// FINAL (the full-resolution level, launched after the half-resolution one): the pixel's diffmap value takes the
// half-resolution diffmap (AddSupersampled2x, weight 0.5) and goes straight into the score reductions - max, sum d^3, d^6,
// d^12 per tile - instead of to memory; !FINAL (the half-resolution level) writes its diffmap.
template <int MR, int NT, bool FINAL>
__global__ __launch_bounds__(NT, NT == 256 ? 3 : 4) void k_ba_malta_l2_xy(const float *__restrict__ psy, const uint32_t *__restrict__ pair_ref,
                                                           const float *__restrict__ blurred, const float *__restrict__ mask_vals,
                                                           float *__restrict__ diffmap, geom g, uint32_t max_refs,
                                                           malta_bands mb, const uint2 *__restrict__ work, uint32_t tiles_x,
                                                           const float *__restrict__ sub_map, geom gs, int has_sub,
                                                           float *__restrict__ blk_max, double *__restrict__ blk_sums, uint32_t n_blocks)
{
    constexpr int MLR = MR + 2 * MH, RSTEP = NT / 32;  // tile rows; output rows per step of the whole block
    __shared__ __attribute__((aligned(16))) ba_f2 s[MLR * ML];
    __shared__ __attribute__((aligned(16))) ba_f2 s_acc[MR * MT];  // the block's running sums; each thread owns its entries
    // XCD-aware 1-D launch (ce_build_xcd_list): the work list says which (tile, pair) this workgroup is
    const uint2 wi = work[blockIdx.x];
    if (wi.x == ~0u) return;  // padding entry
    const uint32_t p = wi.y, bx = wi.x % tiles_x, by = wi.x / tiles_x;
    const int x0 = (int)bx * MT - MH, y0 = (int)by * MR - MH;
    const float *a = psy + (size_t)pair_ref[p] * PSY * g.plane;
    const float *b = psy + (size_t)(max_refs + p) * PSY * g.plane;
    // thread -> two adjacent outputs (columns 2*tq, 2*tq+1 of row 8*sub + ty), four row groups per tile
    const int tq = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const float wmul[9] = {400.0f, 1.50815703118f, 0.0f, 2150.0f, 10.6195433239f, 16.2176043152f, 29.2353797994f, 0.844626970982f, 0.703646627719f};
    const float hf_asymmetry = 1.0f;
#pragma unroll
    for (int band = 0; band < 3; band++) {
        const malta_params mpx = mb.p[0][band], mpy = mb.p[1][band];
        const uint32_t plane0 = band == 0 ? UHF0 : band == 1 ? HF0 : MF0;  // channel X's plane; channel Y's is the next one
        const float *pax = a + (size_t)plane0 * g.plane, *pbx = b + (size_t)plane0 * g.plane;
        const float *pay = pax + g.plane, *pby = pbx + g.plane;
        for (int i = threadIdx.x; i < MLR * (ML / 4); i += NT) {
            const int ly = i / (ML / 4), lq = i % (ML / 4), gx = x0 + 4 * lq, gy = y0 + ly;
            float4 vax = make_float4(0.f, 0.f, 0.f, 0.f), vbx = vax, vay = vax, vby = vax;
            const bool in = gx >= 0 && gy >= 0 && gx < (int)g.pitch && gy < (int)g.h;
            if (in) {
                const size_t o = (size_t)gy * g.pitch + gx;
                vax = *reinterpret_cast<const float4 *>(pax + o);
                vbx = *reinterpret_cast<const float4 *>(pbx + o);
                vay = *reinterpret_cast<const float4 *>(pay + o);
                vby = *reinterpret_cast<const float4 *>(pby + o);
            }
            const bool i0 = in && gx < (int)g.w, i1 = in && gx + 1 < (int)g.w, i2 = in && gx + 2 < (int)g.w, i3 = in && gx + 3 < (int)g.w;
            float4 r01, r23;  // (x0, y0, x1, y1), (x2, y2, x3, y3)
            r01.x = i0 ? malta_pre_diff(vax.x, vbx.x, mpx) : 0.0f;
            r01.y = i0 ? malta_pre_diff(vay.x, vby.x, mpy) : 0.0f;
            r01.z = i1 ? malta_pre_diff(vax.y, vbx.y, mpx) : 0.0f;
            r01.w = i1 ? malta_pre_diff(vay.y, vby.y, mpy) : 0.0f;
            r23.x = i2 ? malta_pre_diff(vax.z, vbx.z, mpx) : 0.0f;
            r23.y = i2 ? malta_pre_diff(vay.z, vby.z, mpy) : 0.0f;
            r23.z = i3 ? malta_pre_diff(vax.w, vbx.w, mpx) : 0.0f;
            r23.w = i3 ? malta_pre_diff(vay.w, vby.w, mpy) : 0.0f;
            float4 *dst = reinterpret_cast<float4 *>(s + ly * ML + 4 * lq);
            dst[0] = r01;
            dst[1] = r23;
            // The HF / MF samples of the tile's OWN pixels are in registers here: their L2DiffAsymmetric (hf) / L2Diff (mf)
            // terms join the running block_diff_ac sums now (after the previous band's Malta sums, before this band's),
            // instead of being loaded again - 32 B per pixel - by the epilogue.  The same in-place accumulations as the
            // lineage's, in another order: the oracle's switch "ba_l2_early" is this order bit for bit.
            if (band >= 1 && ly >= MH && ly < MH + MR && lq >= MH / 4 && lq < (MH + MT) / 4 && gy < (int)g.h) {
                float4 *pa = reinterpret_cast<float4 *>(s_acc + (ly - MH) * MT + 4 * lq - MH);
                float4 t01 = pa[0], t23 = pa[1];  // (x0, y0, x1, y1), (x2, y2, x3, y3)
                if (band == 1) {
                    const float gx_ = wmul[0] * hf_asymmetry * 0.8f, lx_ = wmul[0] / hf_asymmetry * 0.8f;
                    const float gy_ = wmul[1] * hf_asymmetry * 0.8f, ly_ = wmul[1] / hf_asymmetry * 0.8f;
                    if (i0) t01.x = l2_asym_acc(t01.x, vax.x, vbx.x, gx_, lx_), t01.y = l2_asym_acc(t01.y, vay.x, vby.x, gy_, ly_);
                    if (i1) t01.z = l2_asym_acc(t01.z, vax.y, vbx.y, gx_, lx_), t01.w = l2_asym_acc(t01.w, vay.y, vby.y, gy_, ly_);
                    if (i2) t23.x = l2_asym_acc(t23.x, vax.z, vbx.z, gx_, lx_), t23.y = l2_asym_acc(t23.y, vay.z, vby.z, gy_, ly_);
                    if (i3) t23.z = l2_asym_acc(t23.z, vax.w, vbx.w, gx_, lx_), t23.w = l2_asym_acc(t23.w, vay.w, vby.w, gy_, ly_);
                } else {
                    auto l2 = [](float total, float v0, float v1, float w) {
                        const float diff = v0 - v1;
                        return __builtin_fmaf(diff * diff, w, total);
                    };
                    if (i0) t01.x = l2(t01.x, vax.x, vbx.x, wmul[3]), t01.y = l2(t01.y, vay.x, vby.x, wmul[4]);
                    if (i1) t01.z = l2(t01.z, vax.y, vbx.y, wmul[3]), t01.w = l2(t01.w, vay.y, vby.y, wmul[4]);
                    if (i2) t23.x = l2(t23.x, vax.z, vbx.z, wmul[3]), t23.y = l2(t23.y, vay.z, vby.z, wmul[4]);
                    if (i3) t23.z = l2(t23.z, vax.w, vbx.w, wmul[3]), t23.w = l2(t23.w, vay.w, vby.w, wmul[4]);
                }
                pa[0] = t01;
                pa[1] = t23;
            }
        }
        __syncthreads();
#pragma unroll 1
        for (int sub = 0; sub < MR / RSTEP; sub++) {
            ba_f2 acc[2] = {{0.f, 0.f}, {0.f, 0.f}};  // this band's sums; 0 + ret is exact, so adding them afterwards is the same sum
            const ba_f2 *base = s + (RSTEP * sub + ty) * ML + 2 * tq;
            // block_diff_ac accumulates band by band in the lineage: same order here
            if (band == 0)
                malta_rows_xy<false>(base, acc);
            else
                malta_rows_xy<true>(base, acc);
            float4 *pacc = reinterpret_cast<float4 *>(s_acc + (RSTEP * sub + ty) * MT + 2 * tq);
            float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
            if (band != 0) t = *pacc;
            *pacc = make_float4(t.x + acc[0].x, t.y + acc[0].y, t.z + acc[1].x, t.w + acc[1].y);
        }
        __syncthreads();
    }
    float red_m = 0.0f;  // the diffmap is non-negative
    double red_s3 = 0.0, red_s6 = 0.0, red_s12 = 0.0;
#pragma unroll 1
    for (int sub = 0; sub < MR / RSTEP; sub++)
#pragma unroll
    for (int r = 0; r < 2; r++) {
        const uint32_t x = bx * MT + 2 * tq + r, y = by * MR + RSTEP * sub + ty;
        if (x >= g.w || y >= g.h) continue;
        const size_t o = (size_t)y * g.pitch + x;
        const ba_f2 sums = s_acc[(RSTEP * sub + ty) * MT + 2 * tq + r];
        float acv[3], dcv[3];
#pragma unroll
        for (uint32_t c = 0; c < 3; c++) {
            // channels X and Y: the Malta sums of the three bands and the hf / mf L2 terms are all in `sums` (see the tile
            // fill); channel B has no Malta term and no hf: L2Diff on mf[2] only
            float total = c == 0 ? sums.x : c == 1 ? sums.y : 0.0f;
            if (c == 2) {
                const float diff = a[(MF0 + c) * g.plane + o] - b[(MF0 + c) * g.plane + o];
                total = __builtin_fmaf(diff * diff, wmul[3 + c], total);
            }
            acv[c] = total;
            {  // SetL2Diff on lf[c]
                const float diff = a[(LF0 + c) * g.plane + o] - b[(LF0 + c) * g.plane + o];
                dcv[c] = (diff * diff) * wmul[6 + c];
            }
        }
        // CombineChannelsToDiffmap for this pixel (the mask values come from k_ba_mask_vals, once per reference): the
        // block_diff_ac / block_diff_dc triples never leave registers
        const size_t rslot = pair_ref[p];
        const float maskval = mask_vals[(rslot * 2 + 0) * g.plane + o], dc_maskval = mask_vals[(rslot * 2 + 1) * g.plane + o];
        const float mdiff = blurred[rslot * g.plane + o] - blurred[((size_t)max_refs + p) * g.plane + o];
        float ac0 = acv[0], ac1 = acv[1], ac2 = acv[2];
        float dc0 = dcv[0], dc1 = dcv[1], dc2 = dcv[2];
        ac1 += 10.0f * mdiff * mdiff;  // kMaskToErrorMul
        const float xmul = 1.0f;
        ac0 *= xmul;
        dc0 *= xmul;
        const float mc_dc = dc0 * dc_maskval + dc1 * dc_maskval + dc2 * dc_maskval;
        const float mc_ac = ac0 * maskval + ac1 * maskval + ac2 * maskval;
        float d = sqrtf(mc_dc + mc_ac);
        if (!FINAL) {
            diffmap[(size_t)p * g.plane + o] = d;
            continue;
        }
        if (has_sub) {  // AddSupersampled2x
            const float kHeuristicMixingValue = 0.3f, wgt = 0.5f;
            d *= 1.0f - kHeuristicMixingValue * wgt;
            d += wgt * sub_map[(size_t)p * gs.plane + (size_t)(y / 2) * gs.pitch + x / 2];
        }
        const double dd = d, d3 = dd * dd * dd, d6 = d3 * d3;
        red_s3 += d3;
        red_s6 += d6;
        red_s12 += d6 * d6;
        red_m = fmaxf(red_m, d);
    }
    if (FINAL) {  // one partial per (pair, tile): thread order, then wave order - fixed, so the sums are reproducible
        __shared__ float s_max[NT / 64];
        __shared__ double s_sum[3][NT / 64];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            red_m = fmaxf(red_m, __shfl_down(red_m, off, 64));
            red_s3 += __shfl_down(red_s3, off, 64);
            red_s6 += __shfl_down(red_s6, off, 64);
            red_s12 += __shfl_down(red_s12, off, 64);
        }
        const int wv = threadIdx.x >> 6;
        if ((threadIdx.x & 63) == 0) {
            s_max[wv] = red_m;
            s_sum[0][wv] = red_s3;
            s_sum[1][wv] = red_s6;
            s_sum[2][wv] = red_s12;
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            float mm = s_max[0];
            double sa = 0, sb = 0, sc = 0;
            for (int k = 0; k < NT / 64; k++) {
                mm = fmaxf(mm, s_max[k]);
                sa += s_sum[0][k];
                sb += s_sum[1][k];
                sc += s_sum[2][k];
            }
            const size_t bi = (size_t)p * n_blocks + wi.x;
            blk_max[bi] = mm;
            blk_sums[bi * 3] = sa;
            blk_sums[bi * 3 + 1] = sb;
            blk_sums[bi * 3 + 2] = sc;
        }
    }
}

// ---- per pair: mask -----------------------------------------------------------------------------------------------
// DiffPrecompute(sqrt(xdiff^2 + ydiff^2)) of the reference (m0) and the test image (m1) of a pair

// StoreMin3: keep the three smallest values seen, sorted.  The state (min0 <= min1 <= min2) starts sorted for the
// non-negative inputs of the mask, so the insertion is a min/max network - no branches, no indexed temporaries
// (the branchy form was compiled into a scratch-memory array).
__device__ __forceinline__ void store_min3(float v, float &min0, float &min1, float &min2)
{
    const float a = fminf(min0, v), t = fmaxf(min0, v);
    const float b = fminf(min1, t), u = fmaxf(min1, t);
    min2 = fminf(min2, u);
    min1 = b;
    min0 = a;
}

// mask = FuzzyErosion(blurred0); ac[1] += 10 (blurred0 - blurred1)^2; then CombineChannelsToDiffmap — one pass
// MaskPsychoImage's per-pixel mask values depend on the REFERENCE's blurred mask input only (FuzzyErosion, then the two
// f64 rational curves of MaskY / MaskDcY): formed once per reference slot - not once per pair - into two planes that a
// reference handle keeps with the PsychoImage; the pair part of CombineChannelsToDiffmap is the Malta kernel's epilogue.
__global__ __launch_bounds__(TPB) void k_ba_mask_vals(const float *__restrict__ blurred, float *__restrict__ vals, geom g)
{
    const uint32_t z = blockIdx.z;  // reference slot
    BA_XY;
    const float *from = blurred + (size_t)z * g.plane;
    const int X = (int)x, Y = (int)y, W = (int)g.w, H = (int)g.h, S = 3;
    const uint32_t pitch = g.pitch;
#define at(yy, xx) from[(size_t)(yy) * pitch + (xx)]
    // FuzzyErosion: the three smallest of the centre and its eight neighbours at distance 3 that exist.  All nine
    // loads are issued unconditionally at clamped coordinates (independent, so they overlap); a neighbour that does
    // not exist enters as +inf, which store_min3 ignores.
    const int xl = max(X - S, 0), xr = min(X + S, W - 1), yu = max(Y - S, 0), yd = min(Y + S, H - 1);
    const float c0 = at(Y, X);
    const float v_l = at(Y, xl), v_lu = at(yu, xl), v_ld = at(yd, xl), v_r = at(Y, xr), v_ru = at(yu, xr), v_rd = at(yd, xr),
                v_u = at(yu, X), v_d = at(yd, X);
    const bool hl = X >= S, hr = X < W - S, hu = Y >= S, hd = Y < H - S;
    const float INF = __builtin_inff();
    float min0 = c0, min1 = 2 * min0, min2 = min1;
    store_min3(hl ? v_l : INF, min0, min1, min2);
    store_min3(hl && hu ? v_lu : INF, min0, min1, min2);
    store_min3(hl && hd ? v_ld : INF, min0, min1, min2);
    store_min3(hr ? v_r : INF, min0, min1, min2);
    store_min3(hr && hu ? v_ru : INF, min0, min1, min2);
    store_min3(hr && hd ? v_rd : INF, min0, min1, min2);
    store_min3(hu ? v_u : INF, min0, min1, min2);
    store_min3(hd ? v_d : INF, min0, min1, min2);
    const float mask = 0.45f * min0 + 0.3f * min1 + 0.25f * min2;
#undef at
    const double kGlobalScale = 1.0 / (17.83 * 0.790799174);
    const double val = (double)mask;
    double c = 2.5485944793 / ((0.451936922203 * val) + 0.829591754942);
    double rv = kGlobalScale * (1.0 + c);
    vals[((size_t)z * 2 + 0) * g.plane + o] = (float)(rv * rv);  // maskval
    c = 0.505054525019 / ((3.87449418804 * val) + 0.20025578522);
    rv = kGlobalScale * (1.0 + c);
    vals[((size_t)z * 2 + 1) * g.plane + o] = (float)(rv * rv);  // dc_maskval
}

// one block per pair: max and the three power sums over the block partials (fixed order), then the norms
__global__ __launch_bounds__(TPB) void k_ba_score(const float *__restrict__ blk_max, const double *__restrict__ blk_sums,
                                                  ce_dev_scores *__restrict__ scores, double *__restrict__ pnorm,
                                                  uint32_t n_blocks, uint32_t used_blocks, double npix)
{
    __shared__ float s_max[TPB];
    __shared__ double s_sum[3][TPB];
    const uint32_t p = blockIdx.x, t = threadIdx.x;
    float mx = 0.0f;
    double s[3] = {0, 0, 0};
    for (uint32_t k = t; k < used_blocks; k += TPB) {
        mx = fmaxf(mx, blk_max[(size_t)p * n_blocks + k]);
        for (int q = 0; q < 3; q++) s[q] += blk_sums[((size_t)p * n_blocks + k) * 3 + q];
    }
    s_max[t] = mx;
    for (int q = 0; q < 3; q++) s_sum[q][t] = s[q];
    __syncthreads();
    for (int off = TPB / 2; off > 0; off >>= 1) {
        if ((int)t < off) {
            s_max[t] = fmaxf(s_max[t], s_max[t + off]);
            for (int q = 0; q < 3; q++) s_sum[q][t] += s_sum[q][t + off];
        }
        __syncthreads();
    }
    if (t == 0) {
        scores[p].butteraugli = (double)s_max[0];
        const double opp = 1.0 / npix;
        pnorm[p] = (pow(opp * s_sum[0][0], 1.0 / 3.0) + pow(opp * s_sum[1][0], 1.0 / 6.0) + pow(opp * s_sum[2][0], 1.0 / 12.0)) / 3.0;
    }
}

// debug: div2_shared_rcp against operator/ on pseudo-random operands of the ranges Malta uses (and wider)
__global__ __launch_bounds__(256) void k_div_sweep(uint64_t seed, uint64_t count, unsigned long long *out)
{
    unsigned long long bad = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (uint64_t)gridDim.x * blockDim.x) {
        uint64_t x = (i + seed) * 0x9E3779B97F4A7C15ull;  // splitmix64
        x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
        x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
        x ^= x >> 31;
        // b, a0, a1 in [2^-40, 2^40), either sign for the numerators, a zero numerator now and then: random mantissas,
        // exponents from the hash - the range every call site of ce_div_noscale / div2_shared_rcp stays inside
        const uint32_t mb = (uint32_t)x & 0x7fffffu, ma = (uint32_t)(x >> 23) & 0x7fffffu, m2 = (uint32_t)(x >> 41) & 0x7fffffu;
        const uint32_t eb = 87u + (uint32_t)((x >> 17) % 80u), ea = 87u + (uint32_t)((x >> 7) % 80u), e2 = 87u + (uint32_t)((x >> 3) % 80u);
        const float b = __uint_as_float((eb << 23) | mb);
        float a0 = __uint_as_float((ea << 23) | ma | ((uint32_t)(x >> 60) & 1u) << 31), a1 = __uint_as_float((e2 << 23) | m2);
        if ((x >> 50 & 1023u) == 0) a0 = 0.0f;
        float q0, q1;
        div2_shared_rcp(a0, a1, b, q0, q1);
        bad += (__float_as_uint(q0) != __float_as_uint(a0 / b)) + (__float_as_uint(q1) != __float_as_uint(a1 / b));
        bad += __float_as_uint(ce_div_noscale(a1, b)) != __float_as_uint(a1 / b);
    }
    if (bad) atomicAdd(out, bad);
}

blur_kernel make_kernel(float sigma)
{
    blur_kernel bk{};
    const float m = 2.25f;
    const double scaler = -1.0 / (2.0 * (double)sigma * (double)sigma);
    int diff = (int)(m * std::fabs(sigma));
    if (diff < 1) diff = 1;
    for (int i = -diff; i <= diff; i++) bk.k[i + diff] = (float)std::exp(scaler * i * i);
    bk.len = 2 * diff + 1;
    for (int d = 0; d < diff && d < 16; d++) {
        float lo = 0.0f, hi = 0.0f;
        for (int j = diff - d; j < bk.len; j++) lo += bk.k[j];
        for (int j = 0; j <= diff + d; j++) hi += bk.k[j];
        bk.lo[d] = 1.0f / lo;
        bk.hi[d] = 1.0f / hi;
    }
    return bk;
}

float inv_weight_sum(const blur_kernel &bk)
{
    float wsum = 0.0f;
    for (int j = 0; j < bk.len; j++) wsum += bk.k[j];
    return 1.0f / wsum;
}

malta_params make_malta(double w_0gt1, double w_0lt1, double norm1, bool lf)
{
    const double len = 3.75, mulli = lf ? 0.611612573796 : 0.39905817637;
    const float kWeight0 = 0.5f, kWeight1 = 0.33f;
    const double w_pre0gt1 = mulli * std::sqrt(kWeight0 * w_0gt1) / (len * 2 + 1);
    const double w_pre0lt1 = mulli * std::sqrt(kWeight1 * w_0lt1) / (len * 2 + 1);
    return malta_params{(float)(w_pre0gt1 * norm1), (float)(w_pre0lt1 * norm1), (float)norm1};
}

}  // namespace

int ce_butteraugli_div_sweep(ce_ctx *ctx, uint64_t seed, uint64_t count, uint64_t *mismatches)
{
    unsigned long long *d = nullptr, h = 0;
    CE_HIP(ctx, hipMalloc(&d, sizeof(h)));
    CE_HIP(ctx, hipMemsetAsync(d, 0, sizeof(h), ctx->stream));
    CE_LAUNCH(ctx, "div_sweep", k_div_sweep, dim3(4096), dim3(256), 0, seed, count, d);
    CE_HIP(ctx, hipMemcpyAsync(&h, d, sizeof(h), hipMemcpyDeviceToHost, ctx->stream));
    CE_HIP(ctx, hipStreamSynchronize(ctx->stream));
    CE_HIP(ctx, hipFree(d));
    if (mismatches) *mismatches = h;
    return CE_OK;
}

void ce_butteraugli_free(ce_batch *b)
{
    for (int l = 0; l < 2; l++) {
        hipFree(b->ba_psy[l]);
        hipFree(b->ba_diff[l]);
        hipFree(b->ba_mask[l]);
        hipFree(b->ba_mask_vals[l]);
        b->ba_mask_vals[l] = nullptr;
        ce_free_xcd_list(&b->ba_work[l]);
        b->ba_psy[l] = b->ba_diff[l] = b->ba_mask[l] = nullptr;
    }
    for (auto &p : b->ba_s) hipFree(p), p = nullptr;
    for (auto &p : b->ba_s_half) hipFree(p), p = nullptr;
    if (b->ba_half_stream) hipStreamSynchronize(b->ba_half_stream);  // the context's stream: drained, not destroyed
    if (b->ev_ba_fork) hipEventDestroy(b->ev_ba_fork);
    if (b->ev_ba_join) hipEventDestroy(b->ev_ba_join);
    b->ba_half_stream = nullptr;
    b->ev_ba_fork = b->ev_ba_join = nullptr;
    hipFree(b->ba_blk_max);
    hipFree(b->ba_blk_sums);
    hipFree(b->ba_pnorm);
    b->ba_blk_max = nullptr;
    b->ba_blk_sums = nullptr;
    b->ba_pnorm = nullptr;
    b->ba_ref_src = nullptr;
    b->ba_ready = false;
}

static int ba_allocate(ce_batch *b)
{
    ce_ctx *ctx = b->ctx;
    auto set = [](ce_batch::ba_level &d, uint32_t w, uint32_t h) {
        d.w = w;
        d.h = h;
        d.pitch = (w + 31u) & ~31u;
        d.plane = (size_t)d.pitch * h;
    };
    set(b->ba[0], b->w, b->h);
    set(b->ba[1], (b->w + 1) / 2, (b->h + 1) / 2);
    b->ba_levels = (b->ba[1].w >= 8 && b->ba[1].h >= 8) ? 2 : 1;
    const size_t slots = (size_t)b->max_refs + b->max_pairs, P = b->max_pairs, p0 = b->ba[0].plane;
    for (int l = 0; l < b->ba_levels; l++) {
        CE_HIP(ctx, hipMalloc(&b->ba_psy[l], slots * PSY * b->ba[l].plane * sizeof(float)));
        if (l == 1) CE_HIP(ctx, hipMalloc(&b->ba_diff[l], P * b->ba[l].plane * sizeof(float)));  // the full-resolution diffmap is never stored
    }
    for (auto &p : b->ba_s) CE_HIP(ctx, hipMalloc(&p, slots * 3 * p0 * sizeof(float)));
    // blurred mask input per image slot and level, the references' two mask-value planes per level (both persist like the
    // PsychoImage)
    for (int l = 0; l < b->ba_levels; l++) {
        CE_HIP(ctx, hipMalloc(&b->ba_mask[l], slots * b->ba[l].plane * sizeof(float)));
        CE_HIP(ctx, hipMalloc(&b->ba_mask_vals[l], (size_t)b->max_refs * 2 * b->ba[l].plane * sizeof(float)));
    }
    b->ba_blocks = ((b->ba[0].w + 63) / 64) * ((b->ba[0].h + 3) / 4);
    CE_HIP(ctx, hipMalloc(&b->ba_blk_max, P * b->ba_blocks * sizeof(float)));
    CE_HIP(ctx, hipMalloc(&b->ba_blk_sums, P * b->ba_blocks * 3 * sizeof(double)));
    CE_HIP(ctx, hipMalloc(&b->ba_pnorm, P * sizeof(double)));
    return CE_OK;
}

static int ba_prepare(ce_batch *b)
{
    if (b->ba_ready) return CE_OK;
    const int rc = ba_allocate(b);
    if (rc != CE_OK) {  // all or nothing (see ce_ssim2_prepare)
        const std::string why = b->ctx->err;
        ce_butteraugli_free(b);
        (void)hipGetLastError();
        b->ctx->err = why;
        return rc;
    }
    b->ba_ready = true;
    return CE_OK;
}

static bool ba_one_stream()  // CE_BA_LEVEL_STREAMS=1: both resolution levels on one stream whatever the batch size (A/B knob)
{
    static const bool v = [] {
        const char *e = std::getenv("CE_BA_LEVEL_STREAMS");
        return e && e[0] == '1';
    }();
    return v;
}

int ce_launch_butteraugli(ce_batch *b, const uint8_t *d_refs, uint32_t n_refs_used, uint32_t n_pairs, float intensity_target)
{
    ce_ctx *ctx = b->ctx;
    int rc = ba_prepare(b);
    if (rc != CE_OK) return rc;
    const uint32_t n_slots = n_refs_used + n_pairs, mr = b->max_refs;
    // reference handle (ce_ref_*): the references' PsychoImage (both resolutions) of an earlier launch with the same
    // intensity target is still in ba_psy, so only the distorted slots go through the per-image chain
    const bool cached = b->keep_ref_pyramid && b->ba_ref_src == d_refs && b->ba_ref_count >= n_refs_used && b->ba_ref_intensity == intensity_target;
    const uint32_t z0 = cached ? n_refs_used : 0, nz = n_slots - z0;
    if (!cached) b->ref_builds[2]++;
    const blur_kernel k12 = make_kernel(1.2f), kLf = make_kernel(7.15593339443f), kHf = make_kernel(3.22489901262f),
                      kUhf = make_kernel(1.56416327805f), kMask = make_kernel(2.7f);
    float sw = 0.0f;
    for (int j = 0; j < 5; j++) sw += k12.k[j];
    const float sc = 1.0f / sw, w0 = k12.k[2] * sc, w1 = k12.k[3] * sc, w2 = k12.k[4] * sc;
    const double hf_asymmetry = 1.0;
    const malta_params mUhfY = make_malta(1.10039032555 * hf_asymmetry, 1.10039032555 / hf_asymmetry, 71.7800275169, false);
    const malta_params mUhfX = make_malta(173.5 * hf_asymmetry, 173.5 / hf_asymmetry, 5.0, false);
    const malta_params mHfY = make_malta(18.7237414387 * std::sqrt(hf_asymmetry), 18.7237414387 / std::sqrt(hf_asymmetry), 4498534.45232, true);
    const malta_params mHfX = make_malta(6923.99476109 * std::sqrt(hf_asymmetry), 6923.99476109 / std::sqrt(hf_asymmetry), 8051.15833247, true);
    const malta_params mMfY = make_malta(37.0819870399, 37.0819870399, 130262059.556, true);
    const malta_params mMfX = make_malta(8246.75321353, 8246.75321353, 1009002.70582, true);

    // half resolution first: the full-resolution level's Malta kernel folds that diffmap into its own values and reduces
    // them to the score partials.
    // A small batch (the one-pair-per-call regime) is a chain of 16 dependent launches of a few workgroups each; the two
    // resolution levels are independent until the full-resolution Malta kernel, so the half-resolution chain runs on a stream
    // of its own (with its own scratch planes) beside the full-resolution one: two cross-stream events instead of seven
    // launches on the critical path (profiles/r03_experiments.md section 14).
    const bool two_streams = b->ba_levels == 2 && !ctx->prof_serial && (double)n_pairs * b->w * b->h <= 4e6 && !ba_one_stream();
    hipStream_t s_main = CE_STREAM(ctx), s_half = s_main;
    if (two_streams) {
        if (!b->ba_half_stream) {
            if (!(b->ba_half_stream = ce_ctx_aux_stream(ctx, ce_ctx::AUX_BA_HALF))) return CE_ERR_BACKEND;
            CE_HIP(ctx, hipEventCreateWithFlags(&b->ev_ba_fork, hipEventDisableTiming));
            CE_HIP(ctx, hipEventCreateWithFlags(&b->ev_ba_join, hipEventDisableTiming));
            const size_t slots = (size_t)b->max_refs + b->max_pairs;
            for (auto &p : b->ba_s_half) CE_HIP(ctx, hipMalloc(&p, slots * 3 * b->ba[1].plane * sizeof(float)));
        }
        s_half = b->ba_half_stream;
        CE_HIP(ctx, hipEventRecord(b->ev_ba_fork, s_main));
        CE_HIP(ctx, hipStreamWaitEvent(s_half, b->ev_ba_fork, 0));
    }
    uint32_t final_tiles = 0;
    for (int l = b->ba_levels - 1; l >= 0; l--) {
        const auto &d = b->ba[l];
        const geom g{d.w, d.h, d.pitch, d.plane};
        const dim3 gx((d.w + 63) / 64, (d.h + 3) / 4, 1);
        auto G = [&](uint32_t z) { return dim3(gx.x, gx.y, z); };
        hipStream_t st = l == 1 ? s_half : s_main;
        float *const *scr = (two_streams && l == 1) ? b->ba_s_half : b->ba_s;
        float *psy = b->ba_psy[l], *sA = scr[0], *sC = scr[2];
        // ---- per image slot: PsychoImage ----
        const plane_sel s3{3, 0, 3};
        const dim3 ft((d.w + FT - 1) / FT, (d.h + FT - 1) / FT, nz);
        if (l == 0) {
            CE_LAUNCH_ON(ctx, st, "ba_front_u8", k_ba_front<false>, ft, dim3(TPB), 0, d_refs, b->d_tests, ctx->d_lut_ssim2, g, sC, g, w0, w1, w2,
                      intensity_target, b->img_bytes, n_refs_used, mr, z0);
        } else {
            const auto &pd = b->ba[0];
            CE_LAUNCH_ON(ctx, st, "ba_front_half", k_ba_front<true>, ft, dim3(TPB), 0, d_refs, b->d_tests, ctx->d_lut_ssim2,
                      geom{pd.w, pd.h, pd.pitch, pd.plane}, sC, g, w0, w1, w2, intensity_target, b->img_bytes, n_refs_used, mr, z0);
        }
        // LF = blur(xyb, 7.156) -> psy[LF0..2]
        // SeparateFrequencies: row blur of a band, then column blur fused with the pointwise split (k_ba_blur_v_split)
        {
            if (kLf.len != 33 || kHf.len != 15 || kUhf.len != 7) {
                ctx->err = "unexpected blur kernel length";
                return CE_ERR_BACKEND;
            }
            const dim3 gh3((g.w + 255) / 256, (g.h + 8 * BH_TILES - 1) / (8 * BH_TILES), nz * 3);
            const dim3 gvs((g.w + 63) / 64, (g.h + 63) / 64, nz);
            float *sB = scr[1];
            // LF: row pass sC -> sA, column pass + split: LF -> psy, raw MF -> sB
            CE_LAUNCH_ON(ctx, st, "ba_blur_h33", k_ba_blur_h<33>, gh3, dim3(TPB), 0, (const float *)sC, sA, g, s3, s3, kLf, inv_weight_sum(kLf),
                      n_refs_used, mr, 1, z0);
            // rows per block of the column / fused stages: 32 (default; 26 / 21 KB of LDS and ~100 / 56 registers: five blocks per CU)
            // or 64 (CE_HV_ROWS=64: smaller halo, 45 / 38 KB, three or four blocks).  Measured (profiles/r02_experiments.md
            // section 18): 32 rows 0.50 + 0.33 ms per step solo against 0.53 + 0.41 (MF, HF), 0.48 against 0.53 (LF); headline
            // equal, 4K grid +3 %
            static const int hv_rows = [] {
                const char *e = std::getenv("CE_HV_ROWS");
                return e && std::atoi(e) == 64 ? 64 : 32;
            }();
            const dim3 gvs32((g.w + 63) / 64, (g.h + 31) / 32, nz);
            if (hv_rows == 32)
                CE_LAUNCH_ON(ctx, st, "ba_blur_v_lf", (k_ba_blur_v_split<33, EPI_LF, false, 32>), gvs32, dim3(TPB), 0, (const float *)sA,
                          (const float *)sC, psy, g, kLf, inv_weight_sum(kLf), n_refs_used, mr, z0, (float *)nullptr, sB);
            else
            CE_LAUNCH_ON(ctx, st, "ba_blur_v_lf", (k_ba_blur_v_split<33, EPI_LF, false>), gvs, dim3(TPB), 0, (const float *)sA, (const float *)sC, psy,
                      g, kLf, inv_weight_sum(kLf), n_refs_used, mr, z0, (float *)nullptr, sB);
            // MF: row + column pass of raw MF (sB) + split: MF -> psy, raw HF -> sA (its old contents are dead)
            if (hv_rows == 32)
                CE_LAUNCH_ON(ctx, st, "ba_blur_hv_mf", (k_ba_blur_v_split<15, EPI_MF, true, 32>), gvs32, dim3(TPB), 0, (const float *)sB,
                          (const float *)nullptr, psy, g, kHf, inv_weight_sum(kHf), n_refs_used, mr, z0, (float *)nullptr, sA);
            else
            CE_LAUNCH_ON(ctx, st, "ba_blur_hv_mf", (k_ba_blur_v_split<15, EPI_MF, true>), gvs, dim3(TPB), 0, (const float *)sB, (const float *)nullptr,
                      psy, g, kHf, inv_weight_sum(kHf), n_refs_used, mr, z0, (float *)nullptr, sA);
            // HF: row + column pass of raw HF (sA) + split: HF, UHF -> psy, the mask input -> sB (raw MF is dead)
            if (hv_rows == 32)
                CE_LAUNCH_ON(ctx, st, "ba_blur_hv_hf", (k_ba_blur_v_split<7, EPI_HF, true, 32>), gvs32, dim3(TPB), 0, (const float *)sA,
                          (const float *)nullptr, psy, g, kUhf, inv_weight_sum(kUhf), n_refs_used, mr, z0, sB, (float *)nullptr);
            else
            CE_LAUNCH_ON(ctx, st, "ba_blur_hv_hf", (k_ba_blur_v_split<7, EPI_HF, true>), gvs, dim3(TPB), 0, (const float *)sA, (const float *)nullptr,
                      psy, g, kUhf, inv_weight_sum(kUhf), n_refs_used, mr, z0, sB, (float *)nullptr);
        }

        // mask input: DiffPrecompute of HF + UHF (written by the HF split's epilogue into ba_s[1]), blurred with sigma 2.7 -
        // per image slot (the references' once per reference; cached with the PsychoImage for reference handles), into
        // the level's own per-slot planes; then the
        // references' mask values (FuzzyErosion + the two mask curves), also once per reference
        {
            if (kMask.len != 13) {
                ctx->err = "unexpected blur kernel length";
                return CE_ERR_BACKEND;
            }
            const dim3 gm((g.w + 63) / 64, (g.h + 31) / 32, nz);
            CE_LAUNCH_ON(ctx, st, "ba_blur_hv_mask", (k_ba_blur_v_split<13, EPI_MASK, true, 32>), gm, dim3(TPB), 0, (const float *)scr[1],
                      (const float *)nullptr, psy, g, kMask, inv_weight_sum(kMask), n_refs_used, mr, z0, (float *)nullptr, b->ba_mask[l]);
        }
        if (!cached)
            CE_LAUNCH_ON(ctx, st, "ba_mask_vals", k_ba_mask_vals, G(n_refs_used), dim3(TPB), 0, (const float *)b->ba_mask[l], b->ba_mask_vals[l], g);

        // ---- per pair: Malta + L2 terms + CombineChannelsToDiffmap -> the level's diffmap ----
        // tile height: 64 rows / 512 threads for images that fill the chip with such tiles, else 32 rows / 256 threads
        // (CE_MALTA_ROWS=32|64 forces one: measurement knob)
        static const int forced_rows = [] {
            const char *e = std::getenv("CE_MALTA_ROWS");
            const int v = e ? std::atoi(e) : 0;
            return v == 32 || v == 64 ? v : 0;
        }();
        const int malta_rows = forced_rows ? forced_rows : 32;
        malta_bands mb;
        mb.p[0][0] = mUhfX; mb.p[0][1] = mHfX; mb.p[0][2] = mMfX;
        mb.p[1][0] = mUhfY; mb.p[1][1] = mHfY; mb.p[1][2] = mMfY;
        const uint32_t tiles_x = (d.w + MT - 1) / MT, tiles_y = (d.h + (uint32_t)malta_rows - 1) / (uint32_t)malta_rows;
        if ((rc = ce_build_xcd_list(b, n_pairs, tiles_x * tiles_y, &b->ba_work[l])) != CE_OK) return rc;
        const bool has_sub = b->ba_levels == 2;
        const auto &ds = b->ba[has_sub ? 1 : 0];
        const geom gsub{ds.w, ds.h, ds.pitch, ds.plane};
#define CE_MALTA_LAUNCH(ROWS, THREADS, FINAL)                                                                                      \
    CE_LAUNCH_ON(ctx, st, "ba_malta_l2", (k_ba_malta_l2_xy<ROWS, THREADS, FINAL>), dim3(b->ba_work[l].len), dim3(THREADS), 0, psy, b->d_pair_ref, \
              (const float *)b->ba_mask[l], (const float *)b->ba_mask_vals[l], FINAL ? (float *)nullptr : b->ba_diff[1], g, mr, mb,       \
              (const uint2 *)b->ba_work[l].d, tiles_x, FINAL ? (const float *)b->ba_diff[1] : (const float *)nullptr, gsub,               \
              has_sub ? 1 : 0, b->ba_blk_max, b->ba_blk_sums, b->ba_blocks)
        if (l == 0) {
            final_tiles = tiles_x * tiles_y;
            if (two_streams) CE_HIP(ctx, hipStreamWaitEvent(s_main, b->ev_ba_join, 0));  // the half-resolution diffmap
            if (malta_rows == 64)
                CE_MALTA_LAUNCH(64, 512, true);
            else
                CE_MALTA_LAUNCH(32, 256, true);
        } else {
            if (malta_rows == 64)
                CE_MALTA_LAUNCH(64, 512, false);
            else
                CE_MALTA_LAUNCH(32, 256, false);
            if (two_streams) CE_HIP(ctx, hipEventRecord(b->ev_ba_join, s_half));
        }
#undef CE_MALTA_LAUNCH
    }
    if (b->keep_ref_pyramid && !cached) {
        b->ba_ref_src = d_refs;
        b->ba_ref_count = n_refs_used;
        b->ba_ref_intensity = intensity_target;
    }
    const auto &d0 = b->ba[0];
    CE_LAUNCH(ctx, "ba_score", k_ba_score, dim3(n_pairs), dim3(TPB), 0, b->ba_blk_max, b->ba_blk_sums, b->d_scores, b->ba_pnorm,
              b->ba_blocks, final_tiles, (double)d0.w * (double)d0.h);
    CE_HIP(ctx, hipGetLastError());
    return CE_OK;
}
