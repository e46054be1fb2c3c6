// DSSIM on gfx950 — replaces rgb8_to_dssim_image + calculate_dssim
// (/root/reference/src/metrics/dssim.rs:102-114, 40-71 -> dssim_core::Dssim::{create_image, compare}).
//
// Per level (5 levels, each the 2x2 average of the previous in LINEAR RGB, floor sizes):
//   per image slot ("create_image", references once per reference, not once per pair):
//     linear RGB -> normalised L*a*b*  ->  chroma planes pre-blurred (= img);  references also: mu = blur(img),
//     sq = blur(img*img), kept per level for all their pairs (and across launches for a reference handle)
//   per pair ("compare"):
//     the distorted image's mu / sq and i12 = blur(img1*img2) from the one img tile the kernel loads anyway ->
//     SSIM map over the channel-averaged statistics -> mean -> mean |avg - ssim|
//   (a distorted image's mu / sq are used by exactly one compare: computing them there instead of in create_image
//   saves their round trip through HBM - 48 B per pixel and level - and moves their arithmetic from the VALU-bound
//   create kernels into the compare kernel, which waits on memory)
// "blur" is the fixed 3x3 kernel applied twice with edge replication.  Every plane op keeps the
// oracle's f32 operation order (oracle/dssim.c), so planes are bit-identical; sums are f64.
// Build with -ffp-contract=off.
#include <algorithm>
#include <cstdlib>

#include "ce_internal.h"
#include "dssim_common.h"

namespace {

constexpr int TPB = 256;

__device__ __forceinline__ uint32_t slot_of(uint32_t z, uint32_t n_refs_used, uint32_t max_refs)
{
    return z < n_refs_used ? z : max_refs + (z - n_refs_used);
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

// One 3x3 pass with edge replication, evaluated on an LDS region of width RW whose local (0,0) is global
// (gx0, gy0).  The centre (lx, ly) must be inside the image; neighbours are clamped in GLOBAL coordinates,
// which is exactly "replicate the edge of that pass's input".  SQ squares every tap (blur of the squared
// image).  Summation order as in oracle/dssim.c: corners, edges, centre.
// INTERIOR: the whole region lies inside the image (block-uniform), so no neighbour is clamped and the nine taps
// are LDS reads at constant offsets from the centre.
template <int RW, bool SQ, bool INTERIOR = false>
__device__ __forceinline__ float pass3x3(const float *__restrict__ A, int lx, int ly, int gx0, int gy0, int w, int h)
{
    if (INTERIOR) {
        const float *c = A + ly * RW + lx;
        auto sqv = [](float v) { return SQ ? v * v : v; };
        const float K0 = 0.095332f, K1 = 0.118095f, K4 = 0.146293f;
        return (sqv(c[-RW - 1]) + sqv(c[-RW + 1]) + sqv(c[RW - 1]) + sqv(c[RW + 1])) * K0 +
               (sqv(c[-RW]) + sqv(c[-1]) + sqv(c[1]) + sqv(c[RW])) * K1 + sqv(c[0]) * K4;
    }
    const int X = gx0 + lx, Y = gy0 + ly;
    const int xm = max(X - 1, 0) - gx0, xp = min(X + 1, w - 1) - gx0;
    const int ym = max(Y - 1, 0) - gy0, yp = min(Y + 1, h - 1) - gy0;
    auto at = [&](int yy, int xx) {
        const float v = A[yy * RW + xx];
        return SQ ? v * v : v;
    };
    const float K0 = 0.095332f, K1 = 0.118095f, K4 = 0.146293f;
    return (at(ym, xm) + at(ym, xp) + at(yp, xm) + at(yp, xp)) * K0 + (at(ym, lx) + at(ly, xm) + at(ly, xp) + at(yp, lx)) * K1 +
           at(ly, lx) * K4;
}


// ---- Dssim::create_image for one level, fused: 32x32 tile + halo 4 in LDS ------------------------------------
// linear RGB (level 0: sRGB u8 through the host-powf table) -> L*a*b* -> chroma pre-blur (2 passes) ->
// mu = blur(img) (2 passes), sq = blur(img*img) (2 passes); also the next level's linear RGB.
// Writes img, mu, sq (9 planes) for the tile; every intermediate lives in LDS only.
constexpr int DT = 32, DR = DT + 8;

// The elements of an R x R LDS region that lie at least M away from its border, TPB at a time.  The loop runs over ALL
// R * R elements with a margin test, so consecutive lanes always touch consecutive LDS words: looping over the compact
// (R - 2M)^2 index space instead saves a tenth of the wave-instructions but makes every second half-wave straddle two
// region rows (R - 2M = 34 lanes of one, then the next), a two-way bank conflict on every access - measured in round 2:
// SQ_LDS_BANK_CONFLICT / SQ_ACTIVE_INST_LDS 0.00 -> 1.0 (compare), -> 1.8 (create), no time gained.
#define CE_MARGIN_LOOP(R, M, i, lx, ly)                                                                               \
    _Pragma("unroll 1") for (int i = threadIdx.x, lx = i % (R), ly = i / (R); i < (R) * (R); i += TPB, lx = i % (R), ly = i / (R)) \
        if (lx >= (M) && lx < (R) - (M) && ly >= (M) && ly < (R) - (M))

// FULL = a reference slot (img, mu, sq; region RW = tile + 8); otherwise a distorted image (img only: S1, S2 and the img
// store; its region is the tile + 4 the two chroma passes need - a quarter less L*a*b* work than the reference's 40 x 40)
template <bool IN, bool FULL, int RW>
__device__ __forceinline__ void dssim_create_stages(float (&P)[6][DR * DR], float *__restrict__ img, float *__restrict__ mu,
                                                    float *__restrict__ sq, const lvl_geom &g, uint32_t slot, int x0, int y0)
{
    constexpr int HL = (RW - DT) / 2;
    static_assert(FULL ? HL == 4 : HL == 2, "halo");
    const int w = (int)g.w, h = (int)g.h, gx0 = x0 - HL, gy0 = y0 - HL;
    auto inside = [&](int lx, int ly) { return gx0 + lx >= 0 && gx0 + lx < w && gy0 + ly >= 0 && gy0 + ly < h; };
    // Six LDS planes are enough (38 KB, four blocks per CU instead of two): planes are reused as soon as their
    // contents are dead, and mu / sq are produced one after the other through the same three planes.
    // S1: chroma pre-blur pass 1 (margin 1): P1,P2 -> P3,P4
    CE_MARGIN_LOOP(RW, 1, i, lx, ly) {
        if (IN || inside(lx, ly)) {
            P[3][i] = pass3x3<RW, false, IN>(P[1], lx, ly, gx0, gy0, w, h);
            P[4][i] = pass3x3<RW, false, IN>(P[2], lx, ly, gx0, gy0, w, h);
        }
    }
    __syncthreads();
    // S2: chroma pre-blur pass 2 (margin 2): P3,P4 -> P1,P2 (their old contents are dead) ; img = (P0, P1, P2)
    CE_MARGIN_LOOP(RW, 2, i, lx, ly) {
        if (IN || inside(lx, ly)) {
            P[1][i] = pass3x3<RW, false, IN>(P[3], lx, ly, gx0, gy0, w, h);
            P[2][i] = pass3x3<RW, false, IN>(P[4], lx, ly, gx0, gy0, w, h);
        }
    }
    __syncthreads();
    if (!FULL) {  // a distorted image: img is all that leaves (its mu / sq are formed by the compare kernel)
        for (int i = threadIdx.x; i < DT * DT; i += TPB) {
            const int tx = i % DT, ty = i / DT, X = x0 + tx, Y = y0 + ty;
            if (IN || (X < w && Y < h)) {
                const size_t o = (size_t)slot * 3 * g.plane + (size_t)Y * g.pitch + X;
                const int li = (ty + HL) * RW + tx + HL;
                img[o] = P[0][li];
                img[o + g.plane] = P[1][li];
                img[o + 2 * g.plane] = P[2][li];
            }
        }
        return;
    }
    // S3a: first pass of mu (margin 3): img -> P3,P4,P5
    CE_MARGIN_LOOP(RW, 3, i, lx, ly) {
        if (IN || inside(lx, ly)) {
            P[3][i] = pass3x3<RW, false, IN>(P[0], lx, ly, gx0, gy0, w, h);
            P[4][i] = pass3x3<RW, false, IN>(P[1], lx, ly, gx0, gy0, w, h);
            P[5][i] = pass3x3<RW, false, IN>(P[2], lx, ly, gx0, gy0, w, h);
        }
    }
    __syncthreads();
    // S4a: second pass of mu on the tile itself; write img and mu
    for (int i = threadIdx.x; i < DT * DT; i += TPB) {
        const int tx = i % DT, ty = i / DT, lx = tx + HL, ly = ty + HL, X = x0 + tx, Y = y0 + ty;
        if (IN || (X < w && Y < h)) {
            const size_t o = (size_t)slot * 3 * g.plane + (size_t)Y * g.pitch + X;
            const int li = ly * RW + lx;
            img[o] = P[0][li];
            img[o + g.plane] = P[1][li];
            img[o + 2 * g.plane] = P[2][li];
            mu[o] = pass3x3<RW, false, IN>(P[3], lx, ly, gx0, gy0, w, h);
            mu[o + g.plane] = pass3x3<RW, false, IN>(P[4], lx, ly, gx0, gy0, w, h);
            mu[o + 2 * g.plane] = pass3x3<RW, false, IN>(P[5], lx, ly, gx0, gy0, w, h);
        }
    }
    __syncthreads();
    // S3b: first pass of sq = blur(img * img): img -> P3,P4,P5
    CE_MARGIN_LOOP(RW, 3, i, lx, ly) {
        if (IN || inside(lx, ly)) {
            P[3][i] = pass3x3<RW, true, IN>(P[0], lx, ly, gx0, gy0, w, h);
            P[4][i] = pass3x3<RW, true, IN>(P[1], lx, ly, gx0, gy0, w, h);
            P[5][i] = pass3x3<RW, true, IN>(P[2], lx, ly, gx0, gy0, w, h);
        }
    }
    __syncthreads();
    // S4b: second pass of sq; write sq
    for (int i = threadIdx.x; i < DT * DT; i += TPB) {
        const int tx = i % DT, ty = i / DT, lx = tx + HL, ly = ty + HL, X = x0 + tx, Y = y0 + ty;
        if (IN || (X < w && Y < h)) {
            const size_t o = (size_t)slot * 3 * g.plane + (size_t)Y * g.pitch + X;
            sq[o] = pass3x3<RW, false, IN>(P[3], lx, ly, gx0, gy0, w, h);
            sq[o + g.plane] = pass3x3<RW, false, IN>(P[4], lx, ly, gx0, gy0, w, h);
            sq[o + 2 * g.plane] = pass3x3<RW, false, IN>(P[5], lx, ly, gx0, gy0, w, h);
        }
    }
}

template <bool FROM_U8>
__global__ __launch_bounds__(TPB) void k_dssim_create(const uint8_t *__restrict__ refs, const uint8_t *__restrict__ tests,
                                                      const float *__restrict__ lut, const float *__restrict__ lin_in,
                                                      float *__restrict__ lin_out, float *__restrict__ img,
                                                      float *__restrict__ rimg, float *__restrict__ rmu,
                                                      float *__restrict__ rsq, lvl_geom g, lvl_geom gn, int has_next,
                                                      size_t img_bytes, uint32_t n_refs_used, uint32_t max_refs, uint32_t z0)
{
    __shared__ float P[6][DR * DR];
    __shared__ float s_lut[256];
    if (FROM_U8) s_lut[threadIdx.x] = lut[threadIdx.x];
    // z0 > 0: the references' planes of this level are cached (reference handle), only the distorted slots are built
    const uint32_t z = blockIdx.z + z0, slot = slot_of(z, n_refs_used, max_refs);
    // a reference's img / mu / sq go to its own per-level buffers (slot z there) so that they survive the level loop
    // and the next launches; a distorted image's img goes to the shared per-launch buffer (slot = pair index)
    const bool is_ref = z < n_refs_used;
    const uint32_t oslot = is_ref ? z : z - n_refs_used;
    const int w = (int)g.w, h = (int)g.h;
    const int x0 = blockIdx.x * DT, y0 = blockIdx.y * DT;
    const uint8_t *src8 = nullptr;
    if (FROM_U8) src8 = z < n_refs_used ? refs + (size_t)z * img_bytes : tests + (size_t)(z - n_refs_used) * img_bytes;
    const float *srcf = lin_in + (size_t)slot * 3 * g.plane;
    __syncthreads();
    auto load_rgb = [&](int X, int Y, float &r, float &gg, float &b) {
        if (FROM_U8) {
            // 32-bit: DSSIM keeps > 300 B per pixel resident, so an image that fits the device is far below 2^32 / 3 pixels
            const uint8_t *px = src8 + ((uint32_t)Y * (uint32_t)w + (uint32_t)X) * 3u;
            r = s_lut[px[0]];
            gg = s_lut[px[1]];
            b = s_lut[px[2]];
        } else {
            const uint32_t o = (uint32_t)Y * g.pitch + (uint32_t)X;
            r = srcf[o];
            gg = srcf[o + g.plane];
            b = srcf[o + 2 * g.plane];
        }
    };
    // next level: (a + b + c + d) * 0.25 over the tile's own 2x2 quads, floor sizes (odd last row/column dropped)
    if (has_next) {
        const int ox = x0 / 2 + (threadIdx.x & 15), oy = y0 / 2 + (threadIdx.x >> 4);
        if (ox < (int)gn.w && oy < (int)gn.h) {
            float q[4][3];
#pragma unroll
            for (int k = 0; k < 4; k++) load_rgb(2 * ox + (k & 1), 2 * oy + (k >> 1), q[k][0], q[k][1], q[k][2]);
#pragma unroll
            for (int c = 0; c < 3; c++)
                lin_out[((size_t)slot * 3 + c) * gn.plane + (size_t)oy * gn.pitch + ox] = (q[0][c] + q[1][c] + q[2][c] + q[3][c]) * 0.25f;
        }
    }
    // S0: L*a*b* of the clamped region (tile + 4 for a reference, tile + 2 for a distorted image), then S1..S4: a block
    // whose region is wholly inside the image takes the variant without clamping or bounds tests
    auto region = [&](auto rw_tag, auto full_tag, float *oimg, float *omu, float *osq) {
        constexpr int RW = decltype(rw_tag)::value;
        constexpr bool FULL = decltype(full_tag)::value;
        constexpr int HL = (RW - DT) / 2;
        const int gx0 = x0 - HL, gy0 = y0 - HL;
        for (int i = threadIdx.x; i < RW * RW; i += TPB) {
            const int lx = i % RW, ly = i / RW;
            const int X = min(max(gx0 + lx, 0), w - 1), Y = min(max(gy0 + ly, 0), h - 1);
            float r, gg, b;
            load_rgb(X, Y, r, gg, b);
            rgb_to_lab(r, gg, b, P[0][i], P[1][i], P[2][i]);
        }
        __syncthreads();
        if (gx0 >= 0 && gy0 >= 0 && gx0 + RW <= w && gy0 + RW <= h)
            dssim_create_stages<true, FULL, RW>(P, oimg, omu, osq, g, oslot, x0, y0);
        else
            dssim_create_stages<false, FULL, RW>(P, oimg, omu, osq, g, oslot, x0, y0);
    };
    if (is_ref)
        region(std::integral_constant<int, DR>{}, std::true_type{}, rimg, rmu, rsq);
    else
        region(std::integral_constant<int, DT + 4>{}, std::false_type{}, img, nullptr, nullptr);
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

// ---- Dssim::compare for one level, fused: i12 = blur(img1*img2) in LDS (tile + halo 2), then compare_scale ----
constexpr int CR = DT + 4;

// (A variant in which a block walks ALL distorted images of one reference, so that the reference's nine planes are read
// from cache after the first, was measured in round 2: 0.85 ms per step against 0.70 ms for this one-pair-per-block
// form on the Kodak grid - the per-block loop with two barriers per distorted image costs more latency than the
// reference re-reads cost bandwidth; with 24 x 16 tiles per image the blocks of consecutive pairs of a reference
// already land on the same XCD, 384 = 0 mod 8.)
// The distorted image's mu = blur(img2) and sq = blur(img2 * img2), and i12 = blur(img1 * img2): three two-pass 3x3 blurs
// of the SAME 36x36 region (tile + halo 2), each the same pass3x3 sequence create_image runs for a reference (same taps,
// same order, same edge replication: bit-identical to Dssim::create_image of the distorted image).  LDS planes: A = img2,
// M = img1 * img2, T = first-pass output; M is reused as first-pass output once its own first pass is done:
//   P1  T = pass(M)                       | barrier
//   P2  s12 = pass(T);  M = pass(A)       | barrier
//   P3  mu2 = pass(M);  T = pass(A^2)     | barrier
//   P4  sq2 = pass(T);  SSIM
// IN = the block's whole 36x36 region is inside the image (no clamping, no bounds tests).
constexpr int CMP_OUT = DT * DT / TPB;  // output pixels per thread (4)
// Left alone, the compiler reads the 27 taps of each of a thread's four pixels in P2 / P3 and SINKS the sums to their use
// in P4, keeping (spilling) the raw taps instead of the 12 results: 290 registers, or 500 B of scratch under a cap.
// CE_KEEP pins a result in a register where it is computed; the memory fence keeps one pixel's taps live at a time.
#define CE_SCHED_FENCE() asm volatile("" ::: "memory")
#define CE_KEEP(x) asm volatile("" : "+v"(x))
struct cmp_stats {
    float u1[CMP_OUT][3], q1[CMP_OUT][3];  // mu and blur(img^2) of the REFERENCE at this thread's pixels
};

template <bool IN>
__device__ __forceinline__ double dssim_compare_stages(float (&A)[3][CR * CR], float (&M)[3][CR * CR], float (&T)[3][CR * CR],
                                                       const float *__restrict__ r_mu, const float *__restrict__ r_sq,
                                                       float *__restrict__ map, const lvl_geom &g, int x0, int y0)
{
    const int w = (int)g.w, h = (int)g.h, gx0 = x0 - 2, gy0 = y0 - 2;
    const uint32_t plane = (uint32_t)g.plane;
    auto inside = [&](int lx, int ly) { return gx0 + lx >= 0 && gx0 + lx < w && gy0 + ly >= 0 && gy0 + ly < h; };
    float s12[CMP_OUT][3], u2[CMP_OUT][3], q2[CMP_OUT][3];
    // P1
    CE_MARGIN_LOOP(CR, 1, i, lx, ly) {
        if (IN || inside(lx, ly)) {
#pragma unroll
            for (int c = 0; c < 3; c++) T[c][i] = pass3x3<CR, false, IN>(M[c], lx, ly, gx0, gy0, w, h);
        }
    }
    __syncthreads();
    // P2
#pragma unroll
    for (int k = 0; k < CMP_OUT; k++) {
        const int i = k * TPB + (int)threadIdx.x, tx = i % DT, ty = i / DT;
        if (IN || (x0 + tx < w && y0 + ty < h)) {
#pragma unroll
            for (int c = 0; c < 3; c++) {
                s12[k][c] = pass3x3<CR, false, IN>(T[c], tx + 2, ty + 2, gx0, gy0, w, h);
                CE_KEEP(s12[k][c]);
            }
        }
        CE_SCHED_FENCE();
    }
    CE_MARGIN_LOOP(CR, 1, i, lx, ly) {
        if (IN || inside(lx, ly)) {
#pragma unroll
            for (int c = 0; c < 3; c++) M[c][i] = pass3x3<CR, false, IN>(A[c], lx, ly, gx0, gy0, w, h);
        }
    }
    __syncthreads();
    // P3
#pragma unroll
    for (int k = 0; k < CMP_OUT; k++) {
        const int i = k * TPB + (int)threadIdx.x, tx = i % DT, ty = i / DT;
        if (IN || (x0 + tx < w && y0 + ty < h)) {
#pragma unroll
            for (int c = 0; c < 3; c++) {
                u2[k][c] = pass3x3<CR, false, IN>(M[c], tx + 2, ty + 2, gx0, gy0, w, h);
                CE_KEEP(u2[k][c]);
            }
        }
        CE_SCHED_FENCE();
    }
    // the reference's mu / blur(img^2) at this thread's pixels: requested here, consumed in P4 behind the last first pass
    // (pixels outside the image are clamped, never used)
    cmp_stats st;
#pragma unroll
    for (int k = 0; k < CMP_OUT; k++) {
        const int i = k * TPB + (int)threadIdx.x;
        const int X = min(x0 + i % DT, w - 1), Y = min(y0 + i / DT, h - 1);
        const uint32_t o = (uint32_t)Y * g.pitch + (uint32_t)X;
#pragma unroll
        for (int c = 0; c < 3; c++) {
            st.u1[k][c] = r_mu[c * plane + o];
            st.q1[k][c] = r_sq[c * plane + o];
        }
    }
    CE_MARGIN_LOOP(CR, 1, i, lx, ly) {
        if (IN || inside(lx, ly)) {
#pragma unroll
            for (int c = 0; c < 3; c++) T[c][i] = pass3x3<CR, true, IN>(A[c], lx, ly, gx0, gy0, w, h);
        }
    }
    __syncthreads();
    // P4
    double val = 0.0;
#pragma unroll
    for (int k = 0; k < CMP_OUT; k++) {
        const int i = k * TPB + (int)threadIdx.x;
        const int tx = i % DT, ty = i / DT, lx = tx + 2, ly = ty + 2, X = x0 + tx, Y = y0 + ty;
        if (IN || (X < w && Y < h)) {
            const uint32_t o = (uint32_t)Y * g.pitch + (uint32_t)X;
            const float c1 = 0.01f * 0.01f, c2 = 0.03f * 0.03f, third = 1.0f / 3.0f;
            float m11[3], m12[3], m22[3], s1[3], s2[3], s12c[3];
#pragma unroll
            for (int c = 0; c < 3; c++) {
                q2[k][c] = pass3x3<CR, false, IN>(T[c], lx, ly, gx0, gy0, w, h);
                const float u1 = st.u1[k][c], u2v = u2[k][c];
                m11[c] = u1 * u1;
                m12[c] = u1 * u2v;
                m22[c] = u2v * u2v;
                s1[c] = st.q1[k][c] - m11[c];
                s2[c] = q2[k][c] - m22[c];
                s12c[c] = s12[k][c] - m12[c];
            }
#define AVG3(v) (((v)[0] + (v)[1] + (v)[2]) * third)
            const float mu1_sq = AVG3(m11), mu2_sq = AVG3(m22), mu1_mu2 = AVG3(m12);
            const float sigma1_sq = AVG3(s1), sigma2_sq = AVG3(s2), sigma12 = AVG3(s12c);
#undef AVG3
            const float ssim = (2.0f * mu1_mu2 + c1) * (2.0f * sigma12 + c2) / ((mu1_sq + mu2_sq + c1) * (sigma1_sq + sigma2_sq + c2));
            map[o] = ssim;
            val += (double)ssim;
        }
        CE_SCHED_FENCE();
    }
    return val;
}

__global__ __launch_bounds__(TPB, 3) void k_dssim_compare(const float *__restrict__ img, const float *__restrict__ rimg,
                                                       const float *__restrict__ rmu, const float *__restrict__ rsq,
                                                       const uint32_t *__restrict__ pair_ref,
                                                       float *__restrict__ map, double *__restrict__ part, lvl_geom g,
                                                       uint32_t level, uint32_t n_levels, uint32_t n_blocks,
                                                       const uint2 *__restrict__ work, uint32_t tiles_x)
{
    __shared__ float A[3][CR * CR], M[3][CR * CR], T[3][CR * CR];
    __shared__ double s_red[TPB / 64];
    // XCD-aware 1-D launch (ce_build_xcd_list): the pairs of a reference run the same tile back to back on one XCD
    const uint2 wi = work[blockIdx.x];
    if (wi.x == ~0u) return;  // padding entry
    const uint32_t p = wi.y;
    const int w = (int)g.w, h = (int)g.h;
    const int x0 = (int)(wi.x % tiles_x) * DT, y0 = (int)(wi.x / tiles_x) * DT, gx0 = x0 - 2, gy0 = y0 - 2;
    // block-uniform bases (scalar registers); everything a thread adds to them fits 32 bits (three planes of ONE image)
    const float *r_img = rimg + (size_t)pair_ref[p] * 3 * g.plane, *r_mu = rmu + (size_t)pair_ref[p] * 3 * g.plane,
                *r_sq = rsq + (size_t)pair_ref[p] * 3 * g.plane, *t_img = img + (size_t)p * 3 * g.plane;
    const uint32_t plane = (uint32_t)g.plane;
    for (int i = threadIdx.x; i < CR * CR; i += TPB) {
        const int lx = i % CR, ly = i / CR;
        const int X = min(max(gx0 + lx, 0), w - 1), Y = min(max(gy0 + ly, 0), h - 1);
        const uint32_t o = (uint32_t)Y * g.pitch + (uint32_t)X;
#pragma unroll
        for (int c = 0; c < 3; c++) {
            const float t = t_img[c * plane + o];
            A[c][i] = t;
            M[c][i] = r_img[c * plane + o] * t;
        }
    }
    __syncthreads();
    const double val = (gx0 >= 0 && gy0 >= 0 && gx0 + CR <= w && gy0 + CR <= h)
                           ? dssim_compare_stages<true>(A, M, T, r_mu, r_sq, map + (size_t)p * g.plane, g, x0, y0)
                           : dssim_compare_stages<false>(A, M, T, r_mu, r_sq, map + (size_t)p * g.plane, g, x0, y0);
    const double t = block_sum(val, s_red);
    if (threadIdx.x == 0) part[(((size_t)p * n_levels + level) * 2 + 0) * n_blocks + wi.x] = t;
}

// ---- the tail of every level in two launches (round 3: one avg + one absdev launch per LEVEL were ten of DSSIM's 21 dependent
// launches, each ~4.5 us on the critical path of a single-pair call) ------------------------------------------------------
struct ds_tail {
    uint32_t w[CE_DSSIM_SCALES], h[CE_DSSIM_SCALES], pitch[CE_DSSIM_SCALES], n_part[CE_DSSIM_SCALES], gx[CE_DSSIM_SCALES];
    uint32_t blk_end[CE_DSSIM_SCALES];  // absdev blocks of levels 0 .. l, running total (the launch's x extent is the last one)
    size_t plane[CE_DSSIM_SCALES], map_off[CE_DSSIM_SCALES];  // the level's SSIM maps start at map + map_off[l], one plane per pair
};

// avg = max(mean, 0)^(0.5^level): one block per (pair, level) reduces the compare kernel's partial sums in a fixed order
__global__ __launch_bounds__(TPB) void k_dssim_avg(const double *__restrict__ part, double *__restrict__ avg_out, ds_tail t, uint32_t n_levels,
                                                   uint32_t n_blocks)
{
    __shared__ double s_red[TPB];
    const uint32_t p = blockIdx.x, level = blockIdx.y;
    const double *pp = part + ((size_t)p * n_levels + level) * 2 * n_blocks;
    const uint32_t used_blocks = t.n_part[level];
    double v = 0.0;
    for (uint32_t k = threadIdx.x; k < used_blocks; k += TPB) v += pp[k];
    s_red[threadIdx.x] = v;
    __syncthreads();
    for (int off = TPB / 2; off > 0; off >>= 1) {
        if ((int)threadIdx.x < off) s_red[threadIdx.x] += s_red[threadIdx.x + off];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        double avg = s_red[0] / (double)((size_t)t.w[level] * t.h[level]);
        if (!(avg > 0.0)) avg = 0.0;
        avg_out[(size_t)p * n_levels + level] = pow(avg, pow(0.5, (double)level));
    }
}

// mean absolute deviation of the SSIM map from avg; 1-D grid: per pair, the blocks of all levels one after the other
constexpr int AD_ROWS = 32;  // rows per block
__global__ __launch_bounds__(TPB) void k_dssim_absdev(const float *__restrict__ map, const double *__restrict__ avg_in,
                                                      double *__restrict__ part, ds_tail t, uint32_t n_levels, uint32_t n_blocks)
{
    __shared__ double s_red[TPB / 64];
    const uint32_t per_pair = t.blk_end[n_levels - 1], p = blockIdx.x / per_pair, bi = blockIdx.x % per_pair;
    uint32_t level = 0, first = 0;
#pragma unroll
    for (int l = 0; l + 1 < CE_DSSIM_SCALES; l++)
        if (l + 1 < (int)n_levels && bi >= t.blk_end[l]) level = l + 1, first = t.blk_end[l];
    const uint32_t w = t.w[level], h = t.h[level], pitch = t.pitch[level], gx = t.gx[level];
    const uint32_t bx = (bi - first) % gx, by = (bi - first) / gx;
    double *pp = part + ((size_t)p * n_levels + level) * 2 * n_blocks;
    const double avg = avg_in[(size_t)p * n_levels + level];
    const float *m = map + t.map_off[level] + (size_t)p * t.plane[level];
    // block = 64 columns x 32 rows, a thread walks 8 rows (stride 4): one block reduction per 2048 pixels
    const uint32_t x = bx * 64 + (threadIdx.x & 63);
    double val = 0.0;
#pragma unroll
    for (int k = 0; k < AD_ROWS / 4; k++) {
        const uint32_t y = by * AD_ROWS + 4 * k + (threadIdx.x >> 6);
        if (x < w && y < h) val += fabs(avg - (double)m[(size_t)y * pitch + x]);
    }
    const double s = block_sum(val, s_red);
    if (threadIdx.x == 0) pp[n_blocks + by * gx + bx] = s;
}

struct ds_geom {
    uint32_t npix[CE_DSSIM_SCALES];
    uint32_t nblk[CE_DSSIM_SCALES];
};

// one wave per pair: the per-block deviation sums of every level are added lane-strided and then across the wave
// (a fixed order, so the result is reproducible), lane 0 forms the level scores and the final value
__global__ __launch_bounds__(64) void k_dssim_finalize_pairs(const double *__restrict__ part, double *__restrict__ level_scores,
                                                             ce_dev_scores *__restrict__ scores, uint32_t n_pairs,
                                                             uint32_t n_levels, uint32_t n_blocks, ds_geom g)
{
    const uint32_t p = blockIdx.x, lane = threadIdx.x;
    if (p >= n_pairs) return;
    const double W[CE_DSSIM_SCALES] = {0.028, 0.197, 0.322, 0.298, 0.155};
    double ssim_sum = 0.0, weight_sum = 0.0;
    for (uint32_t l = 0; l < n_levels; l++) {
        const double *pp = part + (((size_t)p * n_levels + l) * 2 + 1) * n_blocks;
        double dev = 0.0;
        for (uint32_t k = lane; k < g.nblk[l]; k += 64) dev += pp[k];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) dev += __shfl_down(dev, off, 64);
        dev = __shfl(dev, 0, 64);
        const double score = 1.0 - dev / (double)g.npix[l];
        if (lane == 0) level_scores[(size_t)p * CE_DSSIM_SCALES + l] = score;
        ssim_sum += score * W[l];
        weight_sum += W[l];
    }
    double ssim = ssim_sum / weight_sum;
    if (!(ssim > 2.220446049250313e-16)) ssim = 2.220446049250313e-16;
    if (lane == 0) scores[p].dssim = 1.0 / ssim - 1.0;
}

}  // namespace

void ce_dssim_free(ce_batch *b)
{
    for (auto &p : b->ds_lin) hipFree(p), p = nullptr;
    hipFree(b->ds_img); hipFree(b->ds_map);
    for (int l = 0; l < CE_DSSIM_SCALES; l++) {
        hipFree(b->ds_rimg[l]); hipFree(b->ds_rmu[l]); hipFree(b->ds_rsq[l]);
        b->ds_rimg[l] = b->ds_rmu[l] = b->ds_rsq[l] = nullptr;
        ce_free_xcd_list(&b->ds_work[l]);
        hipFree(b->ds_gwork[l].d);
        b->ds_gwork[l] = ce_group_list{};
    }
    b->ds_ref_src = nullptr;
    hipFree(b->ds_part); hipFree(b->ds_level_scores);
    b->ds_img = b->ds_map = nullptr;
    b->ds_part = b->ds_level_scores = nullptr;
    b->dssim_ready = false;
}

static int dssim_allocate(ce_batch *b)
{
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
    // linear RGB ping-pong for levels >= 1 (level 0 is read from the u8 slabs): level l lives in ds_lin[l & 1]
    const size_t p1 = n > 1 ? b->ds[1].plane : 1;
    CE_HIP(ctx, hipMalloc(&b->ds_lin[0], slots * 3 * p1 * sizeof(float)));
    CE_HIP(ctx, hipMalloc(&b->ds_lin[1], slots * 3 * p1 * sizeof(float)));
    CE_HIP(ctx, hipMalloc(&b->ds_img, (size_t)b->max_pairs * 3 * p0 * sizeof(float)));  // the distorted images' img, one level at a time
    // the references' planes, one set per level: they outlive the level loop (Dssim::create_image of the reference is
    // run once per reference, dssim.rs:54-59; a reference handle keeps them across compares)
    for (int l = 0; l < n; l++) {
        const size_t rb = (size_t)b->max_refs * 3 * b->ds[l].plane * sizeof(float);
        CE_HIP(ctx, hipMalloc(&b->ds_rimg[l], rb));
        CE_HIP(ctx, hipMalloc(&b->ds_rmu[l], rb));
        CE_HIP(ctx, hipMalloc(&b->ds_rsq[l], rb));
    }
    size_t map_floats = 0;  // every level's SSIM maps stay until the tail kernels have run
    for (int l = 0; l < n; l++) map_floats += (size_t)b->max_pairs * b->ds[l].plane;
    CE_HIP(ctx, hipMalloc(&b->ds_map, map_floats * sizeof(float)));
    // partial sums per (pair, level): the absdev kernel's blocks, or the compare kernel's strip tiles (>= 2 rows each)
    b->ds_blocks = std::max(((b->ds[0].w + 63) / 64) * ((b->ds[0].h + 3) / 4), ((b->ds[0].w + CE_DSSIM_STRIP - 1) / CE_DSSIM_STRIP) * ((b->ds[0].h + 1) / 2));
    CE_HIP(ctx, hipMalloc(&b->ds_part, (size_t)b->max_pairs * CE_DSSIM_SCALES * 2 * b->ds_blocks * sizeof(double)));
    CE_HIP(ctx, hipMalloc(&b->ds_level_scores, (size_t)b->max_pairs * CE_DSSIM_SCALES * sizeof(double)));  // avg, then score
    return CE_OK;
}

static int dssim_prepare(ce_batch *b)
{
    if (b->dssim_ready) return CE_OK;
    const int rc = dssim_allocate(b);
    if (rc != CE_OK) {  // all or nothing (see ce_ssim2_prepare)
        const std::string why = b->ctx->err;
        ce_dssim_free(b);
        (void)hipGetLastError();
        b->ctx->err = why;
        return rc;
    }
    b->dssim_ready = true;
    return CE_OK;
}

int ce_launch_dssim(ce_batch *b, const uint8_t *d_refs, uint32_t n_refs_used, uint32_t n_pairs)
{
    ce_ctx *ctx = b->ctx;
    int rc = dssim_prepare(b);
    if (rc != CE_OK) return rc;
    const uint32_t n_slots = n_refs_used + n_pairs, mr = b->max_refs;
    // reference handle (ce_ref_*): the references' img / mu / sq pyramid of an earlier launch is still valid
    const bool cached = b->keep_ref_pyramid && b->ds_ref_src == d_refs && b->ds_ref_count >= n_refs_used;
    const uint32_t z0 = cached ? n_refs_used : 0;
    if (!cached) b->ref_builds[1]++;
    static const bool tile_compare = [] {  // CE_DSSIM_COMPARE=tile: round 2's LDS-tile compare kernel (A/B knob)
        const char *e = std::getenv("CE_DSSIM_COMPARE");
        return e && std::string(e) == "tile";
    }();
    static const bool tile_create = [] {  // CE_DSSIM_CREATE=tile: round 2's LDS-tile create_image kernel (A/B knob)
        const char *e = std::getenv("CE_DSSIM_CREATE");
        return e && std::string(e) == "tile";
    }();
    ds_geom g{};
    ds_tail tail{};
    size_t map_off = 0;
    for (int l = 0; l < b->ds_levels; l++) {
        const auto &d = b->ds[l];
        const bool has_next = l + 1 < b->ds_levels;
        const auto &nd = b->ds[has_next ? l + 1 : l];
        const lvl_geom lg{d.w, d.h, d.pitch, d.plane}, ng{nd.w, nd.h, nd.pitch, nd.plane};
        const dim3 tiles((d.w + DT - 1) / DT, (d.h + DT - 1) / DT, 1);
        // create_image for every used slot (references once per reference)
        if (!tile_create) {
            if ((rc = ce_dssim_create_stream(b, l, d_refs, n_refs_used, n_pairs, z0)) != CE_OK) return rc;
        } else if (l == 0)
            CE_LAUNCH(ctx, "dssim_create_u8", k_dssim_create<true>, dim3(tiles.x, tiles.y, n_slots - z0), dim3(TPB), 0, d_refs, b->d_tests,
                      ctx->d_lut_powf, (const float *)nullptr, b->ds_lin[1], b->ds_img, b->ds_rimg[l], b->ds_rmu[l],
                      b->ds_rsq[l], lg, ng, has_next ? 1 : 0, b->img_bytes, n_refs_used, mr, z0);
        else
            CE_LAUNCH(ctx, "dssim_create", k_dssim_create<false>, dim3(tiles.x, tiles.y, n_slots - z0), dim3(TPB), 0, d_refs, b->d_tests,
                      ctx->d_lut_powf, (const float *)b->ds_lin[l & 1], b->ds_lin[(l + 1) & 1], b->ds_img,
                      b->ds_rimg[l], b->ds_rmu[l], b->ds_rsq[l], lg, ng, has_next ? 1 : 0, b->img_bytes, n_refs_used, mr, z0);
        // compare per pair
        uint32_t n_part = 0;
        if (tile_compare) {
            if ((rc = ce_build_xcd_list(b, n_pairs, tiles.x * tiles.y, &b->ds_work[l])) != CE_OK) return rc;
            CE_LAUNCH(ctx, "dssim_compare", k_dssim_compare, dim3(b->ds_work[l].len), dim3(TPB), 0, (const float *)b->ds_img,
                      (const float *)b->ds_rimg[l], (const float *)b->ds_rmu[l], (const float *)b->ds_rsq[l], b->d_pair_ref, b->ds_map + map_off,
                      b->ds_part, lg, (uint32_t)l, (uint32_t)b->ds_levels, b->ds_blocks, (const uint2 *)b->ds_work[l].d, tiles.x);
            n_part = tiles.x * tiles.y;
        } else {
            if ((rc = ce_dssim_compare_stream(b, l, n_pairs, b->ds_map + map_off, &n_part)) != CE_OK) return rc;
        }
        tail.w[l] = d.w, tail.h[l] = d.h, tail.pitch[l] = d.pitch, tail.plane[l] = d.plane, tail.n_part[l] = n_part;
        tail.gx[l] = (d.w + 63) / 64;
        tail.map_off[l] = map_off;
        map_off += (size_t)b->max_pairs * d.plane;
        g.npix[l] = d.w * d.h;
        g.nblk[l] = tail.gx[l] * ((d.h + AD_ROWS - 1) / AD_ROWS);
        tail.blk_end[l] = (l ? tail.blk_end[l - 1] : 0u) + g.nblk[l];
    }
    {
        const uint32_t nl = (uint32_t)b->ds_levels;
        CE_LAUNCH(ctx, "dssim_avg", k_dssim_avg, dim3(n_pairs, nl), dim3(TPB), 0, b->ds_part, b->ds_level_scores, tail, nl, b->ds_blocks);
        const dim3 gp(tail.blk_end[nl - 1] * n_pairs);
        CE_LAUNCH(ctx, "dssim_absdev", k_dssim_absdev, gp, dim3(TPB), 0, (const float *)b->ds_map, b->ds_level_scores, b->ds_part, tail, nl,
                  b->ds_blocks);
    }
    if (b->keep_ref_pyramid && !cached) {
        b->ds_ref_src = d_refs;
        b->ds_ref_count = n_refs_used;
    }
    CE_LAUNCH(ctx, "dssim_finalize", k_dssim_finalize_pairs, dim3(n_pairs), dim3(64), 0, b->ds_part,
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
