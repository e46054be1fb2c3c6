// DSSIM on gfx950, the streaming kernels (see dssim.hip for the metric's structure and the reference lines it replaces:
// /root/reference/src/metrics/dssim.rs:40-71 -> dssim_core::Dssim::compare).
// This file is compiled with -fno-slp-vectorize: packed f32 instructions do not pay on gfx950 (profiles/r02_experiments.md
// section 4) and pairing registers for them costs moves in a kernel that is nothing but register-to-register arithmetic.
// Build with -ffp-contract=off.
#include <algorithm>
#include <cstdlib>
#include <string>
#include <type_traits>

#include "ce_internal.h"
#include "dssim_common.h"

namespace {

// ---- Dssim::compare for one level as a STREAM: no LDS, no barrier --------------------------------------------------
// One wave owns a 64-column strip (60 output columns + halo 2 on either side) of one pair and walks down its rows.
// Every value a 3x3 pass needs from the neighbouring columns comes from the neighbouring LANES (DPP wave shifts, a
// full-rate VALU move), every value it needs from the neighbouring rows from four registers per plane (rows9 below):
// the partial sums of the oracle's order "corners, edges, centre".  Nine planes (img2, img1 * img2, img2^2 of three
// channels) x two passes = 72 window registers; the pass-1 output of a row feeds pass 2 in the same step, so nothing is
// ever staged.  Same taps, same order, same edge replication as dssim.hip's pass3x3: bit-identical planes.
// The waves of a block take CONSECUTIVE entries of a work list in which the distorted images of one reference follow each
// other on the same strip: they request the same reference lines at the same time, so the reference's nine planes come
// from HBM once per reference, not once per pair.
constexpr int CS_OUT = CE_DSSIM_STRIP;  // output columns of a strip
constexpr int CS_WAVES = 4;

__device__ __forceinline__ float lane_left(float v)  // the value of lane - 1
{
    return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x138, 0xf, 0xf, true));
}
__device__ __forceinline__ float lane_right(float v)  // the value of lane + 1
{
    return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x130, 0xf, 0xf, true));
}
__device__ __forceinline__ float ld_at(const float *base, uint32_t byte_off)  // uniform base + 32-bit lane offset: saddr form
{
    return *reinterpret_cast<const float *>(reinterpret_cast<const char *>(base) + byte_off);
}

// One row of the nine planes as a 3x3 pass keeps it.  Everything a row contributes through its neighbouring columns is
// formed when the row ARRIVES (two lane shifts per value): lr = l + r, the corner pair it will contribute as the row above, and e = (c_above + l) + r, the first
// three of the four "edges" terms of the output it will be the middle row of.  The middle row keeps (c, e, lr), the row
// above only lr.  Three such rows rotate BY NAME through the unrolled loop body (no register moves).
template <int N>
struct rowsN {
    float c[N], e[N], lr[N];
};
typedef rowsN<9> rows9;
struct rowin {
    float a[3], r[3];  // the distorted / the reference image's row, one row ahead of pass 1
};

// the row below (N, value v) has arrived; returns the pass's output for the middle row M (oracle order: corners, edges,
// centre).  c_above = the value above v as the pass's input has it (M.c, or v itself where v is the image's first row).
template <int NP>
__device__ __forceinline__ float pass_rows(const rowsN<NP> &P, const rowsN<NP> &M, rowsN<NP> &N, int k, float v, float c_above)
{
    const float K0 = 0.095332f, K1 = 0.118095f, K4 = 0.146293f;
    const float corners = (P.lr[k] + lane_left(v)) + lane_right(v);  // ((tl + tr) + bl) + br
    const float edges = M.e[k] + v;                                  // ((t + l) + r) + b
    N.c[k] = v;
    N.e[k] = (c_above + lane_left(v)) + lane_right(v);
    N.lr[k] = lane_left(v) + lane_right(v);
    return corners * K0 + edges * K1 + M.c[k] * K4;
}

struct cmp_planes {  // wave-uniform plane bases (scalar registers); a lane adds a 32-bit byte offset
    const float *t[3], *r[3], *u[3], *q[3];
};

template <bool EDGE>
__device__ __forceinline__ double dssim_compare_strip(const cmp_planes &pl, float *__restrict__ map, const lvl_geom &g, int xs, int y0, int y1)
{
    const int w = (int)g.w, h = (int)g.h, lane = (int)(threadIdx.x & 63);
    const uint32_t pitch = g.pitch;
    const int X = xs + lane - 2;
    const uint32_t Xc = (uint32_t)min(max(X, 0), w - 1);  // clamped loads: pass 1 sees the replicated edge columns for free
    const bool out_lane = lane >= 2 && lane < 2 + CS_OUT && X < w;
    // pass 2 replicates ITS input's edge columns: the lane left of column 0 / right of column w - 1 takes its neighbour's value
    const bool before_first = X == -1, after_last = X == w;
    auto row_off = [&](int y) { return ((uint32_t)min(max(y, 0), h - 1) * pitch + Xc) * 4u; };
    double val = 0.0;
    // step i: input row i arrives (cur; nxt is requested for step i + 1), pass 1 puts out row t = i - 1, pass 2 row y = i - 2
    // ENDS: the step may be the one that meets the first / last row of the image (first and last loop body only)
    auto step = [&](auto ends, const rows9 &P1, rows9 &M1, rows9 &N1, const rows9 &P2, rows9 &M2, rows9 &N2, const rowin &cur, rowin &nxt, int i)
                    __attribute__((always_inline)) {
        constexpr bool ENDS = decltype(ends)::value;
        const int t = i - 1, y = i - 2;
        float u1[3], q1[3];  // the reference's mu / blur(img^2) of the row pass 2 puts out at the end of this step
        {
            const uint32_t o = row_off(i + 1), os = row_off(y);
#pragma unroll
            for (int c = 0; c < 3; c++) {
                nxt.a[c] = ld_at(pl.t[c], o);
                nxt.r[c] = ld_at(pl.r[c], o);
                u1[c] = ld_at(pl.u[c], os);
                q1[c] = ld_at(pl.q[c], os);
            }
        }
        // channel by channel (the scheduling barrier keeps the compiler from starting all nine planes at once)
        float O[9];
#pragma unroll
        for (int c = 0; c < 3; c++) {
            const float v3[3] = {cur.a[c], cur.r[c] * cur.a[c], cur.a[c] * cur.a[c]};
#pragma unroll
            for (int j = 0; j < 3; j++) {
                const int k = 3 * c + j;
                float T = pass_rows(P1, M1, N1, k, v3[j], M1.c[k]);
                if (EDGE) {
                    const float fl = lane_right(T), fr = lane_left(T);
                    T = before_first ? fl : (after_last ? fr : T);
                }
                // the rows above / below the image replicate the first / last row of pass 2's INPUT (wave-uniform tests)
                float c_above = M2.c[k];
                if (ENDS && t == h) T = M2.c[k];
                if (ENDS && t == 0) c_above = T;
                O[k] = pass_rows(P2, M2, N2, k, T, c_above);
                if (ENDS && t == 0) M2.lr[k] = N2.lr[k];  // row -1 = row 0: the next step's corner pair from above
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if (y >= y0 && y < y1) {
            const float c1 = 0.01f * 0.01f, c2 = 0.03f * 0.03f, third = 1.0f / 3.0f;
            float m11[3], m12[3], m22[3], s1[3], s2[3], s12c[3];
#pragma unroll
            for (int c = 0; c < 3; c++) {
                const float u2v = O[3 * c + 0], s12 = O[3 * c + 1], q2 = O[3 * c + 2];
                m11[c] = u1[c] * u1[c];
                m12[c] = u1[c] * u2v;
                m22[c] = u2v * u2v;
                s1[c] = q1[c] - m11[c];
                s2[c] = q2 - m22[c];
                s12c[c] = s12 - m12[c];
            }
#define AVG3(v) (((v)[0] + (v)[1] + (v)[2]) * third)
            const float mu1_sq = AVG3(m11), mu2_sq = AVG3(m22), mu1_mu2 = AVG3(m12);
            const float sigma1_sq = AVG3(s1), sigma2_sq = AVG3(s2), sigma12 = AVG3(s12c);
#undef AVG3
            const float ssim = (2.0f * mu1_mu2 + c1) * (2.0f * sigma12 + c2) / ((mu1_sq + mu2_sq + c1) * (sigma1_sq + sigma2_sq + c2));
            if (out_lane) {
                map[(uint32_t)y * pitch + (uint32_t)X] = ssim;
                val += (double)ssim;
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    };
    rows9 s0, s1, s2, z0, z1, z2;
#pragma unroll
    for (int k = 0; k < 9; k++) s0.lr[k] = s1.c[k] = s1.e[k] = s1.lr[k] = z0.lr[k] = z1.c[k] = z1.e[k] = z1.lr[k] = 0.f;
    rowin L0, L1, L2;
    // input rows y0 - 2 .. y1 + 1, three per loop body (up to two surplus steps at the end put nothing out).  The top
    // block starts at row -1, so that the step that meets the image's first row (t = 0, i = 1) is in the first body; the
    // one that meets its last row (t = h, i = y1 + 1) is in the last body by construction: the bodies between are plain.
    const int i0 = y0 == 0 ? -1 : y0 - 2, n_bodies = (y1 + 1 - i0) / 3 + 1;
    {
        const uint32_t o = row_off(i0);
#pragma unroll
        for (int c = 0; c < 3; c++) L0.a[c] = ld_at(pl.t[c], o), L0.r[c] = ld_at(pl.r[c], o);
    }
    auto body = [&](auto ends, int i) __attribute__((always_inline)) {
        step(ends, s0, s1, s2, z0, z1, z2, L0, L1, i);
        step(ends, s1, s2, s0, z1, z2, z0, L1, L2, i + 1);
        step(ends, s2, s0, s1, z2, z0, z1, L2, L0, i + 2);
    };
    int i = i0;
    body(std::true_type{}, i);
    i += 3;
#pragma unroll 1
    for (int b = 1; b < n_bodies - 1; b++, i += 3) body(std::false_type{}, i);
    if (n_bodies > 1) body(std::true_type{}, i);
    return val;
}

__global__ __launch_bounds__(CS_WAVES * 64) __attribute__((amdgpu_waves_per_eu(3, 3))) void k_dssim_compare_stream(const float *__restrict__ img, const float *__restrict__ rimg,
                                                                        const float *__restrict__ rmu, const float *__restrict__ rsq,
                                                                        const uint32_t *__restrict__ pair_ref, float *__restrict__ map,
                                                                        double *__restrict__ part, lvl_geom g, uint32_t level,
                                                                        uint32_t n_levels, uint32_t n_blocks,
                                                                        const uint2 *__restrict__ work, uint32_t strips, uint32_t rows)
{
    // wave j of block b takes entry 4 * (b / 8) + j of XCD class b % 8 (the list interleaves the eight classes)
    const uint32_t wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint2 wi = work[((blockIdx.x >> 3) * CS_WAVES + wv) * 8 + (blockIdx.x & 7)];
    const uint32_t tile = __builtin_amdgcn_readfirstlane(wi.x), p = __builtin_amdgcn_readfirstlane(wi.y);
    if (tile == ~0u) return;  // padding entry; no barrier anywhere below: a wave may leave alone
    const uint32_t ref = __builtin_amdgcn_readfirstlane(pair_ref[p]);
    const int xs = (int)(tile % strips) * CS_OUT, y0 = (int)((tile / strips) * rows), y1 = min(y0 + (int)rows, (int)g.h);
    cmp_planes pl;
#pragma unroll
    for (int c = 0; c < 3; c++) {
        pl.t[c] = img + ((size_t)p * 3 + c) * g.plane;
        pl.r[c] = rimg + ((size_t)ref * 3 + c) * g.plane;
        pl.u[c] = rmu + ((size_t)ref * 3 + c) * g.plane;
        pl.q[c] = rsq + ((size_t)ref * 3 + c) * g.plane;
    }
    float *pmap = map + (size_t)p * g.plane;
    double val = (xs == 0 || xs + CS_OUT + 2 >= (int)g.w) ? dssim_compare_strip<true>(pl, pmap, g, xs, y0, y1)
                                                          : dssim_compare_strip<false>(pl, pmap, g, xs, y0, y1);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) val += __shfl_down(val, off, 64);
    if ((threadIdx.x & 63) == 0) part[(((size_t)p * n_levels + level) * 2 + 0) * n_blocks + tile] = val;
}

// ---- Dssim::create_image for one level as a STREAM --------------------------------------------------------------------------
// The same walk for one image slot: the input row arrives as linear RGB (level 0: sRGB u8 through the host-powf table),
// becomes L*a*b*, the chroma planes run through their two-pass pre-blur (rows i - 1, i - 2), and img = (L, a', b') of row
// i - 2 leaves; a REFERENCE slot goes on with mu = blur(img) and sq = blur(img^2) (two more passes on six planes, row i - 4)
// - its strip has halo 4 (56 output columns), a distorted image's halo 2 (60): L*a*b* is converted for 1.29 x / 1.13 x the
// pixels instead of the 1.56 x / 1.27 x of the 32 x 32 LDS tiles.  The next level's linear RGB (2 x 2 average) is formed from
// the row pair in registers.  Every pass after the first reads the OUTPUT of a pass, so the lanes beside the image take their
// neighbour's value and the rows above / below it the first / last row's (replicate that pass's input), as in the compare.
constexpr int CR_HALO_REF = 4, CR_HALO_DIST = 2;

template <bool FROM_U8, bool REF, bool EDGE>
__device__ __forceinline__ void dssim_create_strip(const uint8_t *__restrict__ src8, const float *const (&srcf)[3], const float *s_lut,
                                                   float *const (&lin_out)[3], float *const (&oimg)[3], float *const (&omu)[3],
                                                   float *const (&osq)[3], const lvl_geom &g, const lvl_geom &gn, int has_next, int xs,
                                                   int y0, int y1)
{
    constexpr int H = REF ? CR_HALO_REF : CR_HALO_DIST, OUT = 64 - 2 * H;
    const int w = (int)g.w, h = (int)g.h, lane = (int)(threadIdx.x & 63);
    const uint32_t pitch = g.pitch;
    const int X = xs + lane - H;
    const uint32_t Xc = (uint32_t)min(max(X, 0), w - 1);
    const bool out_lane = lane >= H && lane < H + OUT && X < w;
    const bool before_first = X == -1, after_last = X == w;
    auto edge_fix = [&](float T) {  // a pass output that feeds another pass: replicate its edge columns
        if (!EDGE) return T;
        const float fl = lane_right(T), fr = lane_left(T);
        return before_first ? fl : (after_last ? fr : T);
    };
    struct rgb { float v[3]; };
    // Two stages ahead of the arithmetic: row i + 2 is REQUESTED (raw: the three bytes, or the three floats) while row i + 1
    // goes through the sRGB table (an LDS read that must not wait for the global load in the same step) and row i is converted
    struct raw3 { uint32_t v[3]; };
    auto request = [&](int y, raw3 &o) {
        const uint32_t yc = (uint32_t)min(max(y, 0), h - 1);
        if (FROM_U8) {
            // 32-bit: DSSIM keeps > 300 B per pixel resident, so an image that fits the device is far below 2^32 / 3 pixels
            const uint8_t *px = src8 + (yc * (uint32_t)w + Xc) * 3u;
            o.v[0] = px[0], o.v[1] = px[1], o.v[2] = px[2];
        } else {
            const uint32_t ob = (yc * pitch + Xc) * 4u;
#pragma unroll
            for (int c = 0; c < 3; c++) o.v[c] = __float_as_uint(ld_at(srcf[c], ob));
        }
    };
    auto linear = [&](const raw3 &r, rgb &o) {
#pragma unroll
        for (int c = 0; c < 3; c++) o.v[c] = FROM_U8 ? s_lut[r.v[c]] : __uint_as_float(r.v[c]);
    };
    constexpr int NS = REF ? 6 : 1;  // planes of the mu / sq passes (a distorted image has none; 1 keeps the types well-formed)
    // a pass after the first at input row r: rows outside the image replicate the first / last row of ITS input
    auto later_pass = [&](auto ends, auto &P, auto &M, auto &N, int k, float v, int r) __attribute__((always_inline)) {
        constexpr bool ENDS = decltype(ends)::value;
        float c_above = M.c[k];
        if (ENDS && r == h) v = M.c[k];
        if (ENDS && r == 0) c_above = v;
        const float o = pass_rows(P, M, N, k, v, c_above);
        if (ENDS && r == 0) M.lr[k] = N.lr[k];
        return o;
    };
    // step i: input row i arrives; img leaves for row i - 2, a reference's mu / sq for row i - 4
    auto step = [&](auto ends, const rowsN<2> &Pa, rowsN<2> &Ma, rowsN<2> &Na, const rowsN<2> &Pb, rowsN<2> &Mb, rowsN<2> &Nb,
                    const rowsN<NS> &Pm, rowsN<NS> &Mm, rowsN<NS> &Nm, const rowsN<NS> &Pq, rowsN<NS> &Mq, rowsN<NS> &Nq,
                    const rgb &cur, rgb &nxt, const rgb &prev, const raw3 &arrived, raw3 &requested, float &L_new, const float &L_out,
                    int i) __attribute__((always_inline)) {
        request(i + 2, requested);
        linear(arrived, nxt);
        // next level: (a + b + c + d) * 0.25 over the strip's own 2 x 2 quads, floor sizes (odd last row / column dropped)
        if (has_next && (i & 1) && i >= y0 && i < y1) {
            const int oy = i >> 1;
            float q[3];
#pragma unroll
            for (int c = 0; c < 3; c++) q[c] = (((prev.v[c] + lane_right(prev.v[c])) + cur.v[c]) + lane_right(cur.v[c])) * 0.25f;
            if (out_lane && !(X & 1) && (X >> 1) < (int)gn.w && oy < (int)gn.h) {
                const uint32_t o = (uint32_t)oy * gn.pitch + (uint32_t)(X >> 1);
#pragma unroll
                for (int c = 0; c < 3; c++) lin_out[c][o] = q[c];
            }
        }
        float L, A, B;
        rgb_to_lab(cur.v[0], cur.v[1], cur.v[2], L, A, B);
        L_new = L;
        // chroma pre-blur: pass 1 sees clamped loads (its input's edges are replicated for free), pass 2 the output of pass 1
        const float ab[2] = {A, B};
        float img[3];
        img[0] = L_out;  // L of row i - 2
#pragma unroll
        for (int k = 0; k < 2; k++) {
            const float T = edge_fix(pass_rows(Pa, Ma, Na, k, ab[k], Ma.c[k]));
            img[1 + k] = later_pass(ends, Pb, Mb, Nb, k, T, i - 1);
        }
        __builtin_amdgcn_sched_barrier(0);
        const int j = i - 2;
        if (j >= y0 && j < y1 && out_lane) {
            const uint32_t o = (uint32_t)j * pitch + (uint32_t)X;
#pragma unroll
            for (int c = 0; c < 3; c++) oimg[c][o] = img[c];
        }
        if (REF) {
            float O[6];
#pragma unroll
            for (int c = 0; c < 3; c++) {
                const float v = c == 0 ? img[0] : edge_fix(img[c]);  // L comes from clamped loads: its edges are replicated already
                const float two[2] = {v, v * v};
#pragma unroll
                for (int t = 0; t < 2; t++) {
                    const int k = 2 * c + t;
                    const float T = edge_fix(later_pass(ends, Pm, Mm, Nm, k, two[t], j));
                    O[k] = later_pass(ends, Pq, Mq, Nq, k, T, j - 1);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            const int y = i - 4;
            if (y >= y0 && y < y1 && out_lane) {
                const uint32_t o = (uint32_t)y * pitch + (uint32_t)X;
#pragma unroll
                for (int c = 0; c < 3; c++) omu[c][o] = O[2 * c], osq[c][o] = O[2 * c + 1];
            }
        }
    };
    rowsN<2> a0, a1, a2, b0, b1, b2;
    rowsN<NS> m0, m1, m2, q0, q1, q2;
#pragma unroll
    for (int k = 0; k < 2; k++) a0.lr[k] = a1.c[k] = a1.e[k] = a1.lr[k] = b0.lr[k] = b1.c[k] = b1.e[k] = b1.lr[k] = 0.f;
#pragma unroll
    for (int k = 0; k < NS; k++) m0.lr[k] = m1.c[k] = m1.e[k] = m1.lr[k] = q0.lr[k] = q1.c[k] = q1.e[k] = q1.lr[k] = 0.f;
    rgb R0, R1, R2;
    float l0 = 0.f, l1 = 0.f, l2 = 0.f;  // L of the last rows: written at step i, read at step i + 2
    // input rows y0 - 2H' .. y1 + 2H' - 1 with H' = passes / 2; the top block starts at row -1 (see the compare kernel).  The
    // steps that meet the image's first / last row as some pass's input are i = 1 .. 3 and i = h + 1 .. h + 3: the first
    // two and the last two bodies are the ENDS instantiation.
    const int i0 = y0 == 0 ? -1 : y0 - H, i_last = y1 + H - 1, n_bodies = (i_last - i0) / 3 + 1;
    raw3 W0, W1, W2;
    request(i0, W2);
    request(i0 + 1, W0);
    linear(W2, R0);
    R2 = R0;
    auto body = [&](auto ends, int i) __attribute__((always_inline)) {
        step(ends, a0, a1, a2, b0, b1, b2, m0, m1, m2, q0, q1, q2, R0, R1, R2, W0, W1, l0, l1, i);
        step(ends, a1, a2, a0, b1, b2, b0, m1, m2, m0, q1, q2, q0, R1, R2, R0, W1, W2, l1, l2, i + 1);
        step(ends, a2, a0, a1, b2, b0, b1, m2, m0, m1, q2, q0, q1, R2, R0, R1, W2, W0, l2, l0, i + 2);
    };
    int i = i0, b = 0;
#pragma unroll 1
    for (; b < min(2, n_bodies); b++, i += 3) body(std::true_type{}, i);
#pragma unroll 1
    for (; b < n_bodies - 2; b++, i += 3) body(std::false_type{}, i);
#pragma unroll 1
    for (; b < n_bodies; b++, i += 3) body(std::true_type{}, i);
}

// grid (tile groups of CS_WAVES, image slots z0 ..): a block's waves are neighbouring strip tiles of ONE slot
template <bool FROM_U8>
__global__ __launch_bounds__(CS_WAVES * 64) void k_dssim_create_stream(const uint8_t *__restrict__ refs, const uint8_t *__restrict__ tests,
                                                                       const float *__restrict__ lut, const float *__restrict__ lin_in,
                                                                       float *__restrict__ lin_out, float *__restrict__ img,
                                                                       float *__restrict__ rimg, float *__restrict__ rmu,
                                                                       float *__restrict__ rsq, lvl_geom g, lvl_geom gn, int has_next,
                                                                       size_t img_bytes, uint32_t n_refs_used, uint32_t max_refs, uint32_t z0,
                                                                       uint32_t rows)
{
    __shared__ float s_lut[256];
    if (FROM_U8) {
        s_lut[threadIdx.x] = lut[threadIdx.x];
        __syncthreads();  // the only barrier: before any wave leaves
    }
    const uint32_t wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t z = blockIdx.y + z0;
    const bool is_ref = z < n_refs_used;
    const uint32_t slot = is_ref ? z : max_refs + (z - n_refs_used), oslot = is_ref ? z : z - n_refs_used;
    const uint32_t out_cols = 64 - 2 * (is_ref ? CR_HALO_REF : CR_HALO_DIST), strips = (g.w + out_cols - 1) / out_cols;
    const uint32_t tile = blockIdx.x * CS_WAVES + wv;
    if (tile >= strips * ((g.h + rows - 1) / rows)) return;
    const int xs = (int)((tile % strips) * out_cols), y0 = (int)((tile / strips) * rows), y1 = min(y0 + (int)rows, (int)g.h);
    const uint8_t *src8 = FROM_U8 ? (is_ref ? refs + (size_t)z * img_bytes : tests + (size_t)(z - n_refs_used) * img_bytes) : nullptr;
    const float *srcf[3];
    float *lo[3], *oi[3], *om[3], *oq[3];
#pragma unroll
    for (int c = 0; c < 3; c++) {
        srcf[c] = lin_in + ((size_t)slot * 3 + c) * g.plane;
        lo[c] = lin_out + ((size_t)slot * 3 + c) * gn.plane;
        oi[c] = (is_ref ? rimg : img) + ((size_t)oslot * 3 + c) * g.plane;
        om[c] = rmu + ((size_t)oslot * 3 + c) * g.plane;
        oq[c] = rsq + ((size_t)oslot * 3 + c) * g.plane;
    }
    const bool edge = xs == 0 || xs + (int)out_cols + (is_ref ? CR_HALO_REF : CR_HALO_DIST) >= (int)g.w;
    if (is_ref) {
        if (edge)
            dssim_create_strip<FROM_U8, true, true>(src8, srcf, s_lut, lo, oi, om, oq, g, gn, has_next, xs, y0, y1);
        else
            dssim_create_strip<FROM_U8, true, false>(src8, srcf, s_lut, lo, oi, om, oq, g, gn, has_next, xs, y0, y1);
    } else {
        if (edge)
            dssim_create_strip<FROM_U8, false, true>(src8, srcf, s_lut, lo, oi, om, oq, g, gn, has_next, xs, y0, y1);
        else
            dssim_create_strip<FROM_U8, false, false>(src8, srcf, s_lut, lo, oi, om, oq, g, gn, has_next, xs, y0, y1);
    }
}

// Launch order of k_dssim_compare_stream: entries (strip tile, pair); a block's CS_WAVES waves take consecutive entries of
// one XCD class.  Workgroups reach the 8 XCDs round-robin by launch id; all entries of one (reference, row block) - its
// strips, and on each strip the reference's distorted images one after the other - go to ONE class, so that the
// 128-byte lines two neighbouring strips share (a strip starts two columns left of a multiple of 60) and the reference's
// lines are fetched into one L2 (and mostly by one CU) while they are in use.
int build_stream_list(ce_batch *b, uint32_t n_pairs, uint32_t strips, uint32_t rows, uint32_t h, ce_group_list *L)
{
    if (L->d && L->version == b->pair_ref_version && L->pairs == n_pairs && L->strips == strips && L->rows == rows && L->h == h)
        return CE_OK;
    ce_ctx *ctx = b->ctx;
    std::vector<std::vector<uint32_t>> pairs_of(b->max_refs);
    for (uint32_t p = 0; p < n_pairs; p++) pairs_of[b->h_pair_ref[p]].push_back(p);
    const uint32_t row_blocks = (h + rows - 1) / rows;
    std::vector<uint2> cls[8];
    uint32_t k = 0;
    for (uint32_t r = 0; r < b->max_refs; r++) {
        if (pairs_of[r].empty()) continue;
        for (uint32_t rb = 0; rb < row_blocks; rb++, k++)
            for (uint32_t s = 0; s < strips; s++)
                for (uint32_t p : pairs_of[r]) cls[k & 7].push_back(make_uint2(rb * strips + s, p));
    }
    size_t longest = 0;
    for (auto &v : cls) longest = std::max(longest, v.size());
    longest = (longest + CS_WAVES - 1) / CS_WAVES * CS_WAVES;
    std::vector<uint2> flat(longest * 8, make_uint2(~0u, 0u));
    for (uint32_t x = 0; x < 8; x++)
        for (size_t sl = 0; sl < cls[x].size(); sl++) flat[sl * 8 + x] = cls[x][sl];
    if (flat.size() > L->cap) {
        if (L->d) CE_HIP(ctx, hipFree(L->d));
        L->d = nullptr;
        L->cap = 0;
        CE_HIP(ctx, hipMalloc(&L->d, flat.size() * sizeof(uint2)));
        L->cap = (uint32_t)flat.size();
    }
    if (int rc = ce_upload_table(b, L->d, flat.data(), flat.size() * sizeof(uint2))) return rc;  // `flat` is pageable and goes out of scope
    L->len = (uint32_t)flat.size();
    L->version = b->pair_ref_version;
    L->pairs = n_pairs;
    L->strips = strips;
    L->rows = rows;
    L->h = h;
    return CE_OK;
}

// rows a wave of the streaming kernels walks: as many as still leave every SIMD a few waves (a wave re-reads 4 rows of halo)
uint32_t stream_rows(uint32_t strips, uint32_t h, uint32_t n_images)
{
    static const size_t min_waves = [] {  // CE_STREAM_MIN_WAVES: A/B knob
        const char *e = std::getenv("CE_STREAM_MIN_WAVES");
        return e ? (size_t)std::atol(e) : (size_t)8192;
    }();
    uint32_t rows = 64;
    while (rows > 8 && (size_t)strips * ((h + rows - 1) / rows) * n_images < min_waves) rows /= 2;
    // a launch that cannot fill the chip anyway is bound by the latency of ONE wave's walk (rows + 4 steps of ~370 dependent
    // instructions): shorter walks, more waves
    while (rows > 2 && (size_t)strips * ((h + rows - 1) / rows) * n_images < 1024) rows /= 2;
    return rows;
}

}  // namespace

int ce_dssim_create_stream(ce_batch *b, int l, const uint8_t *d_refs, uint32_t n_refs_used, uint32_t n_pairs, uint32_t z0)
{
    ce_ctx *ctx = b->ctx;
    const auto &d = b->ds[l];
    const bool has_next = l + 1 < b->ds_levels;
    const auto &nd = b->ds[has_next ? l + 1 : l];
    const lvl_geom lg{d.w, d.h, d.pitch, d.plane}, ng{nd.w, nd.h, nd.pitch, nd.plane};
    const uint32_t n_slots = n_refs_used + n_pairs;
    if (n_slots <= z0) return CE_OK;
    const uint32_t strips_ref = (d.w + 64 - 2 * CR_HALO_REF - 1) / (64 - 2 * CR_HALO_REF), strips = (d.w + CS_OUT - 1) / CS_OUT;
    const uint32_t rows = stream_rows(strips, d.h, n_slots - z0);
    const uint32_t tiles = std::max(z0 < n_refs_used ? strips_ref : 0u, strips) * ((d.h + rows - 1) / rows);
#define CE_CREATE_LAUNCH(NAME, U8, GRID, Z0)                                                                                        \
    CE_LAUNCH(ctx, NAME, (k_dssim_create_stream<U8>), GRID, dim3(CS_WAVES * 64), 0, d_refs, (const uint8_t *)b->d_tests,            \
              (const float *)ctx->d_lut_powf, (const float *)(l == 0 ? nullptr : b->ds_lin[l & 1]), b->ds_lin[(l + 1) & 1], b->ds_img,      \
              b->ds_rimg[l], b->ds_rmu[l], b->ds_rsq[l], lg, ng, has_next ? 1 : 0, b->img_bytes, n_refs_used, b->max_refs, Z0, rows)
    const dim3 grid((tiles + CS_WAVES - 1) / CS_WAVES, n_slots - z0);
    if (l == 0) CE_CREATE_LAUNCH("dssim_create_u8", true, grid, z0);
    else CE_CREATE_LAUNCH("dssim_create", false, grid, z0);
#undef CE_CREATE_LAUNCH
    return CE_OK;
}

int ce_dssim_compare_stream(ce_batch *b, int l, uint32_t n_pairs, float *level_map, uint32_t *n_part)
{
    ce_ctx *ctx = b->ctx;
    const auto &d = b->ds[l];
    const lvl_geom lg{d.w, d.h, d.pitch, d.plane};
    const uint32_t strips = (d.w + CS_OUT - 1) / CS_OUT, rows = stream_rows(strips, d.h, n_pairs);
    const int rc = build_stream_list(b, n_pairs, strips, rows, d.h, &b->ds_gwork[l]);
    if (rc != CE_OK) return rc;
    CE_LAUNCH(ctx, "dssim_compare", k_dssim_compare_stream, dim3(b->ds_gwork[l].len / CS_WAVES), dim3(CS_WAVES * 64), 0,
              (const float *)b->ds_img, (const float *)b->ds_rimg[l], (const float *)b->ds_rmu[l], (const float *)b->ds_rsq[l],
              b->d_pair_ref, level_map, b->ds_part, lg, (uint32_t)l, (uint32_t)b->ds_levels, b->ds_blocks,
              (const uint2 *)b->ds_gwork[l].d, strips, rows);
    *n_part = strips * ((d.h + rows - 1) / rows);
    return CE_OK;
}
