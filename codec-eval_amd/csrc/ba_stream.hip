// Butteraugli's LF stage (blur sigma 7.156, 33 taps), column pass + split, as a STREAM - round 3.
// Replaces k_ba_blur_v_split<33, EPI_LF> (butteraugli.hip) behind butteraugli::butteraugli(..).score
// (/root/reference/src/metrics/butteraugli.rs:72-80); the arithmetic is oracle/butteraugli.c conv_line_renorm's column pass
// and separate_frequencies' LF split, operation for operation.
//
// The tile kernel forms every output from its own 33 products: 33 multiplies + 33 adds.  The kernel is symmetric BIT FOR BIT
// (make_kernel: exp(scaler * i * i) for -i and +i; checked on the host), so the product in[j] * k[16 + e] that output row
// j + e adds as tap 16 - e is the very float that output row j - e adds as tap 16 + e.  A wave that walks DOWN a 64-column
// strip therefore multiplies each arriving row by the 17 distinct weights ONCE and adds the products into the 33 output rows
// in flight - an accumulator each, rotating by name through 33 instantiations of the row step.  Every output still receives
// its taps in ascending row order starting from 0.0f (ConvolutionWithTranspose's sum exactly), with 17 multiplies + 33 adds
// instead of 33 + 33, no LDS and no barrier.  Rows outside the image contribute x = 0 (sum + 0.0f * k == sum); an output row
// is scaled by 1 / (sum of its valid weights), read from a per-row table the host sums in tap order.
//
// A wave owns rows [ya, yb) of its strip and reads 16 rows of halo on either side; the launcher cuts a column into segments
// only when whole columns would not fill the chip (a segment re-forms the products of its 32 halo rows).
//
// Compiled with -fno-slp-vectorize (Makefile): SLP pairs the accumulator adds into v_pk_add_f32 and pays for the pairing
// with register moves (1 477 v_mov in the first build of this kernel).
#include <algorithm>
#include <cstdlib>
#include <type_traits>
#include <utility>

#include "ba_common.h"

namespace {

using namespace ce_ba;

constexpr int LF_LEN = 33, LF_OFF = 16;

template <class F, int... S>
__device__ __forceinline__ void for_each_phase(F &&f, std::integer_sequence<int, S...>)
{
    (f(std::integral_constant<int, S>{}), ...);
}

// Planes [Q0, Q0 + NQ) of one strip segment.  The three planes are split over TWO kinds of waves - X alone, Y and B together
// (XybLowFreqToVals needs lf[Y] for lf[B]) - because the 33 instantiated phases of all three planes are ~50 KB of straight-line
// code, more than the instruction cache keeps: the first build of this kernel (one wave = three planes) ran at 0.55 VALU busy,
// slower than the tile kernel it replaces (profiles/r03_experiments.md section 18).
template <int Q0, int NQ>
__device__ __forceinline__ void lf_stream_body(const float *__restrict__ tmp, const float *__restrict__ xyb, float *__restrict__ psy,
                                               const geom &g, const blur_kernel &bk, const float *__restrict__ row_scale, uint32_t slot,
                                               float *__restrict__ aux_out, int ya, int yb, uint32_t xa, bool live)
{
    constexpr int LEN = LF_LEN, off = LF_OFF;
    const int h = (int)g.h;
    const size_t pl = g.plane;
    // wave-uniform plane bases; a lane adds only its 32-bit column (scalar base + vector offset addressing)
    const float *t0 = tmp + ((size_t)slot * 3 + Q0) * pl;
    const float *xs0 = xyb + ((size_t)slot * 3 + Q0) * pl;
    float *ao0 = aux_out + ((size_t)slot * 3 + Q0) * pl;
    float *ps0 = psy + (size_t)slot * PSY * pl;
    // row j of the row-blurred planes; zero outside the image and beyond the segment's reach (wave-uniform test)
    auto load_row = [&](int j, float (&v)[NQ]) __attribute__((always_inline)) {
        const bool in = j >= 0 && j < h && j < yb + off;
        const float *rp = t0 + (size_t)(in ? j : 0) * g.pitch;
#pragma unroll
        for (int q = 0; q < NQ; q++) {
            const float a = (rp + q * pl)[xa];
            v[q] = in ? a : 0.0f;
        }
    };
    // the unblurred XYB of output row y (clamped: rows outside [ya, yb) are never stored)
    auto load_xs = [&](int y, float (&v)[NQ]) __attribute__((always_inline)) {
        const float *rp = xs0 + (size_t)min(max(y, 0), h - 1) * g.pitch;
#pragma unroll
        for (int q = 0; q < NQ; q++) v[q] = (rp + q * pl)[xa];
    };
    float acc[NQ][LEN];
#pragma unroll
    for (int q = 0; q < NQ; q++)
#pragma unroll
        for (int a = 0; a < LEN; a++) acc[q][a] = 0.0f;
    // three rows of input in flight (ring slot = phase % 3; 33 is a multiple of 3, so the slots keep their names), the XYB of
    // the next two output rows
    float ring[3][NQ], xs_ring[3][NQ];
    const int j_first = ya - off;
    load_row(j_first, ring[0]);
    load_row(j_first + 1, ring[1]);
    load_row(j_first + 2, ring[2]);
    load_xs(j_first - off, xs_ring[0]);
    load_xs(j_first - off + 1, xs_ring[1]);
    load_xs(j_first - off + 2, xs_ring[2]);
    int jb = j_first;
    // one row = one phase; the 33 phases are 33 INSTANTIATIONS (a fold over an integer sequence, not `#pragma unroll`: the body
    // is past the pragma's size limit, and a partly unrolled loop would index the accumulators dynamically, i.e. from scratch)
    auto phase = [&](auto sc) __attribute__((always_inline)) {
        constexpr int s = decltype(sc)::value;
        const int j = jb + s, y = j - off;
        float xv[NQ], xs[NQ];
#pragma unroll
        for (int q = 0; q < NQ; q++) xv[q] = ring[s % 3][q], xs[q] = xs_ring[s % 3][q];
        const float scale = row_scale[min(max(y, 0), h - 1)];  // scalar load, used after the phase's arithmetic
        load_row(j + 3, ring[s % 3]);
        load_xs(y + 3, xs_ring[s % 3]);
#pragma unroll
        for (int q = 0; q < NQ; q++) {
            __builtin_amdgcn_sched_barrier(0);  // one plane's 17 products at a time
            float p[off + 1];
#pragma unroll
            for (int e = 0; e <= off; e++) p[e] = xv[q] * bk.k[off + e];
#pragma unroll
            for (int d = -off; d <= off; d++) {  // output row j + d takes tap off - d of this row: weight k[off - d] == k[off + |d|]
                const int a = (s + d + off + 1) % LEN;
                const float t = p[d < 0 ? -d : d];
                acc[q][a] = d == off ? 0.0f + t : acc[q][a] + t;  // the row's first tap opens the accumulator its last output left
            }
        }
        // Every accumulator is MATERIALISED at the end of its phase.  Without this the optimiser sinks an output's whole chain
        // of adds (and their multiplies) into the conditional block that finally stores it - each output then re-forms its own
        // 33 products from 33 live input rows, which is the tile kernel's arithmetic with 1 000 spilled registers on top.
#pragma unroll
        for (int q = 0; q < NQ; q++)
#pragma unroll
            for (int a = 0; a < LEN; a += 11)
                asm volatile("" : "+v"(acc[q][a]), "+v"(acc[q][a + 1]), "+v"(acc[q][a + 2]), "+v"(acc[q][a + 3]), "+v"(acc[q][a + 4]),
                                  "+v"(acc[q][a + 5]), "+v"(acc[q][a + 6]), "+v"(acc[q][a + 7]), "+v"(acc[q][a + 8]), "+v"(acc[q][a + 9]),
                                  "+v"(acc[q][a + 10]));
        __builtin_amdgcn_sched_barrier(0);
        if (y >= ya && y < yb) {  // wave-uniform: row y has just received its last tap
            float lf[NQ];
#pragma unroll
            for (int q = 0; q < NQ; q++) lf[q] = acc[q][(s + 1) % LEN] * scale;
            if (live) {
                const size_t o = (size_t)y * g.pitch;
                float *ap = ao0 + o, *pp = ps0 + o;
#pragma unroll
                for (int q = 0; q < NQ; q++) (ap + q * pl)[xa] = xs[q] - lf[q];  // raw MF, read with a halo by the MF stage
                // XybLowFreqToVals
                const float xmul = 33.832837186260f, ymul = 14.458268100570f, bmul = 49.87984651440f, y_to_b_mul = -0.362267051518f;
                if (Q0 == 0) {
                    (pp + (size_t)LF0 * pl)[xa] = lf[0] * xmul;
                } else {
                    const float bb = __builtin_fmaf(y_to_b_mul, lf[0], lf[NQ - 1]);
                    (pp + (size_t)LF2 * pl)[xa] = bb * bmul;
                    (pp + (size_t)LF1 * pl)[xa] = lf[0] * ymul;
                }
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    };
#pragma unroll 1
    for (; jb < yb + off; jb += LEN) for_each_phase(phase, std::make_integer_sequence<int, LEN>{});
}

__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(4, 4))) void k_ba_blur_v_lf_stream(
    const float *__restrict__ tmp, const float *__restrict__ xyb, float *__restrict__ psy, geom g, blur_kernel bk,
    const float *__restrict__ row_scale, uint32_t n_refs_used, uint32_t max_refs, uint32_t z0, float *__restrict__ aux_out,
    uint32_t seg_rows)
{
    const uint32_t slot = slot_of((blockIdx.z >> 1) + z0, n_refs_used, max_refs);
    const uint32_t x = blockIdx.x * 64 + threadIdx.x;
    const bool live = x < g.w;
    const int ya = (int)(blockIdx.y * seg_rows), yb = min(ya + (int)seg_rows, (int)g.h);
    if (ya >= (int)g.h) return;
    const uint32_t xa = live ? x : 0u;  // lanes right of the image walk column 0 and store nothing
    if (blockIdx.z & 1)
        lf_stream_body<1, 2>(tmp, xyb, psy, g, bk, row_scale, slot, aux_out, ya, yb, xa, live);
    else
        lf_stream_body<0, 1>(tmp, xyb, psy, g, bk, row_scale, slot, aux_out, ya, yb, xa, live);
}

}  // namespace

namespace ce_ba {

int ce_ba_launch_v_lf_stream(ce_ctx *ctx, hipStream_t stream, const float *tmp, const float *xyb, float *psy, const geom &g,
                             const blur_kernel &bk, const float *row_scale, uint32_t n_refs_used, uint32_t max_refs, uint32_t z0,
                             float *aux_out, uint32_t nz)
{
    if (bk.len != LF_LEN) {
        ctx->err = "butteraugli: the streaming LF pass is built for the 33-tap kernel";
        return CE_ERR_BACKEND;
    }
    for (int e = 1; e <= LF_OFF; e++)
        if (bk.k[LF_OFF - e] != bk.k[LF_OFF + e]) {  // the shared products rely on it, bit for bit
            ctx->err = "butteraugli: the LF kernel is not symmetric";
            return CE_ERR_BACKEND;
        }
    // A wave walks a whole column of its strip when that still gives the chip ~6 waves per SIMD (CE_BA_LF_WAVES); otherwise
    // the column is cut into segments of a multiple of 32 rows.
    static const uint32_t want_waves = [] {
        const char *e = std::getenv("CE_BA_LF_WAVES");
        return (uint32_t)(e ? std::max(1, std::atoi(e)) : 6144);
    }();
    const uint32_t strips = (g.w + 63) / 64;
    uint32_t segs = std::max<uint32_t>(1, (want_waves + 2 * strips * nz - 1) / (2 * strips * nz));
    uint32_t seg_rows = std::max<uint32_t>(32, (g.h + segs - 1) / segs);
    seg_rows = std::min<uint32_t>(g.h, (seg_rows + 31) / 32 * 32);
    segs = (g.h + seg_rows - 1) / seg_rows;
    CE_LAUNCH_ON(ctx, stream, "ba_blur_v_lf", k_ba_blur_v_lf_stream, dim3(strips, segs, 2 * nz), dim3(64), 0, tmp, xyb, psy, g, bk,
                 row_scale, n_refs_used, max_refs, z0, aux_out, seg_rows);
    return CE_OK;
}

}  // namespace ce_ba
