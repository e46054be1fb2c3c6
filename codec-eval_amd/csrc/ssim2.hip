// SSIMULACRA2 on gfx950 — replaces fast_ssim2::compute_ssimulacra2 behind
// /root/reference/src/metrics/ssimulacra2.rs:96 and GpuSsim2::compute
// (crates/codec-iter/src/gpu.rs:83-109).
//
// Pipeline per pyramid level (all pairs of the batch in one launch each):
//   linear RGB (level 0: sRGB u8 through a 256-entry LDS table; level s>0: 2x2 box of level s-1)
//   -> positive XYB planes
//   -> row pass of the sigma=1.5 recursive Gaussian over the five streams a, b, a*a, b*b, a*b
//   -> column pass of the same filter fused with the SSIM / edge-difference maps and their
//      mean / 4-norm pooling (nothing but per-block partial sums is written)
//   -> fixed-order reduction, 108-weight polynomial and remap to the score.
//
// The blur is the f32 second-order recursion of the libjxl lineage, executed with the same
// operation order as the CPU restatement (oracle/ssimulacra2.c), so every plane is
// bit-identical to it; the filter's poles sit on the unit circle and its f32 round-off is
// NOT negligible at the score level (DESIGN.md "Why the recursive form is kept").
// Build with -ffp-contract=off: every fused multiply-add below is explicit.
#include <algorithm>
#include <cstdlib>

#include "ce_internal.h"

namespace {

struct rg_consts {
    float mul_in[3];
    float mul_prev[3];
};

constexpr int kColsPerBlock = 64;

// Pyramid levels 1.. share ONE row-pass launch and ONE column-pass launch (kernel instance LEVEL = -1): the
// launch's blockIdx.x runs over the levels' blocks back to back and this table maps a block to its level.
// Five tiny, latency-bound launches per pass become one that fills the machine.
constexpr int kTailLevels = CE_MAX_SCALES - 1;
struct lvl_table {
    uint32_t n, first_level;
    uint32_t blk_end[kTailLevels];  // exclusive prefix ends of the levels' blockIdx.x ranges
    uint32_t w[kTailLevels], h[kTailLevels], pitch[kTailLevels];
    size_t plane[kTailLevels];
    const float *xyb[kTailLevels];
    float *hbuf[kTailLevels];
};

__device__ __forceinline__ uint32_t slot_of(uint32_t z, uint32_t n_refs_used, uint32_t max_refs)
{
    return z < n_refs_used ? z : max_refs + (z - n_refs_used);
}

// ---- linear RGB -> positive XYB -----------------------------------------------------------
// cube root, reference form: bit-trick seed and two f64 Halley steps rounded once to f32 (the msun cbrtf
// scheme); IEEE basic operations only, so host and device agree bit for bit.
__device__ __noinline__ float cbrt_f32_exact(float x)
{
    uint32_t hx = __float_as_uint(x) & 0x7fffffffu;
    if (hx == 0) return x;
    float t;
    if (hx < 0x00800000u) {
        t = __uint_as_float(0x4b800000u) * x;
        t = __uint_as_float((__float_as_uint(t) & 0x7fffffffu) / 3 + 642849266u);
    } else {
        t = __uint_as_float(hx / 3 + 709958130u);
    }
    double T = (double)t, r;
    const double xd = (double)x;
    r = T * T * T;
    T = T * (xd + xd + r) / (xd + r + r);
    r = T * T * T;
    T = T * (xd + xd + r) / (xd + r + r);
    return (float)T;
}

// The same f32 result without the two f64 divisions (they are ~2/3 of the front end's instructions).
// Both this and the reference form compute an f64 value T within 2^-45 (relative) of the true cube root
// and round it once to f32, so they can only differ when T lies within 2^-45 of an f32 rounding boundary.
// The fast form: seed s = exp2(log2(x)/3) (hardware transcendentals, ~2^-21), then with d = (s^3-x)/s^3
//     cbrt(x) = s (1-d)^(1/3) = s (1 - d/3 - d^2/9 - O(d^3)),   |d| < 2^-19, so the cubic term is < 2^-59;
// 1/s^3 comes from v_rcp_f32 plus one f64 Newton step.  If T is closer than 2^-39 (64x the bound) to a
// rounding boundary - about one input in 2^14 - the reference form is evaluated instead.
// tests/test_gpu_parity.py sweeps every positive normal f32 through both forms (ce_debug_cbrt_sweep).
__device__ __forceinline__ float cbrt_f32(float x, uint32_t *slow = nullptr)
{
    if (x >= 1.0e-30f && x <= 1.0e30f) {
        const float s = __builtin_amdgcn_exp2f(__builtin_amdgcn_logf(x) * 0.333333343f);
        const double S = (double)s, xd = (double)x;
        const double s3 = S * S * S;
        const double diff = s3 - xd;
        const double inv0 = (double)__builtin_amdgcn_rcpf((float)s3);
        const double e = __builtin_fma(-s3, inv0, 1.0);
        const double inv = __builtin_fma(inv0, e, inv0);
        const double d = diff * inv;
        const double poly = __builtin_fma(d, 1.0 / 9.0, 1.0 / 3.0) * d;
        const double T = __builtin_fma(-S, poly, S);
        // the low 29 mantissa bits of T are its position between two adjacent f32 values; the boundary is 2^28
        const int32_t low = (int32_t)((uint32_t)__double2loint(T) & 0x1fffffffu) - 0x10000000;
        if ((low < 0 ? -low : low) >= 0x4000) return (float)T;
    }
    if (slow) *slow += 1;
    return cbrt_f32_exact(x);
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

__device__ __forceinline__ void linear_to_xyb_positive(float r, float g, float b, float cbrt_bias, float &X, float &Y,
                                                       float &B)
{
    const float m01 = K_M01, m11 = K_M11, m22 = K_M22;
    float m0 = __builtin_fmaf(K_M00, r, __builtin_fmaf(m01, g, __builtin_fmaf(K_M02, b, K_B0)));
    float m1 = __builtin_fmaf(K_M10, r, __builtin_fmaf(m11, g, __builtin_fmaf(K_M12, b, K_B0)));
    float m2 = __builtin_fmaf(K_M20, r, __builtin_fmaf(K_M21, g, __builtin_fmaf(m22, b, K_B0)));
    m0 = cbrt_f32(m0 < 0.0f ? 0.0f : m0) - cbrt_bias;
    m1 = cbrt_f32(m1 < 0.0f ? 0.0f : m1) - cbrt_bias;
    m2 = cbrt_f32(m2 < 0.0f ? 0.0f : m2) - cbrt_bias;
    X = 0.5f * (m0 - m1);
    Y = 0.5f * (m0 + m1);
    B = m2;
    B = (B - Y) + 0.55f;
    X = __builtin_fmaf(X, 14.0f, 0.42f);
    Y = Y + 0.01f;
}

// ---- per-level front end, one 2x2 quad per thread ------------------------------------------------
// Reads the level's linear RGB (level 0: sRGB u8 through the 256-entry table, no linear plane is ever
// materialised at full resolution), writes the level's positive-XYB planes and the NEXT level's
// linear RGB (2x2 box average, edge-clamped, ceil sizes; summed in the lineage's (y, x) order).
template <bool FROM_U8>
__global__ __launch_bounds__(256) void k_ssim2_prep(const uint8_t *__restrict__ refs, const uint8_t *__restrict__ tests,
                                                    const float *__restrict__ lut, const float *__restrict__ lin_in,
                                                    float *__restrict__ xyb, float *__restrict__ lin_out, uint32_t w,
                                                    uint32_t h, uint32_t pitch, size_t plane, uint32_t opitch,
                                                    size_t oplane, int has_next, size_t img_bytes, uint32_t n_refs_used,
                                                    uint32_t max_refs, uint32_t z0)
{
    __shared__ float s_lut[256];
    if (FROM_U8) {
        s_lut[threadIdx.x] = lut[threadIdx.x];
        __syncthreads();
    }
    const uint32_t z = blockIdx.z + z0, slot = slot_of(z, n_refs_used, max_refs);  // z0 > 0: references are cached
    const uint32_t qx = blockIdx.x * 64 + (threadIdx.x & 63), qy = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (qx >= (w + 1) / 2 || qy >= (h + 1) / 2) return;
    const uint8_t *src8 = nullptr;
    if (FROM_U8) src8 = z < n_refs_used ? refs + (size_t)z * img_bytes : tests + (size_t)(z - n_refs_used) * img_bytes;
    const float *srcf = lin_in + (size_t)slot * 3 * plane;
    const float cbrt_bias = cbrt_f32(K_B0);
    float sum[3] = {0.0f, 0.0f, 0.0f};
    float *xo = xyb + (size_t)slot * 3 * plane;
#pragma unroll
    for (uint32_t dy = 0; dy < 2; dy++) {
        const uint32_t yr = 2 * qy + dy, y = min(yr, h - 1);
        float X[2], Y[2], B[2];
        uint32_t lo = 0, hi = 0;  // the quad row's six bytes (two pixels), whatever their alignment
        if (FROM_U8) {
            // three aligned dwords around them instead of six byte loads (the slabs carry 16 bytes of padding)
            const uintptr_t a = reinterpret_cast<uintptr_t>(src8 + ((size_t)y * w + 2 * qx) * 3);
            const uint32_t *q = reinterpret_cast<const uint32_t *>(a & ~(uintptr_t)3);
            const uint32_t d0 = q[0], d1 = q[1], d2 = q[2], sh = (uint32_t)(a & 3);
            lo = __builtin_amdgcn_alignbyte(d1, d0, sh);
            hi = __builtin_amdgcn_alignbyte(d2, d1, sh);
            if (2 * qx + 1 >= w) {  // odd width: the second pixel is the first one again (clamped)
                hi = (lo >> 8) & 0xffffu;
                lo = (lo & 0x00ffffffu) | (lo << 24);
            }
        }
#pragma unroll
        for (uint32_t dx = 0; dx < 2; dx++) {
            const uint32_t xr = 2 * qx + dx, x = min(xr, w - 1);
            float r, g, bl;
            if (FROM_U8) {
                r = s_lut[dx ? lo >> 24 : lo & 255u];
                g = s_lut[dx ? hi & 255u : (lo >> 8) & 255u];
                bl = s_lut[dx ? (hi >> 8) & 255u : (lo >> 16) & 255u];
            } else {
                const size_t o = (size_t)y * pitch + x;
                r = srcf[o];
                g = srcf[o + plane];
                bl = srcf[o + 2 * plane];
            }
            sum[0] += r;
            sum[1] += g;
            sum[2] += bl;
            linear_to_xyb_positive(r, g, bl, cbrt_bias, X[dx], Y[dx], B[dx]);
        }
        if (yr < h) {
            // the quad's two pixels of a row leave as one 8-byte store per plane (2 qx is even, the pitch a multiple of 32)
            const size_t o = (size_t)y * pitch + 2 * qx;
            if (2 * qx + 1 < w) {
                *reinterpret_cast<float2 *>(xo + o) = make_float2(X[0], X[1]);
                *reinterpret_cast<float2 *>(xo + o + plane) = make_float2(Y[0], Y[1]);
                *reinterpret_cast<float2 *>(xo + o + 2 * plane) = make_float2(B[0], B[1]);
            } else {
                xo[o] = X[0];
                xo[o + plane] = Y[0];
                xo[o + 2 * plane] = B[0];
            }
        }
    }
    if (has_next) {
        float *lo = lin_out + (size_t)slot * 3 * oplane + (size_t)qy * opitch + qx;
        lo[0] = sum[0] * 0.25f;
        lo[oplane] = sum[1] * 0.25f;
        lo[2 * oplane] = sum[2] * 0.25f;
    }
}

// one step of the three second-order sections for one stream; returns the filter output
__device__ __forceinline__ float rg_step(float sum, float (&prev)[3], float (&prev2)[3], const rg_consts &rg)
{
    float o[3];
#pragma unroll
    for (int k = 0; k < 3; k++) {
        float v = sum * rg.mul_in[k];
        v = __builtin_fmaf(-1.0f, prev2[k], v);
        prev2[k] = prev[k];
        v = __builtin_fmaf(rg.mul_prev[k], prev[k], v);
        prev[k] = v;
        o[k] = v;
    }
    return o[0] + o[1] + o[2];
}

// Workgroup barrier that orders LDS traffic only.  __syncthreads() also drains the vector-memory
// queue (s_waitcnt vmcnt(0)), which would serialise the prefetch loads and the fire-and-forget
// row stores with every barrier; here global memory is never exchanged between waves.
__device__ __forceinline__ void lds_barrier()
{
    asm volatile("" ::: "memory");
    __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0): this wave's LDS reads/writes have landed
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

// ---- row pass, LDS-staged line tiles ------------------------------------------------------
// One block = HB_ROWS rows of one (pair, channel) and 256 threads: thread t < 5 * HB_ROWS owns one (stream, row) pair -
// stream t / HB_ROWS of {a, b, a*a, b*b, a*b}, row t % HB_ROWS - and carries that row's 3-section filter state.
// HB_ROWS = 32 since round 3 (rounds 1-2: 48): 37 KB of LDS instead of 56, four resident blocks per CU instead of two, and the
// small levels waste fewer rows (a 16-row level 5 filled a third of a 48-row block) - solo, default sweep: level 0 6.43 ->
// 6.25 ms per step, levels 1-5 2.98 -> 2.39 (24 rows: 6.54 / 2.55, 40 rows: in between); profiles/r03_experiments.md section 20.
// (Five 64-lane waves, one per stream, look more natural, but a 320-thread block with this much LDS gets ONE
// resident block per CU on gfx950 - measured with a spin kernel - and then SIMD0 hosts two of its five waves
// while the other SIMDs idle half the time; 256-thread blocks get two per CU.)
// Columns advance in chunks of 32.  The chunk's a/b values are fetched with coalesced 16-byte row loads into
// registers ahead of the filter, then laid into LDS column-major ([column][row], row stride 49 -> per-lane row
// access without bank conflicts inside a stream).  The LDS input tile holds two 32-column halves (double buffer);
// outputs leave through a second LDS tile as coalesced 16-byte row stores of whole 128-byte lines.
#ifndef CE_HB_ROWS
#define CE_HB_ROWS 32
#endif
constexpr int HB_ROWS = CE_HB_ROWS, HB_CW = 32, HB_LD = HB_ROWS + 1, HB_THREADS = 256;
constexpr int HB_LOADS = 2 * HB_ROWS * 8;                    // float4 loads per chunk (two planes)
constexpr int HB_LSLOTS = (HB_LOADS + 63) / 64;             // load slots per lane of the loader wave
static_assert(HB_ROWS * CE_SSIM2_STREAMS <= HB_THREADS, "one task per thread");
constexpr int HB_HALF = HB_CW * HB_LD;
constexpr int HB_STORES = CE_SSIM2_STREAMS * HB_ROWS * (HB_CW / 4);  // float4 stores per chunk: 1920

// Every stream is in[i] = P[i] * Q[i] with (P, Q) = (a, 1), (b, 1), (a, a), (b, b), (a, b).
// One chunk = 32 filter steps: step e consumes the tile's column e (right tap) and the input 10 steps
// back (left tap, kept in a register delay line - for the product streams the delay line holds the
// PRODUCT, so each product is formed once) and emits one output.  The three second-order sections run
// as one packed pair (sections 0, 1: v_pk_mul / v_pk_add / v_pk_fma, two IEEE operations each) plus one
// scalar section - the same operations as rg_step(), in the same order.
typedef float hb_f2 __attribute__((ext_vector_type(2)));

struct hblur_state {
    hb_f2 p01, q01;  // prev / prev2 of sections 0 and 1
    float p2, q2;    // ... of section 2
    float d[10];     // the last 10 inputs: slot e % 10 holds input e - 10
};

__device__ __forceinline__ void hblur_chunk(const float *__restrict__ p_cur, const float *__restrict__ q_cur, bool plain,
                                            float *__restrict__ so, hblur_state &st, const rg_consts &rg)
{
    const hb_f2 n01 = {rg.mul_in[0], rg.mul_in[1]}, d01 = {rg.mul_prev[0], rg.mul_prev[1]};
    // The chunk's inputs are lifted out of LDS 16 columns at a time BEFORE the steps that use them: left to the
    // compiler, every step's ds_read sits behind the previous step's ds_write (same address space, possible alias)
    // and is followed by s_waitcnt lgkmcnt(0) - 32 exposed LDS round trips per chunk.
#pragma unroll
    for (int half = 0; half < 2; half++) {
        float in[HB_CW / 2];
#pragma unroll
        for (int i = 0; i < HB_CW / 2; i++) in[i] = p_cur[(half * (HB_CW / 2) + i) * HB_LD];
#pragma unroll
        for (int i = 0; i < HB_CW / 2; i++) {
            const float q = q_cur[(half * (HB_CW / 2) + i) * HB_LD];
            in[i] = in[i] * (plain ? 1.0f : q);  // x * 1.0f == x exactly: the plain streams are untouched
        }
#pragma unroll
        for (int i = 0; i < HB_CW / 2; i++) {
            const int e = half * (HB_CW / 2) + i;
            const float sum = st.d[e % 10] + in[i];  // left tap + right tap
            st.d[e % 10] = in[i];
            // rg_step(): v = sum * n2; v = v - prev2; v = fma(d1, prev, v)
            hb_f2 v01 = hb_f2{sum, sum} * n01;
            v01 = v01 - st.q01;
            st.q01 = st.p01;
            v01 = __builtin_elementwise_fma(d01, st.p01, v01);
            st.p01 = v01;
            float v2 = sum * rg.mul_in[2];
            v2 = v2 - st.q2;
            st.q2 = st.p2;
            v2 = __builtin_fmaf(rg.mul_prev[2], st.p2, v2);
            st.p2 = v2;
            so[e * HB_LD] = v01.x + v01.y + v2;
        }
    }
    // 32 steps advance the ring phase by 2: rotate so that the next chunk starts at slot 0 again
    float t[10];
#pragma unroll
    for (int j = 0; j < 10; j++) t[j] = st.d[(j + HB_CW) % 10];
#pragma unroll
    for (int j = 0; j < 10; j++) st.d[j] = t[j];
}

// Plane geometry (all planar f32 buffers): pitch = 32 * ceil(w/32) + 32, so every row segment a block loads or
// stores is in bounds without a branch; rows outside the image are clamped on load (and zeroed in LDS) and
// skipped on store.
//
// Chunk c (0..N, N = ceil(w/32)) consumes input columns [32c-28, 32c+4) and emits output columns
// [32c-32, 32c): the input tile is the one that sits 16 bytes off the 128-byte grid, so that the
// (2.5x larger) output stores are whole aligned 128-byte lines.  Chunk 0 only primes the filter
// (columns < 0 are the zero padding; its outputs n < 0 do not exist).
// Wave v loads rows [24*(v&1), +24) of plane (v>>1); the 1920 float4 of a chunk's five output tiles are stored
// by all 256 threads, consecutive threads writing consecutive 16-byte pieces of a row.
template <int LEVEL>
__global__ __launch_bounds__(HB_THREADS) void k_ssim2_hblur_lds(const float *__restrict__ xyb,
                                                                const uint32_t *__restrict__ pair_ref,
                                                                float *__restrict__ hbuf, uint32_t w, uint32_t h,
                                                                uint32_t pitch, size_t plane, uint32_t max_refs,
                                                                rg_consts rg, lvl_table tab, const uint2 *__restrict__ work,
                                                                const uint32_t *__restrict__ pair_first)
{
    __shared__ float s_in[2][2 * HB_HALF];  // [plane a|b][half][column][row]
    __shared__ float s_out[CE_SSIM2_STREAMS * HB_HALF];
    const uint32_t tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    uint32_t bx = blockIdx.x, c = blockIdx.y, p = blockIdx.z;
    if (work) {  // XCD-aware 1-D launch: the work list says which (block, channel, pair) this id is
        const uint2 wi = work[blockIdx.x];
        if (wi.x == ~0u) return;  // padding entry
        bx = wi.x & 0xffffu, c = wi.x >> 16, p = wi.y;
    }
    if (LEVEL < 0) {  // merged launch: which level does this block belong to?
        uint32_t l = 0;
        while (l + 1 < tab.n && bx >= tab.blk_end[l]) l++;
        bx -= l ? tab.blk_end[l - 1] : 0;
        xyb = tab.xyb[l], hbuf = tab.hbuf[l];
        w = tab.w[l], h = tab.h[l], pitch = tab.pitch[l], plane = tab.plane[l];
    }
    const uint32_t y0 = bx * HB_ROWS;
    const float *ga = xyb + ((size_t)pair_ref[p] * 3 + c) * plane;
    const float *gb = xyb + ((size_t)(max_refs + p) * 3 + c) * plane;
    const int n_chunks = (int)(pitch / HB_CW);  // N + 1
    // Loads and stores are kept in DIFFERENT waves.  vmcnt retires in order, so in a wave that does both a load
    // issued after a burst of row stores cannot be seen complete before those stores are acknowledged by the memory
    // side - and under write pressure that takes far longer than the load itself: the chunk loop ran at the pace of
    // store acknowledgements, whatever the arithmetic cost.  Wave 3 is the only one that loads (768 float4 per
    // chunk, 12 per lane, one chunk ahead in registers) and never stores; waves 0-2 store (1920 float4 per chunk,
    // 10 per thread) and never wait on vmcnt.
    const bool loader = wv == 3;
    // load slot m = 0..11 of lane: j = 64 m + lane -> plane j / 384, row (j % 384) / 8, float4 (j % 8) of the row
    const float *l_src[HB_LSLOTS];  // the slot's row in its plane (row clamped; zeroed in LDS below)
    uint32_t l_dst[HB_LSLOTS];      // LDS float index of the slot (within one half)
    uint32_t l_row[HB_LSLOTS];      // image row of the slot; ~0 for a slot past the end of the list
    const uint32_t lq = lane & 7;
#pragma unroll
    for (int m = 0; m < HB_LSLOTS; m++) {
        const uint32_t j = min(64u * m + lane, (uint32_t)HB_LOADS - 1), pl = j / (HB_ROWS * 8), row = (j % (HB_ROWS * 8)) >> 3;
        l_row[m] = 64u * m + lane < (uint32_t)HB_LOADS ? y0 + row : ~0u;
        l_src[m] = (pl ? gb : ga) + (size_t)min(y0 + row, h - 1) * pitch;
        l_dst[m] = pl * (2 * HB_HALF) + (4 * lq) * HB_LD + row;
    }
    float4 pf[HB_LSLOTS];
    auto load_chunk = [&](int k) {
        if (loader) {
            const int col = max(HB_CW * k - 28 + 4 * (int)lq, 0);  // clamped: always a readable address
#pragma unroll
            for (int m = 0; m < HB_LSLOTS; m++) pf[m] = *reinterpret_cast<const float4 *>(l_src[m] + col);
        }
    };
    auto stash_chunk = [&](int k) {
        if (loader) {
            const int col = HB_CW * k - 28 + 4 * (int)lq;  // outside [0, w) the filter sees zeros
            float *base = &s_in[0][0] + (k & 1) * HB_HALF;
#pragma unroll
            for (int m = 0; m < HB_LSLOTS; m++) {
                if (l_row[m] == ~0u) continue;  // past the end of the slot list (only when 2*HB_ROWS*8 is not a multiple of 64)
                float *dst = base + l_dst[m];
                const bool rv = l_row[m] < h && col >= 0;  // col is a multiple of 4: sign is per float4
                dst[0] = (rv && col < (int)w) ? pf[m].x : 0.0f;
                dst[HB_LD] = (rv && col + 1 < (int)w) ? pf[m].y : 0.0f;
                dst[2 * HB_LD] = (rv && col + 2 < (int)w) ? pf[m].z : 0.0f;
                dst[3 * HB_LD] = (rv && col + 3 < (int)w) ? pf[m].w : 0.0f;
            }
        }
    };

    load_chunk(0);
    stash_chunk(0);
    if (n_chunks > 1) load_chunk(1);
    __syncthreads();

    // filter task of this thread.  The reference-only streams (a, a*a) are the same for every distorted image of a
    // reference: only the reference's first pair (pair_first[p] == p) produces them, the other pairs run the three
    // streams that involve b - the column pass reads streams 0 and 2 from the first pair's planes.
    const bool full = pair_first[p] == p;
    const uint32_t n_jobs = full ? CE_SSIM2_STREAMS : 3;
    const bool worker = tid < n_jobs * HB_ROWS;
    const uint32_t tj = worker ? tid / HB_ROWS : 0, tr = worker ? tid % HB_ROWS : 0;
    const uint32_t ts = full ? tj : (tj == 0 ? 1u : tj + 2u);  // jobs of a later pair: streams 1, 3, 4
    const bool plain = ts < 2;
    const float *sp = &s_in[(ts == 1 || ts == 3) ? 1 : 0][tr];
    const float *sq = &s_in[(ts == 2) ? 0 : 1][tr];  // a*a -> a, b*b and a*b -> b (ignored when plain)
    float *so = &s_out[ts * HB_HALF + tr];
    hblur_state st;
    st.p01 = st.q01 = hb_f2{0.0f, 0.0f};
    st.p2 = st.q2 = 0.0f;
#pragma unroll
    for (int j = 0; j < 10; j++) st.d[j] = 0.0f;  // inputs before column -28 of chunk 0: zero padding
    float *obase = hbuf + ((size_t)p * 3 + c) * CE_SSIM2_STREAMS * plane;

    // Iteration k: (wave 3) lay chunk k+1 into the half nobody reads and request chunk k+2; filter chunk k; barrier;
    // (waves 0-2) store chunk k's outputs; barrier.
    for (int k = 0; k < n_chunks; k++) {
        if (k + 1 < n_chunks) stash_chunk(k + 1);
        if (k + 2 < n_chunks) load_chunk(k + 2);
        const uint32_t oc = (k & 1) * HB_HALF;
        if (worker) hblur_chunk(sp + oc, sq + oc, plain, so, st, rg);
        lds_barrier();  // the five output tiles of chunk k are complete; chunk k+1 is complete in LDS
        if (k > 0 && !loader) {
#pragma unroll
            for (int it = 0; it < (HB_STORES + 191) / 192; it++) {
                const uint32_t idx = it * 192 + tid;
                if (idx >= n_jobs * (HB_ROWS * 8)) break;
                const uint32_t oj = idx / (HB_ROWS * 8), rem = idx % (HB_ROWS * 8), orow = rem >> 3, oq = rem & 7;
                const uint32_t os = full ? oj : (oj == 0 ? 1u : oj + 2u);
                const float *src = &s_out[os * HB_HALF + (4 * oq) * HB_LD + orow];
                const float4 v = make_float4(src[0], src[HB_LD], src[2 * HB_LD], src[3 * HB_LD]);
                if (y0 + orow < h)
                    *reinterpret_cast<float4 *>(obase + (size_t)os * plane + (size_t)(y0 + orow) * pitch + HB_CW * (k - 1) + 4 * oq) = v;
            }
        }
        lds_barrier();  // the output tiles may be overwritten by chunk k+1
    }
}

__device__ __forceinline__ double wave_sum(double v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;
}

// ---- column pass + SSIM/edge maps + pooling: LDS-DMA ring, one wave per 64-column strip -----------------------------
// Thread = column, all five streams (15 independent recurrences per lane).  The five row-blurred
// planes and the two XYB planes arrive by LDS-DMA (global_load_lds_dwordx4: one instruction moves
// 4 rows x 64 columns, no VGPRs) into a wave-private 2-group ring; group g+2 is requested as soon
// as group g has been consumed, so one or two groups (7-14 KB per wave) are always in flight.
// The filter's left tap (10 rows back) comes from a 16-deep register ring; the loop is unrolled
// over 16 rows so every ring index is static.  No block barrier: waves only wait on vmcnt.
constexpr int VB_G = 4;                       // rows per DMA group
constexpr int VB_PLANES = 7;                  // 5 streams + xyb(ref) + xyb(test)
constexpr int VB_SLOT = VB_G * 64;            // floats per plane per group
constexpr int VB_GROUP = VB_PLANES * VB_SLOT; // floats per group

struct vblur_state {
    float prev[CE_SSIM2_STREAMS][3], prev2[CE_SSIM2_STREAMS][3];
    float rr[16][CE_SSIM2_STREAMS];  // the last 16 rows of the five streams: the left tap is 10 rows back
    double acc[6];
};

// The four filter steps of DMA group g (ring phase GG = g mod 4, so every register-ring index is static).
// CHECKED = false is the steady state: all four rows lie in [4, h), so nothing is masked and nothing
// branches; CHECKED = true handles the first group (rows 0..3 only prime the filter) and the tail.
template <int GG, bool CHECKED>
__device__ __forceinline__ void vblur_group(vblur_state &st, const float *__restrict__ slot, int g, uint32_t h,
                                            const rg_consts &rg)
{
    const float C2 = 0.0009f;
    float gacc[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int e = 0; e < VB_G; e++) {
        const uint32_t i = (uint32_t)(4 * g + e);
        constexpr int base = 4 * GG;
        const int re = base + e;  // i mod 16
        float v[CE_SSIM2_STREAMS];
#pragma unroll
        for (int s = 0; s < CE_SSIM2_STREAMS; s++) {
            const float ld = slot[s * VB_SLOT + e * 64];
            const float right = (!CHECKED || i < h) ? ld : 0.0f;
            const float left = st.rr[(re + 6) & 15][s];  // row i-10
            st.rr[re][s] = right;
            v[s] = rg_step(left + right, st.prev[s], st.prev2[s], rg);
        }
        if (!CHECKED || (i >= 4 && i < h + 4)) {
            const float img1 = slot[5 * VB_SLOT + e * 64], img2 = slot[6 * VB_SLOT + e * 64];
            const float mu1 = v[0], mu2 = v[1], s11 = v[2], s22 = v[3], s12 = v[4];
            const float mu11 = mu1 * mu1, mu22 = mu2 * mu2, mu12 = mu1 * mu2;
            const float mu_diff = mu1 - mu2;
            const float num_m = __builtin_fmaf(mu_diff, -mu_diff, 1.0f);
            const float num_s = __builtin_fmaf(2.0f, s12 - mu12, C2);
            const float denom_s = (s11 - mu11) + (s22 - mu22) + C2;
            // The lineage widens here and pools in f64.  On the device the per-pixel terms stay f32: the SSIM
            // ratio keeps the exact f32 divide, 1 - ratio is exact for ratio in [0.5, 2]; the edge ratio is formed
            // as (|e2| - |e1|) * rcp(1 + |e1|) (relative error ~2e-7 instead of an f64 divide).  The six terms are
            // summed in f32 over the 4 rows of a DMA group and only then added to the f64 accumulators.
            float d = 1.0f - (num_m * num_s) / denom_s;
            d = d > 0.0f ? d : 0.0f;
            const float dd = d * d;
            const float e1 = fabsf(img1 - mu1), e2 = fabsf(img2 - mu2);
            const float d1 = (e2 - e1) * __builtin_amdgcn_rcpf(1.0f + e1);
            const float artifact = d1 > 0.0f ? d1 : 0.0f;
            const float detail = d1 < 0.0f ? -d1 : 0.0f;
            const float aa = artifact * artifact, ll = detail * detail;
            gacc[0] += d;
            gacc[1] += dd * dd;
            gacc[2] += artifact;
            gacc[3] += aa * aa;
            gacc[4] += detail;
            gacc[5] += ll * ll;
        }
    }
#pragma unroll
    for (int q = 0; q < 6; q++) st.acc[q] += (double)gacc[q];
}

template <int LEVEL>
__global__ __launch_bounds__(64) void k_ssim2_vblur_dma(const float *__restrict__ hbuf, const float *__restrict__ xyb,
                                                        const uint32_t *__restrict__ pair_ref,
                                                        double *__restrict__ partials, uint32_t w, uint32_t h,
                                                        uint32_t pitch, size_t plane, uint32_t max_refs, uint32_t scale,
                                                        uint32_t max_vblocks, rg_consts rg, lvl_table tab,
                                                        const uint2 *__restrict__ work, const uint32_t *__restrict__ pair_first)
{
    __shared__ __attribute__((aligned(16))) float ring[2 * VB_GROUP];
    uint32_t bx = blockIdx.x, c = blockIdx.y, p = blockIdx.z;
    if (work) {  // XCD-aware 1-D launch
        const uint2 wi = work[blockIdx.x];
        if (wi.x == ~0u) return;
        bx = wi.x & 0xffffu, c = wi.x >> 16, p = wi.y;
    }
    if (LEVEL < 0) {  // merged launch: which level does this block belong to?
        uint32_t l = 0;
        while (l + 1 < tab.n && bx >= tab.blk_end[l]) l++;
        bx -= l ? tab.blk_end[l - 1] : 0;
        xyb = tab.xyb[l], hbuf = tab.hbuf[l];
        w = tab.w[l], h = tab.h[l], pitch = tab.pitch[l], plane = tab.plane[l];
        scale = tab.first_level + l;
    }
    const uint32_t lane = threadIdx.x, x0 = bx * 64;
    const bool active = x0 + lane < w;
    const float *hb = hbuf + ((size_t)p * 3 + c) * CE_SSIM2_STREAMS * plane + x0;
    // streams 0 and 2 (blurred a, a*a) exist once per reference, in the planes of its first pair
    const float *hb_ref = hbuf + ((size_t)pair_first[p] * 3 + c) * CE_SSIM2_STREAMS * plane + x0;
    const float *xa = xyb + ((size_t)pair_ref[p] * 3 + c) * plane + x0;
    const float *xb = xyb + ((size_t)(max_refs + p) * 3 + c) * plane + x0;
    const uint32_t dr = lane >> 4, dc = (lane & 15) * 4;  // DMA: 16 lanes x 16 B per row, 4 rows per instruction

    using gptr = const __attribute__((address_space(1))) void *;
    using lptr = __attribute__((address_space(3))) void *;
    // group g: rows 4g..4g+3 of the five streams and rows 4g-4..4g-1 of the XYB planes (what the
    // steps of group g consume).  Rows outside the image are clamped (their values are never used).
    auto issue_group = [&](int g) {
        float *dst = ring + (g & 1) * VB_GROUP;
        const uint32_t row = min((uint32_t)(4 * g) + dr, h - 1);
        const size_t off = (size_t)row * pitch + dc;
#pragma unroll
        for (int s = 0; s < CE_SSIM2_STREAMS; s++)
            __builtin_amdgcn_global_load_lds((gptr)(((s == 0 || s == 2) ? hb_ref : hb) + (size_t)s * plane + off), (lptr)(dst + s * VB_SLOT), 16, 0, 0);
        const int rx = 4 * g - 4 + (int)dr;
        const size_t offx = (size_t)min((uint32_t)max(rx, 0), h - 1) * pitch + dc;
        __builtin_amdgcn_global_load_lds((gptr)(xa + offx), (lptr)(dst + 5 * VB_SLOT), 16, 0, 0);
        __builtin_amdgcn_global_load_lds((gptr)(xb + offx), (lptr)(dst + 6 * VB_SLOT), 16, 0, 0);
    };

    vblur_state st;
#pragma unroll
    for (int s = 0; s < CE_SSIM2_STREAMS; s++) {
#pragma unroll
        for (int k = 0; k < 3; k++) st.prev[s][k] = st.prev2[s][k] = 0.0f;
#pragma unroll
        for (int e = 0; e < 16; e++) st.rr[e][s] = 0.0f;
    }
#pragma unroll
    for (int q = 0; q < 6; q++) st.acc[q] = 0.0;
    const int n_groups = (int)((h + 4 + VB_G - 1) / VB_G);
    const int full_groups = (int)(h / VB_G);  // groups 1 .. full_groups-1 have all four rows in [4, h)

    issue_group(0);
    issue_group(1);
    // one turn of the loop = 4 groups = 16 rows = one turn of the register ring
#define CE_VGROUP(GG, CHECKED)                                                                         \
    do {                                                                                               \
        /* group g has landed once at most the 7 requests of group g+1 are outstanding */              \
        asm volatile("s_waitcnt vmcnt(7)" ::: "memory");                                               \
        vblur_group<GG, CHECKED>(st, ring + ((GG) & 1) * VB_GROUP + lane, g0 + (GG), h, rg);           \
        /* the slot is free once its LDS reads have returned; refill it with group g+2 */              \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                             \
        issue_group(g0 + (GG) + 2);                                                                    \
    } while (0)
    // three loops so that the steady state is a branch-free body of its own: head (rows 0..15, the first four
    // only prime the filter), interior turns, and the tail that runs the filter 4 rows past the image
    int g0 = 0;
    {
        CE_VGROUP(0, true);
        CE_VGROUP(1, true);
        if (2 < n_groups) CE_VGROUP(2, true);
        if (3 < n_groups) CE_VGROUP(3, true);
        g0 = 4;
    }
    for (; g0 + 3 < full_groups; g0 += 4) {
        CE_VGROUP(0, false);
        CE_VGROUP(1, false);
        CE_VGROUP(2, false);
        CE_VGROUP(3, false);
    }
    for (; g0 < n_groups; g0 += 4) {
        CE_VGROUP(0, true);
        if (g0 + 1 < n_groups) CE_VGROUP(1, true);
        if (g0 + 2 < n_groups) CE_VGROUP(2, true);
        if (g0 + 3 < n_groups) CE_VGROUP(3, true);
    }
#undef CE_VGROUP
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // no DMA may land after the wave has retired
    double *dst = partials + ((((size_t)p * CE_MAX_SCALES + scale) * 3 + c) * max_vblocks + bx) * 6;
#pragma unroll
    for (int q = 0; q < 6; q++) {
        const double sum = wave_sum(active ? st.acc[q] : 0.0);
        if (lane == 0) dst[q] = sum;
    }
}

// debug: every f32 bit pattern in [first, first+count) through both cube-root forms
__global__ __launch_bounds__(256) void k_cbrt_sweep(uint32_t first, uint64_t count, unsigned long long *out)
{
    uint64_t mism = 0, slow = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (uint64_t)gridDim.x * blockDim.x) {
        const float x = __uint_as_float(first + (uint32_t)i);
        uint32_t sl = 0;
        const float a = cbrt_f32(x, &sl), b = cbrt_f32_exact(x);
        mism += __float_as_uint(a) != __float_as_uint(b);
        slow += sl;
    }
    if (mism) atomicAdd(out, (unsigned long long)mism);
    if (slow) atomicAdd(out + 1, (unsigned long long)slow);
}

__constant__ double c_weight[108] = {
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

struct scale_geom {
    uint32_t npix[CE_MAX_SCALES];
    uint32_t nblk[CE_MAX_SCALES];
};

// ---- fixed-order reduction of the block partials, then Msssim::score -------------------
__global__ __launch_bounds__(128) void k_ssim2_finalize(const double *__restrict__ partials, double *__restrict__ avg,
                                                        ce_dev_scores *__restrict__ scores, uint32_t n_scales,
                                                        uint32_t max_vblocks, scale_geom g)
{
    __shared__ double s_avg[CE_MAX_SCALES * 3 * 6];
    const uint32_t p = blockIdx.x, t = threadIdx.x;
    if (t < n_scales * 18) {
        const uint32_t s = t / 18, c = (t / 6) % 3, q = t % 6;
        const double *src = partials + ((((size_t)p * CE_MAX_SCALES + s) * 3 + c) * max_vblocks) * 6 + q;
        double sum = 0.0;
        for (uint32_t k = 0; k < g.nblk[s]; k++) sum += src[(size_t)k * 6];
        const double one_per_pixels = 1.0 / (double)g.npix[s];
        double v = one_per_pixels * sum;
        if (q & 1) v = sqrt(sqrt(v));
        s_avg[t] = v;
        avg[(size_t)p * CE_MAX_SCALES * 18 + t] = v;
    }
    __syncthreads();
    if (t == 0) {
        double ssim = 0.0;
        int i = 0;
        for (uint32_t c = 0; c < 3; c++)
            for (uint32_t s = 0; s < n_scales; s++) {
                const double *a = s_avg + (s * 3 + c) * 6;
                for (int n = 0; n < 2; n++) {
                    ssim = fma(c_weight[i++], fabs(a[0 + n]), ssim);
                    ssim = fma(c_weight[i++], fabs(a[2 + n]), ssim);
                    ssim = fma(c_weight[i++], fabs(a[4 + n]), ssim);
                }
            }
        ssim *= 0.9562382616834844;
        ssim = fma(6.248496625763138e-5 * ssim * ssim, ssim,
                   fma(2.326765642916932, ssim, -0.020884521182843837 * ssim * ssim));
        if (ssim > 0.0)
            ssim = fma(pow(ssim, 0.6276336467831387), -10.0, 100.0);
        else
            ssim = 100.0;
        scores[p].ssimulacra2 = ssim;
    }
}

}  // namespace

// ---- host side ---------------------------------------------------------------------------

void ce_ssim2_free(ce_batch *b)
{
    for (int s = 0; s < CE_MAX_SCALES; s++) {
        hipFree(b->d_lin[s]);
        hipFree(b->d_xyb[s]);
        hipFree(b->d_hbuf[s]);
        b->d_lin[s] = b->d_xyb[s] = b->d_hbuf[s] = nullptr;
        if (b->lvl_stream[s]) hipStreamSynchronize(b->lvl_stream[s]);  // the context's stream: drained, not destroyed
        if (b->ev_prep[s]) hipEventDestroy(b->ev_prep[s]);
        if (b->ev_done[s]) hipEventDestroy(b->ev_done[s]);
        b->lvl_stream[s] = nullptr;
        b->ev_prep[s] = b->ev_done[s] = nullptr;
    }
    hipFree(b->d_partials);
    hipFree(b->d_avg);
    b->d_partials = nullptr;
    b->d_avg = nullptr;
    b->ssim2_ready = false;
}

static int ssim2_allocate(ce_batch *b)
{
    ce_ctx *ctx = b->ctx;
    const int ns = b->n_scales;
    const size_t slots = (size_t)b->max_refs + b->max_pairs;
    for (int s = 1; s < ns; s++)  // level 0 is read straight from the u8 slabs
        CE_HIP(ctx, hipMalloc(&b->d_lin[s], slots * 3 * b->sd[s].plane * sizeof(float)));
    for (int s = 0; s < ns; s++) {
        CE_HIP(ctx, hipMalloc(&b->d_xyb[s], slots * 3 * b->sd[s].plane * sizeof(float)));
        CE_HIP(ctx, hipMalloc(&b->d_hbuf[s], (size_t)b->max_pairs * 3 * CE_SSIM2_STREAMS * b->sd[s].plane * sizeof(float)));
        if (s == 0 && !(b->lvl_stream[0] = ce_ctx_aux_stream(ctx, ce_ctx::AUX_SSIM2_L0))) return CE_ERR_BACKEND;  // level 0's passes; the other levels follow the front end
        CE_HIP(ctx, hipEventCreateWithFlags(&b->ev_prep[s], hipEventDisableTiming));
        CE_HIP(ctx, hipEventCreateWithFlags(&b->ev_done[s], hipEventDisableTiming));
    }
    b->max_vblocks = (b->sd[0].w + kColsPerBlock - 1) / kColsPerBlock;
    CE_HIP(ctx, hipMalloc(&b->d_partials, (size_t)b->max_pairs * CE_MAX_SCALES * 3 * b->max_vblocks * 6 * sizeof(double)));
    CE_HIP(ctx, hipMalloc(&b->d_avg, (size_t)b->max_pairs * CE_MAX_SCALES * 18 * sizeof(double)));
    return CE_OK;
}

int ce_ssim2_prepare(ce_batch *b)
{
    if (b->ssim2_ready) return CE_OK;
    uint32_t w = b->w, h = b->h;
    int ns = 0;
    // The lineage tests the size BEFORE halving (`if w < 8 || h < 8 {break}; if scale > 0
    // {downscale}`), so a level smaller than 8 px exists whenever its parent was >= 8.
    for (int s = 0; s < CE_MAX_SCALES; s++) {
        if (w < 8 || h < 8) break;
        if (s > 0) {
            w = (w + 1) / 2;
            h = (h + 1) / 2;
        }
        ce_scale_dims &d = b->sd[s];
        d.w = w;
        d.h = h;
        d.pitch = ((w + HB_CW - 1) / HB_CW) * HB_CW + HB_CW;
        d.plane = (size_t)d.pitch * (((size_t)h + HB_ROWS - 1) / HB_ROWS * HB_ROWS);
        d.hpitch = d.pitch;
        d.hplane = d.plane;
        ns++;
    }
    b->n_scales = ns;
    if (ns == 0) return CE_OK;
    // all or nothing: a partial working set (an allocation failed half-way) is released, so that a retry
    // after the caller has made room starts clean instead of overwriting live pointers
    const int rc = ssim2_allocate(b);
    if (rc != CE_OK) {
        const std::string why = b->ctx->err;
        ce_ssim2_free(b);
        (void)hipGetLastError();
        b->ctx->err = why;
        return rc;
    }
    b->ssim2_ready = true;
    return CE_OK;
}

// XCD-aware launch order for the level-0 passes.  Workgroups go to the 8 XCDs round-robin by launch id, and each
// XCD has its own L2.  The blocks that read the same rows (row pass) or columns (column pass) of one REFERENCE
// for its different distorted images should therefore carry ids that are congruent mod 8 and close together: the
// reference's planes are then fetched from HBM once per reference, not once per pair.  The list is built on the
// host whenever the pair -> reference table changes: keys (reference, channel, block) are dealt to the 8 classes
// in turn, each key followed by all the pairs of that reference; entry id = slot * 8 + class.
// chunk_pairs > 0 (experiment, CE_SSIM2_L0_CHUNK): the list is cut into segments of whole references holding at most
// that many pairs; `chunks` receives (offset, length) of each segment and every segment is launched on its own
// (row pass then column pass of one segment back to back, so that the intermediate of a segment is still in the
// Infinity Cache when its column pass reads it).
static int build_one_list(ce_batch *b, uint32_t n_pairs, uint32_t n_blocks, uint2 **d_list, uint32_t *len, uint32_t *cap,
                          uint32_t chunk_pairs = 0, std::vector<uint2> *chunks = nullptr)
{
    ce_ctx *ctx = b->ctx;
    std::vector<std::vector<uint32_t>> pairs_of(b->max_refs);
    for (uint32_t p = 0; p < n_pairs; p++) pairs_of[b->h_pair_ref[p]].push_back(p);
    std::vector<uint2> flat;
    if (chunks) chunks->clear();
    uint32_t r = 0;
    while (r < b->max_refs) {
        std::vector<uint2> cls[8];
        uint32_t k = 0, in_chunk = 0;
        for (; r < b->max_refs; r++) {
            if (pairs_of[r].empty()) continue;
            if (chunk_pairs && in_chunk && in_chunk + pairs_of[r].size() > chunk_pairs) break;
            in_chunk += (uint32_t)pairs_of[r].size();
            for (uint32_t c = 0; c < 3; c++)
                for (uint32_t bx = 0; bx < n_blocks; bx++, k++)
                    for (uint32_t p : pairs_of[r]) cls[k & 7].push_back(make_uint2(bx | (c << 16), p));
        }
        size_t longest = 0;
        for (auto &v : cls) longest = std::max(longest, v.size());
        if (longest == 0) continue;
        const size_t base = flat.size();
        flat.resize(base + longest * 8, make_uint2(~0u, 0u));
        for (uint32_t x = 0; x < 8; x++)
            for (size_t sl = 0; sl < cls[x].size(); sl++) flat[base + sl * 8 + x] = cls[x][sl];
        if (chunks) chunks->push_back(make_uint2((uint32_t)base, (uint32_t)(longest * 8)));
    }
    if (flat.size() > *cap) {
        if (*d_list) CE_HIP(ctx, hipFree(*d_list));
        *d_list = nullptr;
        CE_HIP(ctx, hipMalloc(d_list, flat.size() * sizeof(uint2)));
        *cap = (uint32_t)flat.size();
    }
    if (int rc = ce_upload_table(b, *d_list, flat.data(), flat.size() * sizeof(uint2))) return rc;  // `flat` is pageable and goes out of scope
    *len = (uint32_t)flat.size();
    return CE_OK;
}

static uint32_t ssim2_l0_chunk()
{
    static const uint32_t v = [] {
        const char *e = std::getenv("CE_SSIM2_L0_CHUNK");
        return e ? (uint32_t)std::max(0, std::atoi(e)) : 0u;
    }();
    return v;
}

static int build_work_lists(ce_batch *b, uint32_t n_pairs, uint32_t hblk, uint32_t vblk)
{
    if (b->work_version == b->pair_ref_version && b->work_pairs == n_pairs && b->d_work_h) return CE_OK;
    if (hblk > 0xffffu || vblk > 0xffffu) {
        b->ctx->err = "SSIMULACRA2: image too large for the block index of the work list";
        return CE_ERR_INVALID_ARG;
    }
    int rc = build_one_list(b, n_pairs, hblk, &b->d_work_h, &b->work_len_h, &b->work_cap_h, ssim2_l0_chunk(), &b->work_chunks_h);
    if (rc != CE_OK) return rc;
    rc = build_one_list(b, n_pairs, vblk, &b->d_work_v, &b->work_len_v, &b->work_cap_v, ssim2_l0_chunk(), &b->work_chunks_v);
    if (rc != CE_OK) return rc;
    b->work_version = b->pair_ref_version;
    b->work_pairs = n_pairs;
    return CE_OK;
}

// the same for the merged launch of levels 1..: the block index runs over the blocks of all those levels
static int build_tail_lists(ce_batch *b, uint32_t n_pairs, uint32_t hblk, uint32_t vblk)
{
    if (b->work_version_t == b->pair_ref_version && b->work_pairs_t == n_pairs && b->work_blk_ht == hblk && b->work_blk_vt == vblk &&
        b->d_work_ht)
        return CE_OK;
    if (hblk > 0xffffu || vblk > 0xffffu) {
        b->ctx->err = "SSIMULACRA2: image too large for the block index of the work list";
        return CE_ERR_INVALID_ARG;
    }
    int rc = build_one_list(b, n_pairs, hblk, &b->d_work_ht, &b->work_len_ht, &b->work_cap_ht);
    if (rc != CE_OK) return rc;
    rc = build_one_list(b, n_pairs, vblk, &b->d_work_vt, &b->work_len_vt, &b->work_cap_vt);
    if (rc != CE_OK) return rc;
    b->work_version_t = b->pair_ref_version;
    b->work_pairs_t = n_pairs;
    b->work_blk_ht = hblk;
    b->work_blk_vt = vblk;
    return CE_OK;
}

int ce_launch_ssim2(ce_batch *b, const uint8_t *d_refs, uint32_t n_refs_used, uint32_t n_pairs)
{
    ce_ctx *ctx = b->ctx;
    int rc = ce_ssim2_prepare(b);
    if (rc != CE_OK) return rc;
    rg_consts rg;
    ce_ssim2_recursive_gaussian(rg.mul_in, rg.mul_prev);
    const uint32_t n_slots = n_refs_used + n_pairs;
    // Ssimulacra2Reference semantics (crates/codec-iter/src/eval.rs:138-149): a reference handle keeps the
    // references' XYB pyramid between compares, so the front end then only runs over the distorted slots
    const bool cached = b->keep_ref_pyramid && b->ssim2_ref_src == d_refs && b->ssim2_ref_count >= n_refs_used &&
                        b->ssim2_ref_levels == std::min(b->n_scales, b->debug_max_scales);
    const uint32_t z0 = cached ? n_refs_used : 0;
    if (!cached) b->ref_builds[0]++;
    using hblur_fn = void (*)(const float *, const uint32_t *, float *, uint32_t, uint32_t, uint32_t, size_t, uint32_t,
                              rg_consts, lvl_table, const uint2 *, const uint32_t *);
    using vblur_fn = void (*)(const float *, const float *, const uint32_t *, double *, uint32_t, uint32_t, uint32_t,
                              size_t, uint32_t, uint32_t, uint32_t, rg_consts, lvl_table, const uint2 *, const uint32_t *);
    const hblur_fn h_l0 = k_ssim2_hblur_lds<0>, h_tail = k_ssim2_hblur_lds<-1>;
    const vblur_fn v_l0 = k_ssim2_vblur_dma<0>, v_tail = k_ssim2_vblur_dma<-1>;
    scale_geom g{};
    const int levels = std::min(b->n_scales, b->debug_max_scales);
    // Front end on the context's stream, level by level (level s+1 needs level s's linear planes).  Level 0's row
    // and column pass run on their own stream as soon as level 0's planes exist; levels 1.. share ONE row-pass and
    // ONE column-pass launch that simply follow the front end on the context's stream, so they overlap level 0.
    // Two streams per batch, not three: HIP multiplexes streams onto a few hardware queues (GPU_MAX_HW_QUEUES, 4 by
    // default) and a stream that shares a hardware queue with a busy one waits for it - with three streams per batch
    // the second shape bucket's front end sat behind the first bucket's kernels for most of a step.
    // In the serial profiling mode everything stays on one stream so that per-kernel times do not overlap.
    hipStream_t s0 = ctx->prof_serial ? CE_STREAM(ctx) : b->lvl_stream[0];
    hipStream_t s1 = CE_STREAM(ctx);
    lvl_table tab{};
    lvl_table tab_v{};
    for (int s = 0; s < levels; s++) {
        const ce_scale_dims &d = b->sd[s];
        const bool has_next = s + 1 < levels;
        const ce_scale_dims &nd = b->sd[has_next ? s + 1 : s];
        const dim3 quad_grid(((d.w + 1) / 2 + 63) / 64, ((d.h + 1) / 2 + 3) / 4, n_slots - z0);
        if (s == 0)
            CE_LAUNCH(ctx, "ssim2_prep_u8", k_ssim2_prep<true>, quad_grid, dim3(256), 0, d_refs, b->d_tests, ctx->d_lut_ssim2,
                      (const float *)nullptr, b->d_xyb[0], b->d_lin[1], d.w, d.h, d.pitch, d.plane, nd.pitch, nd.plane,
                      has_next ? 1 : 0, b->img_bytes, n_refs_used, b->max_refs, z0);
        else
            CE_LAUNCH(ctx, "ssim2_prep", k_ssim2_prep<false>, quad_grid, dim3(256), 0, d_refs, b->d_tests, ctx->d_lut_ssim2,
                      (const float *)b->d_lin[s], b->d_xyb[s], b->d_lin[has_next ? s + 1 : s], d.w, d.h, d.pitch, d.plane,
                      nd.pitch, nd.plane, has_next ? 1 : 0, b->img_bytes, n_refs_used, b->max_refs, z0);
        const uint32_t hblk = (d.h + HB_ROWS - 1) / HB_ROWS, vblk = (d.w + kColsPerBlock - 1) / kColsPerBlock;
        g.npix[s] = d.w * d.h;
        g.nblk[s] = vblk;
        if (s == 0) {
            if (s0 != CE_STREAM(ctx)) {
                CE_HIP(ctx, hipEventRecord(b->ev_prep[0], CE_STREAM(ctx)));
                CE_HIP(ctx, hipStreamWaitEvent(s0, b->ev_prep[0], 0));
            }
            rc = build_work_lists(b, n_pairs, hblk, vblk);
            if (rc != CE_OK) return rc;
            if (ssim2_l0_chunk() && b->work_chunks_h.size() > 1 && s0 != CE_STREAM(ctx) && b->ev_done[2]) {
                // experiment: segment by segment on two alternating streams (row pass, then column pass of the same segment)
                if (!b->lvl_stream[1] && !(b->lvl_stream[1] = ce_ctx_aux_stream(ctx, ce_ctx::AUX_SSIM2_L0B))) return CE_ERR_BACKEND;
                CE_HIP(ctx, hipStreamWaitEvent(b->lvl_stream[1], b->ev_prep[0], 0));
                for (size_t ck = 0; ck < b->work_chunks_h.size(); ck++) {
                    hipStream_t sc = (ck & 1) ? b->lvl_stream[1] : s0;
                    const uint2 ch = b->work_chunks_h[ck], cv = b->work_chunks_v[ck];
                    CE_LAUNCH_ON(ctx, sc, "ssim2_hblur_L0", h_l0, dim3(ch.y), dim3(HB_THREADS), 0, b->d_xyb[0], b->d_pair_ref,
                                 b->d_hbuf[0], d.w, d.h, d.pitch, d.plane, b->max_refs, rg, tab, (const uint2 *)b->d_work_h + ch.x,
                                 (const uint32_t *)b->d_pair_first);
                    CE_LAUNCH_ON(ctx, sc, "ssim2_vblur_ssim_L0", v_l0, dim3(cv.y), dim3(64), 0, b->d_hbuf[0], b->d_xyb[0],
                                 b->d_pair_ref, b->d_partials, d.w, d.h, d.pitch, d.plane, b->max_refs, 0u, b->max_vblocks, rg, tab,
                                 (const uint2 *)b->d_work_v + cv.x, (const uint32_t *)b->d_pair_first);
                }
                CE_HIP(ctx, hipEventRecord(b->ev_done[2], b->lvl_stream[1]));
                CE_HIP(ctx, hipStreamWaitEvent(s0, b->ev_done[2], 0));
            } else {
            CE_LAUNCH_ON(ctx, s0, "ssim2_hblur_L0", h_l0, dim3(b->work_len_h), dim3(HB_THREADS), 0, b->d_xyb[0], b->d_pair_ref,
                         b->d_hbuf[0], d.w, d.h, d.pitch, d.plane, b->max_refs, rg, tab, (const uint2 *)b->d_work_h,
                         (const uint32_t *)b->d_pair_first);
            CE_LAUNCH_ON(ctx, s0, "ssim2_vblur_ssim_L0", v_l0, dim3(b->work_len_v), dim3(64), 0, b->d_hbuf[0], b->d_xyb[0],
                         b->d_pair_ref, b->d_partials, d.w, d.h, d.pitch, d.plane, b->max_refs, 0u, b->max_vblocks, rg, tab,
                         (const uint2 *)b->d_work_v, (const uint32_t *)b->d_pair_first);
            }
            if (s0 != CE_STREAM(ctx)) CE_HIP(ctx, hipEventRecord(b->ev_done[0], s0));
        } else {
            const uint32_t l = tab.n++;
            tab_v.n = tab.n;
            tab.first_level = tab_v.first_level = 1;
            tab.blk_end[l] = (l ? tab.blk_end[l - 1] : 0) + hblk;
            tab_v.blk_end[l] = (l ? tab_v.blk_end[l - 1] : 0) + vblk;
            tab.w[l] = tab_v.w[l] = d.w;
            tab.h[l] = tab_v.h[l] = d.h;
            tab.pitch[l] = tab_v.pitch[l] = d.pitch;
            tab.plane[l] = tab_v.plane[l] = d.plane;
            tab.xyb[l] = tab_v.xyb[l] = b->d_xyb[s];
            tab.hbuf[l] = tab_v.hbuf[l] = b->d_hbuf[s];
        }
    }
    if (tab.n) {
        if (s1 != CE_STREAM(ctx)) {
            CE_HIP(ctx, hipEventRecord(b->ev_prep[1], CE_STREAM(ctx)));
            CE_HIP(ctx, hipStreamWaitEvent(s1, b->ev_prep[1], 0));
        }
        rc = build_tail_lists(b, n_pairs, tab.blk_end[tab.n - 1], tab_v.blk_end[tab_v.n - 1]);
        if (rc != CE_OK) return rc;
        CE_LAUNCH_ON(ctx, s1, "ssim2_hblur_L1-5", h_tail, dim3(b->work_len_ht), dim3(HB_THREADS), 0,
                     (const float *)nullptr, b->d_pair_ref, (float *)nullptr, 0u, 0u, 0u, (size_t)0, b->max_refs, rg, tab,
                     (const uint2 *)b->d_work_ht, (const uint32_t *)b->d_pair_first);
        CE_LAUNCH_ON(ctx, s1, "ssim2_vblur_ssim_L1-5", v_tail, dim3(b->work_len_vt), dim3(64), 0,
                     (const float *)nullptr, (const float *)nullptr, b->d_pair_ref, b->d_partials, 0u, 0u, 0u, (size_t)0, b->max_refs,
                     0u, b->max_vblocks, rg, tab_v, (const uint2 *)b->d_work_vt, (const uint32_t *)b->d_pair_first);
        if (s1 != CE_STREAM(ctx)) CE_HIP(ctx, hipEventRecord(b->ev_done[1], s1));
    }
    if (s0 != CE_STREAM(ctx)) {
        CE_HIP(ctx, hipStreamWaitEvent(CE_STREAM(ctx), b->ev_done[0], 0));
        if (tab.n && s1 != CE_STREAM(ctx)) CE_HIP(ctx, hipStreamWaitEvent(CE_STREAM(ctx), b->ev_done[1], 0));
    }
    if (b->keep_ref_pyramid && !cached) {
        b->ssim2_ref_src = d_refs;
        b->ssim2_ref_count = n_refs_used;
        b->ssim2_ref_levels = levels;
    }
    CE_LAUNCH(ctx, "ssim2_finalize", k_ssim2_finalize, dim3(n_pairs), dim3(128), 0, b->d_partials, b->d_avg,
              b->d_scores, (uint32_t)levels, b->max_vblocks, g);
    CE_HIP(ctx, hipGetLastError());
    return CE_OK;
}

int ce_ssim2_cbrt_sweep(ce_ctx *ctx, uint32_t first_bits, uint64_t count, uint64_t *mismatches, uint64_t *slow_path)
{
    unsigned long long *d = nullptr, h[2] = {0, 0};
    CE_HIP(ctx, hipMalloc(&d, sizeof(h)));
    CE_HIP(ctx, hipMemsetAsync(d, 0, sizeof(h), ctx->stream));
    CE_LAUNCH(ctx, "cbrt_sweep", k_cbrt_sweep, dim3(4096), dim3(256), 0, first_bits, count, d);
    CE_HIP(ctx, hipMemcpyAsync(h, d, sizeof(h), hipMemcpyDeviceToHost, ctx->stream));
    CE_HIP(ctx, hipStreamSynchronize(ctx->stream));
    CE_HIP(ctx, hipFree(d));
    if (mismatches) *mismatches = h[0];
    if (slow_path) *slow_path = h[1];
    return CE_OK;
}

// debug: resident blocks per CU the runtime computes for the two blur kernels (which: 0 = row pass, 1 = column pass)
int ce_ssim2_occupancy(int which)
{
    int n = -1;
    if (which == 0)
        hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k_ssim2_hblur_lds<0>, HB_THREADS, 0);
    else
        hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k_ssim2_vblur_dma<0>, 64, 0);
    return n;
}
