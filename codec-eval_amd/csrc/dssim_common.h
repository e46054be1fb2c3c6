// Shared by dssim.hip (tile kernels, launch sequence) and dssim_stream.hip (streaming kernels).  Not part of the ABI.
#pragma once

#include "ce_internal.h"

struct lvl_geom {
    uint32_t w, h, pitch;
    size_t plane;
};

#if defined(__HIPCC__)
namespace {
// ---- linear RGB -> normalised (L, a, b) ------------------------------------------------------------
__device__ __forceinline__ float cbrt_poly(float x)
{
    // x in (216/24389, ~1], y in [0.2, 1.1]: numerators and denominators in (0.005, 3.5) - the two IEEE quotients take the
    // expansion without range scaling (ce_internal.h: 8 instructions instead of 11, bit for bit; this kernel is VALU-bound)
    float y = (-0.5f * x + 1.51f) * x + 0.2f;
    float y3 = y * y * y;
    y = ce_div_noscale(y * (y3 + 2.0f * x), 2.0f * y3 + x);
    y3 = y * y * y;
    y = ce_div_noscale(y * (y3 + 2.0f * x), 2.0f * y3 + x);
    return y;
}

__device__ __forceinline__ void rgb_to_lab(float r, float g, float b, float &L, float &A, float &B)
{
    const float D65X = 0.9505f, D65Y = 1.0f, D65Z = 1.089f;
    const float EPS = 216.0f / 24389.0f, K = 24389.0f / (27.0f * 116.0f);
    const float fx = __builtin_fmaf(b, 0.1805f / D65X, __builtin_fmaf(g, 0.3576f / D65X, r * (0.4124f / D65X)));
    const float fy = __builtin_fmaf(b, 0.0722f / D65Y, __builtin_fmaf(g, 0.7152f / D65Y, r * (0.2126f / D65Y)));
    const float fz = __builtin_fmaf(b, 0.9505f / D65Z, __builtin_fmaf(g, 0.1192f / D65Z, r * (0.0193f / D65Z)));
    // Both arms are evaluated and the result is SELECTED: left alone the compiler puts each cube root behind a branch of its
    // own (exec-mask save, compare, branch), which serialises three independent 35-instruction dependency chains that a wave
    // otherwise interleaves.  The empty asm pins the cube root where it is computed (it cannot be sunk into a conditional
    // block).  cbrt_poly of a value <= EPS is harmless (x >= 0: every denominator is >= 0.016) and discarded.
    float cx = cbrt_poly(fx), cy = cbrt_poly(fy), cz = cbrt_poly(fz);
    asm volatile("" : "+v"(cx), "+v"(cy), "+v"(cz));
    const float X = fx > EPS ? cx - 16.0f / 116.0f : K * fx;
    const float Y = fy > EPS ? cy - 16.0f / 116.0f : K * fy;
    const float Z = fz > EPS ? cz - 16.0f / 116.0f : K * fz;
    L = Y * 1.05f;
    A = __builtin_fmaf(500.0f / 220.0f, X - Y, 86.2f / 220.0f);
    B = __builtin_fmaf(200.0f / 220.0f, Y - Z, 107.9f / 220.0f);
}

}  // namespace
#endif

#define CE_DSSIM_STRIP 60  // output columns of a streaming kernel's 64-lane strip (halo 2 on either side)

// Dssim::create_image of level `level` for the image slots z0 .. n_slots - 1 (references first; dssim_stream.hip)
int ce_dssim_create_stream(ce_batch *b, int level, const uint8_t *d_refs, uint32_t n_refs_used, uint32_t n_pairs, uint32_t z0);
// Dssim::compare of level `level` for the first n_pairs pairs (dssim_stream.hip); level_map = the level's SSIM maps (one plane per pair); *n_part = partial sums written per pair
int ce_dssim_compare_stream(ce_batch *b, int level, uint32_t n_pairs, float *level_map, uint32_t *n_part);
