// DSSIM on gfx950 — replaces rgb8_to_dssim_image + calculate_dssim
// (/root/reference/src/metrics/dssim.rs:102-114, 40-71 -> dssim_core::Dssim::{create_image, compare}).
//
// Per level (5 levels, each the 2x2 average of the previous in LINEAR RGB, floor sizes):
//   per image slot ("create_image", references once per reference, not once per pair):
//     linear RGB -> normalised L*a*b*  ->  chroma planes pre-blurred (= img);  references also: mu = blur(img),
//     sq = blur(img*img), kept per level for all their pairs (and across launches for a reference handle)
//   per pair ("compare"):
//     the distorted image's mu / sq and i12 = blur(img1*img2) -> SSIM map over the channel-averaged statistics
//   after the last level, for all levels at once: mean -> mean |avg - ssim| -> weighted score
// "blur" is the fixed 3x3 kernel applied twice with edge replication.  The per-level kernels are the streaming kernels of
// dssim_stream.hip (round 3; rounds 1-2 ran 32 x 32 LDS-tile kernels here: profiles/r03_experiments.md sections 1, 7); this
// file holds the tail reductions, the working set and the launch sequence.  Every plane op keeps the oracle's f32 operation
// order (oracle/dssim.c), so planes are bit-identical; sums are f64.  Build with -ffp-contract=off.
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
    b->ds_blocks = std::max(((b->ds[0].w + 63) / 64) * ((b->ds[0].h + AD_ROWS - 1) / AD_ROWS), ((b->ds[0].w + CE_DSSIM_STRIP - 1) / CE_DSSIM_STRIP) * ((b->ds[0].h + 1) / 2));
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
    // reference handle (ce_ref_*): the references' img / mu / sq pyramid of an earlier launch is still valid
    const bool cached = b->keep_ref_pyramid && b->ds_ref_src == d_refs && b->ds_ref_count >= n_refs_used;
    const uint32_t z0 = cached ? n_refs_used : 0;
    if (!cached) b->ref_builds[1]++;
    ds_geom g{};
    ds_tail tail{};
    size_t map_off = 0;
    for (int l = 0; l < b->ds_levels; l++) {
        const auto &d = b->ds[l];
        // create_image for every used slot (references once per reference), then compare per pair: dssim_stream.hip
        if ((rc = ce_dssim_create_stream(b, l, d_refs, n_refs_used, n_pairs, z0)) != CE_OK) return rc;
        uint32_t n_part = 0;
        if ((rc = ce_dssim_compare_stream(b, l, n_pairs, b->ds_map + map_off, &n_part)) != CE_OK) return rc;
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
