// Internal declarations shared by the host runtime (ce_api.cpp) and the gfx950 kernels.
// Nothing here is part of the ABI; the ABI is include/ce_metrics.h.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <map>
#include <string>
#include <tuple>
#include <utility>
#include <vector>

#include "ce_metrics.h"
#include "ce_metrics_debug.h"

#define CE_MAX_SCALES 6      // SSIMULACRA2 pyramid depth
#define CE_SSIM2_STREAMS 5   // blur(a), blur(b), blur(a*a), blur(b*b), blur(a*b)
#define CE_DSSIM_SCALES 5    // dssim-core DEFAULT_WEIGHTS.len()

struct ce_scale_dims {
    uint32_t w, h, pitch;  // pitch in floats, multiple of 32 (128-byte rows)
    size_t plane;          // pitch * h
    uint32_t hpitch;       // row-blur planes: pitch padded to 32 * ceil((w + 4) / 32) floats,
    size_t hplane;         //   rows padded to a multiple of 64 (branch-free row stores)
};

// per-pair device results; PSNR leaves the device as the exact integer SSE and is
// finished on the host with the host libm (bit-identical to the reference's f64 log10).
struct ce_dev_scores {
    double dssim;
    double ssimulacra2;
    double butteraugli;
    unsigned long long sse;
};

struct ce_kernel_stat {
    std::string name;
    uint64_t launches = 0;
    double total_ms = 0.0;
};

// metric chains of a batch run side by side when the batch holds at most this many megapixels of pairs (ce_api.cpp,
// ce_batch_launch; measured in profiles/r02_experiments.md section 15); CE_METRIC_FORK_BELOW_MP overrides it
#ifndef CE_DEFAULT_FORK_BELOW_MP
#define CE_DEFAULT_FORK_BELOW_MP 4.0
#endif
#ifndef CE_DEFAULT_FORK_ALONE_BELOW_MP
#define CE_DEFAULT_FORK_ALONE_BELOW_MP 64.0
#endif

struct ce_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = true;
    std::string err;

    // device constant tables (built on the host at context creation)
    float *d_lut_ssim2 = nullptr;  // sRGB->linear, f64 formula rounded to f32 (SSIMULACRA2 front end)
    float *d_lut_powf = nullptr;   // sRGB->linear via f32 powf(2.4) (dssim.rs:78-85, xyb.rs:60-66)
    float *d_xyb_thresh = nullptr; // linear->sRGB u8 decision thresholds (xyb.rs:86-88)

    // profiling
    bool prof = false;         // record a HIP event pair around every launch (on the launch's own stream)
    bool prof_serial = false;  // ...and keep everything on the context's stream so kernel times do not overlap
    std::string prof_filter;   // non-empty: only kernels whose name contains this get events
    std::vector<ce_kernel_stat> stats;
    struct pending { int stat; hipEvent_t e0, e1; };
    std::vector<pending> pend;
    std::vector<hipEvent_t> event_pool;
    hipEvent_t t0 = nullptr, t1 = nullptr;

    // scratch batches for the single-pair / mixed-shape entry points, keyed by shape
    // scratch pool for the host-buffer entry points: up to kPoolRing batches per shape (ce_eval_batch streams a large
    // bucket through them in chunks so that the upload of one chunk overlaps the kernels of the previous one)
    static constexpr uint32_t kPoolRing = 3;
    // largest device footprint one ce_eval_batch chunk is sized for (ce_api.cpp: chunk_budget)
    static constexpr size_t kChunkBytesMax = (size_t)48 << 30;
    std::map<std::tuple<uint32_t, uint32_t, uint32_t>, ce_batch *> shape_pool;

    // grow-only scratch of the leaf entry points that take one host image and return one (ce_xyb_roundtrip,
    // ce_rgb8_to_dssim_image): device in / out and a page-locked staging buffer, kept between calls
    uint8_t *leaf_d_in = nullptr, *leaf_d_out = nullptr, *leaf_h = nullptr;
    size_t leaf_in_cap = 0, leaf_out_cap = 0, leaf_h_cap = 0;

    // Auxiliary streams of the context, shared by all its batches (made on first use, destroyed with the context): the
    // three metric chains of a forked batch, SSIMULACRA2's level-0 passes (+ the chunking experiment's second one),
    // Butteraugli's half-resolution chain.  Rounds 1-2 made them per BATCH: a context with a scratch batch and a reference
    // handle then held 13 streams, HIP maps streams onto GPU_MAX_HW_QUEUES hardware queues in creation order, and two
    // chains of one call could land on one queue - the same call took 0.45 or 0.8 ms depending on what had been created
    // before it (profiles/r03_experiments.md section 15).
    enum { AUX_METRIC0 = 0, AUX_SSIM2_L0 = 3, AUX_SSIM2_L0B = 4, AUX_BA_HALF = 5, AUX_COUNT = 6 };
    hipStream_t aux_stream[AUX_COUNT] = {};
    hipStream_t up2_stream = nullptr;  // second DMA stream of ce_eval_batch's page-locked uploads (CE_UPLOAD_STREAMS=2)
    hipEvent_t ev_up2 = nullptr;

    // two parked host threads that enqueue the other metric chains of a forked batch (ce_api.cpp: ce_fork_helpers); made on
    // the first forked launch, joined by ce_ctx_destroy
    struct ce_fork_helpers *helpers = nullptr;
};

// XCD-aware 1-D launch order for per-pair tile kernels (ce_build_xcd_list, ce_api.cpp): entry = (tile, pair)
struct ce_xcd_list {
    uint2 *d = nullptr;
    uint32_t len = 0, cap = 0, version = ~0u, pairs = 0, tiles = 0;
};

// launch order of the streaming per-pair kernels (dssim.hip): one entry per BLOCK = a strip tile and the (up to four)
// distorted images of one reference that its waves walk side by side
struct ce_group_list {
    void *d = nullptr;
    uint32_t len = 0, cap = 0, version = ~0u, pairs = 0, strips = 0, rows = 0, h = 0;
};

struct ce_batch {
    ce_ctx *ctx = nullptr;
    uint32_t w = 0, h = 0, max_refs = 0, max_pairs = 0;
    size_t img_bytes = 0;  // w*h*3

    uint8_t *d_refs = nullptr;     // [max_refs][h][w][3]
    uint8_t *d_refs_rt = nullptr;  // XYB-roundtripped references (lazily allocated)
    uint8_t *d_tests = nullptr;    // [max_pairs][h][w][3]
    uint32_t *d_pair_ref = nullptr;
    uint32_t *d_pair_first = nullptr;  // second part of the d_pair_ref allocation
    uint32_t *d_ref_off = nullptr, *d_ref_idx = nullptr;  // ... then reference -> its pairs (CSR: [max_refs + 1] offsets, [max_pairs] pair indices)
    std::vector<uint32_t> h_pair_ref;
    bool pair_ref_dirty = true;
    std::tuple<uint32_t, uint32_t, uint32_t> pool_key{0, 0, 0};  // (w, h, ring slot) when owned by a context's scratch pool
    uint32_t pair_ref_version = 0;  // bumped whenever the pair -> reference table changes
    // pinned staging ring for host -> device uploads: the host copy into slot k overlaps the DMA of slot k-1
    static constexpr int kStages = 8;  // two per upload worker (ce_eval_batch fills a bucket with 4 host threads)
    uint8_t *h_stage[kStages] = {};
    hipEvent_t ev_stage[kStages] = {};
    bool stage_busy[kStages] = {};
    int next_stage = 0;
    // uploads run on their own stream so that filling one batch overlaps another batch's kernels
    hipStream_t up_stream = nullptr;
    hipEvent_t ev_up = nullptr, ev_run = nullptr;  // uploads done / last launch done
    bool uploads_pending = false, run_pending = false;
    bool counted_in_flight = false;  // this batch is in the device's launched-and-not-collected count (ce_api.cpp: g_in_flight)
    // wide ingest (RGBA8 / 16-bit sources): one pinned + one device staging image of 8 B/px, made on first use
    // (two of each: while image k's copy and conversion are in flight the host fills the other pair, so a sweep of wide
    // decoded images does not synchronise the upload stream per image)
    uint8_t *h_wide[2] = {}, *d_wide[2] = {};
    hipEvent_t ev_wide[2] = {};
    bool wide_busy[2] = {};
    int next_wide = 0;

    // SSIMULACRA2 working set.  Image slots: [0, max_refs) references, then tests.
    int n_scales = 0;
    ce_scale_dims sd[CE_MAX_SCALES];
    float *d_lin[CE_MAX_SCALES] = {};  // [slots][3][plane_s] linear RGB pyramid
    float *d_xyb[CE_MAX_SCALES] = {};   // [slots][3][plane_s] positive XYB, one buffer per level
    float *d_hbuf[CE_MAX_SCALES] = {};  // [pairs][3][5][plane_s] row-blurred streams, one buffer per level
    // level 0's row/column pass runs on lvl_stream[0], fenced by events against the front end and the final reduction;
    // the other levels follow the front end on the context's stream (entries 1.. are unused)
    hipStream_t lvl_stream[CE_MAX_SCALES] = {};
    // one stream per metric chain when a launch runs several of them (SSIMULACRA2, DSSIM, Butteraugli side by side)
    hipStream_t metric_stream[3] = {};
    hipEvent_t ev_fork = nullptr, ev_join[3] = {};
    hipEvent_t ev_prep[CE_MAX_SCALES] = {}, ev_done[CE_MAX_SCALES] = {};
    double *d_partials = nullptr;      // [pairs][scales][3][max_blocks][6]
    uint32_t max_vblocks = 0;
    double *d_avg = nullptr;           // [pairs][6][3][6]
    ce_dev_scores *d_scores = nullptr;
    ce_dev_scores *h_scores = nullptr;  // pinned
    bool ssim2_ready = false;
    // XCD-aware work lists of the level-0 row / column pass (ssim2.hip): launch id -> (block, channel, pair)
    uint2 *d_work_h = nullptr, *d_work_v = nullptr;    // level 0
    uint2 *d_work_ht = nullptr, *d_work_vt = nullptr;  // the merged launch of levels 1..
    uint32_t work_len_h = 0, work_len_v = 0, work_cap_h = 0, work_cap_v = 0, work_version = ~0u, work_pairs = 0;
    uint32_t work_len_ht = 0, work_len_vt = 0, work_cap_ht = 0, work_cap_vt = 0, work_version_t = ~0u, work_pairs_t = 0;
    uint32_t work_blk_ht = 0, work_blk_vt = 0;
    std::vector<uint2> work_chunks_h, work_chunks_v;  // (offset, length) segments of the level-0 lists (CE_SSIM2_L0_CHUNK experiment)
    // reference handles (ce_ref_*): the references' XYB pyramid of the last SSIMULACRA2 run stays valid
    // until a reference is replaced, so later compares only build the distorted side
    bool keep_ref_pyramid = false;
    const uint8_t *ssim2_ref_src = nullptr;  // reference slab the cached pyramid was built from
    uint32_t ssim2_ref_count = 0;            // ...and how many references it covers
    int ssim2_ref_levels = 0;
    bool refs_rt_valid = false;              // d_refs_rt holds the XYB roundtrip of the current references
    int debug_max_scales = CE_MAX_SCALES;  // test hook: stop the pyramid early
    // launches in which the reference-side state of [SSIMULACRA2, DSSIM, Butteraugli] was (re)built (ce_ref_stats)
    uint32_t ref_builds[3] = {0, 0, 0};

    // DSSIM working set (dssim.hip); planes are [slot][3][plane] with the level's own geometry
    struct dssim_level { uint32_t w, h, pitch; size_t plane; };
    int ds_levels = 0;
    dssim_level ds[CE_DSSIM_SCALES];
    float *ds_lin[2] = {};     // linear RGB of the current and the next level
    float *ds_img = nullptr;   // [max_pairs][3][plane_0]: L, a', b' (chroma pre-blurred) of the distorted images, one level at a time
    float *ds_rimg[CE_DSSIM_SCALES] = {}, *ds_rmu[CE_DSSIM_SCALES] = {}, *ds_rsq[CE_DSSIM_SCALES] = {};  // the references' planes, per level: [max_refs][3][plane_l]
    const uint8_t *ds_ref_src = nullptr;  // reference slab those planes were built from (valid while keep_ref_pyramid)
    uint32_t ds_ref_count = 0;
    float *ds_map = nullptr;   // [pairs][plane] SSIM map
    double *ds_part = nullptr; // [pairs][levels][2][blocks] partial sums (sum, abs-dev)
    double *ds_level_scores = nullptr;  // [pairs][levels]
    uint32_t ds_blocks = 0;
    ce_group_list ds_gwork[CE_DSSIM_SCALES];  // k_dssim_compare_stream's launch order, per level
    bool dssim_ready = false;

    // Butteraugli working set (butteraugli.hip): level 0 = full resolution, 1 = 2x-subsampled
    struct ba_level { uint32_t w, h, pitch; size_t plane; };
    ba_level ba[2];
    int ba_levels = 0;
    float *ba_psy[2] = {};    // [slot][10][plane_l]  PsychoImage
    float *ba_diff[2] = {};   // [pair][plane_1]      the half-resolution diffmap ([1]; the full-resolution one is reduced in registers)
    float *ba_mask[2] = {};   // [slot][plane_l]      blurred mask input (DiffPrecompute of HF + UHF, sigma 2.7)
    float *ba_s[3] = {};      // per-slot scratch, 3 planes each
    // a small batch runs its half-resolution chain on a stream of its own, beside the full-resolution one, with its own
    // scratch (butteraugli.hip: ce_launch_butteraugli); made on first use
    float *ba_s_half[3] = {};
    hipStream_t ba_half_stream = nullptr;
    hipEvent_t ev_ba_fork = nullptr, ev_ba_join = nullptr;
    float *ba_mask_vals[2] = {};  // [ref][2][plane_l]    maskval / dc_maskval of the references (FuzzyErosion + mask curves)
    float *ba_blk_max = nullptr;
    double *ba_blk_sums = nullptr;
    double *ba_pnorm = nullptr;  // [pair] libjxl 3-norm of the last run
    uint32_t ba_blocks = 0;
    ce_xcd_list ba_work[2];  // Malta's launch order, per resolution level
    bool ba_ready = false;
    const uint8_t *ba_ref_src = nullptr;  // reference slab the references' PsychoImage in ba_psy was built from
    uint32_t ba_ref_count = 0;
    float ba_ref_intensity = 0.0f;

    uint32_t last_n_pairs = 0;
    bool caller_blocks = false;  // set by entry points that collect before returning: page-locked sources need no staging copy
    uint32_t last_mask = 0;
};

#define CE_HIP(ctx_, expr)                                                                         \
    do {                                                                                           \
        hipError_t e__ = (expr);                                                                   \
        if (e__ != hipSuccess) {                                                                   \
            (ctx_)->err = std::string(#expr) + ": " + hipGetErrorString(e__);                      \
            return CE_ERR_BACKEND;                                                                 \
        }                                                                                          \
    } while (0)

// host table -> device memory of `b`, complete on return, without draining the context's stream (ce_api.cpp)
int ce_upload_table(ce_batch *b, void *dst, const void *src, size_t bytes);

// the context's auxiliary stream `which` (ce_ctx::AUX_*), made on first use; nullptr + ctx->err on failure (ce_api.cpp)
hipStream_t ce_ctx_aux_stream(ce_ctx *ctx, int which);

// profiling hooks around a launch (ce_api.cpp)
int ce_prof_begin(ce_ctx *ctx, const char *name, hipStream_t stream);
void ce_prof_end(ce_ctx *ctx, int token, hipStream_t stream);

// The stream a metric's launch function enqueues on: the context's stream, unless the calling thread is one of the helper
// threads that enqueue a forked batch's chains side by side (ce_api.cpp: ce_batch_launch) - those set the override.
extern thread_local hipStream_t ce_tls_stream;
#define CE_STREAM(ctx_) (ce_tls_stream ? ce_tls_stream : (ctx_)->stream)

#define CE_LAUNCH_ON(ctx_, stream_, name_, kern_, grid_, block_, shmem_, ...)                      \
    do {                                                                                           \
        int tok__ = (ctx_)->prof ? ce_prof_begin((ctx_), name_, (stream_)) : -1;                   \
        hipLaunchKernelGGL(kern_, grid_, block_, shmem_, (stream_), __VA_ARGS__);                  \
        if (tok__ >= 0) ce_prof_end((ctx_), tok__, (stream_));                                     \
    } while (0)
#define CE_LAUNCH(ctx_, name_, kern_, grid_, block_, shmem_, ...)                                  \
    CE_LAUNCH_ON(ctx_, CE_STREAM(ctx_), name_, kern_, grid_, block_, shmem_, __VA_ARGS__)

#if defined(__HIPCC__)
// Correctly rounded f32 quotients without the range-scaling steps of the compiler's expansion.  hipcc turns a / b into
// v_div_scale x2, v_rcp, two fused steps refining the reciprocal, a product, two fused quotient corrections, v_div_fmas
// and v_div_fixup (11 instructions).  When neither operand is zero-denominator / infinite / NaN / denormal and the
// quotient is far from overflow and underflow - every call site below divides values between 2^-40 and 2^40 (a zero
// numerator included) - v_div_scale returns its operands unchanged, v_div_fmas is a plain fma and v_div_fixup returns
// the quotient, so the same arithmetic is 8 instructions, bit for bit.  ce_debug_div_sweep checks both forms against
// operator/ on the device over exactly that range.
__device__ __forceinline__ float ce_div_refined(float a, float b, float r)  // r = refined reciprocal of b
{
    float q = a * r;
    q = __builtin_fmaf(__builtin_fmaf(-b, q, a), r, q);
    return __builtin_fmaf(__builtin_fmaf(-b, q, a), r, q);
}
__device__ __forceinline__ float ce_rcp_refined(float b)
{
    const float r0 = __builtin_amdgcn_rcpf(b);
    return __builtin_fmaf(__builtin_fmaf(-b, r0, 1.0f), r0, r0);
}
__device__ __forceinline__ float ce_div_noscale(float a, float b) { return ce_div_refined(a, b, ce_rcp_refined(b)); }
#endif

// ---- kernel launchers (one .hip file per metric) ---------------------------------------
int ce_launch_psnr(ce_batch *b, const uint8_t *d_refs, uint32_t n_pairs);
size_t ce_pixel_bytes(int format);
int ce_launch_ingest(ce_ctx *ctx, hipStream_t stream, int format, const void *d_src, uint8_t *d_dst, size_t n_pixels);
int ce_launch_lut_expand(ce_ctx *ctx, hipStream_t stream, const uint8_t *d_packed, uint32_t *d_table);
int ce_launch_lut_apply(ce_ctx *ctx, hipStream_t stream, uint8_t *d_rgb, const uint32_t *d_table, size_t n_pixels);
int ce_ssim2_prepare(ce_batch *b);
void ce_ssim2_free(ce_batch *b);
int ce_launch_ssim2(ce_batch *b, const uint8_t *d_refs, uint32_t n_refs_used, uint32_t n_pairs);
int ce_ssim2_occupancy(int which);
int ce_ssim2_cbrt_sweep(ce_ctx *ctx, uint32_t first_bits, uint64_t count, uint64_t *mismatches, uint64_t *slow_path);
int ce_launch_xyb_roundtrip(ce_ctx *ctx, const uint8_t *d_in, uint8_t *d_out, size_t n_pixels);
int ce_launch_dssim(ce_batch *b, const uint8_t *d_refs, uint32_t n_refs_used, uint32_t n_pairs);
void ce_dssim_free(ce_batch *b);
int ce_launch_butteraugli(ce_batch *b, const uint8_t *d_refs, uint32_t n_refs_used, uint32_t n_pairs, float intensity_target);
void ce_butteraugli_free(ce_batch *b);
int ce_butteraugli_div_sweep(ce_ctx *ctx, uint64_t seed, uint64_t count, uint64_t *mismatches);
int ce_calibrate_traffic(ce_ctx *ctx, size_t bytes);
int ce_build_xcd_list(ce_batch *b, uint32_t n_pairs, uint32_t n_tiles, ce_xcd_list *list);
void ce_free_xcd_list(ce_xcd_list *list);
int ce_launch_rgb8_to_dssim_image(ce_ctx *ctx, const uint8_t *d_rgb, float *d_rgba, size_t n_pixels);

// host-side constant builders (ce_tables.cpp)
void ce_build_srgb_lut_f64(float lut[256]);
void ce_build_srgb_lut_powf(float lut[256]);
void ce_ssim2_recursive_gaussian(float mul_in[3], float mul_prev[3]);
bool ce_build_xyb_srgb_thresholds(float thresh[256]);
