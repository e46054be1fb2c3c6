// Host runtime behind the C ABI (include/ce_metrics.h): contexts, the HBM-resident pair
// grid, shape bucketing for mixed batches, input validation with the reference's error
// kinds, profiling hooks.  All device work is in the .hip files; there is no CPU compute
// path here — if HIP is unavailable every entry point fails with CE_ERR_BACKEND.
#include <algorithm>
#include <array>
#include <atomic>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <thread>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <new>

#include "ce_internal.h"

namespace {

thread_local std::string g_err_noctx;

int fail(ce_ctx *ctx, int code, const std::string &msg)
{
    if (ctx) ctx->err = msg; else g_err_noctx = msg;
    return code;
}

// validation order of calculate_ssimulacra2 / calculate_butteraugli
// (src/metrics/ssimulacra2.rs:65-82, src/metrics/butteraugli.rs:51-67)
int validate_pair(ce_ctx *ctx, size_t ref_len, size_t test_len, size_t w, size_t h)
{
    if (ref_len != test_len)
        return fail(ctx, CE_ERR_DIM_MISMATCH, "Dimension mismatch: reference " + std::to_string(ref_len) +
                                                   " bytes, test " + std::to_string(test_len) + " bytes");
    if (ref_len != w * h * 3)
        return fail(ctx, CE_ERR_BAD_LENGTH, "Invalid image size: expected " + std::to_string(w * h * 3) +
                                                 " bytes, got " + std::to_string(ref_len));
    return CE_OK;
}

double psnr_from_sse(unsigned long long sse, size_t w, size_t h)
{
    // src/metrics/mod.rs:317,324-330
    const double pixel_count = (double)(w * h * 3);
    const double mse = (double)sse / pixel_count;
    if (mse == 0.0) return INFINITY;
    return 10.0 * std::log10(255.0 * 255.0 / mse);
}

}  // namespace

// ---- profiling -----------------------------------------------------------------------------

int ce_prof_begin(ce_ctx *ctx, const char *name, hipStream_t stream)
{
    if (!ctx->prof_filter.empty()) {  // "=name": that kernel only; otherwise a substring
        const char *f = ctx->prof_filter.c_str();
        if (f[0] == '=' ? std::strcmp(name, f + 1) != 0 : !std::strstr(name, f)) return -1;
    }
    int idx = -1;
    for (size_t i = 0; i < ctx->stats.size(); i++)
        if (ctx->stats[i].name == name) { idx = (int)i; break; }
    if (idx < 0) {
        ctx->stats.push_back(ce_kernel_stat{name, 0, 0.0});
        idx = (int)ctx->stats.size() - 1;
    }
    ce_ctx::pending pd{idx, nullptr, nullptr};
    for (hipEvent_t *e : {&pd.e0, &pd.e1}) {
        if (!ctx->event_pool.empty()) {
            *e = ctx->event_pool.back();
            ctx->event_pool.pop_back();
        } else if (hipEventCreate(e) != hipSuccess) {
            return -1;
        }
    }
    hipEventRecord(pd.e0, stream);
    ctx->pend.push_back(pd);
    return (int)ctx->pend.size() - 1;
}

void ce_prof_end(ce_ctx *ctx, int token, hipStream_t stream) { hipEventRecord(ctx->pend[token].e1, stream); }

static void prof_drain(ce_ctx *ctx)
{
    if (ctx->pend.empty()) return;
    hipSetDevice(ctx->device);
    hipStreamSynchronize(ctx->stream);
    for (auto &pd : ctx->pend) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, pd.e0, pd.e1) == hipSuccess) {
            ctx->stats[pd.stat].launches++;
            ctx->stats[pd.stat].total_ms += ms;
        }
        ctx->event_pool.push_back(pd.e0);
        ctx->event_pool.push_back(pd.e1);
    }
    ctx->pend.clear();
}

// XCD-aware launch order for kernels whose workgroup = (tile, pair).  Workgroups reach the 8 XCDs round-robin by launch
// id and every XCD has its own L2, so the workgroups that read the same tile of one REFERENCE for its different
// distorted images should carry ids that are congruent mod 8 and adjacent: the reference's planes are then fetched into
// one XCD's L2 once and hit there by the reference's other pairs (the scheme of the SSIMULACRA2 passes, ssim2.hip).
// Keys (reference, tile) are dealt to the 8 classes in turn, each key followed by all pairs of its reference;
// entry id = slot * 8 + class; classes are padded with (~0, 0) entries (the kernel returns at once).  Rebuilt only when
// the pair -> reference table, the pair count or the tile count changes.
int ce_build_xcd_list(ce_batch *b, uint32_t n_pairs, uint32_t n_tiles, ce_xcd_list *L)
{
    if (L->d && L->version == b->pair_ref_version && L->pairs == n_pairs && L->tiles == n_tiles) return CE_OK;
    ce_ctx *ctx = b->ctx;
    std::vector<std::vector<uint32_t>> pairs_of(b->max_refs);
    for (uint32_t p = 0; p < n_pairs; p++) pairs_of[b->h_pair_ref[p]].push_back(p);
    static const bool natural = [] {  // CE_XCD_ORDER=0: pair-major, tile-minor (what a 3-D grid would do) - A/B knob
        const char *e = std::getenv("CE_XCD_ORDER");
        return e && std::atoi(e) == 0;
    }();
    std::vector<uint2> flat;
    if (natural) {
        for (uint32_t p = 0; p < n_pairs; p++)
            for (uint32_t t = 0; t < n_tiles; t++) flat.push_back(make_uint2(t, p));
    } else {
        std::vector<uint2> cls[8];
        uint32_t k = 0;
        for (uint32_t r = 0; r < b->max_refs; r++) {
            if (pairs_of[r].empty()) continue;
            for (uint32_t t = 0; t < n_tiles; t++, k++)
                for (uint32_t p : pairs_of[r]) cls[k & 7].push_back(make_uint2(t, p));
        }
        size_t longest = 0;
        for (auto &v : cls) longest = std::max(longest, v.size());
        flat.assign(longest * 8, make_uint2(~0u, 0u));
        for (uint32_t x = 0; x < 8; x++)
            for (size_t sl = 0; sl < cls[x].size(); sl++) flat[sl * 8 + x] = cls[x][sl];
    }
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
    L->tiles = n_tiles;
    return CE_OK;
}

void ce_free_xcd_list(ce_xcd_list *L)
{
    hipFree(L->d);
    *L = ce_xcd_list{};
}

thread_local hipStream_t ce_tls_stream = nullptr;  // ce_internal.h: CE_STREAM

// Host table -> device memory of batch `b`, complete when this returns.  Work lists and the pair -> reference table are
// built in pageable vectors, so the host has to wait for the copy - but NOT for the context's stream: round 2 copied on
// that stream and drained it, which made every ce_eval_batch chunk whose pair count differs from the slot's previous one
// wait for the kernels of the chunk before it, and its uploads no longer overlapped them (profiles/r03_experiments.md
// section 17: 141 -> see there).  The copy runs on the batch's upload stream, behind the previous launch of THIS batch
// (which may still read the old table) and behind the image uploads already queued there.
int ce_upload_table(ce_batch *b, void *dst, const void *src, size_t bytes)
{
    ce_ctx *ctx = b->ctx;
    if (bytes == 0) return CE_OK;
    CE_HIP(ctx, hipStreamWaitEvent(b->up_stream, b->ev_run, 0));  // never recorded = no wait
    CE_HIP(ctx, hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, b->up_stream));
    CE_HIP(ctx, hipStreamSynchronize(b->up_stream));
    return CE_OK;
}

hipStream_t ce_ctx_aux_stream(ce_ctx *ctx, int which)
{
    if (which < 0 || which >= ce_ctx::AUX_COUNT) return nullptr;
    if (!ctx->aux_stream[which]) {
        const hipError_t e = hipStreamCreateWithFlags(&ctx->aux_stream[which], hipStreamNonBlocking);
        if (e != hipSuccess) {
            ctx->aux_stream[which] = nullptr;
            ctx->err = std::string("hipStreamCreateWithFlags (auxiliary stream): ") + hipGetErrorString(e);
        }
    }
    return ctx->aux_stream[which];
}

extern "C" {

const char *ce_version(void) { return "codec-eval_amd 0.3.0 (gfx950)"; }

int ce_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return -1;
    return n;
}

int ce_ctx_create_on_stream(int device, void *hip_stream, ce_ctx **out)
{
    if (!out) return CE_ERR_INVALID_ARG;
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0)
        return fail(nullptr, CE_ERR_BACKEND, "no HIP device visible");
    if (device < 0 || device >= n) return fail(nullptr, CE_ERR_INVALID_ARG, "device index out of range");
    ce_ctx *ctx = new (std::nothrow) ce_ctx();
    if (!ctx) return CE_ERR_BACKEND;
    ctx->device = device;
    auto bail = [&](const char *what, hipError_t e) {
        g_err_noctx = std::string(what) + ": " + hipGetErrorString(e);
        delete ctx;
        return CE_ERR_BACKEND;
    };
    hipError_t e;
    if ((e = hipSetDevice(device)) != hipSuccess) return bail("hipSetDevice", e);
    if (hip_stream) {
        ctx->stream = (hipStream_t)hip_stream;
        ctx->own_stream = false;
    } else {
        if ((e = hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking)) != hipSuccess)
            return bail("hipStreamCreate", e);
    }
    float lut[256];
    if ((e = hipMalloc(&ctx->d_lut_ssim2, sizeof(lut))) != hipSuccess) return bail("hipMalloc", e);
    if ((e = hipMalloc(&ctx->d_lut_powf, sizeof(lut))) != hipSuccess) return bail("hipMalloc", e);
    ce_build_srgb_lut_f64(lut);
    if ((e = hipMemcpy(ctx->d_lut_ssim2, lut, sizeof(lut), hipMemcpyHostToDevice)) != hipSuccess)
        return bail("hipMemcpy", e);
    ce_build_srgb_lut_powf(lut);
    if ((e = hipMemcpy(ctx->d_lut_powf, lut, sizeof(lut), hipMemcpyHostToDevice)) != hipSuccess)
        return bail("hipMemcpy", e);
    float thresh[256];
    if (!ce_build_xyb_srgb_thresholds(thresh)) {
        g_err_noctx = "host powf is not monotone around an sRGB code boundary; cannot build the XYB table";
        delete ctx;
        return CE_ERR_BACKEND;
    }
    if ((e = hipMalloc(&ctx->d_xyb_thresh, sizeof(thresh))) != hipSuccess) return bail("hipMalloc", e);
    if ((e = hipMemcpy(ctx->d_xyb_thresh, thresh, sizeof(thresh), hipMemcpyHostToDevice)) != hipSuccess)
        return bail("hipMemcpy", e);
    if ((e = hipEventCreate(&ctx->t0)) != hipSuccess) return bail("hipEventCreate", e);
    if ((e = hipEventCreate(&ctx->t1)) != hipSuccess) return bail("hipEventCreate", e);
    *out = ctx;
    return CE_OK;
}

int ce_ctx_create(int device, ce_ctx **out) { return ce_ctx_create_on_stream(device, nullptr, out); }

// Parked helper threads of a context.  Round 2 spawned a std::thread per chain and call: creating a thread and making its
// first HIP call cost ~50-100 us, more than the 0.2 ms of launches it was meant to overlap (a single 768x512 pair traced in
// round 3: 96 us between the upload and the first kernel, the third chain starting 190 us after the first).  A helper sets
// its device once and sleeps on a condition variable between jobs.
struct ce_fork_helpers {
    struct worker {
        std::thread th;
        std::mutex m;
        std::condition_variable cv;
        std::function<void()> job;
        bool has_job = false, done = true, stop = false;
    } w[2];
    explicit ce_fork_helpers(int device)
    {
        for (auto &x : w)
            x.th = std::thread([&x, device] {
                (void)hipSetDevice(device);
                std::unique_lock<std::mutex> lk(x.m);
                for (;;) {
                    x.cv.wait(lk, [&] { return x.has_job || x.stop; });
                    if (x.stop) return;
                    x.has_job = false;
                    lk.unlock();
                    x.job();
                    lk.lock();
                    x.done = true;
                    x.cv.notify_all();
                }
            });
    }
    void submit(int i, std::function<void()> f)
    {
        std::lock_guard<std::mutex> lk(w[i].m);
        w[i].job = std::move(f);
        w[i].has_job = true;
        w[i].done = false;
        w[i].cv.notify_all();
    }
    void wait(int i)
    {
        std::unique_lock<std::mutex> lk(w[i].m);
        w[i].cv.wait(lk, [&] { return w[i].done; });
    }
    ~ce_fork_helpers()
    {
        for (auto &x : w) {
            {
                std::lock_guard<std::mutex> lk(x.m);
                x.stop = true;
                x.cv.notify_all();
            }
            if (x.th.joinable()) x.th.join();
        }
    }
};

void ce_ctx_destroy(ce_ctx *ctx)
{
    if (!ctx) return;
    hipSetDevice(ctx->device);
    hipStreamSynchronize(ctx->stream);
    delete ctx->helpers;
    ctx->helpers = nullptr;

    prof_drain(ctx);
    for (auto &kv : ctx->shape_pool) ce_batch_destroy(kv.second);
    ctx->shape_pool.clear();
    if (ctx->up2_stream) hipStreamSynchronize(ctx->up2_stream), hipStreamDestroy(ctx->up2_stream), hipEventDestroy(ctx->ev_up2);
    for (auto &st : ctx->aux_stream)  // after the last batch that may still drain them
        if (st) hipStreamSynchronize(st), hipStreamDestroy(st), st = nullptr;
    for (hipEvent_t ev : ctx->event_pool) hipEventDestroy(ev);
    if (ctx->t0) hipEventDestroy(ctx->t0);
    if (ctx->t1) hipEventDestroy(ctx->t1);
    hipFree(ctx->leaf_d_in);
    hipFree(ctx->leaf_d_out);
    if (ctx->leaf_h) hipHostFree(ctx->leaf_h);
    hipFree(ctx->d_lut_ssim2);
    hipFree(ctx->d_lut_powf);
    hipFree(ctx->d_xyb_thresh);
    if (ctx->own_stream && ctx->stream) hipStreamDestroy(ctx->stream);
    delete ctx;
}

int ce_ctx_synchronize(ce_ctx *ctx)
{
    if (!ctx) return CE_ERR_INVALID_ARG;
    CE_HIP(ctx, hipSetDevice(ctx->device));
    CE_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return CE_OK;
}

void *ce_ctx_stream(ce_ctx *ctx) { return ctx ? (void *)ctx->stream : nullptr; }

// batches launched and not yet collected, per device, over all contexts of the process: ce_batch_launch forks a larger
// batch's metric chains only when nothing else is in flight beside it
static std::atomic<int> g_in_flight[64];
static std::atomic<int> &in_flight_of(const ce_ctx *ctx) { return g_in_flight[(unsigned)ctx->device % 64u]; }
static void leave_flight(ce_batch *b)
{
    if (b->counted_in_flight) {
        in_flight_of(b->ctx).fetch_sub(1, std::memory_order_relaxed);
        b->counted_in_flight = false;
    }
}

const char *ce_last_error(const ce_ctx *ctx) { return ctx ? ctx->err.c_str() : g_err_noctx.c_str(); }

// ---- resident batch ------------------------------------------------------------------------

int ce_batch_create(ce_ctx *ctx, uint32_t width, uint32_t height, uint32_t max_refs, uint32_t max_pairs,
                    ce_batch **out)
{
    if (!ctx || !out || width == 0 || height == 0 || max_refs == 0 || max_pairs == 0) return CE_ERR_INVALID_ARG;
    *out = nullptr;
    CE_HIP(ctx, hipSetDevice(ctx->device));
    ce_batch *b = new (std::nothrow) ce_batch();
    if (!b) return CE_ERR_BACKEND;
    b->ctx = ctx;
    b->w = width;
    b->h = height;
    b->max_refs = max_refs;
    b->max_pairs = max_pairs;
    b->img_bytes = (size_t)width * height * 3;
    b->h_pair_ref.assign(max_pairs, 0);
    int rc = CE_OK;
    auto chk = [&](hipError_t e, const char *what) {
        if (e != hipSuccess && rc == CE_OK) {
            ctx->err = std::string(what) + ": " + hipGetErrorString(e);
            rc = CE_ERR_BACKEND;
        }
    };
    // +16 bytes: the PSNR kernel reads whole 16-byte words
    chk(hipMalloc(&b->d_refs, b->img_bytes * max_refs + 16), "hipMalloc refs");
    chk(hipMalloc(&b->d_tests, b->img_bytes * max_pairs + 16), "hipMalloc tests");
    // [pair_ref (P) | pair_first (P) | ref_off (R + 1) | ref_idx (P)]
    chk(hipMalloc(&b->d_pair_ref, sizeof(uint32_t) * (3 * (size_t)max_pairs + max_refs + 1)), "hipMalloc pair_ref");
    b->d_pair_first = b->d_pair_ref ? b->d_pair_ref + max_pairs : nullptr;
    b->d_ref_off = b->d_pair_ref ? b->d_pair_ref + 2 * (size_t)max_pairs : nullptr;
    b->d_ref_idx = b->d_pair_ref ? b->d_ref_off + max_refs + 1 : nullptr;
    chk(hipMalloc(&b->d_scores, sizeof(ce_dev_scores) * max_pairs), "hipMalloc scores");
    chk(hipHostMalloc(&b->h_scores, sizeof(ce_dev_scores) * max_pairs, hipHostMallocDefault), "hipHostMalloc scores");
    chk(hipStreamCreateWithFlags(&b->up_stream, hipStreamNonBlocking), "hipStreamCreate upload");
    chk(hipEventCreateWithFlags(&b->ev_up, hipEventDisableTiming), "hipEventCreate");
    chk(hipEventCreateWithFlags(&b->ev_run, hipEventDisableTiming), "hipEventCreate");
    for (int k = 0; k < ce_batch::kStages; k++) {
        chk(hipHostMalloc(&b->h_stage[k], b->img_bytes, hipHostMallocDefault), "hipHostMalloc stage");
        chk(hipEventCreateWithFlags(&b->ev_stage[k], hipEventDisableTiming), "hipEventCreate stage");
    }
    if (rc == CE_OK) chk(hipMemsetAsync(b->d_scores, 0, sizeof(ce_dev_scores) * max_pairs, ctx->stream), "memset");
    if (rc != CE_OK) {
        ce_batch_destroy(b);
        return rc;
    }
    *out = b;
    return CE_OK;
}

void ce_batch_destroy(ce_batch *b)
{
    if (!b) return;
    hipSetDevice(b->ctx->device);
    hipStreamSynchronize(b->ctx->stream);
    leave_flight(b);
    if (b->up_stream) hipStreamSynchronize(b->up_stream), hipStreamDestroy(b->up_stream);
    for (int k = 0; k < 2; k++) {
        if (b->h_wide[k]) hipHostFree(b->h_wide[k]);
        hipFree(b->d_wide[k]);
        if (b->ev_wide[k]) hipEventDestroy(b->ev_wide[k]);
    }
    if (b->ev_up) hipEventDestroy(b->ev_up);
    if (b->ev_run) hipEventDestroy(b->ev_run);
    if (b->ev_fork) hipEventDestroy(b->ev_fork);
    hipFree(b->d_refs);
    hipFree(b->d_refs_rt);
    hipFree(b->d_tests);
    hipFree(b->d_pair_ref);
    hipFree(b->d_scores);
    if (b->h_scores) hipHostFree(b->h_scores);
    for (int k = 0; k < ce_batch::kStages; k++) {
        if (b->h_stage[k]) hipHostFree(b->h_stage[k]);
        if (b->ev_stage[k]) hipEventDestroy(b->ev_stage[k]);
    }
    for (int l = 0; l < 3; l++) {
        if (b->metric_stream[l]) hipStreamSynchronize(b->metric_stream[l]);  // the context's stream: drained, not destroyed
        if (b->ev_join[l]) hipEventDestroy(b->ev_join[l]);
    }
    ce_ssim2_free(b);
    hipFree(b->d_work_h);
    hipFree(b->d_work_v);
    hipFree(b->d_work_ht);
    hipFree(b->d_work_vt);
    ce_dssim_free(b);
    ce_butteraugli_free(b);
    delete b;
}

// the reference slab is about to change: whatever was derived from it (XYB roundtrip, SSIMULACRA2 XYB pyramid, DSSIM
// img / mu / sq pyramid, Butteraugli PsychoImage) is rebuilt by the next launch
static void invalidate_reference_state(ce_batch *b)
{
    b->ssim2_ref_src = nullptr;
    b->ds_ref_src = nullptr;
    b->ba_ref_src = nullptr;
    b->refs_rt_valid = false;
}

static bool is_pinned_host(const void *p);

static int upload(ce_batch *b, uint8_t *dst, const uint8_t *src, bool allow_inline = true)
{
    ce_ctx *ctx = b->ctx;
    // pageable source -> pinned staging ring -> device on the batch's upload stream.  The caller's buffer is
    // consumed before this returns; the DMA of this slot overlaps the host copy into the next one and the
    // kernels of other batches.  A launched-but-uncollected run of THIS batch still reads the slabs: wait for it.
    CE_HIP(ctx, hipSetDevice(ctx->device));  // the calling thread's current device may be another one (multi-device hosts)
    // A small batch (the one-pair-per-call regime of a reference handle) uploads on the context's own stream: its launch
    // follows at once, and a cross-stream event between the copy and the first kernel costs ~25 us of its ~0.5 ms
    // (not for an image that a format conversion or a colour table follows on the upload stream: allow_inline = false)
    const bool inline_copy = allow_inline && (double)b->max_pairs * b->w * b->h <= 4e6;
    hipStream_t us = inline_copy ? ctx->stream : b->up_stream;
    if (b->run_pending && !inline_copy) {
        CE_HIP(ctx, hipStreamWaitEvent(b->up_stream, b->ev_run, 0));
        b->run_pending = false;  // ordered from here on
    }
    // A BLOCKING entry point (ce_ref_compare*, which collects before it returns) whose caller's image is page-locked
    // (ce_host_alloc) needs no staging copy: the DMA engine reads the caller's buffer, which outlives the call's kernels.
    if (b->caller_blocks && is_pinned_host(src)) {
        CE_HIP(ctx, hipMemcpyAsync(dst, src, b->img_bytes, hipMemcpyHostToDevice, us));
        if (!inline_copy) b->uploads_pending = true;
        return CE_OK;
    }
    const int k = b->next_stage;
    b->next_stage = (k + 1) % ce_batch::kStages;
    if (b->stage_busy[k]) CE_HIP(ctx, hipEventSynchronize(b->ev_stage[k]));
    std::memcpy(b->h_stage[k], src, b->img_bytes);
    CE_HIP(ctx, hipMemcpyAsync(dst, b->h_stage[k], b->img_bytes, hipMemcpyHostToDevice, us));
    CE_HIP(ctx, hipEventRecord(b->ev_stage[k], us));
    b->stage_busy[k] = true;
    if (!inline_copy) b->uploads_pending = true;
    return CE_OK;
}

// kernels (context stream) must see everything uploaded so far
static int flush_uploads(ce_batch *b)
{
    if (!b->uploads_pending) return CE_OK;
    ce_ctx *ctx = b->ctx;
    CE_HIP(ctx, hipEventRecord(b->ev_up, b->up_stream));
    CE_HIP(ctx, hipStreamWaitEvent(ctx->stream, b->ev_up, 0));
    b->uploads_pending = false;
    return CE_OK;
}

// Many images at once (ce_eval_batch): the host copies into the pinned ring are spread over a few threads, each
// with its own pair of ring slots, because one thread's memcpy (~12 GB/s) is slower than the PCIe link.
struct upload_job {
    uint8_t *dst;
    const uint8_t *src;
};

// true if the runtime knows `p` as page-locked host memory (hipHostMalloc / hipHostRegister): the DMA engine can
// read it directly
static bool is_pinned_host(const void *p)
{
    hipPointerAttribute_t a{};
    if (hipPointerGetAttributes(&a, p) != hipSuccess) {
        (void)hipGetLastError();  // an ordinary pageable pointer is reported as an error: clear it
        return false;
    }
    return a.type == hipMemoryTypeHost;
}

static int upload_many(ce_batch *b, const std::vector<upload_job> &jobs)
{
    ce_ctx *ctx = b->ctx;
    if (jobs.empty()) return CE_OK;
    CE_HIP(ctx, hipSetDevice(ctx->device));
    // Page-locked sources skip the staging ring: one asynchronous copy per image straight from the caller's buffer.
    // Only ce_eval_batch comes through here, and it collects (synchronises) before it returns, so the buffers
    // outlive the copies.
    bool all_pinned = true;
    for (const auto &j : jobs)
        if (!is_pinned_host(j.src)) {
            all_pinned = false;
            break;
        }
    // One stream moves a 786 KB image in 38 us (20.6 GB/s): the 1.88 GB of the Kodak + CID22 sweep would take as long as its
    // kernels.  The copies of a chunk therefore alternate between the batch's upload stream and a second one of the context,
    // which is fenced on both sides so that everything else keeps seeing "the uploads are on up_stream".
    static const int n_up = [] {
        const char *e = std::getenv("CE_UPLOAD_STREAMS");
        return e ? std::max(1, std::min(2, std::atoi(e))) : 2;  // 2000 pairs of 512x512, page-locked: 99.9 -> 95.4 ms; pageable: see r03_experiments 17
    }();
    const bool two_up = n_up == 2 && jobs.size() >= 16;
    auto up2_begin = [&]() -> int {
        if (!ctx->up2_stream) {
            CE_HIP(ctx, hipStreamCreateWithFlags(&ctx->up2_stream, hipStreamNonBlocking));
            CE_HIP(ctx, hipEventCreateWithFlags(&ctx->ev_up2, hipEventDisableTiming));
        }
        CE_HIP(ctx, hipEventRecord(ctx->ev_up2, b->up_stream));  // behind whatever up_stream already waits for
        CE_HIP(ctx, hipStreamWaitEvent(ctx->up2_stream, ctx->ev_up2, 0));
        return CE_OK;
    };
    auto up2_end = [&]() -> int {
        CE_HIP(ctx, hipEventRecord(ctx->ev_up2, ctx->up2_stream));
        CE_HIP(ctx, hipStreamWaitEvent(b->up_stream, ctx->ev_up2, 0));
        return CE_OK;
    };
    if (all_pinned) {
        if (b->run_pending) {
            CE_HIP(ctx, hipStreamWaitEvent(b->up_stream, b->ev_run, 0));
            b->run_pending = false;
        }
        if (two_up) {
            int rc = up2_begin();
            if (rc != CE_OK) return rc;
            for (size_t i = 0; i < jobs.size(); i++)
                CE_HIP(ctx, hipMemcpyAsync(jobs[i].dst, jobs[i].src, b->img_bytes, hipMemcpyHostToDevice, (i & 1) ? ctx->up2_stream : b->up_stream));
            rc = up2_end();
            if (rc != CE_OK) return rc;
        } else {
            for (const auto &j : jobs)
                CE_HIP(ctx, hipMemcpyAsync(j.dst, j.src, b->img_bytes, hipMemcpyHostToDevice, b->up_stream));
        }
        b->uploads_pending = true;
        return CE_OK;
    }
    const int n_threads = (int)std::min<size_t>({(size_t)ce_batch::kStages / 2, jobs.size(),
                                                 (size_t)std::max(1u, std::thread::hardware_concurrency())});
    if (n_threads <= 1 || b->img_bytes < (64u << 10)) {
        for (const auto &j : jobs) {
            int rc = upload(b, j.dst, j.src, false);
            if (rc != CE_OK) return rc;
        }
        return CE_OK;
    }
    if (b->run_pending) {
        CE_HIP(ctx, hipStreamWaitEvent(b->up_stream, b->ev_run, 0));
        b->run_pending = false;
    }
    if (two_up) {
        int rc = up2_begin();
        if (rc != CE_OK) return rc;
    }
    std::atomic<size_t> next{0};
    std::atomic<int> err{(int)hipSuccess};
    const int device = ctx->device;
    auto worker = [&](int t) {
        if (hipSetDevice(device) != hipSuccess) return;
        const hipStream_t us = (two_up && (t & 1)) ? ctx->up2_stream : b->up_stream;  // a worker's two ring slots stay on its stream
        int flip = 0;
        for (;;) {
            const size_t i = next.fetch_add(1);
            if (i >= jobs.size()) break;
            const int k = 2 * t + flip;
            flip ^= 1;
            hipError_t e = hipSuccess;
            if (b->stage_busy[k]) e = hipEventSynchronize(b->ev_stage[k]);
            if (e == hipSuccess) {
                std::memcpy(b->h_stage[k], jobs[i].src, b->img_bytes);
                e = hipMemcpyAsync(jobs[i].dst, b->h_stage[k], b->img_bytes, hipMemcpyHostToDevice, us);
            }
            if (e == hipSuccess) e = hipEventRecord(b->ev_stage[k], us);
            b->stage_busy[k] = true;
            if (e != hipSuccess) {
                err.store((int)e);
                break;
            }
        }
    };
    std::vector<std::thread> pool;
    for (int t = 1; t < n_threads; t++) {
        try {
            pool.emplace_back(worker, t);
        } catch (...) {  // no thread to be had: the calling thread's loop below takes whatever is left (nothing may be thrown across the C ABI)
            break;
        }
    }
    worker(0);
    for (auto &th : pool) th.join();
    b->uploads_pending = true;
    if (two_up) {
        int rc = up2_end();
        if (rc != CE_OK) return rc;
    }
    if (err.load() != (int)hipSuccess) {
        ctx->err = std::string("upload: ") + hipGetErrorString((hipError_t)err.load());
        return CE_ERR_BACKEND;
    }
    return CE_OK;
}

int ce_batch_set_reference(ce_batch *b, uint32_t ref_index, const uint8_t *rgb, size_t len)
{
    if (!b || !rgb) return CE_ERR_INVALID_ARG;
    if (ref_index >= b->max_refs) return fail(b->ctx, CE_ERR_INVALID_ARG, "ref_index out of range");
    if (len != b->img_bytes)
        return fail(b->ctx, CE_ERR_BAD_LENGTH, "Invalid image size: expected " + std::to_string(b->img_bytes) +
                                                    " bytes, got " + std::to_string(len));
    invalidate_reference_state(b);  // cached reference-side planes are stale
    return upload(b, b->d_refs + (size_t)ref_index * b->img_bytes, rgb);
}

// pixels in a decoder's format -> wide staging -> device -> ingest kernel writes the RGB8 slab slot
static int upload_fmt(ce_batch *b, uint8_t *dst, const void *pixels, size_t len, int format)
{
    ce_ctx *ctx = b->ctx;
    const size_t bpp = ce_pixel_bytes(format), n_px = (size_t)b->w * b->h;
    if (bpp == 0) return fail(ctx, CE_ERR_INVALID_ARG, "unknown pixel format");
    if (len != n_px * bpp)
        return fail(ctx, CE_ERR_BAD_LENGTH, "Invalid image size: expected " + std::to_string(n_px * bpp) + " bytes, got " +
                                                std::to_string(len));
    if (format == CE_PIXEL_RGB8) return upload(b, dst, static_cast<const uint8_t *>(pixels), false);
    CE_HIP(ctx, hipSetDevice(ctx->device));  // the staging allocations and the ingest launch below go to the context's device
    const int k = b->next_wide;
    b->next_wide ^= 1;
    if (!b->h_wide[k]) {
        CE_HIP(ctx, hipHostMalloc((void **)&b->h_wide[k], n_px * 8, hipHostMallocDefault));
        CE_HIP(ctx, hipMalloc((void **)&b->d_wide[k], n_px * 8));
        CE_HIP(ctx, hipEventCreateWithFlags(&b->ev_wide[k], hipEventDisableTiming));
    }
    if (b->run_pending) {
        CE_HIP(ctx, hipStreamWaitEvent(b->up_stream, b->ev_run, 0));
        b->run_pending = false;
    }
    if (b->wide_busy[k]) CE_HIP(ctx, hipEventSynchronize(b->ev_wide[k]));  // this staging pair's previous image has been converted
    std::memcpy(b->h_wide[k], pixels, len);
    CE_HIP(ctx, hipMemcpyAsync(b->d_wide[k], b->h_wide[k], len, hipMemcpyHostToDevice, b->up_stream));
    int rc = ce_launch_ingest(ctx, b->up_stream, format, b->d_wide[k], dst, n_px);
    if (rc != CE_OK) return rc;
    CE_HIP(ctx, hipEventRecord(b->ev_wide[k], b->up_stream));
    b->wide_busy[k] = true;
    b->uploads_pending = true;
    return CE_OK;
}

int ce_batch_set_reference_fmt(ce_batch *b, uint32_t ref_index, const void *pixels, size_t len, int format)
{
    if (!b || !pixels) return CE_ERR_INVALID_ARG;
    if (ref_index >= b->max_refs) return fail(b->ctx, CE_ERR_INVALID_ARG, "ref_index out of range");
    invalidate_reference_state(b);
    return upload_fmt(b, b->d_refs + (size_t)ref_index * b->img_bytes, pixels, len, format);
}

// ---- ICC -> sRGB colour tables ----------------------------------------------------------------------------------------
struct ce_lut {
    ce_ctx *ctx;
    uint32_t *d_table;  // [2^24] r | g << 8 | b << 16
};

int ce_lut_create(ce_ctx *ctx, const uint8_t *table, size_t table_len, ce_lut **out)
{
    if (!ctx || !table || !out) return CE_ERR_INVALID_ARG;
    *out = nullptr;
    const size_t want = (size_t)3 << 24;
    if (table_len != want)
        return fail(ctx, CE_ERR_BAD_LENGTH, "Invalid colour table size: expected " + std::to_string(want) + " bytes, got " + std::to_string(table_len));
    CE_HIP(ctx, hipSetDevice(ctx->device));
    uint8_t *d_packed = nullptr;
    uint32_t *d_table = nullptr;
    CE_HIP(ctx, hipMalloc(&d_packed, want));
    if (hipMalloc(&d_table, sizeof(uint32_t) << 24) != hipSuccess) {
        hipFree(d_packed);
        return fail(ctx, CE_ERR_BACKEND, "hipMalloc failed (colour table)");
    }
    int rc = CE_OK;
    if (hipMemcpyAsync(d_packed, table, want, hipMemcpyHostToDevice, ctx->stream) != hipSuccess) rc = fail(ctx, CE_ERR_BACKEND, "H2D failed (colour table)");
    if (rc == CE_OK) rc = ce_launch_lut_expand(ctx, ctx->stream, d_packed, d_table);
    if (rc == CE_OK && hipStreamSynchronize(ctx->stream) != hipSuccess) rc = fail(ctx, CE_ERR_BACKEND, "sync failed (colour table)");
    hipFree(d_packed);
    if (rc != CE_OK) {
        hipFree(d_table);
        return rc;
    }
    *out = new ce_lut{ctx, d_table};
    return CE_OK;
}

void ce_lut_destroy(ce_lut *lut)
{
    if (!lut) return;
    hipSetDevice(lut->ctx->device);
    hipStreamSynchronize(lut->ctx->stream);
    hipFree(lut->d_table);
    delete lut;
}

// the table runs on the batch's upload stream, behind the copy (and the format conversion) of the same image
static int apply_lut(ce_batch *b, uint8_t *slot, const ce_lut *lut)
{
    if (!lut) return CE_OK;
    if (lut->ctx->device != b->ctx->device) return fail(b->ctx, CE_ERR_INVALID_ARG, "colour table and batch are on different devices");
    // the table was built on its context's stream and ce_lut_create synchronised: it is complete
    return ce_launch_lut_apply(b->ctx, b->up_stream, slot, lut->d_table, (size_t)b->w * b->h);
}

int ce_batch_set_reference_lut(ce_batch *b, uint32_t ref_index, const void *pixels, size_t len, int format, const ce_lut *lut)
{
    int rc = ce_batch_set_reference_fmt(b, ref_index, pixels, len, format);
    if (rc != CE_OK) return rc;
    return apply_lut(b, b->d_refs + (size_t)ref_index * b->img_bytes, lut);
}

int ce_batch_set_test_lut(ce_batch *b, uint32_t pair_index, uint32_t ref_index, const void *pixels, size_t len, int format,
                          const ce_lut *lut)
{
    int rc = ce_batch_set_test_fmt(b, pair_index, ref_index, pixels, len, format);
    if (rc != CE_OK) return rc;
    return apply_lut(b, b->d_tests + (size_t)pair_index * b->img_bytes, lut);
}

int ce_batch_bind_pair(ce_batch *b, uint32_t pair_index, uint32_t ref_index)
{
    if (!b) return CE_ERR_INVALID_ARG;
    if (pair_index >= b->max_pairs || ref_index >= b->max_refs)
        return fail(b->ctx, CE_ERR_INVALID_ARG, "pair/ref index out of range");
    if (b->h_pair_ref[pair_index] != ref_index) {
        b->h_pair_ref[pair_index] = ref_index;
        b->pair_ref_dirty = true;
        b->pair_ref_version++;  // device-side tables derived from it (pair_ref, XCD work lists) are rebuilt at the next launch
    }
    return CE_OK;
}

int ce_batch_set_test(ce_batch *b, uint32_t pair_index, uint32_t ref_index, const uint8_t *rgb, size_t len)
{
    if (!b || !rgb) return CE_ERR_INVALID_ARG;
    int rc = ce_batch_bind_pair(b, pair_index, ref_index);
    if (rc != CE_OK) return rc;
    if (len != b->img_bytes)
        return fail(b->ctx, CE_ERR_BAD_LENGTH, "Invalid image size: expected " + std::to_string(b->img_bytes) +
                                                    " bytes, got " + std::to_string(len));
    return upload(b, b->d_tests + (size_t)pair_index * b->img_bytes, rgb);
}

int ce_batch_set_test_fmt(ce_batch *b, uint32_t pair_index, uint32_t ref_index, const void *pixels, size_t len, int format)
{
    if (!b || !pixels) return CE_ERR_INVALID_ARG;
    int rc = ce_batch_bind_pair(b, pair_index, ref_index);
    if (rc != CE_OK) return rc;
    return upload_fmt(b, b->d_tests + (size_t)pair_index * b->img_bytes, pixels, len, format);
}

void *ce_batch_reference_slab(ce_batch *b)
{
    if (!b) return nullptr;
    invalidate_reference_state(b);  // the caller may overwrite references behind our back
    return b->d_refs;
}
void *ce_batch_test_slab(ce_batch *b) { return b ? b->d_tests : nullptr; }

int ce_batch_launch(ce_batch *b, uint32_t n_pairs, uint32_t metric_mask, uint32_t flags, float intensity_target)
{
    if (!b) return CE_ERR_INVALID_ARG;
    ce_ctx *ctx = b->ctx;
    if (n_pairs == 0 || n_pairs > b->max_pairs) return fail(ctx, CE_ERR_INVALID_ARG, "n_pairs out of range");
    const uint32_t known = CE_METRIC_DSSIM | CE_METRIC_SSIMULACRA2 | CE_METRIC_BUTTERAUGLI | CE_METRIC_PSNR;
    if (metric_mask & ~known) return fail(ctx, CE_ERR_INVALID_ARG, "unknown metric bit");
    CE_HIP(ctx, hipSetDevice(ctx->device));
    {
        int rc = flush_uploads(b);
        if (rc != CE_OK) return rc;
    }
    if (b->pair_ref_dirty) {
        // pair_first[p] = the lowest pair index bound to the same reference as p: that pair's row pass also produces
        // the two reference-only blur streams (a, a*a) which every pair of the reference then reads (ssim2.hip)
        // ... and the inverse table, reference -> its pairs in ascending pair order (kernels that walk a reference's
        // distorted images with the reference's planes held in registers: dssim.hip)
        const size_t P = b->max_pairs, R = b->max_refs;
        std::vector<uint32_t> table(3 * P + R + 1), first_of(R, ~0u), count(R + 1, 0);
        for (uint32_t i = 0; i < P; i++) {
            const uint32_t r = b->h_pair_ref[i];
            if (first_of[r] == ~0u) first_of[r] = i;
            table[i] = r;
            table[P + i] = first_of[r];
            count[r + 1]++;
        }
        for (size_t r = 0; r < R; r++) count[r + 1] += count[r];
        for (size_t r = 0; r <= R; r++) table[2 * P + r] = count[r];
        std::vector<uint32_t> fill(count.begin(), count.end() - 1);
        for (uint32_t i = 0; i < P; i++) table[2 * P + R + 1 + fill[b->h_pair_ref[i]]++] = i;
        if (int rc = ce_upload_table(b, b->d_pair_ref, table.data(), sizeof(uint32_t) * table.size())) return rc;  // `table` is pageable
        b->pair_ref_dirty = false;
    }
    uint32_t n_refs_used = 0;
    for (uint32_t i = 0; i < n_pairs; i++) n_refs_used = std::max(n_refs_used, b->h_pair_ref[i] + 1);

    const uint8_t *d_refs = b->d_refs;
    if (flags & CE_FLAG_XYB_ROUNDTRIP) {
        // MetricConfig::xyb_roundtrip: every metric sees the roundtripped reference (session.rs:447-456)
        if (!b->d_refs_rt) CE_HIP(ctx, hipMalloc(&b->d_refs_rt, b->img_bytes * b->max_refs + 16));
        if (!(b->keep_ref_pyramid && b->refs_rt_valid)) {
            int rc = ce_launch_xyb_roundtrip(ctx, b->d_refs, b->d_refs_rt, (size_t)n_refs_used * b->w * b->h);
            if (rc != CE_OK) return rc;
            b->refs_rt_valid = true;
        }
        d_refs = b->d_refs_rt;
    }
    // The three perceptual metrics are independent chains over their own buffers (PSNR - two short launches on the
    // context's stream - is enqueued after them, so that forked chains do not wait behind it).
    const bool run_ssim2 = (metric_mask & CE_METRIC_SSIMULACRA2) && b->w >= 8 && b->h >= 8;
    const bool run_dssim = (metric_mask & CE_METRIC_DSSIM) != 0;
    const bool run_ba = (metric_mask & CE_METRIC_BUTTERAUGLI) && b->w >= 8 && b->h >= 8;
    // Side by side or back to back?  Measured (profiles/r02_experiments.md sections 1, 12, 15): a SMALL batch is bound by
    // the latency of its ~130 dependent launches, and three chains side by side hide each other's gaps (one Kodak pair
    // 0.76 -> 0.54 ms, eight 1.37 -> 1.23 ms, one 4K pair 3.65 -> 2.94 ms); a LARGE grid fills the GPU from one chain,
    // and with other batches in flight beside it (two shape buckets, two steps) forked chains evict each other's
    // reference planes from L2 and halve each other's resident workgroups (Kodak grid 6.55 -> 7.55 ms per step).  So the
    // chains fork when the batch holds at most CE_METRIC_FORK_BELOW_MP megapixels of pairs (default 4), or at most
    // CE_METRIC_FORK_ALONE_BELOW_MP (default 64) while no other batch of this device is launched and uncollected, and run
    // back to back on the context's stream otherwise (SSIMULACRA2 still overlaps its level-0 passes with its tail
    // levels).  Scores do not depend on the schedule.
    // fork mask: bit k = metric chain k (0 SSIMULACRA2, 1 DSSIM, 2 Butteraugli) runs on its own stream beside the others.
    // CE_METRIC_STREAMS = "fork" (all three, always), "fork:dssim", "fork:ssim2,ba", ... (those, always), "serial" (never);
    // unset / "auto" = by size.
    static const int fork_mask_env = [] {
        const char *e = std::getenv("CE_METRIC_STREAMS");
        if (!e || std::strcmp(e, "auto") == 0) return -1;
        if (std::strncmp(e, "fork", 4) != 0) return 0;
        if (e[4] != ':') return 7;
        int m = 0;
        if (std::strstr(e + 5, "ssim2")) m |= 1;
        if (std::strstr(e + 5, "dssim")) m |= 2;
        if (std::strstr(e + 5, "ba")) m |= 4;
        return m;
    }();
    static const double fork_below_mp = [] {
        const char *e = std::getenv("CE_METRIC_FORK_BELOW_MP");
        return e ? std::atof(e) : CE_DEFAULT_FORK_BELOW_MP;
    }();
    static const double fork_alone_below_mp = [] {
        const char *e = std::getenv("CE_METRIC_FORK_ALONE_BELOW_MP");
        return e ? std::atof(e) : CE_DEFAULT_FORK_ALONE_BELOW_MP;
    }();
    const double batch_mp = (double)n_pairs * b->w * b->h * 1e-6;
    const int others = in_flight_of(ctx).load(std::memory_order_relaxed) - (b->counted_in_flight ? 1 : 0);
    const bool by_size = batch_mp <= fork_below_mp || (others <= 0 && batch_mp <= fork_alone_below_mp);
    const unsigned fork_wanted = fork_mask_env >= 0 ? (unsigned)fork_mask_env : (by_size ? 7u : 0u);
    const unsigned fork_mask = (!ctx->prof_serial && (int)run_ssim2 + (int)run_dssim + (int)run_ba > 1) ? fork_wanted : 0u;
    hipStream_t base = ctx->stream;
    if (fork_mask) {
        if (!b->ev_fork) CE_HIP(ctx, hipEventCreateWithFlags(&b->ev_fork, hipEventDisableTiming));
        CE_HIP(ctx, hipEventRecord(b->ev_fork, base));
    }
    unsigned joined = 0;
    auto prepare_fork = [&](int k) -> int {  // the chain's stream exists and waits for the fork point
        if (!b->metric_stream[k]) {
            if (!(b->metric_stream[k] = ce_ctx_aux_stream(ctx, ce_ctx::AUX_METRIC0 + k))) return CE_ERR_BACKEND;
            CE_HIP(ctx, hipEventCreateWithFlags(&b->ev_join[k], hipEventDisableTiming));
        }
        CE_HIP(ctx, hipStreamWaitEvent(b->metric_stream[k], b->ev_fork, 0));
        return CE_OK;
    };
    auto launch_metric = [&](int k) -> int {
        return k == 0   ? ce_launch_ssim2(b, d_refs, n_refs_used, n_pairs)
               : k == 1 ? ce_launch_dssim(b, d_refs, n_refs_used, n_pairs)
                        : ce_launch_butteraugli(b, d_refs, n_refs_used, n_pairs, intensity_target);
    };
    const bool runs[3] = {run_ssim2, run_dssim, run_ba};
    // launch order of the chains (0 SSIMULACRA2, 1 DSSIM, 2 Butteraugli); CE_FORK_ORDER=<permutation> for A/B runs of
    // the forked schedule (profiles/r02_experiments.md section 19)
    static const std::array<int, 3> fork_order = [] {
        std::array<int, 3> o{0, 1, 2};
        const char *e = std::getenv("CE_FORK_ORDER");
        if (e && std::strlen(e) == 3) {
            std::array<int, 3> t{e[0] - '0', e[1] - '0', e[2] - '0'};
            if ((1 << t[0] | 1 << t[1] | 1 << t[2]) == 7 && t[0] >= 0 && t[1] >= 0 && t[2] >= 0) o = t;
        }
        return o;
    }();
    // A forked SMALL batch is bound by the host: the ~50 launches of the three chains take ~0.2 ms to enqueue one after
    // the other, and a chain cannot start before its first launch is enqueued.  So the chains of a fully forked batch are
    // enqueued by one host thread each (the caller's + two helpers; CE_STREAM makes a launch function enqueue on its
    // thread's chain stream); CE_FORK_THREADS=0 keeps the single-threaded enqueue for A/B runs.  Per-kernel event timing
    // (ce_prof_*) keeps its bookkeeping on one thread.
    static const bool fork_threads = [] { const char *e = std::getenv("CE_FORK_THREADS"); return !(e && e[0] == '0'); }();
    if (fork_mask == 7u && fork_threads && !ctx->prof) {
        int rcs[3] = {CE_OK, CE_OK, CE_OK};
        std::string errs[3];
        int last = -1;
        for (int i = 0; i < 3; i++)
            if (runs[fork_order[i]]) {
                int rc = prepare_fork(fork_order[i]);
                if (rc != CE_OK) return rc;
                last = fork_order[i];
            }
        auto body = [&](int k) {
            ce_tls_stream = b->metric_stream[k];
            rcs[k] = launch_metric(k);
            ce_tls_stream = nullptr;
            if (rcs[k] == CE_OK && hipEventRecord(b->ev_join[k], b->metric_stream[k]) != hipSuccess) rcs[k] = CE_ERR_BACKEND;
        };
        if (!ctx->helpers) {
            try {
                ctx->helpers = new ce_fork_helpers(ctx->device);
            } catch (...) {  // no threads to be had: every chain is enqueued here (nothing may be thrown across the C ABI)
                ctx->helpers = nullptr;
            }
        }
        int used = 0;
        for (int i = 0; i < 3; i++) {
            const int k = fork_order[i];
            if (!runs[k] || k == last) continue;
            if (ctx->helpers && used < 2)
                ctx->helpers->submit(used++, [&body, k] { body(k); });
            else
                body(k);
        }
        body(last);  // the caller's thread takes the chain that is enqueued last in the single-threaded order
        for (int i = 0; i < used; i++) ctx->helpers->wait(i);
        for (int k = 0; k < 3; k++) {
            if (!runs[k]) continue;
            if (rcs[k] != CE_OK) return rcs[k];
            joined |= 1u << k;
        }
    } else {
        for (int i = 0; i < 3; i++) {
            const int k = fork_mask == 7u ? fork_order[i] : i;
            if (!runs[k]) continue;
            if (!(fork_mask & (1u << k))) {
                int rc = launch_metric(k);
                if (rc != CE_OK) return rc;
                continue;
            }
            int rc = prepare_fork(k);
            if (rc != CE_OK) return rc;
            ce_tls_stream = b->metric_stream[k];  // the chain's launches go to its own stream
            rc = launch_metric(k);
            ce_tls_stream = nullptr;
            if (rc != CE_OK) return rc;
            CE_HIP(ctx, hipEventRecord(b->ev_join[k], b->metric_stream[k]));
            joined |= 1u << k;  // the context's stream waits for it after every chain has been launched
        }
    }
    if (metric_mask & CE_METRIC_PSNR) {
        int rc = ce_launch_psnr(b, d_refs, n_pairs);
        if (rc != CE_OK) return rc;
    }
    for (int k = 0; k < 3; k++)
        if (joined & (1u << k)) CE_HIP(ctx, hipStreamWaitEvent(base, b->ev_join[k], 0));
    b->last_n_pairs = n_pairs;
    b->last_mask = metric_mask;
    // the scores come back behind the last kernel of THIS launch and ev_run marks them: ce_batch_collect waits for the event,
    // not for the stream (round 2 copied at collect time and drained the context's stream, so collecting one batch waited
    // for every batch launched after it - in ce_eval_batch the next chunk's upload then started only when the device was idle)
    CE_HIP(ctx, hipMemcpyAsync(b->h_scores, b->d_scores, sizeof(ce_dev_scores) * n_pairs, hipMemcpyDeviceToHost, ctx->stream));
    CE_HIP(ctx, hipEventRecord(b->ev_run, ctx->stream));
    b->run_pending = true;
    if (!b->counted_in_flight) {
        in_flight_of(ctx).fetch_add(1, std::memory_order_relaxed);
        b->counted_in_flight = true;
    }
    return CE_OK;
}

int ce_batch_collect(ce_batch *b, uint32_t n_pairs, ce_scores *out)
{
    if (!b || !out) return CE_ERR_INVALID_ARG;
    ce_ctx *ctx = b->ctx;
    if (n_pairs == 0 || n_pairs > b->max_pairs) return fail(ctx, CE_ERR_INVALID_ARG, "n_pairs out of range");
    CE_HIP(ctx, hipSetDevice(ctx->device));
    if (n_pairs > b->last_n_pairs) return fail(ctx, CE_ERR_INVALID_ARG, "collect asks for more pairs than the last launch ran");
    CE_HIP(ctx, hipEventSynchronize(b->ev_run));  // the launch's kernels and the copy of its scores into h_scores (ce_batch_launch)
    b->run_pending = false;
    leave_flight(b);
    const uint32_t mask = b->last_mask;
    for (uint32_t i = 0; i < n_pairs; i++) {
        ce_scores s{};
        s.status = CE_OK;
        const ce_dev_scores &d = b->h_scores[i];
        if (mask & CE_METRIC_PSNR) {
            s.psnr = psnr_from_sse(d.sse, b->w, b->h);
            s.valid |= CE_METRIC_PSNR;
        }
        if (mask & CE_METRIC_DSSIM) {
            s.dssim = d.dssim;
            s.valid |= CE_METRIC_DSSIM;
        }
        if (mask & CE_METRIC_BUTTERAUGLI) {
            if (b->w < 8 || b->h < 8) {  // "minimum 8x8 for butteraugli", src/eval/helpers.rs:89
                s.status = CE_ERR_TOO_SMALL;
            } else {
                s.butteraugli = d.butteraugli;
                s.valid |= CE_METRIC_BUTTERAUGLI;
            }
        }
        if (mask & CE_METRIC_SSIMULACRA2) {
            if (b->w < 8 || b->h < 8) {
                s.status = CE_ERR_TOO_SMALL;
            } else {
                s.ssimulacra2 = d.ssimulacra2;
                s.valid |= CE_METRIC_SSIMULACRA2;
            }
        }
        out[i] = s;
    }
    return CE_OK;
}

int ce_batch_butteraugli_pnorm3(ce_batch *b, uint32_t n_pairs, double *out)
{
    if (!b || !out || !b->ba_ready || n_pairs == 0 || n_pairs > b->max_pairs) return CE_ERR_INVALID_ARG;
    ce_ctx *ctx = b->ctx;
    CE_HIP(ctx, hipSetDevice(ctx->device));
    CE_HIP(ctx, hipStreamSynchronize(ctx->stream));
    CE_HIP(ctx, hipMemcpy(out, b->ba_pnorm, sizeof(double) * n_pairs, hipMemcpyDeviceToHost));
    return CE_OK;
}

int ce_batch_run(ce_batch *b, uint32_t n_pairs, uint32_t metric_mask, uint32_t flags, float intensity_target,
                 ce_scores *out)
{
    int rc = ce_batch_launch(b, n_pairs, metric_mask, flags, intensity_target);
    if (rc != CE_OK) return rc;
    return ce_batch_collect(b, n_pairs, out);
}

// ---- single pair / mixed batch ----------------------------------------------------------------

// Device bytes a batch of this shape needs once the metrics in `mask` have run (working sets are allocated lazily,
// per metric).  Mirrors ce_ssim2_prepare / dssim_prepare / ba_prepare, with the plane padding folded into one factor.
size_t ce_estimate_batch_bytes(uint32_t w, uint32_t h, uint32_t n_refs, uint32_t n_pairs, uint32_t metric_mask)
{
    const double px = (double)w * h, slots = (double)n_refs + n_pairs, pairs = n_pairs;
    double bytes = 3.0 * px * slots + 64.0 * pairs;  // u8 slabs, scores
    double allocations = 8;                          // each device allocation is rounded up; budget 256 KiB of slack apiece
    if (metric_mask & CE_METRIC_SSIMULACRA2) {  // linear pyramid (levels >= 1) 4, XYB pyramid 16 per slot; 15 row-blurred planes x 1.333 per pair
        bytes += px * (20.0 * slots + 80.0 * pairs);
        allocations += 24;
    }
    if (metric_mask & CE_METRIC_DSSIM) {  // linear ping-pong 6 per slot; img 12 + the SSIM maps of all levels 5.4 per pair; the references' per-level img / mu / sq 48
        bytes += px * (6.0 * slots + 48.0 * n_refs + 17.4 * pairs);
        allocations += 24;
    }
    if (metric_mask & CE_METRIC_BUTTERAUGLI) {  // PsychoImage 50, mask input 5, three 3-plane scratch sets 36 per slot; mask values 10 per reference; half-resolution diffmap 1 per pair
        bytes += px * (91.0 * slots + 10.0 * n_refs + 1.0 * pairs);
        allocations += 20;
    }
    if (metric_mask & CE_METRIC_PSNR) bytes += 8.0 * pairs;
    return (size_t)(bytes * 1.2) + (size_t)(allocations * (256u << 10)) + (8u << 20);  // row / pitch padding of the planar buffers
}

int ce_ctx_memory_info(ce_ctx *ctx, size_t *free_bytes, size_t *total_bytes)
{
    if (!ctx || !free_bytes || !total_bytes) return CE_ERR_INVALID_ARG;
    CE_HIP(ctx, hipSetDevice(ctx->device));
    CE_HIP(ctx, hipMemGetInfo(free_bytes, total_bytes));
    return CE_OK;
}

int ce_host_alloc(ce_ctx *ctx, size_t bytes, void **out)
{
    if (!ctx || !out || bytes == 0) return CE_ERR_INVALID_ARG;
    *out = nullptr;
    CE_HIP(ctx, hipSetDevice(ctx->device));
    CE_HIP(ctx, hipHostMalloc(out, bytes, hipHostMallocDefault));
    return CE_OK;
}

int ce_host_free(ce_ctx *ctx, void *p)
{
    if (!p) return CE_OK;
    const hipError_t e = hipHostFree(p);
    if (e != hipSuccess) {
        if (ctx) ctx->err = std::string("hipHostFree: ") + hipGetErrorString(e);
        return CE_ERR_BACKEND;
    }
    return CE_OK;
}

// bytes one ce_eval_batch chunk may allocate: CE_EVAL_BATCH_BYTES if set (tests), else a share of what is free now
static size_t chunk_budget(ce_ctx *ctx)
{
    if (const char *e = std::getenv("CE_EVAL_BATCH_BYTES")) {
        const long long v = std::atoll(e);
        if (v > 0) return (size_t)v;
    }
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) return (size_t)8 << 30;
    // the pooled batches of this context are reused, so what they hold already counts as available
    size_t pooled = 0;
    for (auto &kv : ctx->shape_pool) {
        const ce_batch *pb = kv.second;
        const uint32_t held = (pb->ssim2_ready ? CE_METRIC_SSIMULACRA2 : 0u) | (pb->dssim_ready ? CE_METRIC_DSSIM : 0u) |
                              (pb->ba_ready ? CE_METRIC_BUTTERAUGLI : 0u);
        pooled += ce_estimate_batch_bytes(pb->w, pb->h, pb->max_refs, pb->max_pairs, held);
    }
    // ... and no chunk asks for more than kChunkBytesMax: on a 288 GB device a third of the free memory is a 70+ GB batch whose
    // hipMalloc calls alone take seconds (measured: 2000 pairs of 512x512, three metrics - first call 3.7 s + 6.5 s for the
    // second with 73 GB chunks, 0.19 s with 48 GiB ones; steady state 128 ms vs 139 ms; profiles/r03_experiments.md §17)
    const size_t share = (size_t)((double)(free_b + pooled) * 0.8 / ce_ctx::kPoolRing);
    return std::min(share, ce_ctx::kChunkBytesMax);
}

// free every pooled batch that has nothing in flight (called when a new one does not fit)
static void pool_evict_idle(ce_ctx *ctx)
{
    for (auto it = ctx->shape_pool.begin(); it != ctx->shape_pool.end();) {
        if (!it->second->run_pending && !it->second->uploads_pending) {
            ce_batch_destroy(it->second);
            it = ctx->shape_pool.erase(it);
        } else {
            ++it;
        }
    }
}

static int shape_batch(ce_ctx *ctx, uint32_t w, uint32_t h, uint32_t need_pairs, uint32_t ring_slot, ce_batch **out)
{
    auto key = std::make_tuple(w, h, ring_slot);
    auto it = ctx->shape_pool.find(key);
    if (it != ctx->shape_pool.end() && it->second->max_pairs >= need_pairs) {
        *out = it->second;
        return CE_OK;
    }
    if (it != ctx->shape_pool.end()) {
        ce_batch_destroy(it->second);
        ctx->shape_pool.erase(it);
    }
    ce_batch *b = nullptr;
    int rc = ce_batch_create(ctx, w, h, need_pairs, need_pairs, &b);
    if (rc == CE_ERR_BACKEND) {  // out of device memory: drop the idle pooled batches of other shapes / sizes and retry once
        pool_evict_idle(ctx);
        rc = ce_batch_create(ctx, w, h, need_pairs, need_pairs, &b);
    }
    if (rc != CE_OK) return rc;
    ctx->shape_pool[key] = b;
    *out = b;
    return CE_OK;
}

int ce_eval_batch(ce_ctx *ctx, size_t n, const ce_pair_desc *pairs, uint32_t metric_mask, uint32_t flags,
                  float intensity_target, ce_scores *out)
{
    return ce_eval_batch_lut(ctx, n, pairs, nullptr, metric_mask, flags, intensity_target, out);
}

int ce_eval_batch_lut(ce_ctx *ctx, size_t n, const ce_pair_desc *pairs, const ce_lut *const *test_luts, uint32_t metric_mask,
                      uint32_t flags, float intensity_target, ce_scores *out)
{
    if (!ctx || (!pairs && n) || (!out && n)) return CE_ERR_INVALID_ARG;
    // bucket by shape (Kodak mixes 768x512 and 512x768); invalid items never reach the device
    std::map<std::pair<uint32_t, uint32_t>, std::vector<size_t>> buckets;
    for (size_t i = 0; i < n; i++) {
        out[i] = ce_scores{};
        const ce_pair_desc &d = pairs[i];
        int rc = (!d.reference || !d.test || d.width == 0 || d.height == 0)
                     ? CE_ERR_INVALID_ARG
                     : validate_pair(ctx, d.reference_len, d.test_len, d.width, d.height);
        if (rc != CE_OK) {
            out[i].status = rc;
            continue;
        }
        buckets[{d.width, d.height}].push_back(i);
    }
    // Phase 1: fill and launch.  A bucket is streamed through up to kPoolRing pooled batches in chunks of whole
    // references (all pairs of a reference stay together, identical reference pointers share one device slot): the
    // uploads of a chunk run on that batch's upload stream and overlap the kernels of the chunk before it.
    // Phase 2: collect, in launch order.
    struct chunk {
        ce_batch *b;
        std::vector<size_t> items;  // indices into pairs[] / out[]
    };
    std::vector<chunk> chunks;
    uint32_t ring = 0;
    for (auto &kv : buckets) {
        const std::vector<size_t> &idx = kv.second;
        // group the bucket's items by reference pointer, first-appearance order
        std::map<const uint8_t *, size_t> group_of;
        std::vector<std::vector<size_t>> groups;
        for (size_t i : idx) {
            auto it = group_of.find(pairs[i].reference);
            if (it == group_of.end()) {
                group_of[pairs[i].reference] = groups.size();
                groups.emplace_back();
                groups.back().push_back(i);
            } else {
                groups[it->second].push_back(i);
            }
        }
        // How many chunks a bucket that fits is cut into (the upload of one chunk then overlaps the kernels of the one
        // before).  Measured on the 54-pair Kodak bucket with three metrics (round 2): one chunk 9.1 ms per grid, two 12.7,
        // three 10.1 - small launches cost more than the hidden upload saves - so a bucket is only cut once it holds at
        // least 64 pairs per chunk; buckets of different shapes still overlap (each has its own batch and upload stream).
        // CE_EVAL_BATCH_CHUNKS (1..3) forces the count for A/B runs.
        static const size_t forced_chunks = [] {
            const char *e = std::getenv("CE_EVAL_BATCH_CHUNKS");
            const int v = e ? std::atoi(e) : 0;
            return (size_t)(v >= 1 && v <= (int)ce_ctx::kPoolRing ? v : 0);
        }();
        const size_t n_chunks = forced_chunks ? std::min<size_t>(forced_chunks, std::max<size_t>(1, idx.size()))
                                              : std::min<size_t>(ce_ctx::kPoolRing, std::max<size_t>(1, idx.size() / 64));
        size_t target = (idx.size() + n_chunks - 1) / n_chunks;
        // ... and a chunk must fit the device: cap the pairs per chunk by bytes per pair (every pair budgeted with a
        // reference of its own) against a share of the free memory; a grid larger than that streams through the ring
        // in more chunks.  A reference with more tests than the cap is split (its reference is uploaded once per part).
        {
            const size_t per_pair = ce_estimate_batch_bytes(kv.first.first, kv.first.second, 1, 2, metric_mask) -
                                    ce_estimate_batch_bytes(kv.first.first, kv.first.second, 1, 1, metric_mask) +
                                    ce_estimate_batch_bytes(kv.first.first, kv.first.second, 2, 1, metric_mask) -
                                    ce_estimate_batch_bytes(kv.first.first, kv.first.second, 1, 1, metric_mask);  // a pair with a reference of its own
            const size_t cap = std::max<size_t>(1, chunk_budget(ctx) / std::max<size_t>(per_pair, 1));
            target = std::min(target, cap);
            // a pooled batch of this shape that is a little smaller than today's target (the estimate above and what
            // hipMemGetInfo reports move by a few per cent between calls) is used as it is rather than reallocated
            {
                auto it0 = ctx->shape_pool.find(std::make_tuple(kv.first.first, kv.first.second, 0u));
                if (it0 != ctx->shape_pool.end() && it0->second->max_pairs < target && (size_t)it0->second->max_pairs * 4 >= target * 3)
                    target = it0->second->max_pairs;
            }
            std::vector<std::vector<size_t>> split;
            for (auto &g : groups)
                for (size_t o = 0; o < g.size(); o += target)
                    split.emplace_back(g.begin() + o, g.begin() + std::min(g.size(), o + target));
            groups.swap(split);
        }
        // The upload of the FIRST chunk of a call overlaps nothing, so a bucket that needs several chunks starts with a
        // small one and grows geometrically up to the target (an upload costs about half of what the kernels of the same
        // pairs do, so each chunk's kernels still cover the upload of the next, twice as large).  CE_EVAL_BATCH_RAMP = pairs
        // of the first chunk (0 = every chunk at the target).
        static const size_t ramp0 = [] {
            const char *e = std::getenv("CE_EVAL_BATCH_RAMP");
            return (size_t)(e ? std::max(0, std::atoi(e)) : 64);  // 2000 pairs of 512x512: off 106.5 ms, 32 -> 102.1, 64 -> 100.0, 128 -> 101.1
        }();
        const bool several = idx.size() > target;
        size_t limit = (several && ramp0 && ring == 0) ? std::min(ramp0, target) : target;
        size_t g0 = 0;
        while (g0 < groups.size()) {
            size_t g1 = g0, count = 0;
            while (g1 < groups.size() && (count == 0 || count + groups[g1].size() <= limit)) count += groups[g1++].size();
            limit = std::min(target, limit * 2);
            // a ring slot may still be in flight from an earlier chunk of this call: collect it first
            const uint32_t slot = ring++ % ce_ctx::kPoolRing;
            for (auto &c : chunks)
                if (c.b && std::get<2>(c.b->pool_key) == slot && std::get<0>(c.b->pool_key) == kv.first.first &&
                    std::get<1>(c.b->pool_key) == kv.first.second && !c.items.empty() && c.b->run_pending) {
                    std::vector<ce_scores> tmp(c.items.size());
                    int rc = ce_batch_collect(c.b, (uint32_t)c.items.size(), tmp.data());
                    if (rc != CE_OK) return rc;
                    for (size_t k = 0; k < c.items.size(); k++) out[c.items[k]] = tmp[k];
                    c.b = nullptr;  // collected
                }
            ce_batch *b = nullptr;
            // (a bucket cut into several chunks sizes every ring slot for the target: the small first chunks do not make a
            // slot that a later, larger chunk of the same call would have to reallocate)
            int rc = shape_batch(ctx, kv.first.first, kv.first.second, (uint32_t)(several ? std::max(count, target) : count), slot, &b);
            if (rc != CE_OK) return rc;
            b->pool_key = std::make_tuple(kv.first.first, kv.first.second, slot);
            chunk ch{b, {}};
            std::vector<upload_job> jobs;
            uint32_t k = 0;
            for (size_t g = g0; g < g1; g++) {
                const uint32_t ref_slot = (uint32_t)(g - g0);
                jobs.push_back({b->d_refs + (size_t)ref_slot * b->img_bytes, pairs[groups[g][0]].reference});
                for (size_t i : groups[g]) {
                    rc = ce_batch_bind_pair(b, k, ref_slot);
                    if (rc != CE_OK) return rc;
                    jobs.push_back({b->d_tests + (size_t)k * b->img_bytes, pairs[i].test});
                    ch.items.push_back(i);
                    k++;
                }
            }
            invalidate_reference_state(b);
            rc = upload_many(b, jobs);
            if (rc != CE_OK) return rc;
            if (test_luts)  // ICC -> sRGB of the decoded images, on the upload stream behind their copies (icc.rs:69-103)
                for (uint32_t kk = 0; kk < k; kk++)
                    if (const ce_lut *lut = test_luts[ch.items[kk]]) {
                        rc = apply_lut(b, b->d_tests + (size_t)kk * b->img_bytes, lut);
                        if (rc != CE_OK) return rc;
                    }
            rc = ce_batch_launch(b, k, metric_mask, flags, intensity_target);
            if (rc != CE_OK) return rc;
            chunks.push_back(std::move(ch));
            g0 = g1;
        }
    }
    for (auto &c : chunks) {
        if (!c.b) continue;  // already collected when its ring slot was reused
        std::vector<ce_scores> tmp(c.items.size());
        int rc = ce_batch_collect(c.b, (uint32_t)c.items.size(), tmp.data());
        if (rc != CE_OK) return rc;
        for (size_t k = 0; k < c.items.size(); k++) out[c.items[k]] = tmp[k];
    }
    return CE_OK;
}

int ce_eval_pair(ce_ctx *ctx, const uint8_t *reference, size_t reference_len, const uint8_t *test, size_t test_len,
                 uint32_t width, uint32_t height, uint32_t metric_mask, uint32_t flags, float intensity_target,
                 ce_scores *out)
{
    if (!ctx || !out || !reference || !test) return CE_ERR_INVALID_ARG;
    ce_pair_desc d{reference, reference_len, test, test_len, width, height};
    int rc = ce_eval_batch(ctx, 1, &d, metric_mask, flags, intensity_target, out);
    if (rc != CE_OK) return rc;
    return out->status;
}

static int leaf(ce_ctx *ctx, const uint8_t *reference, size_t reference_len, const uint8_t *test, size_t test_len,
                size_t width, size_t height, uint32_t metric, float intensity, double *out)
{
    if (!ctx || !out) return CE_ERR_INVALID_ARG;
    ce_scores s{};
    int rc = ce_eval_pair(ctx, reference, reference_len, test, test_len, (uint32_t)width, (uint32_t)height, metric, 0,
                          intensity, &s);
    if (rc != CE_OK) return rc;
    *out = metric == CE_METRIC_PSNR          ? s.psnr
           : metric == CE_METRIC_SSIMULACRA2 ? s.ssimulacra2
           : metric == CE_METRIC_DSSIM       ? s.dssim
                                             : s.butteraugli;
    return CE_OK;
}

int ce_calculate_psnr(ce_ctx *ctx, const uint8_t *reference, size_t reference_len, const uint8_t *test,
                      size_t test_len, size_t width, size_t height, double *out)
{
    return leaf(ctx, reference, reference_len, test, test_len, width, height, CE_METRIC_PSNR, 0.f, out);
}

int ce_calculate_ssimulacra2(ce_ctx *ctx, const uint8_t *reference, size_t reference_len, const uint8_t *test,
                             size_t test_len, size_t width, size_t height, double *out)
{
    return leaf(ctx, reference, reference_len, test, test_len, width, height, CE_METRIC_SSIMULACRA2, 0.f, out);
}

int ce_calculate_dssim(ce_ctx *ctx, const uint8_t *reference, size_t reference_len, const uint8_t *test,
                       size_t test_len, size_t width, size_t height, double *out)
{
    return leaf(ctx, reference, reference_len, test, test_len, width, height, CE_METRIC_DSSIM, 0.f, out);
}

int ce_calculate_butteraugli(ce_ctx *ctx, const uint8_t *reference, size_t reference_len, const uint8_t *test,
                             size_t test_len, size_t width, size_t height, float intensity_target, double *out)
{
    return leaf(ctx, reference, reference_len, test, test_len, width, height, CE_METRIC_BUTTERAUGLI, intensity_target,
                out);
}

// leaf scratch (ce_internal.h): device buffers of at least in_bytes / out_bytes and a pinned staging buffer of the larger
static int leaf_scratch(ce_ctx *ctx, size_t in_bytes, size_t out_bytes)
{
    CE_HIP(ctx, hipSetDevice(ctx->device));
    auto grow = [&](uint8_t *&p, size_t &cap, size_t want, bool host) -> int {
        if (cap >= want) return CE_OK;
        if (p) CE_HIP(ctx, host ? hipHostFree(p) : hipFree(p));
        p = nullptr;
        cap = 0;
        const size_t sz = want + want / 4;  // a little head room: a sweep over nearby shapes does not reallocate each time
        CE_HIP(ctx, host ? hipHostMalloc((void **)&p, sz, hipHostMallocDefault) : hipMalloc((void **)&p, sz));
        cap = sz;
        return CE_OK;
    };
    int rc = grow(ctx->leaf_d_in, ctx->leaf_in_cap, in_bytes, false);
    if (rc == CE_OK) rc = grow(ctx->leaf_d_out, ctx->leaf_out_cap, out_bytes, false);
    if (rc == CE_OK) rc = grow(ctx->leaf_h, ctx->leaf_h_cap, std::max(in_bytes, out_bytes), true);
    return rc;
}

// host image in -> kernel -> host image out through the leaf scratch, everything on the context's stream
static int leaf_roundtrip(ce_ctx *ctx, const void *in, size_t in_bytes, void *out, size_t out_bytes,
                          const std::function<int(uint8_t *, uint8_t *)> &launch)
{
    int rc = leaf_scratch(ctx, in_bytes, out_bytes);
    if (rc != CE_OK) return rc;
    std::memcpy(ctx->leaf_h, in, in_bytes);
    CE_HIP(ctx, hipMemcpyAsync(ctx->leaf_d_in, ctx->leaf_h, in_bytes, hipMemcpyHostToDevice, ctx->stream));
    rc = launch(ctx->leaf_d_in, ctx->leaf_d_out);
    if (rc != CE_OK) return rc;
    CE_HIP(ctx, hipMemcpyAsync(ctx->leaf_h, ctx->leaf_d_out, out_bytes, hipMemcpyDeviceToHost, ctx->stream));
    CE_HIP(ctx, hipStreamSynchronize(ctx->stream));
    std::memcpy(out, ctx->leaf_h, out_bytes);
    return CE_OK;
}

int ce_xyb_roundtrip(ce_ctx *ctx, const uint8_t *rgb, size_t rgb_len, size_t width, size_t height, uint8_t *out)
{
    if (!ctx || !rgb || !out) return CE_ERR_INVALID_ARG;
    if (rgb_len != width * height * 3)
        return fail(ctx, CE_ERR_BAD_LENGTH, "Buffer size mismatch");  // xyb.rs:227
    if (rgb_len == 0) return CE_OK;
    return leaf_roundtrip(ctx, rgb, rgb_len, out, rgb_len,
                          [&](uint8_t *d_in, uint8_t *d_out) { return ce_launch_xyb_roundtrip(ctx, d_in, d_out, width * height); });
}

int ce_rgb8_to_dssim_image(ce_ctx *ctx, const uint8_t *rgb, size_t rgb_len, size_t width, size_t height,
                           float *rgba_out)
{
    if (!ctx || !rgb || !rgba_out) return CE_ERR_INVALID_ARG;
    if (rgb_len != width * height * 3) return fail(ctx, CE_ERR_BAD_LENGTH, "Buffer size mismatch");
    const size_t n = width * height;
    if (n == 0) return CE_OK;
    return leaf_roundtrip(ctx, rgb, rgb_len, rgba_out, n * 4 * sizeof(float), [&](uint8_t *d_in, uint8_t *d_out) {
        return ce_launch_rgb8_to_dssim_image(ctx, d_in, reinterpret_cast<float *>(d_out), n);
    });
}

// ---- reference handle -------------------------------------------------------------------------

struct ce_ref {
    ce_ctx *ctx;
    ce_batch *batch;
    uint32_t flags;
};

int ce_ref_create(ce_ctx *ctx, const uint8_t *reference, size_t reference_len, uint32_t width, uint32_t height,
                  uint32_t flags, ce_ref **out)
{
    if (!ctx || !reference || !out) return CE_ERR_INVALID_ARG;
    *out = nullptr;
    if (reference_len != (size_t)width * height * 3)
        return fail(ctx, CE_ERR_BAD_LENGTH, "Invalid image size: expected " +
                                                 std::to_string((size_t)width * height * 3) + " bytes, got " +
                                                 std::to_string(reference_len));
    ce_batch *b = nullptr;
    int rc = ce_batch_create(ctx, width, height, 1, 1, &b);
    if (rc != CE_OK) return rc;
    rc = ce_batch_set_reference(b, 0, reference, reference_len);
    if (rc != CE_OK) {
        ce_batch_destroy(b);
        return rc;
    }
    b->keep_ref_pyramid = true;
    *out = new ce_ref{ctx, b, flags};
    return CE_OK;
}

int ce_ref_compare_many(ce_ref *ref, const uint8_t *const *tests, const size_t *test_lens, uint32_t n_tests,
                        uint32_t metric_mask, float intensity_target, ce_scores *out)
{
    if (!ref || !tests || !test_lens || !out) return CE_ERR_INVALID_ARG;
    if (n_tests == 0) return CE_OK;
    ce_ctx *ctx = ref->ctx;
    ce_batch *b = ref->batch;
    CE_HIP(ctx, hipSetDevice(ctx->device));
    if (n_tests > b->max_pairs) {
        // grow the handle: a new batch of the same shape takes over the resident reference (device copy)
        ce_batch *nb = nullptr;
        int rc = ce_batch_create(ctx, b->w, b->h, 1, n_tests, &nb);
        if (rc != CE_OK) return rc;
        rc = flush_uploads(b);
        if (rc != CE_OK) return rc;
        CE_HIP(ctx, hipMemcpyAsync(nb->d_refs, b->d_refs, b->img_bytes, hipMemcpyDeviceToDevice, ctx->stream));
        CE_HIP(ctx, hipStreamSynchronize(ctx->stream));
        nb->keep_ref_pyramid = true;
        for (int k = 0; k < 3; k++) nb->ref_builds[k] = b->ref_builds[k];  // the handle's history (ce_ref_stats) carries over
        ce_batch_destroy(b);
        ref->batch = b = nb;
    }
    bool any_ok = false;
    for (uint32_t i = 0; i < n_tests; i++) {
        out[i] = ce_scores{};
        if (!tests[i]) return CE_ERR_INVALID_ARG;
        if (test_lens[i] != b->img_bytes) {
            out[i].status = fail(ctx, CE_ERR_DIM_MISMATCH, "Dimension mismatch: reference " + std::to_string(b->img_bytes) +
                                                               " bytes, test " + std::to_string(test_lens[i]) + " bytes");
            continue;
        }
        any_ok = true;
    }
    if (!any_ok) return CE_OK;
    // Rejected items keep their slot (their scores are discarded); every valid test is uploaded to its own slot.
    std::vector<ce_scores> tmp(n_tests);
    b->caller_blocks = true;  // this call returns after its kernels: page-locked test images are read in place (upload())
    int rc = CE_OK;
    for (uint32_t i = 0; i < n_tests && rc == CE_OK; i++)
        rc = out[i].status != CE_OK ? ce_batch_bind_pair(b, i, 0) : ce_batch_set_test(b, i, 0, tests[i], test_lens[i]);
    if (rc == CE_OK) rc = ce_batch_run(b, n_tests, metric_mask, ref->flags, intensity_target, tmp.data());
    if (rc != CE_OK) {  // copies straight from the caller's memory may still be queued: drain them before the buffers go away
        hipStreamSynchronize(b->up_stream);
        hipStreamSynchronize(ctx->stream);
    }
    b->caller_blocks = false;
    if (rc != CE_OK) return rc;
    for (uint32_t i = 0; i < n_tests; i++)
        if (out[i].status == CE_OK) out[i] = tmp[i];
    return CE_OK;
}

int ce_ref_compare(ce_ref *ref, const uint8_t *test, size_t test_len, uint32_t metric_mask, float intensity_target,
                   ce_scores *out)
{
    if (!ref || !test || !out) return CE_ERR_INVALID_ARG;
    int rc = ce_ref_compare_many(ref, &test, &test_len, 1, metric_mask, intensity_target, out);
    if (rc != CE_OK) return rc;
    return out->status;
}

int ce_ref_stats(const ce_ref *ref, uint32_t builds[3])
{
    if (!ref || !builds) return CE_ERR_INVALID_ARG;
    for (int k = 0; k < 3; k++) builds[k] = ref->batch->ref_builds[k];
    return CE_OK;
}

void ce_ref_destroy(ce_ref *ref)
{
    if (!ref) return;
    ce_batch_destroy(ref->batch);
    delete ref;
}

// ---- measurement hooks ----------------------------------------------------------------------

int ce_prof_enable(ce_ctx *ctx, int on)
{
    if (!ctx) return CE_ERR_INVALID_ARG;
    prof_drain(ctx);
    ctx->prof = on != 0;
    ctx->prof_serial = on == 1;
    return CE_OK;
}

int ce_prof_filter(ce_ctx *ctx, const char *substring)
{
    if (!ctx) return CE_ERR_INVALID_ARG;
    prof_drain(ctx);
    ctx->prof_filter = substring ? substring : "";
    return CE_OK;
}

int ce_prof_reset(ce_ctx *ctx)
{
    if (!ctx) return CE_ERR_INVALID_ARG;
    prof_drain(ctx);
    ctx->stats.clear();
    return CE_OK;
}

int ce_prof_count(ce_ctx *ctx)
{
    if (!ctx) return 0;
    prof_drain(ctx);
    return (int)ctx->stats.size();
}

int ce_prof_get(ce_ctx *ctx, int index, const char **name, uint64_t *launches, double *total_ms)
{
    if (!ctx || index < 0 || index >= (int)ctx->stats.size()) return CE_ERR_INVALID_ARG;
    prof_drain(ctx);
    if (name) *name = ctx->stats[index].name.c_str();
    if (launches) *launches = ctx->stats[index].launches;
    if (total_ms) *total_ms = ctx->stats[index].total_ms;
    return CE_OK;
}

int ce_timer_start(ce_ctx *ctx)
{
    if (!ctx) return CE_ERR_INVALID_ARG;
    CE_HIP(ctx, hipSetDevice(ctx->device));
    CE_HIP(ctx, hipEventRecord(ctx->t0, ctx->stream));
    return CE_OK;
}

int ce_timer_stop(ce_ctx *ctx, double *elapsed_ms)
{
    if (!ctx || !elapsed_ms) return CE_ERR_INVALID_ARG;
    CE_HIP(ctx, hipSetDevice(ctx->device));
    CE_HIP(ctx, hipEventRecord(ctx->t1, ctx->stream));
    CE_HIP(ctx, hipEventSynchronize(ctx->t1));
    float ms = 0.f;
    CE_HIP(ctx, hipEventElapsedTime(&ms, ctx->t0, ctx->t1));
    *elapsed_ms = ms;
    return CE_OK;
}

// ---- test hooks -------------------------------------------------------------------------------

int ce_debug_ssim2_planes(ce_batch *b, int scale, int which, int channel, float *out, size_t out_floats,
                          uint32_t *w_out, uint32_t *h_out)
{
    if (!b || !out || !b->ssim2_ready || scale < 0 || scale >= b->n_scales) return CE_ERR_INVALID_ARG;
    ce_ctx *ctx = b->ctx;
    const ce_scale_dims &d = b->sd[scale];
    const int nplanes = which == 4 ? CE_SSIM2_STREAMS : 3;
    if (out_floats < (size_t)nplanes * d.w * d.h) return CE_ERR_INVALID_ARG;
    const uint32_t ref_slot = b->h_pair_ref[0], test_slot = b->max_refs;
    const float *src = nullptr;
    switch (which) {
        case 0:
        case 1:
            if (scale == 0) return CE_ERR_INVALID_ARG;  // level 0 has no linear plane (read from u8 on the fly)
            src = b->d_lin[scale] + (size_t)(which == 0 ? ref_slot : test_slot) * 3 * d.plane;
            break;
        case 2: src = b->d_xyb[scale] + (size_t)ref_slot * 3 * d.plane; break;
        case 3: src = b->d_xyb[scale] + (size_t)test_slot * 3 * d.plane; break;
        case 4:
            if (channel < 0 || channel > 2) return CE_ERR_INVALID_ARG;
            src = b->d_hbuf[scale] + (size_t)channel * CE_SSIM2_STREAMS * d.hplane;
            break;
        default: return CE_ERR_INVALID_ARG;
    }
    CE_HIP(ctx, hipSetDevice(ctx->device));
    CE_HIP(ctx, hipStreamSynchronize(ctx->stream));
    const size_t spl = which == 4 ? d.hplane : d.plane, spitch = which == 4 ? d.hpitch : d.pitch;
    for (int p = 0; p < nplanes; p++)
        CE_HIP(ctx, hipMemcpy2D(out + (size_t)p * d.w * d.h, (size_t)d.w * sizeof(float), src + (size_t)p * spl,
                                spitch * sizeof(float), (size_t)d.w * sizeof(float), d.h, hipMemcpyDeviceToHost));
    if (w_out) *w_out = d.w;
    if (h_out) *h_out = d.h;
    return CE_OK;
}

int ce_debug_ssim2_limit_scales(ce_batch *b, int max_scales)
{
    if (!b || max_scales < 1 || max_scales > CE_MAX_SCALES) return CE_ERR_INVALID_ARG;
    b->debug_max_scales = max_scales;
    return CE_OK;
}

int ce_debug_ssim2_averages(ce_batch *b, uint32_t pair_index, double *avg, int *n_scales)
{
    if (!b || !avg || !b->ssim2_ready || pair_index >= b->max_pairs) return CE_ERR_INVALID_ARG;
    ce_ctx *ctx = b->ctx;
    CE_HIP(ctx, hipSetDevice(ctx->device));
    CE_HIP(ctx, hipStreamSynchronize(ctx->stream));
    CE_HIP(ctx, hipMemcpy(avg, b->d_avg + (size_t)pair_index * CE_MAX_SCALES * 18, sizeof(double) * CE_MAX_SCALES * 18,
                          hipMemcpyDeviceToHost));
    if (n_scales) *n_scales = b->n_scales;
    return CE_OK;
}

int ce_debug_ssim2_occupancy(int which) { return ce_ssim2_occupancy(which); }

int ce_debug_div_sweep(ce_ctx *ctx, uint64_t seed, uint64_t count, uint64_t *mismatches)
{
    if (!ctx || count == 0) return CE_ERR_INVALID_ARG;
    CE_HIP(ctx, hipSetDevice(ctx->device));
    return ce_butteraugli_div_sweep(ctx, seed, count, mismatches);
}

int ce_debug_calibrate_traffic(ce_ctx *ctx, size_t bytes)
{
    if (!ctx) return CE_ERR_INVALID_ARG;
    CE_HIP(ctx, hipSetDevice(ctx->device));
    return ce_calibrate_traffic(ctx, bytes);
}

int ce_debug_cbrt_sweep(ce_ctx *ctx, uint32_t first_bits, uint64_t count, uint64_t *mismatches, uint64_t *slow_path)
{
    if (!ctx || count == 0 || (uint64_t)first_bits + count > (1ull << 32)) return CE_ERR_INVALID_ARG;
    CE_HIP(ctx, hipSetDevice(ctx->device));
    return ce_ssim2_cbrt_sweep(ctx, first_bits, count, mismatches, slow_path);
}

}  // extern "C"
