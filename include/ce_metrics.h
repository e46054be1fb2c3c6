/*
 * ce_metrics.h — C ABI of libce_metrics_hip.so, the MI355X (gfx950) backend for the
 * perceptual-metric hot path of imazen/codec-eval.
 *
 * Every entry point replaces one reference interface; the citation after each
 * declaration is the Rust item (path relative to the reference checkout) whose
 * FFI binding would call it.  Conventions (reference: SURVEY.md §8b):
 *
 *   pixels     tightly packed interleaved RGB8, row-major, no stride
 *              (src/metrics/ssimulacra2.rs:41,72; src/eval/session.rs:98-117)
 *   ownership  every pointer is borrowed for the duration of the call only; the
 *              library copies what it needs and owns all device memory
 *   errors     integer status, never abort/panic; ce_last_error() gives the text.
 *              CE_ERR_DIM_MISMATCH  <-> Error::DimensionMismatch  (src/error.rs:31-38)
 *              CE_ERR_BAD_LENGTH    <-> Error::MetricCalculation{"Invalid image size"}
 *                                       (src/metrics/ssimulacra2.rs:73-82)
 *              CE_ERR_TOO_SMALL     <-> Error::MetricCalculation from the metric crate
 *                                       for images under 8x8 (src/eval/helpers.rs:89)
 *              CE_ERR_BACKEND       <-> Error::MetricCalculation{reason: backend text}
 *              PSNR in the reference asserts (src/metrics/mod.rs:313-314); here it
 *              returns the code and the Rust shim re-raises the panic.
 *   results    double, as MetricResult's Option<f64> (src/metrics/mod.rs:140-149)
 *   threading  one in-flight call per context (GpuSsim2::compute takes &mut self,
 *              crates/codec-iter/src/gpu.rs:83); any number of contexts per device.
 *              Process-wide state: one counter per device of launched-and-uncollected
 *              batches, read by ce_batch_launch to choose between running a batch's metric
 *              chains side by side or back to back (a scheduling hint: it never changes a
 *              score), and the environment knobs DESIGN.md lists, read once.
 *   no torch / no C++ types in any signature.
 */
#ifndef CE_METRICS_H
#define CE_METRICS_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct ce_ctx ce_ctx;     /* device + stream + scratch pool       */
typedef struct ce_batch ce_batch; /* HBM-resident grid of (ref, test) pairs of one shape */
typedef struct ce_ref ce_ref;     /* one reference image held on device   */
typedef struct ce_lut ce_lut;     /* one colour transform (ICC -> sRGB) held on device as a 2^24-entry table */

enum ce_status {
    CE_OK = 0,
    CE_ERR_DIM_MISMATCH = 1,
    CE_ERR_BAD_LENGTH = 2,
    CE_ERR_TOO_SMALL = 3,
    CE_ERR_BACKEND = 4,
    CE_ERR_INVALID_ARG = 5
};

/* MetricConfig (src/metrics/mod.rs:46-63) as a bit mask */
enum ce_metric {
    CE_METRIC_DSSIM = 1u << 0,
    CE_METRIC_SSIMULACRA2 = 1u << 1,
    CE_METRIC_BUTTERAUGLI = 1u << 2,
    CE_METRIC_PSNR = 1u << 3
};
enum ce_flag {
    CE_FLAG_XYB_ROUNDTRIP = 1u << 0 /* MetricConfig::xyb_roundtrip: reference side only (session.rs:447-456) */
};

#define CE_DEFAULT_INTENSITY_TARGET 80.0f /* src/metrics/butteraugli.rs:94 */

/* MetricResult (src/metrics/mod.rs:140-149): a score is meaningful iff its bit is
 * set in `valid`; `status` is the ce_status of this pair. */
typedef struct ce_scores {
    double dssim;
    double ssimulacra2;
    double butteraugli;
    double psnr;
    uint32_t valid;
    int32_t status;
} ce_scores;

/* one work item of the (image x codec x quality) grid */
typedef struct ce_pair_desc {
    const uint8_t *reference;
    size_t reference_len;
    const uint8_t *test;
    size_t test_len;
    uint32_t width;
    uint32_t height;
} ce_pair_desc;

/* ---- library / device -------------------------------------------------------- */
const char *ce_version(void);
int ce_device_count(void); /* number of visible HIP devices; <0 on runtime failure */

/* GpuSsim2::new (crates/codec-iter/src/gpu.rs:40-80) without the fixed-shape limit:
 * scratch is keyed by (w,h) and grows on demand. */
int ce_ctx_create(int device, ce_ctx **out);
/* same, launching on a caller-owned hipStream_t (passed as void*); NULL = own stream */
int ce_ctx_create_on_stream(int device, void *hip_stream, ce_ctx **out);
/* Drop for GpuSsim2 (gpu.rs:118-133): synchronises the stream, then frees */
void ce_ctx_destroy(ce_ctx *ctx);
int ce_ctx_synchronize(ce_ctx *ctx);
void *ce_ctx_stream(ce_ctx *ctx); /* the hipStream_t kernels are launched on */
const char *ce_last_error(const ce_ctx *ctx);

/* ---- leaf metric calls: one per reference leaf function ------------------------ */
/* calculate_psnr                       src/metrics/mod.rs:312 */
int ce_calculate_psnr(ce_ctx *ctx, const uint8_t *reference, size_t reference_len, const uint8_t *test,
                      size_t test_len, size_t width, size_t height, double *out);
/* calculate_ssimulacra2                src/metrics/ssimulacra2.rs:59 ; GpuSsim2::compute gpu.rs:83 */
int ce_calculate_ssimulacra2(ce_ctx *ctx, const uint8_t *reference, size_t reference_len,
                             const uint8_t *test, size_t test_len, size_t width, size_t height,
                             double *out);
/* rgb8_to_dssim_image x2 + calculate_dssim   src/metrics/dssim.rs:102,40 (session.rs:467-476) */
int ce_calculate_dssim(ce_ctx *ctx, const uint8_t *reference, size_t reference_len, const uint8_t *test,
                       size_t test_len, size_t width, size_t height, double *out);
/* calculate_butteraugli / _with_intensity    src/metrics/butteraugli.rs:45,99 */
int ce_calculate_butteraugli(ce_ctx *ctx, const uint8_t *reference, size_t reference_len,
                             const uint8_t *test, size_t test_len, size_t width, size_t height,
                             float intensity_target, double *out);
/* xyb_roundtrip                         src/metrics/xyb.rs:225 ; out has rgb_len bytes */
int ce_xyb_roundtrip(ce_ctx *ctx, const uint8_t *rgb, size_t rgb_len, size_t width, size_t height,
                     uint8_t *out);
/* rgb8_to_dssim_image                   src/metrics/dssim.rs:102 ; out has 4*w*h floats (a = 1.0) */
int ce_rgb8_to_dssim_image(ce_ctx *ctx, const uint8_t *rgb, size_t rgb_len, size_t width, size_t height,
                           float *rgba_out);

/* ---- dispatcher --------------------------------------------------------------- */
/* EvalSession::calculate_metrics        src/eval/session.rs:437-497
 * evaluate_single                       src/eval/helpers.rs:105-173 */
int ce_eval_pair(ce_ctx *ctx, const uint8_t *reference, size_t reference_len, const uint8_t *test,
                 size_t test_len, uint32_t width, uint32_t height, uint32_t metric_mask, uint32_t flags,
                 float intensity_target, ce_scores *out);

/* The (codec x quality) double loop of EvalSession::evaluate_image (session.rs:375-376)
 * and images.par_iter() of crates/codec-compare/src/full_comparison.rs:319-328, as one
 * call: n independent pairs, any mix of shapes (bucketed by shape internally).
 * Per-pair failures are reported in out[i].status; the return value is CE_OK unless the
 * call itself could not run. */
int ce_eval_batch(ce_ctx *ctx, size_t n, const ce_pair_desc *pairs, uint32_t metric_mask, uint32_t flags,
                  float intensity_target, ce_scores *out);
/* the same with a colour table per pair for the DISTORTED image (NULL entries / a NULL array: none): the decoded image's
 * ICC -> sRGB step of evaluate_image (session.rs:394, ImageData::to_rgb8_srgb) runs on the device - see ce_lut_create */
int ce_eval_batch_lut(ce_ctx *ctx, size_t n, const ce_pair_desc *pairs, const ce_lut *const *test_luts, uint32_t metric_mask,
                      uint32_t flags, float intensity_target, ce_scores *out);

/* Memory planning for callers that size their own batches (EvalSession::evaluate_corpus streams a corpus through
 * batches that fit the device): an upper estimate of the device bytes a batch of this shape holds once the metrics
 * in metric_mask have run, and the device's free / total memory.  ce_eval_batch uses the same estimate to split a
 * grid that does not fit into chunks (each at most a third of the free memory; CE_EVAL_BATCH_BYTES overrides). */
size_t ce_estimate_batch_bytes(uint32_t width, uint32_t height, uint32_t n_refs, uint32_t n_pairs, uint32_t metric_mask);
int ce_ctx_memory_info(ce_ctx *ctx, size_t *free_bytes, size_t *total_bytes);

/* Page-locked host memory for the images a caller hands to ce_eval_batch / ce_batch_set_*: buffers from here are copied by
 * the DMA engines straight from the caller's memory (one asynchronous copy per image, overlapped with the kernels of the
 * previous chunk); any other host pointer is first copied through the library's own page-locked staging ring by host
 * threads.  A decoder writes its output into such a buffer and scores it in place - the role of the cudarse pinned buffers
 * behind GpuSsim2 (crates/codec-iter/src/gpu.rs:96-108).  Freed with ce_host_free (before or after the context is gone). */
int ce_host_alloc(ce_ctx *ctx, size_t bytes, void **out);
int ce_host_free(ce_ctx *ctx, void *p);

/* ---- HBM-resident grid (what bench.py times; inputs already on device) ---------- */
/* One shape, up to max_refs reference images and max_pairs (reference, test) items. */
int ce_batch_create(ce_ctx *ctx, uint32_t width, uint32_t height, uint32_t max_refs, uint32_t max_pairs,
                    ce_batch **out);
void ce_batch_destroy(ce_batch *b);
/* host -> device copies (pinned staging inside) */
int ce_batch_set_reference(ce_batch *b, uint32_t ref_index, const uint8_t *rgb, size_t len);
int ce_batch_set_test(ce_batch *b, uint32_t pair_index, uint32_t ref_index, const uint8_t *rgb, size_t len);
/* Decoded-image ingest: the same two calls for pixels as a decoder hands them over.  The conversion to the
 * packed RGB8 the metrics read happens on the device (no per-pixel pass on the host):
 *   CE_PIXEL_RGBA8         alpha dropped          ImageData::to_rgb8_vec, src/eval/session.rs:98-117
 *   CE_PIXEL_RGB16_10BIT   ((v*255+512)/1023).min(255) per sample, u16 little endian
 *   CE_PIXEL_RGBA16_10BIT  both                   to_8bit / pixel_data_to_rgb8, crates/codec-iter/src/avif_config.rs:122-170
 * len is in bytes and must be width * height * bytes-per-pixel of the format. */
enum {
    CE_PIXEL_RGB8 = 0,
    CE_PIXEL_RGBA8 = 1,
    CE_PIXEL_RGB16_10BIT = 2,
    CE_PIXEL_RGBA16_10BIT = 3
};
int ce_batch_set_reference_fmt(ce_batch *b, uint32_t ref_index, const void *pixels, size_t len, int format);
int ce_batch_set_test_fmt(ce_batch *b, uint32_t pair_index, uint32_t ref_index, const void *pixels, size_t len, int format);
/* ICC -> sRGB on the device, EXACTLY (transform_to_srgb, src/metrics/icc.rs:69-103, called on every decoded image at
 * src/eval/session.rs:394 through ImageData::to_rgb8_srgb).  The reference's transform (moxcms, 8-bit RGB -> 8-bit RGB)
 * is a pure function of a pixel's three bytes, so the table of its outputs on all 2^24 colours reproduces it bit for bit:
 * the host runs ITS colour management once per profile over the identity colour cube (50 MB, tens of milliseconds) and
 * the device applies the table to every decoded image of that profile - no per-pixel colour management on the host and
 * no interpolation error.  table[((r << 16) | (g << 8) | b) * 3 + c] = channel c of the transformed colour;
 * table_len must be 3 * 2^24.  A table belongs to its context's device and must be destroyed before the context. */
int ce_lut_create(ce_ctx *ctx, const uint8_t *table, size_t table_len, ce_lut **out);
void ce_lut_destroy(ce_lut *lut);
/* ce_batch_set_{reference,test}_fmt followed by the table, all on the device (lut == NULL: no transform) */
int ce_batch_set_reference_lut(ce_batch *b, uint32_t ref_index, const void *pixels, size_t len, int format, const ce_lut *lut);
int ce_batch_set_test_lut(ce_batch *b, uint32_t pair_index, uint32_t ref_index, const void *pixels, size_t len, int format,
                          const ce_lut *lut);
/* device pointers of the packed u8 slabs ([max_refs][h][w][3], [max_pairs][h][w][3]) so a caller that
 * already has pixels in HBM (e.g. a GPU decoder) can write them in place */
void *ce_batch_reference_slab(ce_batch *b);
void *ce_batch_test_slab(ce_batch *b);
int ce_batch_bind_pair(ce_batch *b, uint32_t pair_index, uint32_t ref_index);
/* run the hot path over pairs [0, n_pairs); blocks until scores are on the host */
int ce_batch_run(ce_batch *b, uint32_t n_pairs, uint32_t metric_mask, uint32_t flags, float intensity_target,
                 ce_scores *out);
/* same, without the final wait (for back-to-back timed steps and pipelines of several batches): the launch queues the
 * kernels and, behind them, the copy of the scores into the batch's page-locked buffer; ce_batch_collect waits for THAT
 * launch only (not for batches launched after it on the same context) and converts the first n_pairs scores
 * (n_pairs <= the launched count, CE_ERR_INVALID_ARG otherwise) */
int ce_batch_launch(ce_batch *b, uint32_t n_pairs, uint32_t metric_mask, uint32_t flags,
                    float intensity_target);
int ce_batch_collect(ce_batch *b, uint32_t n_pairs, ce_scores *out);
/* libjxl's p-norm of the Butteraugli diffmap (mean of the 3-, 6- and 12-norms) for the pairs of the last run
 * that asked for CE_METRIC_BUTTERAUGLI.  The reference only ever reads `.score` (the max-norm,
 * src/metrics/butteraugli.rs:80); BASELINE.json's configs[2] also names the 3-norm. */
int ce_batch_butteraugli_pnorm3(ce_batch *b, uint32_t n_pairs, double *out);

/* ---- reference handle: Ssimulacra2Reference::{new,compare} ------------------------
 * crates/codec-iter/src/eval.rs:138-149,83-89; crates/codec-compare/src/brute_force_sweep.rs:197-201,256
 * The handle keeps the reference resident in HBM together with its reference-side state for EVERY metric: the XYB
 * roundtrip (CE_FLAG_XYB_ROUNDTRIP), the SSIMULACRA2 XYB pyramid, DSSIM's img / mu / blur(img^2) pyramid
 * (Dssim::create_image of the reference) and Butteraugli's PsychoImage at both resolutions are built by the first
 * compare that needs them and reused by every later one (Butteraugli's also depends on intensity_target: a compare
 * with another target rebuilds it).  The row-blurred SSIMULACRA2 reference planes are NOT kept: the blur passes are
 * bound by HBM traffic and a compare would read a cached plane just as it reads a recomputed one. */
int ce_ref_create(ce_ctx *ctx, const uint8_t *reference, size_t reference_len, uint32_t width, uint32_t height,
                  uint32_t flags, ce_ref **out);
int ce_ref_compare(ce_ref *ref, const uint8_t *test, size_t test_len, uint32_t metric_mask,
                   float intensity_target, ce_scores *out);
/* The quality sweep of one reference in one launch: the loop
 *   for q in qualities { reference.compare(decoded[q]) }      (eval.rs:83-89, brute_force_sweep.rs:256)
 * as a single batch.  tests[i] / test_lens[i]: the i-th distorted image; out[i].status is per item
 * (CE_ERR_DIM_MISMATCH for a wrong length, the others still run); the return value reports call-level
 * failures only. */
int ce_ref_compare_many(ce_ref *ref, const uint8_t *const *tests, const size_t *test_lens, uint32_t n_tests,
                        uint32_t metric_mask, float intensity_target, ce_scores *out);
/* builds[k] = number of compares so far that had to (re)build the reference-side state of SSIMULACRA2 (k = 0),
 * DSSIM (1), Butteraugli (2): 1 each after any number of compares of one handle with one intensity target */
int ce_ref_stats(const ce_ref *ref, uint32_t builds[3]);
void ce_ref_destroy(ce_ref *ref);

/* ---- measurement hooks (bench.py) ------------------------------------------------ */
/* Bracket every kernel launch with a HIP event pair, recorded on the stream the kernel is launched on,
 * and accumulate per-kernel time.  on = 0: off (default).  on = 2: events only; the batch keeps its
 * multi-stream schedule, so a kernel's time includes whatever it shares the GPU with (what rocprofv3's
 * kernel trace sees).  on = 1: additionally keep all launches on the context's stream, one kernel at a
 * time ("solo" times). */
int ce_prof_enable(ce_ctx *ctx, int on);
/* restrict the events to kernels whose name contains `substring` (NULL or "" = all kernels; "=name" = exactly that kernel) */
int ce_prof_filter(ce_ctx *ctx, const char *substring);
int ce_prof_reset(ce_ctx *ctx);
/* number of distinct kernels seen; then per index: name, launches, total ms */
int ce_prof_count(ce_ctx *ctx);
int ce_prof_get(ce_ctx *ctx, int index, const char **name, uint64_t *launches, double *total_ms);
/* plain HIP-event stopwatch on the context's stream */
int ce_timer_start(ce_ctx *ctx);
int ce_timer_stop(ce_ctx *ctx, double *elapsed_ms);

/* Test hooks (plane-level parity, arithmetic sweeps, counter calibration) are declared in ce_metrics_debug.h: they are
 * exported by the same library but are not part of the boundary a host binds. */

#ifdef __cplusplus
}
#endif
#endif
