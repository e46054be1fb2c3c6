// codec_eval.hpp — C++17 host-side mirror of the reference's interface for the metric hot path, written
// on top of the C ABI (include/ce_metrics.h).  The reference is a Rust crate and no Rust toolchain exists
// in the build image, so this header plays the part the Rust `hip` feature would play:
// same names, argument meaning and error behaviour as
//   src/metrics/mod.rs        MetricConfig, MetricResult, PerceptionLevel, calculate_psnr
//   src/metrics/{ssimulacra2,dssim,butteraugli,xyb}.rs   the leaf functions
//   src/eval/session.rs       ImageData, EncodeRequest, EvalConfig, EvalSession::evaluate_image
//   src/eval/helpers.rs       evaluate_single, assert_quality, assert_perception_level
//   src/eval/report.rs        CodecResult, ImageReport
// The one deliberate difference: evaluate_image runs every encode/decode callback first and then scores
// the whole (codec x quality) grid with ONE ce_eval_batch call (the reference scores inside the double
// loop, session.rs:375-410); results are emitted in the reference's loop order.
#pragma once

#include <chrono>
#include <cmath>
#include <cstdint>
#include <functional>
#include <limits>
#include <map>
#include <memory>
#include <optional>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "ce_metrics.h"

namespace codec_eval {

// ---- src/error.rs:31-75 (the variants this path can raise) -----------------------------------------
struct Error : std::runtime_error {
    enum class Kind { DimensionMismatch, MetricCalculation, QualityBelowThreshold, Codec };
    Kind kind;
    Error(Kind k, const std::string &what) : std::runtime_error(what), kind(k) {}
};

// ---- src/metrics/mod.rs:46-136 ------------------------------------------------------------------------
struct MetricConfig {
    bool dssim = false, ssimulacra2 = false, butteraugli = false, psnr = false, xyb_roundtrip = false;
    static MetricConfig all() { return {true, true, true, true, false}; }
    static MetricConfig fast() { return {false, false, false, true, false}; }
    static MetricConfig perceptual() { return {true, true, true, false, false}; }
    static MetricConfig perceptual_xyb() { return {true, true, true, false, true}; }
    static MetricConfig ssimulacra2_only() { return {false, true, false, false, false}; }
    MetricConfig with_xyb_roundtrip() const { MetricConfig c = *this; c.xyb_roundtrip = true; return c; }
    uint32_t mask() const
    {
        return (dssim ? (uint32_t)CE_METRIC_DSSIM : 0u) | (ssimulacra2 ? (uint32_t)CE_METRIC_SSIMULACRA2 : 0u) |
               (butteraugli ? (uint32_t)CE_METRIC_BUTTERAUGLI : 0u) | (psnr ? (uint32_t)CE_METRIC_PSNR : 0u);
    }
    uint32_t flags() const { return xyb_roundtrip ? (uint32_t)CE_FLAG_XYB_ROUNDTRIP : 0u; }
};

// ---- src/metrics/mod.rs:172-284 -----------------------------------------------------------------------
enum class PerceptionLevel : uint8_t { Imperceptible, Marginal, Subtle, Noticeable, Degraded };

inline PerceptionLevel perception_from_dssim(double d)
{
    return d < 0.0003 ? PerceptionLevel::Imperceptible : d < 0.0007 ? PerceptionLevel::Marginal
         : d < 0.0015 ? PerceptionLevel::Subtle : d < 0.003 ? PerceptionLevel::Noticeable : PerceptionLevel::Degraded;
}
inline PerceptionLevel perception_from_ssimulacra2(double s)
{
    return s > 90.0 ? PerceptionLevel::Imperceptible : s > 80.0 ? PerceptionLevel::Marginal
         : s > 70.0 ? PerceptionLevel::Subtle : s > 50.0 ? PerceptionLevel::Noticeable : PerceptionLevel::Degraded;
}
inline PerceptionLevel perception_from_butteraugli(double b)
{
    return b < 1.0 ? PerceptionLevel::Imperceptible : b < 2.0 ? PerceptionLevel::Marginal
         : b < 3.0 ? PerceptionLevel::Subtle : b < 5.0 ? PerceptionLevel::Noticeable : PerceptionLevel::Degraded;
}
inline const char *perception_code(PerceptionLevel l)
{
    static const char *const c[] = {"IMP", "MAR", "SUB", "NOT", "DEG"};
    return c[(int)l];
}

// ---- src/metrics/mod.rs:138-169 -----------------------------------------------------------------------
struct MetricResult {
    std::optional<double> dssim, ssimulacra2, butteraugli, psnr;
    std::optional<PerceptionLevel> perception_level() const
    {
        return dssim ? std::optional<PerceptionLevel>(perception_from_dssim(*dssim)) : std::nullopt;
    }
    static MetricResult from_c(const ce_scores &s)
    {
        MetricResult r;
        if (s.valid & CE_METRIC_DSSIM) r.dssim = s.dssim;
        if (s.valid & CE_METRIC_SSIMULACRA2) r.ssimulacra2 = s.ssimulacra2;
        if (s.valid & CE_METRIC_BUTTERAUGLI) r.butteraugli = s.butteraugli;
        if (s.valid & CE_METRIC_PSNR) r.psnr = s.psnr;
        return r;
    }
};

// ---- the device context (GpuSsim2::new / Drop, crates/codec-iter/src/gpu.rs:40-80,118-133) ------------
class HipBackend {
public:
    explicit HipBackend(int device = 0)
    {
        if (ce_ctx_create(device, &ctx_) != CE_OK)
            throw Error(Error::Kind::MetricCalculation, std::string("HIP init failed: ") + ce_last_error(nullptr));
    }
    ~HipBackend() { ce_ctx_destroy(ctx_); }
    HipBackend(const HipBackend &) = delete;
    HipBackend &operator=(const HipBackend &) = delete;
    ce_ctx *ctx() const { return ctx_; }
    std::string last_error() const { return ce_last_error(ctx_); }
    static int device_count() { return ce_device_count(); }

private:
    ce_ctx *ctx_ = nullptr;
};

namespace detail {
inline void check(const HipBackend &be, int rc, const char *metric, size_t w, size_t h, size_t test_len)
{
    if (rc == CE_OK) return;
    if (rc == CE_ERR_DIM_MISMATCH)  // ssimulacra2.rs:65-70
        throw Error(Error::Kind::DimensionMismatch, "Dimension mismatch: expected (" + std::to_string(w) + ", " + std::to_string(h) +
                                                        "), got (" + std::to_string(h ? test_len / 3 / h : 0) + ", " + std::to_string(h) + ")");
    throw Error(Error::Kind::MetricCalculation, std::string("Metric calculation failed: ") + metric + ": " + be.last_error());
}
}  // namespace detail

namespace metrics {
using Bytes = std::vector<uint8_t>;

// calculate_psnr, src/metrics/mod.rs:312-331.  The reference asserts on bad lengths; so does this.
inline double calculate_psnr(const HipBackend &be, const Bytes &reference, const Bytes &test, size_t width, size_t height)
{
    if (reference.size() != test.size()) throw std::logic_error("assertion failed: reference.len() == test.len()");
    if (reference.size() != width * height * 3) throw std::logic_error("assertion failed: reference.len() == width * height * 3");
    double out = 0;
    detail::check(be, ce_calculate_psnr(be.ctx(), reference.data(), reference.size(), test.data(), test.size(), width, height, &out),
                  "PSNR", width, height, test.size());
    return out;
}
// calculate_ssimulacra2, src/metrics/ssimulacra2.rs:59-100
inline double calculate_ssimulacra2(const HipBackend &be, const Bytes &reference, const Bytes &test, size_t width, size_t height)
{
    double out = 0;
    detail::check(be, ce_calculate_ssimulacra2(be.ctx(), reference.data(), reference.size(), test.data(), test.size(), width, height, &out),
                  "SSIMULACRA2", width, height, test.size());
    return out;
}
// rgb8_to_dssim_image x2 + calculate_dssim, src/metrics/dssim.rs:102-114,40-71 (session.rs:467-476 composes them)
inline double calculate_dssim(const HipBackend &be, const Bytes &reference, const Bytes &test, size_t width, size_t height)
{
    double out = 0;
    detail::check(be, ce_calculate_dssim(be.ctx(), reference.data(), reference.size(), test.data(), test.size(), width, height, &out),
                  "DSSIM", width, height, test.size());
    return out;
}
// calculate_butteraugli / _with_intensity, src/metrics/butteraugli.rs:45-136
inline double calculate_butteraugli_with_intensity(const HipBackend &be, const Bytes &reference, const Bytes &test, size_t width,
                                                   size_t height, float intensity_target)
{
    double out = 0;
    detail::check(be, ce_calculate_butteraugli(be.ctx(), reference.data(), reference.size(), test.data(), test.size(), width, height,
                                               intensity_target, &out),
                  "Butteraugli", width, height, test.size());
    return out;
}
inline double calculate_butteraugli(const HipBackend &be, const Bytes &reference, const Bytes &test, size_t width, size_t height)
{
    return calculate_butteraugli_with_intensity(be, reference, test, width, height, CE_DEFAULT_INTENSITY_TARGET);
}
// xyb_roundtrip, src/metrics/xyb.rs:225-253 (asserts on the length, :227)
inline Bytes xyb_roundtrip(const HipBackend &be, const Bytes &rgb, size_t width, size_t height)
{
    if (rgb.size() != width * height * 3) throw std::logic_error("Buffer size mismatch");
    Bytes out(rgb.size());
    detail::check(be, ce_xyb_roundtrip(be.ctx(), rgb.data(), rgb.size(), width, height, out.data()), "XYB", width, height, rgb.size());
    return out;
}
// rgb8_to_dssim_image, src/metrics/dssim.rs:102-114: RGBA f32, linear light, a = 1.0
inline std::vector<float> rgb8_to_dssim_image(const HipBackend &be, const Bytes &rgb, size_t width, size_t height)
{
    std::vector<float> out(width * height * 4);
    detail::check(be, ce_rgb8_to_dssim_image(be.ctx(), rgb.data(), rgb.size(), width, height, out.data()), "DSSIM", width, height, rgb.size());
    return out;
}
}  // namespace metrics

namespace eval {

// ---- src/eval/session.rs:25-147 (the slice variants; imgref variants collapse to them in C++) ----------
struct ImageData {
    enum class Format { Rgb8, Rgba8 } format = Format::Rgb8;
    std::vector<uint8_t> data;
    size_t width = 0, height = 0;
    std::optional<std::vector<uint8_t>> icc_profile;  // RgbSliceWithIcc, session.rs:60-77
    static ImageData rgb(std::vector<uint8_t> d, size_t w, size_t h) { return {Format::Rgb8, std::move(d), w, h, std::nullopt}; }
    static ImageData rgba(std::vector<uint8_t> d, size_t w, size_t h) { return {Format::Rgba8, std::move(d), w, h, std::nullopt}; }
    static ImageData rgb_with_icc(std::vector<uint8_t> d, size_t w, size_t h, std::vector<uint8_t> icc)
    {
        return {Format::Rgb8, std::move(d), w, h, std::move(icc)};
    }
    // to_rgb8_vec, session.rs:98-117: alpha is dropped
    std::vector<uint8_t> to_rgb8_vec() const
    {
        if (format == Format::Rgb8) return data;
        std::vector<uint8_t> out;
        out.reserve(width * height * 3);
        for (size_t i = 0; i + 3 < data.size(); i += 4) out.insert(out.end(), {data[i], data[i + 1], data[i + 2]});
        return out;
    }
};

// ---- ICC -> sRGB (src/metrics/icc.rs:69-103) as a device colour table --------------------------------------------------
// Cms: the host's colour management, 8-bit RGB -> 8-bit RGB for one profile (the reference's default build uses moxcms).
// It is evaluated once per profile on the identity colour cube; the table then lives on the device (ce_lut_*).
using Cms = std::function<std::vector<uint8_t>(const std::vector<uint8_t> &icc_profile, const std::vector<uint8_t> &rgb)>;

class HipColorTable {
public:
    HipColorTable(const HipBackend &be, const Cms &cms, const std::vector<uint8_t> &icc_profile)
    {
        std::vector<uint8_t> cube((size_t)3 << 24);
        for (uint32_t v = 0; v < (1u << 24); v++) {
            cube[3 * (size_t)v] = (uint8_t)(v >> 16);
            cube[3 * (size_t)v + 1] = (uint8_t)(v >> 8);
            cube[3 * (size_t)v + 2] = (uint8_t)v;
        }
        const std::vector<uint8_t> table = cms(icc_profile, cube);
        if (ce_lut_create(be.ctx(), table.data(), table.size(), &lut_) != CE_OK)
            throw Error(Error::Kind::MetricCalculation, "Metric calculation failed: ICC: Failed to create ICC transform: " + be.last_error());
    }
    ~HipColorTable() { ce_lut_destroy(lut_); }
    HipColorTable(const HipColorTable &) = delete;
    HipColorTable &operator=(const HipColorTable &) = delete;
    const ce_lut *get() const { return lut_; }

private:
    ce_lut *lut_ = nullptr;
};

struct EncodeRequest {  // session.rs:151-177
    double quality = 0;
    std::map<std::string, std::string> params;
};
using EncodeFn = std::function<std::vector<uint8_t>(const ImageData &, const EncodeRequest &)>;  // session.rs:181
using DecodeFn = std::function<ImageData(const std::vector<uint8_t> &)>;                          // session.rs:186

struct EvalConfig {  // session.rs:190-279; the default quality sweep is :273-275
    MetricConfig metrics = MetricConfig::all();
    std::vector<double> quality_levels = {50.0, 60.0, 70.0, 80.0, 85.0, 90.0, 95.0};
    float intensity_target = CE_DEFAULT_INTENSITY_TARGET;
};

struct CodecResult {  // src/eval/report.rs:16-52
    std::string codec_id, codec_version;
    double quality = 0;
    size_t file_size = 0;
    double bits_per_pixel = 0;
    std::chrono::nanoseconds encode_time{0};
    std::optional<std::chrono::nanoseconds> decode_time;
    MetricResult metrics;
    std::optional<PerceptionLevel> perception;
    std::map<std::string, std::string> codec_params;
};
struct ImageReport {  // report.rs:68-136
    std::string name;
    uint32_t width = 0, height = 0;
    std::vector<CodecResult> results;
};

class EvalSession {  // session.rs:281-497
public:
    EvalSession(std::shared_ptr<HipBackend> backend, EvalConfig config) : be_(std::move(backend)), config_(std::move(config)) {}
    EvalSession &add_codec(std::string id, std::string version, EncodeFn encode)
    {
        codecs_.push_back({std::move(id), std::move(version), std::move(encode), nullptr});
        return *this;
    }
    EvalSession &add_codec_with_decode(std::string id, std::string version, EncodeFn encode, DecodeFn decode)
    {
        codecs_.push_back({std::move(id), std::move(version), std::move(encode), std::move(decode)});
        return *this;
    }
    size_t codec_count() const { return codecs_.size(); }
    // the host's ICC -> sRGB transform for tagged DECODED images (session.rs:394); without one a tagged image fails like
    // a build without the `icc` feature (icc.rs:105-113).  Tables are built once per distinct profile and kept.
    EvalSession &set_cms(Cms cms)
    {
        cms_ = std::move(cms);
        return *this;
    }

    // evaluate_image, session.rs:368-434
    ImageReport evaluate_image(const std::string &name, const ImageData &image) const
    {
        std::vector<const ce_lut *> luts;
        ImageReport report{name, (uint32_t)image.width, (uint32_t)image.height, {}};
        const std::vector<uint8_t> reference_rgb = image.to_rgb8_vec();
        std::vector<std::vector<uint8_t>> decoded;  // kept alive until the batch has run
        std::vector<size_t> result_of_pair;
        decoded.reserve(codecs_.size() * config_.quality_levels.size());
        for (const auto &codec : codecs_)
            for (double quality : config_.quality_levels) {
                EncodeRequest request{quality, {}};
                const auto t0 = std::chrono::steady_clock::now();
                const std::vector<uint8_t> encoded = codec.encode(image, request);
                const auto t1 = std::chrono::steady_clock::now();
                CodecResult r;
                r.codec_id = codec.id;
                r.codec_version = codec.version;
                r.quality = quality;
                r.file_size = encoded.size();
                r.bits_per_pixel = (double)(encoded.size() * 8) / ((double)image.width * (double)image.height);
                r.encode_time = t1 - t0;
                r.codec_params = request.params;
                if (codec.decode) {
                    const auto d0 = std::chrono::steady_clock::now();
                    const ImageData dec = codec.decode(encoded);
                    r.decode_time = std::chrono::steady_clock::now() - d0;
                    decoded.push_back(dec.to_rgb8_vec());
                    luts.push_back(table_for(dec));  // to_rgb8_srgb (session.rs:394) runs on the device
                    result_of_pair.push_back(report.results.size());
                }
                report.results.push_back(std::move(r));
            }
        if (!decoded.empty()) {
            std::vector<ce_pair_desc> pairs(decoded.size());
            for (size_t i = 0; i < decoded.size(); i++)
                pairs[i] = {reference_rgb.data(), reference_rgb.size(), decoded[i].data(), decoded[i].size(), (uint32_t)image.width,
                            (uint32_t)image.height};
            std::vector<ce_scores> scores(decoded.size());
            const int rc = ce_eval_batch_lut(be_->ctx(), pairs.size(), pairs.data(), luts.data(), config_.metrics.mask(),
                                             config_.metrics.flags(), config_.intensity_target, scores.data());
            detail::check(*be_, rc, "batch", image.width, image.height, reference_rgb.size());
            for (size_t i = 0; i < decoded.size(); i++) {
                detail::check(*be_, scores[i].status, "metric", image.width, image.height, decoded[i].size());  // `?` in :394-396
                CodecResult &r = report.results[result_of_pair[i]];
                r.metrics = MetricResult::from_c(scores[i]);
                r.perception = r.metrics.perception_level();  // session.rs:407
            }
        }
        return report;
    }

private:
    struct CodecEntry {
        std::string id, version;
        EncodeFn encode;
        DecodeFn decode;
    };
    const ce_lut *table_for(const ImageData &decoded) const
    {
        if (!decoded.icc_profile) return nullptr;  // ColorProfile::Srgb: a plain copy (icc.rs:73)
        if (!cms_)
            throw Error(Error::Kind::MetricCalculation, "Metric calculation failed: ICC: ICC profile support requires the 'icc' feature");
        auto it = tables_.find(*decoded.icc_profile);
        if (it == tables_.end()) it = tables_.emplace(*decoded.icc_profile, std::make_unique<HipColorTable>(*be_, cms_, *decoded.icc_profile)).first;
        return it->second->get();
    }
    std::shared_ptr<HipBackend> be_;
    EvalConfig config_;
    std::vector<CodecEntry> codecs_;
    Cms cms_;
    mutable std::map<std::vector<uint8_t>, std::unique_ptr<HipColorTable>> tables_;
};

// ---- crates/codec-iter: the SSIMULACRA2 plug point ---------------------------------------------------------
// Ssimulacra2Reference::{new, compare} (fast-ssim2; used at crates/codec-iter/src/eval.rs:138-149,83-89 and
// crates/codec-compare/src/brute_force_sweep.rs:197-201,256): the source image is uploaded once and its
// reference-side state stays on the device for the whole quality sweep.
class Ssimulacra2Reference {
public:
    Ssimulacra2Reference(std::shared_ptr<HipBackend> be, const std::vector<uint8_t> &rgb, size_t width, size_t height)
        : be_(std::move(be)), w_(width), h_(height)
    {
        const int rc = ce_ref_create(be_->ctx(), rgb.data(), rgb.size(), (uint32_t)width, (uint32_t)height, 0, &ref_);
        if (rc != CE_OK) throw Error(Error::Kind::MetricCalculation, "SSIM2 reference error: " + be_->last_error());
    }
    ~Ssimulacra2Reference() { ce_ref_destroy(ref_); }
    Ssimulacra2Reference(const Ssimulacra2Reference &) = delete;
    Ssimulacra2Reference &operator=(const Ssimulacra2Reference &) = delete;

    double compare(const std::vector<uint8_t> &distorted) const
    {
        ce_scores s{};
        const int rc = ce_ref_compare(ref_, distorted.data(), distorted.size(), CE_METRIC_SSIMULACRA2, CE_DEFAULT_INTENSITY_TARGET, &s);
        detail::check(*be_, rc, "SSIMULACRA2", w_, h_, distorted.size());
        return s.ssimulacra2;
    }
    // the sweep `for q in qualities { reference.compare(decoded[q]) }` (eval.rs:83-89) as one launch
    std::vector<double> compare_many(const std::vector<std::vector<uint8_t>> &distorted) const
    {
        std::vector<const uint8_t *> ptrs;
        std::vector<size_t> lens;
        for (const auto &d : distorted) {
            ptrs.push_back(d.data());
            lens.push_back(d.size());
        }
        std::vector<ce_scores> s(distorted.size());
        const int rc = ce_ref_compare_many(ref_, ptrs.data(), lens.data(), (uint32_t)distorted.size(), CE_METRIC_SSIMULACRA2,
                                           CE_DEFAULT_INTENSITY_TARGET, s.data());
        detail::check(*be_, rc, "SSIMULACRA2", w_, h_, 0);
        std::vector<double> out;
        for (size_t i = 0; i < s.size(); i++) {
            detail::check(*be_, s[i].status, "SSIMULACRA2", w_, h_, lens[i]);
            out.push_back(s[i].ssimulacra2);
        }
        return out;
    }

private:
    std::shared_ptr<HipBackend> be_;
    ce_ref *ref_ = nullptr;
    size_t w_, h_;
};

// GpuSsim2 (crates/codec-iter/src/gpu.rs:21-134): `new(w, h)` fixes the shape, `compute(&mut self, ref, dis)`
// takes packed RGB8 from host memory and returns the score; one call in flight per object.
class GpuSsim2 {
public:
    GpuSsim2(uint32_t width, uint32_t height, int device = 0) : be_(std::make_shared<HipBackend>(device)), w_(width), h_(height) {}
    double compute(const std::vector<uint8_t> &reference, const std::vector<uint8_t> &distorted)
    {
        const size_t expected = (size_t)w_ * h_ * 3;
        if (reference.size() != expected || distorted.size() != expected)  // gpu.rs:84-94
            throw std::runtime_error("Image size mismatch: expected " + std::to_string(expected) + " bytes (" + std::to_string(w_) + "x" +
                                     std::to_string(h_) + "x3), got ref=" + std::to_string(reference.size()) +
                                     " dis=" + std::to_string(distorted.size()));
        double out = 0.0;
        const int rc = ce_calculate_ssimulacra2(be_->ctx(), reference.data(), reference.size(), distorted.data(), distorted.size(), w_, h_, &out);
        if (rc != CE_OK) throw std::runtime_error("SSIM2 HIP compute failed: " + be_->last_error());
        return out;
    }
    std::pair<uint32_t, uint32_t> dimensions() const { return {w_, h_}; }
    const std::shared_ptr<HipBackend> &backend() const { return be_; }

private:
    std::shared_ptr<HipBackend> be_;
    uint32_t w_, h_;
};

// Ssim2Backend (eval.rs:56-92).  The reference's enum has a Gpu and a Cpu arm; this library IS the device arm,
// so the mirror has that arm only (there is no CPU path in the product).  With a precomputed reference handle
// the source image is not uploaded again.
class Ssim2Backend {
public:
    explicit Ssim2Backend(std::unique_ptr<GpuSsim2> gpu) : gpu_(std::move(gpu)) {}
    double compare_with_precomputed(const std::vector<uint8_t> &source, const std::vector<uint8_t> &decoded,
                                    const Ssimulacra2Reference *reference, const std::string &image_name, unsigned quality)
    {
        try {
            return reference ? reference->compare(decoded) : gpu_->compute(source, decoded);
        } catch (const std::exception &e) {  // eval.rs:88
            throw std::runtime_error("SSIM2 error for " + image_name + " q" + std::to_string(quality) + ": " + e.what());
        }
    }
    GpuSsim2 &gpu() { return *gpu_; }

private:
    std::unique_ptr<GpuSsim2> gpu_;
};

// ---- src/eval/helpers.rs ---------------------------------------------------------------------------------
// evaluate_single, helpers.rs:105-173 (RGB8 images)
inline MetricResult evaluate_single(const HipBackend &be, const ImageData &reference, const ImageData &encoded, const MetricConfig &config)
{
    if (reference.width != encoded.width || reference.height != encoded.height)  // :111-116
        throw Error(Error::Kind::DimensionMismatch, "Dimension mismatch: expected (" + std::to_string(reference.width) + ", " +
                                                        std::to_string(reference.height) + "), got (" + std::to_string(encoded.width) +
                                                        ", " + std::to_string(encoded.height) + ")");
    const auto r = reference.to_rgb8_vec(), e = encoded.to_rgb8_vec();
    ce_scores s{};
    const int rc = ce_eval_pair(be.ctx(), r.data(), r.size(), e.data(), e.size(), (uint32_t)reference.width, (uint32_t)reference.height,
                                config.mask(), config.flags(), CE_DEFAULT_INTENSITY_TARGET, &s);
    detail::check(be, rc, "evaluate_single", reference.width, reference.height, e.size());
    return MetricResult::from_c(s);
}
// assert_quality, helpers.rs:212-255
inline void assert_quality(const HipBackend &be, const ImageData &reference, const ImageData &encoded,
                           std::optional<double> min_ssimulacra2, std::optional<double> max_dssim)
{
    MetricConfig cfg;
    cfg.dssim = max_dssim.has_value();
    cfg.ssimulacra2 = min_ssimulacra2.has_value();
    const MetricResult res = evaluate_single(be, reference, encoded, cfg);
    if (min_ssimulacra2 && res.ssimulacra2 && *res.ssimulacra2 < *min_ssimulacra2)
        throw Error(Error::Kind::QualityBelowThreshold, "SSIMULACRA2 quality below threshold: " + std::to_string(*res.ssimulacra2) +
                                                            " (threshold: " + std::to_string(*min_ssimulacra2) + ")");
    if (max_dssim && res.dssim && *res.dssim > *max_dssim)
        throw Error(Error::Kind::QualityBelowThreshold, "DSSIM quality below threshold: " + std::to_string(*res.dssim) +
                                                            " (threshold: " + std::to_string(*max_dssim) + ")");
}
// assert_perception_level, helpers.rs:291-321 (DSSIM only, ordinal compare)
inline void assert_perception_level(const HipBackend &be, const ImageData &reference, const ImageData &encoded, PerceptionLevel min_level)
{
    MetricConfig cfg;
    cfg.dssim = true;
    const MetricResult res = evaluate_single(be, reference, encoded, cfg);
    if (res.dssim && (uint8_t)perception_from_dssim(*res.dssim) > (uint8_t)min_level)
        throw Error(Error::Kind::QualityBelowThreshold, "PerceptionLevel (DSSIM " + std::to_string(*res.dssim) + ") below threshold");
}

}  // namespace eval
}  // namespace codec_eval
