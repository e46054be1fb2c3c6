// codec_eval_multi.hpp — ONE process, every GPU of the node.
//
// The reference's EvalSession is Send + Sync and evaluate_image takes &self (src/eval/session.rs:368-434; the codec
// callbacks are `+ Send + Sync`, :181,186), and its multi-core tools fan whole images out to workers
// (`images.par_iter()`, crates/codec-compare/src/full_comparison.rs:319-328).  The device analogue (SURVEY.md §7 step 7,
// §8e): one host thread + one context (stream family) per GPU, all pulling WHOLE REFERENCES - every (codec, quality)
// cell of one source image - from one shared queue, so a reference is uploaded once, its reference-side planes are
// shared by its cells, and no work item ever crosses devices.  No collective, no peer traffic: scores come back to the
// host per device and land in result slots that were fixed before any worker started, so the output order is the
// input order whichever device scored what.
//
// Scheduling is guided self-scheduling: a worker takes max(1, remaining / (2 * workers)) references per pull (bounded
// by a device-memory budget), i.e. large batches while there is plenty of work (big launches fill the GPU) and single
// references at the end (devices finish together).  References are queued largest first (pixels x cells).
//
// Everything that is not a HIP call is a template parameter or a plain function, so the queue and the ordering are
// unit-tested on a host without GPUs (tests/cpp/test_host_mirror.cpp mocks the device count and the scorer).
#pragma once

#include <algorithm>
#include <atomic>
#include <exception>
#include <mutex>
#include <numeric>
#include <thread>

#include "codec_eval.hpp"

namespace codec_eval {
namespace eval {

// ---- the shared queue: indices 0..n-1 in a fixed service order, handed out in guided chunks ---------------------
class GuidedQueue {
public:
    // order: the service order (e.g. largest job first); cost[i] > 0: bytes (or any unit) job i needs on a device;
    // budget: the most a single pull may sum to (0 = unlimited; one job is always handed out even if it exceeds it)
    GuidedQueue(std::vector<size_t> order, std::vector<size_t> cost, size_t workers, size_t budget)
        : order_(std::move(order)), cost_(std::move(cost)), workers_(std::max<size_t>(workers, 1)), budget_(budget)
    {
    }
    // next chunk of job indices for one worker; empty when the queue is drained
    std::vector<size_t> pull()
    {
        std::lock_guard<std::mutex> lock(mu_);
        std::vector<size_t> out;
        const size_t remaining = order_.size() - next_;
        if (remaining == 0) return out;
        const size_t want = std::max<size_t>(1, remaining / (2 * workers_));
        size_t bytes = 0;
        while (next_ < order_.size() && out.size() < want) {
            const size_t j = order_[next_];
            if (!out.empty() && budget_ && bytes + cost_[j] > budget_) break;
            bytes += cost_[j];
            out.push_back(j);
            next_++;
        }
        return out;
    }
    size_t size() const { return order_.size(); }

private:
    std::vector<size_t> order_, cost_;
    size_t workers_, budget_, next_ = 0;
    std::mutex mu_;
};

// largest first, index as the tie-break (deterministic)
inline std::vector<size_t> largest_first(const std::vector<size_t> &load)
{
    std::vector<size_t> order(load.size());
    std::iota(order.begin(), order.end(), (size_t)0);
    std::stable_sort(order.begin(), order.end(), [&](size_t a, size_t b) { return load[a] > load[b]; });
    return order;
}

// ---- one reference's worth of work ----------------------------------------------------------------------------------
struct ReferenceJob {
    const uint8_t *reference = nullptr;  // packed RGB8, borrowed for the duration of run()
    uint32_t width = 0, height = 0;
    std::vector<const uint8_t *> tests;  // the distorted images of this reference (same shape), borrowed
    std::vector<int> test_profile;       // optional, per test: index into the profile list given to run(), -1 = untagged (sRGB)
    std::vector<ce_scores> scores;       // filled by run(): one per test, in the order of `tests`
    int device = -1;                     // which device scored it (diagnostics; never affects the values)
};

// Scorer: int(int worker, std::vector<ReferenceJob*> &chunk) - scores every job of the chunk on worker's device,
// returns a ce_status.  The default one drives ce_eval_batch on a per-worker context.
using ChunkScorer = std::function<int(int, std::vector<ReferenceJob *> &)>;

struct MultiDeviceStats {
    std::vector<size_t> jobs_per_device, pulls_per_device;
    std::vector<double> seconds_per_device;
};

// ---- the pool ----------------------------------------------------------------------------------------------------------
class DevicePool {
public:
    // n_devices <= 0: every visible HIP device.  Contexts are created here, one per device, each on its own device.
    explicit DevicePool(int n_devices = 0)
    {
        const int visible = HipBackend::device_count();
        const int n = n_devices > 0 ? n_devices : visible;
        if (n <= 0 || n > visible)
            throw Error(Error::Kind::MetricCalculation, "HIP init failed: " + std::to_string(visible) + " device(s) visible, " + std::to_string(n) + " requested");
        for (int d = 0; d < n; d++) backends_.push_back(std::make_shared<HipBackend>(d));
    }
    // for tests: no contexts, `workers` mock devices, every chunk goes to `scorer`
    DevicePool(int workers, ChunkScorer scorer) : mock_workers_(workers), scorer_(std::move(scorer)) {}

    int devices() const { return mock_workers_ ? mock_workers_ : (int)backends_.size(); }
    const std::shared_ptr<HipBackend> &backend(int d) const { return backends_.at((size_t)d); }

    // Scores every job.  jobs[i].scores is filled for every i; results do not depend on the device count or on which
    // device took which job.  Throws the first worker's error after all workers have stopped.
    // profiles + cms: the ICC profiles the jobs' test_profile indices refer to and the host colour management; every
    // worker builds the colour table of a profile on ITS device the first time it meets it (a table belongs to a device)
    MultiDeviceStats run(std::vector<ReferenceJob> &jobs, const MetricConfig &metrics, float intensity_target = CE_DEFAULT_INTENSITY_TARGET,
                         size_t device_budget_bytes = 0, const std::vector<std::vector<uint8_t>> *profiles = nullptr, const Cms *cms = nullptr)
    {
        const int n_workers = devices();
        profiles_ = profiles;
        cms_ = cms;
        tables_.clear();
        tables_.resize((size_t)n_workers);
        std::vector<size_t> load(jobs.size()), cost(jobs.size());
        for (size_t i = 0; i < jobs.size(); i++) {
            load[i] = (size_t)jobs[i].width * jobs[i].height * std::max<size_t>(jobs[i].tests.size(), 1);
            cost[i] = ce_estimate_batch_bytes(jobs[i].width, jobs[i].height, 1, (uint32_t)jobs[i].tests.size(), metrics.mask());
            jobs[i].scores.assign(jobs[i].tests.size(), ce_scores{});
            jobs[i].device = -1;
        }
        size_t budget = device_budget_bytes;
        if (!budget && !mock_workers_) {  // a third of the smallest device's free memory: ce_eval_batch streams chunks through a ring of three
            size_t fr = 0, tot = 0, smallest = ~(size_t)0;
            for (auto &be : backends_)
                if (ce_ctx_memory_info(be->ctx(), &fr, &tot) == CE_OK) smallest = std::min(smallest, fr);
            if (smallest != ~(size_t)0) budget = smallest / 3;
        }
        GuidedQueue queue(largest_first(load), cost, (size_t)n_workers, budget);
        MultiDeviceStats st;
        st.jobs_per_device.assign((size_t)n_workers, 0);
        st.pulls_per_device.assign((size_t)n_workers, 0);
        st.seconds_per_device.assign((size_t)n_workers, 0.0);
        std::atomic<bool> failed{false};
        std::mutex err_mu;
        std::exception_ptr first_error;
        auto worker = [&](int w) {
            const auto t0 = std::chrono::steady_clock::now();
            try {
                for (;;) {
                    if (failed.load()) break;
                    std::vector<size_t> idx = queue.pull();
                    if (idx.empty()) break;
                    std::vector<ReferenceJob *> chunk;
                    for (size_t j : idx) {
                        jobs[j].device = w;
                        chunk.push_back(&jobs[j]);
                    }
                    const int rc = scorer_ ? scorer_(w, chunk) : score_chunk(w, chunk, metrics, intensity_target);
                    if (rc != CE_OK)
                        throw Error(Error::Kind::MetricCalculation,
                                    "Metric calculation failed: device " + std::to_string(w) + ": " + (scorer_ ? std::string("scorer") : backends_[(size_t)w]->last_error()));
                    st.jobs_per_device[(size_t)w] += idx.size();
                    st.pulls_per_device[(size_t)w]++;
                }
            } catch (...) {
                failed.store(true);
                std::lock_guard<std::mutex> lock(err_mu);
                if (!first_error) first_error = std::current_exception();
            }
            st.seconds_per_device[(size_t)w] = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        };
        std::vector<std::thread> threads;
        for (int w = 1; w < n_workers; w++) threads.emplace_back(worker, w);
        worker(0);
        for (auto &t : threads) t.join();
        if (first_error) std::rethrow_exception(first_error);
        return st;
    }

private:
    // every cell of every job of the chunk in ONE ce_eval_batch call on this worker's context (identical reference
    // pointers share one device slot; mixed shapes are bucketed inside)
    int score_chunk(int w, std::vector<ReferenceJob *> &chunk, const MetricConfig &metrics, float intensity_target)
    {
        std::vector<ce_pair_desc> pairs;
        std::vector<const ce_lut *> luts;
        for (ReferenceJob *j : chunk) {
            const size_t len = (size_t)j->width * j->height * 3;
            for (size_t t = 0; t < j->tests.size(); t++) {
                pairs.push_back({j->reference, len, j->tests[t], len, j->width, j->height});
                const int pi = t < j->test_profile.size() ? j->test_profile[t] : -1;
                const ce_lut *lut = nullptr;
                if (pi >= 0) {
                    if (!profiles_ || !cms_ || (size_t)pi >= profiles_->size()) return CE_ERR_INVALID_ARG;
                    auto &slot = tables_[(size_t)w][pi];
                    if (!slot) slot = std::make_unique<HipColorTable>(*backends_[(size_t)w], *cms_, (*profiles_)[(size_t)pi]);
                    lut = slot->get();
                }
                luts.push_back(lut);
            }
        }
        if (pairs.empty()) return CE_OK;
        std::vector<ce_scores> out(pairs.size());
        const int rc = ce_eval_batch_lut(backends_[(size_t)w]->ctx(), pairs.size(), pairs.data(), luts.data(), metrics.mask(), metrics.flags(),
                                         intensity_target, out.data());
        if (rc != CE_OK) return rc;
        size_t k = 0;
        for (ReferenceJob *j : chunk)
            for (size_t t = 0; t < j->tests.size(); t++) j->scores[t] = out[k++];
        return CE_OK;
    }

    std::vector<std::shared_ptr<HipBackend>> backends_;
    int mock_workers_ = 0;
    ChunkScorer scorer_;
    const std::vector<std::vector<uint8_t>> *profiles_ = nullptr;
    const Cms *cms_ = nullptr;
    std::vector<std::map<int, std::unique_ptr<HipColorTable>>> tables_;  // [worker][profile index], each touched by its worker only
};

// ---- EvalSession over all devices ------------------------------------------------------------------------------------------
// evaluate_corpus: every image through the (codec x quality) sweep (callbacks run on the calling thread, in the
// reference's loop order - they are the caller's code), then all decoded cells of all images are scored by the pool.
// reports[i] belongs to images[i]; within a report the rows are in the reference's loop order (session.rs:375-410).
class MultiDeviceEvalSession {
public:
    MultiDeviceEvalSession(std::shared_ptr<DevicePool> pool, EvalConfig config) : pool_(std::move(pool)), config_(std::move(config)) {}
    MultiDeviceEvalSession &add_codec(std::string id, std::string version, EncodeFn encode)
    {
        codecs_.push_back({std::move(id), std::move(version), std::move(encode), nullptr});
        return *this;
    }
    MultiDeviceEvalSession &add_codec_with_decode(std::string id, std::string version, EncodeFn encode, DecodeFn decode)
    {
        codecs_.push_back({std::move(id), std::move(version), std::move(encode), std::move(decode)});
        return *this;
    }
    size_t codec_count() const { return codecs_.size(); }
    MultiDeviceEvalSession &set_cms(Cms cms)  // EvalSession::set_cms: ICC -> sRGB of tagged decoded images (session.rs:394)
    {
        cms_ = std::move(cms);
        return *this;
    }

    std::vector<ImageReport> evaluate_corpus(const std::vector<std::pair<std::string, ImageData>> &images, MultiDeviceStats *stats = nullptr) const
    {
        std::vector<ImageReport> reports(images.size());
        std::vector<std::vector<uint8_t>> references(images.size());
        std::vector<std::vector<std::vector<uint8_t>>> decoded(images.size());
        std::vector<std::vector<size_t>> row_of_cell(images.size());
        std::vector<std::vector<int>> profile_of_cell(images.size());
        std::vector<std::vector<uint8_t>> profiles;  // distinct ICC profiles met among the decoded images
        for (size_t i = 0; i < images.size(); i++) {
            const ImageData &image = images[i].second;
            reports[i] = ImageReport{images[i].first, (uint32_t)image.width, (uint32_t)image.height, {}};
            references[i] = image.to_rgb8_vec();
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
                        if (dec.width != image.width || dec.height != image.height)  // calculate_metrics' length check
                            throw Error(Error::Kind::DimensionMismatch, "Dimension mismatch: expected (" + std::to_string(image.width) + ", " +
                                                                            std::to_string(image.height) + "), got (" + std::to_string(dec.width) + ", " +
                                                                            std::to_string(dec.height) + ")");
                        decoded[i].push_back(dec.to_rgb8_vec());
                        int pi = -1;
                        if (dec.icc_profile) {
                            if (!cms_)
                                throw Error(Error::Kind::MetricCalculation, "Metric calculation failed: ICC: ICC profile support requires the 'icc' feature");
                            const auto it = std::find(profiles.begin(), profiles.end(), *dec.icc_profile);
                            pi = (int)(it - profiles.begin());
                            if (it == profiles.end()) profiles.push_back(*dec.icc_profile);
                        }
                        profile_of_cell[i].push_back(pi);
                        row_of_cell[i].push_back(reports[i].results.size());
                    }
                    reports[i].results.push_back(std::move(r));
                }
        }
        std::vector<ReferenceJob> jobs(images.size());
        for (size_t i = 0; i < images.size(); i++) {
            jobs[i].reference = references[i].data();
            jobs[i].width = (uint32_t)images[i].second.width;
            jobs[i].height = (uint32_t)images[i].second.height;
            for (const auto &d : decoded[i]) jobs[i].tests.push_back(d.data());
            jobs[i].test_profile = profile_of_cell[i];
        }
        const MultiDeviceStats st = pool_->run(jobs, config_.metrics, config_.intensity_target, 0, &profiles, cms_ ? &cms_ : nullptr);
        if (stats) *stats = st;
        for (size_t i = 0; i < images.size(); i++)
            for (size_t c = 0; c < jobs[i].scores.size(); c++) {
                const ce_scores &s = jobs[i].scores[c];
                if (s.status != CE_OK)
                    throw Error(s.status == CE_ERR_DIM_MISMATCH ? Error::Kind::DimensionMismatch : Error::Kind::MetricCalculation,
                                "Metric calculation failed: " + images[i].first + ": status " + std::to_string(s.status));
                CodecResult &r = reports[i].results[row_of_cell[i][c]];
                r.metrics = MetricResult::from_c(s);
                r.perception = r.metrics.perception_level();  // session.rs:407
            }
        return reports;
    }

private:
    struct CodecEntry {
        std::string id, version;
        EncodeFn encode;
        DecodeFn decode;
    };
    std::shared_ptr<DevicePool> pool_;
    EvalConfig config_;
    std::vector<CodecEntry> codecs_;
    Cms cms_;
};

}  // namespace eval
}  // namespace codec_eval
