// Report writers of the C++ host mirror: the reference's wire formats, byte for byte.
//   ImageReport JSON  = serde_json::to_string_pretty (src/eval/session.rs:500-508, structs src/eval/report.rs:14-107)
//   CSV summary       = src/eval/session.rs:526-584 (csv crate: quote when necessary, "\n" terminator)
// serde_json prints f64 with ryu: shortest digits that round-trip, laid out by ryu's rules (decimal for exponents in
// [-5, 16), otherwise d.ddde±x), non-finite values as null, 2-space indent, fields in declaration order.  The same
// rules are implemented in codec-eval_amd/reports.py (pinned on the reference's own baselines/*.json); the test
// tests/test_host_cpp.py compares this header's output with that module's.
#pragma once
#include <charconv>
#include <cmath>
#include <cstdio>
#include <sstream>
#include <string>

#include "codec_eval.hpp"

namespace codec_eval {
namespace report {

// shortest round-trip digits and decimal exponent: |x| = 0.d1..dn * 10^(n + k)
inline void digits_exp(double x, std::string &digits, int &k)
{
    char buf[64];
    auto r = std::to_chars(buf, buf + sizeof buf, std::fabs(x), std::chars_format::scientific);  // shortest, d.ddde±xx
    std::string s(buf, r.ptr);
    const size_t e = s.find('e');
    const int exp10 = std::stoi(s.substr(e + 1));
    std::string m = s.substr(0, e);
    digits.clear();
    for (char c : m)
        if (c != '.') digits.push_back(c);
    while (digits.size() > 1 && digits.back() == '0') digits.pop_back();
    k = exp10 - (int)digits.size() + 1;
}

inline std::string format_f64(double x)  // ryu's format64 as serde_json uses it
{
    if (!std::isfinite(x)) return "null";
    if (x == 0.0) return std::signbit(x) ? "-0.0" : "0.0";
    std::string d;
    int k;
    digits_exp(x, d, k);
    const int n = (int)d.size(), kk = n + k;
    std::string out = x < 0 ? "-" : "";
    if (0 <= k && kk <= 16) return out + d + std::string((size_t)k, '0') + ".0";
    if (0 < kk && kk <= 16) return out + d.substr(0, (size_t)kk) + "." + d.substr((size_t)kk);
    if (-5 < kk && kk <= 0) return out + "0." + std::string((size_t)(-kk), '0') + d;
    const int e = kk - 1;
    if (n == 1) return out + d + "e" + std::to_string(e);
    return out + d.substr(0, 1) + "." + d.substr(1) + "e" + std::to_string(e);
}

inline std::string rust_f64_display(double x)  // f64::to_string(): no exponent, no trailing ".0"
{
    if (std::isnan(x)) return "NaN";
    if (std::isinf(x)) return x > 0 ? "inf" : "-inf";
    if (x == 0.0) return std::signbit(x) ? "-0" : "0";
    std::string d;
    int k;
    digits_exp(x, d, k);
    const int n = (int)d.size(), kk = n + k;
    std::string out = x < 0 ? "-" : "";
    if (k >= 0) return out + d + std::string((size_t)k, '0');
    if (kk > 0) return out + d.substr(0, (size_t)kk) + "." + d.substr((size_t)kk);
    return out + "0." + std::string((size_t)(-kk), '0') + d;
}

inline std::string json_str(const std::string &s)
{
    std::string o = "\"";
    for (unsigned char c : s) {
        switch (c) {
            case '"': o += "\\\""; break;
            case '\\': o += "\\\\"; break;
            case '\b': o += "\\b"; break;
            case '\f': o += "\\f"; break;
            case '\n': o += "\\n"; break;
            case '\r': o += "\\r"; break;
            case '\t': o += "\\t"; break;
            default:
                if (c < 0x20) {
                    char b[8];
                    std::snprintf(b, sizeof b, "\\u%04x", c);
                    o += b;
                } else {
                    o.push_back((char)c);
                }
        }
    }
    return o + "\"";
}

inline const char *perception_name(PerceptionLevel l)
{
    switch (l) {
        case PerceptionLevel::Imperceptible: return "Imperceptible";
        case PerceptionLevel::Marginal: return "Marginal";
        case PerceptionLevel::Subtle: return "Subtle";
        case PerceptionLevel::Noticeable: return "Noticeable";
        default: return "Degraded";
    }
}

inline std::string opt_f64(const std::optional<double> &v) { return v ? format_f64(*v) : "null"; }

// `timestamp` is the RFC 3339 text chrono's to_rfc3339() would print (the caller owns the clock)
inline std::string image_report_json(const eval::ImageReport &r, const std::string &timestamp, int indent = 0)
{
    const std::string p0((size_t)indent * 2, ' '), p1 = p0 + "  ", p2 = p1 + "  ", p3 = p2 + "  ", p4 = p3 + "  ";
    std::ostringstream o;
    o << "{\n" << p1 << "\"name\": " << json_str(r.name) << ",\n" << p1 << "\"source_path\": null,\n";
    o << p1 << "\"width\": " << r.width << ",\n" << p1 << "\"height\": " << r.height << ",\n";
    o << p1 << "\"uncompressed_size\": " << (size_t)r.width * r.height * 3 << ",\n" << p1 << "\"results\": ";
    if (r.results.empty()) {
        o << "[]";
    } else {
        o << "[\n";
        for (size_t i = 0; i < r.results.size(); i++) {
            const eval::CodecResult &c = r.results[i];
            const auto ms = [](std::chrono::nanoseconds d) { return std::chrono::duration_cast<std::chrono::milliseconds>(d).count(); };
            o << p2 << "{\n";
            o << p3 << "\"codec_id\": " << json_str(c.codec_id) << ",\n" << p3 << "\"codec_version\": " << json_str(c.codec_version) << ",\n";
            o << p3 << "\"quality\": " << format_f64(c.quality) << ",\n" << p3 << "\"file_size\": " << c.file_size << ",\n";
            o << p3 << "\"bits_per_pixel\": " << format_f64(c.bits_per_pixel) << ",\n" << p3 << "\"encode_time\": " << ms(c.encode_time) << ",\n";
            o << p3 << "\"decode_time\": " << (c.decode_time ? std::to_string(ms(*c.decode_time)) : std::string("null")) << ",\n";
            o << p3 << "\"metrics\": {\n" << p4 << "\"dssim\": " << opt_f64(c.metrics.dssim) << ",\n" << p4 << "\"ssimulacra2\": "
              << opt_f64(c.metrics.ssimulacra2) << ",\n" << p4 << "\"butteraugli\": " << opt_f64(c.metrics.butteraugli) << ",\n" << p4
              << "\"psnr\": " << opt_f64(c.metrics.psnr) << "\n" << p3 << "},\n";
            o << p3 << "\"perception\": " << (c.perception ? json_str(perception_name(*c.perception)) : std::string("null")) << ",\n";
            o << p3 << "\"cached_path\": null,\n" << p3 << "\"codec_params\": ";
            if (c.codec_params.empty()) {
                o << "{}";
            } else {  // std::map: sorted keys (a Rust HashMap has no stable order)
                o << "{\n";
                size_t k = 0;
                for (const auto &kv : c.codec_params)
                    o << p4 << json_str(kv.first) << ": " << json_str(kv.second) << (++k < c.codec_params.size() ? ",\n" : "\n");
                o << p3 << "}";
            }
            o << "\n" << p2 << "}" << (i + 1 < r.results.size() ? ",\n" : "\n");
        }
        o << p1 << "]";
    }
    o << ",\n" << p1 << "\"timestamp\": " << json_str(timestamp) << "\n" << p0 << "}";
    return o.str();
}

inline std::string csv_field(const std::string &s)
{
    if (s.find_first_of(",\"\n\r") == std::string::npos) return s;
    std::string o = "\"";
    for (char c : s) {
        if (c == '"') o.push_back('"');
        o.push_back(c);
    }
    return o + "\"";
}

inline std::string fixed(const std::optional<double> &v, int places)
{
    if (!v) return "";
    if (std::isnan(*v)) return "NaN";
    if (std::isinf(*v)) return *v > 0 ? "inf" : "-inf";
    char b[64];
    std::snprintf(b, sizeof b, "%.*f", places, *v);  // correctly rounded from the exact binary value, like Rust's {:.N}
    return b;
}

// the 13-column summary of a corpus (session.rs:526-584); `images` in report order
inline std::string csv_summary(const std::vector<eval::ImageReport> &images)
{
    std::string o = "image,codec,version,quality,file_size,bpp,encode_ms,decode_ms,dssim,ssimulacra2,butteraugli,psnr,perception\n";
    const auto ms = [](std::chrono::nanoseconds d) { return std::to_string(std::chrono::duration_cast<std::chrono::milliseconds>(d).count()); };
    for (const auto &img : images)
        for (const auto &c : img.results) {
            const std::string row[13] = {img.name, c.codec_id, c.codec_version, rust_f64_display(c.quality), std::to_string(c.file_size),
                                         fixed(c.bits_per_pixel, 4), ms(c.encode_time), c.decode_time ? ms(*c.decode_time) : std::string(),
                                         fixed(c.metrics.dssim, 6), fixed(c.metrics.ssimulacra2, 2), fixed(c.metrics.butteraugli, 4),
                                         fixed(c.metrics.psnr, 2), c.perception ? std::string(perception_code(*c.perception)) : std::string()};
            for (int i = 0; i < 13; i++) o += csv_field(row[i]) + (i < 12 ? "," : "\n");
        }
    return o;
}

}  // namespace report
}  // namespace codec_eval
