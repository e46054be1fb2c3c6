// Prints the C++ report writers' output for a fixed report; tests/test_host_cpp.py compares it byte for byte with
// codec-eval_amd/reports.py (which is pinned on the reference's own baselines/*.json).
#include <cstdio>
#include <limits>

#include "codec_eval_report.hpp"

using namespace codec_eval;

int main()
{
    eval::ImageReport img;
    img.name = "kodim01.png";
    img.width = 768;
    img.height = 512;
    eval::CodecResult a;
    a.codec_id = "mozjpeg";
    a.codec_version = "4.1.1";
    a.quality = 80.0;
    a.file_size = 65536;
    a.bits_per_pixel = 1.3333333333333333;
    a.encode_time = std::chrono::milliseconds(12);
    a.decode_time = std::chrono::milliseconds(3);
    a.metrics.dssim = 0.00045678912;
    a.metrics.ssimulacra2 = 83.456;
    a.metrics.butteraugli = 1.23456789;
    a.metrics.psnr = std::numeric_limits<double>::infinity();
    a.perception = PerceptionLevel::Marginal;
    a.codec_params = {{"subsampling", "4:2:0"}, {"a", "say \"hi\", ok"}};
    eval::CodecResult b;
    b.codec_id = "size,only";
    b.codec_version = "0.1";
    b.quality = 62.5;
    b.file_size = 1000;
    b.bits_per_pixel = 0.02;
    b.encode_time = std::chrono::milliseconds(7);
    img.results = {a, b};
    std::printf("%s\n----\n%s----\n", report::image_report_json(img, "2025-01-02T03:04:05.678+00:00").c_str(), report::csv_summary({img}).c_str());
    for (double v : {0.0, 1.0, 80.0, 0.72332763671875, 1e-5, 1e-6, 1.5e-7, 1e16, 1e15, 123456789012345680.0, 5e-324, 28.130803608679102, -2.5})
        std::printf("%s %s\n", report::format_f64(v).c_str(), report::rust_f64_display(v).c_str());
    return 0;
}
