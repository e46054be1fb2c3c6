"""Result wire formats: byte-compatibility with what the reference's tools read and write.

Pinned by `tests/golden/ref_baseline_jpeg.json`, a data file of the reference (`baselines/jpeg.json`, written by
codec-iter's `save_baseline` = `serde_json::to_string_pretty`): load -> save must reproduce it byte for byte, which
fixes the JSON layout, field order, ryu float formatting and chrono timestamp text."""
import datetime as dt
import glob
import importlib
import json
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
R = importlib.import_module("codec-eval_amd.reports")


def test_baseline_round_trips_byte_for_byte(tmp_path):
    src = os.path.join(ROOT, "tests", "golden", "ref_baseline_jpeg.json")
    raw = open(src, "rb").read()
    b = R.load_baseline(os.path.dirname(src), "ref_baseline_jpeg")
    assert b.format == "jpeg" and len(b.points) == 75 and b.points[0].image == "pexels-photo-951408.png"
    out = R.save_baseline(str(tmp_path), b)
    assert os.path.basename(out) == "jpeg.json"
    assert open(out, "rb").read() == raw


@pytest.mark.skipif(not os.path.isdir("/root/reference/baselines"), reason="reference checkout not present")
def test_every_reference_baseline_round_trips(tmp_path):
    files = sorted(glob.glob("/root/reference/baselines/*.json"))
    assert files
    for f in files:
        raw = open(f, "rb").read()
        d = json.loads(raw)
        b = R.Baseline(d["format"], d["config_summary"], d["corpus_path"], d["created_at"], [R.EvalPoint(**p) for p in d["points"]])
        assert R.to_string_pretty(b.to_obj()).encode() == raw, f


def test_ryu_float_text():
    # ryu `format64` layouts (serde_json): integers keep ".0", small/large magnitudes switch to e-notation
    cases = {
        0.0: "0.0", 1.0: "1.0", 80.0: "80.0", 0.72332763671875: "0.72332763671875", 67.06036004649532: "67.06036004649532",
        1e-5: "0.00001", 1e-6: "1e-6", 1.5e-7: "1.5e-7", 0.0003: "0.0003", 123456789012345680.0: "1.2345678901234568e17",
        1e16: "1e16", 1e15: "1000000000000000.0", 9007199254740993.0: "9007199254740992.0", 1.5e300: "1.5e300", -2.5: "-2.5", 5e-324: "5e-324",
        28.130803608679102: "28.130803608679102",
    }
    for x, want in cases.items():
        assert R.format_f64(x) == want, (x, R.format_f64(x), want)
    assert R.format_f64(float("inf")) == "null" and R.format_f64(float("nan")) == "null"
    # f64::to_string(): no exponent, no ".0"
    assert R.rust_f64_display(80.0) == "80" and R.rust_f64_display(85.5) == "85.5" and R.rust_f64_display(1e-7) == "0.0000001"
    assert R.rust_f64_display(1e21) == "1000000000000000000000"


def test_timestamps():
    t = dt.datetime(2026, 2, 12, 18, 4, 36, tzinfo=dt.timezone.utc)
    assert R.chrono_utc_default(t, 348566804) == "2026-02-12T18:04:36.348566804Z"  # the text in baselines/jpeg.json
    assert R.rfc3339(t, 348566804) == "2026-02-12T18:04:36.348566804+00:00"
    assert R.rfc3339(t, 0) == "2026-02-12T18:04:36+00:00"
    assert R.rfc3339(t, 120_000_000) == "2026-02-12T18:04:36.120+00:00"
    assert R.rfc3339(t.replace(microsecond=1500)) == "2026-02-12T18:04:36.001500+00:00"


def _corpus():
    t = dt.datetime(2025, 1, 2, 3, 4, 5, 678000, tzinfo=dt.timezone.utc)
    img = R.ImageReport("kodim01.png", 768, 512, timestamp=t)
    img.results.append(R.CodecResult("mozjpeg", "4.1.1", 80.0, 65536, 1.3333333333333333, 12, 3, dssim=0.00045678912, ssimulacra2=83.456,
                                     butteraugli=1.23456789, psnr=float("inf"), perception="MAR",
                                     codec_params={"subsampling": "4:2:0", "a": 'say "hi", ok'}))
    img.results.append(R.CodecResult("size,only", "0.1", 62.5, 1000, 0.02, 7))
    rep = R.CorpusReport("corpus", [img], t, "metrics: all")
    return rep


def test_report_json_layout():
    rep = _corpus()
    text = R.to_string_pretty(rep.to_obj())
    d = json.loads(text)
    assert list(d) == ["name", "images", "timestamp", "config_summary"]
    im = d["images"][0]
    assert list(im) == ["name", "source_path", "width", "height", "uncompressed_size", "results", "timestamp"]
    assert im["uncompressed_size"] == 768 * 512 * 3 and im["timestamp"] == "2025-01-02T03:04:05.678+00:00"
    r0 = im["results"][0]
    assert list(r0) == ["codec_id", "codec_version", "quality", "file_size", "bits_per_pixel", "encode_time", "decode_time",
                        "metrics", "perception", "cached_path", "codec_params"]
    assert list(r0["metrics"]) == ["dssim", "ssimulacra2", "butteraugli", "psnr"]  # src/metrics/mod.rs:140-149
    assert r0["perception"] == "Marginal" and r0["metrics"]["psnr"] is None  # +inf -> null
    assert '"quality": 80.0,' in text and '"encode_time": 12,' in text and '"decode_time": null' in text
    assert '      "results": [\n        {\n          "codec_id": "mozjpeg",' in text
    assert '"codec_params": {}' in text  # empty map on one line, like serde_json
    assert im["results"][1]["perception"] is None


def test_csv_summary(tmp_path):
    rep = _corpus()
    jpath, cpath = R.write_corpus_report(str(tmp_path / "reports"), rep)
    lines = open(cpath, newline="").read().split("\n")
    assert lines[0] == "image,codec,version,quality,file_size,bpp,encode_ms,decode_ms,dssim,ssimulacra2,butteraugli,psnr,perception"
    assert lines[1] == "kodim01.png,mozjpeg,4.1.1,80,65536,1.3333,12,3,0.000457,83.46,1.2346,inf,MAR"
    assert lines[2] == 'kodim01.png,"size,only",0.1,62.5,1000,0.0200,7,,,,,,'
    assert lines[3] == "" and len(lines) == 4
    assert json.load(open(jpath))["name"] == "corpus"
    p = R.write_image_report(str(tmp_path / "reports"), rep.images[0])
    assert os.path.basename(p) == "kodim01.png.json"


def test_report_queries_and_baseline_compare():
    rep = _corpus()
    img = rep.images[0]
    assert rep.total_results() == 2 and rep.codec_ids() == ["mozjpeg", "size,only"]
    assert img.best_at_size(70000).codec_id == "mozjpeg" and img.best_at_size(10) is None
    assert img.smallest_at_quality(0.001).codec_id == "mozjpeg" and img.smallest_at_quality(1e-6) is None
    assert abs(img.results[0].compression_ratio(768 * 512 * 3) - 18.0) < 1e-12
    base = R.load_baseline(os.path.join(ROOT, "tests", "golden"), "ref_baseline_jpeg")
    pts = [R.EvalPoint(p.image, p.quality, p.bpp * 0.9, p.ssim2 + 1.0, p.size_bytes, p.encode_ms) for p in base.points]
    rows = R.compare_with_baseline(pts, base)
    assert [r.quality for r in rows] == sorted({p.quality for p in base.points})
    for r in rows:
        assert abs(r.delta_ssim2 - 1.0) < 1e-9 and r.delta_bpp < 0 and abs(r.pareto - (r.delta_ssim2 - 10 * r.delta_bpp)) < 1e-12
