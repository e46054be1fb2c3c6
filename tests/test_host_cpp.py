"""The C++ host mirror (codec-eval_amd/host/codec_eval.hpp) compiled against the C ABI with plain g++ and run
as the reference's own unit tests would run (tests/cpp/test_host_mirror.cpp)."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _build(ce, tmp_path):
    exe = str(tmp_path / "test_host_mirror")
    libdir = os.path.dirname(ce.LIB_PATH)
    subprocess.check_call([
        "g++", "-std=c++17", "-O1", "-Wall", "-Wextra", "-Werror",
        "-I", os.path.join(ROOT, "include"), "-I", os.path.join(ROOT, "codec-eval_amd", "host"),
        os.path.join(ROOT, "tests", "cpp", "test_host_mirror.cpp"), "-o", exe,
        "-L", libdir, "-lce_metrics_hip", f"-Wl,-rpath,{libdir}", "-pthread",
    ])
    return exe


def test_host_mirror_compiles_and_host_logic(ce, tmp_path):
    exe = _build(ce, tmp_path)
    out = subprocess.run([exe, "cpu"], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr


@pytest.mark.gpu
def test_host_mirror_on_gpu(ce, tmp_path):
    exe = _build(ce, tmp_path)
    out = subprocess.run([exe, "gpu"], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr


def test_cpp_report_writers_match_the_python_ones(ce, tmp_path):
    """codec_eval_report.hpp (JSON + CSV of the reference's wire formats) against codec-eval_amd/reports.py, which is
    pinned on the reference's own baselines/*.json."""
    import datetime as dt
    import importlib

    R = importlib.import_module("codec-eval_amd.reports")
    exe = str(tmp_path / "test_report_writers")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-Wall", "-Wextra", "-Werror", "-I", os.path.join(ROOT, "include"),
                           "-I", os.path.join(ROOT, "codec-eval_amd", "host"), os.path.join(ROOT, "tests", "cpp", "test_report_writers.cpp"),
                           "-o", exe])
    out = subprocess.run([exe], capture_output=True, text=True)
    assert out.returncode == 0, out.stderr
    json_text, csv_text, floats = out.stdout.split("----\n")
    t = dt.datetime(2025, 1, 2, 3, 4, 5, 678000, tzinfo=dt.timezone.utc)
    img = R.ImageReport("kodim01.png", 768, 512, timestamp=t)
    img.results.append(R.CodecResult("mozjpeg", "4.1.1", 80.0, 65536, 1.3333333333333333, 12, 3, dssim=0.00045678912, ssimulacra2=83.456,
                                     butteraugli=1.23456789, psnr=float("inf"), perception="MAR",
                                     codec_params={"subsampling": "4:2:0", "a": 'say "hi", ok'}))
    img.results.append(R.CodecResult("size,only", "0.1", 62.5, 1000, 0.02, 7))
    assert json_text == R.to_string_pretty(img.to_obj()) + "\n"
    assert csv_text == R.csv_summary(R.CorpusReport("c", [img], t))
    vals = [0.0, 1.0, 80.0, 0.72332763671875, 1e-5, 1e-6, 1.5e-7, 1e16, 1e15, 123456789012345680.0, 5e-324, 28.130803608679102, -2.5]
    assert floats.splitlines() == [f"{R.format_f64(v)} {R.rust_f64_display(v)}" for v in vals]
