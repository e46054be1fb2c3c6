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
        "-L", libdir, "-lce_metrics_hip", f"-Wl,-rpath,{libdir}",
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
