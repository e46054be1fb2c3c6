"""CPU-side checks of the drop-in boundary: the C-ABI library builds, loads and exports every symbol
include/ce_metrics.h declares; without a GPU every entry point fails loudly (no CPU fallback)."""
import ctypes
import os
import re
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_functions():
    """Every function any header under include/ declares (the boundary, ce_metrics.h, and the test hooks, ce_metrics_debug.h)."""
    inc = os.path.join(ROOT, "include")
    text = "".join(open(os.path.join(inc, f)).read() for f in sorted(os.listdir(inc)) if f.endswith(".h"))
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(ce_[a-z0-9_]+)\s*\(", text)))


def test_header_is_plain_c(tmp_path):
    """extern "C", plain pointers and sizes: the header must compile as C99 and as C++."""
    src = tmp_path / "t.c"
    src.write_text('#include "ce_metrics.h"\n#include "ce_metrics_debug.h"\nint main(void){ce_scores s; (void)s; return CE_OK;}\n')
    inc = os.path.join(ROOT, "include")
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Werror", "-pedantic", "-I", inc, "-c", str(src), "-o", str(tmp_path / "t.o")])
    subprocess.check_call(["g++", "-std=c++17", "-Wall", "-Werror", "-I", inc, "-x", "c++", "-c", str(src), "-o", str(tmp_path / "t2.o")])
    text = open(os.path.join(inc, "ce_metrics.h")).read() + open(os.path.join(inc, "ce_metrics_debug.h")).read()
    assert "torch" not in text.lower().replace("no torch", "") and "std::" not in text
    # the boundary itself declares no test hook
    assert "ce_debug_" not in re.sub(r"/\*.*?\*/", "", open(os.path.join(inc, "ce_metrics.h")).read(), flags=re.S)


def test_library_exports_every_declared_symbol(ce):
    declared = _declared_functions()
    assert len(declared) >= 30
    lib = ce.lib()
    missing = [n for n in declared if not hasattr(lib, n)]
    assert not missing, f"declared in ce_metrics.h but not exported: {missing}"
    unbound = [n for n in declared if n not in ce.ABI_SYMBOLS]
    assert not unbound, f"declared in ce_metrics.h but not bound in the ctypes layer: {unbound}"
    out = subprocess.check_output(["nm", "-D", "--defined-only", ce.LIB_PATH], text=True)
    exported = set(re.findall(r" T (ce_[a-z0-9_]+)", out))
    assert set(declared) <= exported


def test_library_has_gfx950_code_object(ce, tmp_path):
    # llvm-objdump --offloading drops the extracted code objects next to its input: work on a copy
    import shutil

    lib_copy = shutil.copy(ce.LIB_PATH, tmp_path / "lib.so")
    out = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-objdump", "--offloading", str(lib_copy)], capture_output=True, text=True)
    blob = out.stdout + out.stderr
    if "gfx950" not in blob:  # older objdump: fall back to the embedded bundle id
        blob = subprocess.check_output(["strings", ce.LIB_PATH], text=True)
    assert "gfx950" in blob


def test_enum_values_match_header(ce):
    text = open(os.path.join(ROOT, "include", "ce_metrics.h")).read()
    for name, val in [("CE_OK", 0), ("CE_ERR_DIM_MISMATCH", 1), ("CE_ERR_BAD_LENGTH", 2), ("CE_ERR_TOO_SMALL", 3),
                      ("CE_ERR_BACKEND", 4), ("CE_ERR_INVALID_ARG", 5)]:
        assert re.search(rf"{name}\s*=\s*{val}\b", text)
        assert getattr(ce, name) == val
    assert (ce.METRIC_DSSIM, ce.METRIC_SSIMULACRA2, ce.METRIC_BUTTERAUGLI, ce.METRIC_PSNR) == (1, 2, 4, 8)
    assert ctypes.sizeof(ce.CeScores) == 40 and ctypes.sizeof(ce.CePairDesc) == 40


def test_no_cpu_fallback_without_gpu(ce):
    """On a host with no HIP device the product must fail loudly, never compute on the CPU."""
    if ce.device_count() > 0:
        pytest.skip("a GPU is visible; the no-device path is exercised on CPU-only hosts")
    with pytest.raises(ce.CodecEvalError) as e:
        ce.Context(0)
    assert e.value.status == ce.CE_ERR_BACKEND
    a = np.zeros(8 * 8 * 3, np.uint8)
    out = ctypes.c_double()
    assert ce.lib().ce_calculate_psnr(None, a.ctypes.data, a.size, a.ctypes.data, a.size, 8, 8, ctypes.byref(out)) == ce.CE_ERR_INVALID_ARG
    # the page-locked allocator needs a context too; freeing nothing is fine
    p = ctypes.c_void_p()
    assert ce.lib().ce_host_alloc(None, 4096, ctypes.byref(p)) == ce.CE_ERR_INVALID_ARG and not p.value
    assert ce.lib().ce_host_free(None, None) == ce.CE_OK


def test_product_never_touches_the_oracle():
    """The oracle is test infrastructure: nothing under codec-eval_amd/ may import, link or dlopen it."""
    pkg = os.path.join(ROOT, "codec-eval_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".h", "Makefile")):
                text = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "libce_oracle" not in text and "from oracle" not in text and "import oracle" not in text, f
    out = subprocess.check_output(["ldd", os.path.join(pkg, "libce_metrics_hip.so")], text=True)
    assert "oracle" not in out


def test_metric_config_presets(ce):
    # src/metrics/mod.rs:65-136
    assert ce.MetricConfig.all().mask == 15 and ce.MetricConfig.all().flags == 0
    assert ce.MetricConfig.fast().mask == ce.METRIC_PSNR
    assert ce.MetricConfig.perceptual().mask == 7
    assert ce.MetricConfig.perceptual_xyb().flags == ce.FLAG_XYB_ROUNDTRIP
    assert ce.MetricConfig.ssimulacra2_only().mask == ce.METRIC_SSIMULACRA2
    assert ce.MetricConfig.fast().with_xyb_roundtrip().xyb_roundtrip


def test_perception_levels(ce):
    # src/metrics/mod.rs:338-366
    f = ce.perception_from_dssim
    assert f(0.0001) == "Imperceptible" and f(0.0003) == "Marginal" and f(0.0005) == "Marginal"
    assert f(0.0007) == "Subtle" and f(0.001) == "Subtle" and f(0.0015) == "Noticeable"
    assert f(0.002) == "Noticeable" and f(0.003) == "Degraded" and f(0.01) == "Degraded"
    assert ce.perception_from_ssimulacra2(90.0) == "Marginal" and ce.perception_from_ssimulacra2(90.1) == "Imperceptible"
    assert ce.perception_from_butteraugli(0.99) == "Imperceptible" and ce.perception_from_butteraugli(5.0) == "Degraded"
    assert ce.MetricResult(dssim=0.0001).perception_level() == "Imperceptible"
    assert ce.MetricResult().perception_level() is None


def test_image_data_icc_semantics_on_the_host(ce):
    """ImageData mirrors session.rs:25-149: to_rgb8_vec never applies a profile; to_rgb8_srgb applies it through the
    caller's CMS (transform_to_srgb, icc.rs:69-103) and, without one, fails like a build without the `icc` feature."""
    import importlib

    import numpy as np
    import pytest

    S = importlib.import_module("codec-eval_amd.session")
    px = (np.arange(4 * 3 * 3) % 256).astype(np.uint8)
    plain, tagged = S.ImageData.rgb(px, 4, 3), S.ImageData.rgb_with_icc(px, 4, 3, b"profile")
    assert np.array_equal(plain.to_rgb8_srgb(), px) and np.array_equal(tagged.to_rgb8_vec(), px)
    with pytest.raises(ce.MetricCalculation, match="requires the 'icc' feature"):
        tagged.to_rgb8_srgb()
    seen = []

    def cms(profile, rgb):
        seen.append((profile, rgb.shape))
        return 255 - rgb

    assert np.array_equal(tagged.to_rgb8_srgb(cms), 255 - px) and seen == [(b"profile", (12, 3))]
    assert np.array_equal(plain.to_rgb8_srgb(cms), px) and len(seen) == 1  # untagged = sRGB: a plain copy (icc.rs:73)
    cube = ce.ColorTable.identity_cube()
    assert cube.shape == (1 << 24, 3) and cube.dtype == np.uint8
    assert cube[0].tolist() == [0, 0, 0] and cube[(7 << 16) | (9 << 8) | 11].tolist() == [7, 9, 11] and cube[-1].tolist() == [255, 255, 255]
