"""ctypes front-end of the CPU oracle (oracle/libce_oracle.so).

TEST INFRASTRUCTURE ONLY.  Importers allowed: tests/, bench.py's cpu_baseline leg,
__graft_entry__.smoke().  The product package (codec-eval_amd/) never imports this.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libce_oracle.so")

OK, DIM_MISMATCH, BAD_LENGTH, TOO_SMALL, BACKEND = 0, 1, 2, 3, 4


def build(force: bool = False) -> str:
    """Compile the oracle with gcc (seconds)."""
    srcs = [os.path.join(_HERE, f) for f in os.listdir(_HERE) if f.endswith((".c", ".h"))]
    stale = (not os.path.exists(_SO)) or any(os.path.getmtime(s) > os.path.getmtime(_SO) for s in srcs)
    if force or stale:
        subprocess.check_call(["make", "-C", _HERE, "-B", "libce_oracle.so"], stdout=subprocess.DEVNULL)
    return _SO


_lib = None


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(_SO)
        u8p, f32p, f64p = C.POINTER(C.c_uint8), C.POINTER(C.c_float), C.POINTER(C.c_double)
        sz = C.c_size_t
        _lib.ceo_psnr.argtypes = [u8p, sz, u8p, sz, sz, sz, f64p]
        _lib.ceo_sse_u8.argtypes = [u8p, u8p, sz]
        _lib.ceo_sse_u8.restype = C.c_uint64
        _lib.ceo_srgb_u8_to_linear.argtypes = [C.c_uint8]
        _lib.ceo_srgb_u8_to_linear.restype = C.c_float
        _lib.ceo_rgb8_to_dssim_image.argtypes = [u8p, sz, f32p]
        _lib.ceo_rgb8_to_dssim_image.restype = None
        _lib.ceo_xyb_roundtrip.argtypes = [u8p, sz, sz, sz, u8p]
        _lib.ceo_set_variant.argtypes = [C.c_int, C.c_int]
        _lib.ceo_set_variant.restype = None
        _lib.ceo_get_variant.argtypes = [C.c_int]
        _lib.ceo_cbrtf_compare.argtypes = [f32p, sz, f32p, f32p]
        _lib.ceo_cbrtf_compare.restype = None
        _lib.ceo_ssimulacra2.argtypes = [u8p, sz, u8p, sz, sz, sz, C.c_int, f64p]
        _lib.ceo_ssimulacra2_detail.argtypes = [u8p, u8p, sz, sz, C.c_int, f64p, C.POINTER(C.c_int), f64p]
        _lib.ceo_ssimulacra2_score.argtypes = [f64p, C.c_int]
        _lib.ceo_ssimulacra2_score.restype = C.c_double
        _lib.ceo_ssim2_srgb_lut.argtypes = [f32p]
        _lib.ceo_ssim2_srgb_lut.restype = None
        _lib.ceo_ssim2_blur_taps.argtypes = [f32p, f64p]
        _lib.ceo_ssim2_blur_taps.restype = None
        _lib.ceo_ssim2_linear_planar.argtypes = [u8p, sz, f32p]
        _lib.ceo_ssim2_linear_planar.restype = None
        _lib.ceo_ssim2_downscale.argtypes = [f32p, sz, sz, f32p]
        _lib.ceo_ssim2_downscale.restype = None
        _lib.ceo_ssim2_xyb_positive.argtypes = [f32p, sz, f32p]
        _lib.ceo_ssim2_xyb_positive.restype = None
        _lib.ceo_ssim2_blur_plane.argtypes = [f32p, sz, sz, C.c_int, f32p]
        _lib.ceo_ssim2_blur_plane.restype = None
        _lib.ceo_dssim_rgb8.argtypes = [u8p, sz, u8p, sz, sz, sz, f64p]
        _lib.ceo_dssim_rgbaf.argtypes = [f32p, sz, sz, f32p, sz, sz, f64p]
        _lib.ceo_dssim_detail.argtypes = [u8p, u8p, sz, sz, f64p, C.POINTER(C.c_int), f64p]
        _lib.ceo_butteraugli.argtypes = [u8p, sz, u8p, sz, sz, sz, C.c_float, f64p, f64p]
    return _lib


def _u8(a) -> np.ndarray:
    return np.ascontiguousarray(np.asarray(a, dtype=np.uint8).reshape(-1))


def _p(a: np.ndarray, ty):
    return a.ctypes.data_as(C.POINTER(ty))


class OracleError(Exception):
    def __init__(self, code: int):
        super().__init__(f"oracle status {code}")
        self.code = code


def _chk(rc: int):
    if rc != OK:
        raise OracleError(rc)


def psnr(ref, test, w: int, h: int) -> float:
    r, t = _u8(ref), _u8(test)
    out = C.c_double()
    _chk(lib().ceo_psnr(_p(r, C.c_uint8), r.size, _p(t, C.c_uint8), t.size, w, h, C.byref(out)))
    return out.value


def sse(ref, test) -> int:
    r, t = _u8(ref), _u8(test)
    assert r.size == t.size
    return int(lib().ceo_sse_u8(_p(r, C.c_uint8), _p(t, C.c_uint8), r.size))


def srgb_u8_to_linear(v: int) -> float:
    return float(lib().ceo_srgb_u8_to_linear(int(v)))


VARIANTS = {"ssim2_srgb_f32_powf": 0, "ssim2_host_cbrtf": 1, "ssim2_iir_no_fma": 2, "dssim_lab_no_fma": 3,
            "dssim_f32_final": 4, "ba_malta_f32": 5, "ba_libm_log2": 6, "ssim2_f32_pool": 7, "ba_l2_early": 8, "ba_blur_fma": 9}


def set_variant(name: str, value: int):
    """Sensitivity switch of ce_oracle.h (tests/golden/sensitivity.py only; 0 = the restatement)."""
    lib().ceo_set_variant(VARIANTS[name], int(value))


def variants_all_default() -> bool:
    return all(lib().ceo_get_variant(k) == 0 for k in VARIANTS.values())


def cbrtf_compare(x) -> tuple:
    """(pinned, host): the oracle's pinned cbrtf (glibc 2.35 restated) and the host libm's on the same inputs."""
    a = np.ascontiguousarray(np.asarray(x, dtype=np.float32).reshape(-1))
    pinned, host = np.empty_like(a), np.empty_like(a)
    lib().ceo_cbrtf_compare(_p(a, C.c_float), a.size, _p(pinned, C.c_float), _p(host, C.c_float))
    return pinned, host


def rgb8_to_dssim_image(rgb, w: int, h: int) -> np.ndarray:
    r = _u8(rgb)
    out = np.empty((h, w, 4), np.float32)
    lib().ceo_rgb8_to_dssim_image(_p(r, C.c_uint8), w * h, _p(out, C.c_float))
    return out


def xyb_roundtrip(rgb, w: int, h: int) -> np.ndarray:
    r = _u8(rgb)
    out = np.empty(r.size, np.uint8)
    _chk(lib().ceo_xyb_roundtrip(_p(r, C.c_uint8), r.size, w, h, _p(out, C.c_uint8)))
    return out


def ssimulacra2(ref, test, w: int, h: int, blur_mode: int = 0) -> float:
    r, t = _u8(ref), _u8(test)
    out = C.c_double()
    _chk(lib().ceo_ssimulacra2(_p(r, C.c_uint8), r.size, _p(t, C.c_uint8), t.size, w, h, blur_mode, C.byref(out)))
    return out.value


def ssimulacra2_detail(ref, test, w: int, h: int, blur_mode: int = 0):
    """-> (score, avg[n_scales,3,6])"""
    r, t = _u8(ref), _u8(test)
    avg = np.zeros((6, 3, 6), np.float64)
    ns, out = C.c_int(), C.c_double()
    _chk(lib().ceo_ssimulacra2_detail(_p(r, C.c_uint8), _p(t, C.c_uint8), w, h, blur_mode,
                                      _p(avg, C.c_double), C.byref(ns), C.byref(out)))
    return out.value, avg[: ns.value].copy()


def ssimulacra2_score(avg: np.ndarray) -> float:
    a = np.ascontiguousarray(avg, dtype=np.float64)
    return float(lib().ceo_ssimulacra2_score(_p(a, C.c_double), a.shape[0]))


def ssim2_srgb_lut() -> np.ndarray:
    out = np.empty(256, np.float32)
    lib().ceo_ssim2_srgb_lut(_p(out, C.c_float))
    return out


def ssim2_blur_taps():
    t32, t64 = np.empty(5, np.float32), np.empty(5, np.float64)
    lib().ceo_ssim2_blur_taps(_p(t32, C.c_float), _p(t64, C.c_double))
    return t32, t64


def ssim2_linear_planar(rgb, w: int, h: int) -> np.ndarray:
    r = _u8(rgb)
    out = np.empty((3, h, w), np.float32)
    lib().ceo_ssim2_linear_planar(_p(r, C.c_uint8), w * h, _p(out, C.c_float))
    return out


def ssim2_downscale(planes: np.ndarray) -> np.ndarray:
    p = np.ascontiguousarray(planes, np.float32)
    _, h, w = p.shape
    out = np.empty((3, (h + 1) // 2, (w + 1) // 2), np.float32)
    lib().ceo_ssim2_downscale(_p(p, C.c_float), w, h, _p(out, C.c_float))
    return out


def ssim2_xyb_positive(planes: np.ndarray) -> np.ndarray:
    p = np.ascontiguousarray(planes, np.float32)
    _, h, w = p.shape
    out = np.empty_like(p)
    lib().ceo_ssim2_xyb_positive(_p(p, C.c_float), w * h, _p(out, C.c_float))
    return out


def ssim2_blur_plane(plane: np.ndarray, blur_mode: int = 0) -> np.ndarray:
    p = np.ascontiguousarray(plane, np.float32)
    h, w = p.shape
    out = np.empty_like(p)
    lib().ceo_ssim2_blur_plane(_p(p, C.c_float), w, h, blur_mode, _p(out, C.c_float))
    return out


def dssim(ref, test, w: int, h: int) -> float:
    r, t = _u8(ref), _u8(test)
    out = C.c_double()
    _chk(lib().ceo_dssim_rgb8(_p(r, C.c_uint8), r.size, _p(t, C.c_uint8), t.size, w, h, C.byref(out)))
    return out.value


def dssim_rgbaf(ref_rgba: np.ndarray, test_rgba: np.ndarray) -> float:
    a = np.ascontiguousarray(ref_rgba, np.float32)
    b = np.ascontiguousarray(test_rgba, np.float32)
    out = C.c_double()
    _chk(lib().ceo_dssim_rgbaf(_p(a, C.c_float), a.shape[1], a.shape[0], _p(b, C.c_float), b.shape[1], b.shape[0],
                               C.byref(out)))
    return out.value


def dssim_detail(ref, test, w: int, h: int):
    r, t = _u8(ref), _u8(test)
    sc = np.zeros(5, np.float64)
    ns, out = C.c_int(), C.c_double()
    _chk(lib().ceo_dssim_detail(_p(r, C.c_uint8), _p(t, C.c_uint8), w, h, _p(sc, C.c_double), C.byref(ns), C.byref(out)))
    return out.value, sc[: ns.value].copy()


def butteraugli(ref, test, w: int, h: int, intensity_target: float = 80.0):
    """-> (score (max-norm), 3-norm)"""
    r, t = _u8(ref), _u8(test)
    s, p3 = C.c_double(), C.c_double()
    _chk(lib().ceo_butteraugli(_p(r, C.c_uint8), r.size, _p(t, C.c_uint8), t.size, w, h, intensity_target,
                               C.byref(s), C.byref(p3)))
    return s.value, p3.value
