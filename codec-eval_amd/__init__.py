"""codec-eval_amd — MI355X (gfx950) backend for the perceptual-metric hot path of imazen/codec-eval.

The product is ``libce_metrics_hip.so`` (hand-written HIP kernels behind the C ABI declared in
``include/ce_metrics.h``).  This module is the thin ctypes binding used by the tests, bench.py and
Python callers; every metric call goes through the C ABI and there is NO CPU fallback: if the shared
library is missing or no HIP device is visible the call raises.

Import name: the directory is ``codec-eval_amd`` (hyphen, as the repo layout prescribes); use
``import codec_eval_amd`` (the shim module at the repo root) or
``importlib.import_module("codec-eval_amd")``.
"""
from __future__ import annotations

import ctypes as C
import os
from dataclasses import dataclass
from typing import Iterable, List, Optional, Sequence

import numpy as np

# HIP maps streams onto GPU_MAX_HW_QUEUES hardware queues (4 by default) and a stream that shares a queue with a busy
# one waits for it.  A context uses two streams per batch plus an upload stream, and callers keep several contexts /
# batches in flight, so ask for 16 (read once, when the HIP runtime initialises: this only takes effect if the package
# is imported before the process's first HIP call; 32 is far slower on MI355X, so it is a setdefault, not a maximum).
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("CE_METRICS_LIB") or os.path.join(_HERE, "libce_metrics_hip.so")
INCLUDE_DIR = os.path.join(os.path.dirname(_HERE), "include")

# ---- enums of include/ce_metrics.h ---------------------------------------------------------
CE_OK, CE_ERR_DIM_MISMATCH, CE_ERR_BAD_LENGTH, CE_ERR_TOO_SMALL, CE_ERR_BACKEND, CE_ERR_INVALID_ARG = range(6)
METRIC_DSSIM, METRIC_SSIMULACRA2, METRIC_BUTTERAUGLI, METRIC_PSNR = 1, 2, 4, 8
FLAG_XYB_ROUNDTRIP = 1
PIXEL_RGB8, PIXEL_RGBA8, PIXEL_RGB16_10BIT, PIXEL_RGBA16_10BIT = 0, 1, 2, 3
DEFAULT_INTENSITY_TARGET = 80.0

_STATUS_NAMES = {
    CE_ERR_DIM_MISMATCH: "DimensionMismatch",
    CE_ERR_BAD_LENGTH: "MetricCalculation(invalid image size)",
    CE_ERR_TOO_SMALL: "MetricCalculation(image too small)",
    CE_ERR_BACKEND: "MetricCalculation(backend)",
    CE_ERR_INVALID_ARG: "InvalidArgument",
}


class CeScores(C.Structure):
    _fields_ = [
        ("dssim", C.c_double),
        ("ssimulacra2", C.c_double),
        ("butteraugli", C.c_double),
        ("psnr", C.c_double),
        ("valid", C.c_uint32),
        ("status", C.c_int32),
    ]


class CePairDesc(C.Structure):
    _fields_ = [
        ("reference", C.c_void_p),
        ("reference_len", C.c_size_t),
        ("test", C.c_void_p),
        ("test_len", C.c_size_t),
        ("width", C.c_uint32),
        ("height", C.c_uint32),
    ]


class CodecEvalError(RuntimeError):
    """Mirrors codec_eval::Error for this path (src/error.rs:31-49)."""

    def __init__(self, status: int, message: str = ""):
        self.status = status
        self.kind = _STATUS_NAMES.get(status, f"status {status}")
        super().__init__(f"{self.kind}: {message}" if message else self.kind)


class DimensionMismatch(CodecEvalError):
    pass


class MetricCalculation(CodecEvalError):
    pass


def _raise(status: int, message: str):
    if status == CE_ERR_DIM_MISMATCH:
        raise DimensionMismatch(status, message)
    if status in (CE_ERR_BAD_LENGTH, CE_ERR_TOO_SMALL, CE_ERR_BACKEND):
        raise MetricCalculation(status, message)
    raise CodecEvalError(status, message)


def _error_obj(status: int, message: str) -> CodecEvalError:
    try:
        _raise(status, message)
    except CodecEvalError as e:
        return e


_lib: Optional[C.CDLL] = None

# (name, restype, argtypes) — must list every function include/ce_metrics.h declares
_vp, _u8p, _sz, _u32, _f32, _i = C.c_void_p, C.c_void_p, C.c_size_t, C.c_uint32, C.c_float, C.c_int
_dp = C.POINTER(C.c_double)
_PROTOTYPES = [
    ("ce_version", C.c_char_p, []),
    ("ce_device_count", _i, []),
    ("ce_ctx_create", _i, [_i, C.POINTER(_vp)]),
    ("ce_ctx_create_on_stream", _i, [_i, _vp, C.POINTER(_vp)]),
    ("ce_ctx_destroy", None, [_vp]),
    ("ce_ctx_synchronize", _i, [_vp]),
    ("ce_ctx_stream", _vp, [_vp]),
    ("ce_last_error", C.c_char_p, [_vp]),
    ("ce_calculate_psnr", _i, [_vp, _u8p, _sz, _u8p, _sz, _sz, _sz, _dp]),
    ("ce_calculate_ssimulacra2", _i, [_vp, _u8p, _sz, _u8p, _sz, _sz, _sz, _dp]),
    ("ce_calculate_dssim", _i, [_vp, _u8p, _sz, _u8p, _sz, _sz, _sz, _dp]),
    ("ce_calculate_butteraugli", _i, [_vp, _u8p, _sz, _u8p, _sz, _sz, _sz, _f32, _dp]),
    ("ce_xyb_roundtrip", _i, [_vp, _u8p, _sz, _sz, _sz, _u8p]),
    ("ce_rgb8_to_dssim_image", _i, [_vp, _u8p, _sz, _sz, _sz, _vp]),
    ("ce_eval_pair", _i, [_vp, _u8p, _sz, _u8p, _sz, _u32, _u32, _u32, _u32, _f32, C.POINTER(CeScores)]),
    ("ce_eval_batch", _i, [_vp, _sz, C.POINTER(CePairDesc), _u32, _u32, _f32, C.POINTER(CeScores)]),
    ("ce_estimate_batch_bytes", _sz, [_u32, _u32, _u32, _u32, _u32]),
    ("ce_ctx_memory_info", _i, [_vp, C.POINTER(_sz), C.POINTER(_sz)]),
    ("ce_host_alloc", _i, [_vp, _sz, C.POINTER(_vp)]),
    ("ce_host_free", _i, [_vp, _vp]),
    ("ce_eval_batch_lut", _i, [_vp, _sz, C.POINTER(CePairDesc), C.POINTER(_vp), _u32, _u32, _f32, C.POINTER(CeScores)]),
    ("ce_batch_create", _i, [_vp, _u32, _u32, _u32, _u32, C.POINTER(_vp)]),
    ("ce_batch_destroy", None, [_vp]),
    ("ce_batch_set_reference", _i, [_vp, _u32, _u8p, _sz]),
    ("ce_batch_set_test", _i, [_vp, _u32, _u32, _u8p, _sz]),
    ("ce_batch_set_reference_fmt", _i, [_vp, _u32, _vp, _sz, _i]),
    ("ce_batch_set_test_fmt", _i, [_vp, _u32, _u32, _vp, _sz, _i]),
    ("ce_lut_create", _i, [_vp, _u8p, _sz, C.POINTER(_vp)]),
    ("ce_lut_destroy", None, [_vp]),
    ("ce_batch_set_reference_lut", _i, [_vp, _u32, _vp, _sz, _i, _vp]),
    ("ce_batch_set_test_lut", _i, [_vp, _u32, _u32, _vp, _sz, _i, _vp]),
    ("ce_batch_reference_slab", _vp, [_vp]),
    ("ce_batch_test_slab", _vp, [_vp]),
    ("ce_batch_bind_pair", _i, [_vp, _u32, _u32]),
    ("ce_batch_run", _i, [_vp, _u32, _u32, _u32, _f32, C.POINTER(CeScores)]),
    ("ce_batch_launch", _i, [_vp, _u32, _u32, _u32, _f32]),
    ("ce_batch_collect", _i, [_vp, _u32, C.POINTER(CeScores)]),
    ("ce_batch_butteraugli_pnorm3", _i, [_vp, _u32, _dp]),
    ("ce_ref_create", _i, [_vp, _u8p, _sz, _u32, _u32, _u32, C.POINTER(_vp)]),
    ("ce_ref_compare", _i, [_vp, _u8p, _sz, _u32, _f32, C.POINTER(CeScores)]),
    ("ce_ref_compare_many", _i, [_vp, C.POINTER(_u8p), C.POINTER(_sz), _u32, _u32, _f32, C.POINTER(CeScores)]),
    ("ce_ref_stats", _i, [_vp, C.POINTER(_u32 * 3)]),
    ("ce_ref_destroy", None, [_vp]),
    ("ce_prof_enable", _i, [_vp, _i]),
    ("ce_prof_filter", _i, [_vp, C.c_char_p]),
    ("ce_prof_reset", _i, [_vp]),
    ("ce_prof_count", _i, [_vp]),
    ("ce_prof_get", _i, [_vp, _i, C.POINTER(C.c_char_p), C.POINTER(C.c_uint64), _dp]),
    ("ce_timer_start", _i, [_vp]),
    ("ce_timer_stop", _i, [_vp, _dp]),
    ("ce_debug_ssim2_planes", _i, [_vp, _i, _i, _i, _vp, _sz, C.POINTER(_u32), C.POINTER(_u32)]),
    ("ce_debug_ssim2_limit_scales", _i, [_vp, _i]),
    ("ce_debug_ssim2_averages", _i, [_vp, _u32, _dp, C.POINTER(_i)]),
    ("ce_debug_ssim2_occupancy", _i, [_i]),
    ("ce_debug_cbrt_sweep", _i, [_vp, _u32, C.c_uint64, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
    ("ce_debug_div_sweep", _i, [_vp, C.c_uint64, C.c_uint64, C.POINTER(C.c_uint64)]),
    ("ce_debug_calibrate_traffic", _i, [_vp, _sz]),
]
ABI_SYMBOLS = [p[0] for p in _PROTOTYPES]


def lib() -> C.CDLL:
    """Load libce_metrics_hip.so (built by __graft_entry__.build() / codec-eval_amd/build.py)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                f"{LIB_PATH} is missing: build it first (python -c 'import __graft_entry__ as g; g.build()'). "
                "There is no CPU fallback."
            )
        _lib = C.CDLL(LIB_PATH)
        for name, restype, argtypes in _PROTOTYPES:
            fn = getattr(_lib, name)  # AttributeError here = ABI symbol missing
            fn.restype = restype
            fn.argtypes = argtypes
    return _lib


def estimate_batch_bytes(width: int, height: int, n_refs: int, n_pairs: int, config: "MetricConfig") -> int:
    """Upper estimate of the device bytes a Batch of this shape holds once `config`'s metrics have run."""
    return int(lib().ce_estimate_batch_bytes(width, height, n_refs, n_pairs, config.mask))


def version() -> str:
    return lib().ce_version().decode()


def device_count() -> int:
    return int(lib().ce_device_count())


def _buf(a) -> np.ndarray:
    """Borrowed view as a flat contiguous u8 array (copy only if the input is not already one)."""
    arr = np.asarray(a)
    if arr.dtype != np.uint8:
        raise TypeError("pixel buffers must be uint8")
    return np.ascontiguousarray(arr).reshape(-1)


def _pinned_block(address: int, nbytes: int):
    """One ce_host_alloc block as a ctypes array (buffer protocol: numpy keeps it alive through `.base`) that frees the block
    when the last array on it is gone."""
    def _free(self):
        a, self._ce_address = getattr(self, "_ce_address", 0), 0
        if a:
            lib().ce_host_free(None, a)
    cls = type("CePinnedBlock", (C.c_ubyte * nbytes,), {"__del__": _free})
    blk = cls.from_address(address)
    blk._ce_address = address
    return blk


class PairList:
    """The ce_pair_desc array of a grid, built once: (reference, test, width, height) tuples -> descriptors that borrow the
    callers' buffers (kept alive here).  Identical reference objects share one descriptor address, which is what makes
    ce_eval_batch upload a reference once for all its distorted images."""

    def __init__(self, pairs: Sequence[tuple]):
        self.n = len(pairs)
        self.descs = (CePairDesc * self.n)()
        keep = {}  # id(buffer) -> (flat view, address, length, the object): a reference shared by many pairs is looked at once

        def addr(a):
            e = keep.get(id(a))
            if e is None:
                v = _buf(a)
                e = keep[id(a)] = (v, v.__array_interface__["data"][0], v.size, a)
            return e
        for i, (ref, test, w, h) in enumerate(pairs):
            r, t, d = addr(ref), addr(test), self.descs[i]
            d.reference, d.reference_len, d.test, d.test_len, d.width, d.height = r[1], r[2], t[1], t[2], w, h
        self._keep = keep

    def __len__(self):
        return self.n


# ---- MetricConfig / MetricResult mirrors (src/metrics/mod.rs:46-149) -----------------------
@dataclass
class MetricConfig:
    dssim: bool = False
    ssimulacra2: bool = False
    butteraugli: bool = False
    psnr: bool = False
    xyb_roundtrip: bool = False

    @staticmethod
    def all() -> "MetricConfig":
        return MetricConfig(True, True, True, True, False)

    @staticmethod
    def fast() -> "MetricConfig":
        return MetricConfig(psnr=True)

    @staticmethod
    def perceptual() -> "MetricConfig":
        return MetricConfig(True, True, True, False, False)

    @staticmethod
    def perceptual_xyb() -> "MetricConfig":
        return MetricConfig(True, True, True, False, True)

    @staticmethod
    def ssimulacra2_only() -> "MetricConfig":
        return MetricConfig(ssimulacra2=True)

    def with_xyb_roundtrip(self) -> "MetricConfig":
        return MetricConfig(self.dssim, self.ssimulacra2, self.butteraugli, self.psnr, True)

    @property
    def mask(self) -> int:
        return (
            (METRIC_DSSIM if self.dssim else 0)
            | (METRIC_SSIMULACRA2 if self.ssimulacra2 else 0)
            | (METRIC_BUTTERAUGLI if self.butteraugli else 0)
            | (METRIC_PSNR if self.psnr else 0)
        )

    @property
    def flags(self) -> int:
        return FLAG_XYB_ROUNDTRIP if self.xyb_roundtrip else 0


PERCEPTION_LEVELS = ("Imperceptible", "Marginal", "Subtle", "Noticeable", "Degraded")


def perception_from_dssim(d: float) -> str:  # src/metrics/mod.rs:189-201
    for level, t in zip(PERCEPTION_LEVELS, (0.0003, 0.0007, 0.0015, 0.003)):
        if d < t:
            return level
    return "Degraded"


def perception_from_ssimulacra2(s: float) -> str:  # mod.rs:206-218
    for level, t in zip(PERCEPTION_LEVELS, (90.0, 80.0, 70.0, 50.0)):
        if s > t:
            return level
    return "Degraded"


def perception_from_butteraugli(b: float) -> str:  # mod.rs:223-235
    for level, t in zip(PERCEPTION_LEVELS, (1.0, 2.0, 3.0, 5.0)):
        if b < t:
            return level
    return "Degraded"


@dataclass
class MetricResult:
    dssim: Optional[float] = None
    ssimulacra2: Optional[float] = None
    butteraugli: Optional[float] = None
    psnr: Optional[float] = None

    def perception_level(self) -> Optional[str]:  # mod.rs:154-156 (DSSIM only)
        return None if self.dssim is None else perception_from_dssim(self.dssim)

    @staticmethod
    def from_c(s: CeScores) -> "MetricResult":
        return MetricResult(
            s.dssim if s.valid & METRIC_DSSIM else None,
            s.ssimulacra2 if s.valid & METRIC_SSIMULACRA2 else None,
            s.butteraugli if s.valid & METRIC_BUTTERAUGLI else None,
            s.psnr if s.valid & METRIC_PSNR else None,
        )


class Context:
    """One device + one stream (GpuSsim2::new, crates/codec-iter/src/gpu.rs:40)."""

    def __init__(self, device: int = 0, stream: int | None = None):
        self._h = C.c_void_p()
        L = lib()
        rc = L.ce_ctx_create_on_stream(device, stream, C.byref(self._h)) if stream else L.ce_ctx_create(device, C.byref(self._h))
        if rc != CE_OK:
            msg = (L.ce_last_error(None) or b"").decode()
            self._h = C.c_void_p()
            _raise(rc, msg)
        self.device = device

    # -- lifetime
    def close(self):
        if self._h:
            lib().ce_ctx_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def _err(self) -> str:
        return (lib().ce_last_error(self._h) or b"").decode()

    def _check(self, rc: int):
        if rc != CE_OK:
            _raise(rc, self._err())

    @property
    def stream(self) -> int:
        return int(lib().ce_ctx_stream(self._h) or 0)

    def synchronize(self):
        self._check(lib().ce_ctx_synchronize(self._h))

    def memory_info(self):
        """(free, total) device bytes."""
        free, total = C.c_size_t(), C.c_size_t()
        self._check(lib().ce_ctx_memory_info(self._h, C.byref(free), C.byref(total)))
        return free.value, total.value

    def debug_div_sweep(self, seed: int, count: int) -> int:
        """Mismatches between the shared-reciprocal division of the Malta pre-scaling and operator/ (must be 0)."""
        bad = C.c_uint64()
        self._check(lib().ce_debug_div_sweep(self._h, seed, count, C.byref(bad)))
        return bad.value

    def debug_calibrate_traffic(self, nbytes: int):
        """Run the known-byte-count calibration streams (profiles/make_traffic.py reads them from a PMC pass)."""
        self._check(lib().ce_debug_calibrate_traffic(self._h, nbytes))

    def debug_cbrt_sweep(self, first_bits: int, count: int):
        """(mismatches, fallbacks) of the fast vs reference cube root over f32 bit patterns."""
        mism, slow = C.c_uint64(), C.c_uint64()
        self._check(lib().ce_debug_cbrt_sweep(self._h, first_bits, count, C.byref(mism), C.byref(slow)))
        return mism.value, slow.value

    # -- leaf calls, same names / argument order as the reference
    def _leaf(self, fn, reference, test, width, height, *extra) -> float:
        r, t = _buf(reference), _buf(test)
        out = C.c_double()
        self._check(fn(self._h, r.ctypes.data, r.size, t.ctypes.data, t.size, width, height, *extra, C.byref(out)))
        return out.value

    def calculate_psnr(self, reference, test, width: int, height: int) -> float:
        """calculate_psnr, src/metrics/mod.rs:312 (the reference panics on bad lengths; this raises)."""
        return self._leaf(lib().ce_calculate_psnr, reference, test, width, height)

    def calculate_ssimulacra2(self, reference, test, width: int, height: int) -> float:
        """calculate_ssimulacra2, src/metrics/ssimulacra2.rs:59."""
        return self._leaf(lib().ce_calculate_ssimulacra2, reference, test, width, height)

    def calculate_dssim(self, reference, test, width: int, height: int) -> float:
        """rgb8_to_dssim_image x2 + calculate_dssim, src/metrics/dssim.rs:102,40."""
        return self._leaf(lib().ce_calculate_dssim, reference, test, width, height)

    def calculate_butteraugli(self, reference, test, width: int, height: int) -> float:
        """calculate_butteraugli, src/metrics/butteraugli.rs:45."""
        return self._leaf(lib().ce_calculate_butteraugli, reference, test, width, height, DEFAULT_INTENSITY_TARGET)

    def calculate_butteraugli_with_intensity(self, reference, test, width: int, height: int, intensity_target: float) -> float:
        """calculate_butteraugli_with_intensity, src/metrics/butteraugli.rs:99."""
        return self._leaf(lib().ce_calculate_butteraugli, reference, test, width, height, float(intensity_target))

    def xyb_roundtrip(self, rgb, width: int, height: int) -> np.ndarray:
        """xyb_roundtrip, src/metrics/xyb.rs:225."""
        r = _buf(rgb)
        out = np.empty(r.size, np.uint8)
        self._check(lib().ce_xyb_roundtrip(self._h, r.ctypes.data, r.size, width, height, out.ctypes.data))
        return out

    def rgb8_to_dssim_image(self, rgb, width: int, height: int) -> np.ndarray:
        """rgb8_to_dssim_image, src/metrics/dssim.rs:102 -> (h, w, 4) float32, a = 1.0."""
        r = _buf(rgb)
        out = np.empty((height, width, 4), np.float32)
        self._check(lib().ce_rgb8_to_dssim_image(self._h, r.ctypes.data, r.size, width, height, out.ctypes.data))
        return out

    # -- dispatcher
    def calculate_metrics(self, reference, test, width: int, height: int, config: MetricConfig,
                          intensity_target: float = DEFAULT_INTENSITY_TARGET) -> MetricResult:
        """EvalSession::calculate_metrics, src/eval/session.rs:437-497."""
        r, t = _buf(reference), _buf(test)
        s = CeScores()
        self._check(lib().ce_eval_pair(self._h, r.ctypes.data, r.size, t.ctypes.data, t.size, width, height,
                                       config.mask, config.flags, intensity_target, C.byref(s)))
        return MetricResult.from_c(s)

    def host_buffer(self, nbytes: int) -> np.ndarray:
        """`nbytes` of page-locked host memory as a flat uint8 array (ce_host_alloc; freed when the array and every view of
        it are gone): images placed here are uploaded by DMA straight from the buffer, overlapped with the kernels."""
        p = C.c_void_p()
        self._check(lib().ce_host_alloc(self._h, nbytes, C.byref(p)))
        return np.frombuffer(_pinned_block(p.value, nbytes), dtype=np.uint8)

    def eval_batch(self, pairs, config: MetricConfig,
                   intensity_target: float = DEFAULT_INTENSITY_TARGET, test_tables: Optional[Sequence] = None) -> List[CeScores]:
        """pairs: (reference, test, width, height) tuples, or a PairList built from them once (a caller that scores the same
        buffers again and again - a decoder writing into fixed page-locked images - then pays the descriptor marshalling
        once, as a compiled caller of ce_eval_batch does).  The (codec x quality) grid of session.rs:375-376.
        test_tables: optional per-pair ColorTable (or None) applied to the distorted image on the device."""
        pl = pairs if isinstance(pairs, PairList) else PairList(pairs)
        n = pl.n
        out = (CeScores * n)()
        if test_tables is not None:
            luts = (C.c_void_p * n)(*[(t._h if t is not None else None) for t in test_tables])
            self._check(lib().ce_eval_batch_lut(self._h, n, pl.descs, luts, config.mask, config.flags, intensity_target, out))
        else:
            self._check(lib().ce_eval_batch(self._h, n, pl.descs, config.mask, config.flags, intensity_target, out))
        return list(out)

    # -- measurement hooks
    def prof_enable(self, on=True, serial: bool = True):
        """Per-kernel HIP-event timing.  serial=True: one kernel at a time on the context's stream (solo
        times); serial=False: keep the batch's multi-stream schedule (times as rocprofv3 sees them)."""
        self._check(lib().ce_prof_enable(self._h, 0 if not on else (1 if serial else 2)))

    def prof_filter(self, substring: str = ""):
        """Only kernels whose name contains `substring` get events ("" = all, "=name" = exactly that kernel)."""
        self._check(lib().ce_prof_filter(self._h, substring.encode()))

    def prof_reset(self):
        self._check(lib().ce_prof_reset(self._h))

    def prof_stats(self) -> dict:
        L = lib()
        out = {}
        for i in range(L.ce_prof_count(self._h)):
            name, n, ms = C.c_char_p(), C.c_uint64(), C.c_double()
            self._check(L.ce_prof_get(self._h, i, C.byref(name), C.byref(n), C.byref(ms)))
            out[name.value.decode()] = (int(n.value), float(ms.value))
        return out

    def timer_start(self):
        self._check(lib().ce_timer_start(self._h))

    def timer_stop(self) -> float:
        ms = C.c_double()
        self._check(lib().ce_timer_stop(self._h, C.byref(ms)))
        return ms.value


class ColorTable:
    """ICC -> sRGB as a complete colour table on the device (ce_lut_*): `table[r, g, b] = transform(r, g, b)`, uint8,
    shape (256, 256, 256, 3) - the host CMS evaluated once on the identity colour cube (`identity_cube()`), which
    reproduces an 8-bit RGB -> 8-bit RGB transform such as the reference's (icc.rs:69-103) bit for bit."""

    def __init__(self, ctx: "Context", table):
        t = np.ascontiguousarray(np.asarray(table, dtype=np.uint8)).reshape(-1)
        self.ctx = ctx
        self._h = C.c_void_p()
        ctx._check(lib().ce_lut_create(ctx._h, t.ctypes.data, t.size, C.byref(self._h)))

    @staticmethod
    def identity_cube() -> np.ndarray:
        """All 2^24 colours as an (2^24, 3) uint8 array in table order: feed it to the CMS, pass the result to ColorTable."""
        v = np.arange(1 << 24, dtype=np.uint32)
        return np.stack([(v >> 16) & 255, (v >> 8) & 255, v & 255], axis=1).astype(np.uint8)

    def close(self):
        if self._h and self.ctx._h:
            lib().ce_lut_destroy(self._h)
        self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Batch:
    """HBM-resident grid of (reference, test) pairs of one shape (ce_batch_*)."""

    def __init__(self, ctx: Context, width: int, height: int, max_refs: int, max_pairs: int):
        self.ctx, self.width, self.height = ctx, width, height
        self.max_refs, self.max_pairs = max_refs, max_pairs
        self._h = C.c_void_p()
        ctx._check(lib().ce_batch_create(ctx._h, width, height, max_refs, max_pairs, C.byref(self._h)))

    def close(self):
        if self._h and self.ctx._h:
            lib().ce_batch_destroy(self._h)
        self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_reference(self, ref_index: int, rgb):
        r = _buf(rgb)
        self.ctx._check(lib().ce_batch_set_reference(self._h, ref_index, r.ctypes.data, r.size))

    def set_test(self, pair_index: int, ref_index: int, rgb):
        t = _buf(rgb)
        self.ctx._check(lib().ce_batch_set_test(self._h, pair_index, ref_index, t.ctypes.data, t.size))

    # decoded-image ingest: pixels as a decoder hands them over, converted to RGB8 on the device
    def set_reference_fmt(self, ref_index: int, pixels, fmt: int):
        a = np.ascontiguousarray(pixels)
        self.ctx._check(lib().ce_batch_set_reference_fmt(self._h, ref_index, a.ctypes.data, a.nbytes, fmt))

    def set_test_fmt(self, pair_index: int, ref_index: int, pixels, fmt: int):
        a = np.ascontiguousarray(pixels)
        self.ctx._check(lib().ce_batch_set_test_fmt(self._h, pair_index, ref_index, a.ctypes.data, a.nbytes, fmt))

    # ... and through a colour table (ICC -> sRGB on the device)
    def set_reference_lut(self, ref_index: int, pixels, fmt: int, table: Optional["ColorTable"]):
        a = np.ascontiguousarray(pixels)
        self.ctx._check(lib().ce_batch_set_reference_lut(self._h, ref_index, a.ctypes.data, a.nbytes, fmt, table._h if table else None))

    def set_test_lut(self, pair_index: int, ref_index: int, pixels, fmt: int, table: Optional["ColorTable"]):
        a = np.ascontiguousarray(pixels)
        self.ctx._check(lib().ce_batch_set_test_lut(self._h, pair_index, ref_index, a.ctypes.data, a.nbytes, fmt, table._h if table else None))

    def bind_pair(self, pair_index: int, ref_index: int):
        self.ctx._check(lib().ce_batch_bind_pair(self._h, pair_index, ref_index))

    @property
    def reference_slab(self) -> int:
        return int(lib().ce_batch_reference_slab(self._h))

    @property
    def test_slab(self) -> int:
        return int(lib().ce_batch_test_slab(self._h))

    def run(self, n_pairs: int, config: MetricConfig, intensity_target: float = DEFAULT_INTENSITY_TARGET) -> List[CeScores]:
        out = (CeScores * n_pairs)()
        self.ctx._check(lib().ce_batch_run(self._h, n_pairs, config.mask, config.flags, intensity_target, out))
        return list(out)

    def launch(self, n_pairs: int, config: MetricConfig, intensity_target: float = DEFAULT_INTENSITY_TARGET):
        self.ctx._check(lib().ce_batch_launch(self._h, n_pairs, config.mask, config.flags, intensity_target))

    def collect(self, n_pairs: int) -> List[CeScores]:
        out = (CeScores * n_pairs)()
        self.ctx._check(lib().ce_batch_collect(self._h, n_pairs, out))
        return list(out)

    def butteraugli_pnorm3(self, n_pairs: int) -> np.ndarray:
        out = np.zeros(n_pairs, np.float64)
        self.ctx._check(lib().ce_batch_butteraugli_pnorm3(self._h, n_pairs, out.ctypes.data_as(_dp)))
        return out

    # -- test hooks
    def debug_limit_scales(self, n: int):
        self.ctx._check(lib().ce_debug_ssim2_limit_scales(self._h, n))

    def debug_planes(self, scale: int, which: int, channel: int = 0) -> np.ndarray:
        nplanes = 5 if which == 4 else 3
        buf = np.empty(nplanes * self.width * self.height, np.float32)
        w, h = C.c_uint32(), C.c_uint32()
        self.ctx._check(lib().ce_debug_ssim2_planes(self._h, scale, which, channel, buf.ctypes.data, buf.size,
                                                     C.byref(w), C.byref(h)))
        return buf[: nplanes * w.value * h.value].reshape(nplanes, h.value, w.value).copy()

    def debug_averages(self, pair_index: int) -> np.ndarray:
        avg = np.zeros((6, 3, 6), np.float64)
        ns = C.c_int()
        self.ctx._check(lib().ce_debug_ssim2_averages(self._h, pair_index, avg.ctypes.data_as(_dp), C.byref(ns)))
        return avg[: ns.value].copy()


class ReferenceHandle:
    """Ssimulacra2Reference::{new, compare} (crates/codec-iter/src/eval.rs:138-149, 83-89)."""

    def __init__(self, ctx: Context, reference, width: int, height: int, xyb_roundtrip: bool = False):
        self.ctx = ctx
        r = _buf(reference)
        self._h = C.c_void_p()
        ctx._check(lib().ce_ref_create(ctx._h, r.ctypes.data, r.size, width, height,
                                       FLAG_XYB_ROUNDTRIP if xyb_roundtrip else 0, C.byref(self._h)))

    def compare(self, test, config: MetricConfig = None, intensity_target: float = DEFAULT_INTENSITY_TARGET) -> MetricResult:
        config = config or MetricConfig.ssimulacra2_only()
        t = _buf(test)
        s = CeScores()
        self.ctx._check(lib().ce_ref_compare(self._h, t.ctypes.data, t.size, config.mask, intensity_target, C.byref(s)))
        return MetricResult.from_c(s)

    def compare_many(self, tests, config: MetricConfig = None, intensity_target: float = DEFAULT_INTENSITY_TARGET):
        """The quality sweep of this reference in one launch; returns one MetricResult (or error) per test."""
        config = config or MetricConfig.ssimulacra2_only()
        bufs = [_buf(t) for t in tests]
        n = len(bufs)
        if n == 0:
            return []
        ptrs = (C.c_void_p * n)(*[b.ctypes.data for b in bufs])
        lens = (_sz * n)(*[b.size for b in bufs])
        out = (CeScores * n)()
        self.ctx._check(lib().ce_ref_compare_many(self._h, ptrs, lens, n, config.mask, intensity_target, out))
        res = []
        for s in out:
            res.append(MetricResult.from_c(s) if s.status == 0 else _error_obj(s.status, self.ctx._err()))
        return res

    def stats(self):
        """(ssimulacra2, dssim, butteraugli): compares so far that had to build that metric's reference-side state."""
        out = (_u32 * 3)()
        self.ctx._check(lib().ce_ref_stats(self._h, C.byref(out)))
        return tuple(out)

    def close(self):
        if self._h and self.ctx._h:
            lib().ce_ref_destroy(self._h)
        self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


# ---- codec-iter's plug point (crates/codec-iter/src/eval.rs:56-92, gpu.rs:21-134) ------------
Ssimulacra2Reference = ReferenceHandle  # fast_ssim2::Ssimulacra2Reference::{new, compare}


class GpuSsim2:
    """GpuSsim2::new(w, h) / compute(&mut self, reference, distorted) (gpu.rs:40-116): fixed shape, packed
    RGB8 from host memory, one call in flight per object."""

    def __init__(self, width: int, height: int, device: int = 0):
        self.ctx = Context(device)
        self.width, self.height = int(width), int(height)

    def compute(self, reference, distorted) -> float:
        r, d = _buf(reference), _buf(distorted)
        expected = self.width * self.height * 3
        if r.size != expected or d.size != expected:  # gpu.rs:84-94
            raise RuntimeError(f"Image size mismatch: expected {expected} bytes ({self.width}x{self.height}x3), "
                               f"got ref={r.size} dis={d.size}")
        return self.ctx.calculate_ssimulacra2(r, d, self.width, self.height)

    def dimensions(self):
        return (self.width, self.height)

    def close(self):
        self.ctx.close()


class Ssim2Backend:
    """eval.rs:56-92.  The reference's enum has a Gpu and a Cpu arm; this package is the device arm and has
    no CPU path, so only that arm exists here."""

    def __init__(self, gpu: GpuSsim2):
        self.gpu = gpu

    def compare_with_precomputed(self, source, decoded, reference: Optional[ReferenceHandle], image_name: str, quality: int) -> float:
        try:
            if reference is not None:
                return reference.compare(decoded).ssimulacra2
            return self.gpu.compute(source, decoded)
        except (CodecEvalError, RuntimeError) as e:  # eval.rs:88
            raise RuntimeError(f"SSIM2 error for {image_name} q{quality}: {e}") from e


# ---- eval helpers (src/eval/helpers.rs) ----------------------------------------------------
class QualityBelowThreshold(CodecEvalError):
    def __init__(self, metric: str, value: float, threshold: float):
        RuntimeError.__init__(self, f"{metric} quality below threshold: {value} (threshold: {threshold})")
        self.status, self.kind = -1, "QualityBelowThreshold"
        self.metric, self.value, self.threshold = metric, value, threshold


def _as_rgb8(img) -> np.ndarray:
    a = np.asarray(img)
    if a.dtype != np.uint8 or a.ndim != 3 or a.shape[2] != 3:
        raise TypeError("expected an (h, w, 3) uint8 image")
    return a


def evaluate_single(ctx: Context, reference, encoded, config: MetricConfig) -> MetricResult:
    """evaluate_single, src/eval/helpers.rs:105-173: (h, w, 3) uint8 images."""
    r, e = _as_rgb8(reference), _as_rgb8(encoded)
    if r.shape != e.shape:  # helpers.rs:111-116
        raise DimensionMismatch(CE_ERR_DIM_MISMATCH, f"expected {(r.shape[1], r.shape[0])}, got {(e.shape[1], e.shape[0])}")
    return ctx.calculate_metrics(r, e, r.shape[1], r.shape[0], config)


def assert_quality(ctx: Context, reference, encoded, min_ssimulacra2: Optional[float], max_dssim: Optional[float]) -> None:
    """assert_quality, src/eval/helpers.rs:212-255."""
    cfg = MetricConfig(dssim=max_dssim is not None, ssimulacra2=min_ssimulacra2 is not None)
    res = evaluate_single(ctx, reference, encoded, cfg)
    if min_ssimulacra2 is not None and res.ssimulacra2 is not None and res.ssimulacra2 < min_ssimulacra2:
        raise QualityBelowThreshold("SSIMULACRA2", res.ssimulacra2, min_ssimulacra2)
    if max_dssim is not None and res.dssim is not None and res.dssim > max_dssim:
        raise QualityBelowThreshold("DSSIM", res.dssim, max_dssim)


def assert_perception_level(ctx: Context, reference, encoded, min_level: str) -> None:
    """assert_perception_level, src/eval/helpers.rs:291-321 (DSSIM only, ordinal compare)."""
    res = evaluate_single(ctx, reference, encoded, MetricConfig(dssim=True))
    if res.dssim is not None:
        actual = PERCEPTION_LEVELS.index(perception_from_dssim(res.dssim))
        want = PERCEPTION_LEVELS.index(min_level)
        if actual > want:
            raise QualityBelowThreshold(f"PerceptionLevel (DSSIM {res.dssim:.6f})", float(actual), float(want))
