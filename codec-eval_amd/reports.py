"""Result wire formats of the reference, byte for byte.

What downstream tools (`pareto`, `stats`, `rd_knee`, codec-iter's baseline compare) read:

* `ImageReport` / `CorpusReport` JSON  — `serde_json::to_string_pretty` of the structs in
  src/eval/report.rs:14-166 (written by src/eval/session.rs:500-523);
* the 13-column CSV summary       — src/eval/session.rs:526-584 (`csv` crate, default writer);
* `Baseline` / `EvalPoint` JSON       — crates/codec-iter/src/baseline.rs:11-47, eval.rs:21-29.

Everything here is host-side formatting of scores the device path produced; there is no arithmetic
on pixels.  serde_json prints f64 with ryu (shortest digits that round-trip, its own exponent
rules), non-finite floats as `null`, struct fields in declaration order, 2-space indent.
`tests/test_reports.py` round-trips the reference's own `baselines/jpeg.json` byte for byte.
"""
from __future__ import annotations

import datetime as _dt
import io
import json
import os
from dataclasses import dataclass, field
from typing import Dict, List, Optional

__all__ = [
    "format_f64", "to_string_pretty", "rust_f64_display", "CodecResult", "ImageReport", "CorpusReport",
    "EvalPoint", "Baseline", "write_image_report", "write_corpus_report", "csv_summary", "save_baseline",
    "load_baseline", "rfc3339", "chrono_utc_default",
]


# ---- numbers ---------------------------------------------------------------------------------
def _digits_exp(x: float):
    """Shortest round-trip decimal digits d1..dn and exponent k with x = 0.d1..dn * 10^kk, kk = n + k."""
    r = repr(abs(x))  # Python's repr is also the shortest round-trip digit string
    if "e" in r:
        mant, e = r.split("e")
        e = int(e)
    else:
        mant, e = r, 0
    if "." in mant:
        ip, fp = mant.split(".")
    else:
        ip, fp = mant, ""
    digits = (ip + fp).lstrip("0")
    # value = int(ip+fp) * 10^(e - len(fp))
    k = e - len(fp)
    stripped = digits.rstrip("0")
    k += len(digits) - len(stripped)
    digits = stripped or "0"
    return digits, k


def format_f64(x: float) -> str:
    """ryu's `format64` as serde_json uses it (`ryu::Buffer::format_finite`), e.g. 0.72332763671875,
    80.0, 1e-7, 1.5e300, 0.00001."""
    if x != x or x in (float("inf"), float("-inf")):
        return "null"  # serde_json: non-finite f64 serialises as null
    if x == 0.0:
        return "-0.0" if str(x).startswith("-") else "0.0"
    sign = "-" if x < 0 else ""
    digits, k = _digits_exp(x)
    n = len(digits)
    kk = n + k
    if 0 <= k and kk <= 16:
        return f"{sign}{digits}{'0' * k}.0"
    if 0 < kk <= 16:
        return f"{sign}{digits[:kk]}.{digits[kk:]}"
    if -5 < kk <= 0:
        return f"{sign}0.{'0' * (-kk)}{digits}"
    e = kk - 1
    if n == 1:
        return f"{sign}{digits}e{e}"
    return f"{sign}{digits[0]}.{digits[1:]}e{e}"


def rust_f64_display(x: float) -> str:
    """`f64::to_string()` / `{}`: shortest round-trip digits, never an exponent, no trailing `.0`."""
    if x != x:
        return "NaN"
    if x in (float("inf"), float("-inf")):
        return "inf" if x > 0 else "-inf"
    if x == 0.0:
        return "-0" if str(x).startswith("-") else "0"
    sign = "-" if x < 0 else ""
    digits, k = _digits_exp(x)
    n = len(digits)
    kk = n + k
    if k >= 0:
        return f"{sign}{digits}{'0' * k}"
    if kk > 0:
        return f"{sign}{digits[:kk]}.{digits[kk:]}"
    return f"{sign}0.{'0' * (-kk)}{digits}"


# ---- serde_json::to_string_pretty ----------------------------------------------------------------
_ESC = {'"': '\\"', "\\": "\\\\", "\b": "\\b", "\f": "\\f", "\n": "\\n", "\r": "\\r", "\t": "\\t"}


def _json_str(s: str) -> str:
    out = ['"']
    for ch in s:
        if ch in _ESC:
            out.append(_ESC[ch])
        elif ord(ch) < 0x20:
            out.append("\\u%04x" % ord(ch))
        else:
            out.append(ch)  # serde_json leaves non-ASCII as UTF-8
    out.append('"')
    return "".join(out)


def _emit(v, ind: int, out: List[str]):
    pad = "  " * (ind + 1)
    if v is None:
        out.append("null")
    elif v is True:
        out.append("true")
    elif v is False:
        out.append("false")
    elif isinstance(v, int):
        out.append(str(v))
    elif isinstance(v, float):
        out.append(format_f64(v))
    elif isinstance(v, str):
        out.append(_json_str(v))
    elif isinstance(v, dict):
        if not v:
            out.append("{}")
            return
        out.append("{\n")
        items = list(v.items())
        for i, (k, x) in enumerate(items):
            out.append(pad + _json_str(str(k)) + ": ")
            _emit(x, ind + 1, out)
            out.append(",\n" if i + 1 < len(items) else "\n")
        out.append("  " * ind + "}")
    elif isinstance(v, (list, tuple)):
        if not v:
            out.append("[]")
            return
        out.append("[\n")
        for i, x in enumerate(v):
            out.append(pad)
            _emit(x, ind + 1, out)
            out.append(",\n" if i + 1 < len(v) else "\n")
        out.append("  " * ind + "]")
    else:
        raise TypeError(f"cannot serialise {type(v).__name__}")


def to_string_pretty(value) -> str:
    out: List[str] = []
    _emit(value, 0, out)
    return "".join(out)


# ---- timestamps ------------------------------------------------------------------------------------
def _auto_si(nanos: int) -> str:
    """chrono's SecondsFormat::AutoSi: no fraction, or 3, 6 or 9 digits."""
    if nanos == 0:
        return ""
    if nanos % 1_000_000 == 0:
        return ".%03d" % (nanos // 1_000_000)
    if nanos % 1_000 == 0:
        return ".%06d" % (nanos // 1_000)
    return ".%09d" % nanos


def rfc3339(t: _dt.datetime, nanos: Optional[int] = None) -> str:
    """`DateTime<Utc>::to_rfc3339()` (report.rs `chrono_serde`): `+00:00` offset, AutoSi fraction."""
    t = t.astimezone(_dt.timezone.utc)
    ns = t.microsecond * 1000 if nanos is None else nanos
    return t.strftime("%Y-%m-%dT%H:%M:%S") + _auto_si(ns) + "+00:00"


def chrono_utc_default(t: _dt.datetime, nanos: Optional[int] = None) -> str:
    """chrono's own serde impl for `DateTime<Utc>` (baseline.rs `created_at`): `Z` suffix, AutoSi fraction."""
    t = t.astimezone(_dt.timezone.utc)
    ns = t.microsecond * 1000 if nanos is None else nanos
    return t.strftime("%Y-%m-%dT%H:%M:%S") + _auto_si(ns) + "Z"


def _now() -> _dt.datetime:
    return _dt.datetime.now(_dt.timezone.utc)


# ---- report structs (src/eval/report.rs) -------------------------------------------------------------
_PERCEPTION_NAMES = {"IMP": "Imperceptible", "MAR": "Marginal", "SUB": "Subtle", "NOT": "Noticeable", "DEG": "Degraded"}
_PERCEPTION_CODES = {v: k for k, v in _PERCEPTION_NAMES.items()}


def _perception_name(p) -> Optional[str]:
    if p is None:
        return None
    p = str(p)
    return _PERCEPTION_NAMES.get(p, p)  # accepts "IMP" or "Imperceptible"


@dataclass
class CodecResult:  # report.rs:14-52
    codec_id: str
    codec_version: str
    quality: float
    file_size: int
    bits_per_pixel: float
    encode_time_ms: int
    decode_time_ms: Optional[int] = None
    dssim: Optional[float] = None
    ssimulacra2: Optional[float] = None
    butteraugli: Optional[float] = None
    psnr: Optional[float] = None
    perception: Optional[str] = None  # "Imperceptible" ... or its 3-letter code
    cached_path: Optional[str] = None
    codec_params: Dict[str, str] = field(default_factory=dict)

    def compression_ratio(self, original_size: int) -> float:  # report.rs:54-63
        return 0.0 if self.file_size == 0 else original_size / self.file_size

    def to_obj(self) -> dict:
        f = lambda v: None if v is None else float(v)
        return {
            "codec_id": self.codec_id,
            "codec_version": self.codec_version,
            "quality": float(self.quality),
            "file_size": int(self.file_size),
            "bits_per_pixel": float(self.bits_per_pixel),
            "encode_time": int(self.encode_time_ms),
            "decode_time": None if self.decode_time_ms is None else int(self.decode_time_ms),
            "metrics": {"dssim": f(self.dssim), "ssimulacra2": f(self.ssimulacra2), "butteraugli": f(self.butteraugli),
                        "psnr": f(self.psnr)},
            "perception": _perception_name(self.perception),
            "cached_path": self.cached_path,
            # a Rust HashMap has no stable order; sorted here so that output is reproducible
            "codec_params": dict(sorted(self.codec_params.items())),
        }


@dataclass
class ImageReport:  # report.rs:66-107
    name: str
    width: int
    height: int
    results: List[CodecResult] = field(default_factory=list)
    source_path: Optional[str] = None
    timestamp: _dt.datetime = field(default_factory=_now)

    @property
    def uncompressed_size(self) -> int:
        return self.width * self.height * 3

    def to_obj(self) -> dict:
        return {
            "name": self.name,
            "source_path": self.source_path,
            "width": int(self.width),
            "height": int(self.height),
            "uncompressed_size": self.uncompressed_size,
            "results": [r.to_obj() for r in self.results],
            "timestamp": rfc3339(self.timestamp),
        }

    def results_for_codec(self, codec_id: str):
        return [r for r in self.results if r.codec_id == codec_id]

    def best_at_size(self, max_bytes: int) -> Optional[CodecResult]:  # report.rs:109-124 (last maximum wins)
        best, best_q = None, None
        for r in self.results:
            if r.file_size > max_bytes:
                continue
            q = -r.dssim if r.dssim is not None else float("-inf")
            if best is None or q >= best_q:
                best, best_q = r, q
        return best

    def smallest_at_quality(self, max_dssim: float) -> Optional[CodecResult]:  # report.rs:126-134 (first minimum wins)
        ok = [r for r in self.results if r.dssim is not None and r.dssim <= max_dssim]
        return min(ok, key=lambda r: r.file_size) if ok else None


@dataclass
class CorpusReport:  # report.rs:137-186
    name: str
    images: List[ImageReport] = field(default_factory=list)
    timestamp: _dt.datetime = field(default_factory=_now)
    config_summary: str = ""

    def to_obj(self) -> dict:
        return {"name": self.name, "images": [i.to_obj() for i in self.images], "timestamp": rfc3339(self.timestamp),
                "config_summary": self.config_summary}

    def total_results(self) -> int:
        return sum(len(i.results) for i in self.images)

    def codec_ids(self) -> List[str]:
        return sorted({r.codec_id for i in self.images for r in i.results})


# ---- CSV (session.rs:526-584; csv crate defaults: quote when necessary, "\n" terminator) --------------------
CSV_HEADER = ["image", "codec", "version", "quality", "file_size", "bpp", "encode_ms", "decode_ms", "dssim",
              "ssimulacra2", "butteraugli", "psnr", "perception"]


def _csv_field(s: str) -> str:
    if s == "" or not any(c in s for c in ',"\n\r'):
        return s
    return '"' + s.replace('"', '""') + '"'


def _fixed(v: Optional[float], places: int) -> str:
    if v is None:
        return ""
    if v != v:
        return "NaN"
    if v in (float("inf"), float("-inf")):
        return "inf" if v > 0 else "-inf"
    return format(v, f".{places}f")  # Rust's {:.N} and Python's both round the exact binary value correctly


def csv_summary(report: CorpusReport) -> str:
    out = io.StringIO()
    out.write(",".join(CSV_HEADER) + "\n")
    for img in report.images:
        for r in img.results:
            code = None if r.perception is None else _PERCEPTION_CODES.get(_perception_name(r.perception), str(r.perception))
            row = [img.name, r.codec_id, r.codec_version, rust_f64_display(float(r.quality)), str(int(r.file_size)),
                   _fixed(r.bits_per_pixel, 4), str(int(r.encode_time_ms)),
                   "" if r.decode_time_ms is None else str(int(r.decode_time_ms)), _fixed(r.dssim, 6), _fixed(r.ssimulacra2, 2),
                   _fixed(r.butteraugli, 4), _fixed(r.psnr, 2), "" if code is None else code]
            out.write(",".join(_csv_field(x) for x in row) + "\n")
    return out.getvalue()


def write_image_report(report_dir: str, report: ImageReport) -> str:  # session.rs:500-508
    os.makedirs(report_dir, exist_ok=True)
    path = os.path.join(report_dir, f"{report.name}.json")
    with open(path, "w", encoding="utf-8", newline="") as f:
        f.write(to_string_pretty(report.to_obj()))
    return path


def write_corpus_report(report_dir: str, report: CorpusReport):  # session.rs:511-523
    os.makedirs(report_dir, exist_ok=True)
    jpath = os.path.join(report_dir, f"{report.name}.json")
    with open(jpath, "w", encoding="utf-8", newline="") as f:
        f.write(to_string_pretty(report.to_obj()))
    cpath = os.path.join(report_dir, f"{report.name}.csv")
    with open(cpath, "w", encoding="utf-8", newline="") as f:
        f.write(csv_summary(report))
    return jpath, cpath


# ---- codec-iter baselines (crates/codec-iter/src/eval.rs:21-29, baseline.rs:11-47) ---------------------------
@dataclass
class EvalPoint:
    image: str
    quality: int
    bpp: float
    ssim2: float
    size_bytes: int
    encode_ms: int

    def to_obj(self) -> dict:
        return {"image": self.image, "quality": int(self.quality), "bpp": float(self.bpp), "ssim2": float(self.ssim2),
                "size_bytes": int(self.size_bytes), "encode_ms": int(self.encode_ms)}


@dataclass
class Baseline:
    format: str
    config_summary: str
    corpus_path: str
    created_at: str  # kept as the RFC 3339 string chrono wrote (nanosecond digits survive a round trip)
    points: List[EvalPoint] = field(default_factory=list)

    def to_obj(self) -> dict:
        return {"format": self.format, "config_summary": self.config_summary, "corpus_path": self.corpus_path,
                "created_at": self.created_at, "points": [p.to_obj() for p in self.points]}


def baseline_path(baselines_dir: str, fmt: str) -> str:
    return os.path.join(baselines_dir, f"{fmt}.json")


def save_baseline(baselines_dir: str, baseline: Baseline) -> str:
    os.makedirs(baselines_dir, exist_ok=True)
    path = baseline_path(baselines_dir, baseline.format)
    with open(path, "w", encoding="utf-8", newline="") as f:
        f.write(to_string_pretty(baseline.to_obj()))
    return path


def load_baseline(baselines_dir: str, fmt: str) -> Optional[Baseline]:
    path = baseline_path(baselines_dir, fmt)
    if not os.path.exists(path):
        return None
    with open(path, encoding="utf-8") as f:
        d = json.load(f)
    return Baseline(d["format"], d["config_summary"], d["corpus_path"], d["created_at"], [EvalPoint(**p) for p in d["points"]])


def aggregate_by_quality(points: List[EvalPoint]) -> Dict[int, tuple]:
    """baseline.rs `aggregate_by_quality`: mean bpp and mean ssim2 per quality, summed in point order."""
    acc: Dict[int, list] = {}
    for p in points:
        a = acc.setdefault(p.quality, [0.0, 0.0, 0])
        a[0] += p.bpp
        a[1] += p.ssim2
        a[2] += 1
    return {q: (a[0] / a[2], a[1] / a[2]) for q, a in acc.items()}


@dataclass
class ComparisonRow:  # baseline.rs:49-56
    quality: int
    bpp: float
    ssim2: float
    delta_bpp: float
    delta_ssim2: float
    pareto: float


def compare_with_baseline(points: List[EvalPoint], baseline: Baseline) -> List[ComparisonRow]:
    """baseline.rs:58-86: per quality, mean bpp / ssim2 now vs the baseline; pareto = d_ssim2 - 10 d_bpp."""
    cur, base = aggregate_by_quality(points), aggregate_by_quality(baseline.points)
    rows = []
    for q in sorted(cur):
        bpp, s2 = cur[q]
        d_bpp, d_s2 = (bpp - base[q][0], s2 - base[q][1]) if q in base else (0.0, 0.0)
        rows.append(ComparisonRow(q, bpp, s2, d_bpp, d_s2, d_s2 - d_bpp * 10.0))
    return rows
