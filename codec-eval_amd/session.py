"""EvalSession — the reference's primary entry (src/eval/session.rs:281-497), batched for the device.

Same names, argument meaning and error behaviour as the reference; what differs is the schedule:
the reference scores one (codec, quality) cell at a time on the calling thread (session.rs:375-410),
here every cell of an image — or of a whole corpus — is encoded/decoded first, then all decoded
images go to the device as ONE batch per shape (reference uploaded once per image, decoder output
converted to RGB8 on the device), and the scores are filled into the same `CodecResult` rows in the
reference's loop order.  Report types and writers are in `reports.py`.
"""
from __future__ import annotations

import os
import time
from dataclasses import dataclass, field
from typing import Callable, Dict, List, Optional, Sequence, Tuple

import numpy as np

from . import (PIXEL_RGB8, PIXEL_RGBA8, Batch, CodecEvalError, ColorTable, Context, DimensionMismatch, MetricCalculation,
               MetricConfig, MetricResult, _error_obj, estimate_batch_bytes, CE_ERR_BACKEND)
from . import reports as R

__all__ = ["ImageData", "EncodeRequest", "EvalConfig", "EvalConfigBuilder", "EvalSession"]


@dataclass
class ImageData:
    """session.rs:25-149.  `RgbSlice`, `RgbaSlice`, `RgbSliceWithIcc` (the imgref variants carry the same bytes)."""
    data: np.ndarray  # packed u8, RGB or RGBA
    width: int
    height: int
    channels: int = 3
    icc_profile: Optional[bytes] = None

    @staticmethod
    def rgb(data, width: int, height: int) -> "ImageData":
        return ImageData(np.ascontiguousarray(data, dtype=np.uint8).reshape(-1), int(width), int(height), 3)

    @staticmethod
    def rgba(data, width: int, height: int) -> "ImageData":
        return ImageData(np.ascontiguousarray(data, dtype=np.uint8).reshape(-1), int(width), int(height), 4)

    @staticmethod
    def rgb_with_icc(data, width: int, height: int, icc_profile: bytes) -> "ImageData":
        return ImageData(np.ascontiguousarray(data, dtype=np.uint8).reshape(-1), int(width), int(height), 3, bytes(icc_profile))

    def to_rgb8_vec(self) -> np.ndarray:  # session.rs:98-117 (host copy; the session itself strips alpha on the device)
        if self.channels == 3:
            return self.data
        return np.ascontiguousarray(self.data.reshape(-1, 4)[:, :3]).reshape(-1)

    def to_rgb8_srgb(self, cms: Optional[Callable[[bytes, np.ndarray], np.ndarray]] = None) -> np.ndarray:
        """session.rs:143-147 -> transform_to_srgb, metrics/icc.rs:69-113: on the HOST (the session itself applies the
        profile on the device through a colour table).  `cms(profile_bytes, rgb_nx3_u8) -> rgb_nx3_u8` is the colour
        management to use (the reference's is moxcms); without one a tagged image fails like a build without `icc`."""
        rgb = self.to_rgb8_vec()
        if self.icc_profile is None:
            return rgb
        if cms is None:
            self._check_profile()
        return np.ascontiguousarray(cms(self.icc_profile, rgb.reshape(-1, 3)), dtype=np.uint8).reshape(-1)

    def _check_profile(self):
        if self.icc_profile is not None:
            # the reference built without its `icc` feature (icc.rs:105-113): no colour management available
            raise MetricCalculation(CE_ERR_BACKEND, "Metric calculation failed: ICC: ICC profile support requires the 'icc' feature")

    @property
    def pixel_format(self) -> int:
        return PIXEL_RGB8 if self.channels == 3 else PIXEL_RGBA8


@dataclass
class EncodeRequest:  # session.rs:151-177
    quality: float
    params: Dict[str, str] = field(default_factory=dict)

    def with_param(self, key: str, value: str) -> "EncodeRequest":
        self.params[str(key)] = str(value)
        return self


EncodeFn = Callable[[ImageData, EncodeRequest], bytes]
DecodeFn = Callable[[bytes], ImageData]


@dataclass
class EvalConfig:  # session.rs:188-279
    report_dir: str
    cache_dir: Optional[str] = None
    viewing: Optional[object] = None  # ViewingCondition: carried, not used by any metric (dssim.rs:40 ignores it too)
    metrics: MetricConfig = field(default_factory=MetricConfig.all)
    quality_levels: List[float] = field(default_factory=lambda: [50.0, 60.0, 70.0, 80.0, 85.0, 90.0, 95.0])

    @staticmethod
    def builder() -> "EvalConfigBuilder":
        return EvalConfigBuilder()


class EvalConfigBuilder:
    def __init__(self):
        self._report_dir = self._cache_dir = self._viewing = self._metrics = self._levels = None

    def report_dir(self, path):
        self._report_dir = str(path)
        return self

    def cache_dir(self, path):
        self._cache_dir = str(path)
        return self

    def viewing(self, viewing):
        self._viewing = viewing
        return self

    def metrics(self, metrics: MetricConfig):
        self._metrics = metrics
        return self

    def quality_levels(self, levels: Sequence[float]):
        self._levels = [float(q) for q in levels]
        return self

    def build(self) -> EvalConfig:
        if self._report_dir is None:
            raise ValueError("report_dir is required")  # session.rs:269 `.expect("report_dir is required")`
        cfg = EvalConfig(self._report_dir, self._cache_dir, self._viewing)
        if self._metrics is not None:
            cfg.metrics = self._metrics
        if self._levels is not None:
            cfg.quality_levels = self._levels
        return cfg


@dataclass
class _CodecEntry:
    id: str
    version: str
    encode: EncodeFn
    decode: Optional[DecodeFn]


class EvalSession:
    """session.rs:309-497.  One session = one device context; `evaluate_image` may be called for any shape."""

    def __init__(self, config: EvalConfig, ctx: Optional[Context] = None, device: int = 0,
                 cms: Optional[Callable[[bytes, np.ndarray], np.ndarray]] = None):
        """cms(profile_bytes, rgb (n, 3) uint8) -> (n, 3) uint8: the host's ICC -> sRGB transform (the reference's default
        build uses moxcms, icc.rs:69-103).  It is evaluated ONCE per distinct profile on the identity colour cube; the
        resulting 2^24-entry table lives on the device and is applied to every decoded image tagged with that profile
        (session.rs:394: only DECODED images go through to_rgb8_srgb, the source image does not).  Without a cms a tagged
        decoded image raises what a build without the `icc` feature raises."""
        self.config = config
        self.ctx = ctx or Context(device)
        self._own_ctx = ctx is None
        self._codecs: List[_CodecEntry] = []
        self._cms = cms
        self._tables: Dict[bytes, ColorTable] = {}

    def close(self):
        for t in self._tables.values():
            t.close()
        self._tables.clear()
        if self._own_ctx:
            self.ctx.close()

    def _table_for(self, image: ImageData) -> Optional[ColorTable]:
        """The device colour table of a decoded image's profile (None for untagged = sRGB images, icc.rs:73)."""
        if image.icc_profile is None:
            return None
        if self._cms is None:
            image._check_profile()
        key = bytes(image.icc_profile)
        t = self._tables.get(key)
        if t is None:
            out = np.asarray(self._cms(key, ColorTable.identity_cube()), dtype=np.uint8)
            if out.shape != (1 << 24, 3):
                raise MetricCalculation(CE_ERR_BACKEND, f"Metric calculation failed: ICC: Failed to apply ICC transform: the cms returned shape {out.shape}")
            t = self._tables[key] = ColorTable(self.ctx, out)
        return t

    def add_codec(self, id: str, version: str, encode: EncodeFn) -> "EvalSession":  # :325-334
        self._codecs.append(_CodecEntry(id, version, encode, None))
        return self

    def add_codec_with_decode(self, id: str, version: str, encode: EncodeFn, decode: DecodeFn) -> "EvalSession":  # :336-351
        self._codecs.append(_CodecEntry(id, version, encode, decode))
        return self

    def codec_count(self) -> int:
        return len(self._codecs)

    # -- the sweep -------------------------------------------------------------------------------
    def _sweep(self, name: str, image: ImageData):
        """Encode/decode every (codec, quality) cell of one image (session.rs:375-428 minus the metrics).
        Returns the report with metric-less rows and the list of (row index, decoded ImageData)."""
        width, height = image.width, image.height
        report = R.ImageReport(name, width, height)
        pending: List[Tuple[int, ImageData]] = []
        for codec in self._codecs:
            for quality in self.config.quality_levels:
                request = EncodeRequest(float(quality))
                t0 = time.perf_counter()
                encoded = codec.encode(image, request)
                encode_ms = int((time.perf_counter() - t0) * 1000)
                row = R.CodecResult(codec.id, codec.version, float(quality), len(encoded),
                                    (len(encoded) * 8) / (float(width) * float(height)), encode_ms,
                                    codec_params=dict(request.params))
                if codec.decode is not None:
                    t0 = time.perf_counter()
                    decoded = codec.decode(encoded)
                    row.decode_time_ms = int((time.perf_counter() - t0) * 1000)
                    if self._cms is None:
                        decoded._check_profile()  # to_rgb8_srgb's failure mode without colour management, before anything reaches the device
                    pending.append((len(report.results), decoded))
                report.results.append(row)
        return report, pending

    def _score(self, jobs: List[Tuple[ImageData, R.ImageReport, List[Tuple[int, ImageData]]]]):
        """All cells of all images, one device batch per shape; references uploaded once per image."""
        cfg = self.config.metrics
        buckets: Dict[Tuple[int, int], list] = {}
        for image, report, pending in jobs:
            if pending:
                buckets.setdefault((image.width, image.height), []).append((image, report, pending))
        for (w, h), group in buckets.items():
            # a shape's cells go through device batches that fit the device: whole images while they fit, an image with
            # more cells than one batch holds is split (its reference is uploaded once per part)
            free, _total = self.ctx.memory_info()
            budget = int(os.environ.get("CE_SESSION_BATCH_BYTES", 0)) or int(free * 0.6)
            fixed = estimate_batch_bytes(w, h, 0, 0, cfg)
            per_ref = estimate_batch_bytes(w, h, 1, 0, cfg) - fixed
            per_pair = estimate_batch_bytes(w, h, 0, 1, cfg) - fixed
            max_pairs = max(1, (budget - fixed - per_ref) // max(per_pair, 1))
            parts: List[list] = [[]]  # each part: [(image, report, cells)]
            used = fixed
            for image, report, pending in group:
                for o in range(0, len(pending), max_pairs):
                    cells = pending[o:o + max_pairs]
                    need = per_ref + per_pair * len(cells)
                    if parts[-1] and used + need > budget:
                        parts.append([])
                        used = fixed
                    parts[-1].append((image, report, cells))
                    used += need
            for part in parts:
                self._score_part(w, h, part, cfg)

    def _score_part(self, w: int, h: int, group, cfg: MetricConfig):
        n_refs = len(group)
        n_pairs = sum(len(p) for _, _, p in group)
        batch = Batch(self.ctx, w, h, n_refs, n_pairs)
        try:
            rows = []
            k = 0
            for ri, (image, report, pending) in enumerate(group):
                batch.set_reference_fmt(ri, image.data, image.pixel_format)
                for row_index, decoded in pending:
                    if (decoded.width, decoded.height) != (w, h):  # calculate_metrics' length check, ssimulacra2.rs:65-70
                        raise DimensionMismatch(1, f"Dimension mismatch: expected ({w}, {h}), got ({decoded.width}, {decoded.height})")
                    batch.set_test_lut(k, ri, decoded.data, decoded.pixel_format, self._table_for(decoded))  # to_rgb8_srgb, session.rs:394
                    rows.append((report, row_index))
                    k += 1
            scores = batch.run(n_pairs, cfg)
        finally:
            batch.close()
        for (report, row_index), s in zip(rows, scores):
            if s.status != 0:
                raise _error_obj(s.status, self.ctx._err())
            m = MetricResult.from_c(s)
            row = report.results[row_index]
            row.dssim, row.ssimulacra2, row.butteraugli, row.psnr = m.dssim, m.ssimulacra2, m.butteraugli, m.psnr
            row.perception = m.perception_level()  # session.rs:407

    def evaluate_image(self, name: str, image: ImageData) -> R.ImageReport:  # session.rs:368-434
        # the source image enters as to_rgb8_vec() (session.rs:373): its own profile, if any, is NOT applied
        report, pending = self._sweep(name, image)
        self._score([(image, report, pending)])
        return report

    def evaluate_corpus(self, name: str, images: Sequence[Tuple[str, ImageData]], rank: int = 0, world: int = 1) -> R.CorpusReport:
        """Every image of a corpus in one pass (the loop a caller of evaluate_image writes, e.g. examples/ and
        crates/codec-compare): all cells of all images of a shape share one device batch.  With world > 1 the
        images are partitioned by reference (sharding.assign_references) and this rank scores its own."""
        from .sharding import assign_references

        corpus = R.CorpusReport(name, config_summary=f"metrics: {self.config.metrics}")
        mine = list(range(len(images)))
        if world > 1:
            mine = assign_references([im.width * im.height for _, im in images], world)[rank]
        jobs = []
        for i in mine:
            img_name, image = images[i]
            report, pending = self._sweep(img_name, image)
            jobs.append((image, report, pending))
            corpus.images.append(report)
        self._score(jobs)
        return corpus

    # -- writers (session.rs:500-584) ---------------------------------------------------------------
    def write_image_report(self, report: R.ImageReport) -> str:
        return R.write_image_report(self.config.report_dir, report)

    def write_corpus_report(self, report: R.CorpusReport):
        return R.write_corpus_report(self.config.report_dir, report)
