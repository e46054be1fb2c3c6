"""One process, every GPU — the Python mirror of `host/codec_eval_multi.hpp`.

The reference's `EvalSession` is `Send + Sync` and `evaluate_image` takes `&self` (src/eval/session.rs:368-434); its
multi-core tools fan whole images out to workers (`images.par_iter()`, crates/codec-compare/src/full_comparison.rs:319-328).
Here: one host thread + one device context per GPU, all pulling WHOLE REFERENCES (every (codec, quality) cell of one
source image) from a shared largest-first queue in guided chunks; one `ce_eval_batch` per chunk; result slots fixed
before the workers start, so the output does not depend on the device count or on which device scored what.  ctypes
releases the GIL inside the library calls, so the device threads run concurrently.

`GuidedQueue` and `DevicePool` take the scorer as a parameter: the queue / ordering / failure logic is tested on a host
without GPUs (tests/test_multidevice.py) with a mocked device count.
"""
from __future__ import annotations

import threading
import time
from dataclasses import dataclass, field
from typing import Callable, Dict, List, Optional, Sequence, Tuple

import numpy as np

from . import (CE_ERR_BACKEND, ColorTable, Context, MetricCalculation, MetricConfig, MetricResult, _error_obj, device_count,
               estimate_batch_bytes)
from . import reports as R
from .session import EvalConfig, EvalSession, ImageData

__all__ = ["GuidedQueue", "ReferenceJob", "DevicePool", "MultiDeviceEvalSession", "largest_first"]


def largest_first(load: Sequence[int]) -> List[int]:
    """Service order: largest job first, index as the tie-break (deterministic)."""
    return sorted(range(len(load)), key=lambda i: (-load[i], i))


class GuidedQueue:
    """Indices in a fixed service order, handed out in guided chunks: max(1, remaining // (2 * workers)) per pull, capped
    by `budget` (sum of `cost`; 0 = unlimited; a single job is always handed out even if it exceeds the budget)."""

    def __init__(self, order: Sequence[int], cost: Sequence[int], workers: int, budget: int = 0):
        self._order, self._cost = list(order), list(cost)
        self._workers, self._budget, self._next = max(int(workers), 1), int(budget), 0
        self._lock = threading.Lock()

    def pull(self) -> List[int]:
        with self._lock:
            remaining = len(self._order) - self._next
            if remaining == 0:
                return []
            want = max(1, remaining // (2 * self._workers))
            out, used = [], 0
            while self._next < len(self._order) and len(out) < want:
                j = self._order[self._next]
                if out and self._budget and used + self._cost[j] > self._budget:
                    break
                used += self._cost[j]
                out.append(j)
                self._next += 1
            return out


@dataclass
class ReferenceJob:
    """One reference's worth of work: the source image and every decoded cell of its sweep."""
    reference: np.ndarray
    width: int
    height: int
    tests: List[np.ndarray] = field(default_factory=list)
    test_profiles: List[Optional[bytes]] = field(default_factory=list)  # ICC profile of each decoded image (None = sRGB)
    scores: list = field(default_factory=list)  # filled by DevicePool.run: one CeScores per test
    device: int = -1


class DevicePool:
    """devices: device indices (default: every visible one).  scorer(worker, jobs) is injectable for tests; the default
    scores a chunk with ONE Context.eval_batch on that worker's context (colour tables built per device on first use)."""

    def __init__(self, devices: Optional[Sequence[int]] = None, scorer: Optional[Callable[[int, List[ReferenceJob]], None]] = None,
                 mock_workers: int = 0, cms: Optional[Callable[[bytes, np.ndarray], np.ndarray]] = None):
        self._scorer, self._cms = scorer, cms
        self._ctxs: List[Context] = []
        self._tables: List[Dict[bytes, ColorTable]] = []
        if scorer is not None and mock_workers:
            self._n = int(mock_workers)
            return
        visible = device_count()
        devs = list(range(visible)) if devices is None else list(devices)
        if not devs or any(d < 0 or d >= visible for d in devs):
            raise MetricCalculation(CE_ERR_BACKEND, f"HIP init failed: {visible} device(s) visible, {devs} requested")
        self._ctxs = [Context(d) for d in devs]
        self._tables = [{} for _ in devs]
        self._n = len(devs)

    @property
    def devices(self) -> int:
        return self._n

    def close(self):
        for tabs in self._tables:
            for t in tabs.values():
                t.close()
        for c in self._ctxs:
            c.close()
        self._ctxs, self._tables = [], []

    def _table(self, w: int, profile: Optional[bytes]) -> Optional[ColorTable]:
        if profile is None:
            return None
        if self._cms is None:
            raise MetricCalculation(CE_ERR_BACKEND, "Metric calculation failed: ICC: ICC profile support requires the 'icc' feature")
        t = self._tables[w].get(profile)
        if t is None:
            t = self._tables[w][profile] = ColorTable(self._ctxs[w], self._cms(profile, ColorTable.identity_cube()))
        return t

    def _score(self, w: int, jobs: List[ReferenceJob], cfg: MetricConfig, intensity_target: float):
        pairs, tables = [], []
        for j in jobs:
            profiles = j.test_profiles or [None] * len(j.tests)
            for t, p in zip(j.tests, profiles):
                pairs.append((j.reference, t, j.width, j.height))
                tables.append(self._table(w, p))
        if not pairs:
            return
        out = self._ctxs[w].eval_batch(pairs, cfg, intensity_target, test_tables=tables if any(t is not None for t in tables) else None)
        k = 0
        for j in jobs:
            j.scores = out[k:k + len(j.tests)]
            k += len(j.tests)

    def run(self, jobs: List[ReferenceJob], cfg: MetricConfig, intensity_target: float = 80.0, device_budget_bytes: int = 0) -> dict:
        """Scores every job; jobs[i].scores is filled for every i.  Returns per-device statistics.  The first worker
        error is re-raised after all workers have stopped."""
        load = [j.width * j.height * max(len(j.tests), 1) for j in jobs]
        cost = [estimate_batch_bytes(j.width, j.height, 1, len(j.tests), cfg) if self._ctxs else 1 for j in jobs]
        budget = device_budget_bytes
        if not budget and self._ctxs:
            budget = min(c.memory_info()[0] for c in self._ctxs) // 3  # ce_eval_batch streams through a ring of three
        for j in jobs:
            j.scores, j.device = [], -1
        q = GuidedQueue(largest_first(load), cost, self._n, budget)
        stats = {"jobs_per_device": [0] * self._n, "pulls_per_device": [0] * self._n, "seconds_per_device": [0.0] * self._n}
        errors: List[BaseException] = []
        failed = threading.Event()

        def worker(w: int):
            t0 = time.perf_counter()
            try:
                while not failed.is_set():
                    idx = q.pull()
                    if not idx:
                        break
                    chunk = [jobs[i] for i in idx]
                    for j in chunk:
                        j.device = w
                    if self._scorer is not None:
                        self._scorer(w, chunk)
                    else:
                        self._score(w, chunk, cfg, intensity_target)
                    stats["jobs_per_device"][w] += len(idx)
                    stats["pulls_per_device"][w] += 1
            except BaseException as e:  # noqa: BLE001 - re-raised on the calling thread
                failed.set()
                errors.append(e)
            stats["seconds_per_device"][w] = time.perf_counter() - t0

        threads = [threading.Thread(target=worker, args=(w,)) for w in range(1, self._n)]
        for t in threads:
            t.start()
        worker(0)
        for t in threads:
            t.join()
        if errors:
            raise errors[0]
        return stats


class MultiDeviceEvalSession:
    """EvalSession over all devices: `evaluate_corpus(images)` runs every image's (codec x quality) sweep (callbacks on the
    calling thread, in the reference's loop order), then the pool scores all decoded cells; reports[i] belongs to
    images[i] and its rows are in the reference's loop order (session.rs:375-410)."""

    def __init__(self, config: EvalConfig, pool: Optional[DevicePool] = None, cms: Optional[Callable[[bytes, np.ndarray], np.ndarray]] = None):
        self.config = config
        self.pool = pool or DevicePool(cms=cms)
        self._own_pool = pool is None
        self._sweeper = EvalSession.__new__(EvalSession)  # the sweep logic only: no device context of its own
        self._sweeper.config, self._sweeper._codecs, self._sweeper._cms = config, [], cms or (pool._cms if pool else None)

    def add_codec(self, id: str, version: str, encode) -> "MultiDeviceEvalSession":
        EvalSession.add_codec(self._sweeper, id, version, encode)
        return self

    def add_codec_with_decode(self, id: str, version: str, encode, decode) -> "MultiDeviceEvalSession":
        EvalSession.add_codec_with_decode(self._sweeper, id, version, encode, decode)
        return self

    def codec_count(self) -> int:
        return len(self._sweeper._codecs)

    def close(self):
        if self._own_pool:
            self.pool.close()

    def evaluate_corpus(self, name: str, images: Sequence[Tuple[str, ImageData]]) -> Tuple[R.CorpusReport, dict]:
        corpus = R.CorpusReport(name, config_summary=f"metrics: {self.config.metrics}")
        jobs, rows = [], []
        for img_name, image in images:
            report, pending = EvalSession._sweep(self._sweeper, img_name, image)
            corpus.images.append(report)
            job = ReferenceJob(image.to_rgb8_vec(), image.width, image.height)
            for row_index, decoded in pending:
                if (decoded.width, decoded.height) != (image.width, image.height):
                    raise _error_obj(1, f"Dimension mismatch: expected ({image.width}, {image.height}), got ({decoded.width}, {decoded.height})")
                job.tests.append(decoded.to_rgb8_vec())
                job.test_profiles.append(decoded.icc_profile)
            jobs.append(job)
            rows.append((report, [ri for ri, _ in pending]))
        stats = self.pool.run(jobs, self.config.metrics)
        for job, (report, row_indices) in zip(jobs, rows):
            for s, ri in zip(job.scores, row_indices):
                if s.status != 0:
                    raise _error_obj(s.status, f"{report.name}: status {s.status}")
                m = MetricResult.from_c(s)
                row = report.results[ri]
                row.dssim, row.ssimulacra2, row.butteraugli, row.psnr = m.dssim, m.ssimulacra2, m.butteraugli, m.psnr
                row.perception = m.perception_level()  # session.rs:407
        return corpus, stats
