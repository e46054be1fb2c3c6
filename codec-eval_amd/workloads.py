"""Synthetic stand-ins for the corpora BASELINE.json's configs name (no Kodak / CID22 images exist
offline; SURVEY.md §8d fixes shapes, seeds and the distortion model).

Reference images: seeded smooth noise (three octaves of bilinearly upsampled uniform noise) with
8x8-aligned hard edges, full u8 range.  Distorted images: a JPEG-like codec model in pure numpy/scipy —
RGB->YCbCr, optional 4:2:0 chroma averaging, 8x8 block DCT, quantisation with the libjpeg tables scaled
by quality, inverse.  Only the shapes and the (image x quality) grid structure matter to the hot path.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import List, Tuple

import numpy as np
from scipy.fft import dctn, idctn

_LUMA_Q = np.array([
    [16, 11, 10, 16, 24, 40, 51, 61], [12, 12, 14, 19, 26, 58, 60, 55], [14, 13, 16, 24, 40, 57, 69, 56],
    [14, 17, 22, 29, 51, 87, 80, 62], [18, 22, 37, 56, 68, 109, 103, 77], [24, 35, 55, 64, 81, 104, 113, 92],
    [49, 64, 78, 87, 103, 121, 120, 101], [72, 92, 95, 98, 112, 100, 103, 99]], np.float64)
_CHROMA_Q = np.array([
    [17, 18, 24, 47, 99, 99, 99, 99], [18, 21, 26, 66, 99, 99, 99, 99], [24, 26, 56, 99, 99, 99, 99, 99],
    [47, 66, 99, 99, 99, 99, 99, 99], [99, 99, 99, 99, 99, 99, 99, 99], [99, 99, 99, 99, 99, 99, 99, 99],
    [99, 99, 99, 99, 99, 99, 99, 99], [99, 99, 99, 99, 99, 99, 99, 99]], np.float64)


def _upsample_bilinear(a: np.ndarray, h: int, w: int) -> np.ndarray:
    sh, sw = a.shape[:2]
    ys = np.linspace(0, sh - 1, h)
    xs = np.linspace(0, sw - 1, w)
    y0 = np.floor(ys).astype(int)
    x0 = np.floor(xs).astype(int)
    y1 = np.minimum(y0 + 1, sh - 1)
    x1 = np.minimum(x0 + 1, sw - 1)
    fy = (ys - y0)[:, None, None]
    fx = (xs - x0)[None, :, None]
    top = a[y0][:, x0] * (1 - fx) + a[y0][:, x1] * fx
    bot = a[y1][:, x0] * (1 - fx) + a[y1][:, x1] * fx
    return top * (1 - fy) + bot * fy


def make_reference(width: int, height: int, seed: int, kind: str = "natural") -> np.ndarray:
    """(h, w, 3) uint8.  kind: natural | flat | highfreq."""
    rng = np.random.default_rng(seed)
    if kind == "flat":
        return np.full((height, width, 3), rng.integers(30, 226, 3), np.uint8)
    if kind == "highfreq":
        return rng.integers(0, 256, (height, width, 3), dtype=np.uint8)
    img = np.zeros((height, width, 3))
    for octave, amp in ((16, 0.55), (64, 0.30), (256, 0.15)):
        gh, gw = max(2, height // (512 // octave) // 8 + 2), max(2, width // (512 // octave) // 8 + 2)
        img += amp * _upsample_bilinear(rng.random((gh, gw, 3)), height, width)
    # 8x8-aligned rectangles with hard edges
    for _ in range(6):
        x0, x1 = sorted(rng.integers(0, width // 8 + 1, 2) * 8)
        y0, y1 = sorted(rng.integers(0, height // 8 + 1, 2) * 8)
        img[y0:y1, x0:x1] = 0.5 * img[y0:y1, x0:x1] + 0.5 * rng.random(3)
    img = (img - img.min()) / max(img.max() - img.min(), 1e-9)
    return np.clip(np.rint(img * 255.0), 0, 255).astype(np.uint8)


def _quality_scale(q: float) -> float:
    q = min(max(q, 1.0), 100.0)
    return (5000.0 / q if q < 50 else 200.0 - 2.0 * q) / 100.0


def _block_quant(plane: np.ndarray, table: np.ndarray) -> np.ndarray:
    h, w = plane.shape
    blocks = plane.reshape(h // 8, 8, w // 8, 8).transpose(0, 2, 1, 3)
    coef = dctn(blocks, axes=(2, 3), norm="ortho")
    coef = np.rint(coef / table) * table
    out = idctn(coef, axes=(2, 3), norm="ortho")
    return out.transpose(0, 2, 1, 3).reshape(h, w)


def distort(reference: np.ndarray, quality: float, subsampling_420: bool = False) -> np.ndarray:
    """JPEG-like distortion of an (h, w, 3) uint8 image at the given quality (1..100)."""
    h, w, _ = reference.shape
    ph, pw = (-h) % 16, (-w) % 16
    rgb = np.pad(reference.astype(np.float64), ((0, ph), (0, pw), (0, 0)), mode="edge")
    r, g, b = rgb[..., 0], rgb[..., 1], rgb[..., 2]
    y = 0.299 * r + 0.587 * g + 0.114 * b - 128.0
    cb = -0.168736 * r - 0.331264 * g + 0.5 * b
    cr = 0.5 * r - 0.418688 * g - 0.081312 * b
    s = _quality_scale(quality)
    lq = np.clip(np.floor(_LUMA_Q * s + 0.5), 1, 255)
    cq = np.clip(np.floor(_CHROMA_Q * s + 0.5), 1, 255)
    y = _block_quant(y, lq)
    if subsampling_420:
        H, W = cb.shape
        cb2 = _block_quant(cb.reshape(H // 2, 2, W // 2, 2).mean(axis=(1, 3)), cq)
        cr2 = _block_quant(cr.reshape(H // 2, 2, W // 2, 2).mean(axis=(1, 3)), cq)
        cb = np.repeat(np.repeat(cb2, 2, 0), 2, 1)
        cr = np.repeat(np.repeat(cr2, 2, 0), 2, 1)
    else:
        cb = _block_quant(cb, cq)
        cr = _block_quant(cr, cq)
    y = y + 128.0
    out = np.stack([y + 1.402 * cr, y - 0.344136 * cb - 0.714136 * cr, y + 1.772 * cb], -1)
    return np.clip(np.rint(out[:h, :w]), 0, 255).astype(np.uint8)


@dataclass
class Grid:
    """A (reference x quality) grid of one shape: references[i] is (h, w, 3); pairs[k] = (ref_index, test).
    When only a shard of the grid was generated (`only` / `units`), ref_ids[i] is the GLOBAL index of references[i]
    and pair_ids[k] = (global reference index, variant index, quality index) of pairs[k]."""
    name: str
    width: int
    height: int
    references: List[np.ndarray]
    pairs: List[Tuple[int, np.ndarray]]
    ref_ids: List[int] = None
    pair_ids: List[Tuple[int, int, int]] = None

    @property
    def megapixels(self) -> float:
        return len(self.pairs) * self.width * self.height / 1e6


def _one_reference(job):
    """(reference, [distorted images of its units, unit-major then quality]) of one source image: the work a pool worker does."""
    w, h, seed, kind, s420_list, qualities = job
    ref = make_reference(w, h, seed, kind)
    return ref, [distort(ref, q, s420) for s420 in s420_list for q in qualities]


_POOL_WORKERS = 0  # set_generation_workers(): > 1 = build grids with a fork pool (call before anything initialises HIP)


def set_generation_workers(n: int):
    global _POOL_WORKERS
    _POOL_WORKERS = max(0, int(n))


def _grid(name, w, h, n_refs, seed0, qualities, variants=((False,),), kinds=None, only=None, units=None) -> Grid:
    """only: global reference indices to generate (a rank's shard; default all).  units: (reference, variant) pairs
    to generate instead (partition by (image, codec-config)) - a reference listed with several
    variants is generated once.  Every image depends on its GLOBAL index only, so any shard of the grid holds the
    same bytes as the same cells of the whole grid."""
    if units is None:
        units = [(i, v) for i in (range(n_refs) if only is None else only) for v in range(len(variants))]
    order, by_ref = [], {}
    for i, v in units:
        if i not in by_ref:
            by_ref[i] = []
            order.append(i)
        by_ref[i].append(v)
    jobs = [(w, h, seed0 + i, (kinds or {}).get(i, "natural"), [variants[v][0] for v in by_ref[i]], tuple(qualities)) for i in order]
    if _POOL_WORKERS > 1 and len(jobs) > 1:
        import multiprocessing as mp

        with mp.get_context("fork").Pool(min(_POOL_WORKERS, len(jobs))) as pool:
            made = pool.map(_one_reference, jobs, chunksize=1)
    else:
        made = [_one_reference(j) for j in jobs]
    refs, pairs, ref_ids, pair_ids, local = [], [], [], [], {}
    tests_of = {}
    for i, (ref, tests) in zip(order, made):
        local[i] = len(refs)
        refs.append(ref)
        ref_ids.append(i)
        for k, v in enumerate(by_ref[i]):
            tests_of[(i, v)] = tests[k * len(qualities):(k + 1) * len(qualities)]
    for i, v in units:  # the caller's unit order, as before
        for qi in range(len(qualities)):
            pairs.append((local[i], tests_of[(i, v)][qi]))
            pair_ids.append((i, v, qi))
    return Grid(name, w, h, refs, pairs, ref_ids, pair_ids)


KODAK_LANDSCAPE, KODAK_PORTRAIT = 18, 6  # 768x512 x18 + 512x768 x6


def kodak_like(qualities=(75, 85, 95), n_landscape=KODAK_LANDSCAPE, n_portrait=KODAK_PORTRAIT, seed0=1000) -> List[Grid]:
    """BASELINE configs[0]/[1]: Kodak-24 shapes; one Grid per shape bucket."""
    out = []
    if n_landscape:
        out.append(_grid("kodak-768x512", 768, 512, n_landscape, seed0, qualities))
    if n_portrait:
        out.append(_grid("kodak-512x768", 512, 768, n_portrait, seed0 + KODAK_LANDSCAPE, qualities))
    return out


def kodak_corpus_shapes(copies: int = 1) -> List[Tuple[int, int]]:
    """(width, height) of every reference of `copies` Kodak-24-shaped sets, global reference order: copy c holds
    references 24c .. 24c+23, the first 18 of a copy are 768x512, the last 6 are 512x768."""
    return [((768, 512) if i % 24 < KODAK_LANDSCAPE else (512, 768)) for i in range(24 * copies)]


def kodak_corpus_shard(ref_indices, qualities=(75, 85, 95), seed0=1000) -> List[Grid]:
    """The references `ref_indices` (global indices into kodak_corpus_shapes) with their quality sweep, one Grid per
    shape.  Reference g is generated from seed0 + g, so shards of different ranks are cells of ONE global grid."""
    shapes = kodak_corpus_shapes(max(ref_indices) // 24 + 1) if len(ref_indices) else []
    out = []
    for (w, h) in ((768, 512), (512, 768)):
        mine = [g for g in ref_indices if shapes[g] == (w, h)]
        if mine:
            out.append(_grid(f"kodak-{w}x{h}", w, h, 0, seed0, qualities, only=mine))
    return out


def uhd_pairs(n=16, seed0=2000, quality=85, only=None) -> Grid:
    """BASELINE configs[2]: synthetic 3840x2160 pairs."""
    return _grid("uhd-3840x2160", 3840, 2160, n, seed0, (quality,), only=only)


STANDARD_QUALITIES = (50, 60, 70, 75, 80, 85, 90, 95)  # crates/codec-iter/src/main.rs:198
DENSE_QUALITIES = tuple(range(50, 99, 2))  # main.rs:199


def cid22_like(n_refs=250, qualities=STANDARD_QUALITIES, seed0=3000, only=None) -> Grid:
    """BASELINE configs[3]: CID22-512 shapes x the standard 8-quality sweep."""
    return _grid("cid22-512x512", 512, 512, n_refs, seed0, qualities, only=only)


DENSE_VARIANTS = ((False,), (True,))  # 4:4:4, 4:2:0 (crates/codec-iter/src/main.rs:474-499)


def codec_iter_dense(n_refs=15, qualities=DENSE_QUALITIES, seed0=4000, only=None, units=None) -> Grid:
    """BASELINE configs[4]: 15-image tier x 25 qualities x {4:4:4, 4:2:0} (the XYB on/off axis is a
    metric flag, applied by the caller).  units: (reference, variant) cells of a rank's shard."""
    return _grid("codec-iter-512x512", 512, 512, n_refs, seed0, qualities, variants=DENSE_VARIANTS, only=only, units=units)
