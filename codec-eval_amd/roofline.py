"""Algorithmic HBM bytes of every kernel of the hot path, per step of a resident grid.

`bench.py` prices a kernel's HIP-event duration against these counts (`roofline.achieved`), DESIGN.md §4
states them, `tests/test_roofline_model.py` pins them against SURVEY.md §8(d)'s per-pixel figures.

Counting rules (applied per KERNEL): a kernel is charged the compulsory traffic of the
stage decomposition it implements — every input plane it needs read once, every output plane written once;
whatever it keeps in LDS / registers is free (fused stages are therefore charged LESS than SURVEY's per-stage
sums, which makes `frac` conservative).  Reference-side inputs that all distorted images of one reference
share (the reference's planes, the two reference-only SSIMULACRA2 blur streams) are charged once per
REFERENCE, not once per pair: the device computes them once per reference, and the pairs of a reference sit
next to each other in the launch order so the repeats are cache hits.

A grid is described by `Bucket(width, height, n_refs, n_pairs)` — one per shape; bytes are summed over the
buckets of a step.  Names are the launch names `ce_prof_*` reports (the `name_` argument of CE_LAUNCH).
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Dict, Iterable, List, Tuple

HBM_PEAK_GBPS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: 8.0 TB/s spec (about 6.3 TB/s is achievable)


@dataclass(frozen=True)
class Bucket:
    width: int
    height: int
    n_refs: int
    n_pairs: int

    @property
    def px(self) -> int:
        return self.width * self.height


def ssim2_levels(w: int, h: int) -> List[Tuple[int, int]]:
    """SSIMULACRA2 pyramid: ceil halving, a level exists while its PARENT is >= 8 px (ssim2.hip: ce_ssim2_prepare)."""
    out = []
    for s in range(6):
        if w < 8 or h < 8:
            break
        if s > 0:
            w, h = (w + 1) // 2, (h + 1) // 2
        out.append((w, h))
    return out


def dssim_levels(w: int, h: int) -> List[Tuple[int, int]]:
    """dssim-core scales: floor halving, 5 at most, a level is halved only while it is >= 8 px (dssim.hip: dssim_prepare)."""
    out = []
    for _ in range(5):
        out.append((w, h))
        if w < 8 or h < 8:
            break
        w, h = w // 2, h // 2
        if w == 0 or h == 0:
            break
    return out


def butteraugli_levels(w: int, h: int) -> List[Tuple[int, int]]:
    """full resolution + the 2x-subsampled pass when that is still >= 8 px (butteraugli.hip: ba_allocate)."""
    w1, h1 = (w + 1) // 2, (h + 1) // 2
    return [(w, h)] + ([(w1, h1)] if w1 >= 8 and h1 >= 8 else [])


def _add(acc: Dict[str, float], name: str, nbytes: float):
    acc[name] = acc.get(name, 0.0) + nbytes


def ssim2_bytes(b: Bucket, acc: Dict[str, float]):
    lv = ssim2_levels(b.width, b.height)
    slots = b.n_refs + b.n_pairs
    for s, (w, h) in enumerate(lv):
        n = w * h
        has_next = s + 1 < len(lv)
        nn = lv[s + 1][0] * lv[s + 1][1] if has_next else 0
        # front end: level's linear RGB in (u8 at level 0), next level's linear RGB out.  The XYB planes it also writes
        # are NOT charged (the conversion is pointwise, i.e. fusable into the passes that consume it), so
        # they show up as measured traffic above the algorithmic count - which is what they are
        _add(acc, "ssim2_prep_u8" if s == 0 else "ssim2_prep", slots * ((3 if s == 0 else 12) * n + 12 * nn))
        # a blur pass moves its blurred streams (4 B x 3 channels each: three per pair, the two reference-only ones per
        # reference) plus its inputs: the XYB planes of both images (R1: u8 at level 0 = 3 B, f32 above = 12 B)
        in_b = 3 if s == 0 else 12
        streams = 12.0 * n * (3 * b.n_pairs + 2 * b.n_refs)
        inputs = in_b * n * (b.n_pairs + b.n_refs)
        suffix = "_L0" if s == 0 else "_L1-5"
        _add(acc, "ssim2_hblur" + suffix, streams + inputs)
        _add(acc, "ssim2_vblur_ssim" + suffix, streams + inputs)


def dssim_bytes(b: Bucket, acc: Dict[str, float]):
    lv = dssim_levels(b.width, b.height)
    slots = b.n_refs + b.n_pairs
    for l, (w, h) in enumerate(lv):
        n = w * h
        nn = lv[l + 1][0] * lv[l + 1][1] if l + 1 < len(lv) else 0
        # create_image: linear RGB in, next level's linear RGB out, and img (3 planes) of a distorted image / img, mu, sq
        # (9 planes) of a reference; L*a*b*, the chroma pre-blur and the 3x3 blur pairs never leave LDS
        _add(acc, "dssim_create_u8" if l == 0 else "dssim_create",
             slots * ((3 if l == 0 else 12) * n + 12 * nn) + 12 * n * b.n_pairs + 36 * n * b.n_refs)
        # compare: img of the distorted image per pair (its mu / sq are formed in the kernel), nine planes of the
        # reference per reference, the SSIM map out
        _add(acc, "dssim_compare", n * (12 * b.n_pairs + 36 * b.n_refs + 4 * b.n_pairs))
        _add(acc, "dssim_absdev", n * 4 * b.n_pairs)


def butteraugli_bytes(b: Bucket, acc: Dict[str, float]):
    lv = butteraugli_levels(b.width, b.height)
    slots, P, R = b.n_refs + b.n_pairs, b.n_pairs, b.n_refs
    for l, (w, h) in enumerate(lv):
        n = w * h
        if l == 0:
            _add(acc, "ba_front_u8", slots * n * (3 + 12))  # u8 in, XYB out (sigma-1.2 blur + opsin dynamics in LDS)
        else:
            _add(acc, "ba_front_half", slots * (3 * b.px + 12 * n))  # the full-resolution u8 in (2x2 averaged on the fly), XYB out
        # SeparateFrequencies: row blur -> column blur fused with the split.  Planes are 4 B/px.
        _add(acc, "ba_blur_h33", slots * n * (12 + 12))
        _add(acc, "ba_blur_v_lf", slots * n * (12 + 12 + 24))   # row-blurred + XYB in; LF + raw MF out
        _add(acc, "ba_blur_hv_mf", slots * n * (12 + 12 + 8))   # raw MF in (row + column pass in one kernel); MF x3 + raw HF x2 out
        _add(acc, "ba_blur_hv_hf", slots * n * (8 + 16 + 4))    # raw HF in; HF x2 + UHF x2 + the mask input out
        # the mask input's sigma-2.7 blur per image slot; the references' two mask-value planes
        _add(acc, "ba_blur_hv_mask", slots * n * 8)
        _add(acc, "ba_mask_vals", R * n * (4 + 8))
        # per pair: Malta + L2 terms + CombineChannelsToDiffmap read the ten PsychoImage planes and the blurred mask plane
        # of both images and the reference's two mask-value planes (the AC / DC triples stay in registers); the
        # half-resolution level writes its diffmap, the full-resolution level reads that (a quarter of its pixels) and
        # reduces its own values to the score partials without storing them
        has_sub = len(lv) == 2
        io = 4 * P if l == 1 else (1 * P if has_sub else 0)
        _add(acc, "ba_malta_l2", n * ((40 + 4) * P + (40 + 4 + 8) * R) + n * io)


def step_bytes(buckets: Iterable[Bucket], metrics: Iterable[str], xyb_roundtrip: bool = False) -> Dict[str, float]:
    """{launch name: algorithmic bytes per step} for one pass over `buckets` with `metrics` enabled."""
    acc: Dict[str, float] = {}
    metrics = set(metrics)
    for b in buckets:
        if "ssimulacra2" in metrics and min(b.width, b.height) >= 8:
            ssim2_bytes(b, acc)
        if "dssim" in metrics:
            dssim_bytes(b, acc)
        if "butteraugli" in metrics and min(b.width, b.height) >= 8:
            butteraugli_bytes(b, acc)
        if "psnr" in metrics:
            _add(acc, "psnr_sse", 3.0 * b.px * (b.n_pairs + b.n_refs))
        if xyb_roundtrip:
            _add(acc, "xyb_roundtrip", 6.0 * b.px * b.n_refs)
    return acc


# ---- round 2's byte model, FROZEN --------------------------------------------------------------
# As stages fuse, a kernel's compulsory bytes shrink and `pipeline_frac` falls while the step gets faster, so fractions of
# different rounds are not comparable.  This table is round 2's model (BENCH_r02: 16.56 GB per step of the 72-pair Kodak
# grid, all three metrics) in bytes per pixel OF THE KERNEL'S LEVEL: P = per pair, R = per reference, S = per image slot,
# (lvl0, upper) where level 0 reads u8.  bench.py prices every round's step time against it as
# `pipeline_frac_r02_model`; step_bytes() above is the CURRENT kernels' model.
R02_MODEL = {
    # name: (per-level pyramid, [(unit, bytes/px at level 0, bytes/px at levels >= 1, bytes per NEXT-level px)])
    "ssim2_prep": ("ssim2", [("S", 3, 12, 12)]),
    "ssim2_hblur": ("ssim2", [("P", 36 + 3, 36 + 12, 0), ("R", 24 + 3, 24 + 12, 0)]),
    "ssim2_vblur_ssim": ("ssim2", [("P", 36 + 3, 36 + 12, 0), ("R", 24 + 3, 24 + 12, 0)]),
    "dssim_create": ("dssim", [("S", 3, 12, 12), ("P", 12, 12, 0), ("R", 36, 36, 0)]),
    "dssim_compare": ("dssim", [("P", 16, 16, 0), ("R", 36, 36, 0)]),
    "dssim_absdev": ("dssim", [("P", 4, 4, 0)]),
}


def step_bytes_r02_model(buckets: Iterable[Bucket], metrics: Iterable[str], xyb_roundtrip: bool = False) -> float:
    """Bytes per step under round 2's model.  SSIMULACRA2 / DSSIM from R02_MODEL; Butteraugli's round-2 rows are exactly
    butteraugli_bytes() as round 2 left it (restated here so that later fusions do not move it)."""
    metrics = set(metrics)
    total = 0.0
    for b in buckets:
        P, R, S = b.n_pairs, b.n_refs, b.n_pairs + b.n_refs
        unit = {"P": P, "R": R, "S": S}
        for name, (pyr, rows) in R02_MODEL.items():
            if ("ssimulacra2" if pyr == "ssim2" else "dssim") not in metrics:
                continue
            if pyr == "ssim2" and min(b.width, b.height) < 8:
                continue
            lv = ssim2_levels(b.width, b.height) if pyr == "ssim2" else dssim_levels(b.width, b.height)
            for l, (w, h) in enumerate(lv):
                nn = lv[l + 1][0] * lv[l + 1][1] if l + 1 < len(lv) else 0
                for u, b0, b1, bn in rows:
                    total += unit[u] * ((b0 if l == 0 else b1) * w * h + bn * nn)
        if "butteraugli" in metrics and min(b.width, b.height) >= 8:
            lv = butteraugli_levels(b.width, b.height)
            for l, (w, h) in enumerate(lv):
                n = w * h
                total += S * (n * 15 if l == 0 else 3 * b.px + 12 * n)            # front end
                total += S * n * (24 + 48 + 32 + 28 + 8) + R * n * 12               # blur stages, mask values
                io = 4 * P if l == 1 else (P if len(lv) == 2 else 0)
                total += n * (44 * P + 52 * R) + n * io                            # Malta + L2 + combine
        if "psnr" in metrics:
            total += 3.0 * b.px * (P + R)
        if xyb_roundtrip:
            total += 6.0 * b.px * R
    return total


METRIC_OF_PREFIX = (("ssim2_", "ssimulacra2"), ("dssim_", "dssim"), ("ba_", "butteraugli"), ("psnr", "psnr"), ("xyb_", "xyb_roundtrip"))


def metric_of(kernel: str) -> str:
    for prefix, metric in METRIC_OF_PREFIX:
        if kernel.startswith(prefix):
            return metric
    return "other"


def uncached_pair_bytes_per_px0(metric: str) -> float:
    """SURVEY.md §8(d) / BASELINE.md §4 per-stage counts for ONE uncached pair, per scale-0 pixel (for comparison)."""
    return {"ssimulacra2": 210.0, "dssim": 238.0, "butteraugli": 826.0, "psnr": 6.0, "xyb_roundtrip": 6.0}[metric]
