"""Multi-GPU sharding of the (image x codec x quality) grid — SURVEY.md §8(e).

Work items are independent (the reference runs them in a serial double loop, src/eval/session.rs:375-376,
or under images.par_iter(), crates/codec-compare/src/full_comparison.rs:319-328), so the grid shards with
NO data-path collective: one process + one HIP stream per GPU, partitioned BY REFERENCE IMAGE so that all
qualities / codec variants of one source land on the same GPU (one reference upload, shared reference
planes).  Scores (<= 4 doubles per item) are gathered on the host in deterministic (image, variant) order.
torch.distributed is used for control only: barrier, max-over-ranks time, gather of the tiny score lists.
"""
from __future__ import annotations

from typing import List, Sequence, Tuple


def assign_references(pixel_counts: Sequence[int], world_size: int) -> List[List[int]]:
    """Greedy longest-processing-time partition: references sorted by pixel count (desc, index as
    tie-break) go to the currently least-loaded rank.  Deterministic; returns ref indices per rank."""
    order = sorted(range(len(pixel_counts)), key=lambda i: (-pixel_counts[i], i))
    load = [0] * world_size
    out: List[List[int]] = [[] for _ in range(world_size)]
    for i in order:
        r = min(range(world_size), key=lambda k: (load[k], k))
        out[r].append(i)
        load[r] += pixel_counts[i]
    for r in out:
        r.sort()
    return out


def imbalance(loads: Sequence[float]) -> float:
    """max / mean of the per-rank loads (1.0 = perfectly balanced)."""
    mean = sum(loads) / max(len(loads), 1)
    return max(loads) / mean if mean > 0 else 1.0


def plan_partition(ref_pixels: Sequence[int], tests_per_ref: Sequence[int], variants_per_ref: int, world_size: int,
                   tolerance: float = 0.01):
    """partition BY REFERENCE (one upload and one set of reference planes per source image); when
    there are too few references for the ranks (config 5: 15 images on 8 GPUs) fall back to partitioning by
    (image, codec-config) unit if that balances strictly better.  Load = pixels x tests.

    Returns (mode, per_rank_units, per_rank_load): mode is "reference" or "image-x-codec-config"; a unit is
    (reference index, variant index); in "reference" mode a rank owns every variant of its references."""
    V = max(1, variants_per_ref)
    ref_load = [p * t for p, t in zip(ref_pixels, tests_per_ref)]
    by_ref = assign_references(ref_load, world_size)
    load_ref = [sum(ref_load[i] for i in r) for r in by_ref]
    units_ref = [[(i, v) for i in r for v in range(V)] for r in by_ref]
    if V == 1:
        return "reference", units_ref, load_ref
    unit_load = [ref_load[i] / V for i in range(len(ref_pixels)) for _ in range(V)]
    by_unit = assign_references(unit_load, world_size)
    load_unit = [sum(unit_load[u] for u in r) for r in by_unit]
    if imbalance(load_unit) < imbalance(load_ref) - tolerance:
        return "image-x-codec-config", [[(u // V, u % V) for u in r] for r in by_unit], load_unit
    return "reference", units_ref, load_ref


def shard_items(item_refs: Sequence[int], ref_owner: Sequence[int], rank: int) -> List[int]:
    """Indices of the work items (each tagged with its reference index) owned by `rank`."""
    return [k for k, ri in enumerate(item_refs) if ref_owner[ri] == rank]


def owner_table(assignment: List[List[int]], n_refs: int) -> List[int]:
    owner = [-1] * n_refs
    for r, refs in enumerate(assignment):
        for i in refs:
            owner[i] = r
    assert all(o >= 0 for o in owner)
    return owner


def gather_scores(local: List[Tuple[int, tuple]], dist=None, dst: int = 0):
    """local = [(global_item_index, scores...)].  Returns the full list ordered by item index on `dst`
    (None elsewhere).  With dist=None (single process) it just sorts."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return sorted(local, key=lambda t: t[0])
    world = dist.get_world_size()
    bucket = [None] * world if dist.get_rank() == dst else None
    dist.gather_object(local, bucket, dst=dst)
    if dist.get_rank() != dst:
        return None
    merged = [t for part in bucket for t in part]
    merged.sort(key=lambda t: t[0])
    return merged


def max_over_ranks(seconds: float, dist=None, device=None) -> float:
    """The bench contract's 'take the MAX over ranks' of the timed region."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return seconds
    import torch

    t = torch.tensor([seconds], dtype=torch.float64, device=device or "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
