#!/usr/bin/env python3
"""bench.py — BASELINE.json's metric on the MI355X hot path.

    python bench.py --gpus N --steps K --warmup W
    (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

Workload (N = 1): BASELINE.json configs[1] — "Kodak-24 x 3 quality levels, SSIMULACRA2 only, one
MI355X (768x512 buffers)": 24 synthetic references in Kodak's shapes (18 of 768x512, 6 of 512x768)
x q in {75, 85, 95} = 72 (reference, distorted) pairs, 28.3 MP per step.  Inputs are uploaded once
and are resident in HBM when the timed region starts.  A step = one pass of the hot path over the
whole grid, scores returned to the host.

N > 1: the grid shards by reference image with no data-path collective; every rank owns its own
Kodak-24-shaped corpus (different seeds), i.e. per-GPU work is fixed => "scaling": "weak".
torch.distributed (RCCL) is used only for the barriers and the max-over-ranks of the time.

Extra JSON objects: "roofline" (dominant kernel, HIP-event timed in the timed region) and "cpu_baseline"
(the C oracle timed on this host's cores, rank 0, N = 1 only).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

# before anything initialises HIP (torch does): see codec-eval_amd/__init__.py
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: 8.0 TB/s spec
# SURVEY.md §8(d) algorithmic bytes, counting rules R1-R6
SSIM2_BYTES_PER_PX0_TOTAL = 210.0  # whole metric, per scale-0 pixel of a pair
# the other leaves, counted with the same rules from the oracle's stage lists (BASELINE.md §4 has the sums)
METRIC_BYTES_PER_PX0 = {"ssimulacra2": SSIM2_BYTES_PER_PX0_TOTAL, "dssim": 238.0, "butteraugli": 826.0, "psnr": 6.0}
SSIM2_PASS_BYTES_L0 = 66.0  # one blur pass at level 0 for an uncached pair: 60 B of blurred planes + 6 B of u8 input
# The two reference-only blur streams (a, a*a: 24 B of the 60) are produced once per REFERENCE and shared by its
# distorted images, so a pass over a grid moves 12 B x (3 streams per pair + 2 per reference) + its inputs.  The
# roofline is quoted on these (smaller) counts, not on 66 B for every pair.
SSIM2_STREAM_BYTES = 12.0  # one blurred stream, three channels, f32


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--quick", action="store_true", help="6 references instead of 24 (smoke runs)")
    ap.add_argument("--config", type=int, default=2, choices=(2, 3, 4, 5),
                    help="BASELINE.json configs[] entry, 1-based (default 2 = the headline workload). "
                         "3/4/5 are extra measurements, scaled down with --refs")
    ap.add_argument("--refs", type=int, default=0, help="override the number of reference images (configs 3-5)")
    ap.add_argument("--no-events", action="store_true", help="no per-kernel HIP events in the timed region (no roofline)")
    ap.add_argument("--one-context", action="store_true", help="all shape buckets on one context (buckets run back to back)")
    ap.add_argument("--all-events", action="store_true", help="events around every kernel in the timed region, not only level 0")
    ap.add_argument("--solo", action="store_true", help="(default at N = 1) extra untimed pass: time every kernel alone on one stream")
    ap.add_argument("--no-solo", action="store_true", help="skip the solo pass")
    ap.add_argument("--depth", type=int, default=0,
                    help="batches in flight per shape bucket (1 = launch and collect each step before the next); "
                         "default: 2 for configs 2 and 3, 1 for the mixed-metric configs 4 and 5 (measured)")
    return ap.parse_args()


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus N > 1 must be launched with torch.distributed.run --nproc-per-node N")
        args.gpus = world

    import numpy as np
    import torch

    import codec_eval_amd as ce
    import importlib

    wl = importlib.import_module("codec-eval_amd.workloads")

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (torch.cuda.is_available() is False)")
    # CE_BENCH_SHARE_DEVICE=1 (rehearsal of the N > 1 path on a box with fewer GPUs than ranks): ranks share devices
    # and the rendezvous uses gloo.  Never set by the driver.
    share = os.environ.get("CE_BENCH_SHARE_DEVICE") == "1"
    if share:
        local_rank = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist  # noqa: F811

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if share:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    # ---- build this rank's shard of the grid (synthetic, seeded) --------------------------------
    qualities = (75, 85, 95)  # codec-iter "quick" preset, crates/codec-iter/src/main.rs:197
    n_land, n_port = (5, 1) if args.quick else (wl.KODAK_LANDSCAPE, wl.KODAK_PORTRAIT)
    if args.config == 2:
        grids = wl.kodak_like(qualities, n_land, n_port, seed0=1000 + 100 * rank)
        cfg = ce.MetricConfig.ssimulacra2_only()
        workload = ("BASELINE configs[1]: Kodak-24 x 3 quality levels (q75/85/95), SSIMULACRA2 only, "
                    "768x512 x18 + 512x768 x6 buffers")
    elif args.config == 3:
        n = args.refs or 4
        grids = [wl.uhd_pairs(n, seed0=2000 + 100 * rank)]
        cfg = ce.MetricConfig(butteraugli=True)
        workload = f"BASELINE configs[2]: {n} synthetic 3840x2160 pairs, Butteraugli (max-norm + 3-norm)"
    elif args.config == 4:
        n = args.refs or 32
        grids = [wl.cid22_like(n, seed0=3000 + 1000 * rank)]
        cfg = ce.MetricConfig(ssimulacra2=True, dssim=True)
        workload = f"BASELINE configs[3]: {n} CID22-shaped 512x512 refs x 8 qualities, SSIMULACRA2 + DSSIM"
    else:
        n = args.refs or 15
        grids = [wl.codec_iter_dense(n, seed0=4000 + 100 * rank)]
        cfg = ce.MetricConfig.all().with_xyb_roundtrip()
        workload = f"BASELINE configs[4]: {n} 512x512 refs x 25 qualities x {{4:4:4, 4:2:0}}, all metrics, XYB roundtrip on"

    # One context (= one HIP stream family) per shape bucket, so the buckets' kernel chains overlap on the GPU.
    # `--depth` > 1 keeps that many sets of batches in flight (step k is launched before step k-1's scores are
    # collected, the way EvalSession streams a corpus larger than one batch); every timed step's scores are
    # still collected inside the timed region.  Measured with 16 hardware queues: 21.5 k MP/s at depth 1, 22.2 k at
    # depth 2 (the next step's front end fills the end of the previous step), no further gain at 3.  With HIP's default
    # of 4 hardware queues depth 2 was SLOWER than depth 1: the extra streams queued behind busy ones.
    depth = args.depth if args.depth > 0 else (2 if args.config in (2, 3) else 1)
    sets = []
    for _ in range(depth):
        cs = [ce.Context(local_rank) for _ in grids]
        if args.one_context:
            cs = [cs[0]] * len(grids)
        bs = []
        for g, c in zip(grids, cs):
            b = ce.Batch(c, g.width, g.height, len(g.references), len(g.pairs))
            for i, r in enumerate(g.references):
                b.set_reference(i, r)
            for k, (ri, t) in enumerate(g.pairs):
                b.set_test(k, ri, t)
            bs.append((g, b))
        sets.append((cs, bs))
    ctxs, batches = sets[0]
    ctx = ctxs[0]
    pairs_per_step = sum(len(g.pairs) for g in grids)
    mp_per_step = sum(g.megapixels for g in grids)

    def launch(k):
        for g, b in sets[k % depth][1]:
            b.launch(len(g.pairs), cfg)

    def collect(k):
        return [b.collect(len(g.pairs)) for g, b in sets[k % depth][1]]

    def run_steps(n):
        out = None
        for k in range(n):
            launch(k)
            if k >= depth - 1:
                out = collect(k - (depth - 1))
        for k in range(max(0, n - (depth - 1)), n):
            out = collect(k)
        return out

    # Per-kernel HIP events are recorded IN the timed region, each pair on the stream its kernel is launched on
    # and with the batch's normal multi-stream schedule (prof mode 2): the durations are what rocprofv3's kernel
    # trace of this same command reports (profiles/).  --no-events switches them off (A/B: no measurable cost).
    events = not args.no_events
    all_ctxs = list({id(c): c for cs, _ in sets for c in cs}.values())
    for c in all_ctxs:
        # --all-events: every kernel; default: only the level-0 kernels (the candidates for the dominant kernel),
        # which keeps the cost of the events in the timed region near 1 %
        c.prof_filter("" if args.all_events else "_L0")
        c.prof_enable(events, serial=False)

    # one untimed pass over every set of batches first: lazy device allocations and the host-built work lists are part
    # of setting a batch up, not of a step (with --depth 2 a short warm-up would otherwise leave the second set cold)
    for _, bs in sets:
        for g, b in bs:
            b.launch(len(g.pairs), cfg)
            b.collect(len(g.pairs))
    run_steps(args.warmup)
    for c in all_ctxs:
        c.prof_reset()

    barrier()
    t0 = time.perf_counter()
    scores = run_steps(args.steps)
    barrier()
    elapsed = time.perf_counter() - t0

    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if share else "cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # MP/s = reference pixels x (pair, metric) evaluations / wall time
    n_metrics = sum(1 for m in ("dssim", "ssimulacra2", "butteraugli", "psnr") if getattr(cfg, m))
    total_mp = mp_per_step * n_metrics * args.steps * world
    value = total_mp / elapsed
    ms_per_step = elapsed / args.steps * 1e3

    def gather_stats():
        acc = {}
        for c in all_ctxs:
            for k, (n, ms) in c.prof_stats().items():
                n0, ms0 = acc.get(k, (0, 0.0))
                acc[k] = (n0 + n, ms0 + ms)
        return acc

    # ---- roofline: the dominant kernel of the timed region --------------------------------------
    roofline = None
    kernels = gather_stats() if events else {}
    for c in all_ctxs:
        c.prof_enable(False)
    solo = {}
    if rank == 0 and (args.solo or (world == 1 and args.config == 2)) and not args.no_solo:
        # optional extra pass: one kernel at a time on one stream ("solo" durations, no sharing of the GPU)
        for c in ctxs:
            c.prof_reset()
            c.prof_filter("")
            c.prof_enable(True, serial=True)
        for _ in range(args.steps):
            for g, b in batches:
                b.launch(len(g.pairs), cfg)
                b.collect(len(g.pairs))
        for c in ctxs:
            for k, (n, ms) in c.prof_stats().items():
                n0, ms0 = solo.get(k, (0, 0.0))
                solo[k] = (n0 + n, ms0 + ms)
            c.prof_enable(False)
    px0 = sum(len(g.pairs) * g.width * g.height for g in grids)  # scale-0 pixels per step
    px0_first = sum(len({r for r, _ in g.pairs}) * g.width * g.height for g in grids)  # ... of each reference's first pair

    def tail_pixels(w, h):  # pyramid levels 1..5 (ceil halving; a level exists while both sides are >= 8)
        n, lv = 0, 1
        while lv < 6:
            w, h = (w + 1) // 2, (h + 1) // 2
            if min(w, h) < 8:
                break
            n += w * h
            lv += 1
        return n

    px_tail = sum(len(g.pairs) * tail_pixels(g.width, g.height) for g in grids)
    px_tail_first = sum(len({r for r, _ in g.pairs}) * tail_pixels(g.width, g.height) for g in grids)
    # whole SSIMULACRA2: 210 B per scale-0 pixel of an uncached pair, minus the two shared streams of both passes
    ssim2_step_bytes = SSIM2_BYTES_PER_PX0_TOTAL * px0 - 2 * 2 * SSIM2_STREAM_BYTES * ((px0 - px0_first) + (px_tail - px_tail_first))
    step_bytes = (ssim2_step_bytes - SSIM2_BYTES_PER_PX0_TOTAL * px0 if cfg.ssimulacra2 else 0.0) + px0 * (sum(b for m, b in METRIC_BYTES_PER_PX0.items() if getattr(cfg, m)) + (6.0 if cfg.xyb_roundtrip else 0.0))
    if rank == 0 and args.config != 2:
        # the other configs run several metrics' kernel chains side by side: quote the whole step against the HBM peak
        roofline = {
            "bound": "hbm", "kernel": None, "achieved": round(step_bytes / (ms_per_step * 1e-3) / 1e9, 1), "peak": HBM_PEAK_GBPS,
            "unit": "GB/s", "frac": round(step_bytes / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4), "traffic": None,
            "algorithmic_bytes_per_step": step_bytes,
            "note": "whole step (all enabled metrics, uncached-pair byte counts of BASELINE.md §4) over the timed step; "
                    "per-kernel rooflines are quoted on the headline config only",
        }
    elif rank == 0 and kernels:
        total_ms = sum(ms for _, ms in kernels.values())

        # SURVEY.md §8(d) algorithmic bytes per step of each kernel (R1-R6): a blur pass moves its blurred streams (12 B
        # each: three per pair + the two reference-only ones per reference) plus its inputs (6 B of u8 at level 0, 24 B of
        # f32 above); the front end reads/writes 6+6 and 24+6
        pass_l0 = 6.0 * px0 + SSIM2_STREAM_BYTES * (3 * px0 + 2 * px0_first)
        pass_tail = 24.0 * px_tail + SSIM2_STREAM_BYTES * (3 * px_tail + 2 * px_tail_first)
        alg_bytes = {
            "ssim2_hblur_L0": pass_l0, "ssim2_vblur_ssim_L0": pass_l0,
            "ssim2_hblur_L1-5": pass_tail, "ssim2_vblur_ssim_L1-5": pass_tail,
            "ssim2_prep_u8": 12.0 * px0, "ssim2_prep": 30.0 * px_tail,
        }
        # dominant kernel = the one that moves the largest share of the step's algorithmic bytes (ties: the slower)
        name, (launches, ms) = max(((k, v) for k, v in kernels.items() if k in alg_bytes),
                                   key=lambda kv: (alg_bytes[kv[0]], kv[1][1]), default=max(kernels.items(), key=lambda kv: kv[1][1]))
        n_launch_per_step = launches / args.steps
        avg_s = ms / launches * 1e-3
        bytes_per_launch = alg_bytes.get(name, 0.0) / n_launch_per_step
        achieved = bytes_per_launch / avg_s / 1e9
        traffic = None  # HBM bytes per launch from the PMC counters (profiles/traffic_r01.json, separate --pmc passes)
        tpath = os.path.join(ROOT, "profiles", "traffic_r01.json")
        if os.path.exists(tpath):
            with open(tpath) as f:
                rec = json.load(f).get(name)
            if rec:
                traffic = rec["bytes_per_scale0_pixel"] * px0 / n_launch_per_step
        roofline = {
            "bound": "hbm",
            "kernel": name,
            "achieved": round(achieved, 1),
            "peak": HBM_PEAK_GBPS,
            "unit": "GB/s",
            "frac": round(achieved / HBM_PEAK_GBPS, 4),
            "traffic": traffic,
            "algorithmic_bytes_per_launch": bytes_per_launch,
            "avg_launch_ms": round(avg_s * 1e3, 4),
            "launches": launches,
            "timing": "HIP events on the launch stream, in the timed region; kernels of other pyramid levels, of the other "
                      "shape bucket and of the next step run concurrently on other streams, so this duration is the "
                      "share of the GPU the launch gets (solo_* = the same launch alone on the GPU; pipeline_* = all "
                      "kernels of the step over the step time)",
            "dominant_by": "largest share of the step's algorithmic bytes (%.0f %%)" % (100.0 * alg_bytes.get(name, 0.0) / ssim2_step_bytes),
            "bytes_model": "12 B per blurred stream; a, a*a streams once per reference, the other three per pair; "
                           "an uncached pair would be %.0f B per launch" % (SSIM2_PASS_BYTES_L0 * px0 / n_launch_per_step),
            # the whole metric against the same peak: 210 B per scale-0 pixel of an uncached pair less the shared streams
            "pipeline_achieved": round(ssim2_step_bytes / (ms_per_step * 1e-3) / 1e9, 1),
            "pipeline_frac": round(ssim2_step_bytes / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4),
            "pipeline_bytes_per_px0": round(ssim2_step_bytes / px0, 1),
            # the same two fractions on SURVEY.md §8(d)'s uncached-pair counts (66 B per pass, 210 B per pixel), for comparison
            "uncached_pair_model": {
                "frac": round(SSIM2_PASS_BYTES_L0 * px0 / n_launch_per_step / avg_s / 1e9 / HBM_PEAK_GBPS, 4) if name.endswith("_L0") else None,
                "pipeline_frac": round(SSIM2_BYTES_PER_PX0_TOTAL * px0 / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4),
            },
        }
        if solo:
            sn, sms = solo[name]
            roofline["solo_avg_launch_ms"] = round(sms / sn, 4)
            roofline["solo_frac"] = round(bytes_per_launch / (sms / sn * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4)
            roofline["solo_kernel_ms_per_step"] = round(sum(m for _, m in solo.values()) / args.steps, 4)

    # ---- CPU baseline: the C oracle on this host's cores (rank 0, N = 1 only) -------------------
    cpu_baseline = None
    max_dev = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline and args.config == 2:
        from oracle import oracle as O

        O.build()
        items = [(g, ri, t) for g in grids for (ri, t) in g.pairs]

        def one(it):
            g, ri, t = it
            return O.ssimulacra2(g.references[ri], t, g.width, g.height, 1)

        n1 = min(len(items), 12)
        t1 = time.perf_counter()
        cpu_scores_1 = [one(it) for it in items[:n1]]
        dt1 = time.perf_counter() - t1
        mp1 = sum(g.width * g.height for g, _, _ in items[:n1]) / 1e6
        cores = os.cpu_count() or 1
        try:
            cores = len(os.sched_getaffinity(0))
        except Exception:
            pass
        tN = time.perf_counter()
        with ThreadPoolExecutor(cores) as ex:  # ctypes releases the GIL: item-level parallelism,
            cpu_scores = list(ex.map(one, items))  # mirrors images.par_iter() (full_comparison.rs:319-328)
        dtN = time.perf_counter() - tN
        cpu_baseline = {
            "value": round(mp_per_step / dtN, 3),
            "unit": "MP/s",
            "cores": cores,
            "kind": "port",
            "sample": f"all {len(items)} pairs of the workload, once, {cores} threads over items "
                      f"(C restatement of the metric; the Rust crates cannot be built offline)",
            "value_1thread": round(mp1 / dt1, 3),
            "sample_1thread": f"first {n1} pairs, 1 thread",
        }
        gpu_scores = [s.ssimulacra2 for sc in scores for s in sc]
        max_dev = max(abs(a - b) / max(1.0, abs(b)) for a, b in zip(gpu_scores, cpu_scores))

    if rank == 0:
        line = {
            "metric": "metric MP/s (SSIMULACRA2+DSSIM+Butteraugli) at 1/2/4/8 GPU; HBM-roofline %",
            "value": round(value, 2),
            "unit": "MP/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": workload + (" [--quick subset]" if args.quick else ""),
                "pairs_per_gpu_step": pairs_per_step,
                "megapixels_per_gpu_step": round(mp_per_step, 3),
                "metrics": [m for m in ("dssim", "ssimulacra2", "butteraugli", "psnr") if getattr(cfg, m)],
                "sharding": "by reference image, one process + one HIP stream per GPU, no collective",
                "inputs": "resident in HBM (uploaded before the timed region)",
                "batches_in_flight": depth,
                "hip_env": {"GPU_MAX_HW_QUEUES": os.environ.get("GPU_MAX_HW_QUEUES")},
            },
            "roofline": roofline,
            "cpu_baseline": cpu_baseline,
            "max_rel_dev_vs_oracle": max_dev,
            "kernels_ms_per_step": {k: round(ms / args.steps, 4) for k, (n, ms) in sorted(kernels.items())},
            "solo_kernels_ms_per_step": {k: round(ms / args.steps, 4) for k, (n, ms) in sorted(solo.items())} or None,
        }
        print(json.dumps(line), flush=True)

    for cs, bs in sets:
        for _, b in bs:
            b.close()
        for c in {id(c): c for c in cs}.values():
            c.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
