#!/usr/bin/env python3
"""bench.py — BASELINE.json's metric on the MI355X hot path.

    python bench.py --gpus N --steps K --warmup W
    (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

Default workload: the metric AS NAMED — "metric MP/s (SSIMULACRA2+DSSIM+Butteraugli)" — on BASELINE.json configs[1]'s
grid: 24 synthetic references in Kodak's shapes (18 of 768x512, 6 of 512x768) x q in {75, 85, 95} = 72
(reference, distorted) pairs per GPU, every pair scored by all three perceptual metrics in one launch
(the call shape of crates/codec-compare/src/full_comparison.rs:149-177).  Inputs are uploaded once and are resident
in HBM when the timed region starts.  A step = one pass of the hot path over the rank's whole grid, scores returned to
the host.  value = reference pixels x (pair, metric) evaluations / wall time, plus a per-metric
breakdown (each metric alone on the same grid).

--config 2..5 select BASELINE.json's other configs (2 = configs[1] as written: SSIMULACRA2 only; 3 = 4K Butteraugli;
4 = CID22 x 8 qualities, SSIMULACRA2 + DSSIM; 5 = codec-iter dense sweep, all metrics, XYB on/off).

N > 1: ONE global grid is partitioned over the ranks by codec-eval_amd/sharding.py (by reference image; config 5
falls back to (image, codec-config) units when that balances better), no data-path collective, the scores are
gathered to rank 0 in global (image, variant, quality) order and a sample of them is recomputed there.  Default
workload: the global grid is N Kodak-24 sets (24 N references), so per-GPU work is fixed => "scaling": "weak";
configs 4 and 5 are FIXED grids => "scaling": "strong".  torch.distributed is control plane only (barriers,
max-over-ranks of the time, the gather of the tiny score lists).

Extra JSON objects: "roofline" (the dominant kernel: algorithmic bytes per launch over its HIP-event duration, alone
on the GPU = `frac`, and in the timed region = `in_region_*`; every kernel's row in `kernels`), "per_metric",
"end_to_end" (page-locked host buffers in, scores out, uploads included) and "cpu_baseline" (the C oracle — a scalar
restatement, not the Rust crates — on this host's cores; rank 0, N = 1 only).
"""
from __future__ import annotations

import argparse
import importlib
import json
import os
import sys
import time

# before anything initialises HIP (torch does): see codec-eval_amd/__init__.py
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

METRIC_NAMES = ("dssim", "ssimulacra2", "butteraugli", "psnr")


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", type=int, default=0, choices=(0, 2, 3, 4, 5),
                    help="0 (default): BASELINE configs[1]'s Kodak grid scored by SSIMULACRA2 + DSSIM + Butteraugli (the metric as "
                         "named); 2..5: BASELINE.json configs[1..4] as written (1-based), scaled with --refs")
    ap.add_argument("--refs", type=int, default=0, help="override the number of reference images (configs 3-5)")
    ap.add_argument("--quick", action="store_true", help="6 references instead of 24 per Kodak set (smoke runs)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-solo", action="store_true", help="skip the solo pass (no per-kernel table; roofline from in-region events only)")
    ap.add_argument("--no-events", action="store_true", help="no HIP events in the timed region")
    ap.add_argument("--no-per-metric", action="store_true")
    ap.add_argument("--no-end-to-end", action="store_true")
    ap.add_argument("--all-events", action="store_true", help="events around every kernel in the timed region, not only the dominant one")
    ap.add_argument("--serial", action="store_true",
                    help="timed region with every launch on one stream, one kernel at a time (for rocprofv3 traces whose per-kernel "
                         "durations are the kernels' own, not their share of an overlapped schedule)")
    ap.add_argument("--one-shape", action="store_true",
                    help="experiment: all 24 references of a Kodak set in the 768x512 shape (one bucket, same pixel count) - the upper "
                         "bound of what a merged two-bucket launch could gain")
    ap.add_argument("--depth", type=int, default=0, help="batches in flight per shape bucket (default 2: step k+1 is launched before step k's scores are collected)")
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

    wl = importlib.import_module("codec-eval_amd.workloads")
    sh = importlib.import_module("codec-eval_amd.sharding")
    rf = importlib.import_module("codec-eval_amd.roofline")

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (torch.cuda.is_available() is False)")
    # CE_BENCH_SHARE_DEVICE=1 (rehearsal of the N > 1 path on a box with fewer GPUs than ranks): ranks share devices
    # and the rendezvous uses gloo.  Never set by the driver.
    share = os.environ.get("CE_BENCH_SHARE_DEVICE") == "1"
    if share:
        local_rank = local_rank % torch.cuda.device_count()
    # 8 ranks x 16 hardware queues on ONE device is the oversubscribed regime; with one rank per GPU each
    # process owns its device's queues, so 16 per rank is the same setting as N = 1
    torch.cuda.set_device(local_rank)
    dist = None
    gloo = None
    if world > 1:
        import torch.distributed as dist  # noqa: F811

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if share:
            dist.init_process_group("gloo")
            gloo = dist.group.WORLD
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
            gloo = dist.new_group(backend="gloo")  # host-side gather of the score lists (python objects)

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    # ---- the global grid and this rank's shard of it (synthetic, seeded by GLOBAL index) -----------------------
    cfg_id = args.config
    qualities = (75, 85, 95)  # codec-iter "quick" preset, crates/codec-iter/src/main.rs:197
    scaling = "weak"
    partition_mode = "reference"
    launches_cfg = []  # [(grid, MetricConfig)] of this rank: one resident batch each
    if cfg_id in (0, 2):
        shapes_all = wl.kodak_corpus_shapes(world) if not args.one_shape else [(768, 512)] * (24 * world)
        ids = [g for g in range(24 * world) if not args.quick or g % 24 in (0, 1, 2, 3, 4, 18)]  # --quick: 5 + 1 per set
        shapes = [shapes_all[g] for g in ids]
        pix = [w * h for (w, h) in shapes]
        mode, units, loads = sh.plan_partition(pix, [len(qualities)] * len(pix), 1, world)
        mine = sorted({ids[i] for i, _ in units[rank]})
        grids = (wl.kodak_corpus_shard(mine, qualities, seed0=1000) if not args.one_shape else
                 [wl._grid("kodak-768x512", 768, 512, 0, 1000, qualities, only=mine)])
        cfg = ce.MetricConfig.perceptual() if cfg_id == 0 else ce.MetricConfig.ssimulacra2_only()
        launches_cfg = [(g, cfg) for g in grids]
        n_global_refs = len(ids)
        what = "SSIMULACRA2 + DSSIM + Butteraugli on every pair" if cfg_id == 0 else "SSIMULACRA2 only"
        workload = (f"BASELINE configs[1] grid: Kodak-24 x 3 quality levels (q75/85/95; 768x512 x18 + 512x768 x6 buffers) per GPU, {what}"
                    + (f"; global grid = {world} Kodak-24 sets partitioned by reference" if world > 1 else ""))
    elif cfg_id == 3:
        n = args.refs or 4
        pix = [3840 * 2160] * (n * world)
        mode, units, loads = sh.plan_partition(pix, [1] * len(pix), 1, world)
        mine = sorted({i for i, _ in units[rank]})
        cfg = ce.MetricConfig(butteraugli=True)
        launches_cfg = [(wl.uhd_pairs(n * world, seed0=2000, only=mine), cfg)]
        n_global_refs = n * world
        workload = f"BASELINE configs[2]: {n} synthetic 3840x2160 pairs per GPU (the config names 16: --refs 16), Butteraugli (max-norm + 3-norm)"
    elif cfg_id == 4:
        n = args.refs or 250
        scaling = "strong"
        pix = [512 * 512] * n
        mode, units, loads = sh.plan_partition(pix, [len(wl.STANDARD_QUALITIES)] * n, 1, world)
        mine = sorted({i for i, _ in units[rank]})
        cfg = ce.MetricConfig(ssimulacra2=True, dssim=True)
        launches_cfg = [(wl.cid22_like(n, seed0=3000, only=mine), cfg)]
        n_global_refs = n
        workload = f"BASELINE configs[3]: {n} CID22-shaped 512x512 refs x 8 qualities = {8 * n} pairs (fixed grid, partitioned by reference), SSIMULACRA2 + DSSIM"
    else:
        n = args.refs or 15
        scaling = "strong"
        pix = [512 * 512] * n
        # codec configs of crates/codec-iter/src/main.rs:474-499: {4:4:4, 4:2:0} x {XYB off, on}; variant v = 2 * xyb + s420
        mode, units, loads = sh.plan_partition(pix, [4 * len(wl.DENSE_QUALITIES)] * n, 4, world)
        for xyb in (0, 1):
            u = sorted((i, v & 1) for i, v in units[rank] if (v >> 1) == xyb)
            if u:
                c = ce.MetricConfig.all()
                launches_cfg.append((wl.codec_iter_dense(n, seed0=4000, units=u), c.with_xyb_roundtrip() if xyb else c))
        cfg = ce.MetricConfig.all()
        n_global_refs = n
        workload = (f"BASELINE configs[4]: {n} 512x512 refs x 25 qualities x {{4:4:4, 4:2:0}} x {{XYB off, on}} = {100 * n} pairs "
                    f"(fixed grid), PSNR + SSIMULACRA2 + DSSIM + Butteraugli, XYB roundtrip on the reference for the XYB-on half")
    partition_mode = mode
    metrics_on = [m for m in METRIC_NAMES if getattr(cfg, m)]
    # (pair, metric) evaluations per pair: the three perceptual metrics BASELINE's metric names; PSNR rides along for free
    # (6 B/px) and is NOT counted, except in a PSNR-only configuration
    n_eval_metrics = len([m for m in metrics_on if m != "psnr"]) or 1

    # ---- resident batches ----------------------------------------------------------------------------------------
    # One context (= one HIP stream family) per batch, so the buckets' kernel chains overlap on the GPU.  depth > 1 keeps
    # that many sets of batches in flight (step k is launched before step k-1's scores are collected, the way a session
    # streams a corpus larger than one batch); every timed step's scores are still collected inside the timed region.
    depth = 1 if args.serial else (args.depth if args.depth > 0 else 2)
    sets = []
    for _ in range(depth):
        bs = []
        for g, c in launches_cfg:
            ctx = ce.Context(local_rank)
            b = ce.Batch(ctx, g.width, g.height, len(g.references), len(g.pairs))
            for i, r in enumerate(g.references):
                b.set_reference(i, r)
            for k, (ri, t) in enumerate(g.pairs):
                b.set_test(k, ri, t)
            bs.append((g, c, ctx, b))
        sets.append(bs)
    all_ctxs = [ctx for bs in sets for (_, _, ctx, _) in bs]
    pairs_per_step = sum(len(g.pairs) for g, _ in launches_cfg)
    mp_per_step = sum(g.megapixels for g, _ in launches_cfg)

    def launch(k, only_cfg=None):
        for g, c, _, b in sets[k % depth]:
            b.launch(len(g.pairs), only_cfg or c)

    def collect(k):
        return [b.collect(len(g.pairs)) for g, _, _, b in sets[k % depth]]

    def run_steps(n, only_cfg=None):
        out = None
        if args.serial:  # one batch at a time, collected before the next is launched: nothing overlaps, across contexts either
            for k in range(n):
                out = []
                for g, c, _, b in sets[0]:
                    b.launch(len(g.pairs), only_cfg or c)
                    out.append(b.collect(len(g.pairs)))
            return out
        for k in range(n):
            launch(k, only_cfg)
            if k >= depth - 1:
                out = collect(k - (depth - 1))
        for k in range(max(0, n - (depth - 1)), n):
            out = collect(k)
        return out

    def prof_all(on, serial=False, flt=""):
        for c in all_ctxs:
            c.prof_reset()
            c.prof_filter(flt)
            c.prof_enable(on, serial=serial)

    def gather_stats(ctxs=None):
        acc = {}
        for c in ctxs or all_ctxs:
            for k, (n, ms) in c.prof_stats().items():
                n0, ms0 = acc.get(k, (0, 0.0))
                acc[k] = (n0 + n, ms0 + ms)
        return acc

    # one untimed pass over every set of batches first: lazy device allocations and the host-built work lists are part
    # of setting a batch up, not of a step
    for bs in sets:
        for g, c, _, b in bs:
            b.launch(len(g.pairs), c)
            b.collect(len(g.pairs))

    # ---- algorithmic bytes of every kernel for one step of this rank's grid (codec-eval_amd/roofline.py) ----------
    alg = {}
    for g, c in launches_cfg:
        part = rf.step_bytes([rf.Bucket(g.width, g.height, len(g.references), len(g.pairs))],
                             [m for m in METRIC_NAMES if getattr(c, m)], c.xyb_roundtrip)
        for k, v in part.items():
            alg[k] = alg.get(k, 0.0) + v
    step_alg_bytes = sum(alg.values())

    # ---- solo pass (untimed): every kernel alone on the GPU, one stream, HIP events around each launch ------------
    solo = {}
    solo_steps = 0
    if not args.no_solo:
        solo_steps = max(3, min(10, args.steps))
        prof_all(True, serial=True)
        for _ in range(solo_steps):
            for g, c, _, b in sets[0]:
                b.launch(len(g.pairs), c)
                b.collect(len(g.pairs))
        solo = gather_stats([ctx for _, _, ctx, _ in sets[0]])
        prof_all(False)
    # dominant kernel = the one the GPU spends the most time in (solo time per step), among the kernels that move data
    dominant = None
    if solo:
        dominant = max((k for k in solo if k in alg), key=lambda k: solo[k][1], default=None)

    # ---- timed region -----------------------------------------------------------------------------------------------
    events = not args.no_events
    if args.serial:
        prof_all(True, serial=True)
    elif events:
        prof_all(True, serial=False, flt="" if (args.all_events or dominant is None) else "=" + dominant)
    run_steps(args.warmup)
    for c in all_ctxs:
        c.prof_reset()

    barrier()
    t0 = time.perf_counter()
    scores = run_steps(args.steps)
    barrier()
    elapsed = time.perf_counter() - t0
    local_elapsed = elapsed
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if share else "cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    in_region = gather_stats() if (events or args.serial) else {}
    prof_all(False)

    # MP/s = reference pixels x (pair, metric) evaluations / wall time
    all_mp = [mp_per_step]
    all_pairs = [pairs_per_step]
    if dist is not None:
        info = [None] * world
        dist.all_gather_object(info, (mp_per_step, pairs_per_step, local_elapsed), group=gloo)
        all_mp = [x[0] for x in info]
        all_pairs = [x[1] for x in info]
        all_elapsed = [x[2] for x in info]
    else:
        all_elapsed = [local_elapsed]
    total_mp = sum(all_mp) * n_eval_metrics * args.steps
    value = total_mp / elapsed
    ms_per_step = elapsed / args.steps * 1e3

    # ---- gather the scores of the last step in global order; rank 0 recomputes a sample ---------------------------
    shard = None
    if world > 1:
        local = []
        for (g, c, _, _), sc in zip(sets[(args.steps - 1) % depth], scores):
            for pid, s in zip(g.pair_ids, sc):
                key = (pid[0], pid[1] + (2 if c.xyb_roundtrip and cfg_id == 5 else 0), pid[2])
                local.append((key, (s.dssim, s.ssimulacra2, s.butteraugli, s.psnr)))
        merged = [None] * world if rank == 0 else None
        dist.gather_object(local, merged, dst=0, group=gloo)
        if rank == 0:
            flat = sorted((t for part in merged for t in part), key=lambda t: t[0])
            keys = [k for k, _ in flat]
            assert len(set(keys)) == len(keys) == sum(all_pairs), "the shards must tile the global grid exactly once"
            # recompute one item of every other rank's shard here (single-pair call; batch == single is bit-exact)
            foreign = {}
            for r in range(1, world):
                for k, v in merged[r][:1]:
                    foreign[k] = v
            max_diff = 0.0
            with ce.Context(local_rank) as cx:
                for (gi, v, qi), want in foreign.items():
                    if cfg_id in (0, 2):
                        w_, h_ = wl.kodak_corpus_shapes(world)[gi]
                        ref = wl.make_reference(w_, h_, 1000 + gi)
                        test = wl.distort(ref, qualities[qi])
                        c1 = cfg
                    elif cfg_id == 3:
                        w_, h_ = 3840, 2160
                        ref = wl.make_reference(w_, h_, 2000 + gi)
                        test = wl.distort(ref, 85)
                        c1 = cfg
                    elif cfg_id == 4:
                        w_, h_ = 512, 512
                        ref = wl.make_reference(w_, h_, 3000 + gi)
                        test = wl.distort(ref, wl.STANDARD_QUALITIES[qi])
                        c1 = cfg
                    else:
                        w_, h_ = 512, 512
                        ref = wl.make_reference(w_, h_, 4000 + gi)
                        test = wl.distort(ref, wl.DENSE_QUALITIES[qi], bool(v & 1))
                        c1 = cfg.with_xyb_roundtrip() if v >> 1 else cfg
                    m = cx.calculate_metrics(ref, test, w_, h_, c1)
                    got = (m.dssim or 0.0, m.ssimulacra2 or 0.0, m.butteraugli or 0.0, m.psnr or 0.0)
                    max_diff = max(max_diff, max(abs(a - b) for a, b in zip(got, want)))
            shard = {
                "partition": partition_mode, "pairs_per_rank": all_pairs, "megapixels_per_rank": [round(x, 3) for x in all_mp],
                "imbalance_max_over_mean": round(sh.imbalance(all_mp), 4), "seconds_per_rank": [round(x, 4) for x in all_elapsed],
                "gathered_scores": len(flat), "recomputed_on_rank0": len(foreign), "recomputed_max_abs_diff": max_diff,
            }
            assert max_diff == 0.0, f"a gathered score differs from its recomputation on rank 0 ({max_diff})"

    # ---- roofline -----------------------------------------------------------------------------------------------------
    roofline = None
    if rank == 0:
        peak = rf.HBM_PEAK_GBPS
        kernels_tbl = {}
        for k in sorted(set(solo) | set(in_region)):
            row = {"metric": rf.metric_of(k)}
            if k in solo:
                n, ms = solo[k]
                row.update(launches_per_step=round(n / solo_steps, 2), solo_ms_per_step=round(ms / solo_steps, 4))
                if k in alg and ms > 0:
                    row.update(alg_mb_per_step=round(alg[k] / 1e6, 2), solo_gbps=round(alg[k] * solo_steps / (ms * 1e-3) / 1e9, 1),
                               solo_frac=round(alg[k] * solo_steps / (ms * 1e-3) / 1e9 / peak, 4))
            if k in in_region:
                n, ms = in_region[k]
                row.update(in_region_ms_per_step=round(ms / args.steps, 4))
            kernels_tbl[k] = row
        name = dominant or (max((k for k in in_region if k in alg), key=lambda k: in_region[k][1], default=None))
        if name is not None:
            traffic = None  # HBM bytes per launch from the PMC counters (profiles/traffic_r02.json; separate --pmc passes)
            tpath = os.path.join(ROOT, "profiles", "traffic_r02.json")
            px0 = sum(len(g.pairs) * g.width * g.height for g, _ in launches_cfg)
            lps = (solo[name][0] / solo_steps) if name in solo else (in_region[name][0] / args.steps)
            if os.path.exists(tpath):
                with open(tpath) as f:
                    rec = json.load(f).get(name)
                if rec:
                    traffic = rec["bytes_per_scale0_pixel"] * px0 / lps
            bytes_per_launch = alg[name] / lps
            roofline = {"bound": "hbm", "kernel": name, "peak": peak, "unit": "GB/s", "traffic": traffic,
                        "algorithmic_bytes_per_launch": bytes_per_launch, "launches_per_step": lps}
            if name in solo:
                n, ms = solo[name]
                avg = ms / n * 1e-3
                roofline.update(achieved=round(bytes_per_launch / avg / 1e9, 1), frac=round(bytes_per_launch / avg / 1e9 / peak, 4),
                                avg_launch_ms=round(avg * 1e3, 4), launches=n,
                                frac_basis="solo: the launch alone on the GPU, HIP events on its stream, untimed pass of this same run "
                                           "(in the timed region the metrics' chains, shape buckets and steps overlap, so a launch's "
                                           "duration there is its share of the GPU - see in_region_*)")
            if name in in_region:
                n, ms = in_region[name]
                avg = ms / n * 1e-3
                roofline.update(in_region_avg_launch_ms=round(avg * 1e3, 4), in_region_launches=n,
                                in_region_achieved=round(bytes_per_launch / avg / 1e9, 1),
                                in_region_frac=round(bytes_per_launch / avg / 1e9 / peak, 4))
                if "frac" not in roofline:
                    roofline.update(achieved=roofline["in_region_achieved"], frac=roofline["in_region_frac"], avg_launch_ms=roofline["in_region_avg_launch_ms"],
                                    launches=n, frac_basis="in region (no solo pass)")
                elif name in solo:
                    roofline["in_region_share"] = round((solo[name][1] / solo[name][0]) / (ms / n), 3)
            solo_total = sum(ms for _, ms in solo.values()) / solo_steps if solo else None
            roofline.update(
                dominant_by="largest solo time per step (%.0f %% of the step's kernel time)" % (100.0 * solo[name][1] / max(sum(ms for _, ms in solo.values()), 1e-12)) if name in solo else "largest in-region time",
                # all kernels of the step against the same peak
                pipeline_algorithmic_bytes_per_step=step_alg_bytes,
                pipeline_achieved=round(step_alg_bytes / (ms_per_step * 1e-3) / 1e9, 1),
                pipeline_frac=round(step_alg_bytes / (ms_per_step * 1e-3) / 1e9 / peak, 4),
                solo_kernel_ms_per_step=round(solo_total, 4) if solo_total else None,
                bytes_model="codec-eval_amd/roofline.py: compulsory bytes of each kernel (inputs once, outputs once; reference-side "
                            "inputs once per reference); SURVEY.md §8(d)'s uncached-pair stage sums would be "
                            + ", ".join(f"{m} {rf.uncached_pair_bytes_per_px0(m):.0f}" for m in metrics_on) + " B per scale-0 pixel",
                kernels=kernels_tbl,
            )

    # ---- per-metric breakdown: each metric alone on the same resident grid (rank 0 prints; every rank runs) ---------
    per_metric = None
    if not args.no_per_metric and len(metrics_on) > 1 and world == 1:
        per_metric = {}
        for m in metrics_on:
            one = ce.MetricConfig(**{m: True})
            if cfg.xyb_roundtrip:
                one = one.with_xyb_roundtrip()
            n = max(3, min(args.steps, 20))
            run_steps(2, one)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            run_steps(n, one)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t1
            per_metric[m] = {"value": round(mp_per_step * n / dt, 2), "unit": "MP/s", "ms_per_step": round(dt / n * 1e3, 4), "steps": n}
        # restore the combined working state (scores of the combined config) for the checks below
        scores = run_steps(1)

    # ---- end to end: page-locked host buffers in, scores out (uploads inside the timing) -----------------------------
    end_to_end = None
    if not args.no_end_to_end and world == 1 and rank == 0:
        items = []
        keep = []
        for g, c in launches_cfg:
            refs_p = []
            for r in g.references:
                t = torch.empty(r.size, dtype=torch.uint8).pin_memory()
                t.numpy()[:] = r.reshape(-1)
                refs_p.append(t)
            for ri, tt in g.pairs:
                t = torch.empty(tt.size, dtype=torch.uint8).pin_memory()
                t.numpy()[:] = tt.reshape(-1)
                keep.append(t)
                items.append((c, (refs_p[ri].numpy(), t.numpy(), g.width, g.height)))
            keep.extend(refs_p)
        by_cfg = {}
        for c, it in items:
            by_cfg.setdefault((c.mask, c.flags), (c, []))[1].append(it)
        with ce.Context(local_rank) as cx:
            def e2e_step():
                for c, its in by_cfg.values():
                    cx.eval_batch(its, c)
            e2e_step()
            n = max(2, min(args.steps, 10))
            t1 = time.perf_counter()
            for _ in range(n):
                e2e_step()
            dt = time.perf_counter() - t1
        end_to_end = {"value": round(mp_per_step * n_eval_metrics * n / dt, 2), "unit": "MP/s", "ms_per_step": round(dt / n * 1e3, 3), "steps": n,
                      "route": "ce_eval_batch: page-locked host RGB8 in -> H2D on the upload stream (chunked, overlapped with the "
                               "kernels of the previous chunk) -> kernels -> scores on the host; 6 B/px over PCIe per pair "
                               "(3 B/px for the pairs that share an already uploaded reference)",
                      "h2d_megabytes_per_step": round(sum((len(g.references) + len(g.pairs)) * g.width * g.height * 3 for g, _ in launches_cfg) / 1e6, 1)}
        del keep

    # ---- CPU baseline: the C oracle on this host's cores (rank 0, N = 1 only) ------------------------------------------
    cpu_baseline = None
    max_dev = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle import oracle as O

        O.build()
        fn = {"ssimulacra2": lambda r, t, w, h: O.ssimulacra2(r, t, w, h, 1), "dssim": O.dssim,
              "butteraugli": lambda r, t, w, h: O.butteraugli(r, t, w, h)[0], "psnr": O.psnr}
        flat_pairs = [(g, c, ri, t) for g, c in launches_cfg for (ri, t) in g.pairs]
        # bounded sample: at most SAMPLE_MP megapixels of pairs (evenly spaced over the grid), every enabled metric on each
        SAMPLE_MP = 30.0
        mp_pair = [g.width * g.height / 1e6 for g, _, _, _ in flat_pairs]
        stride = max(1, int(np.ceil(sum(mp_pair) / SAMPLE_MP)))
        sample_idx = list(range(0, len(flat_pairs), stride))
        cores = os.cpu_count() or 1
        try:
            cores = len(os.sched_getaffinity(0))
        except Exception:
            pass
        # work items = (pair, metric); the slow metric first so the pool drains evenly
        cost = {"butteraugli": 3, "dssim": 2, "ssimulacra2": 1, "psnr": 0}
        work = sorted(((i, m) for i in sample_idx for m in metrics_on), key=lambda im: -cost[im[1]])

        def one(im):
            i, m = im
            g, c, ri, t = flat_pairs[i]
            ref = g.references[ri]
            if c.xyb_roundtrip:
                ref = O.xyb_roundtrip(ref, g.width, g.height)
            return fn[m](ref, t, g.width, g.height)

        # 1 thread: the first pair of the sample, every metric
        one_thread = {}
        for m in metrics_on:
            t1 = time.perf_counter()
            one((sample_idx[0], m))
            one_thread[m] = round(mp_pair[sample_idx[0]] / (time.perf_counter() - t1), 3)
        threads = max(1, min(cores, len(work)))
        with ThreadPoolExecutor(threads) as ex:  # ctypes releases the GIL: item-level parallelism,
            list(ex.map(one, work[:threads]))       # warm pass (page in the library on every thread)
            tN = time.perf_counter()
            res = list(ex.map(one, work))           # mirrors images.par_iter() (full_comparison.rs:319-328)
            dtN = time.perf_counter() - tN
        sample_mp = sum(mp_pair[i] for i in sample_idx)
        cpu_baseline = {
            "value": round(sample_mp * len(metrics_on) / dtN, 3), "unit": "MP/s", "cores": threads, "kind": "port",
            "sample": f"{len(sample_idx)} of the {len(flat_pairs)} pairs (every {stride}th), {'+'.join(metrics_on)} on each = {len(work)} "
                      f"(pair, metric) items over a pool of {threads} threads (host has {cores} cores; pool = min(cores, items)), "
                      f"one warm pass first; scalar C restatement of the metrics (oracle/*.c), NOT the Rust crates' SIMD code",
            "seconds": round(dtN, 2), "host_cores": cores,
            "value_1thread_per_metric": one_thread,
        }
        # the device against the same oracle values, on the sampled pairs
        got = [s for sc in scores for s in sc]
        want = {}
        for (i, m), v in zip(work, res):
            want[(i, m)] = v
        max_dev = {}
        for m in metrics_on:
            floor = {"ssimulacra2": 1.0, "dssim": 1e-6, "butteraugli": 1e-3, "psnr": 1.0}[m]
            dev = 0.0
            for i in sample_idx:
                a, b = getattr(got[i], m), want[(i, m)]
                if a == b:
                    continue
                dev = max(dev, abs(a - b) / max(abs(b), floor))
            max_dev[m] = dev

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
            "scaling": scaling,
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": workload + (" [--quick subset]" if args.quick else ""),
                "metrics": metrics_on,
                "pairs_per_step": sum(all_pairs),
                "megapixels_per_step": round(sum(all_mp), 3),
                "metric_evaluations_per_pair": n_eval_metrics,
                "sharding": f"global grid of {n_global_refs} references partitioned by {partition_mode}, one process per GPU, no collective on the data path",
                "inputs": "resident in HBM (uploaded before the timed region)",
                "batches_in_flight": depth,
                "schedule": "serial (one stream, one kernel at a time)" if args.serial else
                            ("shape buckets and in-flight steps overlap on their own HIP streams; a batch's metric chains run "
                             + ("as CE_METRIC_STREAMS=%s says" % os.environ["CE_METRIC_STREAMS"] if os.environ.get("CE_METRIC_STREAMS") else
                                "back to back (side by side only for a batch of <= 4 MP of pairs or one launched while nothing else is "
                                "in flight on the device)")),
                "hip_env": {"GPU_MAX_HW_QUEUES": os.environ.get("GPU_MAX_HW_QUEUES")},
            },
            "roofline": roofline,
            "cpu_baseline": cpu_baseline,
            "per_metric": per_metric,
            "end_to_end": end_to_end,
            "shard": shard,
            "max_rel_dev_vs_oracle": max_dev,
        }
        print(json.dumps(line), flush=True)

    for bs in sets:
        for _, _, ctx, b in bs:
            b.close()
            ctx.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
