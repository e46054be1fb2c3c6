"""bench.py's output contract, on the device: one JSON line with the required keys (N = 1), and the N = 2 path
(torch.distributed.run, one process per rank, max-over-ranks time, rank 0 prints) rehearsed with both ranks sharing
this box's single GPU over gloo (CE_BENCH_SHARE_DEVICE=1 - the driver never sets it)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REQUIRED = {"metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
            "dtype", "data", "config", "roofline", "cpu_baseline"}


def _json_lines(out):
    return [json.loads(l) for l in out.splitlines() if l.startswith("{")]


def test_single_gpu_line():
    out = subprocess.run([sys.executable, "bench.py", "--quick", "--steps", "4", "--warmup", "1"], cwd=ROOT, capture_output=True,
                         text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = _json_lines(out.stdout)
    assert len(lines) == 1
    d = lines[0]
    assert REQUIRED <= set(d), REQUIRED - set(d)
    assert d["n_gpus"] == 1 and d["steps"] == 4 and d["warmup"] == 1 and d["unit"] == "MP/s" and d["scaling"] == "weak"
    assert d["higher_is_better"] is True and d["vs_baseline"] is None and d["data"] == "synthetic" and d["dtype"] == "f32"
    assert "workload" in d["config"] and "model" not in d["config"]
    # the line measures the metric it names: all three perceptual metrics on every pair, with a per-metric breakdown
    assert set(d["config"]["metrics"]) == {"ssimulacra2", "dssim", "butteraugli"} and d["config"]["metric_evaluations_per_pair"] == 3
    assert set(d["per_metric"]) == {"ssimulacra2", "dssim", "butteraugli"} and all(v["value"] > 0 for v in d["per_metric"].values())
    # the workload is north_star's sweep (Kodak + CID22 shapes; --quick: 6 x 3 + 8 x 8 pairs), with the Kodak grid alone beside it
    assert d["config"]["pairs_per_step"] == 6 * 3 + 8 * 8 and "CID22" in d["config"]["workload"] and "Kodak" in d["config"]["workload"]
    assert d["kodak_only"]["value"] > 0 and d["kodak_only"]["unit"] == "MP/s"
    # one blocking call per encode (crates/codec-iter/src/gpu.rs:83-109), both shapes, both call forms
    for shape in ("768x512", "512x512"):
        pc = d["per_call"][shape]
        assert 0 < pc["ce_ref_compare_ssimulacra2"] <= pc["ce_ref_compare_all_metrics"] * 1.2 and 0 < pc["ce_eval_pair_all_metrics"] < 50
    r = d["roofline"]
    assert {"bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "algorithmic_bytes_per_launch", "avg_launch_ms"} <= set(r)
    assert r["bound"] in ("hbm", "valu") and r["peak"] == 8000.0 and r["kernel"] in r["kernels"]
    assert (r["bound"] == "valu") == (r["valu_util"] is not None and r["valu_util"] >= 0.6)
    # the step against this round's byte model and against round 2's frozen one
    assert 0.0 < r["pipeline_frac"] <= r["pipeline_frac_r02_model"] < 1.0 and r["traffic_over_algorithmic"] >= 1.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3 and 0.0 < r["frac"] < 1.0
    # frac is recomputable from the line alone: algorithmic bytes per launch / average launch duration / peak
    assert abs(r["algorithmic_bytes_per_launch"] / (r["avg_launch_ms"] * 1e-3) / 1e9 / r["peak"] - r["frac"]) < 2e-3
    assert 0.0 < r["in_region_frac"] <= 1.0 and 0.0 < r["pipeline_frac"] < 1.0
    # the dominant kernel is the one with the largest solo time, and every data-moving kernel has its own row
    rows = {k: v for k, v in r["kernels"].items() if "solo_frac" in v}
    assert r["kernel"] == max(rows, key=lambda k: rows[k]["solo_ms_per_step"])
    assert {"ssim2_vblur_ssim_L0", "dssim_compare", "ba_malta_l2", "ba_blur_h33"} <= set(rows)
    c = d["cpu_baseline"]
    assert {"value", "unit", "cores", "kind", "sample"} <= set(c) and c["kind"] == "port" and c["value"] > 0
    assert c["cores"] <= c["host_cores"] and set(c["value_1thread_per_metric"]) == {"ssimulacra2", "dssim", "butteraugli"}
    # the pool size was swept and the best is reported, with what a worker achieves of the one-process rate
    assert c["threads"] == c["cores"] and str(c["cores"]) in c["pool_sweep_mp_per_s"] and c["value"] == max(c["pool_sweep_mp_per_s"].values())
    assert 0.0 < c["per_thread_efficiency"] <= 1.3
    assert d["value"] > c["value"] and all(v < 1e-4 for v in d["max_rel_dev_vs_oracle"].values())
    e = d["end_to_end"]
    # the host-buffer route runs the same workload with its uploads inside the timing: slower than the resident headline, but
    # the uploads overlap the kernels, so not by the 2x a serial upload + kernels would cost
    # (this test runs the --quick grid: 82 pairs, a step of a few milliseconds, where the call's fixed costs weigh several
    # times what they do on the sweep - 0.9 of the headline there, 0.45-0.7 here depending on the box - hence the loose floor)
    assert e["unit"] == "MP/s" and e["grid"].startswith("the whole workload") and 0.25 * d["value"] < e["value"] <= d["value"] * 1.3
    # ... and from ordinary host memory (host threads stage every image) the route is in the same range
    assert 0.2 * d["value"] < e["pageable"]["value"] <= e["value"] * 1.6


def _check_two_rank_line(out):
    assert out.returncode == 0, out.stderr[-2000:]
    lines = _json_lines(out.stdout)
    assert len(lines) == 1  # rank 0 only
    d = lines[0]
    assert d["n_gpus"] == 2 and d["cpu_baseline"] is None  # the CPU baseline is an N = 1 measurement
    # whole-job value = both ranks' (pair, metric) evaluations over the slower rank's time
    assert abs(d["value"] - 3 * d["config"]["megapixels_per_step"] * d["steps"] / (d["ms_per_step"] * d["steps"] / 1e3)) < 1e-2 * d["value"]
    # ONE global grid was partitioned: the shards tile it, the gathered scores are complete and a foreign shard's item
    # recomputed on rank 0 is bit-identical
    sh = d["shard"]
    n = 2 * (6 * 3 + 8 * 8)  # two --quick sweeps
    assert sh["partition"] == "reference" and len(sh["pairs_per_rank"]) == 2 and sum(sh["pairs_per_rank"]) == d["config"]["pairs_per_step"] == n
    assert sh["gathered_scores"] == n and sh["recomputed_on_rank0"] >= 1 and sh["recomputed_max_abs_diff"] == 0.0
    assert sh["imbalance_max_over_mean"] == 1.0 and len(sh["seconds_per_rank"]) == 2
    # the same line carries the strong-scaling leg: a FIXED grid (BASELINE configs[3] shapes) over the same ranks
    st = d["strong"]
    assert st["scaling"] == "strong" and st["pairs_per_rank"] == [64, 64] and st["imbalance_max_over_mean"] == 1.0
    assert st["value"] > 0 and len(st["seconds_per_rank"]) == 2 and st["recomputed_max_abs_diff"] == 0.0 and st["metrics"] == ["dssim", "ssimulacra2"]


def test_two_rank_path_sharing_one_device():
    """The driver's form: torch.distributed.run starts the ranks."""
    env = dict(os.environ, CE_BENCH_SHARE_DEVICE="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29541", "bench.py", "--gpus", "2", "--quick", "--steps", "4", "--warmup", "1"]
    _check_two_rank_line(subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900))


def test_two_ranks_from_the_plain_command():
    """`python bench.py --gpus 2` with NO launcher: the parent (no GPU call) starts the two ranks itself."""
    env = dict(os.environ, CE_BENCH_SHARE_DEVICE="1")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, "bench.py", "--gpus", "2", "--quick", "--steps", "4", "--warmup", "1"]
    _check_two_rank_line(subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900))


def test_two_rank_fixed_grid_strong_scaling_with_unit_fallback():
    """configs[4] (few references): the fixed grid is split by (image, codec-config) when that balances better."""
    env = dict(os.environ, CE_BENCH_SHARE_DEVICE="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29543", "bench.py", "--gpus", "2", "--config", "5", "--refs", "3", "--steps", "2", "--warmup", "1", "--no-solo"]
    out = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    d = _json_lines(out.stdout)[0]
    assert d["scaling"] == "strong" and d["config"]["pairs_per_step"] == 300
    sh = d["shard"]
    assert sh["partition"] == "image-x-codec-config" and sh["pairs_per_rank"] == [150, 150] and sh["recomputed_max_abs_diff"] == 0.0
