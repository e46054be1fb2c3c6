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
    r = d["roofline"]
    assert {"bound", "achieved", "peak", "unit", "frac", "traffic"} <= set(r) and r["bound"] == "hbm" and r["peak"] == 8000.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3 and 0.0 < r["frac"] < 1.0
    c = d["cpu_baseline"]
    assert {"value", "unit", "cores", "kind", "sample"} <= set(c) and c["kind"] == "port" and c["value"] > 0
    assert d["value"] > c["value"] and d["max_rel_dev_vs_oracle"] < 1e-4


def test_two_rank_path_sharing_one_device():
    env = dict(os.environ, CE_BENCH_SHARE_DEVICE="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29541", "bench.py", "--gpus", "2", "--quick", "--steps", "4", "--warmup", "1"]
    out = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = _json_lines(out.stdout)
    assert len(lines) == 1  # rank 0 only
    d = lines[0]
    assert d["n_gpus"] == 2 and d["cpu_baseline"] is None  # the CPU baseline is an N = 1 measurement
    # whole-job value = both ranks' pixels over the slower rank's time
    assert abs(d["value"] - 2 * d["config"]["megapixels_per_gpu_step"] * d["steps"] / (d["ms_per_step"] * d["steps"] / 1e3)) < 1e-2 * d["value"]
