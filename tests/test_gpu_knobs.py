"""Measurement knobs (environment variables read once per process) choose schedules and tile shapes, never results: every
setting must return the scores of the default build bit for bit.  Each setting runs in its own process."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SCRIPT = r"""
import importlib, json, sys
sys.path.insert(0, %r)
import numpy as np
ce = importlib.import_module("codec-eval_amd")
wl = importlib.import_module("codec-eval_amd.workloads")
ctx = ce.Context(0)
out = []
for (w, h, n_refs, q) in [(200, 136, 2, 3), (97, 131, 1, 2), (768, 512, 1, 2)]:
    b = ce.Batch(ctx, w, h, n_refs, n_refs * q)
    for r in range(n_refs):
        ref = wl.make_reference(w, h, 50 + r)
        b.set_reference(r, ref)
        for k in range(q):
            b.set_test(r * q + k, r, wl.distort(ref, 35 + 25 * k))
    for cfg in (ce.MetricConfig.all(), ce.MetricConfig.all().with_xyb_roundtrip()):
        s = b.run(n_refs * q, cfg)
        out += [[x.psnr, x.ssimulacra2, x.dssim, x.butteraugli] for x in s]
        out.append(b.butteraugli_pnorm3(n_refs * q).tolist())
    b.close()
print(json.dumps(out))
""" % ROOT

SETTINGS = [
    {},
    {"CE_METRIC_STREAMS": "fork"},
    {"CE_METRIC_STREAMS": "serial"},
    {"CE_METRIC_STREAMS": "fork:dssim,ba"},
    {"CE_METRIC_STREAMS": "fork", "CE_FORK_ORDER": "210"},
    {"CE_METRIC_STREAMS": "fork", "CE_FORK_THREADS": "0"},
    {"CE_XCD_ORDER": "0"},
    {"CE_HV_ROWS": "64"},
    {"CE_MALTA_ROWS": "64"},
    {"CE_METRIC_FORK_BELOW_MP": "0", "CE_METRIC_FORK_ALONE_BELOW_MP": "0"},
    {"GPU_MAX_HW_QUEUES": "2"},
]


def _run(env_extra):
    env = dict(os.environ)
    env.update(env_extra)
    r = subprocess.run([sys.executable, "-c", SCRIPT], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, (env_extra, r.stderr[-2000:])
    return json.loads(r.stdout.strip().splitlines()[-1])


def test_knobs_do_not_change_any_score():
    want = _run(SETTINGS[0])
    assert len(want) == 2 * ((6 + 1) + (2 + 1) + (2 + 1))  # per batch and config: its pairs' score rows + one 3-norm list
    for env in SETTINGS[1:]:
        got = _run(env)
        if "CE_MALTA_ROWS" in env:
            # the 64-row Malta tile forms the 3-norm's f64 partial sums per 64 x 64 tile: same terms, another fixed order
            for a, b in zip(got, want):
                if len(a) == 4:  # [psnr, ssimulacra2, dssim, butteraugli] of a pair
                    assert a == b, env
                else:
                    assert all(abs(x - y) <= 1e-12 * max(abs(y), 1.0) for x, y in zip(a, b)), env
        else:
            assert got == want, env


SCRIPT_EVAL_BATCH = r"""
import importlib, json, sys
sys.path.insert(0, %r)
import numpy as np
ce = importlib.import_module("codec-eval_amd")
wl = importlib.import_module("codec-eval_amd.workloads")
ctx = ce.Context(0)
cfg = ce.MetricConfig.all()
out = []
for pinned in (True, False):
    items, keep = [], []
    for (w, h, n_refs, q) in [(96, 64, 7, 5), (64, 96, 3, 4)]:
        slab = ctx.host_buffer(n_refs * (q + 1) * w * h * 3) if pinned else np.empty(n_refs * (q + 1) * w * h * 3, np.uint8)
        keep.append(slab)
        off = 0
        for r in range(n_refs):
            ref = wl.make_reference(w, h, 90 + r)
            rv = slab[off:off + ref.size]; rv[:] = ref.reshape(-1); off += ref.size
            for k in range(q):
                t = wl.distort(ref, 30 + 15 * k)
                tv = slab[off:off + t.size]; tv[:] = t.reshape(-1); off += t.size
                items.append((rv, tv, w, h))
    for _ in range(2):  # the second call reuses the pooled batches
        out.append([[s.psnr, s.ssimulacra2, s.dssim, s.butteraugli, s.status] for s in ctx.eval_batch(items, cfg)])
print(json.dumps(out))
""" % ROOT

# per pair of 96 x 64 with a reference of its own, all metrics: a budget of a few pairs per chunk forces ramps and ring reuse
SETTINGS_EVAL_BATCH = [
    {},
    {"CE_UPLOAD_STREAMS": "1"},
    {"CE_EVAL_BATCH_RAMP": "0"},
    {"CE_EVAL_BATCH_RAMP": "2", "CE_EVAL_BATCH_BYTES": str(40 << 20)},
    {"CE_EVAL_BATCH_BYTES": str(24 << 20), "CE_UPLOAD_STREAMS": "1"},
    {"CE_EVAL_BATCH_CHUNKS": "3"},
]


def test_upload_and_chunking_knobs_do_not_change_any_score():
    def run(env_extra):
        env = dict(os.environ)
        env.update(env_extra)
        r = subprocess.run([sys.executable, "-c", SCRIPT_EVAL_BATCH], env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, (env_extra, r.stderr[-2000:])
        return json.loads(r.stdout.strip().splitlines()[-1])
    want = run(SETTINGS_EVAL_BATCH[0])
    assert len(want) == 4 and want[0] == want[1] == want[2] == want[3] and all(row[4] == 0 for row in want[0]) and len(want[0]) == 7 * 5 + 3 * 4
    for env in SETTINGS_EVAL_BATCH[1:]:
        assert run(env) == want, env
