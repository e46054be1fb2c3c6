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
