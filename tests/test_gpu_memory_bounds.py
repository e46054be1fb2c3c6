"""A grid larger than the device budget streams through the device in chunks instead of failing:
ce_eval_batch splits a shape's grid by bytes per pair (whole references; a reference with more cells than fit is split),
EvalSession does the same for its per-shape batches.  The budget is forced down with CE_EVAL_BATCH_BYTES /
CE_SESSION_BATCH_BYTES so that a small grid exercises many chunks; the scores must be those of the unconstrained run,
bit for bit (pairs never interact)."""
import importlib
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

S = importlib.import_module("codec-eval_amd.session")


def test_estimate_is_monotone_and_covers_a_real_batch(gpu_ctx, ce, workloads):
    cfg = ce.MetricConfig.all()
    e = lambda r, p, c=cfg: ce.estimate_batch_bytes(256, 192, r, p, c)
    assert e(1, 1) < e(1, 2) < e(2, 2) and e(1, 4, ce.MetricConfig.fast()) < e(1, 4, ce.MetricConfig(ssimulacra2=True)) < e(1, 4)
    g = workloads._grid("t", 256, 192, 2, 50, (40, 80))
    # what the first launches of a process load once (code objects, the context's tables and streams) is not the batch's
    gpu_ctx.calculate_metrics(g.references[0], g.pairs[0][1], 256, 192, cfg)
    free0, total = gpu_ctx.memory_info()
    assert 0 < free0 <= total
    b = ce.Batch(gpu_ctx, 256, 192, 2, 4)
    for i, r in enumerate(g.references):
        b.set_reference(i, r)
    for k, (ri, t) in enumerate(g.pairs):
        b.set_test(k, ri, t)
    b.run(4, cfg)
    used = free0 - gpu_ctx.memory_info()[0]
    b.close()
    assert 0 < used <= e(2, 4), (used, e(2, 4))  # the estimate is an upper bound of what the batch really holds


def test_eval_batch_streams_a_grid_that_does_not_fit(gpu_ctx, ce, workloads):
    w, h = 160, 96
    refs = [workloads.make_reference(w, h, 300 + i) for i in range(3)]
    items = []
    for i, r in enumerate(refs):
        for q in ((30, 50, 70, 90, 95, 97, 99) if i == 1 else (40, 85)):  # reference 1 has more cells than one chunk holds
            items.append((r, workloads.distort(r, q), w, h))
    cfg = ce.MetricConfig.all()
    want = [(s.psnr, s.ssimulacra2, s.dssim, s.butteraugli, s.status) for s in gpu_ctx.eval_batch(items, cfg)]
    per_pair = ce.estimate_batch_bytes(w, h, 2, 2, cfg) - ce.estimate_batch_bytes(w, h, 1, 1, cfg)  # one more pair with its own reference
    old = os.environ.get("CE_EVAL_BATCH_BYTES")
    try:
        for pairs_per_chunk in (3, 1):
            os.environ["CE_EVAL_BATCH_BYTES"] = str(per_pair * pairs_per_chunk)
            got = [(s.psnr, s.ssimulacra2, s.dssim, s.butteraugli, s.status) for s in gpu_ctx.eval_batch(items, cfg)]
            assert got == want, pairs_per_chunk
    finally:
        if old is None:
            os.environ.pop("CE_EVAL_BATCH_BYTES", None)
        else:
            os.environ["CE_EVAL_BATCH_BYTES"] = old
    assert all(s[4] == 0 for s in want) and len({s[1] for s in want}) == len(want)


def test_session_splits_a_shape_into_batches_that_fit(gpu_ctx, ce, workloads, tmp_path):
    cfg = S.EvalConfig.builder().report_dir(tmp_path / "rep").metrics(ce.MetricConfig.all()).quality_levels([35, 60, 80, 92]).build()

    def encode(image, request):
        return np.array([request.quality], np.float32).tobytes() + image.to_rgb8_vec().tobytes()

    def decode(blob):
        q = float(np.frombuffer(blob[:4], np.float32)[0])
        rgb = np.frombuffer(blob[4:], np.uint8).reshape(64, 96, 3)
        return S.ImageData.rgb(workloads.distort(rgb, q), 96, 64)

    images = [(f"i{k}.png", S.ImageData.rgb(workloads.make_reference(96, 64, 400 + k), 96, 64)) for k in range(3)]

    def run():
        ses = S.EvalSession(cfg, ctx=gpu_ctx)
        ses.add_codec_with_decode("toy", "1", encode, decode)
        rep = ses.evaluate_corpus("c", images)
        return [(r.quality, r.psnr, r.ssimulacra2, r.dssim, r.butteraugli) for im in rep.images for r in im.results]

    want = run()
    per_pair = ce.estimate_batch_bytes(96, 64, 0, 1, cfg.metrics) - ce.estimate_batch_bytes(96, 64, 0, 0, cfg.metrics)
    fixed = ce.estimate_batch_bytes(96, 64, 1, 0, cfg.metrics)
    old = os.environ.get("CE_SESSION_BATCH_BYTES")
    try:
        os.environ["CE_SESSION_BATCH_BYTES"] = str(fixed + 2 * per_pair)  # two cells per device batch: images are split
        assert run() == want
    finally:
        if old is None:
            os.environ.pop("CE_SESSION_BATCH_BYTES", None)
        else:
            os.environ["CE_SESSION_BATCH_BYTES"] = old
    assert len(want) == 12 and all(v is not None for row in want for v in row)


def test_collect_waits_for_its_own_launch_only(gpu_ctx, ce, workloads):
    """Two batches launched back to back on ONE context, collected in the reverse order: ce_batch_collect waits for the
    event of its batch's launch (the scores are copied behind that launch's kernels), so either order returns each batch's
    own scores; asking for more pairs than were launched, or collecting a batch that never ran, is an error."""
    cfg = ce.MetricConfig.all()
    grids = [workloads._grid("a", 160, 96, 2, 70, (40, 85)), workloads._grid("b", 96, 160, 1, 90, (30, 60, 95))]
    batches, want = [], []
    for g in grids:
        b = ce.Batch(gpu_ctx, g.width, g.height, len(g.references), len(g.pairs))
        for i, r in enumerate(g.references):
            b.set_reference(i, r)
        for k, (ri, t) in enumerate(g.pairs):
            b.set_test(k, ri, t)
        with pytest.raises(ce.CodecEvalError):
            b.collect(1)  # nothing launched yet
        want.append([(s.psnr, s.ssimulacra2, s.dssim, s.butteraugli) for s in b.run(len(g.pairs), cfg)])
        batches.append(b)
    for order in ((0, 1), (1, 0)):
        for b, g in zip(batches, grids):
            b.launch(len(g.pairs), cfg)
        got = {}
        for j in order[::-1]:
            got[j] = [(s.psnr, s.ssimulacra2, s.dssim, s.butteraugli) for s in batches[j].collect(len(grids[j].pairs))]
        assert [got[0], got[1]] == want
    # a prefix of the launched pairs is fine, more than were launched is not
    batches[0].launch(2, cfg)
    assert [(s.psnr, s.ssimulacra2, s.dssim, s.butteraugli) for s in batches[0].collect(1)] == want[0][:1]
    with pytest.raises(ce.CodecEvalError):
        batches[0].collect(3)
    for b in batches:
        b.close()


def test_eval_batch_ramp_and_ring_reuse_keep_every_score(gpu_ctx, ce, workloads):
    """A bucket cut into many chunks (first chunk small, doubling; ring slots reused while later chunks are in flight; the
    page-locked uploads split over two streams) returns exactly the scores of the one-chunk call, in the caller's order."""
    w, h = 96, 64
    refs = [workloads.make_reference(w, h, 500 + i) for i in range(12)]
    pinned = gpu_ctx.host_buffer(12 * 5 * h * w * 3).reshape(12 * 5, h * w * 3)  # ce_host_alloc: the DMA route
    items = []
    for i, r in enumerate(refs):
        pinned[5 * i][:] = r.reshape(-1)
        for k, q in enumerate((35, 60, 80, 95)):
            pinned[5 * i + 1 + k][:] = workloads.distort(r, q).reshape(-1)
            items.append((pinned[5 * i], pinned[5 * i + 1 + k], w, h))
    cfg = ce.MetricConfig.all()
    want = [(s.psnr, s.ssimulacra2, s.dssim, s.butteraugli, s.status) for s in gpu_ctx.eval_batch(items, cfg)]
    per_pair = ce.estimate_batch_bytes(w, h, 2, 2, cfg) - ce.estimate_batch_bytes(w, h, 1, 1, cfg)
    saved = {k: os.environ.get(k) for k in ("CE_EVAL_BATCH_BYTES", "CE_EVAL_BATCH_RAMP")}
    try:
        os.environ["CE_EVAL_BATCH_BYTES"] = str(per_pair * 8)  # 48 pairs in chunks of <= 8: the three ring slots are reused twice
        got = [(s.psnr, s.ssimulacra2, s.dssim, s.butteraugli, s.status) for s in gpu_ctx.eval_batch(items, cfg)]
        assert got == want
        assert [(s.psnr, s.ssimulacra2, s.dssim, s.butteraugli, s.status) for s in gpu_ctx.eval_batch(items[::-1], cfg)] == want[::-1]
    finally:
        for k, v in saved.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    assert all(s[4] == 0 for s in want)
