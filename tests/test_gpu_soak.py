"""Randomised sweep (fixed seed): random shapes in [8, 400)^2, smooth / noise / FLAT content, JPEG-like and random
distortions, with and without the XYB roundtrip; every metric against the oracle.

Flat references are the hard case for Butteraugli: X = c0 - c1 is a difference of two nearly equal values there, and a
logarithm that differs by one ulp between host and device moved the score by up to 4e-3 relative until both sides took
the lineage's FastLog2f (IEEE basic operations only).  Since then the device output equals the oracle's bit for bit."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_random_shapes_and_contents_all_metrics(gpu_ctx, oracle, ce, workloads):
    rng = np.random.default_rng(20260)
    cfg = ce.MetricConfig.all()
    worst = {"ssimulacra2": 0.0, "dssim": 0.0, "butteraugli": 0.0}
    for it in range(80):
        w, h = int(rng.integers(8, 400)), int(rng.integers(8, 400))
        if it % 10 == 0:
            w, h = int(rng.integers(8, 20)), int(rng.integers(8, 20))
        kind = ["natural", "highfreq", "flat"][it % 3]
        ref = workloads.make_reference(w, h, int(rng.integers(0, 1 << 30)), kind)
        test = workloads.distort(ref, int(rng.integers(5, 100))) if it % 7 else rng.integers(0, 256, ref.shape, dtype=np.uint8)
        rt = it % 4 == 0
        m = gpu_ctx.calculate_metrics(ref, test, w, h, cfg.with_xyb_roundtrip() if rt else cfg)
        r = oracle.xyb_roundtrip(ref, w, h) if rt else ref
        assert m.psnr == oracle.psnr(r, test, w, h), (w, h, kind)
        want = {"ssimulacra2": oracle.ssimulacra2(r, test, w, h, 1), "dssim": oracle.dssim(r, test, w, h),
                "butteraugli": oracle.butteraugli(r, test, w, h)[0]}
        for k, floor in (("ssimulacra2", 1.0), ("dssim", 1e-6), ("butteraugli", 1e-3)):
            rel = abs(getattr(m, k) - want[k]) / max(abs(want[k]), floor)
            worst[k] = max(worst[k], rel)
            assert rel <= 1e-4, (k, w, h, kind, rt, getattr(m, k), want[k])
    # far inside the 1e-4 bar: SSIMULACRA2 differs only through f32 pooling of the map terms, DSSIM and Butteraugli
    # follow the oracle's operations one for one
    assert worst["ssimulacra2"] < 1e-6 and worst["dssim"] < 1e-9 and worst["butteraugli"] < 1e-6, worst


def test_random_grids_all_metrics_against_the_oracle(gpu_ctx, oracle, ce, workloads):
    """Random (shape, references, pairs, bindings) grids with all four metrics in one launch: the SSIMULACRA2 row pass
    produces the reference-only blur streams once per reference (the other pairs of the reference read them), and the
    three perceptual metrics run as concurrent kernel chains - every pair must still match the oracle, on the first run
    and on a repeat with the cached tables."""
    rng = np.random.default_rng(11)
    cfg = ce.MetricConfig.all()
    for it in range(14):
        w, h = int(rng.integers(8, 260)), int(rng.integers(8, 260))
        if it % 4 == 0:
            w, h = int(rng.integers(90, 200)), int(rng.integers(40, 60))
        R, P = int(rng.integers(1, 5)), int(rng.integers(1, 9))
        refs = [workloads.make_reference(w, h, int(rng.integers(0, 1 << 30)), ["natural", "highfreq", "flat"][int(rng.integers(0, 3))])
                for _ in range(R)]
        bind = rng.integers(0, R, P)
        tests = [workloads.distort(refs[int(bind[k])], int(rng.integers(5, 100))) for k in range(P)]
        b = ce.Batch(gpu_ctx, w, h, R, P)
        for i, r in enumerate(refs):
            b.set_reference(i, r)
        for k in range(P):
            b.set_test(k, int(bind[k]), tests[k])
        want = []
        for k in range(P):
            r = refs[int(bind[k])]
            want.append({"ssimulacra2": oracle.ssimulacra2(r, tests[k], w, h, 1), "dssim": oracle.dssim(r, tests[k], w, h),
                         "butteraugli": oracle.butteraugli(r, tests[k], w, h)[0], "psnr": oracle.psnr(r, tests[k], w, h)})
        for rep in range(2):
            out = b.run(P, cfg)
            for k in range(P):
                assert out[k].psnr == want[k]["psnr"], (it, k)
                for key, floor in (("ssimulacra2", 1.0), ("dssim", 1e-6), ("butteraugli", 1e-3)):
                    rel = abs(getattr(out[k], key) - want[k][key]) / max(abs(want[k][key]), floor)
                    assert rel <= 1e-6, (key, it, rep, w, h, R, P, k, getattr(out[k], key), want[k][key])
        b.close()


def test_irregular_pair_to_reference_bindings(gpu_ctx, ce, workloads):
    """The level-0 SSIMULACRA2 passes run from host-built XCD-aware work lists keyed by the pair -> reference table.
    Unsorted pairs, references with 0, 1 or many pairs, re-binding between runs: every pair must score exactly what
    it scores alone."""
    rng = np.random.default_rng(5)
    w, h, R, P = 150, 110, 7, 23
    cfg = ce.MetricConfig(ssimulacra2=True, psnr=True)
    refs = [workloads.make_reference(w, h, 100 + i) for i in range(R)]
    b = ce.Batch(gpu_ctx, w, h, R, P)
    for i, r in enumerate(refs):
        b.set_reference(i, r)
    for rnd in range(3):
        bind = rng.integers(0, R if rnd else 3, P)  # round 0: only references 0..2 are used
        tests = [workloads.distort(refs[int(bind[k])], int(rng.integers(20, 98))) for k in range(P)]
        for k in range(P):
            b.set_test(k, int(bind[k]), tests[k])
        n = P if rnd != 1 else 11  # a shorter run in between
        out = b.run(n, cfg)
        for k in range(n):
            solo = gpu_ctx.calculate_metrics(refs[int(bind[k])], tests[k], w, h, cfg)
            assert (out[k].ssimulacra2, out[k].psnr) == (solo.ssimulacra2, solo.psnr), (rnd, k, int(bind[k]))
    b.close()


def test_contexts_on_concurrent_host_threads(oracle, ce, workloads):
    """SURVEY.md 8b threading: one in-flight call per context, any number of contexts per device, created and used from
    different host threads (full_comparison.rs:319-328 calls the leaf functions from rayon workers)."""
    from concurrent.futures import ThreadPoolExecutor

    w, h = 120, 90
    cases = []
    for i in range(6):
        ref = workloads.make_reference(w, h, 900 + i)
        test = workloads.distort(ref, 40 + 9 * i)
        cases.append((ref, test, oracle.ssimulacra2(ref, test, w, h, 1), oracle.psnr(ref, test, w, h)))

    def worker(idx):
        ctx = ce.Context(0)
        out = []
        for rep in range(8):
            ref, test, _, _ = cases[(idx + rep) % len(cases)]
            m = ctx.calculate_metrics(ref, test, w, h, ce.MetricConfig(ssimulacra2=True, psnr=True, dssim=True, butteraugli=True))
            out.append(((idx + rep) % len(cases), m))
        ctx.close()
        return out

    with ThreadPoolExecutor(4) as ex:
        results = list(ex.map(worker, range(4)))
    first = {}
    for res in results:
        for k, m in res:
            _, _, s2, ps = cases[k]
            assert m.psnr == ps and abs(m.ssimulacra2 - s2) <= 1e-4 * max(abs(s2), 1.0)
            key = (m.ssimulacra2, m.dssim, m.butteraugli)
            assert first.setdefault(k, key) == key  # every thread / context gets the same bits


def test_eval_batch_streams_large_buckets_in_chunks(gpu_ctx, ce, workloads):
    """ce_eval_batch streams a bucket of more than ~16 pairs through a ring of pooled batches in chunks of whole
    references (uploads of a chunk overlap the previous chunk's kernels).  Scores must come back in the caller's
    order whatever the interleaving of references and shapes, and repeated calls must reuse the pool correctly."""
    rng = np.random.default_rng(11)
    shapes = [(96, 64), (64, 96)]
    refs = {s: [workloads.make_reference(s[0], s[1], 50 + 10 * si + i) for i in range(7)] for si, s in enumerate(shapes)}
    cfg = ce.MetricConfig(ssimulacra2=True, psnr=True)
    for rnd in range(2):
        pairs = []
        for _ in range(75):
            s = shapes[int(rng.integers(0, 2))]
            r = refs[s][int(rng.integers(0, 7))]
            pairs.append((r, workloads.distort(r, int(rng.integers(10, 99))), s[0], s[1]))
        out = gpu_ctx.eval_batch(pairs, cfg)
        assert len(out) == len(pairs) and all(o.status == 0 for o in out)
        for k in rng.choice(len(pairs), 12, replace=False):
            r, t, w, h = pairs[int(k)]
            solo = gpu_ctx.calculate_metrics(r, t, w, h, cfg)
            assert (out[int(k)].ssimulacra2, out[int(k)].psnr) == (solo.ssimulacra2, solo.psnr), (rnd, int(k))


def test_eval_batch_route_soak_mixed_shapes_memory_kinds_and_budgets(gpu_ctx, ce, workloads):
    """The host-buffer route under random conditions (fixed seed): grids that mix two or three shapes, references with 1-9
    distorted images, page-locked / ordinary / mixed source memory, chunk budgets from "everything in one chunk" down to three
    pairs (ramped chunk sizes, ring slots reused while later chunks are in flight, two DMA streams, work lists rebuilt when a
    slot's pair count changes), the same context reused call after call.  Pairs never interact, so every score must equal -
    exactly - the one the same pair gets alone in a one-pair call."""
    import os

    rng = np.random.default_rng(4242)
    cfg = ce.MetricConfig.all()
    shapes = [(96, 64), (64, 96), (130, 50), (72, 72)]
    pool = {}  # (shape, seed, q) -> (reference, distorted, scores of the pair alone)

    def alone(shape, seed, q):
        key = (shape, seed, q)
        if key not in pool:
            w, h = shape
            ref = workloads.make_reference(w, h, 7000 + seed)
            t = workloads.distort(ref, q)
            m = gpu_ctx.calculate_metrics(ref, t, w, h, cfg)
            pool[key] = (ref, t, (m.psnr, m.ssimulacra2, m.dssim, m.butteraugli))
        return pool[key]
    saved = os.environ.get("CE_EVAL_BATCH_BYTES")
    try:
        for it in range(48):
            use = [shapes[i] for i in rng.choice(len(shapes), size=int(rng.integers(1, 4)), replace=False)]
            cells = []
            for shape in use:
                for seed in rng.choice(12, size=int(rng.integers(1, 7)), replace=False):
                    for q in rng.choice([20, 35, 50, 65, 80, 90, 95, 97, 99], size=int(rng.integers(1, 10)), replace=False):
                        cells.append((shape, int(seed), int(q)))
            order = rng.permutation(len(cells))
            cells = [cells[i] for i in order]
            kind = ["pinned", "pageable", "mixed"][it % 3]
            n_bytes = sum(2 * s[0] * s[1] * 3 for s, _, _ in cells)
            slab = gpu_ctx.host_buffer(n_bytes) if kind != "pageable" else None
            off, refs_placed, items, want = 0, {}, [], []
            for shape, seed, q in cells:
                ref, t, sc = alone(shape, seed, q)
                w, h = shape
                pin = kind == "pinned" or (kind == "mixed" and (seed + q) % 2 == 0)
                if (shape, seed, pin) not in refs_placed:  # one buffer per reference: its pairs share the upload
                    if pin:
                        v = slab[off:off + ref.size]
                        v[:] = ref.reshape(-1)
                        off += ref.size
                    else:
                        v = ref.reshape(-1).copy()
                    refs_placed[(shape, seed, pin)] = v
                if pin:
                    tv = slab[off:off + t.size]
                    tv[:] = t.reshape(-1)
                    off += t.size
                else:
                    tv = t.reshape(-1).copy()
                items.append((refs_placed[(shape, seed, pin)], tv, w, h))
                want.append(sc)
            budget = [None, 3, 8, 20][int(rng.integers(0, 4))]
            if budget is None:
                os.environ.pop("CE_EVAL_BATCH_BYTES", None)
            else:
                per_pair = max(ce.estimate_batch_bytes(w_, h_, 2, 2, cfg) - ce.estimate_batch_bytes(w_, h_, 1, 1, cfg) for w_, h_ in use)
                os.environ["CE_EVAL_BATCH_BYTES"] = str(per_pair * budget)
            got = [(s.psnr, s.ssimulacra2, s.dssim, s.butteraugli) for s in gpu_ctx.eval_batch(items, cfg)]
            assert got == want, (it, kind, budget, len(items))
    finally:
        if saved is None:
            os.environ.pop("CE_EVAL_BATCH_BYTES", None)
        else:
            os.environ["CE_EVAL_BATCH_BYTES"] = saved
