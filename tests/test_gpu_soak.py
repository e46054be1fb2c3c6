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
