"""BASELINE.json configs[3] and configs[4] as WORKLOADS, under -m gpu.

configs[3]: CID22-shaped 512x512 references x the standard 8-quality sweep, SSIMULACRA2 + DSSIM
            (crates/codec-iter/src/main.rs:198; 8 distorted images share one reference's DSSIM pyramid and
            SSIMULACRA2 reference streams)
configs[4]: codec-iter dense sweep: references x qualities x {4:4:4, 4:2:0} x {XYB off, on}, all four metrics
            (main.rs:199,470-499; the XYB roundtrip applies to the reference only, session.rs:447-456)

Scaled-down grids are compared pair by pair with the oracle (1e-4; PSNR and the XYB roundtrip exact); the FULL pair
counts (2000 and 1500) are checked through properties that need no oracle: an identical pair scores the identity
values, a batch equals single calls bit for bit, permuting the pairs permutes the scores, the scores fall with the
quality, and a sample of pairs is compared with the oracle."""
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def rel(got, want, floor):
    return abs(got - want) / max(abs(want), floor)


def fill(ce, ctx, g):
    b = ce.Batch(ctx, g.width, g.height, len(g.references), len(g.pairs))
    for i, r in enumerate(g.references):
        b.set_reference(i, r)
    for k, (ri, t) in enumerate(g.pairs):
        b.set_test(k, ri, t)
    return b


def oracle_scores(oracle, g, idx, metrics, xyb=False, threads=8):
    """{(pair index, metric): value} from the C oracle, a thread pool over the items (ctypes releases the GIL)."""
    rt = {}
    if xyb:
        for ri in {g.pairs[k][0] for k in idx}:
            rt[ri] = oracle.xyb_roundtrip(g.references[ri], g.width, g.height).reshape(g.references[ri].shape)
    fn = {"ssimulacra2": lambda r, t: oracle.ssimulacra2(r, t, g.width, g.height, 1), "dssim": lambda r, t: oracle.dssim(r, t, g.width, g.height),
          "butteraugli": lambda r, t: oracle.butteraugli(r, t, g.width, g.height)[0], "psnr": lambda r, t: oracle.psnr(r, t, g.width, g.height)}
    items = [(k, m) for k in idx for m in metrics]

    def one(km):
        k, m = km
        ri, t = g.pairs[k]
        return fn[m](rt[ri] if xyb else g.references[ri], t)

    with ThreadPoolExecutor(threads) as ex:
        return dict(zip(items, ex.map(one, items)))


FLOOR = {"ssimulacra2": 1.0, "dssim": 1e-6, "butteraugli": 1e-3}


def check_against_oracle(scores, want, idx, metrics):
    for k in idx:
        for m in metrics:
            got = getattr(scores[k], m)
            if m == "psnr":
                assert got == want[(k, m)], (k, m)
            else:
                assert rel(got, want[(k, m)], FLOOR[m]) <= 1e-4, (k, m, got, want[(k, m)])


# ---- configs[3] ---------------------------------------------------------------------------------------------------


def test_config3_cid22_sweep_every_pair_against_the_oracle(gpu_ctx, oracle, ce, workloads):
    g = workloads.cid22_like(4)  # 4 references x 8 qualities = 32 pairs of 512x512
    assert (g.width, g.height, len(g.pairs)) == (512, 512, 32)
    cfg = ce.MetricConfig(ssimulacra2=True, dssim=True)
    b = fill(ce, gpu_ctx, g)
    s = b.run(len(g.pairs), cfg)
    assert all(x.status == 0 and x.valid == cfg.mask for x in s)
    idx = list(range(len(g.pairs)))
    check_against_oracle(s, oracle_scores(oracle, g, idx, ("ssimulacra2", "dssim")), idx, ("ssimulacra2", "dssim"))
    # within one reference the scores follow the quality axis (50 .. 95)
    for r in range(4):
        q = s[8 * r:8 * r + 8]
        assert all(q[i].ssimulacra2 < q[i + 1].ssimulacra2 for i in range(7))
        assert all(q[i].dssim > q[i + 1].dssim for i in range(7))
    # a repeat of the launch (cached tables, XCD work lists) is bit-identical
    s2 = b.run(len(g.pairs), cfg)
    assert [(x.ssimulacra2, x.dssim) for x in s2] == [(x.ssimulacra2, x.dssim) for x in s]
    b.close()


def test_config3_full_pair_count_properties(gpu_ctx, oracle, ce, workloads):
    """The whole configs[3] grid: 250 references x 8 qualities = 2000 pairs in ONE resident batch."""
    n_refs = 250
    with ThreadPoolExecutor(8) as ex:  # generation is the slow part (numpy DCTs); images depend on their index only
        parts = list(ex.map(lambda i: workloads.cid22_like(n_refs, only=[i]), range(n_refs)))
    refs = [p.references[0] for p in parts]
    pairs = [(i, t) for i, p in enumerate(parts) for (_, t) in p.pairs]
    assert len(pairs) == 2000
    cfg = ce.MetricConfig(ssimulacra2=True, dssim=True)
    b = ce.Batch(gpu_ctx, 512, 512, n_refs, len(pairs) + 1)
    for i, r in enumerate(refs):
        b.set_reference(i, r)
    for k, (ri, t) in enumerate(pairs):
        b.set_test(k, ri, t)
    b.set_test(len(pairs), 7, refs[7])  # an identical pair in the last slot
    s = b.run(len(pairs) + 1, cfg)
    assert all(x.status == 0 for x in s)
    assert s[-1].ssimulacra2 == 100.0 and s[-1].dssim == 0.0  # identity
    key = [(x.ssimulacra2, x.dssim) for x in s[:-1]]
    assert all(np.isfinite(a) and np.isfinite(d) and a < 100.0 and d > 0.0 for a, d in key)
    for r in range(n_refs):  # monotone in the quality within every reference
        q = key[8 * r:8 * r + 8]
        assert all(q[i][0] < q[i + 1][0] and q[i][1] > q[i + 1][1] for i in range(7)), r
    # batch == single call, bit for bit, on a spread of pairs; and those pairs against the oracle
    sample = [0, 9, 778, 1234, 1999]
    for k in sample:
        ri, t = pairs[k]
        m = gpu_ctx.calculate_metrics(refs[ri], t, 512, 512, cfg)
        assert (m.ssimulacra2, m.dssim) == key[k], k
    g = workloads.Grid("cid22-512x512", 512, 512, refs, pairs)
    check_against_oracle(s, oracle_scores(oracle, g, sample, ("ssimulacra2", "dssim")), sample, ("ssimulacra2", "dssim"))
    # permutation: reverse the pair order in place (bindings move with the pixels)
    for k, (ri, t) in enumerate(reversed(pairs)):
        b.set_test(k, ri, t)
    s2 = b.run(len(pairs), cfg)
    assert [(x.ssimulacra2, x.dssim) for x in s2] == list(reversed(key))
    b.close()


# ---- configs[4] ---------------------------------------------------------------------------------------------------


@pytest.mark.parametrize("xyb", [False, True])
def test_config4_dense_sweep_every_pair_against_the_oracle(gpu_ctx, oracle, ce, workloads, xyb):
    g = workloads.codec_iter_dense(2, qualities=(50, 74, 98))  # 2 refs x 3 q x {4:4:4, 4:2:0} = 12 pairs
    assert len(g.pairs) == 12
    assert [pid[1] for pid in g.pair_ids[:6]] == [0, 0, 0, 1, 1, 1]  # 4:4:4 then 4:2:0 of reference 0
    cfg = ce.MetricConfig.all().with_xyb_roundtrip() if xyb else ce.MetricConfig.all()
    metrics = ("psnr", "ssimulacra2", "dssim", "butteraugli")
    b = fill(ce, gpu_ctx, g)
    s = b.run(len(g.pairs), cfg)
    assert all(x.status == 0 and x.valid == cfg.mask for x in s)
    idx = list(range(len(g.pairs)))
    check_against_oracle(s, oracle_scores(oracle, g, idx, metrics, xyb=xyb), idx, metrics)
    if xyb:  # the flag changes the reference the metrics see, never the distorted image
        plain = b.run(len(g.pairs), ce.MetricConfig.all())
        assert any(p.psnr != q.psnr for p, q in zip(plain, s))
        for ri, r in enumerate(g.references):
            assert np.array_equal(gpu_ctx.xyb_roundtrip(r, g.width, g.height), oracle.xyb_roundtrip(r, g.width, g.height))
    b.close()


def test_config4_full_pair_count_properties(gpu_ctx, oracle, ce, workloads):
    """The whole configs[4] grid: 15 references x 25 qualities x {4:4:4, 4:2:0} = 750 distorted images, scored with
    XYB off and XYB on = 1500 (pair, codec-config) evaluations of all four metrics."""
    n_refs = 15
    with ThreadPoolExecutor(8) as ex:
        parts = list(ex.map(lambda i: workloads.codec_iter_dense(n_refs, only=[i]), range(n_refs)))
    refs = [p.references[0] for p in parts]
    pairs = [(i, t) for i, p in enumerate(parts) for (_, t) in p.pairs]
    ids = [pid for p in parts for pid in p.pair_ids]
    assert len(pairs) == 750
    g = workloads.Grid("codec-iter-512x512", 512, 512, refs, pairs)
    b = ce.Batch(gpu_ctx, 512, 512, n_refs, len(pairs) + 1)
    for i, r in enumerate(refs):
        b.set_reference(i, r)
    for k, (ri, t) in enumerate(pairs):
        b.set_test(k, ri, t)
    b.set_test(len(pairs), 3, refs[3])
    metrics = ("psnr", "ssimulacra2", "dssim", "butteraugli")
    out = {}
    for xyb in (False, True):
        cfg = ce.MetricConfig.all().with_xyb_roundtrip() if xyb else ce.MetricConfig.all()
        s = b.run(len(pairs) + 1, cfg)
        assert all(x.status == 0 and x.valid == cfg.mask for x in s)
        if not xyb:  # identity only holds against the untouched reference
            assert s[-1].ssimulacra2 == 100.0 and s[-1].dssim == 0.0 and s[-1].butteraugli == 0.0 and np.isinf(s[-1].psnr)
        key = [(x.psnr, x.ssimulacra2, x.dssim, x.butteraugli) for x in s[:-1]]
        assert all(all(np.isfinite(v) for v in row) for row in key)
        out[xyb] = key
        sample = [0, 49, 333, 749]
        for k in sample:  # batch == single, bit for bit
            ri, t = pairs[k]
            m = gpu_ctx.calculate_metrics(refs[ri], t, 512, 512, cfg)
            assert (m.psnr, m.ssimulacra2, m.dssim, m.butteraugli) == key[k], (xyb, k)
        check_against_oracle(s, oracle_scores(oracle, g, sample, metrics, xyb=xyb), sample, metrics)
    # quality axis: within a (reference, subsampling) run of 25 qualities PSNR rises from q50 to q98
    for r in range(n_refs):
        for v in range(2):
            run = [out[False][k][0] for k in range(len(pairs)) if ids[k][0] == r and ids[k][1] == v]
            assert len(run) == 25 and run[0] < run[-1]
    # 4:2:0 loses chroma detail: at the top quality it scores below 4:4:4 on every reference (PSNR)
    for r in range(n_refs):
        top = {v: [out[False][k][0] for k in range(len(pairs)) if ids[k] == (r, v, 24)][0] for v in (0, 1)}
        assert top[1] < top[0], r
    # XYB on changes the scores (a different reference) but leaves the order of magnitude alone
    assert out[True] != out[False]
    # permutation under the XYB flag
    for k, (ri, t) in enumerate(reversed(pairs)):
        b.set_test(k, ri, t)
    s2 = b.run(len(pairs), ce.MetricConfig.all().with_xyb_roundtrip())
    assert [(x.psnr, x.ssimulacra2, x.dssim, x.butteraugli) for x in s2] == list(reversed(out[True]))
    b.close()
