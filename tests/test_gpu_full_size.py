"""Parity at BASELINE.json's full sizes.

configs[2] works on 3840x2160 images: one such pair is scored by every metric and compared with the oracle directly
(the C oracle needs ~40 s of one host core for it), and the whole config-3 batch is checked through properties that do
not depend on size (identity, batch == single, permutation of pairs, monotonic in the distortion strength).
configs[1] sizes (768x512, 512x768) are compared with the oracle directly in test_gpu_parity.py / test_gpu_butteraugli.py,
configs[3] / configs[4] (512x512 grids) in test_gpu_configs.py."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

W, H = 3840, 2160


def rel(got, want, floor):
    return abs(got - want) / max(abs(want), floor)


def test_4k_pair_every_metric_against_the_oracle(gpu_ctx, oracle, ce, workloads):
    g = workloads.uhd_pairs(1, seed0=2000)
    ref, (ri, test) = g.references[0], g.pairs[0]
    assert (g.width, g.height) == (W, H)
    m = gpu_ctx.calculate_metrics(ref, test, W, H, ce.MetricConfig.all())
    assert m.psnr == oracle.psnr(ref, test, W, H)  # bit-exact
    assert rel(m.ssimulacra2, oracle.ssimulacra2(ref, test, W, H, 1), 1.0) <= 1e-4
    assert rel(m.dssim, oracle.dssim(ref, test, W, H), 1e-6) <= 1e-4
    want, want_p3 = oracle.butteraugli(ref, test, W, H)
    assert rel(m.butteraugli, want, 1e-3) <= 1e-4
    # BASELINE configs[2] names the 3-norm (libjxl's mean of the 3-, 6- and 12-norms of the diffmap)
    b = ce.Batch(gpu_ctx, W, H, 1, 1)
    b.set_reference(0, ref)
    b.set_test(0, 0, test)
    s = b.run(1, ce.MetricConfig(butteraugli=True))
    assert s[0].butteraugli == m.butteraugli
    assert rel(float(b.butteraugli_pnorm3(1)[0]), want_p3, 1e-3) <= 1e-4
    b.close()
    rt = gpu_ctx.xyb_roundtrip(ref, W, H)
    assert np.array_equal(rt, oracle.xyb_roundtrip(ref, W, H))  # u8-exact on all 8.3 M pixels


def test_4k_batch_properties(gpu_ctx, ce, workloads):
    """BASELINE configs[2] at its full batch: 16 pairs of 3840x2160 resident at once (4 references x 4 qualities; ~24 GB of
    working planes), Butteraugli + SSIMULACRA2."""
    n_refs, quals = 4, (85, 70, 92, 60)
    refs = [workloads.make_reference(W, H, 2100 + i) for i in range(n_refs)]
    pairs = [(i, workloads.distort(refs[i], q)) for i in range(n_refs) for q in quals]
    n = len(pairs)
    assert n == 16
    cfg = ce.MetricConfig(butteraugli=True, ssimulacra2=True)
    b = ce.Batch(gpu_ctx, W, H, n_refs, n + 3)
    for i, r in enumerate(refs):
        b.set_reference(i, r)
    for k, (ri, t) in enumerate(pairs):
        b.set_test(k, ri, t)
    # extra slots: an identical pair, and reference 0 at two more distortion strengths
    b.set_test(n, 0, refs[0])
    strong, weak = workloads.distort(refs[0], 40), workloads.distort(refs[0], 97)
    b.set_test(n + 1, 0, strong)
    b.set_test(n + 2, 0, weak)
    s = b.run(n + 3, cfg)
    assert all(x.status == 0 for x in s)
    assert s[n].butteraugli == 0.0 and s[n].ssimulacra2 == 100.0  # identity
    assert s[n + 1].butteraugli > s[0].butteraugli > s[n + 2].butteraugli > 0.0  # q40 worse than q85 worse than q97
    assert s[n + 1].ssimulacra2 < s[0].ssimulacra2 < s[n + 2].ssimulacra2 < 100.0
    # within a reference the scores follow the quality: q60 < q70 < q85 < q92
    for i in range(n_refs):
        by_q = {q: s[4 * i + j] for j, q in enumerate(quals)}
        assert by_q[60].ssimulacra2 < by_q[70].ssimulacra2 < by_q[85].ssimulacra2 < by_q[92].ssimulacra2
        assert by_q[60].butteraugli > by_q[70].butteraugli > by_q[85].butteraugli > by_q[92].butteraugli
    # batch == single call, bit for bit (pairs never interact)
    single = gpu_ctx.calculate_metrics(refs[1], pairs[5][1], W, H, cfg)
    assert (single.butteraugli, single.ssimulacra2) == (s[5].butteraugli, s[5].ssimulacra2)
    # permuting the pairs permutes the scores
    for k, (ri, t) in enumerate(reversed(pairs)):
        b.set_test(k, ri, t)
    s2 = b.run(n, cfg)
    assert [(x.butteraugli, x.ssimulacra2) for x in s2] == [(x.butteraugli, x.ssimulacra2) for x in reversed(s[:n])]
    b.close()
