"""The synthetic corpora are seeded and shaped as SURVEY.md §8d / Appendix B prescribe."""
import numpy as np


def test_generators_are_deterministic(workloads):
    a, b = workloads.make_reference(96, 64, 5), workloads.make_reference(96, 64, 5)
    assert a.shape == (64, 96, 3) and a.dtype == np.uint8 and np.array_equal(a, b)
    assert not np.array_equal(a, workloads.make_reference(96, 64, 6))
    assert a.min() == 0 and a.max() == 255  # full u8 range
    d1, d2 = workloads.distort(a, 60), workloads.distort(a, 60)
    assert np.array_equal(d1, d2) and d1.shape == a.shape
    err = [np.abs(workloads.distort(a, q).astype(int) - a).mean() for q in (20, 50, 80, 95)]
    assert all(x > y for x, y in zip(err, err[1:]))
    assert not np.array_equal(workloads.distort(a, 60, True), d1)  # 4:2:0 differs from 4:4:4
    odd = workloads.make_reference(101, 77, 1)
    assert workloads.distort(odd, 70).shape == (77, 101, 3)


def test_config_shapes(workloads):
    grids = workloads.kodak_like((75,), 2, 1)
    assert [(g.width, g.height, len(g.references), len(g.pairs)) for g in grids] == [(768, 512, 2, 2), (512, 768, 1, 1)]
    assert workloads.KODAK_LANDSCAPE + workloads.KODAK_PORTRAIT == 24
    assert workloads.STANDARD_QUALITIES == (50, 60, 70, 75, 80, 85, 90, 95) and len(workloads.DENSE_QUALITIES) == 25
    g = workloads.codec_iter_dense(2, (50, 98))
    assert len(g.pairs) == 2 * 2 * 2 and g.pairs[0][0] == 0 and g.pairs[-1][0] == 1
    flat = workloads.make_reference(32, 32, 3, "flat")
    assert np.array_equal(workloads.distort(flat, 90), flat)  # the identical-pair edge case
