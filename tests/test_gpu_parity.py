"""GPU parity tests proper: the HIP path, called through the C ABI, against the CPU oracle on the
same seeded inputs.  Bars (BASELINE.json north_star): PSNR and the XYB roundtrip bit-exact;
SSIMULACRA2 / DSSIM / Butteraugli within 1e-4 relative.

The synthetic inputs mirror the reference's own tests (src/metrics/*.rs #[cfg(test)]).
"""
import math

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

REL_TOL = 1e-4  # north_star: "within 1e-4 relative"


def rel_close(got, want, rel=REL_TOL, floor=1.0):
    """|got - want| <= rel * max(|want|, floor).  floor=1 for scores whose natural scale is O(1..100)."""
    return abs(got - want) <= rel * max(abs(want), floor)


def ramp(w, h):
    # src/metrics/ssimulacra2.rs:155  (0..w*h*3).map(|i| (i % 256) as u8)
    return (np.arange(w * h * 3) % 256).astype(np.uint8)


def helper_pattern(w, h, pattern):
    # src/eval/helpers.rs:327-335 (wrapping `as u8` casts)
    i = np.arange(w * h)
    base = (i + pattern) % 256
    return np.stack([base, base + 50, base + 100], 1).astype(np.uint8).reshape(h, w, 3)


# ---------------------------------------------------------------- PSNR (bit-exact) ----------


def test_psnr_reference_cases(gpu_ctx, oracle):
    # src/metrics/mod.rs:368-383
    same = np.full(100 * 100 * 3, 128, np.uint8)
    assert math.isinf(gpu_ctx.calculate_psnr(same, same, 100, 100))
    r, t = np.full(30000, 100, np.uint8), np.full(30000, 110, np.uint8)
    got = gpu_ctx.calculate_psnr(r, t, 100, 100)
    assert 28.0 < got < 29.0
    assert got == oracle.psnr(r, t, 100, 100)


@pytest.mark.parametrize("w,h", [(1, 1), (3, 5), (7, 7), (64, 64), (100, 100), (101, 77), (768, 512), (512, 768), (1920, 1080)])
def test_psnr_bit_exact(gpu_ctx, oracle, w, h):
    rng = np.random.default_rng(w * 10007 + h)
    r = rng.integers(0, 256, w * h * 3, dtype=np.uint8)
    t = np.clip(r.astype(np.int16) + rng.integers(-9, 10, r.size), 0, 255).astype(np.uint8)
    assert gpu_ctx.calculate_psnr(r, t, w, h) == oracle.psnr(r, t, w, h)
    # worst case: every sample differs by 255
    z, f = np.zeros(w * h * 3, np.uint8), np.full(w * h * 3, 255, np.uint8)
    assert gpu_ctx.calculate_psnr(z, f, w, h) == oracle.psnr(z, f, w, h) == 0.0


def test_psnr_errors(gpu_ctx, ce):
    a, b = np.zeros(50 * 50 * 3, np.uint8), np.zeros(100 * 100 * 3, np.uint8)
    with pytest.raises(ce.DimensionMismatch):  # mod.rs:313 asserts; here an error code
        gpu_ctx.calculate_psnr(a, b, 100, 100)
    with pytest.raises(ce.MetricCalculation):  # mod.rs:314
        gpu_ctx.calculate_psnr(a, a, 100, 100)


# ---------------------------------------------------------- XYB roundtrip (bit-exact) -------


def test_xyb_roundtrip_reference_cases(gpu_ctx, oracle):
    # src/metrics/xyb.rs:259-272
    rgb = (np.arange(64 * 64 * 3) % 256).astype(np.uint8)
    out = gpu_ctx.xyb_roundtrip(rgb, 64, 64)
    assert out.size == rgb.size
    rgb2 = ((np.arange(32 * 32 * 3) * 7) % 256).astype(np.uint8)
    a, b = gpu_ctx.xyb_roundtrip(rgb2, 32, 32), gpu_ctx.xyb_roundtrip(rgb2, 32, 32)
    assert np.array_equal(a, b)
    assert np.array_equal(out, oracle.xyb_roundtrip(rgb, 64, 64))


def test_xyb_roundtrip_all_colours_bit_exact(gpu_ctx, oracle):
    """Every sRGB colour (2^24 pixels): the function is pointwise, so this is exhaustive.
    Also re-derives the known-answer table of src/metrics/xyb.rs:15-24 from the device output."""
    v = np.arange(1 << 24, dtype=np.uint32)
    rgb = np.stack([(v >> 16) & 255, (v >> 8) & 255, v & 255], 1).astype(np.uint8)
    got = gpu_ctx.xyb_roundtrip(rgb, 4096, 4096).reshape(-1, 3)
    want = oracle.xyb_roundtrip(rgb, 4096, 4096).reshape(-1, 3)
    bad = np.flatnonzero((got != want).any(axis=1))
    assert bad.size == 0, f"{bad.size} colours differ, first: in={rgb[bad[0]]} gpu={got[bad[0]]} oracle={want[bad[0]]}"
    diff = np.abs(got.astype(np.int16) - rgb.astype(np.int16)).max(axis=1)
    n = float(diff.size)
    assert abs((diff == 0).sum() / n * 100 - 15.7) < 0.05
    assert abs((diff <= 1).sum() / n * 100 - 71.3) < 0.05
    assert abs((diff <= 2).sum() / n * 100 - 84.7) < 0.05
    assert abs((diff <= 5).sum() / n * 100 - 95.8) < 0.05
    assert abs((diff <= 10).sum() / n * 100 - 99.3) < 0.05
    assert diff.max() == 26


@pytest.mark.parametrize("w,h", [(1, 1), (2, 1), (3, 1), (5, 7), (101, 77)])
def test_xyb_roundtrip_ragged(gpu_ctx, oracle, w, h):
    rng = np.random.default_rng(w + 31 * h)
    rgb = rng.integers(0, 256, w * h * 3, dtype=np.uint8)
    assert np.array_equal(gpu_ctx.xyb_roundtrip(rgb, w, h), oracle.xyb_roundtrip(rgb, w, h))


# ------------------------------------------------------------------- SSIMULACRA2 -------------


def test_ssim2_reference_cases(gpu_ctx, oracle, ce):
    # src/metrics/ssimulacra2.rs:153-182
    d = ramp(100, 100)
    assert gpu_ctx.calculate_ssimulacra2(d, d, 100, 100) > 99.0
    r, t = np.full(30000, 100, np.uint8), np.full(30000, 200, np.uint8)
    got = gpu_ctx.calculate_ssimulacra2(r, t, 100, 100)
    assert got < 80.0
    assert rel_close(got, oracle.ssimulacra2(r, t, 100, 100, 1))
    small, large = np.full(50 * 50 * 3, 128, np.uint8), np.full(100 * 100 * 3, 128, np.uint8)
    with pytest.raises(ce.CodecEvalError):
        gpu_ctx.calculate_ssimulacra2(small, large, 100, 100)
    with pytest.raises(ce.MetricCalculation):  # ssimulacra2.rs:72-82
        gpu_ctx.calculate_ssimulacra2(small, small, 100, 100)
    with pytest.raises(ce.MetricCalculation):  # below 8x8
        gpu_ctx.calculate_ssimulacra2(small[: 7 * 7 * 3], small[: 7 * 7 * 3], 7, 7)


def test_ssim2_planes_bit_exact(gpu_ctx, oracle, ce, workloads):
    """Stage outputs of level 0 against the oracle's building blocks: every plane identical."""
    w, h = 200, 136
    ref = workloads.make_reference(w, h, 11)
    test = workloads.distort(ref, 70)
    lin_r, lin_t = oracle.ssim2_linear_planar(ref, w, h), oracle.ssim2_linear_planar(test, w, h)
    b = ce.Batch(gpu_ctx, w, h, 1, 1)
    b.set_reference(0, ref)
    b.set_test(0, 0, test)
    # level 0 (read straight from u8, no linear plane exists): XYB planes
    b.debug_limit_scales(1)
    b.run(1, ce.MetricConfig.ssimulacra2_only())
    assert np.array_equal(b.debug_planes(0, 2), oracle.ssim2_xyb_positive(lin_r))
    assert np.array_equal(b.debug_planes(0, 3), oracle.ssim2_xyb_positive(lin_t))
    # level 1: the 2x2 box of linear RGB, its XYB planes, and the five row-blurred streams of channel Y
    b.debug_limit_scales(2)
    b.run(1, ce.MetricConfig.ssimulacra2_only())
    lin1_r, lin1_t = oracle.ssim2_downscale(lin_r), oracle.ssim2_downscale(lin_t)
    assert np.array_equal(b.debug_planes(1, 0), lin1_r)
    assert np.array_equal(b.debug_planes(1, 1), lin1_t)
    x1, x2 = oracle.ssim2_xyb_positive(lin1_r), oracle.ssim2_xyb_positive(lin1_t)
    assert np.array_equal(b.debug_planes(1, 2), x1)
    assert np.array_equal(b.debug_planes(1, 3), x2)
    b.close()


SHAPES = [(8, 8), (9, 15), (16, 8), (33, 17), (64, 64), (100, 100), (101, 77), (255, 129), (768, 512), (512, 768)]


@pytest.mark.gpu
def test_ssim2_cbrt_fast_form_is_exact(gpu_ctx):
    """The division-free cube root of the XYB front end must give the reference form's f32 for EVERY
    positive normal input (0x00800000 .. 0x7f7fffff), and take the fallback only rarely in the
    range the front end feeds it (mixed absorbance + bias lies in [0.0037, ~1.1])."""
    first, last = 0x00800000, 0x7F7FFFFF
    mism, slow = gpu_ctx.debug_cbrt_sweep(first, last - first + 1)
    assert mism == 0
    lo, hi = 0x3B000000, 0x3FC00000  # [2^-9, 1.5)
    mism, slow = gpu_ctx.debug_cbrt_sweep(lo, hi - lo)
    assert mism == 0
    assert slow / (hi - lo) < 2.5e-4  # ~2^-14 expected
    # zero, subnormals, huge values, negatives and NaN patterns go through the reference form itself
    for a, n in ((0, 0x00800000), (0x7F000000, 0x01000000), (0x80000000, 0x01000000)):
        mism, slow = gpu_ctx.debug_cbrt_sweep(a, n)
        assert mism == 0


@pytest.mark.parametrize("w,h", SHAPES)
def test_ssim2_parity_shapes(gpu_ctx, oracle, ce, workloads, w, h):
    ref = workloads.make_reference(w, h, 100 + w)
    for q in (30, 75, 95):
        test = workloads.distort(ref, q)
        want, want_avg = oracle.ssimulacra2_detail(ref, test, w, h, 1)
        b = ce.Batch(gpu_ctx, w, h, 1, 1)
        b.set_reference(0, ref)
        b.set_test(0, 0, test)
        s = b.run(1, ce.MetricConfig.ssimulacra2_only())[0]
        got_avg = b.debug_averages(0)
        b.close()
        assert s.status == 0 and s.valid & ce.METRIC_SSIMULACRA2
        assert got_avg.shape == want_avg.shape
        # blurred planes are bit-identical; the per-pixel map terms are pooled from f32 (see k_ssim2_vblur_dma)
        np.testing.assert_allclose(got_avg, want_avg, rtol=2e-6, atol=1e-12)
        assert rel_close(s.ssimulacra2, want), (w, h, q, s.ssimulacra2, want)


def test_ssim2_edge_content(gpu_ctx, oracle, workloads):
    w, h = 96, 80
    flat = workloads.make_reference(w, h, 5, "flat")
    noise = workloads.make_reference(w, h, 6, "highfreq")
    for a, b in ((flat, flat), (noise, noise), (flat, noise), (noise, flat), (np.zeros_like(flat), np.full_like(flat, 255))):
        got = gpu_ctx.calculate_ssimulacra2(a, b, w, h)
        assert rel_close(got, oracle.ssimulacra2(a, b, w, h, 1)), got
    assert gpu_ctx.calculate_ssimulacra2(noise, noise, w, h) == 100.0


def test_helpers_mirror(gpu_ctx, ce):
    # src/eval/helpers.rs:337-383 (DSSIM / Butteraugli halves are added with those kernels)
    img, shifted = helper_pattern(64, 64, 0), helper_pattern(64, 64, 50)
    res = ce.evaluate_single(gpu_ctx, img, img, ce.MetricConfig(ssimulacra2=True, psnr=True))
    assert res.ssimulacra2 > 99.0 and math.isinf(res.psnr)
    with pytest.raises(ce.DimensionMismatch):
        ce.evaluate_single(gpu_ctx, img, helper_pattern(32, 32, 0), ce.MetricConfig.ssimulacra2_only())
    ce.assert_quality(gpu_ctx, img, img, 90.0, None)
    with pytest.raises(ce.QualityBelowThreshold):
        ce.assert_quality(gpu_ctx, img, shifted, 99.0, None)


# --------------------------------------------------------- batch / grid semantics --------------


def test_batch_grid_matches_single_calls(gpu_ctx, oracle, ce, workloads):
    """One resident grid (2 refs x 3 qualities, shared reference slots) == six single calls == oracle."""
    w, h = 128, 96
    refs = [workloads.make_reference(w, h, 40 + i) for i in range(2)]
    b = ce.Batch(gpu_ctx, w, h, 2, 6)
    tests = []
    for i, r in enumerate(refs):
        b.set_reference(i, r)
        for k, q in enumerate((50, 75, 90)):
            t = workloads.distort(r, q)
            tests.append((i, t))
            b.set_test(i * 3 + k, i, t)
    cfg = ce.MetricConfig(ssimulacra2=True, psnr=True)
    out = b.run(6, cfg)
    for (i, t), s in zip(tests, out):
        assert s.psnr == oracle.psnr(refs[i], t, w, h)
        assert rel_close(s.ssimulacra2, oracle.ssimulacra2(refs[i], t, w, h, 1))
        single = gpu_ctx.calculate_metrics(refs[i], t, w, h, cfg)
        assert single.ssimulacra2 == s.ssimulacra2 and single.psnr == s.psnr  # batching changes nothing
    b.close()


def test_eval_batch_mixed_shapes_and_errors(gpu_ctx, oracle, ce, workloads):
    a = workloads.make_reference(96, 64, 1)
    b_ = workloads.make_reference(64, 96, 2)
    ta, tb = workloads.distort(a, 60), workloads.distort(b_, 60)
    pairs = [(a, ta, 96, 64), (b_, tb, 64, 96), (a, tb[:10], 96, 64), (a, ta, 95, 64)]
    out = gpu_ctx.eval_batch(pairs, ce.MetricConfig(ssimulacra2=True, psnr=True))
    assert out[0].status == 0 and out[1].status == 0
    assert out[2].status == ce.CE_ERR_DIM_MISMATCH and out[3].status == ce.CE_ERR_BAD_LENGTH
    assert out[0].psnr == oracle.psnr(a, ta, 96, 64) and out[1].psnr == oracle.psnr(b_, tb, 64, 96)
    assert rel_close(out[1].ssimulacra2, oracle.ssimulacra2(b_, tb, 64, 96, 1))


def test_xyb_flag_applies_to_reference_only(gpu_ctx, oracle, ce, workloads):
    # session.rs:447-456: the reference is roundtripped, the test image is not
    w, h = 80, 72
    ref = workloads.make_reference(w, h, 3)
    test = workloads.distort(ref, 80)
    got = gpu_ctx.calculate_metrics(ref, test, w, h, ce.MetricConfig(ssimulacra2=True, psnr=True, xyb_roundtrip=True))
    rt = oracle.xyb_roundtrip(ref, w, h)
    assert got.psnr == oracle.psnr(rt, test, w, h)
    assert rel_close(got.ssimulacra2, oracle.ssimulacra2(rt, test, w, h, 1))


def test_reference_handle(gpu_ctx, oracle, ce, workloads):
    # Ssimulacra2Reference::{new, compare}, crates/codec-iter/src/eval.rs:138-149,83-89
    w, h = 112, 88
    ref = workloads.make_reference(w, h, 9)
    hd = ce.ReferenceHandle(gpu_ctx, ref, w, h)
    for q in (40, 70, 92):
        t = workloads.distort(ref, q)
        assert rel_close(hd.compare(t).ssimulacra2, oracle.ssimulacra2(ref, t, w, h, 1))
    with pytest.raises(ce.DimensionMismatch):
        hd.compare(ref[:10])
    # a distorted image in page-locked memory (ce_host_alloc) is read in place by the DMA engine - no staging copy - and may be
    # overwritten by the caller as soon as the (blocking) call returns: same scores as from ordinary memory, call after call
    pinned = gpu_ctx.host_buffer(w * h * 3)
    allm = ce.MetricConfig.all()
    for q in (35, 60, 85, 60):
        t = workloads.distort(ref, q)
        pinned[:] = t.reshape(-1)
        a, b = hd.compare(pinned, allm), hd.compare(t, allm)
        assert (a.psnr, a.ssimulacra2, a.dssim, a.butteraugli) == (b.psnr, b.ssimulacra2, b.dssim, b.butteraugli)
        many = hd.compare_many([pinned, t, pinned], allm)
        assert all((m.ssimulacra2, m.dssim, m.butteraugli) == (a.ssimulacra2, a.dssim, a.butteraugli) for m in many)
    hd.close()


def test_decoded_image_ingest_on_device(gpu_ctx, ce, workloads):
    """RGBA8 / 10-bit RGB16 / RGBA16 decoder outputs are converted to the RGB8 slab on the device; the result must be
    the bytes the reference's host passes produce (session.rs:98-117 alpha strip; avif_config.rs:122-125 to_8bit),
    restated here in numpy.  PSNR against the numpy-converted image is +inf exactly when every byte agrees."""
    w, h = 97, 61  # odd sizes: no alignment to lean on
    rng = np.random.default_rng(5)
    ref = workloads.make_reference(w, h, 11)
    rgb = workloads.distort(ref, 70).reshape(h, w, 3)
    rgba = np.concatenate([rgb, rng.integers(0, 256, (h, w, 1), dtype=np.uint8)], axis=2)
    v10 = rng.integers(0, 1024, (h, w, 3), dtype=np.uint16)
    v10[0, :8] = [[0, 1, 2], [3, 4, 510], [511, 512, 513], [1021, 1022, 1023], [1024, 2047, 65535], [2, 2, 2], [1000, 100, 10], [7, 77, 777]]
    rgb16_as8 = np.minimum((v10.astype(np.uint32) * 255 + 512) // 1023, 255).astype(np.uint8)
    rgba16 = np.concatenate([v10, rng.integers(0, 1024, (h, w, 1), dtype=np.uint16)], axis=2)
    cfg = ce.MetricConfig.all()
    cases = [(rgba, ce.PIXEL_RGBA8, rgb), (v10, ce.PIXEL_RGB16_10BIT, rgb16_as8), (rgba16, ce.PIXEL_RGBA16_10BIT, rgb16_as8),
             (rgb, ce.PIXEL_RGB8, rgb)]
    b = ce.Batch(gpu_ctx, w, h, 2, 2)
    for pixels, fmt, want in cases:
        # slot 0: reference = device-ingested image, test = numpy-converted image -> must be identical
        b.set_reference_fmt(0, pixels, fmt)
        b.set_test(0, 0, want)
        # slot 1: same test image both ways against a real reference -> same scores
        b.set_reference(1, ref)
        b.set_test_fmt(1, 1, pixels, fmt)
        s = b.run(2, cfg)
        assert s[0].psnr == float("inf") and s[0].dssim == 0.0 and s[0].ssimulacra2 == 100.0
        direct = gpu_ctx.calculate_metrics(ref, want, w, h, cfg)
        assert (s[1].psnr, s[1].ssimulacra2, s[1].dssim, s[1].butteraugli) == (direct.psnr, direct.ssimulacra2, direct.dssim, direct.butteraugli)
    with pytest.raises(ce.MetricCalculation):
        b.set_test_fmt(0, 0, rgba[:-1], ce.PIXEL_RGBA8)  # wrong length
    with pytest.raises(ce.CodecEvalError):
        b.set_test_fmt(0, 0, rgba, 9)  # unknown format
    b.close()


def test_codec_iter_plug_point(gpu_ctx, oracle, ce, workloads):
    # GpuSsim2::new/compute (gpu.rs:40-116) and Ssim2Backend::compare_with_precomputed (eval.rs:56-92)
    w, h = 72, 56
    ref = workloads.make_reference(w, h, 3)
    t = workloads.distort(ref, 60)
    gpu = ce.GpuSsim2(w, h)
    assert gpu.dimensions() == (w, h)
    want = oracle.ssimulacra2(ref, t, w, h, 1)
    assert rel_close(gpu.compute(ref, t), want)
    with pytest.raises(RuntimeError, match=r"Image size mismatch: expected 12096 bytes \(72x56x3\), got ref=12096 dis=10"):
        gpu.compute(ref, t.reshape(-1)[:10])
    backend = ce.Ssim2Backend(gpu)
    hd = ce.Ssimulacra2Reference(gpu.ctx, ref, w, h)
    assert backend.compare_with_precomputed(ref, t, None, "img", 60) == gpu.compute(ref, t)
    assert backend.compare_with_precomputed(ref, t, hd, "img", 60) == gpu.compute(ref, t)
    with pytest.raises(RuntimeError, match="SSIM2 error for img q60: "):
        backend.compare_with_precomputed(ref, t.reshape(-1)[:10], hd, "img", 60)
    hd.close()
    gpu.close()


def test_reference_handle_sweep_and_cached_reference_side(gpu_ctx, oracle, ce, workloads):
    """The quality sweep of one reference in one launch (`for q { reference.compare(..) }`,
    brute_force_sweep.rs:256), with the reference-side state (XYB roundtrip, XYB pyramid) built by the
    first compare and reused afterwards: later compares must give exactly what a fresh evaluation gives."""
    w, h = 96, 80
    ref = workloads.make_reference(w, h, 21)
    tests = [workloads.distort(ref, q) for q in (30, 55, 75, 90, 97)]
    cfg = ce.MetricConfig.all()
    for rt in (False, True):
        hd = ce.ReferenceHandle(gpu_ctx, ref, w, h, xyb_roundtrip=rt)
        assert hd.stats() == (0, 0, 0)
        first = hd.compare(tests[0], cfg)  # builds the reference side of every metric
        assert hd.stats() == (1, 1, 1)
        assert hd.compare(tests[1], cfg).dssim > 0 and hd.stats() == (1, 1, 1)  # ... once: this compare built nothing
        many = hd.compare_many(tests, cfg)  # grows the handle to 5 slots (a new batch: one rebuild), reuses nothing stale
        assert hd.stats() == (2, 2, 2)
        again = hd.compare_many(tests[::-1], cfg)[::-1]  # reference side cached: XYB roundtrip, SSIMULACRA2 XYB pyramid,
        assert hd.stats() == (2, 2, 2)                   # DSSIM img / mu / sq pyramid, Butteraugli PsychoImage
        full = cfg.with_xyb_roundtrip() if rt else cfg
        for t, m, a in zip(tests, many, again):
            fresh = gpu_ctx.calculate_metrics(ref, t, w, h, full)
            for name in ("psnr", "ssimulacra2", "dssim", "butteraugli"):
                assert getattr(m, name) == getattr(fresh, name), (rt, name)
                assert getattr(a, name) == getattr(fresh, name), (rt, name)
        assert first.ssimulacra2 == many[0].ssimulacra2
        # a wrong-sized item fails alone
        res = hd.compare_many([tests[1], ref[:7], tests[2]], cfg)
        assert isinstance(res[1], ce.DimensionMismatch)
        assert res[0].ssimulacra2 == many[1].ssimulacra2 and res[2].ssimulacra2 == many[2].ssimulacra2
        # Butteraugli's reference side depends on the intensity target: another target rebuilds it (and only it), and
        # the result equals a fresh evaluation with that target
        before = hd.stats()
        dim = hd.compare(tests[2], ce.MetricConfig(butteraugli=True), intensity_target=250.0)
        assert hd.stats() == (before[0], before[1], before[2] + 1)
        if not rt:
            assert dim.butteraugli == gpu_ctx.calculate_butteraugli_with_intensity(ref, tests[2], w, h, 250.0)
        back = hd.compare(tests[2], cfg)
        assert back.butteraugli == many[2].butteraugli and hd.stats()[2] == before[2] + 2
        hd.close()
    # against the oracle too (no roundtrip)
    hd = ce.ReferenceHandle(gpu_ctx, ref, w, h)
    for t, m in zip(tests, hd.compare_many(tests)):
        assert rel_close(m.ssimulacra2, oracle.ssimulacra2(ref, t, w, h, 1))
    hd.close()


# ------------------------------------------------------------------------ DSSIM ----------------


def test_dssim_reference_cases(gpu_ctx, oracle, ce):
    # src/metrics/dssim.rs:180-273 (the reference feeds linear RGBA f32; the ABI takes RGB8 and stages on device)
    img = gpu_ctx.rgb8_to_dssim_image(np.array([255, 0, 0, 0, 255, 0], np.uint8), 2, 1)
    assert img.shape == (1, 2, 4) and abs(img[0, 0, 0] - 1.0) < 0.001 and abs(img[0, 1, 1] - 1.0) < 0.001
    rng = np.random.default_rng(3)
    rgb = rng.integers(0, 256, 37 * 29 * 3, dtype=np.uint8)
    assert np.array_equal(gpu_ctx.rgb8_to_dssim_image(rgb, 37, 29), oracle.rgb8_to_dssim_image(rgb, 37, 29))  # bit-exact
    grey = np.full(100 * 100 * 3, 128, np.uint8)
    assert gpu_ctx.calculate_dssim(grey, grey, 100, 100) < 0.0001
    a, b = np.full(30000, 77, np.uint8), np.full(30000, 180, np.uint8)
    got = gpu_ctx.calculate_dssim(a, b, 100, 100)
    assert got > 0.0 and rel_close(got, oracle.dssim(a, b, 100, 100), floor=1e-6)
    with pytest.raises(ce.DimensionMismatch):
        gpu_ctx.calculate_dssim(np.zeros(50 * 50 * 3, np.uint8), grey, 100, 100)


DSSIM_SHAPES = [(1, 1), (3, 2), (7, 9), (8, 8), (15, 17), (20, 20), (64, 64), (100, 100), (101, 77), (255, 129), (768, 512), (512, 768),
                (33, 33), (65, 31), (32, 96),  # one pixel over / under the 32 x 32 tile (CE_DSSIM_*=tile kernels)
                # the streaming kernels' strips: 60 output columns per wave for a distorted image / a pair, 56 for a reference,
                # walks of 2 .. 64 rows - one column / row under, on and over their edges, and degenerate strips
                (59, 9), (60, 10), (61, 33), (62, 3), (119, 66), (120, 64), (121, 65), (56, 8), (57, 130), (112, 5), (113, 17),
                (300, 2), (2, 300), (1, 70), (70, 1), (4, 4), (5, 63)]


@pytest.mark.parametrize("w,h", DSSIM_SHAPES)
def test_dssim_parity_shapes(gpu_ctx, oracle, ce, workloads, w, h):
    ref = workloads.make_reference(w, h, 300 + w)
    for q in (30, 75, 95):
        test = workloads.distort(ref, q)
        want = oracle.dssim(ref, test, w, h)
        got = gpu_ctx.calculate_dssim(ref, test, w, h)
        # planes are bit-identical; only the f64 summation order differs
        assert abs(got - want) <= 1e-9 * max(abs(want), 1e-6), (w, h, q, got, want)
    assert gpu_ctx.calculate_dssim(ref, ref, w, h) == 0.0


def test_dssim_in_mixed_metric_batch(gpu_ctx, oracle, ce, workloads):
    w, h = 128, 96
    refs = [workloads.make_reference(w, h, 50 + i) for i in range(2)]
    pairs = [(refs[i], workloads.distort(refs[i], q), w, h) for i in range(2) for q in (40, 70, 90)]
    cfg = ce.MetricConfig(dssim=True, ssimulacra2=True, psnr=True)
    out = gpu_ctx.eval_batch(pairs, cfg)
    for (r, t, _, _), s in zip(pairs, out):
        assert s.status == 0 and s.valid == (ce.METRIC_DSSIM | ce.METRIC_SSIMULACRA2 | ce.METRIC_PSNR)
        assert s.psnr == oracle.psnr(r, t, w, h)
        assert rel_close(s.ssimulacra2, oracle.ssimulacra2(r, t, w, h, 1))
        assert rel_close(s.dssim, oracle.dssim(r, t, w, h), floor=1e-6)
    # xyb_roundtrip flag applies to the reference of every metric
    got = gpu_ctx.calculate_metrics(refs[0], pairs[0][1], w, h, ce.MetricConfig(dssim=True, xyb_roundtrip=True))
    assert rel_close(got.dssim, oracle.dssim(oracle.xyb_roundtrip(refs[0], w, h), pairs[0][1], w, h), floor=1e-6)
    assert got.perception_level() == ce.perception_from_dssim(got.dssim)


def test_helpers_mirror_dssim(gpu_ctx, ce):
    # src/eval/helpers.rs:337-383
    img, shifted = helper_pattern(64, 64, 0), helper_pattern(64, 64, 50)
    res = ce.evaluate_single(gpu_ctx, img, img, ce.MetricConfig(dssim=True, ssimulacra2=True))
    assert res.dssim < 0.0001 and res.ssimulacra2 > 99.0
    ce.assert_quality(gpu_ctx, img, img, 90.0, 0.001)
    ce.assert_perception_level(gpu_ctx, img, img, "Imperceptible")
    with pytest.raises(ce.QualityBelowThreshold):
        ce.assert_perception_level(gpu_ctx, img, shifted, "Imperceptible")
    with pytest.raises(ce.QualityBelowThreshold):
        ce.assert_quality(gpu_ctx, img, shifted, None, 0.0001)


# ---------------------------------------------------------- committed golden vectors ----------


def test_committed_golden_vectors_through_the_c_abi(gpu_ctx, ce):
    """tests/golden/{inputs.npz, scores.json} (made by tests/golden/make_golden.py with the oracle, committed): the HIP
    path against the FILES, with no oracle in the loop - integer / byte results exact, the three perceptual scores
    within the parity bar (they are in fact much closer: the device planes are bit-identical to the oracle's)."""
    import hashlib
    import json
    import os

    here = os.path.dirname(os.path.abspath(__file__))
    with open(os.path.join(here, "golden", "scores.json")) as f:
        scores = json.load(f)
    arrays = np.load(os.path.join(here, "golden", "inputs.npz"))
    assert len(scores) >= 8
    worst = {"ssimulacra2": 0.0, "dssim": 0.0, "butteraugli": 0.0}
    for name, s in scores.items():
        ref, test, w, h = arrays[name + ".ref"], arrays[name + ".test"], s["width"], s["height"]
        assert gpu_ctx.calculate_psnr(ref, test, w, h) == s["psnr"], name
        rt = gpu_ctx.xyb_roundtrip(ref, w, h)
        assert hashlib.sha256(np.asarray(rt).tobytes()).hexdigest() == s["xyb_roundtrip_sha256"], name
        m = gpu_ctx.calculate_metrics(ref, test, w, h, ce.MetricConfig.all())
        assert m.psnr == s["psnr"], name
        for key, got, floor in (("ssimulacra2", m.ssimulacra2, 1.0), ("dssim", m.dssim, 1e-3), ("butteraugli", m.butteraugli, 1.0)):
            assert rel_close(got, s[key], floor=floor), (name, key, got, s[key])
            worst[key] = max(worst[key], abs(got - s[key]) / max(abs(s[key]), floor))
        # the XYB-roundtripped reference (session.rs:447-456) against its own golden values
        mx = gpu_ctx.calculate_metrics(ref, test, w, h, ce.MetricConfig.all().with_xyb_roundtrip())
        assert mx.psnr == s["psnr_xyb_ref"], name
        assert rel_close(mx.ssimulacra2, s["ssimulacra2_xyb_ref"]), name
    assert worst["ssimulacra2"] < 1e-6 and worst["dssim"] < 1e-6 and worst["butteraugli"] < 1e-6, worst
