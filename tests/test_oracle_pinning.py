"""CPU tests that pin the ORACLE (oracle/*.c) against everything the reference's own tests hold for
this path and against the committed golden vectors.

* PSNR and the XYB roundtrip are in-tree reference algorithms: the known answers below are real.
* SSIMULACRA2 / DSSIM / Butteraugli live in crates that are not in the reference tree: only the
  reference's inequalities exist — "parity unpinned" (DESIGN.md).
"""
import hashlib
import json
import math
import os

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


def ramp(w, h):
    return (np.arange(w * h * 3) % 256).astype(np.uint8)


def helper_pattern(w, h, pattern):  # src/eval/helpers.rs:327-335
    i = np.arange(w * h)
    base = (i + pattern) % 256
    return np.stack([base, base + 50, base + 100], 1).astype(np.uint8).reshape(-1)


# ---- PSNR: src/metrics/mod.rs:368-383 ---------------------------------------------------------
def test_psnr_reference_tests(oracle):
    same = np.full(100 * 100 * 3, 128, np.uint8)
    assert math.isinf(oracle.psnr(same, same, 100, 100))
    r, t = np.full(30000, 100, np.uint8), np.full(30000, 110, np.uint8)
    p = oracle.psnr(r, t, 100, 100)
    assert 28.0 < p < 29.0
    assert p == 10.0 * math.log10(255.0 * 255.0 / 100.0)  # the formula of mod.rs:329, bit for bit
    assert abs(p - 28.1308036086791) < 1e-12


def test_psnr_sum_is_exact_integer(oracle):
    rng = np.random.default_rng(0)
    a = rng.integers(0, 256, 4096 * 3, dtype=np.uint8)
    b = rng.integers(0, 256, 4096 * 3, dtype=np.uint8)
    sse = int(((a.astype(np.int64) - b.astype(np.int64)) ** 2).sum())
    assert oracle.sse(a, b) == sse
    assert oracle.psnr(a, b, 64, 64) == 10.0 * math.log10(255.0 * 255.0 / (sse / (4096 * 3)))


def test_psnr_length_errors(oracle):
    a, b = np.zeros(30, np.uint8), np.zeros(27, np.uint8)
    with pytest.raises(oracle.OracleError) as e:
        oracle.psnr(a, b, 10, 1)
    assert e.value.code == oracle.DIM_MISMATCH
    with pytest.raises(oracle.OracleError) as e:
        oracle.psnr(a, a, 3, 3)
    assert e.value.code == oracle.BAD_LENGTH


# ---- sRGB -> linear staging: src/metrics/dssim.rs:252-273 ----------------------------------------
def test_rgb8_to_dssim_image_reference_tests(oracle):
    img = oracle.rgb8_to_dssim_image(np.array([255, 0, 0, 0, 255, 0], np.uint8), 2, 1)
    assert img.shape == (1, 2, 4)
    assert abs(img[0, 0, 0] - 1.0) < 0.001 and abs(img[0, 1, 1] - 1.0) < 0.001
    assert img[0, 0, 3] == 1.0
    assert oracle.srgb_u8_to_linear(0) == 0.0
    assert abs(oracle.srgb_u8_to_linear(128) - 0.2158605) < 1e-6  # ((128/255+0.055)/1.055)^2.4


# ---- XYB roundtrip: src/metrics/xyb.rs:259-301 and the table of xyb.rs:15-24 ------------------------
def test_xyb_roundtrip_reference_tests(oracle):
    rgb = (np.arange(64 * 64 * 3) % 256).astype(np.uint8)
    assert oracle.xyb_roundtrip(rgb, 64, 64).size == rgb.size
    rgb2 = ((np.arange(32 * 32 * 3) * 7) % 256).astype(np.uint8)
    assert np.array_equal(oracle.xyb_roundtrip(rgb2, 32, 32), oracle.xyb_roundtrip(rgb2, 32, 32))
    g = np.arange(0, 256, 16, dtype=np.uint8)
    lattice = np.stack(np.meshgrid(g, g, g, indexing="ij"), -1).reshape(-1, 3)
    out = oracle.xyb_roundtrip(lattice, lattice.shape[0], 1).reshape(-1, 3)
    assert np.abs(out.astype(int) - lattice.astype(int)).max() <= 30


def test_xyb_roundtrip_known_answer_table(oracle):
    """xyb.rs:15-24: over all 2^24 sRGB colours — exact 15.7 %, <=1 71.3 %, <=2 84.7 %, <=5 95.8 %,
    <=10 99.3 %, max 26, MAE ~0.69.  The only quantitative known-answer set on the hot path."""
    v = np.arange(1 << 24, dtype=np.uint32)
    rgb = np.stack([(v >> 16) & 255, (v >> 8) & 255, v & 255], 1).astype(np.uint8)
    out = oracle.xyb_roundtrip(rgb, 4096, 4096).reshape(-1, 3)
    ad = np.abs(out.astype(np.int16) - rgb.astype(np.int16))
    diff = ad.max(axis=1)
    n = float(diff.size)
    pct = lambda k: (diff <= k).sum() / n * 100
    assert round(pct(0), 1) == 15.7
    assert round(pct(1), 1) == 71.3
    assert round(pct(2), 1) == 84.7
    assert round(pct(5), 1) == 95.8
    assert round(pct(10), 1) == 99.3
    assert diff.max() == 26
    assert abs(ad.mean() - 0.69) < 0.005
    worst = rgb[np.argmax(diff)]
    assert worst[0] > 200 and worst[1] > 200 and worst[2] < 100  # "bright saturated yellows"


def test_xyb_cbrtf_is_pinned_in_one_place(oracle):
    """xyb.rs:92-94 calls f32::cbrt = the platform libm's cbrtf, whose algorithm changed between glibc releases.
    Oracle and device both run ONE restatement (glibc 2.35's routine, the one that reproduces the table above).
    On a glibc-2.35 host the restatement must equal the host's cbrtf bit for bit; on a host with a different
    cbrtf the test reports the disagreement rate instead of failing (the pin, not the host, is the contract)."""
    import platform

    rng = np.random.default_rng(5)
    # the roundtrip's inputs: opsin values in [bias, ~1.004]; plus a wide log-uniform sweep of normal floats
    x = np.concatenate([rng.uniform(0.0037930733, 1.01, 1 << 20), np.exp(rng.uniform(-80, 80, 1 << 18))]).astype(np.float32)
    pinned, host = oracle.cbrtf_compare(x)
    differ = int((pinned.view(np.uint32) != host.view(np.uint32)).sum())
    ulps = np.abs(pinned.view(np.int32).astype(np.int64) - host.view(np.int32).astype(np.int64)).max()
    if platform.libc_ver()[1] == "2.35":
        assert differ == 0
    elif differ:
        assert ulps <= 1  # two faithful cube roots never differ by more than one ulp
        pytest.skip(f"host cbrtf ({platform.libc_ver()}) differs from the pinned glibc-2.35 routine on {differ} of {x.size} "
                    f"inputs (max {ulps} ulp); oracle and device both use the pin")


# ---- SSIMULACRA2: src/metrics/ssimulacra2.rs:153-182 (inequalities only) -----------------------------
@pytest.mark.parametrize("mode", [0, 1])
def test_ssimulacra2_reference_tests(oracle, mode):
    d = ramp(100, 100)
    assert oracle.ssimulacra2(d, d, 100, 100, mode) > 99.0
    r, t = np.full(30000, 100, np.uint8), np.full(30000, 200, np.uint8)
    assert oracle.ssimulacra2(r, t, 100, 100, mode) < 80.0
    small, large = np.full(50 * 50 * 3, 128, np.uint8), np.full(100 * 100 * 3, 128, np.uint8)
    with pytest.raises(oracle.OracleError):
        oracle.ssimulacra2(small, large, 100, 100, mode)


def test_ssimulacra2_blur_is_the_published_filter(oracle):
    """the recursion's exact impulse response is the 9-tap FIR h[0..4]."""
    t32, t64 = oracle.ssim2_blur_taps()
    want = [0.264621105488199, 0.212928592010422, 0.10933537277746, 0.0360111146583644, 0.00941436780965374]
    np.testing.assert_allclose(t64, want, rtol=1e-12)
    assert abs(t64[0] + 2 * t64[1:].sum() - 1.0) < 1e-12  # unit DC gain
    imp = np.zeros((31, 31), np.float32)
    imp[15, 15] = 1.0
    fir, iir = oracle.ssim2_blur_plane(imp, 0), oracle.ssim2_blur_plane(imp, 1)
    k = np.concatenate([t64[:0:-1], t64])
    np.testing.assert_allclose(fir[15, 11:20], k * t64[0], rtol=2e-6)
    np.testing.assert_allclose(iir, fir, atol=2e-7)  # same filter, f32 recursion noise only
    assert abs(fir.sum() - 1.0) < 1e-5


def test_ssimulacra2_scale_count_follows_lineage(oracle):
    """`if w < 8 || h < 8 {break}` is tested BEFORE halving, so 100x100 has 5 levels (100,50,25,13,7)."""
    d = ramp(100, 100)
    _, avg = oracle.ssimulacra2_detail(d, d, 100, 100, 1)
    assert avg.shape[0] == 5
    _, avg = oracle.ssimulacra2_detail(ramp(768, 512), ramp(768, 512), 768, 512, 1)
    assert avg.shape[0] == 6
    with pytest.raises(oracle.OracleError) as e:
        oracle.ssimulacra2(ramp(7, 9), ramp(7, 9), 7, 9)
    assert e.value.code == oracle.TOO_SMALL


def test_ssimulacra2_monotone_in_distortion(oracle, workloads):
    ref = workloads.make_reference(128, 96, 3)
    s = [oracle.ssimulacra2(ref, workloads.distort(ref, q), 128, 96, 1) for q in (20, 40, 60, 80, 95)]
    assert all(a < b for a, b in zip(s, s[1:])) and s[-1] < 100.0 and s[0] > -50.0


# ---- DSSIM: src/metrics/dssim.rs:180-250 (inequalities only) -------------------------------------------
def test_dssim_reference_tests(oracle):
    def flat(v, w=100, h=100):
        a = np.full((h, w, 4), v, np.float32)
        a[..., 3] = 1.0
        return a

    assert oracle.dssim_rgbaf(flat(0.5), flat(0.5)) < 0.0001
    assert oracle.dssim_rgbaf(flat(0.3), flat(0.7)) > 0.0
    with pytest.raises(oracle.OracleError) as e:
        oracle.dssim_rgbaf(flat(0.5, 50, 50), flat(0.5))
    assert e.value.code == oracle.DIM_MISMATCH


def test_dssim_monotone_and_scales(oracle, workloads):
    ref = workloads.make_reference(128, 96, 4)
    vals = [oracle.dssim(ref, workloads.distort(ref, q), 128, 96) for q in (20, 50, 80, 95)]
    assert all(a > b for a, b in zip(vals, vals[1:])) and vals[-1] > 0
    _, per_scale = oracle.dssim_detail(ref, workloads.distort(ref, 50), 128, 96)
    assert per_scale.size == 5 and np.all(per_scale <= 1.0)


# ---- Butteraugli: src/metrics/butteraugli.rs:168-207 (inequalities only) ----------------------------
def test_butteraugli_reference_tests(oracle):
    d = ramp(100, 100)
    assert oracle.butteraugli(d, d, 100, 100)[0] < 0.01
    assert oracle.butteraugli(d, d, 100, 100, 250.0)[0] < 0.01
    r, t = np.full(30000, 100, np.uint8), np.full(30000, 200, np.uint8)
    assert oracle.butteraugli(r, t, 100, 100)[0] > 1.0
    small, large = np.full(50 * 50 * 3, 128, np.uint8), np.full(100 * 100 * 3, 128, np.uint8)
    with pytest.raises(oracle.OracleError):
        oracle.butteraugli(small, large, 100, 100)


def test_butteraugli_monotone(oracle, workloads):
    ref = workloads.make_reference(128, 96, 5)
    vals = [oracle.butteraugli(ref, workloads.distort(ref, q), 128, 96) for q in (20, 50, 80, 95)]
    assert all(a[0] > b[0] for a, b in zip(vals, vals[1:]))
    assert all(0 < p3 <= mx for mx, p3 in vals)  # the 3-norm never exceeds the max-norm


# ---- helpers: src/eval/helpers.rs:337-383 -----------------------------------------------------------
def test_helpers_reference_tests(oracle):
    img, shifted = helper_pattern(64, 64, 0), helper_pattern(64, 64, 50)
    assert oracle.dssim(img, img, 64, 64) < 0.0001
    assert oracle.ssimulacra2(img, img, 64, 64, 1) > 99.0
    assert oracle.butteraugli(img, img, 64, 64)[0] < 0.1
    assert oracle.ssimulacra2(img, shifted, 64, 64, 1) < 99.0  # assert_quality(.., Some(99.0), None) must fail


# ---- committed golden vectors -------------------------------------------------------------------------
def _golden():
    with open(os.path.join(HERE, "golden", "scores.json")) as f:
        scores = json.load(f)
    arrays = np.load(os.path.join(HERE, "golden", "inputs.npz"))
    return scores, arrays


def test_golden_vectors(oracle):
    scores, arrays = _golden()
    assert len(scores) >= 8
    for name, s in scores.items():
        ref, test, w, h = arrays[name + ".ref"], arrays[name + ".test"], s["width"], s["height"]
        assert oracle.sse(ref, test) == s["sse"]
        assert oracle.psnr(ref, test, w, h) == s["psnr"]
        rt = oracle.xyb_roundtrip(ref, w, h)
        assert hashlib.sha256(rt.tobytes()).hexdigest() == s["xyb_roundtrip_sha256"]
        assert oracle.psnr(rt, test, w, h) == s["psnr_xyb_ref"]
        # floating-point metrics: same libm / compiler on the GPU box, but allow last-digit drift
        assert abs(oracle.ssimulacra2(ref, test, w, h, 1) - s["ssimulacra2"]) <= 1e-9 * max(1, abs(s["ssimulacra2"]))
        assert abs(oracle.ssimulacra2(ref, test, w, h, 0) - s["ssimulacra2_fir"]) <= 1e-9 * max(1, abs(s["ssimulacra2_fir"]))
        assert abs(oracle.dssim(ref, test, w, h) - s["dssim"]) <= 1e-9 * max(1e-3, abs(s["dssim"]))
        ba, ba3 = oracle.butteraugli(ref, test, w, h)
        assert abs(ba - s["butteraugli"]) <= 1e-6 * max(1, ba) and abs(ba3 - s["butteraugli_3norm"]) <= 1e-6 * max(1, ba3)
