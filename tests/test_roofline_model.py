"""codec-eval_amd/roofline.py: the per-kernel algorithmic byte counts bench.py prices kernels with.

Pinned against SURVEY.md §8(d): per-kernel compulsory bytes (fused stages are free, reference-side inputs once per
reference) must never EXCEED SURVEY's per-stage sums for an uncached pair (210 / 238 / 826 B per scale-0 pixel), and
the level geometry must match what the kernels' host code allocates."""
import importlib

rf = importlib.import_module("codec-eval_amd.roofline")


def test_level_geometry():
    assert rf.ssim2_levels(768, 512) == [(768, 512), (384, 256), (192, 128), (96, 64), (48, 32), (24, 16)]
    assert rf.ssim2_levels(15, 9) == [(15, 9), (8, 5)]  # a level below 8 px exists while its parent is >= 8
    assert rf.ssim2_levels(7, 100) == []
    assert rf.dssim_levels(512, 512) == [(512, 512), (256, 256), (128, 128), (64, 64), (32, 32)]
    assert rf.dssim_levels(20, 9) == [(20, 9), (10, 4)]
    assert rf.butteraugli_levels(3840, 2160) == [(3840, 2160), (1920, 1080)]
    assert rf.butteraugli_levels(15, 15) == [(15, 15), (8, 8)]
    assert rf.butteraugli_levels(14, 14) == [(14, 14)]


def per_px0(metric, n_refs, n_pairs, w=768, h=512):
    b = rf.Bucket(w, h, n_refs, n_pairs)
    return sum(rf.step_bytes([b], [metric]).values()) / (n_pairs * w * h)


def test_uncached_pair_counts_stay_below_the_survey_sums():
    # one reference, one distorted image: the case SURVEY.md §8(d) counts
    s2, ds, ba = per_px0("ssimulacra2", 1, 1), per_px0("dssim", 1, 1), per_px0("butteraugli", 1, 1)
    assert abs(s2 - 210) < 1.0, s2   # SURVEY: 144 N0 + 198 N_s - 30 N_last = 210
    assert 150 <= ds <= 238, ds      # SURVEY: 238
    assert 500 <= ba <= 826, ba      # SURVEY: 826 (counts every blur pass of the fused front end / splits separately)


def test_reference_sharing_lowers_the_count():
    for m in ("ssimulacra2", "dssim", "butteraugli"):
        assert per_px0(m, 1, 8) < per_px0(m, 1, 3) < per_px0(m, 1, 1)
    # SSIMULACRA2 with three distorted images per reference: the two reference-only streams of both passes are shared
    assert abs((per_px0("ssimulacra2", 1, 1) - per_px0("ssimulacra2", 1, 3)) - (2 * 2 * 12 * (2 / 3) * 1.333 + 2 * 18 * (2 / 3) * 1.0)) < 12


def test_step_bytes_names_are_launch_names():
    b = rf.Bucket(512, 512, 2, 6)
    acc = rf.step_bytes([b], ["ssimulacra2", "dssim", "butteraugli", "psnr"], xyb_roundtrip=True)
    expected = {"ssim2_prep_u8", "ssim2_prep", "ssim2_hblur_L0", "ssim2_vblur_ssim_L0", "ssim2_hblur_L1-5", "ssim2_vblur_ssim_L1-5",
                "dssim_create_u8", "dssim_create", "dssim_compare", "dssim_absdev", "ba_front_u8", "ba_front_half",
                "ba_blur_h33", "ba_blur_v_lf", "ba_blur_hv_mf", "ba_blur_hv_hf", "ba_malta_l2",
                "ba_blur_hv_mask", "ba_mask_vals", "psnr_sse", "xyb_roundtrip"}
    assert set(acc) == expected
    assert all(v > 0 for v in acc.values())
    assert [rf.metric_of(k) for k in ("ssim2_prep", "dssim_compare", "ba_malta_l2", "psnr_sse", "xyb_roundtrip")] == \
           ["ssimulacra2", "dssim", "butteraugli", "psnr", "xyb_roundtrip"]
    # the launch names in the sources are exactly these
    import os, re
    src = ""
    root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "codec-eval_amd", "csrc")
    for f in ("ssim2.hip", "dssim.hip", "dssim_stream.hip", "butteraugli.hip", "psnr.hip", "xyb.hip"):
        src += open(os.path.join(root, f)).read()
    names = set(re.findall(r'CE_LAUNCH(?:_ON)?\(ctx,(?: \w+,)? "([a-z0-9_\-A-Z]+)"', src)) | {"ba_blur_h13", "ba_blur_v13"}
    names |= set(re.findall(r'CE_CREATE_LAUNCH\("([a-z0-9_]+)"', src))  # dssim_stream.hip's launch macro
    assert expected <= names, expected - names


def test_round2_model_is_frozen():
    """`pipeline_frac_r02_model` prices every round's step against round 2's byte model.  The frozen table
    must reproduce BENCH_r02's 16.56 GB per step of the Kodak grid (72 pairs, three metrics) to the byte, and config 5's."""
    kodak = [rf.Bucket(768, 512, 18, 54), rf.Bucket(512, 768, 6, 18)]
    three = ["ssimulacra2", "dssim", "butteraugli"]
    assert rf.step_bytes_r02_model(kodak, three) == 16557613056.0
    # while the current kernels' model is unchanged it equals the frozen one, metric by metric
    for m in three:
        assert rf.step_bytes_r02_model(kodak, [m]) == sum(rf.step_bytes(kodak, [m]).values())
    dense = [rf.Bucket(512, 512, 15, 1500)]
    allm = ["ssimulacra2", "dssim", "butteraugli", "psnr"]
    assert rf.step_bytes_r02_model(dense, allm, True) == sum(rf.step_bytes(dense, allm, True).values())
