"""EvalSession mirror (src/eval/session.rs:281-584) on the device: the (codec x quality) sweep of an image or a
corpus is scored as one batch per shape and must equal the per-pair leaf calls, in the reference's loop order;
reports come out in the reference's JSON / CSV layout."""
import importlib
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

S = importlib.import_module("codec-eval_amd.session")
R = importlib.import_module("codec-eval_amd.reports")


def quantiser_codec(keep_alpha=False):
    """A toy codec: 'encode' packs (w, h, step, pixels), 'decode' quantises with a step that shrinks as quality grows."""

    def encode(image, request):
        step = 1 + int((100.0 - request.quality) / 8.0)
        request.with_param("step", str(step))
        rgb = image.to_rgb8_vec()
        head = np.array([image.width, image.height, step], dtype=np.uint32).tobytes()
        return head + rgb.tobytes()[: max(16, rgb.size // (1 + step))] + rgb.tobytes()

    def decode(blob):
        w, h, step = np.frombuffer(blob[:12], dtype=np.uint32)
        rgb = np.frombuffer(blob[-int(w) * int(h) * 3:], dtype=np.uint8)
        q = np.minimum(255, (rgb // step) * step + step // 2).astype(np.uint8)
        if keep_alpha:  # a decoder that hands back RGBA: alpha is dropped on the device
            rgba = np.concatenate([q.reshape(-1, 3), np.full((q.size // 3, 1), 200, np.uint8)], axis=1)
            return S.ImageData.rgba(rgba, int(w), int(h))
        return S.ImageData.rgb(q, int(w), int(h))

    return encode, decode


def test_evaluate_image_matches_leaf_calls(gpu_ctx, ce, workloads, tmp_path):
    cfg = S.EvalConfig.builder().report_dir(tmp_path / "rep").metrics(ce.MetricConfig.all()).quality_levels([50, 75, 95]).build()
    ses = S.EvalSession(cfg, ctx=gpu_ctx)
    enc, dec = quantiser_codec()
    enc_a, dec_a = quantiser_codec(keep_alpha=True)
    ses.add_codec_with_decode("toy", "1.0", enc, dec).add_codec_with_decode("toy-rgba", "1.1", enc_a, dec_a).add_codec("size-only", "0.1", enc)
    assert ses.codec_count() == 3
    w, h = 96, 80
    src = workloads.make_reference(w, h, 7)
    rep = ses.evaluate_image("pattern.png", S.ImageData.rgb(src, w, h))
    assert [r.codec_id for r in rep.results] == ["toy"] * 3 + ["toy-rgba"] * 3 + ["size-only"] * 3
    assert [r.quality for r in rep.results] == [50.0, 75.0, 95.0] * 3 and rep.uncompressed_size == w * h * 3
    for r in rep.results[:6]:
        decoded = dec(enc(S.ImageData.rgb(src, w, h), S.EncodeRequest(r.quality)))
        m = gpu_ctx.calculate_metrics(src, decoded.data, w, h, ce.MetricConfig.all())
        assert (r.psnr, r.ssimulacra2, r.dssim, r.butteraugli) == (m.psnr, m.ssimulacra2, m.dssim, m.butteraugli)
        assert r.perception == m.perception_level() and r.decode_time_ms is not None and r.codec_params == {"step": str(1 + int((100 - r.quality) / 8))}
        assert r.bits_per_pixel == r.file_size * 8 / (float(w) * float(h))
    assert rep.results[0].ssimulacra2 < rep.results[2].ssimulacra2
    for r in rep.results[6:]:  # no decoder: size only, session.rs:411-428
        assert r.psnr is None and r.perception is None and r.decode_time_ms is None
    path = ses.write_image_report(rep)
    d = json.load(open(path))
    assert d["name"] == "pattern.png" and len(d["results"]) == 9 and d["results"][0]["metrics"]["ssimulacra2"] == rep.results[0].ssimulacra2
    # an RGBA source image is accepted too (ImageData::RgbaSlice)
    rgba = np.concatenate([src.reshape(-1, 3), np.full((w * h, 1), 255, np.uint8)], axis=1)
    rep2 = ses.evaluate_image("pattern-rgba.png", S.ImageData.rgba(rgba, w, h))
    assert [r.ssimulacra2 for r in rep2.results] == [r.ssimulacra2 for r in rep.results]


def test_evaluate_corpus_buckets_shapes_and_writes_csv(gpu_ctx, ce, workloads, tmp_path):
    cfg = S.EvalConfig.builder().report_dir(tmp_path / "rep").metrics(ce.MetricConfig.ssimulacra2_only()).quality_levels([60, 90]).build()
    ses = S.EvalSession(cfg, ctx=gpu_ctx)
    enc, dec = quantiser_codec()
    ses.add_codec_with_decode("toy", "1.0", enc, dec)
    images = [(f"img{i}.png", S.ImageData.rgb(workloads.make_reference(w, h, 30 + i), w, h)) for i, (w, h) in
              enumerate([(64, 48), (48, 64), (64, 48), (40, 40)])]
    corpus = ses.evaluate_corpus("mini", images)
    assert corpus.total_results() == 8 and [im.name for im in corpus.images] == [n for n, _ in images]
    for (n, im), rep in zip(images, corpus.images):
        one = ses.evaluate_image(n, im)
        assert [r.ssimulacra2 for r in rep.results] == [r.ssimulacra2 for r in one.results]
    jpath, cpath = ses.write_corpus_report(corpus)
    lines = open(cpath).read().splitlines()
    assert lines[0].startswith("image,codec,version,quality,") and len(lines) == 9
    f = lines[1].split(",")
    assert f[0] == "img0.png" and f[3] == "60" and f[9] == format(corpus.images[0].results[0].ssimulacra2, ".2f") and f[8] == "" and f[12] == ""
    # two-rank partition by reference: the union of the ranks' reports is the single-rank report
    parts = [ses.evaluate_corpus("mini", images, rank=r, world=2) for r in range(2)]
    got = {im.name: [x.ssimulacra2 for x in im.results] for p in parts for im in p.images}
    assert got == {im.name: [x.ssimulacra2 for x in im.results] for im in corpus.images}
    assert sorted(len(p.images) for p in parts) == [2, 2]


def test_session_errors(gpu_ctx, ce, workloads, tmp_path):
    with pytest.raises(ValueError, match="report_dir is required"):
        S.EvalConfig.builder().build()
    cfg = S.EvalConfig.builder().report_dir(tmp_path).build()
    assert cfg.quality_levels == [50.0, 60.0, 70.0, 80.0, 85.0, 90.0, 95.0] and cfg.metrics.mask == ce.MetricConfig.all().mask
    ses = S.EvalSession(cfg, ctx=gpu_ctx)
    src = workloads.make_reference(32, 32, 1)
    # the SOURCE image's profile is never applied (session.rs:373 takes to_rgb8_vec()); a tagged DECODED image needs colour
    # management (session.rs:394, to_rgb8_srgb) and without one fails like a build without the `icc` feature
    assert ses.evaluate_image("x", S.ImageData.rgb_with_icc(src, 32, 32, b"fake")).results == []
    tagged = S.EvalSession(cfg, ctx=gpu_ctx)
    tagged.add_codec_with_decode("tagging", "0", lambda im, rq: b"x", lambda blob: S.ImageData.rgb_with_icc(src, 32, 32, b"fake"))
    with pytest.raises(ce.MetricCalculation, match="ICC profile support requires the 'icc' feature"):
        tagged.evaluate_image("x", S.ImageData.rgb(src, 32, 32))
    ses.add_codec_with_decode("bad", "0", lambda im, rq: b"x", lambda blob: S.ImageData.rgb(np.zeros(16 * 16 * 3, np.uint8), 16, 16))
    with pytest.raises(ce.DimensionMismatch):
        ses.evaluate_image("x", S.ImageData.rgb(src, 32, 32))


def fake_cms(profile: bytes, rgb: np.ndarray) -> np.ndarray:
    """A stand-in for the host's colour management (the reference's is moxcms, icc.rs:69-103): a wide-gamut-like
    matrix + tone curve whose strength depends on the profile bytes, 8-bit RGB in, 8-bit RGB out, pure per pixel."""
    k = 1.0 + (len(profile) % 5) * 0.05
    lin = (rgb.astype(np.float32) / 255.0) ** np.float32(2.2 * k)
    m = np.array([[1.22, -0.17, -0.05], [-0.04, 1.09, -0.05], [-0.02, -0.08, 1.10]], np.float32)
    out = np.clip(lin @ m.T, 0.0, 1.0) ** np.float32(1.0 / 2.2)
    return np.rint(out * 255.0).astype(np.uint8)


def test_icc_profiles_of_decoded_images_are_applied_on_the_device_exactly(gpu_ctx, ce, workloads, tmp_path):
    """transform_to_srgb (icc.rs:69-103) is an 8-bit RGB -> 8-bit RGB function, so the table of the host CMS's outputs on
    all 2^24 colours reproduces it bit for bit.  The session builds one table per distinct profile, keeps it on the device
    and applies it to every decoded image tagged with that profile; scores must equal those of pixels transformed on the
    host with the same CMS."""
    w, h = 80, 56
    src = workloads.make_reference(w, h, 31)
    table = fake_cms(b"P3-like", ce.ColorTable.identity_cube())
    # the table route itself: device output == direct evaluation of the CMS, byte for byte
    lut = ce.ColorTable(gpu_ctx, table)
    b = ce.Batch(gpu_ctx, w, h, 1, 2)
    dec = workloads.distort(src, 70)
    rgba = np.concatenate([dec.reshape(-1, 3), np.full((w * h, 1), 9, np.uint8)], axis=1)
    b.set_reference_lut(0, fake_cms(b"P3-like", dec.reshape(-1, 3)), ce.PIXEL_RGB8, None)  # CMS on the host, no table
    b.set_test_lut(0, 0, dec, ce.PIXEL_RGB8, lut)                                           # CMS as a device table
    b.set_test_lut(1, 0, rgba, ce.PIXEL_RGBA8, lut)                                         # ... behind the RGBA strip
    s = b.run(2, ce.MetricConfig.all())
    assert all(x.psnr == float("inf") and x.dssim == 0.0 and x.ssimulacra2 == 100.0 and x.butteraugli == 0.0 for x in s)
    b.close()
    lut.close()
    with pytest.raises(ce.MetricCalculation):
        ce.ColorTable(gpu_ctx, table[:100])  # not a complete table
    # the session: two profiles, tagged and untagged decoders
    calls = []

    def cms(profile, rgb):
        calls.append((profile, rgb.shape[0]))
        return fake_cms(profile, rgb)

    cfg = S.EvalConfig.builder().report_dir(tmp_path / "rep").metrics(ce.MetricConfig.all()).quality_levels([45, 80]).build()
    ses = S.EvalSession(cfg, ctx=gpu_ctx, cms=cms)
    enc = lambda im, rq: np.array([rq.quality], np.float32).tobytes()
    mk = lambda profile: (lambda blob: (S.ImageData.rgb_with_icc if profile else (lambda d, w_, h_, p: S.ImageData.rgb(d, w_, h_)))(
        workloads.distort(src, float(np.frombuffer(blob, np.float32)[0])), w, h, profile))
    ses.add_codec_with_decode("tag-a", "1", enc, mk(b"profile-A"))
    ses.add_codec_with_decode("tag-b", "1", enc, mk(b"profile-BB"))
    ses.add_codec_with_decode("plain", "1", enc, mk(None))
    rep = ses.evaluate_image("x.png", S.ImageData.rgb_with_icc(src, w, h, b"source-profile-is-ignored"))
    assert len(rep.results) == 6
    assert sorted(c[0] for c in calls) == [b"profile-A", b"profile-BB"] and all(c[1] == 1 << 24 for c in calls)  # once per profile
    for r, profile in zip(rep.results, [b"profile-A"] * 2 + [b"profile-BB"] * 2 + [None] * 2):
        dec = workloads.distort(src, r.quality)
        want_px = fake_cms(profile, dec.reshape(-1, 3)) if profile else dec
        m = gpu_ctx.calculate_metrics(src, want_px, w, h, ce.MetricConfig.all())
        assert (r.psnr, r.ssimulacra2, r.dssim, r.butteraugli) == (m.psnr, m.ssimulacra2, m.dssim, m.butteraugli), (r.codec_id, r.quality)
        # ImageData.to_rgb8_srgb with the same cms is the host route of the same transform
        img = S.ImageData.rgb_with_icc(dec, w, h, profile) if profile else S.ImageData.rgb(dec, w, h)
        assert np.array_equal(img.to_rgb8_srgb(fake_cms), np.asarray(want_px).reshape(-1))
    assert rep.results[0].psnr != rep.results[4].psnr  # the transform really changed the pixels
    ses.evaluate_image("again.png", S.ImageData.rgb(src, w, h))
    assert len(calls) == 2  # tables are cached per profile for the session's lifetime
    ses.close()
