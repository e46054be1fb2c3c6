"""The program the rocprofv3 PMC passes run (one kernel at a time under --pmc; see profiles/README.md).

    python3 profiles/prof_step.py <config> <steps> <sidecar.json>

config 0: 54 pairs of 768x512 (18 references x 3 qualities), SSIMULACRA2 + DSSIM + Butteraugli  (the default bench bucket)
config 2: the same grid, SSIMULACRA2 only
config 3: 2 pairs of 3840x2160, Butteraugli
config 4: 16 references 512x512 x 8 qualities, SSIMULACRA2 + DSSIM
config 5: 4 references 512x512 x 25 qualities x {4:4:4, 4:2:0}, all four metrics, XYB roundtrip on
config 6: 16 references 512x512 x 8 qualities, SSIMULACRA2 + DSSIM + Butteraugli  (the shape of the default sweep's CID22 chunks)

Every run starts with the known-byte-count calibration streams (ce_debug_calibrate_traffic, 256 MiB): make_traffic.py
derives FETCH_SIZE's / WRITE_SIZE's correction factor per access width from them, in the same pass.
The sidecar records the launch geometry make_traffic.py normalises by."""
import importlib
import json
import os
import sys

sys.path.insert(0, os.getcwd())
import numpy as np

import codec_eval_amd as ce

wl = importlib.import_module("codec-eval_amd.workloads")
config = int(sys.argv[1]) if len(sys.argv) > 1 else 0
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 2
sidecar = sys.argv[3] if len(sys.argv) > 3 else None
CALIB_BYTES = 256 << 20

rng = np.random.default_rng(0)


def cheap_grid(w, h, n_refs, q):
    """Inputs whose content is irrelevant to the byte counts: shifted noise + per-quality noise (fast to make)."""
    base = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
    refs, pairs = [], []
    for i in range(n_refs):
        r = np.roll(base, i * 7, axis=1)
        refs.append(r)
        for k in range(q):
            pairs.append((i, np.clip(r.astype(np.int16) + rng.integers(-3 * (k % 8 + 1), 3 * (k % 8 + 1) + 1, base.shape), 0, 255).astype(np.uint8)))
    return refs, pairs


if config in (0, 2):
    w, h, (refs, pairs) = 768, 512, cheap_grid(768, 512, 18, 3)
    cfg = ce.MetricConfig.perceptual() if config == 0 else ce.MetricConfig.ssimulacra2_only()
elif config == 3:
    w, h, (refs, pairs) = 3840, 2160, cheap_grid(3840, 2160, 2, 1)
    cfg = ce.MetricConfig(butteraugli=True)
elif config == 4:
    w, h, (refs, pairs) = 512, 512, cheap_grid(512, 512, 16, 8)
    cfg = ce.MetricConfig(ssimulacra2=True, dssim=True)
elif config == 6:
    w, h, (refs, pairs) = 512, 512, cheap_grid(512, 512, 16, 8)
    cfg = ce.MetricConfig.perceptual()
else:
    w, h, (refs, pairs) = 512, 512, cheap_grid(512, 512, 4, 50)
    cfg = ce.MetricConfig.all().with_xyb_roundtrip()

ctx = ce.Context(0)
ctx.debug_calibrate_traffic(CALIB_BYTES)
b = ce.Batch(ctx, w, h, len(refs), len(pairs))
for i, r in enumerate(refs):
    b.set_reference(i, r)
for k, (ri, t) in enumerate(pairs):
    b.set_test(k, ri, t)
for _ in range(steps):
    s = b.run(len(pairs), cfg)
if sidecar:
    with open(sidecar, "w") as f:
        json.dump({"config": config, "width": w, "height": h, "n_refs": len(refs), "n_pairs": len(pairs), "steps": steps,
                   "metrics": [m for m in ("dssim", "ssimulacra2", "butteraugli", "psnr") if getattr(cfg, m)],
                   "xyb_roundtrip": cfg.xyb_roundtrip, "calib_bytes": CALIB_BYTES}, f)
print("scores[0]", s[0].ssimulacra2, s[0].dssim, s[0].butteraugli, s[0].psnr)
