import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np
import codec_eval_amd as ce
n_refs, q = 18, 3
w, h = 768, 512
rng = np.random.default_rng(0)
ctx = ce.Context(0)
b = ce.Batch(ctx, w, h, n_refs, n_refs * q)
base = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
for i in range(n_refs):
    b.set_reference(i, np.roll(base, i * 7, axis=1))
    for k in range(q):
        t = np.clip(np.roll(base, i * 7, axis=1).astype(np.int16) + rng.integers(-3 * (k + 1), 3 * (k + 1) + 1, base.shape), 0, 255).astype(np.uint8)
        b.set_test(i * q + k, i, t)
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
for _ in range(steps):
    s = b.run(n_refs * q, ce.MetricConfig.ssimulacra2_only())
print("score0", s[0].ssimulacra2)
