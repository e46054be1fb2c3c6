"""Score sensitivity of every assumption the oracle makes about the absent crates (DESIGN.md §2 ledger).

The SSIMULACRA2 / DSSIM / Butteraugli restatements (oracle/*.c) follow the published algorithms, but fast-ssim2 0.8.0,
dssim-core 3.4.0 and butteraugli 0.9.0 could order a few floating-point operations differently.  For each such choice
the oracle has a switch (ce_oracle.h, ceo_set_variant) that flips ONLY that choice; this script measures how far the
score moves, relative to the 1e-4 parity bar, on the committed golden inputs plus one 768x512 pair, and writes
tests/golden/sensitivity.json.  A failing crate pin (tests/test_crate_pin.py) therefore points at a line.

Run:  python tests/golden/sensitivity.py
"""
import importlib
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import oracle as O  # noqa: E402

wl = importlib.import_module("codec-eval_amd.workloads")

# variant -> (metric it touches, what the default restatement does, what the variant does)
LEDGER = {
    "ssim2_blur_fir": ("ssimulacra2", "f32 recursive Gaussian, lineage operation order with mul_add (blur_mode 1)",
                       "exact impulse response as a 9-tap FIR (blur_mode 0)"),
    "ssim2_iir_no_fma": ("ssimulacra2", "recursion steps use fused multiply-add", "separate multiply and add"),
    "ssim2_srgb_f32_powf": ("ssimulacra2", "sRGB->linear table: f64 pow rounded once to f32", "f32 powf per code"),
    "ssim2_host_cbrtf": ("ssimulacra2", "cube root: msun bit-trick seed + two f64 Halley steps", "host libm cbrtf"),
    "ssim2_f32_pool": ("ssimulacra2", "map terms widened to f64 before 1 - ratio and the powers", "f32 terms, f64 sums (what the device does)"),
    "dssim_lab_no_fma": ("dssim", "RGB->XYZ and the a/b affine steps use mul_add", "separate multiply and add"),
    "dssim_f32_final": ("dssim", "scale weighting and 1/ssim - 1 in f64", "in f32, widened by f64::from (dssim.rs:70)"),
    "ba_malta_f32": ("butteraugli", "Malta asymmetry term in f64 (as libjxl)", "in f32"),
    "ba_libm_log2": ("butteraugli", "Gamma() uses the lineage's FastLog2f", "libm log2f"),
    "ba_blur_fma": ("butteraugli", "taps of the long separable blurs: multiply, then add (two roundings, lineage)",
                    "one fused multiply-add per tap (measured on the device in round 3: +3 % throughput; NOT adopted - see the soak run in profiles/r03_experiments.md section 3)"),
    "ba_l2_early": ("butteraugli", "L2DiffAsymmetric(hf) / L2Diff(mf) of X and Y join block_diff_ac after the three Malta bands (lineage call order)",
                    "between the bands: uhf, L2asym(hf), hf, L2(mf), mf - the same in-place accumulations in the order the device's fused kernel meets the bands (what the DEVICE does, with ba_malta_f32)"),
}
FLOOR = {"ssimulacra2": 1.0, "dssim": 1e-6, "butteraugli": 1e-3}


def score(metric, ref, test, w, h, blur_mode=1):
    if metric == "ssimulacra2":
        return O.ssimulacra2(ref, test, w, h, blur_mode)
    if metric == "dssim":
        return O.dssim(ref, test, w, h)
    return O.butteraugli(ref, test, w, h)[0]


def main():
    d = np.load(os.path.join(HERE, "inputs.npz"))
    cases = []
    for name in sorted({k.rsplit(".", 1)[0] for k in d.files}):
        ref, test = d[name + ".ref"], d[name + ".test"]
        cases.append((name, ref, test, ref.shape[1], ref.shape[0]))
    big = wl.make_reference(768, 512, 1000)
    cases.append(("kodak768x512_q85", big, wl.distort(big, 85), 768, 512))
    flat = wl.make_reference(256, 256, 5, "flat")
    noisy = np.clip(flat.astype(np.int16) + np.random.default_rng(3).integers(-2, 3, flat.shape), 0, 255).astype(np.uint8)
    cases.append(("flat256_noise2", flat, noisy, 256, 256))
    out = {}
    for variant, (metric, default, alt) in LEDGER.items():
        rows = {}
        for name, ref, test, w, h in cases:
            assert O.variants_all_default()
            base = score(metric, ref, test, w, h)
            if variant == "ssim2_blur_fir":
                got = score(metric, ref, test, w, h, blur_mode=0)
            else:
                O.set_variant(variant, 1)
                try:
                    got = score(metric, ref, test, w, h)
                finally:
                    O.set_variant(variant, 0)
            rows[name] = {"default": base, "variant": got, "rel": abs(got - base) / max(abs(base), FLOOR[metric])}
        worst = max(rows.values(), key=lambda r: r["rel"])
        out[variant] = {"metric": metric, "default": default, "variant": alt, "max_rel": worst["rel"],
                        "over_the_1e-4_bar": worst["rel"] > 1e-4, "cases": rows}
        print(f"{variant:22s} {metric:12s} max rel {worst['rel']:.3e}")
    with open(os.path.join(HERE, "sensitivity.json"), "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
