#!/usr/bin/env python3
"""Condense the rocprofv3 outputs of a profiling call into the files committed under profiles/.

usage: python3 profiles/make_traffic.py <tag> <fetch_dir> <write_dir> <sidecar.json> [<stats_dir> [<stats_tag>]]

  <fetch_dir>  output of  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d <fetch_dir> -- python3 profiles/prof_step.py <config> 2 <sidecar.json>
  <write_dir>  the same with --pmc WRITE_SIZE (a separate pass: the two TCC counters do not share one)
  <stats_dir>  optional: output of  rocprofv3 --kernel-trace --stats --output-format csv -d <stats_dir> -- python3 bench.py ...

Writes profiles/<tag>_pmc_fetch_write.csv (per-kernel mean counter values per launch, launches per step, calibration rows),
profiles/traffic_<tag>.json (HBM bytes per launch / per step / per scale-0 pixel, next to the algorithmic count of
codec-eval_amd/roofline.py) and, with <stats_dir>, profiles/<stats_tag or tag>_kernel_stats.csv (short kernel names).

Counters are KiB.  MI355X_MICROARCH.md (HBM section): on gfx950 FETCH_SIZE reports half the bytes of a 16-B-per-lane
streaming read and other widths are uncalibrated - so the correction factor of every access width is MEASURED here
from the known-byte-count calibration streams that every prof_step.py pass starts with (k_calib_read<W> /
k_calib_write<W>): factor = bytes moved / bytes reported.  Each kernel is corrected with the factor of its dominant
global-load width.
"""
import collections
import csv
import glob
import importlib
import json
import os
import re
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))

SHORT = [
    (r"k_ssim2_hblur_lds<-1>", "ssim2_hblur_L1-5"), (r"k_ssim2_vblur_dma<-1>", "ssim2_vblur_ssim_L1-5"),
    (r"k_ssim2_hblur_lds<0>", "ssim2_hblur_L0"), (r"k_ssim2_vblur_dma<0>", "ssim2_vblur_ssim_L0"),
    (r"k_ssim2_prep<true>", "ssim2_prep_u8"), (r"k_ssim2_prep<false>", "ssim2_prep"), (r"k_ssim2_finalize", "ssim2_finalize"),
    (r"k_dssim_create_stream<true", "dssim_create_u8"), (r"k_dssim_create_stream<false", "dssim_create"), (r"k_dssim_create<true>", "dssim_create_u8"), (r"k_dssim_create<false>", "dssim_create"), (r"k_dssim_compare", "dssim_compare"),
    (r"k_dssim_avg", "dssim_avg"), (r"k_dssim_absdev", "dssim_absdev"), (r"k_dssim_finalize_pairs", "dssim_finalize"),
    (r"k_ba_front<false>", "ba_front_u8"), (r"k_ba_front<true>", "ba_front_half"), (r"k_ba_subsample2x_u8", "ba_subsample2x"),
    (r"k_ba_blur_h<(\d+)>", "ba_blur_h{}"), (r"k_ba_blur_v<(\d+)>", "ba_blur_v{}"),
    (r"k_ba_blur_v_split<13", "ba_blur_hv_mask"), (r"k_ba_blur_v_split<33", "ba_blur_v_lf"), (r"k_ba_blur_v_split<15, *1, *true", "ba_blur_hv_mf"), (r"k_ba_blur_v_split<7, *2, *true", "ba_blur_hv_hf"),
    (r"k_ba_blur_v_split<15", "ba_blur_v_mf"), (r"k_ba_blur_v_split<7", "ba_blur_v_hf"),
    (r"k_ba_malta_l2_xy", "ba_malta_l2"), (r"k_ba_mask_pre", "ba_mask_pre"), (r"k_ba_mask_vals", "ba_mask_vals"), (r"k_ba_mask_combine", "ba_mask_combine"),
    (r"k_ba_final", "ba_final"), (r"k_ba_score", "ba_score"), (r"k_psnr_sse", "psnr_sse"), (r"k_psnr_clear", "psnr_clear"),
    (r"k_xyb_roundtrip", "xyb_roundtrip"), (r"k_calib_read<(\d+)>", "calib_read_b{}"), (r"k_calib_write<(\d+)>", "calib_write_b{}"),
]
# dominant global-load width of each kernel (bytes per lane): which calibration factor corrects its FETCH_SIZE
READ_WIDTH_16 = ("ssim2_hblur", "ssim2_vblur", "ba_blur_h", "ba_blur_v_", "ba_malta_l2")
READ_WIDTH_1 = ("dssim_create_u8", "ba_subsample2x")
WRITE_WIDTH_16 = ("ssim2_hblur", "ba_blur_h")


def short(name):
    for pat, fmt in SHORT:
        m = re.search(pat, name)
        if m:
            return fmt.format(*m.groups())
    return name.split("(")[0][-48:]


def one(d, suffix):
    f = glob.glob(os.path.join(d, "**", "*" + suffix), recursive=True)
    if not f:
        raise SystemExit(f"no *{suffix} under {d}")
    return max(f, key=os.path.getmtime)  # gpurun merges into gpurun_out/: take the newest run


def counters_of(d, cname):
    acc, n = collections.defaultdict(float), collections.Counter()
    for r in csv.DictReader(open(one(d, "counter_collection.csv"))):
        if r["Counter_Name"] != cname:
            continue
        k = short(r["Kernel_Name"])
        acc[k] += float(r["Counter_Value"])
        n[k] += 1
    return {k: acc[k] / n[k] for k in acc}, dict(n)


def main():
    if sys.argv[1] == "--stats-only":  # python3 profiles/make_traffic.py --stats-only <stats_dir> <stats_tag>: only the kernel_stats condensation
        tag, fetch_dir, write_dir, sidecar = None, None, None, None
        stats_dir, stats_tag = sys.argv[2], sys.argv[3]
    else:
        tag, fetch_dir, write_dir, sidecar = sys.argv[1:5]
        stats_dir = sys.argv[5] if len(sys.argv) > 5 else None
        stats_tag = sys.argv[6] if len(sys.argv) > 6 else tag
    if stats_dir:
        rows = list(csv.DictReader(open(one(stats_dir, "kernel_stats.csv"))))
        with open(os.path.join(HERE, f"{stats_tag}_kernel_stats.csv"), "w", newline="") as fo:
            w = csv.writer(fo)
            w.writerow(["Kernel", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
            # one row per LAUNCH NAME (what bench.py / ce_prof_* call the kernel); a launch name that covers several
            # template instances (ba_malta_l2: the full- and the half-resolution level) gets one merged row - the average
            # bench.py's roofline.avg_launch_ms is compared with - followed by a row per instance
            groups = collections.OrderedDict()
            for r in rows:
                groups.setdefault(short(r["Name"]), []).append(r)
            for name, rs in groups.items():
                if len(rs) == 1:
                    r = rs[0]
                    w.writerow([name, r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"], r["MinNs"], r["MaxNs"]])
                    continue
                calls = sum(int(r["Calls"]) for r in rs)
                total = sum(int(r["TotalDurationNs"]) for r in rs)
                w.writerow([name, calls, total, f"{total / calls:.6f}", f"{sum(float(r['Percentage']) for r in rs):.2f}",
                            min(int(r["MinNs"]) for r in rs), max(int(r["MaxNs"]) for r in rs)])
                for r in rs:
                    inst = re.search(r"<([^>]*)>", r["Name"])
                    w.writerow([f"{name} [instance <{inst.group(1) if inst else '?'}>]", r["Calls"], r["TotalDurationNs"], r["AverageNs"],
                                r["Percentage"], r["MinNs"], r["MaxNs"]])
    if sidecar is None:
        return
    side = json.load(open(sidecar))
    rf = importlib.import_module("codec-eval_amd.roofline")
    alg = rf.step_bytes([rf.Bucket(side["width"], side["height"], side["n_refs"], side["n_pairs"])], side["metrics"], side["xyb_roundtrip"])
    px0 = side["n_pairs"] * side["width"] * side["height"]
    fetch, n_f = counters_of(fetch_dir, "FETCH_SIZE")
    write, n_w = counters_of(write_dir, "WRITE_SIZE")
    cal = side["calib_bytes"]
    f_read = {w_: cal / (fetch[f"calib_read_b{w_}"] * 1024) for w_ in (1, 4, 16) if fetch.get(f"calib_read_b{w_}")}
    f_write = {w_: cal / (write[f"calib_write_b{w_}"] * 1024) for w_ in (4, 16) if write.get(f"calib_write_b{w_}")}
    out = {
        "_provenance": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE, two separate passes over "
                       f"profiles/prof_step.py {side['config']} {side['steps']}; counters are KiB, averaged per launch; corrected by the "
                       "per-width factors MEASURED in the same passes from the calibration streams (bytes moved / bytes reported); "
                       "produced by profiles/make_traffic.py",
        "_launch": dict(side, scale0_pixels=px0),
        "_calibration": {"read_factor_by_lane_bytes": {str(k): round(v, 4) for k, v in f_read.items()},
                         "write_factor_by_lane_bytes": {str(k): round(v, 4) for k, v in f_write.items()}},
    }
    with open(os.path.join(HERE, f"{tag}_pmc_fetch_write.csv"), "w", newline="") as fo:
        w = csv.writer(fo)
        w.writerow(["Kernel", "launches_per_step", "FETCH_SIZE_KiB_per_launch", "WRITE_SIZE_KiB_per_launch"])
        for k in sorted(set(fetch) | set(write)):
            f, wr = fetch.get(k, 0.0), write.get(k, 0.0)
            lps = n_f.get(k, n_w.get(k, 0)) / (1 if k.startswith("calib_") else side["steps"])
            w.writerow([k, round(lps, 2), round(f, 1), round(wr, 1)])
            if k.startswith(("calib_", "__amd")):
                continue
            rw = 16 if k.startswith(READ_WIDTH_16) else 1 if k.startswith(READ_WIDTH_1) else 4
            ww = 16 if k.startswith(WRITE_WIDTH_16) else 4
            fb = f * 1024 * f_read.get(rw, 1.0)
            wb = wr * 1024 * f_write.get(ww, 1.0)
            per_step = (fb + wb) * lps
            out[k] = {"launches_per_step": round(lps, 2), "fetch_bytes_per_launch": fb, "write_bytes_per_launch": wb,
                      "hbm_bytes_per_launch": fb + wb, "hbm_bytes_per_step": per_step, "bytes_per_scale0_pixel": round(per_step / px0, 2),
                      "read_lane_bytes": rw, "write_lane_bytes": ww}
            if k in alg:
                out[k]["algorithmic_bytes_per_step"] = alg[k]
                out[k]["algorithmic_bytes_per_scale0_pixel"] = round(alg[k] / px0, 2)
                out[k]["traffic_over_algorithmic"] = round(per_step / alg[k], 3)
    json.dump(out, open(os.path.join(HERE, f"traffic_{tag}.json"), "w"), indent=1)
    print("calibration", out["_calibration"])
    for k, v in out.items():
        if not k.startswith("_"):
            print(f"{k:24s} {v['bytes_per_scale0_pixel']:8.2f} B/px0   alg {v.get('algorithmic_bytes_per_scale0_pixel', float('nan')):8.2f}")


if __name__ == "__main__":
    main()
