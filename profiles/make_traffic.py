#!/usr/bin/env python3
"""Condense the rocprofv3 outputs of a profiling call into the files committed under profiles/.

usage: python3 profiles/make_traffic.py <round-tag> <stats_dir> <fetch_dir> <write_dir>

  <stats_dir>  output of  rocprofv3 --kernel-trace --stats --output-format csv -d <stats_dir> -- python3 bench.py --steps 20 --warmup 3
  <fetch_dir>  output of  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d <fetch_dir> -- python3 profiles/prof_step.py 2
  <write_dir>  the same with --pmc WRITE_SIZE (a separate pass: the two TCC counters do not share one)

Writes profiles/<tag>_kernel_stats.csv (kernel_stats.csv with short kernel names), profiles/<tag>_pmc_fetch_write.csv
(per-kernel mean counter values) and profiles/traffic_<tag>.json (HBM bytes per launch and per scale-0 pixel).
FETCH_SIZE / WRITE_SIZE count KiB; FETCH_SIZE is doubled for kernels that stream with 16 B per lane, as
MI355X_MICROARCH.md (HBM / rocprofv3 section) prescribes for gfx950.
"""
import collections
import csv
import glob
import json
import os
import re
import sys

SHORT = [
    (r"k_ssim2_hblur_lds<-1>", "ssim2_hblur_L1-5"), (r"k_ssim2_vblur_dma<-1>", "ssim2_vblur_ssim_L1-5"),
    (r"k_ssim2_hblur_lds<(\d)>", "ssim2_hblur_L{}"), (r"k_ssim2_vblur_dma<(\d)>", "ssim2_vblur_ssim_L{}"),
    (r"k_ssim2_prep<true>", "ssim2_prep_u8"), (r"k_ssim2_prep<false>", "ssim2_prep"), (r"k_ssim2_finalize", "ssim2_finalize"),
]
# kernels whose global reads are 16 B per lane (dwordx4 / global_load_lds_dwordx4): FETCH_SIZE x2 on gfx950
DWORDX4_READERS = ("ssim2_hblur_L", "ssim2_vblur_ssim_L")


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


def main():
    tag, stats_dir, fetch_dir, write_dir = sys.argv[1:5]
    here = os.path.dirname(os.path.abspath(__file__))
    rows = list(csv.DictReader(open(one(stats_dir, "kernel_stats.csv"))))
    with open(os.path.join(here, f"{tag}_kernel_stats.csv"), "w", newline="") as fo:
        w = csv.writer(fo)
        w.writerow(["Kernel", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
        for r in rows:
            w.writerow([short(r["Name"]), r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"], r["MinNs"], r["MaxNs"]])
    counters = {}
    for cname, d in (("FETCH_SIZE", fetch_dir), ("WRITE_SIZE", write_dir)):
        acc, n = collections.defaultdict(float), collections.Counter()
        for r in csv.DictReader(open(one(d, "counter_collection.csv"))):
            if r["Counter_Name"] != cname:
                continue
            k = short(r["Kernel_Name"])
            acc[k] += float(r["Counter_Value"])
            n[k] += 1
        counters[cname] = {k: acc[k] / n[k] for k in acc}
    pairs, w_, h_ = 54, 768, 512  # profiles/prof_step.py
    px0 = pairs * w_ * h_
    out = {
        "_provenance": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE, two separate passes over profiles/prof_step.py "
                       "(54 pairs of 768x512, 18 references, SSIMULACRA2, 2 steps), MI355X. Counters are KiB, averaged per launch. "
                       "FETCH_SIZE is doubled for the dwordx4 readers (MI355X_MICROARCH.md, HBM section); WRITE_SIZE is taken as is. "
                       "Produced by profiles/make_traffic.py.",
        "_launch": {"pairs": pairs, "width": w_, "height": h_, "scale0_pixels": px0},
    }
    with open(os.path.join(here, f"{tag}_pmc_fetch_write.csv"), "w", newline="") as fo:
        w = csv.writer(fo)
        w.writerow(["Kernel", "FETCH_SIZE_KiB_per_launch", "WRITE_SIZE_KiB_per_launch"])
        for k in sorted(set(counters["FETCH_SIZE"]) | set(counters["WRITE_SIZE"])):
            f, wr = counters["FETCH_SIZE"].get(k, 0.0), counters["WRITE_SIZE"].get(k, 0.0)
            w.writerow([k, round(f, 1), round(wr, 1)])
            if not k.startswith("ssim2_"):
                continue
            fb = f * 1024 * (2 if k.startswith(DWORDX4_READERS) else 1)
            wb = wr * 1024
            out[k] = {"fetch_bytes_corrected": fb, "write_bytes": wb, "hbm_bytes_per_launch": fb + wb,
                      "bytes_per_scale0_pixel": round((fb + wb) / px0, 2)}
    json.dump(out, open(os.path.join(here, f"traffic_{tag}.json"), "w"), indent=1)
    for k, v in out.items():
        if not k.startswith("_"):
            print(f"{k:24s} {v['bytes_per_scale0_pixel']:8.2f} B/px0")


if __name__ == "__main__":
    main()
