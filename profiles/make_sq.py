#!/usr/bin/env python3
"""Condense an SQ-counter pass into profiles/<tag>_sq_util.csv: per kernel, what the SIMDs were doing.

usage: python3 profiles/make_sq.py <tag> <pmc_dir>

  <pmc_dir>  output of  rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_INST_ANY
             SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_ANY GRBM_GUI_ACTIVE --output-format csv -d <pmc_dir> -- python3 profiles/prof_step.py <config> 2 <sidecar>

Columns (averages per launch): gui_cycles = GRBM_GUI_ACTIVE / 8 (the counter sums the 8 XCDs) = kernel duration in
shader cycles; valu_util / lds_util = SQ_ACTIVE_INST_{VALU,LDS} over the SIMD quad-cycles available in that time
(1024 SIMDs x gui / 4: the SQ counters count quad-cycles, MI355X_MICROARCH.md, cycle-constants table); waves_per_simd =
SQ_WAVE_CYCLES over the same; wait_any / wait_inst = fraction of wave-cycles parked on s_waitcnt/barriers / stalled at
issue; bank_conflict = SQ_LDS_BANK_CONFLICT / SQ_ACTIVE_INST_LDS; valu_minst = SQ_INSTS_VALU in millions.
"""
import collections
import csv
import glob
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from make_traffic import short  # noqa: E402


def main():
    tag, d = sys.argv[1:3]
    f = max(glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True), key=os.path.getmtime)
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    cnt = collections.Counter()
    for r in csv.DictReader(open(f)):
        k = short(r["Kernel_Name"])
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        cnt[(k, r["Counter_Name"])] += 1
    rows = []
    for k in sorted(agg):
        if k.startswith(("__amd", "calib_")):
            continue
        a = {c: v / cnt[(k, c)] for c, v in agg[k].items()}
        gui = a.get("GRBM_GUI_ACTIVE", 0.0) / 8
        quads = max(1024 * gui / 4, 1.0)
        wc = max(a.get("SQ_WAVE_CYCLES", 0.0), 1.0)
        rows.append([k, cnt[(k, "GRBM_GUI_ACTIVE")], round(gui), round(a.get("SQ_ACTIVE_INST_VALU", 0) / quads, 3),
                     round(a.get("SQ_ACTIVE_INST_LDS", 0) / quads, 3), round(a.get("SQ_WAVE_CYCLES", 0) / quads, 2),
                     round(a.get("SQ_WAIT_ANY", 0) / wc, 3), round(a.get("SQ_WAIT_INST_ANY", 0) / wc, 3),
                     round(a.get("SQ_LDS_BANK_CONFLICT", 0) / max(a.get("SQ_ACTIVE_INST_LDS", 0), 1.0), 3),
                     round(a.get("SQ_INSTS_VALU", 0) / 1e6, 2)])
    with open(os.path.join(HERE, f"{tag}_sq_util.csv"), "w", newline="") as fo:
        w = csv.writer(fo)
        w.writerow(["Kernel", "launches", "gui_cycles", "valu_util", "lds_util", "waves_per_simd", "wait_any", "wait_inst", "bank_conflict", "valu_minst"])
        w.writerows(rows)
    for r in rows:
        print("%-24s n %3d gui %9d valu %.2f lds %.2f waves/simd %.2f wait_any %.2f wait_inst %.2f conf %.2f valu_Minst %.1f" % tuple(r))


if __name__ == "__main__":
    main()
