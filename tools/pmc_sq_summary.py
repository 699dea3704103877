#!/usr/bin/env python3
"""Median per-launch SQ counter values of the solve kernel from the passes of tools/pmc_sq.sh, plus the
ratios the roofline discussion uses (SQ_*_CYCLES / WAIT / ACTIVE_INST count quad-cycles; MFMA busy counts cycles)."""
import csv
import glob
import json
import sys
from collections import defaultdict

vals = defaultdict(list)
for f in glob.glob(sys.argv[1] + "/sq_*/**/*_counter_collection.csv", recursive=True):
    with open(f, newline="") as fh:
        for row in csv.DictReader(fh):
            if "qmpc_solve_kernel" in row["Kernel_Name"]:
                vals[row["Counter_Name"]].append(float(row["Counter_Value"]))
med = {k: sorted(v)[len(v) // 2] for k, v in vals.items()}
out = {"launches_sampled": {k: len(v) for k, v in vals.items()}, "median_per_launch": med}
w = med.get("SQ_WAVE_CYCLES")
if w:
    r = {}
    for k in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS",
              "SQ_ACTIVE_INST_SCA", "SQ_WAIT_INST_LDS", "SQ_ACTIVE_INST_MISC"):
        if k in med:
            r[k + "/WAVE_CYCLES"] = med[k] / w
    if "SQ_VALU_MFMA_BUSY_CYCLES" in med and "SQ_BUSY_CYCLES" in med:
        r["MFMA_BUSY_CYCLES/(4*WAVE_CYCLES)"] = med["SQ_VALU_MFMA_BUSY_CYCLES"] / (4.0 * w)
    if "SQ_INSTS_VALU" in med:
        r["VALU_insts_per_wave_quadcycle"] = med["SQ_INSTS_VALU"] / w
    out["ratios"] = r
print(json.dumps(out, indent=1))
