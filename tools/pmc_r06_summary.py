#!/usr/bin/env python3
"""Counter passes of tools/pmc_r06.sh -> the two summaries bench.py reads from profiles/ (per SOLVE = per bench step, which
may be more than one kernel: the lane kernel + its sort + the straggler hand-off's list kernel):
   python tools/pmc_r06_summary.py gpurun_out/<tag> <batch> <horizon>   writes <tag>/r06_pmc_traffic*.json, r06_sq_summary_*.json
FETCH_SIZE / WRITE_SIZE are KiB on gfx950; FETCH_SIZE x2 for this access pattern (profiles/r02_fetch_calibration.txt)."""
import csv
import glob
import json
import sys
from collections import defaultdict
from pathlib import Path

src, B, N = Path(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
per_kernel = defaultdict(lambda: defaultdict(list))      # kernel -> counter -> values per dispatch
for f in glob.glob(str(src / "pmc_*/**/*_counter_collection.csv"), recursive=True):
    with open(f, newline="") as fh:
        for row in csv.DictReader(fh):
            k = row["Kernel_Name"]
            if "qmpc_" not in k:
                continue
            per_kernel[k.split("(")[0]][row["Counter_Name"]].append(float(row["Counter_Value"]))
solve_kernels = [k for k in per_kernel if any(t in k for t in ("qmpc_solve_w_kernel", "qmpc_solve_w_list_kernel", "qmpc_solve_kernel",
                                                                  "qmpc_lane_kernel", "qmpc_lane_sort"))]
main = max(solve_kernels, key=lambda k: sorted(per_kernel[k].get("SQ_WAVE_CYCLES", [0]))[len(per_kernel[k].get("SQ_WAVE_CYCLES", [0])) // 2])


def med(k, c):
    v = sorted(per_kernel[k].get(c, []))
    return v[len(v) // 2] if v else None


def per_solve(c):      # sum over the kernels of one solve (medians per kernel; every kernel runs once per solve)
    return sum((med(k, c) or 0.0) for k in solve_kernels)


fetch, write = per_solve("FETCH_SIZE") * 1024.0, per_solve("WRITE_SIZE") * 1024.0
hit, miss = per_solve("TCC_HIT_sum"), per_solve("TCC_MISS_sum")
short = {k: k.split("::")[-1] for k in solve_kernels}
traffic = {
    "round": "r06", "workload": f"B={B}, N={N}, Go1 (bench.py leg); kernels of one solve: " + ", ".join(sorted(short.values())),
    "source": "tools/pmc_r06.sh: rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE (+TCC hits) in separate runs over bench.py, median per "
              "kernel over the launches, summed over the kernels of a solve; tools/pmc_r06_summary.py",
    "fetch_bytes_per_launch_raw": fetch, "write_bytes_per_launch": write,
    "traffic_bytes_per_launch": fetch + write, "traffic_bytes_per_launch_calibrated": 2.0 * fetch + write,
    "calibration_note": "FETCH_SIZE x2 for 8-byte-per-lane coalesced reads (profiles/r02_fetch_calibration.txt, MI355X_MICROARCH.md), WRITE_SIZE x1",
    "l2_hit_rate": hit / (hit + miss) if hit + miss else None,
    "algorithmic_bytes_per_launch": B * (384 + 96 + 40),
    "per_kernel": {short[k]: {"FETCH_KiB": med(k, "FETCH_SIZE"), "WRITE_KiB": med(k, "WRITE_SIZE")} for k in solve_kernels},
}
w = med(main, "SQ_WAVE_CYCLES")
sq = {
    "round": "r06", "kernel": short[main], "workload": f"B={B}, N={N}, Go1 (bench.py leg); dominant kernel of the solve",
    "source_passes": "tools/pmc_r06.sh (three --pmc passes of 8 SQ counters over bench.py); tools/pmc_r06_summary.py",
    "issue_frac": med(main, "SQ_ACTIVE_INST_ANY") / w, "wait_frac": med(main, "SQ_WAIT_ANY") / w,
    "issue_stall_frac": med(main, "SQ_WAIT_INST_ANY") / w, "valu_frac": med(main, "SQ_ACTIVE_INST_VALU") / w,
    "lds_frac": med(main, "SQ_ACTIVE_INST_LDS") / w, "lds_wait_frac": (med(main, "SQ_WAIT_INST_LDS") or 0.0) / w,
    "mfma_busy": (med(main, "SQ_VALU_MFMA_BUSY_CYCLES") or 0.0) / (4.0 * w),
    "mfma_insts_per_solve": (med(main, "SQ_INSTS_MFMA") or 0.0) / B,
    "valu_insts_per_solve": (med(main, "SQ_INSTS_VALU") or 0.0) / B,
    "fp64_insts_per_launch": {"fma": med(main, "SQ_INSTS_VALU_FMA_F64"), "mul": med(main, "SQ_INSTS_VALU_MUL_F64"),
                              "add": med(main, "SQ_INSTS_VALU_ADD_F64"), "trans": med(main, "SQ_INSTS_VALU_TRANS_F64")},
    "other_kernels_wave_cycles": {short[k]: med(k, "SQ_WAVE_CYCLES") for k in solve_kernels if k != main},
    "wave_cycles": w,
}
tag = "" if (B == 1024 and N == 10) else f"_b{B}_n{N}"
(src / f"r06_pmc_traffic{tag}.json").write_text(json.dumps(traffic, indent=1) + "\n")
(src / f"r06_sq_summary_b{B}_n{N}.json").write_text(json.dumps(sq, indent=1) + "\n")
print(json.dumps({"traffic": traffic, "sq": sq}, indent=1))
