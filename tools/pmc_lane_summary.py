#!/usr/bin/env python3
"""pmc_summary.json of tools/pmc_lane.sh -> the two summaries bench.py reads from profiles/:
   python tools/pmc_lane_summary.py gpurun_out/<tag>/lane_b32768_n10 32768 10 [model] [round prefix, default r03]
writes profiles/<prefix>_pmc_traffic_b{B}_n{N}.json and profiles/<prefix>_sq_summary_b{B}_n{N}.json (go1) or ..._{model}.json"""
import json
import sys
from pathlib import Path

src, B, N = Path(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
model = sys.argv[4] if len(sys.argv) > 4 else "go1"
prefix = sys.argv[5] if len(sys.argv) > 5 else "r03"
m = json.load(open(src / "pmc_summary.json"))["median_per_launch"]
tagm = "" if model == "go1" else "_" + model
name = "Go1" if model == "go1" else "8-contact-point model"
root = Path(__file__).resolve().parent.parent / "profiles"
fetch = m["FETCH_SIZE"] * 1024.0          # rocprofv3 reports KB
write = m["WRITE_SIZE"] * 1024.0
traffic = {
    "workload": f"B={B}, N={N}, {name}, lane-per-instance kernel (qmpc_lane_kernel)",
    "source": "tools/pmc_lane.sh: rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate runs over tools/lane_bench.py, "
              "median of the launches; tools/pmc_lane_summary.py",
    "fetch_bytes_per_launch_raw": fetch,
    "write_bytes_per_launch": write,
    "traffic_bytes_per_launch": fetch + write,
    "traffic_bytes_per_launch_calibrated": 2.0 * fetch + write,
    "calibration_note": "FETCH_SIZE x2 for 8-byte-per-lane coalesced reads (profiles/r02_fetch_calibration.txt, MI355X_MICROARCH.md)",
    "l2_hit_rate": m["TCC_HIT_sum"] / (m["TCC_HIT_sum"] + m["TCC_MISS_sum"]),
}
wave = m["SQ_WAVE_CYCLES"]
sq = {
    "workload": f"B={B}, N={N}, {name}, lane-per-instance kernel",
    "source_passes": "tools/pmc_lane.sh (SQ counters, three --pmc passes); tools/pmc_lane_summary.py",
    "issue_frac": m["SQ_ACTIVE_INST_ANY"] / wave,
    "wait_frac": m["SQ_WAIT_ANY"] / wave,
    "issue_stall_frac": m["SQ_WAIT_INST_ANY"] / wave,
    "valu_frac": m["SQ_ACTIVE_INST_VALU"] / wave,
    "lds_wait_frac": m["SQ_WAIT_INST_LDS"] / wave,
    "mfma_busy": 0.0,
    "valu_insts_per_solve": m["SQ_INSTS_VALU"] / B,
    "fp64_insts_per_launch": {"fma": m["SQ_INSTS_VALU_FMA_F64"], "mul": m["SQ_INSTS_VALU_MUL_F64"],
                              "add": m["SQ_INSTS_VALU_ADD_F64"], "trans": m["SQ_INSTS_VALU_TRANS_F64"]},
    "vmem_rd_insts": m["SQ_INSTS_VMEM_RD"], "vmem_wr_insts": m["SQ_INSTS_VMEM_WR"], "lds_insts": m["SQ_INSTS_LDS"],
    "icache_miss_rate": m["SQC_ICACHE_MISSES"] / max(m["SQC_ICACHE_REQ"], 1.0),
}
for kind, obj in (("pmc_traffic", traffic), ("sq_summary", sq)):
    out = root / f"{prefix}_{kind}_b{B}_n{N}{tagm}.json"
    out.write_text(json.dumps(obj, indent=1) + "\n")
    print(out)
