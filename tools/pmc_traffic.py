#!/usr/bin/env python3
"""Per-launch HBM traffic of the solve kernel from two rocprofv3 --pmc passes.
usage: tools/pmc_traffic.py <fetch counter_collection.csv> <write counter_collection.csv> <out.json> [batch] [horizon]

FETCH_SIZE / WRITE_SIZE are reported in KiB on gfx950 (MI355X_MICROARCH.md: hbm_bytes =
(FETCH_SIZE + WRITE_SIZE) * 1024).  Calibration for THIS access pattern (tools/microbench/fetch_calib.hip, one
wave per 384-byte record read with one 8-byte-per-lane load, 96 + 40 bytes written; known byte counts,
profiles/r02_fetch_calibration.txt): FETCH_SIZE reports exactly 1/2 of the bytes read (196 624.5 KiB for
393 216 KiB), as the guide found for 16-byte streaming reads -> x2; WRITE_SIZE counts 32-byte sectors
(160 B per record for 136 B of payload) -> x1, i.e. it IS the sector traffic."""
FETCH_CALIBRATION = 2.0
WRITE_CALIBRATION = 1.0
import csv
import json
import sys


def per_launch(path, counter):
    vals = []
    with open(path, newline="") as f:
        for row in csv.DictReader(f):
            if "qmpc_solve_kernel" in row["Kernel_Name"] and row["Counter_Name"] == counter:
                vals.append(float(row["Counter_Value"]))
    if not vals:
        raise SystemExit(f"no {counter} rows for qmpc_solve_kernel in {path}")
    vals.sort()
    return vals[len(vals) // 2], len(vals)      # median over launches


fetch, nf = per_launch(sys.argv[1], "FETCH_SIZE")
write, nw = per_launch(sys.argv[2], "WRITE_SIZE")
batch = int(sys.argv[4]) if len(sys.argv) > 4 else 1024
horizon = int(sys.argv[5]) if len(sys.argv) > 5 else 10
out = {
    "workload": f"B={batch}, N={horizon} (bench.py)",
    "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes), kernel qmpc_solve_kernel, "
              f"median over {nf} / {nw} launches",
    "unit_note": "counter values are KiB; hbm_bytes = (FETCH_SIZE + WRITE_SIZE) * 1024; includes instruction "
                 "fetch through the 8 XCD L2s",
    "FETCH_SIZE_KiB_per_launch": fetch,
    "WRITE_SIZE_KiB_per_launch": write,
    "traffic_bytes_per_launch": (fetch + write) * 1024.0,
    "traffic_bytes_per_launch_calibrated": (FETCH_CALIBRATION * fetch + WRITE_CALIBRATION * write) * 1024.0,
    "calibration_note": "FETCH_SIZE x2 (measured with tools/microbench/fetch_calib.hip on this access pattern), "
                        "WRITE_SIZE x1 (32-byte sectors); the calibrated figure includes the kernel's instruction "
                        "fetch through the 8 XCD L2s (~60 KB of code each)",
    "algorithmic_bytes_per_launch": batch * (384 + 96 + 40),
}
json.dump(out, open(sys.argv[3], "w"), indent=1)
print(json.dumps(out))
