#!/usr/bin/env python3
"""Straggler hand-off of large batches (qmpc_hip.hip: launch_solve): the automatic path (lane kernel capped at
QMPC_LANE_CAP iterations + wave-per-instance kernel on the instances left) against the pure lane kernel (QMPC_LANE_CAP=0),
same records, kernel-side time between the handle's events.  GPU box:  python tools/handoff_bench.py [--cases N:B,...]"""
import argparse
import os
import sys
from pathlib import Path

import numpy as np

REPO = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(REPO))
import __graft_entry__ as g  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--cases", default="10:32768,10:65536,20:65536")
ap.add_argument("--caps", default="0,default")
ap.add_argument("--reps", type=int, default=6)
a = ap.parse_args()
pkg = g._load_pkg()
lib = pkg.load_library()
import torch  # noqa: E402

for case in a.cases.split(","):
    N, B = (int(x) for x in case.split(":"))
    p = pkg.default_params(N, 0, lib)
    rec = pkg.random_go1_trot_states(B, config_id=3 if N == 20 else 4)
    d_in = torch.from_numpy(rec.view(np.float64).reshape(B, -1).copy()).cuda()
    ref = None
    for cap in a.caps.split(","):
        if cap == "default":
            os.environ.pop("QMPC_LANE_CAP", None)
        else:
            os.environ["QMPC_LANE_CAP"] = cap
        s = pkg.Solver(p, B, 0, lib)
        d_f = torch.zeros(B, 12, dtype=torch.float64, device="cuda")
        d_i = torch.zeros(B, 5, dtype=torch.float64, device="cuda")
        ms = []
        for r in range(a.reps + 2):
            s.solve_device(B, d_in.data_ptr(), d_f.data_ptr(), d_i.data_ptr())
            s.wait()
            if r >= 2:
                ms.append(s.last_kernel_ms())
        f = d_f.cpu().numpy()
        info = d_i.cpu().numpy().view(pkg.INFO_DTYPE).reshape(B)
        s.close()
        if ref is None:
            ref = (f, info)
        it = info["iterations"]
        print(f"N={N} B={B} cap={cap}: {np.median(ms):.3f} ms -> {B / np.median(ms) / 1e3:.3f} M solves/s; ok {int((info['status'] == 0).sum())}/{B}, "
              f"iterations mean {it.mean():.2f} max {it.max()}; vs cap=0: max |df| {np.abs(f - ref[0]).max():.2e} N, "
              f"iteration counts equal {100 * (it == ref[1]['iterations']).mean():.3f} %", flush=True)
