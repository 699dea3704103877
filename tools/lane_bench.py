"""Lane-per-instance kernel against the wave-per-instance kernels: kernel ms, solves/s, parity between the two and
against the CPU oracle on a sample.  GPU only.  python tools/lane_bench.py [--cases N:B,...] [--model go1|biped8]"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time
from pathlib import Path

import numpy as np

REPO = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(REPO))
from oracle import pyoracle  # noqa: E402  (checker)

pkg = pyoracle.pkg


def run(variant: str, p, rec, reps: int, biped: bool, extra_env=None, convex: bool = False):
    import torch
    os.environ["QMPC_VARIANT"] = variant
    for k, v in (extra_env or {}).items():
        os.environ[k] = v
    B = rec.shape[0]
    nu = 24 if biped else 12
    s = pkg.Solver(p, B, device=0)
    d_in = torch.from_numpy(rec.view(np.float64).reshape(B, -1).copy()).cuda()
    d_f = torch.zeros(B, nu, dtype=torch.float64, device="cuda")
    d_i = torch.zeros(B, 5, dtype=torch.float64, device="cuda")
    torch.cuda.synchronize()
    call = s.solve8_device if biped else (s.convex_solve_device if convex else s.solve_device)
    ms = []
    for r in range(reps + 1):
        call(B, d_in.data_ptr(), d_f.data_ptr(), d_i.data_ptr())
        s.wait()
        if r:
            ms.append(s.last_kernel_ms())
    f = d_f.cpu().numpy()
    info = d_i.cpu().numpy().view(pkg.INFO_DTYPE).reshape(B)
    s.close()
    for k in (extra_env or {}):
        os.environ.pop(k, None)
    return f, info, float(np.median(ms))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", default="10:4096,10:32768,10:65536,20:65536")
    ap.add_argument("--model", default="go1")
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--sample", type=int, default=128)
    ap.add_argument("--skip-wave", action="store_true")
    ap.add_argument("--nosort", action="store_true")
    a = ap.parse_args()
    biped = a.model == "biped8"
    convex = a.model == "convex"
    out = []
    for case in a.cases.split(","):
        N, B = (int(x) for x in case.split(":"))
        if biped:
            p = pkg.default_biped8_params(N)
            rec = pkg.random_biped8_states(B, config_id=5)
        elif convex:
            p = pkg.default_convex_params(N)
            rec = pkg.random_go1_convex_states(B, config_id=12)
        else:
            p = pkg.default_params(N)
            rec = pkg.random_go1_trot_states(B, config_id=3 if N == 20 else 2)
        row = {"N": N, "B": B, "model": a.model}
        fl, il, msl = run("4", p, rec, a.reps, biped, None, convex)
        row["lane_ms"] = msl
        row["lane_solves_per_s"] = B / msl * 1e3
        row["lane_status"] = np.bincount(il["status"], minlength=6).tolist()
        row["lane_iters_mean_max"] = [float(il["iterations"].mean()), int(il["iterations"].max())]
        if a.nosort:
            _, _, msn = run("4", p, rec, a.reps, biped, {"QMPC_LANE_SORT": "0"}, convex)
            row["lane_nosort_ms"] = msn
        if not a.skip_wave:
            # variant 0 = the library's automatic choice, which is the lane kernel from QMPC_LANE_MIN instances on:
            # raise the threshold so that this leg really times the wave-per-instance kernels
            fw, iw, msw = run("0", p, rec, a.reps, biped, {"QMPC_LANE_MIN": str(1 << 30)}, convex)
            row["wave_ms"] = msw
            row["wave_solves_per_s"] = B / msw * 1e3
            row["speedup"] = msw / msl
            row["lane_vs_wave_linf"] = float(np.abs(fl - fw).max())
            row["status_equal"] = bool((il["status"] == iw["status"]).all())
            row["iters_equal_frac"] = float((il["iterations"] == iw["iterations"]).mean())
        ns = min(a.sample, B)
        idx = np.linspace(0, B - 1, ns).astype(int)
        op = pyoracle.default_biped8_params(N, 0) if biped else (pyoracle.default_convex_params(N, 0) if convex else pyoracle.default_params(N, 0))
        fo, io = (pyoracle.solve8 if biped else (pyoracle.convex_solve if convex else pyoracle.solve))(op, rec[idx], threads=8)
        row["lane_vs_oracle_linf"] = float(np.abs(fl[idx] - fo).max())
        row["oracle_iters_equal_frac"] = float((il["iterations"][idx] == io["iterations"]).mean())
        print(json.dumps(row), flush=True)
        out.append(row)
    return out


if __name__ == "__main__":
    main()
