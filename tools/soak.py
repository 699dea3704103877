#!/usr/bin/env python3
"""Soak: many seeded batches through the HIP path and the CPU oracle, every instance compared (status, iteration
count, force L-inf).  Looks for rare-event differences the fixed-seed parity tests cannot see.
Run on the GPU box:  python tools/soak.py [--instances 500000] [--horizon 10] [--model quat|convex|biped8]"""
import argparse
import importlib
import sys
import time
from pathlib import Path

import numpy as np

REPO = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(REPO))
sys.path.insert(0, str(REPO / "tests"))
from conftest import load_pkg  # noqa: E402

pkg = load_pkg()
from oracle import pyoracle as po  # noqa: E402  (checker)

ap = argparse.ArgumentParser()
ap.add_argument("--instances", type=int, default=500000)
ap.add_argument("--horizon", type=int, default=10)
ap.add_argument("--model", choices=["quat", "convex", "biped8"], default="quat")
ap.add_argument("--chunk", type=int, default=32768)
ap.add_argument("--threads", type=int, default=16)
a = ap.parse_args()
lib = pkg.load_library()
gen = {"quat": pkg.random_go1_trot_states, "convex": pkg.random_go1_convex_states, "biped8": pkg.random_biped8_states}[a.model]
dp = {"quat": "default_params", "convex": "default_convex_params", "biped8": "default_biped8_params"}[a.model]
sv = {"quat": "solve", "convex": "convex_solve", "biped8": "solve8"}[a.model]
p = getattr(pkg, dp)(a.horizon, pkg.MODE_CONVERGED, lib)
po_p = getattr(po, dp)(a.horizon, 0)
assert bytes(p) == bytes(po_p)
s = pkg.Solver(p, a.chunk, device=0, lib=lib)
done, worst, n_status_diff, n_iter_diff, n_fail_gpu, n_fail_cpu = 0, 0.0, 0, 0, 0, 0
t0 = time.time()
cfg = 1000
while done < a.instances:
    n = min(a.chunk, a.instances - done)
    rec = gen(n, config_id=cfg, first=done)
    f, info = getattr(s, sv)(rec)
    fo, io = getattr(po, sv)(po_p, rec, threads=a.threads)
    both = (info["status"] == 0) & (io["status"] == 0)
    n_status_diff += int((info["status"] != io["status"]).sum())
    n_iter_diff += int((info["iterations"] != io["iterations"]).sum())
    n_fail_gpu += int((info["status"] != 0).sum())
    n_fail_cpu += int((io["status"] != 0).sum())
    if both.any():
        if a.model == "biped8":
            feet = rec["foot_pos_body"].reshape(-1, 8, 3)
            wr = lambda F: np.concatenate([F.reshape(-1, 8, 3).sum(1), np.cross(feet, F.reshape(-1, 8, 3)).sum(1)], axis=1)
            e = np.abs(wr(f) - wr(fo))[both].max()
        else:
            e = np.abs(f - fo)[both].max()
        worst = max(worst, float(e))
    done += n
s.close()
print(f"soak {a.model} N={a.horizon}: {done} instances in {time.time() - t0:.0f} s; GPU failures {n_fail_gpu}, oracle failures "
      f"{n_fail_cpu}, status differences {n_status_diff}, iteration-count differences {n_iter_diff} "
      f"({100.0 * n_iter_diff / done:.3f} %), worst force difference {worst:.3e} "
      f"{'N (foot wrench)' if a.model == 'biped8' else 'N'}")
