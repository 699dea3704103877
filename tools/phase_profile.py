#!/usr/bin/env python3
"""Phase-level cycle breakdown of the solve kernel (s_memtime instrumentation).
Run on the GPU box:  python tools/phase_profile.py [--batch 1024] [--horizon 10]"""
import argparse
import sys
from pathlib import Path

import numpy as np

REPO = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(REPO))
import __graft_entry__ as g  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=1024)
ap.add_argument("--horizon", type=int, default=10)
a = ap.parse_args()
pkg = g._load_pkg()
lib = pkg.load_library()
p = pkg.default_params(a.horizon, 0, lib)
s = pkg.Solver(p, a.batch, 0, lib)
rec = pkg.random_go1_trot_states(a.batch, config_id=2 if a.horizon == 10 else 3)
s.solve(rec)
c = s.phase_profile(rec)
names = ["setup", "expand", "build", "mfma+terms", "stage_solve", "P_update", "directions", "rollout(rest)", "misc",
         "rot.prepass", "roll:dx+gain", "roll:bcast", "roll:step", "ipm_apply", "mfma drain"]
it = c[:, 15].astype(float)
tot = c[:, :15].sum(1)
print(f"batch {a.batch} N {a.horizon}: mean iterations {it.mean():.2f}; mean cycles/instance {tot.mean():.0f}; "
      f"max {tot.max()}; per iteration {np.mean(tot / np.maximum(it, 1)):.0f}")
for i, n in enumerate(names):
    print(f"  {n:12s} {c[:, i].mean():12.0f} cycles  {100 * c[:, i].sum() / tot.sum():5.1f} %   per-iter {np.mean(c[:, i] / np.maximum(it, 1)):9.0f}")

nc = rec["contacts"].sum(1)
for k in (2, 4):
    m = nc == k
    print(f"stance legs {k}: {m.sum()} instances, cycles/iter {np.mean(tot[m] / it[m]):.0f}, iterations mean {it[m].mean():.2f} max {it[m].max():.0f}, "
          f"total cycles mean {tot[m].mean():.0f} max {tot[m].max()}")
top = np.argsort(tot)[-8:]
print("slowest instances:", [(int(i), int(nc[i]), int(it[i]), int(tot[i])) for i in top])
