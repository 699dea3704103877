#!/usr/bin/env python3
"""Diagnostic: where a reference-mode solve of the wrench-form wave kernel spends its cycles (library built with
-DQMPC_REF_PROF, loaded through QMPC_LIB: the info fields then carry cycle counts).  GPU box only."""
import os, sys
from pathlib import Path
import numpy as np
REPO = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(REPO))
import __graft_entry__ as g  # noqa: E402
pkg = g._load_pkg(); lib = pkg.load_library()
N, B = int(os.environ.get("RP_N", "10")), int(os.environ.get("RP_B", "1024"))
p = pkg.default_params(N, pkg.MODE_REFERENCE, lib)
rec = pkg.random_go1_trot_states(B, config_id=2)
s = pkg.Solver(p, B, 0, lib)
for r in range(3):
    f, info = s.solve(rec)
print("kernel ms", s.last_kernel_ms())
it = info["iterations"].astype(float)
tot, ls, bw = info["cost"], info["max_violation"], info["last_step"]
st = np.floor(info["penalty"]); tr = np.round((info["penalty"] - st) * 1e3)
print(f"iterations mean {it.mean():.2f} max {it.max():.0f}; failed trials mean {tr.mean():.2f} max {tr.max():.0f}")
print(f"cycles/instance: total mean {tot.mean():.0f} max {tot.max():.0f}; backward(+prepass+expected) {bw.mean():.0f}; "
      f"line search {ls.mean():.0f}; stationarity {st.mean():.0f}; rest {np.mean(tot - ls - bw - st):.0f}")
print(f"per iteration: backward {np.mean(bw / it):.0f}, line search {np.mean(ls / it):.0f} ({np.mean(ls / (it + tr)):.0f} per trial), "
      f"stationarity {np.mean(st / it):.0f}, rest {np.mean((tot - ls - bw - st) / it):.0f}")
o = np.argsort(-tot)[:8]
print("slowest:", [(int(i), int(it[i]), int(tr[i]), int(tot[i])) for i in o])
s.close()
