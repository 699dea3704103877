#!/usr/bin/env python3
"""A/B of the wrench-form wave kernel (qmpc_wform.hip, QMPC_WFORM=1) against the round-1 kernel (QMPC_WFORM=0) and the
CPU oracle on the contract workload: forces, status words, iteration counts, kernel time.  GPU box only.
  python tools/wform_check.py [--batch 1024] [--horizon 10] [--oracle 256]"""
import argparse
import os
import sys
import time
from pathlib import Path

import numpy as np

REPO = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(REPO))
import __graft_entry__ as g  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=1024)
ap.add_argument("--horizon", type=int, default=10)
ap.add_argument("--oracle", type=int, default=256)
ap.add_argument("--reps", type=int, default=20)
a = ap.parse_args()
pkg = g._load_pkg()
lib = pkg.load_library()
from oracle import pyoracle  # checker only

p = pkg.default_params(a.horizon, 0, lib)
rec = pkg.random_go1_trot_states(a.batch, config_id=2 if a.horizon == 10 else 3)
res = {}
for wf in (0, 1):
    os.environ["QMPC_WFORM"] = str(wf)
    s = pkg.Solver(p, a.batch, 0, lib)
    f, info = s.solve(rec)
    ts = []
    for _ in range(a.reps):
        s.solve(rec)
        ts.append(s.last_kernel_ms())
    s.close()
    res[wf] = (f, info, np.median(ts), np.min(ts))
    print(f"QMPC_WFORM={wf}: kernel ms median {np.median(ts):.4f} min {np.min(ts):.4f}  -> {a.batch / np.median(ts) / 1e3:.3f} M solves/s; "
          f"status ok {int((info['status'] == 0).sum())}/{a.batch}; iterations mean {info['iterations'].mean():.2f} max {info['iterations'].max()}")
f0, i0 = res[0][:2]
f1, i1 = res[1][:2]
print(f"wform vs round-1 kernel: max |df| {np.abs(f1 - f0).max():.3e} N; status equal {np.array_equal(i0['status'], i1['status'])}; "
      f"iteration counts equal on {100.0 * (i0['iterations'] == i1['iterations']).mean():.2f} %")
n = min(a.oracle, a.batch)
if n > 0:
    t0 = time.time()
    fo, io = pyoracle.solve(pyoracle.default_params(a.horizon, 0), rec[:n])
    print(f"oracle on {n} instances ({time.time() - t0:.1f} s): wform max |df| {np.abs(f1[:n] - fo).max():.3e} N, "
          f"round-1 {np.abs(f0[:n] - fo).max():.3e} N; iteration counts equal: wform {100.0 * (i1['iterations'][:n] == io['iterations']).mean():.2f} %, "
          f"round-1 {100.0 * (i0['iterations'][:n] == io['iterations']).mean():.2f} %; status equal {np.array_equal(i1['status'][:n], io['status'])}")
