#!/usr/bin/env python3
"""Reference mode (AL-iLQR, <= 10 iterations) on the device against the oracle's restatement of the same scheme, per horizon:
status words, iteration counts, share of the truncated iterates' forces within 1e-6 N, kernel rate at 1024 instances.
QMPC_REF_WFORM_MAXN=12 (read once per process) restores the round-4 rule (dense kernels beyond N=12) for an A/B run.
GPU box:  python tools/refmode_check.py [N ...]"""
import sys
from pathlib import Path

import numpy as np

REPO = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(REPO))
import __graft_entry__ as g  # noqa: E402
from oracle import pyoracle as po  # noqa: E402  (checker)

pkg = g._load_pkg()
lib = pkg.load_library()
Ns = [int(a) for a in sys.argv[1:]] or [10, 16, 20, 24, 32]
for N in Ns:
    p = pkg.default_params(N, pkg.MODE_REFERENCE, lib)
    B = 512
    rec = pkg.random_go1_trot_states(B, config_id=3 if N > 12 else 2)
    s = pkg.Solver(p, 1024, 0, lib)
    f, i = s.solve(rec)
    fam = s.kernel_for_batch(B)
    fo, io = po.solve(po.default_params(N, 1), rec, threads=8)
    e = np.abs(f - fo).max(axis=1)
    print(f"N={N} B={B} [{fam}]: status equal {int((i['status'] == io['status']).sum())}/{B}, iterations equal "
          f"{int((i['iterations'] == io['iterations']).sum())}/{B}, forces within 1e-6 N: {int((e < 1e-6).sum())}/{B} "
          f"({100 * (e < 1e-6).mean():.1f} %), median {np.median(e):.2e}, worst {e.max():.2e} N", flush=True)
    for Bt in (1, 256, 1024):
        rt = pkg.random_go1_trot_states(Bt, config_id=3 if N > 12 else 2)
        ms = []
        for r in range(6):
            s.solve(rt)
            if r >= 2:
                ms.append(s.last_kernel_ms())
        print(f"    B={Bt} [{s.kernel_for_batch(Bt)}]: kernel {np.median(ms):.3f} ms -> {Bt / np.median(ms) / 1e3:.3f} M solves/s", flush=True)
    s.close()
