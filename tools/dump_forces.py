#!/usr/bin/env python3
"""Dump the forces of fixed synthetic batches (regression aid while refactoring kernels):
usage: tools/dump_forces.py out.npz   |   tools/dump_forces.py --compare ref.npz"""
import sys
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from bench import load_pkg  # noqa: E402

pkg = load_pkg()
lib = pkg.load_library()
out = {}
for name, N, B in (("q10", 10, 1024), ("q20", 20, 512), ("q10big", 10, 6000), ("q1", 1, 256), ("q32", 32, 128)):
    p = pkg.default_params(N, 0, lib)
    s = pkg.Solver(p, B, 0, lib)
    f, info = s.solve(pkg.random_go1_trot_states(B, config_id=2))
    out[name] = f
    out[name + "_it"] = info["iterations"]
    s.close()
for name, N, B in (("c20", 20, 512), ("c10big", 10, 5000)):
    p = pkg.default_convex_params(N, 0, lib)
    s = pkg.Solver(p, B, 0, lib)
    f, info = s.convex_solve(pkg.random_go1_convex_states(B, config_id=12))
    out[name] = f
    out[name + "_it"] = info["iterations"]
    s.close()
if sys.argv[1] == "--compare":
    ref = np.load(sys.argv[2])
    for k in out:
        same = np.array_equal(ref[k], out[k])
        print(k, "bit-identical" if same else "DIFFERS max %.3e" % np.abs(ref[k].astype(float) - out[k]).max())
else:
    np.savez(sys.argv[1], **out)
    print("saved", sys.argv[1])
