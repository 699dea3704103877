#!/usr/bin/env python3
"""Forces / iteration words of fixed batches through the LANE kernel (QMPC_VARIANT=4: pure lane kernel, every launch form the
size selects: 64-lane wavefronts, lane pairs, warm start, the other two models), for bit comparisons between builds while the
passes are restructured:  QMPC_LIB=... tools/lane_bits.py out.npz   |   tools/lane_bits.py --compare ref.npz"""
import os
import sys
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from bench import load_pkg  # noqa: E402

os.environ["QMPC_VARIANT"] = "4"
pkg = load_pkg()
lib = pkg.load_library()
out = {}
for name, N, B, cfg in (("q10_pair", 10, 3000, 4), ("q10_full", 10, 66000, 4), ("q20_pair", 20, 2000, 3), ("q20_full", 20, 66000, 3)):
    p = pkg.default_params(N, 0, lib)
    s = pkg.Solver(p, B, 0, lib)
    f, info = s.solve(pkg.random_go1_trot_states(B, config_id=cfg))
    out[name] = f
    out[name + "_it"] = info["iterations"]
    s.close()
for name, N, B in (("c20", 20, 66000), ("c10_half", 10, 5000)):
    p = pkg.default_convex_params(N, 0, lib)
    s = pkg.Solver(p, B, 0, lib)
    f, info = s.convex_solve(pkg.random_go1_convex_states(B, config_id=12))
    out[name] = f
    out[name + "_it"] = info["iterations"]
    s.close()
for name, N, B in (("b16", 16, 66000), ("b16_half", 16, 3000)):
    p = pkg.default_biped8_params(N, 0, lib)
    s = pkg.Solver(p, B, 0, lib)
    f, info = s.solve8(pkg.random_biped8_states(B, config_id=5))
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
