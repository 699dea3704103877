#!/usr/bin/env python3
"""Reference mode on the lane passes: forces and status / iteration words of fixed batches (QMPC_VARIANT=4), to compare the pair
form (QMPC_LANE_PAIR=1, half-filled wavefronts) with the masked plain form (QMPC_LANE_PAIR=0) and with a full-wavefront launch
that contains the same instances:  tools/refmode_pair_bits.py out.npz | --compare ref.npz"""
import os, sys
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from bench import load_pkg  # noqa: E402
os.environ["QMPC_VARIANT"] = "4"
pkg = load_pkg(); lib = pkg.load_library()
out = {}
for name, N, B, cfg in (("r10", 10, 30000, 2), ("r20", 20, 20000, 3), ("r10_4", 10, 9000, 4)):
    p = pkg.default_params(N, pkg.MODE_REFERENCE, lib)
    rec = pkg.random_go1_trot_states(70000, config_id=cfg)
    s = pkg.Solver(p, 70000, 0, lib)
    f, info = s.solve(rec[:B])                 # half-filled wavefronts
    out[name] = f; out[name + "_st"] = info["status"]; out[name + "_it"] = info["iterations"]
    ff, fi = s.solve(rec)                      # full wavefronts: the plain form, whatever QMPC_LANE_PAIR says
    out[name + "_full"] = ff[:B]; out[name + "_full_st"] = fi["status"][:B]; out[name + "_full_it"] = fi["iterations"][:B]
    s.close()
    print(name, "shard == its block of the full launch:", np.array_equal(f, ff[:B]), np.array_equal(info["status"], fi["status"][:B]),
          np.array_equal(info["iterations"], fi["iterations"][:B]))
if sys.argv[1] == "--compare":
    ref = np.load(sys.argv[2])
    for k in out:
        print(k, "bit-identical" if np.array_equal(ref[k], out[k]) else "DIFFERS max %.3e" % np.abs(ref[k].astype(float) - out[k]).max())
else:
    np.savez(sys.argv[1], **out); print("saved")
