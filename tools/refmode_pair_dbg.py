import os, sys
import numpy as np
sys.path.insert(0, os.getcwd())
from bench import load_pkg
os.environ["QMPC_VARIANT"] = "4"
pkg = load_pkg(); lib = pkg.load_library()
p = pkg.default_params(10, pkg.MODE_REFERENCE, lib)
p.iterations_max = int(sys.argv[2])
rec = pkg.random_go1_trot_states(3000, config_id=2)
s = pkg.Solver(p, 3000, 0, lib)
f, info, tu, tx = s.solve(rec, want_traj=True)
if sys.argv[1] == "save":
    np.savez("/tmp/refdbg.npz", tu=tu, tx=tx)
else:
    r = np.load("/tmp/refdbg.npz")
    du = np.abs(tu - r["tu"]); dx = np.abs(tx - r["tx"])
    print("iters", sys.argv[2], "max |dU| per knot:", du.max(axis=(0, 2)))
    print("max |dX| per knot:", dx.max(axis=(0, 2)))
    bad = np.where(du.max(axis=(1, 2)) > 0)[0]
    print("instances differing:", len(bad), "of 3000; stance counts of the first:", [int(rec["contacts"][b].sum()) for b in bad[:10]])
