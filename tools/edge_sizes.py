#!/usr/bin/env python3
"""Default path (lane kernel + hand-off) at batch sizes on and around the switch-overs and at sizes that leave the last wavefront
partly filled: every instance converged, forces and iteration words equal to those of the same instances inside a 40000-instance
(full-wavefront) launch.  GPU box: python tools/edge_sizes.py"""
import sys, numpy as np
sys.path.insert(0, '.')
import __graft_entry__ as g
pkg = g._load_pkg(); lib = pkg.load_library()
for N, sizes in ((10, (14336, 14337, 20011, 32767, 32768, 32769)), (20, (14848, 14849, 32767))):
    p = pkg.default_params(N, 0, lib)
    rec = pkg.random_go1_trot_states(40000, config_id=4 if N == 10 else 3)
    s = pkg.Solver(p, 40000, 0, lib)
    ff, fi = s.solve(rec)
    for B in sizes:
        f, info = s.solve(rec[:B])
        ok = (info["status"] == 0).all()
        d = np.abs(f - ff[:B]).max()
        print(f"N={N} B={B}: kernel {s.kernel_for_batch(B)}, all converged {ok}, max |f - full launch| {d:.2e} N, iterations equal {np.array_equal(info['iterations'], fi['iterations'][:B])}", flush=True)
    s.close()
