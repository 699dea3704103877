#!/usr/bin/env python3
"""Reference mode (AL-iLQR, <= 10 iterations): the lane-per-instance kernel (QMPC_VARIANT=4) against the wave-per-instance
kernels (QMPC_LANE_REF_MIN huge), kernel time, status words / iteration counts / forces against each other and against the
oracle on a sample.  GPU box:  python tools/refmode_lane_bench.py [--cases N:B,...]"""
import argparse, os, sys
from pathlib import Path
import numpy as np
REPO = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(REPO))
import __graft_entry__ as g  # noqa: E402
ap = argparse.ArgumentParser()
ap.add_argument("--cases", default="10:4096,10:8192,10:16384,10:32768,10:65536,20:16384,20:65536")
ap.add_argument("--sample", type=int, default=128)
ap.add_argument("--model", default="quat", choices=("quat", "convex", "biped8"))       # convex: ConvexMpc's own mode (five iterations)
a = ap.parse_args()
pkg = g._load_pkg(); lib = pkg.load_library()
from oracle import pyoracle  # noqa: E402  (checker)
import torch  # noqa: E402
for case in a.cases.split(","):
    N, B = (int(x) for x in case.split(":"))
    cvx, b8 = a.model == "convex", a.model == "biped8"
    p = (pkg.default_convex_params if cvx else pkg.default_biped8_params if b8 else pkg.default_params)(N, pkg.MODE_REFERENCE, lib)
    rec = (pkg.random_go1_convex_states(B, config_id=12) if cvx else pkg.random_biped8_states(B, config_id=5) if b8
           else pkg.random_go1_trot_states(B, config_id=3 if N == 20 else 2))
    nu = 24 if b8 else 12
    d_in = torch.from_numpy(rec.view(np.float64).reshape(B, -1).copy()).cuda()
    res = {}
    for tag, env in (("wave", {"QMPC_LANE_REF_MIN": str(1 << 30)}), ("lane", {"QMPC_VARIANT": "4"})):
        for k in ("QMPC_LANE_REF_MIN", "QMPC_VARIANT"):
            os.environ.pop(k, None)
        os.environ.update(env)
        s = pkg.Solver(p, B, 0, lib)
        d_f = torch.zeros(B, nu, dtype=torch.float64, device="cuda"); d_i = torch.zeros(B, 5, dtype=torch.float64, device="cuda")
        ms = []
        for r in range(5):
            (s.convex_solve_device if cvx else s.solve8_device if b8 else s.solve_device)(B, d_in.data_ptr(), d_f.data_ptr(), d_i.data_ptr()); s.wait()
            if r >= 2: ms.append(s.last_kernel_ms())
        res[tag] = (d_f.cpu().numpy(), d_i.cpu().numpy().view(pkg.INFO_DTYPE).reshape(B), float(np.median(ms)))
        s.close()
    (f0, i0, m0), (f1, i1, m1) = res["wave"], res["lane"]
    d = np.abs(f1 - f0).max(axis=1)
    idx = np.arange(0, B, max(B // a.sample, 1))[:a.sample]
    fo, io = (pyoracle.convex_solve(pyoracle.default_convex_params(N, 1), rec[idx], threads=8) if cvx
              else pyoracle.solve8(pyoracle.default_biped8_params(N, 1), rec[idx], threads=8) if b8
              else pyoracle.solve(pyoracle.default_params(N, 1), rec[idx], threads=8))
    do = np.abs(f1[idx] - fo).max(axis=1)
    same_o = (i1["status"][idx] == io["status"]) & (i1["iterations"][idx] == io["iterations"])
    print(f"reference mode ({a.model}) N={N} B={B}: wave kernels {m0:.3f} ms ({B / m0 / 1e3:.3f} M/s), lane kernel {m1:.3f} ms ({B / m1 / 1e3:.3f} M/s); "
          f"lane vs wave: status equal {100 * (i0['status'] == i1['status']).mean():.2f} %, iterations equal {100 * (i0['iterations'] == i1['iterations']).mean():.2f} %, "
          f"forces within 1e-6 N {100 * (d < 1e-6).mean():.1f} % (median {np.median(d):.1e}); lane vs oracle ({len(idx)}): status+iterations equal "
          f"{int(same_o.sum())}/{len(idx)}, within 1e-6 N {int((do < 1e-6).sum())}, worst {do.max():.1e}; mean iterations {i1['iterations'].mean():.2f}", flush=True)
