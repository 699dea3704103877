#!/usr/bin/env python3
"""Reference mode (AL-iLQR, <= 10 iterations) on the device: wrench-form kernels (QMPC_WFORM=1) against the round-1 kernels
(QMPC_WFORM=0), kernel time and parity against each other.  GPU box:  python tools/refmode_bench.py [--cases N:B,...]"""
import argparse, os, sys
from pathlib import Path
import numpy as np
REPO = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(REPO))
import __graft_entry__ as g  # noqa: E402
ap = argparse.ArgumentParser()
ap.add_argument("--cases", default="10:1024,10:2048,10:8192,10:32768,12:1024,3:1024,20:1024,20:16384")
a = ap.parse_args()
pkg = g._load_pkg(); lib = pkg.load_library()
import torch  # noqa: E402
for case in a.cases.split(","):
    N, B = (int(x) for x in case.split(":"))
    p = pkg.default_params(N, pkg.MODE_REFERENCE, lib)
    rec = pkg.random_go1_trot_states(B, config_id=3 if N == 20 else 2)
    d_in = torch.from_numpy(rec.view(np.float64).reshape(B, -1).copy()).cuda()
    res = {}
    os.environ["QMPC_LANE_REF_MIN"] = str(1 << 30)      # the wave kernels at every size
    for wf in (0, 1):
        os.environ["QMPC_WFORM"] = str(wf)
        s = pkg.Solver(p, B, 0, lib)
        d_f = torch.zeros(B, 12, dtype=torch.float64, device="cuda"); d_i = torch.zeros(B, 5, dtype=torch.float64, device="cuda")
        ms = []
        for r in range(6):
            s.solve_device(B, d_in.data_ptr(), d_f.data_ptr(), d_i.data_ptr()); s.wait()
            if r >= 2: ms.append(s.last_kernel_ms())
        res[wf] = (d_f.cpu().numpy(), d_i.cpu().numpy().view(pkg.INFO_DTYPE).reshape(B), float(np.median(ms)))
        s.close()
    (f0, i0, m0), (f1, i1, m1) = res[0], res[1]
    d = np.abs(f1 - f0).max(axis=1)
    print(f"reference mode N={N} B={B}: round-1 kernels {m0:.3f} ms ({B / m0 / 1e3:.3f} M/s), wrench form {m1:.3f} ms ({B / m1 / 1e3:.3f} M/s); "
          f"status equal {np.array_equal(i0['status'], i1['status'])}, iterations equal {100 * (i0['iterations'] == i1['iterations']).mean():.2f} %, "
          f"forces within 1e-6 N on {100 * (d < 1e-6).mean():.1f} % (median {np.median(d):.1e}, worst {d.max():.1e})", flush=True)
