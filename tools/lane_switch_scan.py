#!/usr/bin/env python3
"""Switch-over between the wave-per-instance kernels and the lane kernel (+ hand-off) in converged mode: the library's own
choice with QMPC_LANE_MIN huge (wave side) and 1 (lane side), same records, kernel-side time between the handle's events.
GPU box:  python tools/lane_switch_scan.py [--cases N:B,...]"""
import argparse, os, sys
from pathlib import Path
import numpy as np
REPO = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(REPO))
import __graft_entry__ as g  # noqa: E402
ap = argparse.ArgumentParser()
ap.add_argument("--cases", default="10:12288,10:16384,10:20480,10:24576,16:12288,16:16384,16:20480,20:12288,20:16384,20:20480,24:12288,24:16384")
ap.add_argument("--reps", type=int, default=5)
a = ap.parse_args()
pkg = g._load_pkg(); lib = pkg.load_library()
import torch  # noqa: E402
for case in a.cases.split(","):
    N, B = (int(x) for x in case.split(":"))
    p = pkg.default_params(N, 0, lib)
    rec = pkg.random_go1_trot_states(B, config_id=3 if N == 20 else 4)
    d_in = torch.from_numpy(rec.view(np.float64).reshape(B, -1).copy()).cuda()
    out = {}
    for tag, lm in (("wave", str(1 << 30)), ("lane", "1")):
        os.environ["QMPC_LANE_MIN"] = lm
        s = pkg.Solver(p, B, 0, lib)
        d_f = torch.zeros(B, 12, dtype=torch.float64, device="cuda"); d_i = torch.zeros(B, 5, dtype=torch.float64, device="cuda")
        ms = []
        for r in range(a.reps):
            s.solve_device(B, d_in.data_ptr(), d_f.data_ptr(), d_i.data_ptr()); s.wait()
            if r >= 2: ms.append(s.last_kernel_ms())
        out[tag] = (float(np.median(ms)), d_f.cpu().numpy(), s.kernel_for_batch(B))
        s.close()
    os.environ.pop("QMPC_LANE_MIN", None)
    (mw, fw, kw), (ml, fl, kl) = out["wave"], out["lane"]
    print(f"N={N} B={B}: {kw} {mw:.3f} ms ({B / mw / 1e3:.3f} M/s) vs {kl} {ml:.3f} ms ({B / ml / 1e3:.3f} M/s); max |df| {np.abs(fw - fl).max():.1e} N", flush=True)
