#!/usr/bin/env python3
"""Blocking single-instance latency of the drop-in call (qmpc_solve, batch = 1: H2D + kernel + D2H), i.e. what
legged::QuatMpcHip::grf_update pays per MPC tick; the reference's mpc_thread has a 5 ms budget (Main.cpp:101-119)."""
import sys
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from bench import load_pkg  # noqa: E402

pkg = load_pkg()
lib = pkg.load_library()
for name, N, mk, gen, call in (
        ("QuatMpc   N=10", 10, pkg.default_params, pkg.random_go1_trot_states, "solve"),
        ("QuatMpc   N=20", 20, pkg.default_params, pkg.random_go1_trot_states, "solve"),
        ("ConvexMpc N=20", 20, pkg.default_convex_params, pkg.random_go1_convex_states, "convex_solve")):
    p = mk(N, 0, lib)
    s = pkg.Solver(p, 1, 0, lib)
    recs = gen(200, config_id=2 if call == "solve" else 12)
    getattr(s, call)(recs[:1])
    lat, its = [], []
    for i in range(200):
        t0 = time.perf_counter()
        f, info = getattr(s, call)(recs[i:i + 1])
        lat.append(time.perf_counter() - t0)
        its.append(int(info["iterations"][0]))
    lat = np.array(lat) * 1e3
    print(f"{name}: batch-1 blocking latency mean {lat.mean():.3f} ms, p50 {np.median(lat):.3f}, p99 {np.percentile(lat, 99):.3f}, "
          f"max {lat.max():.3f} ms; mean iterations {np.mean(its):.1f}")
    s.close()

# ---- one trotting robot, the whole tick on the device (front end + solve + plant, qmpc_loop_run_device with batch 1): cold
# (the reference's start from u_ref) and warm-started with a low initial barrier -- the latency floor of a tick in the gait
import torch  # noqa: E402

for N in (10, 20):
    lp = pkg.default_loop_params(lib)
    prm = pkg.default_params(N, 0, lib)
    prm.drop_ang_vel = 0
    st = pkg.loop_states([[0.3, 0.05, 0.30, 0.0, 0.0, 0.2, 0.0]], lp, lib=lib)
    sl = pkg.Solver(prm, 1, 0, lib)
    st = sl.loop_run(st, 8, lp)
    st["movement_mode"] = 1.0
    st = sl.loop_run(st, 100, lp)
    sl.close()
    for name, warm, mu0 in (("cold, mu0 1e-2 (reference semantics)", 0.0, None), ("warm start, mu0 1e-6", 1.0, 1e-6)):
        p2 = pkg.default_params(N, 0, lib)
        p2.drop_ang_vel = 0
        if mu0:
            p2.ipm_mu0 = mu0
        lpw = pkg.default_loop_params(lib)
        lpw.warm_start = warm
        s2 = pkg.Solver(p2, 1, 0, lib)
        d2 = torch.from_numpy(st.view(np.uint8).reshape(1, -1).copy()).cuda()
        s2.loop_run_device(1, d2.data_ptr(), 20, lpw); s2.wait()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        s2.loop_run_device(1, d2.data_ptr(), 400, lpw); s2.wait()
        dt = (time.perf_counter() - t0) / 400
        it = d2.cpu().numpy().view(pkg.LOOP_STATE_DTYPE).reshape(1)["iterations"][0]
        print(f"QuatMpc N={N}, one trotting robot, whole tick on the device (front end + solve + plant), {name}: "
              f"{dt * 1e3:.3f} ms per tick (last tick: {int(it)} iterations)")
        s2.close()
