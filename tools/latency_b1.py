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
