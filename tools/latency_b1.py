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
SAMPLES = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
for name, N, mode, mk, gen, call in (
        ("QuatMpc   N=10 converged", 10, 0, pkg.default_params, pkg.random_go1_trot_states, "qmpc_solve"),
        ("QuatMpc   N=20 converged", 20, 0, pkg.default_params, pkg.random_go1_trot_states, "qmpc_solve"),
        ("QuatMpc   N=10 reference mode (AL-iLQR <= 10 iterations, QuatMpc.cpp:21-26)", 10, 1, pkg.default_params, pkg.random_go1_trot_states, "qmpc_solve"),
        ("QuatMpc   N=20 reference mode (the reference's own operating point, gazebo_go1_quat_mpc.yaml:36-37)", 20, 1, pkg.default_params,
         pkg.random_go1_trot_states, "qmpc_solve"),
        ("ConvexMpc N=20 converged", 20, 0, pkg.default_convex_params, pkg.random_go1_convex_states, "qmpc_convex_solve")):
    p = mk(N, mode, lib)
    s = pkg.Solver(p, 1, 0, lib)
    recs = gen(SAMPLES, config_id=2 if call == "qmpc_solve" else 12)
    fn = getattr(lib, call)
    for kind in ("pinned", "pageable"):
        # caller-owned buffers, as the C++ host class holds them: pinned (qmpc_host_alloc) or plain members
        if kind == "pinned":
            hin = s.pinned((1,), recs.dtype); hf = s.pinned((1, 12)); hi = s.pinned((1,), pkg.INFO_DTYPE)
        else:
            hin = np.zeros(1, dtype=recs.dtype); hf = np.zeros((1, 12)); hi = np.zeros(1, dtype=pkg.INFO_DTYPE)
        a = (s._h, 1, hin.ctypes.data, hf.ctypes.data, hi.ctypes.data)
        lat, its = np.zeros(SAMPLES), np.zeros(SAMPLES, dtype=int)
        first = None
        for i in range(-20, SAMPLES):          # 20 untimed warm-up calls (the first one allocates and loads code objects)
            hin[0] = recs[max(i, 0)]
            t0 = time.perf_counter()
            rc = fn(*a)
            dt = time.perf_counter() - t0
            assert rc == 0
            if first is None:
                first = dt
            if i >= 0:
                lat[i] = dt * 1e3
                its[i] = hi["iterations"][0]
        worst = int(lat.argmax())
        print(f"{name}, {kind} buffers: batch-1 blocking latency mean {lat.mean():.3f} ms, p50 {np.median(lat):.3f}, p99 {np.percentile(lat, 99):.3f}, "
              f"max {lat.max():.3f} ms (call {worst}, {its[worst]} iterations); mean iterations {its.mean():.1f}, max {its.max()}; "
              f"very first call {first * 1e3:.1f} ms; kernel family {s.kernel_for_batch(1)}", flush=True)
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
