#!/usr/bin/env python3
"""Soak: many seeded batches through the HIP path and the CPU oracle, every instance compared (status, iteration
count, force L-inf).  Looks for rare-event differences the fixed-seed parity tests cannot see.
Run on the GPU box:  python tools/soak.py [--instances 500000] [--horizon 10] [--model quat|convex|biped8] [--mode 0|1]
                     python tools/soak.py --closed-loop [--robots 4096] [--ticks 1000]
--mode 1 soaks the reference mode (truncated AL-iLQR iterate: agreement is counted at 1e-6 N, not asserted per instance);
--closed-loop runs the device-resident loop for many robots with random commands and checks a sample of them against the
host classes tick for tick."""
import argparse
import importlib
import sys
import time
from pathlib import Path

import numpy as np

REPO = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(REPO))
sys.path.insert(0, str(REPO / "tests"))
from conftest import load_pkg  # noqa: E402

pkg = load_pkg()
from oracle import pyoracle as po  # noqa: E402  (checker)

ap = argparse.ArgumentParser()
ap.add_argument("--instances", type=int, default=500000)
ap.add_argument("--horizon", type=int, default=10)
ap.add_argument("--model", choices=["quat", "convex", "biped8"], default="quat")
ap.add_argument("--chunk", type=int, default=32768)
ap.add_argument("--threads", type=int, default=16)
ap.add_argument("--mode", type=int, default=0, help="0 converged, 1 reference (AL-iLQR, <= 10 iterations)")
ap.add_argument("--closed-loop", action="store_true")
ap.add_argument("--robots", type=int, default=4096)
ap.add_argument("--ticks", type=int, default=1000)
ap.add_argument("--check", type=int, default=4, help="closed loop: robots replayed on the host classes")
ap.add_argument("--warm", action="store_true", help="closed loop: qmpc_loop_params.warm_start = 1 on the device, set_warm_start(true) "
                "on the host classes")
ap.add_argument("--feed-ang-vel", action="store_true", help="closed loop: params.drop_ang_vel = 0 (the MPC sees the angular "
                "velocity; with the reference's quirk the ideal plant is undamped and robots lose balance after 6-9 s)")
a = ap.parse_args()
lib = pkg.load_library()
if a.closed_loop:
    import ctypes as C
    import __graft_entry__ as g

    host = C.CDLL(str(g.build_host()))
    vp = C.c_void_p
    host.qh_loop_create_opts.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_int, vp, vp]; host.qh_loop_create_opts.restype = vp
    for fn in ("qh_loop_tick", "qh_loop_destroy"):
        getattr(host, fn).argtypes = [vp]
    host.qh_loop_export.argtypes = [vp, vp]
    host.qh_loop_set_command.argtypes = [vp, vp, C.c_double]
    host.qh_loop_set_warm_start.argtypes = [vp, C.c_int]
    lp = pkg.default_loop_params(lib)
    lp.warm_start = 1.0 if a.warm else 0.0
    rng = np.random.default_rng(11)
    B = a.robots
    cmds = np.zeros((B, 7))
    cmds[:, 0] = rng.uniform(-0.5, 0.5, B); cmds[:, 1] = rng.uniform(-0.2, 0.2, B); cmds[:, 2] = rng.uniform(0.26, 0.32, B)
    cmds[:, 5] = rng.uniform(-0.5, 0.5, B); cmds[:, 6] = (rng.random(B) < 0.9).astype(float)
    cmds[cmds[:, 6] == 0, :2] = 0.0      # a standing robot is not asked to translate: its feet stay where they are
    cmds[cmds[:, 6] == 0, 5] = 0.0       # (the rigid-body plant has no leg-length limit that would stop it)
    yaws = rng.uniform(-3.1, 3.1, B)
    stand = cmds.copy(); stand[:, 6] = 0.0
    st_init = pkg.loop_states(stand, lp, height=0.3, yaw=yaws, lib=lib)
    prm = pkg.default_params(a.horizon, pkg.MODE_REFERENCE if a.mode else pkg.MODE_CONVERGED, lib)
    prm.drop_ang_vel = 0 if a.feed_ang_vel else 1
    s = pkg.Solver(prm, B, device=0, lib=lib)
    t0 = time.time()
    st = s.loop_run(st_init, 8, lp)
    st["movement_mode"] = cmds[:, 6]
    st0 = st.copy()
    st, tf, tc = s.loop_run(st, a.ticks, lp, trace=True)
    dt = time.time() - t0
    s.close()
    down = (st["pos_world"][:, 2] < 0.15) | ~np.isfinite(st["pos_world"][:, 2])
    fell = int(down.sum())
    if fell:
        print("down by movement_mode:", {int(m): int((down & (cmds[:, 6] == m)).sum()) for m in (0, 1)})
    nonok = int((st["status"] != 0).sum())
    worst_f, cdiff = 0.0, 0
    for i in range(min(a.check, B)):
        h = host.qh_loop_create_opts(str(pkg.LIB_PATH).encode(), a.horizon, pkg.MODE_REFERENCE if a.mode else pkg.MODE_CONVERGED, 0 if a.feed_ang_vel else 1,
                                     C.addressof(lp), st_init[i:i + 1].ctypes.data)
        e = np.zeros(1, dtype=pkg.LOOP_STATE_DTYPE)
        host.qh_loop_set_warm_start(h, 1 if a.warm else 0)
        for _ in range(8):
            host.qh_loop_tick(h)
        host.qh_loop_set_warm_start(h, 1 if a.warm else 0)      # a new device call starts cold: drop the kept solution
        host.qh_loop_set_command(h, np.ascontiguousarray(cmds[i, :6]).ctypes.data, float(cmds[i, 6]))
        for t in range(a.ticks):
            host.qh_loop_tick(h)
            host.qh_loop_export(h, e.ctypes.data)
            cdiff += int((e[0]["contacts"] != tc[t, i]).sum())
            worst_f = max(worst_f, float(np.abs(e[0]["forces_body"] - tf[t, i]).max()))
        host.qh_loop_destroy(h)
    dist = np.linalg.norm(st["pos_world"][:, :2] - st0["pos_world"][:, :2], axis=1)[~down]
    print(f"closed-loop soak{' (reference mode)' if a.mode else ''}{' (angular velocity fed to the MPC)' if a.feed_ang_vel else ''}{' (warm start)' if a.warm else ''}: {B} robots x {a.ticks} ticks ({a.ticks * 0.005:.1f} s of robot time) in {dt:.1f} s = "
          f"{B * a.ticks / dt:.3g} robot-ticks/s incl. traces; robots down {fell}, last-tick solver status != OK {nonok}; "
          f"distance walked median {np.median(dist):.3f} m, max {dist.max():.3f} m; {min(a.check, B)} robots replayed on the host "
          f"classes: contact-flag differences {cdiff}, worst force difference {worst_f:.3e} N")
    sys.exit(0)
gen = {"quat": pkg.random_go1_trot_states, "convex": pkg.random_go1_convex_states, "biped8": pkg.random_biped8_states}[a.model]
dp = {"quat": "default_params", "convex": "default_convex_params", "biped8": "default_biped8_params"}[a.model]
sv = {"quat": "solve", "convex": "convex_solve", "biped8": "solve8"}[a.model]
p = getattr(pkg, dp)(a.horizon, a.mode, lib)
po_p = p if a.mode else getattr(po, dp)(a.horizon, 0)
assert a.mode or bytes(p) == bytes(po_p)
n_within = 0
s = pkg.Solver(p, a.chunk, device=0, lib=lib)
done, worst, n_status_diff, n_iter_diff, n_fail_gpu, n_fail_cpu = 0, 0.0, 0, 0, 0, 0
t0 = time.time()
cfg = 1000
while done < a.instances:
    n = min(a.chunk, a.instances - done)
    rec = gen(n, config_id=cfg, first=done)
    f, info = getattr(s, sv)(rec)
    fo, io = getattr(po, sv)(po_p, rec, threads=a.threads)
    both = ((info["status"] == 0) & (io["status"] == 0)) if a.mode == 0 else np.ones(n, dtype=bool)
    n_within += int((np.abs(f - fo).max(axis=1) < 1e-6).sum())
    n_status_diff += int((info["status"] != io["status"]).sum())
    n_iter_diff += int((info["iterations"] != io["iterations"]).sum())
    n_fail_gpu += int((info["status"] != 0).sum())
    n_fail_cpu += int((io["status"] != 0).sum())
    if both.any():
        if a.model == "biped8":
            feet = rec["foot_pos_body"].reshape(-1, 8, 3)
            wr = lambda F: np.concatenate([F.reshape(-1, 8, 3).sum(1), np.cross(feet, F.reshape(-1, 8, 3)).sum(1)], axis=1)
            e = np.abs(wr(f) - wr(fo))[both].max()
        else:
            e = np.abs(f - fo)[both].max()
        worst = max(worst, float(e))
    done += n
s.close()
print(f"soak {a.model} N={a.horizon}: {done} instances in {time.time() - t0:.0f} s; GPU failures {n_fail_gpu}, oracle failures "
      f"{n_fail_cpu}, status differences {n_status_diff}, iteration-count differences {n_iter_diff} "
      f"({100.0 * n_iter_diff / done:.3f} %), worst force difference {worst:.3e} "
      f"{'N (foot wrench)' if a.model == 'biped8' else 'N'}; within 1e-6 N: {n_within} ({100.0 * n_within / done:.3f} %)"
      + (" [reference mode: truncated iterates]" if a.mode else ""))
