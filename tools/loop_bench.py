#!/usr/bin/env python3
"""Throughput of the device-resident closed loop on robots with DIFFERENT commands (the Monte-Carlo use): robot-ticks/s of
qmpc_loop_run_device, state resident in HBM.  QMPC_LOOP_FUSED=0 selects the per-tick launch sequence, the default is
the persistent wave-per-robot kernel.  Run on the GPU box:  python tools/loop_bench.py [--robots 1024] [--ticks 200]"""
import argparse
import sys
import time
from pathlib import Path

import numpy as np
import torch

REPO = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(REPO))
sys.path.insert(0, str(REPO / "tests"))
from conftest import load_pkg  # noqa: E402

pkg = load_pkg()
ap = argparse.ArgumentParser()
ap.add_argument("--robots", type=int, default=1024)
ap.add_argument("--ticks", type=int, default=200)
ap.add_argument("--horizon", type=int, default=10)
ap.add_argument("--mu0", type=float, default=0.0, help="initial barrier parameter (0: the default of qmpc_default_params, 1e-2)")
ap.add_argument("--warm", type=int, default=0, nargs="?", const=1, help="qmpc_loop_params.warm_start")
ap.add_argument("--quirk", action="store_true", help="params.drop_ang_vel = 1: the reference's x_init without angular velocity")
ap.add_argument("--tol-step", type=float, default=0.0, help="params.tol_step (0: the default 1e-8 N)")
ap.add_argument("--model", choices=["quat", "convex"], default="quat", help="which controller's tick (handle model)")
ap.add_argument("--sigma-fast", type=float, default=0.0, help="params.ipm_sigma_fast (0: the default)")
ap.add_argument("--mode", type=int, default=0, help="0 converged, 1 reference (AL-iLQR, <= 10 iterations)")
a = ap.parse_args()
lib = pkg.load_library()
lp = pkg.default_loop_params(lib)
lp.warm_start = float(a.warm)
rng = np.random.default_rng(11)
B = a.robots
cmds = np.zeros((B, 7))
cmds[:, 0] = (0.6 if a.model == "convex" else 1.0) * rng.uniform(-0.5, 0.5, B); cmds[:, 1] = rng.uniform(-0.2, 0.2, B)
cmds[:, 2] = rng.uniform(0.26, 0.32, B)
cmds[:, 5] = rng.uniform(-0.5, 0.5, B); cmds[:, 6] = (rng.random(B) < 0.9).astype(float)
cmds[cmds[:, 6] == 0, :2] = 0.0
cmds[cmds[:, 6] == 0, 5] = 0.0
stand = cmds.copy(); stand[:, 6] = 0.0
st = pkg.loop_states(stand, lp, height=0.3, yaw=rng.uniform(-3.1, 3.1, B), lib=lib)
prm = (pkg.default_convex_params if a.model == "convex" else pkg.default_params)(a.horizon, a.mode, lib)
if a.model == "quat":
    prm.drop_ang_vel = 1 if a.quirk else 0     # default: the MPC sees the body's angular velocity
if a.mu0 > 0.0:
    prm.ipm_mu0 = a.mu0
if a.tol_step > 0.0:
    prm.tol_step = a.tol_step
if a.sigma_fast > 0.0:
    prm.ipm_sigma_fast = a.sigma_fast
s = pkg.Solver(prm, B, device=0, lib=lib)
st = s.loop_run(st, 8, lp)
st["movement_mode"] = cmds[:, 6]
d_st = torch.from_numpy(st.view(np.uint8).reshape(B, -1).copy()).cuda()
s.loop_run_device(B, d_st.data_ptr(), 40, lp)          # into the gait
s.wait()
torch.cuda.synchronize()
t0 = time.perf_counter()
s.loop_run_device(B, d_st.data_ptr(), a.ticks, lp)
s.wait()
dt = time.perf_counter() - t0
out = d_st.cpu().numpy().view(pkg.LOOP_STATE_DTYPE).reshape(B)
import os
print(f"closed loop{' (ConvexMpc)' if a.model == 'convex' else ''}{' (reference mode)' if a.mode else ''}, {B} robots with random commands, {a.ticks} ticks, N={a.horizon}{f', mu0={a.mu0:g}' if a.mu0 > 0 else ''}{f', tol_step={a.tol_step:g}' if a.tol_step > 0 else ''}{f', sigma_fast={a.sigma_fast:g}' if a.sigma_fast > 0 else ''}{', drop_ang_vel=1' if a.quirk else ''}{', warm start' if a.warm else ''}, "
      f"{ {'0': 'per-tick launches (QMPC_LOOP_FUSED=0)', '1': 'persistent kernel (QMPC_LOOP_FUSED=1)'}.get(os.environ.get('QMPC_LOOP_FUSED'), 'library default') }: {dt * 1e3 / a.ticks:.3f} ms per tick, "
      f"{B * a.ticks / dt:.4g} robot-ticks/s; last-tick status != OK {int((out['status'] != 0).sum())}, mean iterations "
      f"{out['iterations'].mean():.2f} (max {int(out['iterations'].max())}); checksum {float(out['pos_world'].sum()):.12f}")
s.close()
