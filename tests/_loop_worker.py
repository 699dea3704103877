"""Worker of test_persistent_loop_kernel_equals_the_per_tick_sequence: runs the device-resident closed loop for robots
with different commands and prints a SHA-256 of the final states and of the force / contact traces.  The launch form is
chosen by the environment (QMPC_LOOP_FUSED=0 per-tick kernels, =1 persistent kernel), read once per process."""
import hashlib
import sys
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent))
from conftest import load_pkg  # noqa: E402

pkg = load_pkg()
robots, ticks, horizon = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
mode = int(sys.argv[4]) if len(sys.argv) > 4 else 0          # 0 converged, 1 reference (AL-iLQR, <= 10 iterations)
convex = len(sys.argv) > 5 and sys.argv[5] == "convex"       # the sibling controller (ConvexMpc handle)
warm = len(sys.argv) > 6 and sys.argv[6] == "warm"           # qmpc_loop_params.warm_start
lib = pkg.load_library()
lp = pkg.default_loop_params(lib)
lp.warm_start = 1.0 if warm else 0.0
rng = np.random.default_rng(5)
cmds = np.zeros((robots, 7))
cmds[:, 0] = rng.uniform(-0.5, 0.5, robots); cmds[:, 1] = rng.uniform(-0.2, 0.2, robots)
cmds[:, 2] = rng.uniform(0.26, 0.32, robots); cmds[:, 5] = rng.uniform(-0.5, 0.5, robots)
if convex:
    cmds[:, 0] *= 0.6                        # ConvexMpc's gains (yaml) are tuned for gentler walking
cmds[:, 6] = (rng.random(robots) < 0.85).astype(float)
cmds[cmds[:, 6] == 0, :2] = 0.0
cmds[cmds[:, 6] == 0, 5] = 0.0
stand = cmds.copy(); stand[:, 6] = 0.0
st = pkg.loop_states(stand, lp, height=0.3, yaw=rng.uniform(-3, 3, robots), lib=lib)
if robots > 2:
    st["quat"][2] = np.nan                   # a robot whose records are rejected every tick (QMPC_NAN_INPUT): it keeps ticking
s = pkg.Solver((pkg.default_convex_params if convex else pkg.default_params)(horizon, mode, lib), robots, device=0, lib=lib)
st = s.loop_run(st, 6, lp)
st["movement_mode"] = cmds[:, 6]
st, tf, tc = s.loop_run(st, ticks, lp, trace=True)
s.close()
ok = np.ones(robots, dtype=bool)
if robots > 2:
    ok[2] = False
    assert st["status"][2] == pkg.NAN_INPUT and st["tick"][2] == 6 + ticks
assert (st["tick"] == 6 + ticks).all()
if mode == 0:
    assert (st["status"][ok] == 0).all()
else:            # truncated iterates: OK, MAX_ITER (1) and LINESEARCH_FAIL (4) are all applied, as the reference does
    assert np.isin(st["status"][ok], (0, 1, 4)).all() and (np.abs(st["pos_world"][ok][:, 2] - 0.29) < 0.08).all()
print("SHA", hashlib.sha256(st.tobytes() + tf.tobytes() + tc.tobytes()).hexdigest(),
      "swing-ticks", int((tc[:, ok] == 0).sum()), "walked", f"{float(np.abs(st['pos_world'][ok][:, :2]).max()):.3f}")
