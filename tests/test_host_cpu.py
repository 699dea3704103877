"""CPU suite: the C++ host mirror of the reference's call surface
(quaternion-mpc_amd/host: QuatMpcHipT<State>, LeggedContactFSMHip,
MovingWindowFilterHip) driven through host/libqmpc_host.so.

The contact schedule must be BIT-EXACT (integer/FP64 state machine); it is
checked against an independent Python restatement of
legged_ctrl/src/utils/LeggedContactFSM.cpp, tick for tick.
"""
import ctypes as C
from collections import deque
from pathlib import Path

import numpy as np
import pytest

REPO = Path(__file__).resolve().parent.parent


@pytest.fixture(scope="module")
def host():
    import __graft_entry__ as g

    p = g.build_host()
    lib = C.CDLL(str(p))
    vp = C.c_void_p
    lib.qh_create.argtypes = [C.c_char_p, C.c_int]
    lib.qh_create.restype = vp
    lib.qh_destroy.argtypes = [vp]
    lib.qh_device_status.argtypes = [vp]
    lib.qh_set_feedback.argtypes = [vp, vp]
    lib.qh_set_command.argtypes = [vp, vp, C.c_double]
    for f in ("qh_goal_update", "qh_foot_update"):
        getattr(lib, f).argtypes = [vp]
    lib.qh_grf_update.argtypes = [vp]
    lib.qh_update.argtypes = [vp]
    lib.qh_pack_input.argtypes = [vp, vp]
    lib.qh_get_outputs.argtypes = [vp, vp]
    lib.qh_fsm_run.argtypes = [C.c_double, C.c_int, vp, vp, vp, vp]
    lib.qh_filter_run.argtypes = [C.c_int, C.c_int, vp, vp]
    return lib


# ---- independent restatement of LeggedContactFSM (schedule part) ---------------
class PyFSM:
    SWING, STANCE = 0, 1

    def __init__(self, leg):                      # reset_params + set_default_gait_pattern
        self.pattern = [self.STANCE, self.SWING] if leg in (0, 3) else [self.SWING, self.STANCE]
        self.switch = [0.5, 1.0]
        self.idx, self.prev = 0, 1
        self.s = self.STANCE
        self.phase = 0.0
        self.t0, self.t1 = 0.0, self.switch[0]

    def reset(self):                              # LeggedContactFSM.cpp:11-31
        self.phase = 0.0
        self.idx, self.prev = 0, 1
        self.t0, self.t1 = 0.0, self.switch[0]
        self.s = self.pattern[0]

    def _enter(self):                             # common_enter :208-223
        self.prev = self.idx
        self.idx = (self.idx + 1) % 2
        if self.idx < self.prev:
            self.phase -= 1.0
        self.t0, self.t1 = self.phase, self.switch[self.idx]

    def _pct(self):                               # :261-270
        p = (self.phase - self.t0) / (self.t1 - self.t0)
        return 0.0 if p < 0.0 else (1.0 if p > 1.0 else p)

    def update(self, dt, freq, flag):             # :33-78
        self.phase += freq * dt
        if self.s == self.STANCE:
            if self.phase >= self.t1:
                self._enter(); self.s = self.SWING
        else:
            if self._pct() > 0.9 and flag:
                self.s = self.STANCE; self._enter()
            elif self._pct() >= 1.0:
                self.s = self.STANCE; self._enter()
        return self.phase


def test_contact_schedule_bit_exact(host):
    rng = np.random.default_rng(7)
    T = 6000
    mode = np.ones(T); mode[:40] = 0; mode[3000:3010] = 0      # stand -> trot -> stand -> trot
    # sigmoid-like contact flags: mostly tiny-but-nonzero (any non-zero counts as contact), some exact zeros
    flags = rng.random((T, 4)) * (rng.random((T, 4)) < 0.3)
    contacts = np.zeros((T, 4), dtype=np.int32); phases = np.zeros((T, 4))
    host.qh_fsm_run(2.2, T, mode.ctypes.data, flags.ctypes.data, contacts.ctypes.data, phases.ctypes.data)
    fsm = [PyFSM(i) for i in range(4)]
    for t in range(T):
        if mode[t] == 0:
            for f in fsm:
                f.reset()
            exp_c = [1, 1, 1, 1]; exp_p = [f.phase for f in fsm]
        else:
            exp_p = [f.update(5.0 / 1000.0, 2.2, bool(flags[t, i])) for i, f in enumerate(fsm)]
            exp_c = [f.s for f in fsm]
        assert contacts[t].tolist() == exp_c, t
        assert phases[t].tolist() == exp_p, t          # exact double equality
    # trot: diagonal pairs alternate, never zero stance legs
    walk = contacts[mode == 1]
    assert (walk.sum(1) >= 2).all()
    assert (walk[:, 0] == walk[:, 3]).mean() > 0.95 and (walk[:, 1] == walk[:, 2]).mean() > 0.95


def test_moving_window_filter_exact(host):
    rng = np.random.default_rng(3)
    x = rng.standard_normal(450) * 10.0 ** rng.integers(-6, 6, 450)
    out = np.zeros_like(x)
    host.qh_filter_run(100, len(x), x.ctypes.data, out.ctypes.data)
    # MovingWindowFilter.hpp:26-62
    s = c = 0.0
    dq = deque()
    exp = []

    def neumaier(v):
        nonlocal s, c
        ns = s + v
        c += (s - ns) + v if abs(s) >= abs(v) else (v - ns) + s
        s = ns

    for v in x:
        if len(dq) >= 100:
            neumaier(-dq.popleft())
        neumaier(v); dq.append(v)
        exp.append((s + c) / 100.0)
    assert out.tolist() == exp
    assert abs(out[0] - x[0] / 100.0) < 1e-18      # divides by the window even while filling (:38)


def _feedback(pkg, rec, pos_world=(0.0, 0.0, 0.3), flags=(1, 1, 1, 1)):
    f = np.zeros(38)
    f[0:4] = rec["quat"]; f[4:13] = rec["rot"]; f[13:16] = pos_world
    f[16:19] = rec["rot"].reshape(3, 3) @ rec["lin_vel_body"]     # world-frame velocity
    f[19:22] = rec["ang_vel_body"]; f[22:34] = rec["foot_pos_body"]; f[34:38] = flags
    return f


def test_pack_input_matches_reference_construction(host, pkg):
    """goal_update + foot_update + the record grf_update builds (QuatMpc.cpp:68-176,231-246)."""
    h = host.qh_create(None, 10)
    assert h
    rec = pkg.random_go1_trot_states(1, config_id=2)[0]
    f = _feedback(pkg, rec)
    host.qh_set_feedback(h, f.ctypes.data)
    joy = np.array([0.4, -0.05, 0.28, 0.1, -0.2, 0.3])
    host.qh_set_command(h, joy.ctypes.data, 0.0)
    host.qh_goal_update(h); host.qh_foot_update(h)
    inp = np.zeros(1, dtype=pkg.INPUT_DTYPE)
    host.qh_pack_input(h, inp.ctypes.data)
    r = inp[0]
    R = rec["rot"].reshape(3, 3)
    assert np.array_equal(r["quat"], rec["quat"]) and np.array_equal(r["rot"], rec["rot"])
    assert np.allclose(r["lin_vel_body"], R.T @ (R @ rec["lin_vel_body"]), rtol=0, atol=1e-15)
    assert np.array_equal(r["foot_pos_body"], rec["foot_pos_body"])
    assert r["contacts"].tolist() == [1, 1, 1, 1]                       # movement_mode 0: all stance (:283-289)
    # quat_d <- normalise(quat_d + 0.5 G(quat_d) w_d 5ms), from identity (:128-137)
    w = joy[3:6]
    qd = np.array([1.0, 0, 0, 0]); G = np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0], [0, 0, 1.0]])
    qd = qd + 0.5 * (G @ w) * 5.0 / 1000.0; qd /= np.linalg.norm(qd)
    assert np.allclose(r["quat_d"], qd, rtol=0, atol=1e-16)
    # references: first sample of a 100-window average (:87-89,:103-105)
    yaw = np.arctan2(R[1, 0], R[0, 0]); Rz = np.array([[np.cos(yaw), -np.sin(yaw), 0], [np.sin(yaw), np.cos(yaw), 0], [0, 0, 1]])
    v_world = Rz @ np.array([0.4, -0.05, 0.0])
    assert np.allclose(r["vel_ref_body"], (R.T @ v_world) / 100.0, rtol=0, atol=1e-15)
    pd_world = np.array([0.0 + v_world[0] * 5.0 / 1000.0, 0.0 + v_world[1] * 5.0 / 1000.0, 0.28])
    assert np.allclose(r["pos_ref_body"], (R.T @ (pd_world - np.array([0, 0, 0.3]))) / 100.0, rtol=0, atol=1e-15)
    assert (r["acc_ref_body"] == 0).all()
    host.qh_destroy(h)


def test_walking_mode_contacts_flow_into_the_record(host, pkg):
    h = host.qh_create(None, 10)
    rec = pkg.go1_stand_input()[0]
    f = _feedback(pkg, rec, flags=(0, 0, 0, 0))
    host.qh_set_feedback(h, f.ctypes.data)
    joy = np.array([0.3, 0.0, 0.28, 0, 0, 0])
    host.qh_set_command(h, joy.ctypes.data, 0.0); host.qh_foot_update(h)
    host.qh_set_command(h, joy.ctypes.data, 1.0)
    seen = set()
    inp = np.zeros(1, dtype=pkg.INPUT_DTYPE)
    for _ in range(200):
        host.qh_foot_update(h)
        host.qh_pack_input(h, inp.ctypes.data)
        seen.add(tuple(inp[0]["contacts"].astype(int)))
    assert seen == {(1, 0, 0, 1), (0, 1, 1, 0)}                         # trot pattern :87-108
    host.qh_destroy(h)


def test_grf_update_fails_loudly_without_device(host, pkg):
    import torch

    if torch.cuda.is_available():
        pytest.skip("a GPU is visible here")
    import __graft_entry__ as g

    libpath = str(g.build_hip()).encode()
    h = host.qh_create(libpath, 10)
    assert h
    assert host.qh_device_status(h) == pkg.NO_DEVICE
    rec = pkg.go1_stand_input()[0]
    f = _feedback(pkg, rec)
    host.qh_set_feedback(h, f.ctypes.data)
    assert host.qh_grf_update(h) == 0                                    # no silent CPU fallback
    out = np.zeros(40); host.qh_get_outputs(h, out.ctypes.data)
    assert (out[8:32] == 0).all()
    host.qh_destroy(h)


# ---- ConvexMpcHipT<State> (host mirror of legged::ConvexMpc) -----------------
def _bind_convex(lib):
    vp = C.c_void_p
    lib.qh_convex_create.argtypes = [C.c_char_p, C.c_int]
    lib.qh_convex_create.restype = vp
    lib.qh_convex_device_status.argtypes = [vp]
    lib.qh_convex_set_feedback.argtypes = [vp, vp]
    lib.qh_convex_set_body_xy.argtypes = [vp, C.c_double, C.c_double]
    for f in ("qh_convex_goal_update", "qh_convex_foot_update", "qh_convex_grf_update", "qh_convex_update"):
        getattr(lib, f).argtypes = [vp]
    lib.qh_convex_pack_input.argtypes = [vp, vp]
    lib.qh_convex_get_goal.argtypes = [vp, vp]
    return lib


def _yaw_feedback(yaw=0.0):
    """38-double feedback block of qh_set_feedback for a yaw-only attitude."""
    f = np.zeros(38)
    f[0], f[3] = np.cos(yaw / 2), np.sin(yaw / 2)
    c, s = np.cos(yaw), np.sin(yaw)
    f[4:13] = [c, -s, 0, s, c, 0, 0, 0, 1]
    f[13:16] = [0.1, -0.2, 0.29]
    f[16:19] = [0.3, 0.1, -0.05]
    f[22:34] = [0.2, 0.14, -0.3, 0.2, -0.14, -0.3, -0.2, 0.14, -0.3, -0.2, -0.14, -0.3]
    return f


def test_convex_goal_update_ramps_and_rotates(host):
    host = _bind_convex(host)
    h = host.qh_convex_create(b"", 20)
    assert h and host.qh_convex_device_status(h) == 17           # QMPC_NO_DEVICE: no library given
    yaw = 0.7
    f = _yaw_feedback(yaw)
    host.qh_set_feedback(h, f.ctypes.data)
    joy = np.array([0.02, -0.3, 0.31, 0.0, 0.0, 0.4])            # velx vely body_height roll pitch yaw rates
    host.qh_set_command(h, joy.ctypes.data, 1.0)
    host.qh_convex_set_body_xy(h, 0.5, -0.25)
    vx = 0.0
    out = np.zeros(16)
    for tick in range(8):
        host.qh_convex_goal_update(h)
        # ConvexMpc.cpp:61-65: +-1.0 * h / 1000.0 per tick towards joy.velx (h = 5 ms), never settling exactly
        if vx < joy[0]:
            vx += 1.0 * 5.0 / 1000.0
        elif vx > joy[0]:
            vx -= 1.0 * 5.0 / 1000.0
        host.qh_convex_get_goal(h, out.ctypes.data)
        assert out[0] == vx and out[1] == joy[1] and out[2] == 0.0
        c, s = np.cos(yaw), np.sin(yaw)
        assert np.allclose(out[3:6], [c * vx - s * joy[1], s * vx + c * joy[1], 0.0], rtol=0, atol=1e-15)
        assert list(out[6:9]) == [0.5, -0.25, 0.31] and out[9] == 0.4
    host.qh_destroy(h)


def test_convex_pack_input_and_schedule(host, pkg):
    host = _bind_convex(host)
    h = host.qh_convex_create(b"", 20)
    f = _yaw_feedback(0.3)
    f[34:38] = 0.0
    host.qh_set_feedback(h, f.ctypes.data)
    extra = np.concatenate([[0.01, -0.02, 0.3], [0.1, 0.2, 0.3], np.arange(12) * 0.01])
    host.qh_convex_set_feedback(h, extra.ctypes.data)
    joy = np.array([0.2, 0.05, 0.3, 0.0, 0.0, -0.2])
    host.qh_set_command(h, joy.ctypes.data, 1.0)                 # walking: the FSM runs with dt = h/1000
    fsm = [PyFSM(i) for i in range(4)]
    for tick in range(120):
        host.qh_convex_goal_update(h)
        host.qh_convex_foot_update(h)
        for m in fsm:
            m.update(5.0 / 1000.0, 2.2, False)
        rec = np.zeros(1, dtype=pkg.CONVEX_INPUT_DTYPE)
        host.qh_convex_pack_input(h, rec.ctypes.data)
        assert list(rec["contacts"][0]) == [float(m.s) for m in fsm], tick
    assert list(rec["euler"][0]) == [0.01, -0.02, 0.3]
    assert list(rec["ang_vel_world"][0]) == [0.1, 0.2, 0.3]
    assert list(rec["pos_world"][0]) == list(f[13:16]) and list(rec["lin_vel_world"][0]) == list(f[16:19])
    assert np.array_equal(rec["foot_pos_abs_com"][0], np.arange(12) * 0.01)
    assert rec["yaw_rate_d"][0] == -0.2 and np.all(rec["reserved"][0] == 0.0)
    g = np.zeros(16); host.qh_convex_get_goal(h, g.ctypes.data)
    assert np.array_equal(rec["pos_d_world"][0], g[6:9]) and np.array_equal(rec["lin_vel_d_world"][0], g[3:6])
    assert host.qh_convex_grf_update(h) == 0                     # no device behind it: fails loudly
    host.qh_destroy(h)


# ---- the step before the force path: swing trajectory, Raibert foothold, FSM targets -------
def _py_quintic(t, T, start, fin):
    """Independent restatement of QuinticCurve::get_foot_swing_target (Utils.cpp:236-293): matrix entries in float32
    (products of the float argument T), the 6x6 solve in double, and the polynomial left to right with double
    coefficients (`a_z(2) * t * t` at :263-265 is ((a_z(2) * t) * t) in double: no power of t is formed in float)."""
    f = np.float32
    t, T = f(t), f(T)
    C = np.array([[1, 0, 0, 0, 0, 0],
                  [1, T, T * T, T * T * T, T * T * T * T, T * T * T * T * T],
                  [0, 1, 0, 0, 0, 0],
                  [0, 1, f(2) * T, f(3) * T * T, f(4) * T * T * T, f(5) * T * T * T * T],
                  [1, T / f(2), T * T / f(4), T * T * T / f(8), T * T * T * T / f(16), T * T * T * T * T / f(32)],
                  [0, 1, T, f(3) * T * T / f(4), f(4) * T * T * T / f(8), f(5) * T * T * T * T / f(16)]], dtype=np.float64)
    dx, dy = fin[0] - start[0], fin[1] - start[1]
    k = 1.26 / float(T)
    vm = k * np.sqrt(dx * dx + dy * dy)
    th = np.arctan2(abs(dy), abs(dx))
    vx = (1 if dx >= 0 else -1) * vm * np.cos(th)
    vy = (1 if dy >= 0 else -1) * vm * np.sin(th)
    cons = [[start[0], fin[0], 0, 0, (start[0] + fin[0]) / 2, vx], [start[1], fin[1], 0, 0, (start[1] + fin[1]) / 2, vy],
            [start[2], fin[2], 0.1, -0.1, 0.1, 0.0]]
    out = np.zeros(9)
    td = float(t)
    for ax in range(3):
        a = np.linalg.solve(C, np.array(cons[ax], dtype=float))
        out[ax] = a[0] + a[1] * td + a[2] * td * td + a[3] * td * td * td + a[4] * td * td * td * td + a[5] * td * td * td * td * td
        out[3 + ax] = a[1] + 2 * a[2] * td + 3 * a[3] * td * td + 4 * a[4] * td * td * td + 5 * a[5] * td * td * td * td
        out[6 + ax] = 2 * a[2] + 6 * a[3] * td + 12 * a[4] * td * td + 20 * a[5] * td * td * td
    return out


def test_swing_quintic_matches_restatement_and_boundary_conditions(host):
    host.qh_swing_target.argtypes = [C.c_float, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p]
    rng = np.random.default_rng(9)
    T = 0.5 / 2.2
    for _ in range(50):
        start = rng.uniform(-0.3, 0.3, 3); fin = start + rng.uniform(-0.2, 0.2, 3); fin[2] = start[2] + rng.uniform(-0.02, 0.02)
        for t in (0.0, 0.13 * T, 0.5 * T, 0.9 * T, T):
            out = np.zeros(9)
            host.qh_swing_target(t, T, start.ctypes.data, fin.ctypes.data, out.ctypes.data)
            assert np.abs(out - _py_quintic(t, T, start, fin)).max() < 1e-8
        o0 = np.zeros(9); oT = np.zeros(9); om = np.zeros(9)
        host.qh_swing_target(0.0, T, start.ctypes.data, fin.ctypes.data, o0.ctypes.data)
        host.qh_swing_target(T, T, start.ctypes.data, fin.ctypes.data, oT.ctypes.data)
        host.qh_swing_target(T / 2, T, start.ctypes.data, fin.ctypes.data, om.ctypes.data)
        assert np.abs(o0[:3] - start).max() < 1e-12 and np.abs(o0[3:6] - [0, 0, 0.1]).max() < 1e-12
        # the float32 matrix entries / powers of t limit these to ~1e-5 (the reference's own rounding)
        assert np.abs(oT[:3] - fin).max() < 1e-5 and np.abs(oT[3:6] - [0, 0, -0.1]).max() < 1e-3
        assert abs(om[2] - 0.1) < 1e-5 and abs(om[5]) < 1e-3                     # apex: ABSOLUTE height 0.1, Utils.cpp:259


def test_fsm_full_update_publishes_foot_targets(host):
    host.qh_fsm_leg_run.argtypes = [C.c_int, C.c_double, C.c_double, C.c_int] + [C.c_void_p] * 6
    ticks, freq, dt = 400, 2.2, 0.005
    rng = np.random.default_rng(10)
    cur = np.cumsum(rng.normal(0, 1e-3, (ticks, 3)), axis=0) + [0.2, 0.14, 0.0]
    tgt = cur + [0.05, 0.0, 0.0]
    flags = (rng.random(ticks) < 0.2).astype(float)
    for leg in range(4):
        contacts = np.zeros(ticks, dtype=np.int32); phases = np.zeros(ticks); targets = np.zeros((ticks, 9))
        host.qh_fsm_leg_run(leg, freq, dt, ticks, cur.ctypes.data, tgt.ctypes.data, flags.ctypes.data,
                            contacts.ctypes.data, phases.ctypes.data, targets.ctypes.data)
        # the schedule is the one of the schedule-only update (bit-exact restatement above)
        m = PyFSM(leg); m.reset()
        start = cur[0].copy(); pos = tgt[0].copy(); vel = np.zeros(3); acc = np.zeros(3)
        for t in range(ticks):
            prev = m.s
            m.update(dt, freq, bool(flags[t]))
            assert contacts[t] == m.s and phases[t] == m.phase
            if prev == PyFSM.STANCE and m.s == PyFSM.SWING:
                start = cur[t].copy()                                   # swing_enter
            if prev == PyFSM.SWING and m.s == PyFSM.STANCE:
                pos, vel = cur[t].copy(), np.zeros(3)                   # stance_enter
            if m.s == PyFSM.SWING:
                o = _py_quintic(0.5 * m._pct() / freq, 0.5 / freq, start, tgt[t])
                pos, vel, acc = o[:3], o[3:6], o[6:]
            assert np.abs(targets[t, :3] - pos).max() < 1e-8, (leg, t)
            assert np.abs(targets[t, 3:6] - vel).max() < 1e-7 and np.abs(targets[t, 6:] - acc).max() < 1e-5


class PyLegFull(PyFSM):
    """The full per-leg state machine (schedule + foot targets), LeggedContactFSM.cpp:11-86,225-246, restated
    independently of host/LeggedContactFSMHip.h: reset() sends a swinging foot to its saved target and clears
    not_first_call, so the first walking tick after a stand re-seeds swing start / end and the targets."""

    def __init__(self, leg):
        super().__init__(leg)
        self.first = False                        # not_first_call
        self.start = np.zeros(3); self.end = np.zeros(3)
        self.pos = np.zeros(3); self.vel = np.zeros(3); self.acc = np.zeros(3)

    def reset(self):
        was_swing = self.s == self.SWING
        super().reset()
        if was_swing:                             # :20-25
            self.pos = self.end.copy(); self.vel = np.zeros(3)
        self.first = False                        # :30

    def step(self, dt, freq, cur, tgt, flag):
        if not self.first:                        # :37-43
            self.start, self.end = cur.copy(), tgt.copy()
            self.pos, self.vel = tgt.copy(), np.zeros(3)
            self.first = True
        prev = self.s
        self.update(dt, freq, flag)
        if prev == self.STANCE and self.s == self.SWING:
            self.start = cur.copy()               # swing_enter :225-229
        if prev == self.SWING and self.s == self.STANCE:
            self.pos, self.vel = cur.copy(), np.zeros(3)   # stance_enter :231-235
        if self.s == self.SWING:                  # swing_update :237-246
            o = _py_quintic(0.5 * self._pct() / freq, 0.5 / freq, self.start, tgt)
            self.pos, self.vel, self.acc = o[:3], o[3:6], o[6:]


def _fake_harness(host):
    host.qh_create_fake.argtypes = [C.c_int]; host.qh_create_fake.restype = C.c_void_p
    host.qh_fake_script.argtypes = [C.c_void_p, C.c_int, C.c_int]
    host.qh_get_foot_targets.argtypes = [C.c_void_p, C.c_void_p]
    host.qh_set_foot_world.argtypes = [C.c_void_p] * 3
    return host.qh_create_fake(10)


def test_update_publishes_fsm_foot_targets_tick_for_tick(host, pkg):
    """QuatMpc.cpp:270-272: optimized_state[6+3i], optimized_input[12+3i] and [24+3i] carry the FSM foot position /
    velocity / acceleration targets that BaseInterface.cpp:349,358 servo the legs to.  Stand -> walk -> stand -> walk,
    so the reset semantics (:11-31) are covered, through the whole update() of the drop-in class."""
    h = _fake_harness(host)
    assert h
    f12 = np.arange(12, dtype=float) + 1.0
    host.qh_fake_script(f12.ctypes.data, pkg.OK, pkg.OK)
    rec = pkg.go1_stand_input()[0]
    rng = np.random.default_rng(21)
    T, freq, dt = 700, 2.2, 5.0 / 1000.0          # param.gait_freq default, the reference's hard-wired 5 ms
    mode = np.ones(T); mode[:5] = 0; mode[330:340] = 0
    base = np.array([[0.2, 0.14, 0.02], [0.2, -0.14, 0.02], [-0.2, 0.14, 0.02], [-0.2, -0.14, 0.02]])
    legs = [PyLegFull(i) for i in range(4)]
    joy = np.array([0.3, 0.0, 0.28, 0, 0, 0])
    saw_swing_at_reset = False
    for t in range(T):
        cur = base + np.cumsum(rng.normal(0, 1e-3, (4, 3)), axis=0) + [0.001 * t, 0, 0]
        tgt = cur + [0.06, 0.01, 0.0]
        flags = (rng.random(4) < 0.15).astype(float)
        fb = _feedback(pkg, rec, flags=flags)
        host.qh_set_feedback(h, fb.ctypes.data)
        host.qh_set_foot_world(h, np.ascontiguousarray(cur).ctypes.data, np.ascontiguousarray(tgt).ctypes.data)
        host.qh_set_command(h, joy.ctypes.data, float(mode[t]))
        assert host.qh_update(h) == 1
        out = np.zeros(72); host.qh_get_foot_targets(h, out.ctypes.data)
        o40 = np.zeros(40); host.qh_get_outputs(h, o40.ctypes.data)
        # the three published blocks ARE the FSM members, bit for bit
        assert np.array_equal(out[0:12], out[36:48]) and np.array_equal(out[12:24], out[48:60])
        assert np.array_equal(out[24:36], out[60:72])
        assert np.array_equal(o40[8:20], f12)                          # optimized_input[0:12] = u
        for i, m in enumerate(legs):
            if mode[t] == 0:
                saw_swing_at_reset |= (m.s == PyFSM.SWING and t > 10)
                m.reset()
            else:
                m.step(dt, freq, cur[i], tgt[i], bool(flags[i]))
                assert int(o40[i]) == m.s
            assert np.abs(out[36 + 3 * i:39 + 3 * i] - m.pos).max() < 1e-8, (t, i)
            assert np.abs(out[48 + 3 * i:51 + 3 * i] - m.vel).max() < 1e-7, (t, i)
            assert np.abs(out[60 + 3 * i:63 + 3 * i] - m.acc).max() < 1e-5, (t, i)
    assert saw_swing_at_reset                                           # the :20-25 branch was exercised
    assert np.abs(out[0:12]).max() > 0.1 and np.abs(out[12:24]).max() > 0
    host.qh_destroy(h)


def test_grf_update_keeps_previous_forces_on_instance_failure(host, pkg):
    """Per-instance status words (qmpc.h): zero-force answers (NAN_INPUT, NO_CONTACT) and broken iterates (NOT_PD)
    must not overwrite the forces of the previous tick; MAX_ITER is an iterate like the reference's own."""
    h = _fake_harness(host)
    rec = pkg.go1_stand_input()[0]
    fb = _feedback(pkg, rec); host.qh_set_feedback(h, fb.ctypes.data)
    good = np.linspace(1.0, 12.0, 12)
    host.qh_fake_script(good.ctypes.data, pkg.OK, pkg.OK)
    assert host.qh_grf_update(h) == 1
    o = np.zeros(40); host.qh_get_outputs(h, o.ctypes.data)
    assert np.array_equal(o[8:20], good) and np.array_equal(o[20:32], good)      # R = I
    bad = np.full(12, 777.0)
    for st in (pkg.NAN_INPUT, pkg.NO_CONTACT, pkg.NOT_PD, pkg.LINESEARCH_FAIL):
        host.qh_fake_script(bad.ctypes.data, pkg.OK, st)
        assert host.qh_grf_update(h) == 0
        host.qh_get_outputs(h, o.ctypes.data)
        assert np.array_equal(o[8:20], good) and np.array_equal(o[20:32], good)
    host.qh_fake_script(bad.ctypes.data, pkg.HIP_ERROR, pkg.OK)                     # call-level failure
    assert host.qh_grf_update(h) == 0
    host.qh_get_outputs(h, o.ctypes.data); assert np.array_equal(o[8:20], good)
    host.qh_fake_script(bad.ctypes.data, pkg.OK, pkg.MAX_ITER)
    assert host.qh_grf_update(h) == 1
    host.qh_get_outputs(h, o.ctypes.data); assert np.array_equal(o[8:20], bad)
    host.qh_destroy(h)


def test_raibert_foot_targets(host):
    host.qh_raibert.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    h = host.qh_create(None, 10)
    yaw = -0.4
    f = _yaw_feedback(yaw)
    host.qh_set_feedback(h, f.ctypes.data)
    vd = np.array([0.4, -0.1, 0.0])
    out = np.zeros(36)
    host.qh_raibert(h, vd.ctypes.data, out.ctypes.data)
    c, s = np.cos(yaw), np.sin(yaw)
    Rz = np.array([[c, -s, 0], [s, c, 0], [0, 0, 1.0]])
    vrel = Rz.T @ f[16:19]
    k = np.sqrt(abs(f[15]) / 9.81)
    d = np.array([k * (vrel[0] - vd[0]) + (1 / 2.2) / 2 * vd[0], k * (vrel[1] - vd[1]) + (1 / 2.2) / 2 * vd[1], 0.0])
    d[0] = np.clip(d[0], -0.5, 0.5); d[1] = np.clip(d[1], -0.3, 0.3)
    feet = np.array([[0.20, 0.14, -0.3], [0.20, -0.14, -0.3], [-0.20, 0.14, -0.3], [-0.20, -0.14, -0.3]])
    a = (Rz @ feet.T).T + [*(Rz @ d)[:2], 0.0]
    assert np.abs(out[:12].reshape(4, 3) - a).max() < 1e-14
    assert np.abs(out[12:24].reshape(4, 3) - a @ Rz).max() < 1e-14          # torso_rot_mat' * abs (yaw-only attitude)
    assert np.abs(out[24:].reshape(4, 3) - (a + f[13:16])).max() < 1e-14
    # the clamp (FOOT_DELTA_X/Y_LIMIT, LeggedParams.h:21-22)
    vd2 = np.array([-20.0, 30.0, 0.0])
    host.qh_raibert(h, vd2.ctypes.data, out.ctypes.data)
    d2 = np.array([k * (vrel[0] - vd2[0]) + (1 / 2.2) / 2 * vd2[0], k * (vrel[1] - vd2[1]) + (1 / 2.2) / 2 * vd2[1], 0.0])
    assert abs(d2[0]) > 0.5 and abs(d2[1]) > 0.3                            # both limits are hit
    d2[0] = np.clip(d2[0], -0.5, 0.5); d2[1] = np.clip(d2[1], -0.3, 0.3)
    a2 = (Rz @ feet.T).T + [*(Rz @ d2)[:2], 0.0]
    assert np.abs(out[:12].reshape(4, 3) - a2).max() < 1e-14
    host.qh_destroy(h)


def test_debug_topic_records(host):
    """/debug/{torso_odom,torso_odom_d,mpc_grf,mpc_time} payloads (LeggedLogger.hpp:48-106) from the harness
    state: field-for-field the numbers the reference's logger publishes."""
    host.qh_debug_records.argtypes = [C.c_void_p, C.c_void_p, C.c_char_p]
    host.qh_set_mpc_outputs.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_double]
    host.qh_debug_grf_batch.argtypes = [C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    h = host.qh_create(None, 10)
    f = _yaw_feedback(0.3)
    f[19:22] = [0.01, -0.02, 0.4]
    host.qh_set_feedback(h, f.ctypes.data)
    joy = np.array([0.3, -0.1, 0.28, 0.0, 0.0, 0.2])
    host.qh_set_command(h, joy.ctypes.data, 1.0)
    host.qh_goal_update(h)
    grf = np.array([1.0, 2.0, 2.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, -3.0, 4.0, 12.0])
    con = np.array([1.0, 0.0, 0.0, 1.0])
    host.qh_set_mpc_outputs(h, grf.ctypes.data, con.ctypes.data, 0.734)
    out = np.zeros(39)
    names = C.create_string_buffer(12)
    host.qh_debug_records(h, out.ctypes.data, names)
    assert [names.raw[3 * i:3 * i + 2].decode() for i in range(4)] == ["FL", "FR", "RL", "RR"]
    # torso_odom: world position, attitude (w x y z), BODY-frame linear velocity, body angular velocity
    assert np.array_equal(out[0:3], f[13:16]) and np.array_equal(out[3:7], f[0:4]) and np.array_equal(out[10:13], f[19:22])
    assert np.array_equal(out[7:10], np.zeros(3))      # fbk.torso_lin_vel_body is written by grf_update (QuatMpc.cpp:231)
    # torso_odom_d: the desired quantities goal_update produced (same numbers as qh_get_outputs / pack_input)
    o = np.zeros(40)
    host.qh_get_outputs(h, o.ctypes.data)
    assert np.array_equal(out[16:20], o[32:36])
    assert abs(out[20] - 0.3) < 0.31 and out[25] == pytest.approx(0.2)    # filtered vx command, yaw rate
    # mpc_grf: planned contacts, zero velocities, force norms; mpc_time
    assert np.array_equal(out[26:30], con) and np.array_equal(out[30:34], np.zeros(4))
    assert np.allclose(out[34:38], [3.0, 0.0, 0.0, 13.0], rtol=0, atol=1e-15) and out[38] == 0.734
    # batch form
    rng = np.random.default_rng(5)
    F = rng.normal(size=(64, 12)); Cn = (rng.random((64, 4)) < 0.5).astype(np.float64)
    pos = np.zeros((64, 4)); eff = np.zeros((64, 4))
    host.qh_debug_grf_batch(64, F.ctypes.data, Cn.ctypes.data, pos.ctypes.data, eff.ctypes.data)
    assert np.array_equal(pos, Cn) and np.abs(eff - np.linalg.norm(F.reshape(64, 4, 3), axis=2)).max() < 1e-15
    host.qh_destroy(h)


def test_c_example_builds_and_fails_loudly_without_a_gpu(tmp_path):
    """examples/solve_batch.c: the C ABI from plain C.  It compiles and links against the in-tree library; without
    a GPU the product path refuses to run (QMPC_NO_DEVICE) instead of falling back to anything."""
    import shutil
    import subprocess

    import __graft_entry__ as g

    repo = Path(__file__).resolve().parents[1]
    lib = g.build_hip()
    exe = tmp_path / "solve_batch"
    subprocess.run(["gcc", "-O2", "-Wall", "-Werror", "-I", str(repo / "include"), str(repo / "examples" / "solve_batch.c"),
                    "-o", str(exe), str(lib), f"-Wl,-rpath,{lib.parent}", "-lm"], check=True)
    try:
        import torch
        has_gpu = torch.cuda.is_available()
    except Exception:
        has_gpu = False
    r = subprocess.run([str(exe), "4"], capture_output=True, text=True)
    if has_gpu:
        assert r.returncode == 0, r.stdout + r.stderr
    else:
        assert r.returncode == 2 and "no CPU fallback" in r.stderr
    # the closed-loop example too
    exe2 = tmp_path / "closed_loop"
    subprocess.run(["gcc", "-O2", "-Wall", "-Werror", "-I", str(repo / "include"), str(repo / "examples" / "closed_loop.c"),
                    "-o", str(exe2), str(lib), f"-Wl,-rpath,{lib.parent}", "-lm"], check=True)
    r = subprocess.run([str(exe2), "4", "10"], capture_output=True, text=True)
    if has_gpu:
        assert r.returncode == 0, r.stdout + r.stderr
    else:
        assert r.returncode == 2 and "no CPU fallback" in r.stderr
    shutil.rmtree(tmp_path, ignore_errors=True)


def test_host_closed_loop_stands_in_equilibrium_with_scripted_forces(host, pkg):
    """host/ClosedLoopHost.h with the scripted test double: feet under the hips, a quarter of the weight on every
    foot -> the plant (csrc/qmpc_loop_math.h) stays at rest; the same forces with one leg unloaded tip it over."""
    lib = pkg.load_library()
    vp = C.c_void_p
    host.qh_loop_create.argtypes = [C.c_char_p, C.c_int, vp, vp]; host.qh_loop_create.restype = vp
    for f in ("qh_loop_tick", "qh_loop_destroy"):
        getattr(host, f).argtypes = [vp]
    host.qh_loop_export.argtypes = [vp, vp]
    host.qh_fake_script.argtypes = [vp, C.c_int, C.c_int]
    lp = pkg.default_loop_params(lib)
    st0 = pkg.loop_states([[0, 0, 0.3, 0, 0, 0, 0]], lp, lib=lib)
    assert pkg.LOOP_STATE_DTYPE.itemsize == lib.qmpc_sizeof_loop_state() == 820 * 8
    w = 12.84 * 9.81 / 4
    f = np.array([0, 0, w] * 4)
    host.qh_fake_script(f.ctypes.data, pkg.OK, pkg.OK)
    h = host.qh_loop_create(None, 10, C.addressof(lp), st0.ctypes.data)
    e = np.zeros(1, dtype=pkg.LOOP_STATE_DTYPE)
    for _ in range(100):
        assert host.qh_loop_tick(h) == 1
    host.qh_loop_export(h, e.ctypes.data)
    assert np.abs(e[0]["pos_world"] - [0, 0, 0.3]).max() < 1e-12 and np.abs(e[0]["quat"] - [1, 0, 0, 0]).max() < 1e-12
    assert e[0]["tick"] == 100 and (e[0]["contacts"] == 1).all() and np.array_equal(e[0]["forces_body"], f)
    host.qh_loop_destroy(h)
    f2 = f.copy(); f2[2] = 0.0                     # front-left leg unloaded
    host.qh_fake_script(f2.ctypes.data, pkg.OK, pkg.OK)
    h = host.qh_loop_create(None, 10, C.addressof(lp), st0.ctypes.data)
    for _ in range(40):
        host.qh_loop_tick(h)
    host.qh_loop_export(h, e.ctypes.data)
    assert e[0]["pos_world"][2] < 0.3 - 1e-3 and abs(e[0]["quat"][1]) > 1e-3 and abs(e[0]["quat"][2]) > 1e-3
    host.qh_loop_destroy(h)
