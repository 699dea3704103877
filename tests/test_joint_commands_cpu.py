"""Joint-level command of a tick (BaseInterface::tau_ctrl_update, BaseInterface.cpp:343-408) on the CPU side:
the oracle's restatement (oracle/qo_legkin.c) against kinematic identities and the reference's own test point, and
the host mirror (host/JointCommandsHip.h, the arithmetic the device kernels share) against the oracle."""
import ctypes as C

import numpy as np
import pytest

from conftest import load_pkg

pkg = load_pkg()
from oracle import pyoracle as po  # noqa: E402


def _host():
    import __graft_entry__ as g
    h = C.CDLL(str(g.build_host()))
    vp = C.c_void_p
    h.qh_leg_inverse.argtypes = [C.c_int, vp, vp, vp]
    h.qh_joint_commands.argtypes = [C.c_int, vp, vp]
    return h


def random_joint_angles(rng, B):
    """The joint ranges of the reference's commented sweep (TestInvKin.cpp:37-44): hip +-46 deg, thigh -60..240 deg,
    calf -154.5..-52.5 deg."""
    return np.stack([rng.uniform(-0.8, 0.8, (B, 4)), rng.uniform(-1.0, 4.1, (B, 4)), rng.uniform(-2.69, -0.92, (B, 4))],
                    axis=-1).reshape(B, 12)


def random_feedback(rng, B, walking=None):
    """Feedback records around trotting postures: joint angles near the stand pose, foot targets near the feet."""
    g = po.default_go1_geometry()
    fb = np.zeros(B, dtype=pkg.JOINT_FEEDBACK_DTYPE)
    q = np.tile([0.0, 0.67, -1.3], (B, 4)) + rng.uniform(-0.35, 0.35, (B, 12))
    fb["joint_pos"] = q
    fb["joint_vel"] = rng.uniform(-2, 2, (B, 12))
    fb["torso_pos_world"] = rng.uniform(-1, 1, (B, 3)) + [0, 0, 0.3]
    quat = rng.normal(size=(B, 4)) * [1, 0.15, 0.15, 1]
    fb["torso_quat"] = quat / np.linalg.norm(quat, axis=1, keepdims=True)
    fb["torso_lin_vel_world"] = rng.uniform(-0.5, 0.5, (B, 3))
    p_body, _ = po.leg_kinematics(g, q)
    R = pkg.quat_to_rot(fb["torso_quat"]).reshape(-1, 3, 3)
    tgt_body = p_body.reshape(B, 4, 3) + rng.uniform(-0.04, 0.04, (B, 4, 3))
    fb["foot_pos_target_world"] = (np.einsum("bij,blj->bli", R, tgt_body) + fb["torso_pos_world"][:, None, :]).reshape(B, 12)
    fb["foot_vel_target_world"] = rng.uniform(-1, 1, (B, 12))
    fb["forces_body"] = rng.uniform(-30, 80, (B, 12))
    fb["plan_contacts"] = (rng.random((B, 4)) < 0.6).astype(float)
    fb["movement_mode"] = (rng.random(B) < 0.8).astype(float) if walking is None else float(walking)
    return fb


def test_inverse_kinematics_inverts_the_forward_kinematics_over_the_joint_range():
    g = po.default_go1_geometry()
    rng = np.random.default_rng(0)
    q = random_joint_angles(rng, 20000)
    p, _ = po.leg_kinematics(g, q)
    cur = q.copy()
    cur[:, 1::3] = 0.0      # only the hip angle of cur_q is read (A1Kinematics.cpp:354,394)
    cur[:, 2::3] = 0.0
    qi = po.leg_inverse_kinematics(g, p, cur)
    assert not np.isnan(qi).any()
    # the reference's own (commented) acceptance is 1e-3 rad (TestInvKin.cpp:59); the single-precision atan2
    # polynomial is good to a few 1e-6
    assert np.abs(qi - q).max() < 1e-5
    # the mirrored hip solution is taken when the current hip angle is nearer to it: still a solution of the FK
    far = cur.copy()
    far[:, 0::3] += np.where(rng.random((len(q), 4)) < 0.5, 2.5, -2.5)
    qm = po.leg_inverse_kinematics(g, p, far)
    differs = np.abs(qm - qi).reshape(-1, 4, 3)[:, :, 0] > 1e-3
    assert differs.any()


def test_reference_test_point_is_out_of_reach():
    """TestInvKin.cpp:15-33: FR leg, foot (0.23391, -0.364016, -0.294254), cur_q (-0.224375, 0.466764, -1.39401).
    The point is 0.435 m from the hip with 0.426 m of leg: thigh and calf come out NaN (the case the caller's isnan
    fallback, BaseInterface.cpp:351-353, exists for); the hip angle is finite."""
    g = po.default_go1_geometry()
    p = np.tile([0.2, 0.13, -0.3, 0.2, -0.13, -0.3, -0.2, 0.13, -0.3, -0.2, -0.13, -0.3], (1, 1)).astype(float)
    c = np.zeros((1, 12))
    p[0, 3:6] = (0.23391, -0.364016, -0.294254)
    c[0, 3:6] = (-0.224375, 0.466764, -1.39401)
    q = po.leg_inverse_kinematics(g, p, c).reshape(4, 3)
    assert np.isfinite(q[1, 0]) and np.isnan(q[1, 1]) and np.isnan(q[1, 2])
    assert np.isfinite(q[[0, 2, 3]]).all()
    fb = np.zeros(1, dtype=pkg.JOINT_FEEDBACK_DTYPE)
    fb["joint_pos"] = c + np.tile([0.0, 0.67, -1.3], 4) * (np.arange(12) // 3 != 1)
    fb["joint_vel"] = 0.25
    fb["torso_quat"][0, 0] = 1.0
    fb["foot_pos_target_world"] = p
    fb["plan_contacts"] = 1.0
    fb["movement_mode"] = 1.0
    cmd = po.joint_commands(g, fb)
    np.testing.assert_array_equal(cmd["joint_ang_tgt"][0, 3:6], fb["joint_pos"][0, 3:6])    # the fallback


def test_joint_commands_properties():
    g = po.default_go1_geometry()
    rng = np.random.default_rng(3)
    fb = random_feedback(rng, 4000)
    cmd = po.joint_commands(g, fb)
    walking = fb["movement_mode"] > 0
    stance = fb["plan_contacts"] != 0
    # torques: -J'f for planned stance legs (and for every leg when standing), exactly 0 for swing legs
    tau = po.torque_map(g, fb["joint_pos"], fb["forces_body"], fb["plan_contacts"], walking=True)
    tau_all = po.torque_map(g, fb["joint_pos"], fb["forces_body"], None, walking=False)
    np.testing.assert_array_equal(cmd["joint_tau_tgt"][walking], tau[walking])
    np.testing.assert_array_equal(cmd["joint_tau_tgt"][~walking], tau_all[~walking])
    assert (cmd["joint_tau_tgt"].reshape(-1, 4, 3)[walking][~stance[walking]] == 0).all()
    # standing: the targets are the measurements
    np.testing.assert_array_equal(cmd["joint_ang_tgt"][~walking], fb["joint_pos"][~walking])
    np.testing.assert_array_equal(cmd["joint_vel_tgt"][~walking], fb["joint_vel"][~walking])
    # walking: FK of the angle target is the foot target in the body frame; J qd is the relative foot velocity
    w = np.where(walking)[0]
    R = pkg.quat_to_rot(fb["torso_quat"][w]).reshape(-1, 3, 3)
    tgt_body = np.einsum("bji,blj->bli", R, fb["foot_pos_target_world"][w].reshape(-1, 4, 3) - fb["torso_pos_world"][w][:, None, :])
    p_tgt, _ = po.leg_kinematics(g, cmd["joint_ang_tgt"][w])
    reach = (cmd["joint_ang_tgt"][w] != fb["joint_pos"][w]).reshape(-1, 4, 3).any(axis=2)   # else: the isnan fallback
    assert reach.mean() > 0.95
    err = np.abs(p_tgt.reshape(-1, 4, 3) - tgt_body)[reach].max(axis=1)
    # a few 1e-6 m from the single-precision atan2; the clamp near the straight leg (A1Kinematics.cpp:423-426) costs more
    assert np.quantile(err, 0.99) < 5e-6 and err.max() < 1e-2
    _, J = po.leg_kinematics(g, fb["joint_pos"][w])
    Jm = J.reshape(-1, 4, 3, 3).transpose(0, 1, 3, 2)          # column-major 3x3 -> [row, col]
    v_body = np.einsum("bji,blj->bli", R, fb["foot_vel_target_world"][w].reshape(-1, 4, 3) - fb["torso_lin_vel_world"][w][:, None, :])
    jq = np.einsum("blij,blj->bli", Jm, cmd["joint_vel_tgt"][w].reshape(-1, 4, 3))
    assert np.abs(jq - v_body).max() < 1e-9


def test_host_mirror_matches_oracle():
    """host/JointCommandsHip.h + csrc/qmpc_joint_math.h (compiled for the host) against oracle/qo_legkin.c: two
    restatements of the same expressions, the same libm -> equal to rounding of sums in a different order."""
    host = _host()
    g = po.default_go1_geometry()
    rng = np.random.default_rng(5)
    q = random_joint_angles(rng, 5000)
    p, _ = po.leg_kinematics(g, q)
    cur = q + rng.uniform(-0.2, 0.2, q.shape)
    qo = po.leg_inverse_kinematics(g, p, cur)
    qh = np.zeros_like(qo)
    host.qh_leg_inverse(len(p), p.ctypes.data, cur.ctypes.data, qh.ctypes.data)
    np.testing.assert_array_equal(qh, qo)
    fb = random_feedback(rng, 5000)
    co = po.joint_commands(g, fb)
    ch = np.zeros(len(fb), dtype=pkg.JOINT_COMMAND_DTYPE)
    host.qh_joint_commands(len(fb), fb.ctypes.data, ch.ctypes.data)
    for k in ("joint_ang_tgt", "joint_vel_tgt", "joint_tau_tgt"):
        np.testing.assert_allclose(ch[k], co[k], rtol=0, atol=1e-10, err_msg=k)
    np.testing.assert_array_equal(ch["joint_tau_tgt"], co["joint_tau_tgt"])
