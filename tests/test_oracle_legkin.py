"""CPU checks of the leg kinematics / force -> joint-torque restatement (oracle/qo_legkin.c;
SURVEY.md 8f rank 2: A1Kinematics::fk/jac, BaseInterface::tau_ctrl_update).

Pins: the stand-pose foothold the reference's Gazebo interface starts from
(q = (0, 0.67, -1.3) -> (+-0.1813, +-0.12795, -0.339), GazeboInterface.cpp:86-89 /
SURVEY.md 8d), the Jacobian as the derivative of the forward kinematics, virtual work
(tau . dq = -f . dp), and the stance / swing rule of tau_ctrl_update.
"""
import numpy as np


def test_go1_geometry(oracle):
    g = oracle.default_go1_geometry()
    fix = np.array([list(r) for r in g.rho_fix])
    assert np.array_equal(fix[:, 0], [0.1881, 0.1881, -0.1881, -0.1881])        # BaseInterface.cpp:12-15
    assert np.array_equal(fix[:, 1], [0.04675, -0.04675, 0.04675, -0.04675])    # :16-19
    assert np.array_equal(fix[:, 2], [0.0812, -0.0812, 0.0812, -0.0812])        # :20-23
    assert np.all(fix[:, 3:] == 0.213)                                          # LeggedParams.h:14-15
    assert np.all(np.array([list(r) for r in g.rho_opt]) == 0.0)                # BaseInterface.cpp:31


def test_stand_pose_foothold(oracle):
    g = oracle.default_go1_geometry()
    q = np.tile([0.0, 0.67, -1.3], 4)
    p, _ = oracle.leg_kinematics(g, q)
    p = p.reshape(4, 3)
    assert np.allclose(np.abs(p[:, 0] - [0.1881, 0.1881, -0.1881, -0.1881]), 0.1881 - 0.18131782, atol=1e-8)
    assert np.allclose(p[:, 1], [0.12795, -0.12795, 0.12795, -0.12795], atol=1e-15)
    assert np.allclose(p[:, 2], -0.33906387, atol=1e-8)
    # straight leg: the foot hangs l_thigh + l_calf below the hip joint
    p0, _ = oracle.leg_kinematics(g, np.zeros(12))
    assert np.allclose(p0.reshape(4, 3)[:, 2], -0.426, atol=1e-15)


def test_jacobian_is_the_derivative_of_fk(oracle):
    g = oracle.default_go1_geometry()
    for l in range(4):
        g.rho_opt[l][0], g.rho_opt[l][1], g.rho_opt[l][2] = 0.01 * (l + 1), -0.005 * l, 0.02   # exercise c != 0
    rng = np.random.default_rng(3)
    q = rng.uniform(-1.5, 1.5, (64, 12))
    _, J = oracle.leg_kinematics(g, q)
    eps = 1e-6
    for j in range(3):
        d = np.zeros(12); d[j::3] = eps
        pp, _ = oracle.leg_kinematics(g, q + d)
        pm, _ = oracle.leg_kinematics(g, q - d)
        col = ((pp - pm) / (2 * eps)).reshape(64, 4, 3)
        assert np.abs(J[:, :, 3 * j:3 * j + 3] - col).max() < 1e-9                  # column-major: J[3j+i]


def test_torque_map_rule_and_virtual_work(oracle):
    g = oracle.default_go1_geometry()
    rng = np.random.default_rng(4)
    q = rng.uniform(-1.2, 1.2, (32, 12))
    f = rng.normal(0, 40, (32, 12))
    c = (rng.random((32, 4)) < 0.5).astype(float)
    _, J = oracle.leg_kinematics(g, q)
    Jm = J.reshape(32, 4, 3, 3).transpose(0, 1, 3, 2)                               # [.., i, j] = dp_i/dq_j
    tau_all = -np.einsum("blij,bli->blj", Jm, f.reshape(32, 4, 3)).reshape(32, 12)  # -J' f
    assert np.abs(oracle.torque_map(g, q, f, None, walking=True) - tau_all).max() < 1e-12
    assert np.abs(oracle.torque_map(g, q, f, c, walking=False) - tau_all).max() < 1e-12   # standing: every leg
    tw = oracle.torque_map(g, q, f, c, walking=True)
    mask = np.repeat(c, 3, axis=1)
    assert np.abs(tw - np.where(mask != 0, tau_all, 0.0)).max() < 1e-12
    assert np.all(tw[mask == 0] == 0.0)                                             # swing legs: exactly zero
    # virtual work: tau . dq = -f . (p(q+dq) - p(q)) to first order
    dq = 1e-6 * rng.normal(size=(32, 12))
    p0, _ = oracle.leg_kinematics(g, q)
    p1, _ = oracle.leg_kinematics(g, q + dq)
    assert np.abs((tau_all * dq).sum(1) + (f * (p1 - p0)).sum(1)).max() < 1e-8
