"""CPU suite: pins the oracle (oracle/) on the reference's own golden vectors.

Golden data (tests/golden/*.json) are the reference's committed test OUTPUTS:
  legged_ctrl/src/test/test_altro/quat_mpc_test.json       (TestAltroQuatMpc.cpp)
  legged_ctrl/src/test/test_altro/trot_quat_mpc_test.json  (TestAltroTrotQuatMpc.cpp)
Known-answer numbers quoted inline come from TestDoubleIntegrator.cpp / TestPendulum.cpp.
"""
import ctypes as C
import json
from pathlib import Path

import numpy as np
import pytest

from conftest import golden_problem

GOLDEN = Path(__file__).parent / "golden"


def _golden(name):
    d = json.loads((GOLDEN / name).read_text())
    return np.array(d["state_trajectory"]), np.array(d["input_trajectory"]), np.array(d["reference_state"])


# ---- generic AL-iLQR scheme: reference KATs ---------------------------------
def test_kat_double_integrator_goal_constraint(oracle):
    # TestDoubleIntegrator.cpp:170-256: dist_to_goal < 1e-4, GetIterations() == 3
    it, status, dist, *_ = oracle.kat_double_integrator(0)
    assert status == 0 and it == 3 and dist < 1e-4


def test_kat_double_integrator_control_bounds(oracle):
    # TestDoubleIntegrator.cpp:258-375: u0 == -1 +- 1e-4, GetIterations() == 5
    it, status, dist, u0, u1, *_ = oracle.kat_double_integrator(1)
    assert status == 0 and it == 5 and dist < 1e-4
    assert abs(u0 + 1.0) < 1e-4 and abs(u1 + 1.0) < 1e-4


def test_kat_pendulum_midpoint(oracle):
    # TestPendulum.cpp:13-43
    xn, J = oracle.kat_pendulum_midpoint()
    assert np.linalg.norm(xn - [0.08445158545673655, -0.21395149094594346]) < 1e-6
    Jexp = np.array([[0.9755975228465564, 0.0495, 0.005000000000000001],
                     [-0.967268640223389, 0.9557742592228808, 0.198]])
    assert np.linalg.norm(J - Jexp) < 1e-6


def test_kat_pendulum_swingup(oracle):
    # TestPendulum.cpp:45-115: xN_expected to 1e-5, <= 10 iterations
    it, status, x0, x1, *_ = oracle.kat_pendulum_swingup()
    assert status == 0 and it <= 10
    assert np.hypot(x0 - 3.12099917161669, x1 - 0.0011966258762942175) < 1e-5


# ---- model: dynamics + midpoint + float h pinned by the goldens ---------------
@pytest.mark.parametrize("which,name", [("stand", "quat_mpc_test.json"), ("trot", "trot_quat_mpc_test.json")])
def test_golden_rollout_reproduces_states(oracle, pkg, which, name):
    """Rolling the golden inputs through the restated dynamics reproduces the
    golden states (1.4e-12 / 3.7e-15 with h = 0.01f)."""
    Xg, Ug, _ = _golden(name)
    p, rec, cols = golden_problem(pkg, oracle.default_params(20, 0), which)
    lib = oracle.lib()

    class Model(C.Structure):
        # struct qo_srbd_model (oracle/qo_srbd.h): nleg = 0 means the 4-leg Go1 model
        _fields_ = [("nleg", C.c_int), ("foot", C.c_double * 24), ("inertia", C.c_double * 9),
                    ("inertia_inv", C.c_double * 9), ("mass", C.c_double), ("rot", C.c_double * 9),
                    ("contacts", C.c_double * 8), ("g_body", C.c_double * 3), ("moment_gravity", C.c_double * 3)]

    m = Model()
    m.nleg = 4
    m.foot[:12] = rec["foot_pos_body"][0]
    m.inertia[:] = list(p.inertia)
    m.mass = p.mass
    m.rot[:] = rec["rot"][0]
    m.contacts[:4] = rec["contacts"][0]
    lib.qo_srbd_prepare(C.byref(m))
    lib.qo_srbd_discrete_dynamics.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_float]
    x = Xg[0].copy()
    worst = 0.0
    for k in range(20):
        u = np.zeros(12)
        u[cols] = Ug[k]
        xn = np.zeros(13)
        lib.qo_srbd_discrete_dynamics(C.byref(m), xn.ctypes.data, x.ctypes.data, u.ctypes.data, C.c_float(0.01))
        worst = max(worst, np.abs(xn - Xg[k + 1]).max())
        x = Xg[k + 1].copy()
    assert worst < 1e-11, worst


@pytest.mark.parametrize("which,name,tol", [("stand", "quat_mpc_test.json", 1e-5), ("trot", "trot_quat_mpc_test.json", 1e-5)])
def test_golden_forces_converged(oracle, pkg, which, name, tol):
    """Converged oracle vs the reference solver's output: <= 1e-5 N over the
    horizon (measured 5.4e-6 / 1.6e-6; the JSON is a tolerance-terminated iterate)."""
    Xg, Ug, _ = _golden(name)
    p, rec, cols = golden_problem(pkg, oracle.default_params(20, 0), which)
    f, info, tu, tx = oracle.solve(p, rec, want_traj=True)
    assert info["status"][0] == 0
    assert np.abs(tu[0][:, cols] - Ug).max() < tol
    assert np.abs(tx[0] - Xg).max() < 1e-5
    # k = 0 forces (what the controller applies): BASELINE.md section 2
    if which == "stand":
        ref = [29.46622148056, 29.07350149774, 33.90670213022, 33.51398214740]
        assert np.abs(f[0][[2, 5, 8, 11]] - ref).max() < 1e-6


def test_golden_forces_reference_mode_two_iterations(oracle, pkg):
    """The restated AL-iLQR (reference mode) lands on the same optimum."""
    Xg, Ug, _ = _golden("quat_mpc_test.json")
    p, rec, cols = golden_problem(pkg, oracle.default_params(20, 1), "stand")
    p.tol_stationarity = 1e-9   # do not stop at the loose upstream tolerance
    f, info, tu, _ = oracle.solve(p, rec, want_traj=True)
    assert np.abs(tu[0] - Ug).max() < 1e-5


# ---- problem construction -------------------------------------------------------
def test_default_params_match_yaml(oracle):
    p = oracle.default_params(20, 1)
    assert p.horizon == 20 and abs(p.h - 0.01) < 1e-9 and p.h_ref == 0.01
    assert p.mass == 12.84 and p.w == 50.0 and p.mu == 0.7 and p.fz_max == 100.0
    assert list(p.q_weights) == [2.5, 2.5, 10.0, 0, 0, 0, 0, 0.1, 0.1, 0.1, 0.15, 0.15, 0.15]
    assert all(r == 1e-6 for r in p.r_weights)
    assert abs(p.inertia[0] - 1.2 * 0.0168128557) < 1e-18
    assert p.iterations_max == 10 and p.penalty_scaling == 20.0 and p.drop_ang_vel == 1


def test_reference_trajectory(oracle, pkg):
    p = oracle.default_params(10, 0)
    rec = pkg.random_go1_trot_states(1, config_id=2)
    xref = np.zeros((11, 13)); uref = np.zeros(12)
    oracle.lib().qo_build_reference(C.byref(p), rec.ctypes.data_as(C.c_void_p), xref.ctypes.data_as(C.c_void_p),
                                    uref.ctypes.data_as(C.c_void_p))
    r = rec[0]
    nc = r["contacts"].sum()
    assert np.allclose(uref.reshape(4, 3)[:, 2], r["contacts"] * 12.84 * 9.81 / nc, rtol=0, atol=1e-13)
    for k in range(11):
        assert xref[k][0] == r["pos_ref_body"][0] + r["vel_ref_body"][0] * k * 10.0 / 1000.0
        assert xref[k][2] == r["pos_ref_body"][2]
        assert (xref[k][3:7] == r["quat_d"]).all() and (xref[k][10:] == 0).all()


def test_ang_vel_is_dropped_like_the_reference(oracle, pkg):
    """QuatMpc.cpp:242-245: the ';' ends the initialiser, x_init[10:13] stays 0."""
    p = oracle.default_params(10, 0)
    rec = pkg.random_go1_trot_states(4, config_id=2)
    f1, _ = oracle.solve(p, rec)
    rec2 = rec.copy(); rec2["ang_vel_body"] += 1.0
    f2, _ = oracle.solve(p, rec2)
    assert np.array_equal(f1, f2)
    p.drop_ang_vel = 0
    f3, _ = oracle.solve(p, rec2)
    assert np.abs(f3 - f1).max() > 1e-3


# ---- converged mode on the benchmark workload ----------------------------------
def test_random_trot_states_converge_and_are_feasible(oracle, pkg):
    p = oracle.default_params(10, 0)
    rec = pkg.random_go1_trot_states(128, config_id=2)
    f, info = oracle.solve(p, rec, threads=4)
    assert (info["status"] == 0).all()
    assert info["max_violation"].max() < 1e-8
    # swing legs carry exactly zero force; stance forces obey the world-frame cone
    fw = np.einsum("bij,blj->bli", rec["rot"].reshape(-1, 3, 3), f.reshape(-1, 4, 3))
    sw = rec["contacts"] == 0
    assert (f.reshape(-1, 4, 3)[sw] == 0).all()
    st = ~sw
    assert (fw[st][:, 2] <= 100 + 1e-7).all() and (fw[st][:, 2] >= -1e-7).all()
    assert (np.abs(fw[st][:, 0]) <= 0.7 * fw[st][:, 2] + 1e-7).all()
    assert (np.abs(fw[st][:, 1]) <= 0.7 * fw[st][:, 2] + 1e-7).all()


def test_solution_independent_of_centering_schedule(oracle, pkg):
    """Two different interior-point schedules must meet at the same KKT point."""
    rec = pkg.random_go1_trot_states(64, config_id=2)
    p = oracle.default_params(10, 0)
    f1, i1 = oracle.solve(p, rec, threads=4)
    p.ipm_sigma, p.ipm_sigma_fast = 0.3, 0.3
    f2, i2 = oracle.solve(p, rec, threads=4)
    assert (i1["status"] == 0).all() and (i2["status"] == 0).all()
    assert np.abs(f1 - f2).max() < 1e-7


def test_edge_cases(oracle, pkg):
    p = oracle.default_params(10, 0)
    rec = pkg.random_go1_trot_states(3, config_id=2)
    rec["contacts"][0] = 0           # no stance leg: reference divides 0/0 (QuatMpc.cpp:122)
    rec["quat"][1][2] = np.nan       # non-finite record
    f, info = oracle.solve(p, rec)
    assert info["status"].tolist()[:2] == [2, 3] and info["status"][2] == 0
    assert (f[:2] == 0).all()


def test_scenarios_are_counter_based(pkg):
    a = pkg.random_go1_trot_states(64, config_id=2)
    b = pkg.random_go1_trot_states(16, config_id=2, first=48)
    assert a[48:].tobytes() == b.tobytes()
    c = pkg.random_go1_trot_states(16, config_id=3, first=48)
    assert a[48:].tobytes() != c.tobytes()
    assert np.allclose(np.linalg.norm(a["quat"], axis=1), 1.0)
    assert set(np.unique(a["contacts"].sum(1))) <= {2.0, 4.0}


def test_kat_double_integrator_unconstrained(oracle):
    """TestDoubleIntegrator.cpp:69-168: Success within iterations_max = 3; the final state is closer to
    the goal than x0 = (1, 2, 0, 0) but not on it (:166-167)."""
    it, status, dist, *_ = oracle.kat_double_integrator(2)
    assert status == 0 and it <= 3
    assert 1e-3 < dist < np.hypot(1.0, 2.0)


def test_kat_pendulum_goal_constrained(oracle):
    """TestPendulum.cpp:117-203: terminal equality x_N = (pi, 0) met to 1e-4 in at most 10 iterations."""
    it, status, dist, *_ = oracle.kat_pendulum_goal()
    assert status == 0
    assert dist < 1e-4          # :201
    assert it <= 10             # :202


# ---- regimes no reference golden covers: the oracle's own frozen answers ---------------------
REGIMES = (("quat_n10", "random_go1_trot_states", "default_params", "solve", 10, 2),
           ("quat_n20", "random_go1_trot_states", "default_params", "solve", 20, 3),
           ("convex_n20", "random_go1_convex_states", "default_convex_params", "convex_solve", 20, 13),
           ("biped8_n16", "random_biped8_states", "default_biped8_params", "solve8", 16, 5))


@pytest.mark.parametrize("name,gen,dp,solve,N,cfg", REGIMES)
def test_oracle_reproduces_its_frozen_answers(oracle, pkg, name, gen, dp, solve, N, cfg):
    """tests/golden/oracle_regimes.npz (made by tests/golden/make_oracle_fixtures.py): random attitudes, trot
    contacts, active cone faces.  A change to oracle/ or to the state generators shows up here."""
    fx = np.load(GOLDEN / "oracle_regimes.npz")
    want = fx[name + "_forces"]
    rec = getattr(pkg, gen)(len(want), config_id=cfg)
    f, info = getattr(oracle, solve)(getattr(oracle, dp)(N, 0), rec, threads=4)
    assert (info["status"] == 0).all()
    assert np.abs(f - want).max() < 1e-9
    assert np.array_equal(info["iterations"], fx[name + "_iterations"])
