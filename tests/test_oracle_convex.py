"""CPU checks of the ConvexMpc restatement (oracle/qo_convex.c).

PARITY UNPINNED against the reference: its only artefact for this controller
(src/test/test_altro/convex_mpc.json) comes from a test that is not built, uses
forward Euler, mass 13 and one solver iteration, and cannot be reproduced from the
current model.  What is checked here is the restated model against itself:
  * the discrete Jacobian against central differences where the reference's
    ct_srb_jacobian is exact (all of B; A except the yaw column and the yaw-rate
    column, whose d(I_world^-1)/d(yaw) terms the reference omits, AltroUtils.cpp:354-357)
  * physics of a symmetric stand (weight shared by the four legs)
  * cone feasibility, pinned swing legs, status codes on random states
"""
import numpy as np
import pytest


@pytest.fixture(scope="module")
def cvx(pkg, oracle):
    return pkg.random_go1_convex_states(128, config_id=12)


def _uref(p, rec):
    nc = (rec["contacts"] != 0).sum()
    u = np.zeros(12)
    u[2::3] = p.mass * 9.81 / nc * (rec["contacts"] != 0)
    return u


def test_convex_record_and_defaults(pkg, oracle):
    p = oracle.default_convex_params(20, 0)
    assert p.model == pkg.MODEL_CONVEX and p.horizon == 20
    assert abs(p.h - 0.005) < 1e-9 and p.h_ref == 0.005          # mpc_update_period 5 ms
    assert list(p.q_weights)[:12] == [3, 3, 3, 1, 1, 20, 0, 0, 3, 2, 3, 2]
    assert p.mu == 0.6 and p.fz_max == 200.0 and p.mass == 12.84
    assert p.inertia[0] == 0.0168128557 and p.inertia[4] == 0.063009565 and p.inertia[8] == 0.0716547275
    assert pkg.CONVEX_INPUT_DTYPE.itemsize == 384


def test_discrete_jacobian_vs_central_differences(pkg, oracle, cvx):
    p = oracle.default_convex_params(4, 0)
    A, B, X = oracle.convex_linearize(p, cvx[:6])
    eps = 1e-6
    for b in range(6):
        rec = cvx[b:b + 1]
        x, u = X[b, 0].copy(), _uref(p, cvx[b])
        Afd, Bfd = np.zeros((12, 12)), np.zeros((12, 12))
        for j in range(12):
            d = np.zeros(12); d[j] = eps
            Afd[:, j] = (oracle.convex_step(p, rec, x + d, u) - oracle.convex_step(p, rec, x - d, u)) / (2 * eps)
            Bfd[:, j] = (oracle.convex_step(p, rec, x, u + d) - oracle.convex_step(p, rec, x, u - d)) / (2 * eps)
        assert np.abs(B[b, 0] - Bfd).max() < 1e-8
        keep = [c for c in range(12) if c not in (2, 8)]
        assert np.abs(A[b, 0][:, keep] - Afd[:, keep]).max() < 1e-8
        # the omitted terms are O(h^2): the two inexact columns stay close
        assert np.abs(A[b, 0] - Afd).max() < 1.0


def test_symmetric_stand_shares_the_weight(pkg, oracle):
    p = oracle.default_convex_params(10, 0)
    rec = np.zeros(1, dtype=pkg.CONVEX_INPUT_DTYPE)
    rec["pos_world"][0] = [0, 0, 0.3]
    rec["pos_d_world"][0] = [0, 0, 0.3]
    rec["foot_pos_abs_com"][0] = np.array([[0.2, 0.14, -0.3], [0.2, -0.14, -0.3], [-0.2, 0.14, -0.3], [-0.2, -0.14, -0.3]]).reshape(12)
    rec["contacts"][0] = 1.0
    f, info = oracle.convex_solve(p, rec)
    assert info["status"][0] == 0
    f = f.reshape(4, 3)
    assert np.abs(f[:, :2]).max() < 1e-6
    assert np.abs(f[:, 2] - p.mass * 9.81 / 4).max() < 1e-6


@pytest.mark.parametrize("N", [10, 20])
def test_random_states_converge_feasible(pkg, oracle, cvx, N):
    p = oracle.default_convex_params(N, 0)
    f, info, tu, tx = oracle.convex_solve(p, cvx, threads=4, want_traj=True)
    assert (info["status"] == 0).all()
    assert info["max_violation"].max() < 1e-8
    u = tu.reshape(len(cvx), N, 4, 3)
    swing = (cvx["contacts"] == 0)[:, None, :, None]
    assert np.abs(np.where(swing, u, 0.0)).max() == 0.0                       # pinned exactly
    fz, fx, fy = u[..., 2], u[..., 0], u[..., 1]
    assert (fz >= -1e-9).all() and (fz <= p.fz_max + 1e-9).all()
    assert (np.abs(fx) <= p.mu * fz + 1e-8).all() and (np.abs(fy) <= p.mu * fz + 1e-8).all()
    assert np.array_equal(f, tu[:, 0])
    # the trajectory is the model's own rollout of the inputs
    for b in (0, 7):
        x = tx[b, 0].copy()
        for k in range(N):
            x = oracle.convex_step(p, cvx[b:b + 1], x, tu[b, k])
            assert np.abs(x - tx[b, k + 1]).max() < 1e-12


def test_status_codes(pkg, oracle, cvx):
    p = oracle.default_convex_params(10, 0)
    rec = cvx[:3].copy()
    rec["contacts"][0] = 0.0
    rec["lin_vel_world"][1, 0] = np.nan
    f, info = oracle.convex_solve(p, rec)
    assert list(info["status"]) == [pkg.NO_CONTACT, pkg.NAN_INPUT, 0]
    assert np.abs(f[:2]).max() == 0.0
