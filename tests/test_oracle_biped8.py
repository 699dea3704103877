"""CPU checks of the 8-contact-point generalisation of the oracle (BASELINE config 5; the humanoid
branch is not in the reference checkout, so the robot is synthetic and nothing upstream pins it).

The generalisation itself IS pinned: with Go1 parameters, the Go1 footholds in points 0-3 and points
4-7 in swing, the 8-point problem is the golden-pinned 4-leg problem."""
import numpy as np
import pytest


def _embed(pkg, rec4):
    rec8 = np.zeros(len(rec4), dtype=pkg.INPUT8_DTYPE)
    for f in ("quat", "rot", "lin_vel_body", "ang_vel_body", "pos_ref_body", "vel_ref_body", "acc_ref_body", "quat_d"):
        rec8[f] = rec4[f]
    rec8["foot_pos_body"][:, :12] = rec4["foot_pos_body"]
    rec8["foot_pos_body"][:, 12:] = rec4["foot_pos_body"] + 0.01      # swing points: position irrelevant
    rec8["contacts"][:, :4] = rec4["contacts"]
    return rec8


def test_eight_point_problem_reduces_to_the_four_leg_one(pkg, oracle):
    rec4 = pkg.random_go1_trot_states(48, config_id=2)
    p4 = oracle.default_params(10, 0)
    p8 = oracle.default_params(10, 0)
    p8.model = pkg.MODEL_QUAT8
    f4, i4, u4, x4 = oracle.solve(p4, rec4, threads=4, want_traj=True)
    f8, i8, u8, x8 = oracle.solve8(p8, _embed(pkg, rec4), threads=4, want_traj=True)
    assert (i4["status"] == 0).all() and (i8["status"] == 0).all()
    assert np.abs(f8[:, :12] - f4).max() < 1e-7
    assert np.abs(f8[:, 12:]).max() == 0.0
    assert np.abs(x8 - x4).max() < 1e-9
    assert np.array_equal(i8["iterations"], i4["iterations"])


@pytest.mark.parametrize("N", [16, 6])
def test_biped_states_converge_feasible(pkg, oracle, N):
    p = oracle.default_biped8_params(N, 0)
    assert p.model == pkg.MODEL_QUAT8 and p.mass == 30.0 and p.fz_max == 250.0
    rec = pkg.random_biped8_states(96, config_id=5)
    support = rec["contacts"].reshape(-1, 2, 4)
    assert ((support == support[:, :, :1]).all())                       # a foot's four corners share one flag
    assert set(map(tuple, support[:, :, 0])) <= {(1, 1), (1, 0), (0, 1)}
    f, info, tu, tx = oracle.solve8(p, rec, threads=4, want_traj=True)
    assert (info["status"] == 0).all() and info["max_violation"].max() < 1e-8
    swing = np.repeat(rec["contacts"] == 0, 3, axis=1)
    assert np.abs(f[swing]).max() == 0.0
    # pyramid in the WORLD frame (QuatMpc.cpp:194-205): rotate the body-frame forces with R
    R = rec["rot"].reshape(-1, 3, 3)
    fw = np.einsum("bij,bkj->bki", R, f.reshape(-1, 8, 3))
    assert (fw[..., 2] >= -1e-8).all() and (fw[..., 2] <= p.fz_max + 1e-8).all()
    assert (np.abs(fw[..., 0]) <= p.mu * fw[..., 2] + 1e-7).all()
    assert (np.abs(fw[..., 1]) <= p.mu * fw[..., 2] + 1e-7).all()
