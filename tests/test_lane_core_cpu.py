"""CPU suite: numerics of the lane-per-instance solver core (quaternion-mpc_amd/csrc/qmpc_lane_core.h).

The core is plain C++ that hipcc compiles into qmpc_lane_kernel; tests/native/lane_core_host.cpp compiles THE SAME TEXT
with g++ (one instance after the other, unit strides) so that the wrench-form elimination, the single-precision feedback
gains and the control flow can be checked against the oracle without a GPU.  Test infrastructure only: the product has no
CPU path."""
import ctypes as C
import subprocess
from pathlib import Path

import numpy as np
import pytest

HERE = Path(__file__).resolve().parent
SRC = HERE / "native" / "lane_core_host.cpp"
LIB = HERE / "native" / "liblane_core_host.so"
CORE = HERE.parent / "quaternion-mpc_amd" / "csrc"


@pytest.fixture(scope="module")
def lane(pkg):
    deps = [SRC, CORE / "qmpc_lane_core.h", CORE / "qmpc_params_dev.h", HERE.parent / "include" / "qmpc.h"]
    if not LIB.exists() or any(LIB.stat().st_mtime < d.stat().st_mtime for d in deps):
        # no contraction: the oracle is compiled without it as well
        subprocess.run(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-Wno-unknown-pragmas",
                        "-o", str(LIB), str(SRC)], check=True)
    lib = C.CDLL(str(LIB))
    lib.lane_host_solve.restype = C.c_int

    def solve(p, rec, nu=12):
        rec = np.ascontiguousarray(rec)
        B = rec.shape[0]
        f = np.zeros((B, nu))
        info = np.zeros(B, dtype=pkg.INFO_DTYPE)
        rc = lib.lane_host_solve(C.byref(p), B, rec.ctypes.data_as(C.c_void_p), f.ctypes.data_as(C.c_void_p),
                                 info.ctypes.data_as(C.c_void_p))
        assert rc == 0, rc
        return f, info
    return solve


@pytest.mark.parametrize("N,B,cfg", [(10, 192, 2), (20, 64, 3), (1, 32, 2), (32, 24, 2)])
def test_lane_core_matches_oracle(pkg, oracle, lane, N, B, cfg):
    p = oracle.default_params(N, 0)
    rec = np.concatenate([pkg.go1_stand_input(), pkg.random_go1_trot_states(B - 1, config_id=cfg)])
    rec["contacts"][3] = 0.0
    rec["quat"][6, 0] = np.inf
    f, info = lane(p, rec)
    fo, io = oracle.solve(p, rec, threads=8)
    assert np.array_equal(info["status"], io["status"])
    assert info["status"][3] == pkg.NO_CONTACT and info["status"][6] == pkg.NAN_INPUT
    assert np.abs(f - fo).max() < 1e-6           # measured 3e-10
    di = np.abs(info["iterations"].astype(int) - io["iterations"].astype(int))
    assert (di == 0).mean() >= 0.95 and (di <= 1).mean() >= 0.97
    assert np.abs(f[np.repeat(rec["contacts"] == 0, 3, axis=1)]).max() == 0.0
    ok = info["status"] == 0
    assert np.abs(info["cost"][ok] - io["cost"][ok]).max() < 1e-9 * max(1.0, np.abs(io["cost"][ok]).max())


def test_lane_core_8_point_model(pkg, oracle, lane):
    p = oracle.default_biped8_params(16, 0)
    rec = pkg.random_biped8_states(48, config_id=5)
    f, info = lane(p, rec, nu=24)
    fo, io = oracle.solve8(p, rec, threads=8)
    assert np.array_equal(info["status"], io["status"]) and (info["status"] == 0).all()
    assert np.abs(f - fo).max() < 1e-5 and np.array_equal(info["iterations"], io["iterations"])


@pytest.mark.parametrize("name,gen,dp,N,cfg,nu", [("quat_n10", "random_go1_trot_states", "default_params", 10, 2, 12),
                                                   ("quat_n20", "random_go1_trot_states", "default_params", 20, 3, 12),
                                                   ("biped8_n16", "random_biped8_states", "default_biped8_params", 16, 5, 24)])
def test_lane_core_reaches_the_certified_points(pkg, oracle, lane, name, gen, dp, N, cfg, nu):
    """The algorithm-independent KKT fixtures (tests/test_kkt_certificate.py): the lane core's first-knot forces are the
    certified points' (the lane kernel returns forces only, not the whole input trajectory)."""
    fx = np.load(HERE / "golden" / "kkt_fixtures.npz")
    U = fx[name + "_U"]
    rec = getattr(pkg, gen)(U.shape[0], config_id=cfg)
    f, info = lane(getattr(oracle, dp)(N, 0), rec, nu=nu)
    assert (info["status"] == 0).all()
    assert np.abs(f - U[:, 0, :]).max() < (1e-5 if nu == 24 else 1e-6)


def test_lane_core_outside_the_envelope_returns_finite_forces(pkg, oracle, lane):
    """Long look-ahead on tilted states: failures are reported, never returned as NaN forces (a non-finite
    trial step is not applied), and the converged instances agree with the oracle."""
    p = oracle.default_params(20, 0)
    p.h, p.h_ref = 0.02, 0.02
    rec = pkg.random_go1_trot_states(96, config_id=41)
    f, info = lane(p, rec)
    fo, io = oracle.solve(p, rec, threads=8)
    both = (info["status"] == 0) & (io["status"] == 0)
    assert np.isfinite(f).all() and set(np.unique(info["status"])) <= {pkg.OK, pkg.MAX_ITER, pkg.NOT_PD}
    assert both.mean() > 0.4 and np.abs(f - fo)[both].max() < 1e-5
    assert abs((info["status"] == 0).mean() - (io["status"] == 0).mean()) < 0.1


def test_lane_core_warm_start_matches_oracle(pkg, oracle):
    """The warm-started solve (qmpc_solve_warm*, the warm-started closed loop) on the lane core: the previous solution
    shifted by one knot, a changed contact set (a leg lands, another lifts off), an instance without a usable previous
    solution, the low initial barrier of the loop -- against the oracle's restatement of the same start."""
    deps = [SRC, CORE / "qmpc_lane_core.h"]
    if not LIB.exists() or any(LIB.stat().st_mtime < d.stat().st_mtime for d in deps):
        subprocess.run(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-Wno-unknown-pragmas",
                        "-o", str(LIB), str(SRC)], check=True)
    lib = C.CDLL(str(LIB))
    lib.lane_host_solve_warm.restype = C.c_int

    def warm(p, rec, u_init):
        B, N = rec.shape[0], p.horizon
        f = np.zeros((B, 12)); tu = np.zeros((B, N, 12))
        info = np.zeros(B, dtype=pkg.INFO_DTYPE)
        ui = None if u_init is None else np.ascontiguousarray(u_init)
        rc = lib.lane_host_solve_warm(C.byref(p), B, np.ascontiguousarray(rec).ctypes.data_as(C.c_void_p),
                                      None if ui is None else ui.ctypes.data_as(C.c_void_p), f.ctypes.data_as(C.c_void_p),
                                      info.ctypes.data_as(C.c_void_p), tu.ctypes.data_as(C.c_void_p))
        assert rc == 0, rc
        return f, info, tu

    for N, mu0 in ((10, 0.0), (10, 1e-6), (20, 1e-6)):
        p = oracle.default_params(N, 0)
        if mu0:
            p.ipm_mu0 = mu0
        rec = pkg.random_go1_trot_states(96, config_id=2)
        # cold launch through the warm entry: the same answers as the plain solve, and the trajectory to start from
        f0, i0, tu = warm(p, rec, None)
        fo0, io0, tuo = oracle.solve_warm(p, rec, None)
        assert np.array_equal(i0["status"], io0["status"]) and np.abs(f0 - fo0).max() < 1e-6
        assert np.abs(tu - tuo.reshape(tu.shape)).max() < 1e-6
        # next tick: slightly different states, a changed contact set on some instances
        rec2 = rec.copy()
        rec2["lin_vel_body"] += 0.02
        rec2["contacts"][::7] = rec2["contacts"][::7][:, ::-1]          # stance pair swapped: two legs land, two lift off
        rec2["contacts"][3::11] = 1.0                                     # all four down
        f1, i1, tu1 = warm(p, rec2, tuo.reshape(tu.shape))
        fo1, io1, tuo1 = oracle.solve_warm(p, rec2, tuo)
        assert np.array_equal(i1["status"], io1["status"]) and (i1["status"] == 0).all()
        assert np.abs(f1 - fo1).max() < 1e-6, np.abs(f1 - fo1).max()
        di = np.abs(i1["iterations"].astype(int) - io1["iterations"].astype(int))
        assert (di == 0).mean() >= 0.9 and (di <= 1).mean() >= 0.95, np.bincount(di)
        assert np.abs(tu1 - tuo1.reshape(tu1.shape)).max() < 1e-5
        print(f"N={N} mu0={mu0}: warm iterations {i1['iterations'].mean():.2f} (cold {i0['iterations'].mean():.2f}), "
              f"max force difference to the oracle {np.abs(f1 - fo1).max():.2e} N, iteration counts equal on {(di == 0).mean():.3f}")


@pytest.mark.parametrize("N", [10, 20])
def test_lane_core_convex_model_matches_oracle(pkg, oracle, lane, N):
    """ConvexMpc's problem on the same core (MD_CONVEX): Euler-angle single rigid body, world-frame forces."""
    p = oracle.default_convex_params(N, 0)
    rec = pkg.random_go1_convex_states(96, config_id=12)
    rec["contacts"][4] = 0.0
    f, info = lane(p, rec)
    fo, io = oracle.convex_solve(p, rec, threads=8)
    assert np.array_equal(info["status"], io["status"]) and info["status"][4] == pkg.NO_CONTACT
    assert np.abs(f - fo).max() < 1e-6           # measured 6e-11
    di = np.abs(info["iterations"].astype(int) - io["iterations"].astype(int))
    assert (di == 0).mean() >= 0.95 and (di <= 1).mean() >= 0.97
    ok = info["status"] == 0
    assert np.abs(info["cost"][ok] - io["cost"][ok]).max() < 1e-9 * max(1.0, np.abs(io["cost"][ok]).max())


@pytest.mark.parametrize("N,B,cfg", [(10, 160, 2), (20, 64, 3), (5, 48, 2)])
def test_lane_core_reference_mode_matches_oracle(pkg, oracle, lane, N, B, cfg):
    """The reference's own solver mode on the lane passes (qmpc_lane_core.h: lane_solve_ref -- AL weights through leg_block,
    expected decrease through the wrench form, line-search trial pass, stationarity sweep, dual update) against the oracle's
    restatement of that scheme: identical status words and iteration counts; a truncated iterate does not damp rounding
    -- the AL passes therefore keep their feedback gains in double precision (end of round 5; the packed single-precision
    gains of the converged mode held 95-97 % of the N = 20 forces to 1e-6 N): every force within 1e-7 N (measured: worst 1.1e-10 N)."""
    p = oracle.default_params(N, 1)
    rec = np.concatenate([pkg.go1_stand_input(), pkg.random_go1_trot_states(B - 1, config_id=cfg)])
    rec["contacts"][3] = 0.0
    rec["quat"][6, 0] = np.inf
    f, info = lane(p, rec)
    fo, io = oracle.solve(p, rec, threads=8)
    assert np.array_equal(info["status"], io["status"])
    assert np.array_equal(info["iterations"], io["iterations"])
    assert info["status"][3] == pkg.NO_CONTACT and info["status"][6] == pkg.NAN_INPUT
    d = np.abs(f - fo).max(axis=1)
    print(f"lane core, reference mode N={N}: forces within 1e-6 N on {100 * (d < 1e-6).mean():.1f} %, median {np.median(d):.1e}, worst {d.max():.1e}; "
          f"status counts {np.bincount(info['status'], minlength=6).tolist()}")
    assert d.max() < 1e-7
    assert (info["iterations"] <= 10).all()
    assert np.abs(f[np.repeat(rec["contacts"] == 0, 3, axis=1)]).max() == 0.0
    solved = io["status"] <= 1
    assert np.abs(info["cost"][solved] - io["cost"][solved]).max() < 1e-6 * max(1.0, np.abs(io["cost"][solved]).max())


@pytest.mark.parametrize("N", [10, 20])
def test_lane_core_convex_reference_mode_matches_oracle(pkg, oracle, lane, N):
    """ConvexMpc's OWN solver mode (five AL-iLQR iterations, ConvexMpc.cpp:36-38) on the lane passes (lane_solve_ref<4, MD_CONVEX>):
    status words and iteration counts identical to the oracle's; forces of the truncated iterates within 1e-7 N on every
    instance (double-precision feedback gains in the AL passes; measured: worst 2.2e-11 N)."""
    p = oracle.default_convex_params(N, 1)
    rec = pkg.random_go1_convex_states(96, config_id=12)
    rec["contacts"][4] = 0.0
    f, info = lane(p, rec)
    fo, io = oracle.convex_solve(p, rec, threads=8)
    assert np.array_equal(info["status"], io["status"]) and info["status"][4] == pkg.NO_CONTACT
    assert np.array_equal(info["iterations"], io["iterations"]) and (info["iterations"] <= p.iterations_max).all()
    d = np.abs(f - fo).max(axis=1)
    print(f"lane core, ConvexMpc reference mode N={N}: forces within 1e-6 N on {100 * (d < 1e-6).mean():.1f} %, median {np.median(d):.1e}, "
          f"worst {d.max():.1e}; status counts {np.bincount(info['status'], minlength=6).tolist()}")
    assert d.max() < 1e-7
    solved = io["status"] <= 1
    assert np.abs(info["cost"][solved] - io["cost"][solved]).max() < 1e-6 * max(1.0, np.abs(io["cost"][solved]).max())


def test_lane_core_8_point_reference_mode_matches_oracle(pkg, oracle, lane):
    """The 8-contact-point model in the reference's solver mode on the lane passes (lane_solve_ref<8>)."""
    p = oracle.default_biped8_params(16, 1)
    rec = pkg.random_biped8_states(48, config_id=5)
    f, info = lane(p, rec, nu=24)
    fo, io = oracle.solve8(p, rec, threads=8)
    assert np.array_equal(info["status"], io["status"]) and np.array_equal(info["iterations"], io["iterations"])
    d = np.abs(f - fo).max(axis=1)
    assert d.max() < 1e-7, (np.median(d), d.max())          # measured with the double-precision gains: worst below 1e-9 N
