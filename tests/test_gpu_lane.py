"""GPU suite (-m gpu): the lane-per-instance kernel of large batches (csrc/qmpc_lane.hip, qmpc_lane_core.h) through the C ABI,
against the CPU oracle and against the wave-per-instance kernels.

The lane kernel solves the same problem with the same interior-point iteration; the Newton system of a knot is eliminated
in the wrench form (qmpc_lane_core.h), so iterates agree with the oracle to rounding and iteration counts are the same
except where a stopping test sits on its threshold.  Tolerances: forces 1e-6 N (measured 5e-11), status words equal,
iteration counts equal on >= 97 % of the instances; swing-leg forces exactly 0."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def lib(pkg):
    return pkg.load_library()


def _forced(monkeypatch, variant):
    monkeypatch.setenv("QMPC_VARIANT", str(variant))     # read by qmpc_create


def _edge_records(pkg, rec):
    """rec with a stand pose in front, one record without stance legs and one with a NaN: status paths of the set-up"""
    rec = np.concatenate([pkg.go1_stand_input(), rec])
    rec["contacts"][5] = 0.0
    rec["lin_vel_body"][9, 1] = np.nan
    return rec


@pytest.mark.parametrize("N,B", [(10, 4096), (20, 1024), (5, 700), (1, 130), (32, 256)])
def test_lane_kernel_matches_oracle(pkg, lib, oracle, monkeypatch, N, B):
    _forced(monkeypatch, 4)
    p = pkg.default_params(N, pkg.MODE_CONVERGED, lib)
    rec = _edge_records(pkg, pkg.random_go1_trot_states(B - 1, config_id=2 if N != 20 else 3))
    s = pkg.Solver(p, B, device=0, lib=lib)
    f, info = s.solve(rec)
    s.close()
    fo, io = oracle.solve(p, rec, threads=8)
    assert np.array_equal(info["status"], io["status"])
    assert info["status"][5] == pkg.NO_CONTACT and info["status"][9] == pkg.NAN_INPUT
    assert (np.delete(info["status"], [5, 9]) == 0).all()
    assert np.abs(f - fo).max() < 1e-6
    assert np.abs(f[np.repeat(rec["contacts"] == 0, 3, axis=1)]).max() == 0.0
    assert np.abs(f[[5, 9]]).max() == 0.0
    di = np.abs(info["iterations"].astype(int) - io["iterations"].astype(int))
    # long horizons have instances with 50+ iterations whose paths part (both reach the same point): counts within one on 98 %
    assert (di == 0).mean() >= 0.97 and (di <= 1).mean() >= 0.98
    ok = info["status"] == 0
    assert np.abs(info["cost"][ok] - io["cost"][ok]).max() < 1e-9 * max(1.0, np.abs(io["cost"][ok]).max())
    assert info["max_violation"].max() < 1e-8


def test_lane_kernel_equals_wave_kernel_on_the_same_batch(pkg, lib, monkeypatch):
    """Both kernel families on one batch (device buffers): same status words, forces within 1e-7 N, and the lane kernel's
    result of an instance does not depend on where in the batch (which lane, which wavefront) it is solved."""
    B, N = 8192, 10
    p = pkg.default_params(N, pkg.MODE_CONVERGED, lib)
    rec = pkg.random_go1_trot_states(B, config_id=2)
    out = {}
    for v in (4, 0):
        _forced(monkeypatch, v)
        s = pkg.Solver(p, B, device=0, lib=lib)
        out[v] = s.solve(rec)
        if v == 4:
            perm = np.random.default_rng(3).permutation(B)
            fp, ip = s.solve(rec[perm])
            assert np.array_equal(fp, out[4][0][perm]) and np.array_equal(ip["iterations"], out[4][1]["iterations"][perm])
            f2, _ = s.solve(rec)
            assert np.array_equal(f2, out[4][0])                     # deterministic
            fs, _ = s.solve(rec[:100])                               # a ragged, partly filled wavefront
            assert np.array_equal(fs, out[4][0][:100])
        s.close()
    assert np.array_equal(out[4][1]["status"], out[0][1]["status"])
    assert np.abs(out[4][0] - out[0][0]).max() < 1e-7


def test_lane_kernel_8_point_model(pkg, lib, oracle, monkeypatch):
    _forced(monkeypatch, 4)
    N, B = 16, 512
    p = pkg.default_biped8_params(N, pkg.MODE_CONVERGED, lib)
    rec = pkg.random_biped8_states(B, config_id=5)
    s = pkg.Solver(p, B, device=0, lib=lib)
    f, info = s.solve8(rec)
    s.close()
    fo, io = oracle.solve8(p, rec, threads=8)
    assert np.array_equal(info["status"], io["status"]) and (info["status"] == 0).all()
    assert np.abs(f - fo).max() < 1e-5                                # corner forces of a foot: DESIGN.md 3d
    feet = rec["foot_pos_body"].reshape(B, 8, 3)
    w = lambda x: np.concatenate([x.reshape(B, 8, 3).sum(1), np.cross(feet, x.reshape(B, 8, 3)).sum(1)], axis=1)
    assert np.abs(w(f) - w(fo)).max() < 1e-7
    assert np.abs(f[np.repeat(rec["contacts"] == 0, 3, axis=1)]).max() == 0.0


@pytest.mark.parametrize("name,gen,dp,N,cfg,nu", [("quat_n10", "random_go1_trot_states", "default_params", 10, 2, 12),
                                                   ("quat_n20", "random_go1_trot_states", "default_params", 20, 3, 12),
                                                   ("biped8_n16", "random_biped8_states", "default_biped8_params", 16, 5, 24)])
def test_lane_kernel_reaches_the_certified_points(pkg, lib, monkeypatch, name, gen, dp, N, cfg, nu):
    """First-knot forces of the algorithm-independent KKT fixtures (tests/test_kkt_certificate.py), no oracle involved."""
    from pathlib import Path
    _forced(monkeypatch, 4)
    U = np.load(Path(__file__).parent / "golden" / "kkt_fixtures.npz")[name + "_U"]
    rec = getattr(pkg, gen)(U.shape[0], config_id=cfg)
    s = pkg.Solver(getattr(pkg, dp)(N, pkg.MODE_CONVERGED, lib), U.shape[0], device=0, lib=lib)
    f, info = (s.solve8 if nu == 24 else s.solve)(rec)
    s.close()
    assert (info["status"] == 0).all()
    assert np.abs(f - U[:, 0, :]).max() < (1e-5 if nu == 24 else 1e-6)


def test_lane_kernel_parameters_follow_the_handle(pkg, lib, oracle, monkeypatch):
    """The lane kernel reads its parameter block from constant memory, one slot per handle: two handles with different
    parameters, used alternately, and qmpc_set_params between two solves."""
    _forced(monkeypatch, 4)
    rec = pkg.random_go1_trot_states(256, config_id=2)
    pa = pkg.default_params(10, pkg.MODE_CONVERGED, lib)
    pb = pkg.default_params(10, pkg.MODE_CONVERGED, lib)
    pb.mu, pb.fz_max, pb.w = 0.5, 80.0, 20.0
    sa, sb = pkg.Solver(pa, 256, device=0, lib=lib), pkg.Solver(pb, 256, device=0, lib=lib)
    fa, _ = sa.solve(rec); fb, _ = sb.solve(rec); fa2, _ = sa.solve(rec)
    assert np.array_equal(fa, fa2)
    assert np.abs(fa - oracle.solve(pa, rec, threads=8)[0]).max() < 1e-6
    assert np.abs(fb - oracle.solve(pb, rec, threads=8)[0]).max() < 1e-6
    sa.set_params(pb)
    assert np.array_equal(sa.solve(rec)[0], fb)
    sa.close(); sb.close()


def test_config4_workload_on_one_gpu(pkg, lib, oracle):
    """BASELINE config 4's own workload -- 262144 Go1 instances, N = 10, generator seed 0x5EED0000 + 4 -- on ONE GPU (the
    8-GPU run shards it in blocks of 32768): four rounds of the resident lanes of the lane kernel.  Size-independent
    properties + a 64-instance sample against the oracle."""
    B, N = 262144, 10
    p = pkg.default_params(N, pkg.MODE_CONVERGED, lib)
    rec = pkg.random_go1_trot_states(B, config_id=4)
    s = pkg.Solver(p, B, device=0, lib=lib)
    f, info = s.solve(rec)
    assert (info["status"] == 0).all(), np.unique(info["status"], return_counts=True)
    assert info["max_violation"].max() < 1e-8
    assert np.abs(f[np.repeat(rec["contacts"] == 0, 3, axis=1)]).max() == 0.0
    R = rec["rot"].reshape(B, 3, 3)
    fw = np.einsum("bij,blj->bli", R, f.reshape(B, 4, 3))
    assert (fw[..., 2] >= -1e-8).all() and (fw[..., 2] <= p.fz_max + 1e-8).all()
    assert (np.abs(fw[..., 0]) <= p.mu * fw[..., 2] + 1e-7).all() and (np.abs(fw[..., 1]) <= p.mu * fw[..., 2] + 1e-7).all()
    # the shard a rank of the 8-GPU run would solve gives the same bits
    fsh, _ = s.solve(rec[3 * 32768:4 * 32768])
    assert np.array_equal(fsh, f[3 * 32768:4 * 32768])
    sub = np.arange(0, B, B // 64)[:64]
    fo, io = oracle.solve(p, rec[sub], threads=8)
    assert (io["status"] == 0).all() and np.abs(f[sub] - fo).max() < 1e-6
    s.close()


@pytest.mark.parametrize("N,B", [(10, 4096), (20, 1024), (3, 300)])
def test_lane_kernel_reference_mode(pkg, lib, oracle, monkeypatch, N, B):
    """QMPC_MODE_REFERENCE on the lane kernel (qmpc_lane_ref_kernel: the AL variant of the lane passes, line search in lock
    step): against the oracle's reference mode on every instance -- identical status words and iteration counts, forces
    within 1e-6 N on >= 95 % (truncated iterates) -- and against the wave-per-instance reference kernels; edge records keep
    their status words, swing legs are exactly 0, trajectories come back."""
    p = pkg.default_params(N, pkg.MODE_REFERENCE, lib)
    rec = _edge_records(pkg, pkg.random_go1_trot_states(B - 1, config_id=2 if N != 20 else 3))
    out = {}
    for v in (4, 0):
        _forced(monkeypatch, v)
        monkeypatch.setenv("QMPC_LANE_REF_MIN", str(1 << 30))       # variant 0: the wave kernels at this size
        s = pkg.Solver(p, B, device=0, lib=lib)
        f, info, tu, tx = s.solve(rec, want_traj=True) if v == 0 else (*s.solve(rec), None, None)
        if v == 4:
            f2, info2, tu, tx = s.solve(rec, want_traj=True)        # trajectories come from the lane kernel as well
            assert np.array_equal(f, f2) and np.array_equal(info["iterations"], info2["iterations"])
            assert pkg.KERNEL_FAMILY[s.query(pkg.QUERY_LAST_KERNEL)] == "lane" and np.array_equal(tu[:, 0, :], f2)
        s.close()
        out[v] = (f, info)
        out[("traj", v)] = (tu, tx)
    fl, il = out[4]
    fw, iw = out[0]
    dt = np.abs(out[("traj", 4)][1] - out[("traj", 0)][1]).reshape(B, -1).max(axis=1)
    # (state trajectories of two truncated iterates, every knot of the horizon: with the feedback gains of the AL passes in double
    # precision -- end of round 5 -- the two kernel families agree like their first-knot forces do; the packed single-precision
    # gains left 83 % of the N = 20 trajectories within 1e-6)
    print(f"lane vs wave kernels, reference mode N={N}: state trajectories within 1e-6 on {100 * (dt < 1e-6).mean():.2f} %, worst {dt.max():.1e}")
    assert np.isfinite(dt).all() and (dt < 1e-6).mean() >= 0.995
    fo, io = oracle.solve(p, rec, threads=8)
    assert np.array_equal(il["status"], io["status"]) and np.array_equal(il["iterations"], io["iterations"])
    assert np.array_equal(il["status"], iw["status"]) and np.array_equal(il["iterations"], iw["iterations"])
    assert il["status"][5] == pkg.NO_CONTACT and il["status"][9] == pkg.NAN_INPUT
    d = np.abs(fl - fo).max(axis=1)
    dw = np.abs(fl - fw).max(axis=1)
    print(f"lane kernel, reference mode N={N} B={B}: vs oracle within 1e-6 N on {100 * (d < 1e-6).mean():.1f} % (median {np.median(d):.1e}, worst "
          f"{d.max():.1e}); vs wave kernels {100 * (dw < 1e-6).mean():.1f} %; status counts {np.bincount(il['status'], minlength=6).tolist()}")
    # (double-precision gains in the AL passes: 0.6 M soak instances without one force beyond 7e-9 N of the oracle's; the packed
    # form held 99.9 % / 99.1 % of the N = 10 / 20 forces to 1e-6 N)
    assert (d < 1e-6).mean() >= 0.999 and (dw < 1e-6).mean() >= 0.999
    assert (il["iterations"] <= 10).all() and np.isfinite(fl).all()
    assert (fl.reshape(-1, 4, 3)[rec["contacts"] == 0] == 0).all()
    solved = io["status"] <= 1
    assert np.abs(il["cost"][solved] - io["cost"][solved]).max() < 1e-6 * max(1.0, np.abs(io["cost"][solved]).max())
    assert np.array_equal(il["penalty"][solved], io["penalty"][solved])          # the final penalty


def test_more_live_handles_than_parameter_slots(pkg, lib, oracle, monkeypatch):
    """The lane kernel reads its parameters from a 64-slot constant-memory table, one slot per LIVE handle (free list,
    returned in qmpc_destroy).  70 handles with different friction coefficients alive at once: every one solves its own
    problem (the ones created after the table is full keep the wave-per-instance kernels), also after the early ones have
    solved again; destroyed handles give their slots back."""
    _forced(monkeypatch, 4)
    rec = pkg.random_go1_trot_states(64, config_id=2)
    mus = np.linspace(0.35, 0.9, 70)
    solvers, want = [], []
    for mu in mus:
        p = pkg.default_params(10, pkg.MODE_CONVERGED, lib)
        p.mu = float(mu)
        solvers.append(pkg.Solver(p, 64, device=0, lib=lib))
    for i in (0, 13, 63, 64, 69):
        p = pkg.default_params(10, pkg.MODE_CONVERGED, lib)
        p.mu = float(mus[i])
        want.append((i, oracle.solve(p, rec, threads=8)[0]))
    first = [s.solve(rec)[0] for s in solvers]
    again = [s.solve(rec)[0] for s in solvers]
    for i, fo in want:
        assert np.abs(first[i] - fo).max() < 1e-6 and np.array_equal(first[i], again[i]), i
    assert np.abs(first[0] - first[69]).max() > 1e-3            # different problems
    for s in solvers:
        s.close()
    fresh = [pkg.Solver(pkg.default_params(10, pkg.MODE_CONVERGED, lib), 64, device=0, lib=lib) for _ in range(64)]
    f0 = [s.solve(rec)[0] for s in fresh]
    assert all(np.array_equal(f0[0], f) for f in f0)
    for s in fresh:
        s.close()


@pytest.mark.parametrize("N,B", [(10, 32768), (20, 22528), (24, 20480)])      # (beyond the switch-over of each horizon); N=24: 44 KB of LDS per list-kernel workgroup (512 of them)
def test_straggler_hand_off_of_large_batches(pkg, lib, oracle, monkeypatch, N, B):
    """Cold plain solves the library sends to the lane kernel by itself (qmpc_hip.hip: launch_solve): the lane kernel stops at
    a fixed iteration cap and the wrench-form wave kernel CONTINUES the instances left from their state records.  Against
    the pure lane kernel (QMPC_LANE_CAP=0): same status words, iteration counts equal on >= 99.9 %, forces within 1e-7 N;
    against the oracle on a sample that over-represents the handed-over instances; a second call gives the same bits (the
    cap is a function of the horizon, not of timing); edge records (no stance leg, NaN) keep their status words."""
    p = pkg.default_params(N, pkg.MODE_CONVERGED, lib)
    rec = _edge_records(pkg, pkg.random_go1_trot_states(B - 1, config_id=4 if N == 10 else 3))
    cap = 15 + N // 10
    out = {}
    for tag, env in (("pure", {"QMPC_LANE_CAP": "0"}), ("auto", {}), ("low cap", {"QMPC_LANE_CAP": "12"}),
                     ("restart", {"QMPC_HANDOFF_RESTART": "1"})):       # the wave kernel ignores the records: from scratch
        for k in ("QMPC_LANE_CAP", "QMPC_HANDOFF_RESTART"):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        s = pkg.Solver(p, B, device=0, lib=lib)
        f, info = s.solve(rec)
        f2, info2 = s.solve(rec)
        s.close()
        assert np.array_equal(f, f2) and np.array_equal(info["iterations"], info2["iterations"]), tag
        out[tag] = (f, info)
    fp, ip = out["pure"]
    handed = int((ip["iterations"] > cap).sum())
    assert handed > B // 100, handed          # the workload does leave instances beyond the cap
    for tag in ("auto", "low cap", "restart"):
        f, info = out[tag]
        assert np.array_equal(info["status"], ip["status"]), tag
        same = float((info["iterations"] == ip["iterations"]).mean())
        err = float(np.abs(f - fp).max())
        print(f"hand-off ({tag}), N={N} B={B}: {handed} instances beyond the cap of {cap}; iteration counts equal on {100 * same:.3f} %, "
              f"forces within {err:.2e} N of the pure lane kernel")
        assert same > 0.999 and err < 1e-7
    assert ip["status"][5] == pkg.NO_CONTACT and ip["status"][9] == pkg.NAN_INPUT
    # trajectory outputs across the hand-off: handed-over instances get theirs from the wave kernel
    for k in ("QMPC_LANE_CAP", "QMPC_HANDOFF_RESTART"):
        monkeypatch.delenv(k, raising=False)
    s = pkg.Solver(p, B, device=0, lib=lib)
    ft, it_, tu, tx = s.solve(rec, want_traj=True)
    s.close()
    assert np.array_equal(ft, out["auto"][0]) and np.array_equal(tu[:, 0, :], ft)
    hs = np.where(ip["iterations"] > cap)[0][:64]
    x0 = tx[hs, 0, :]
    assert np.abs(x0[:, 3:7] - rec["quat"][hs]).max() == 0.0 and np.isfinite(tx).all() and np.isfinite(tu).all()
    f, info = out["auto"]
    slow = np.argsort(-ip["iterations"])[:96]                     # the handed-over ones ...
    idx = np.unique(np.concatenate([slow, np.arange(0, B, B // 96)]))     # ... and a spread sample
    idx = idx[(ip["status"][idx] == 0)]
    fo, io = oracle.solve(p, rec[idx], threads=8)
    assert (io["status"] == 0).all()
    assert np.abs(f[idx] - fo).max() < 1e-6
    assert float((info["iterations"][idx] == io["iterations"]).mean()) > 0.97


def _random_params(pkg, lib, rng, t):
    N = int(rng.choice([6, 10, 14, 20]))
    p = pkg.default_params(N, pkg.MODE_CONVERGED, lib)
    p.mu = float(rng.uniform(0.3, 1.0))
    p.fz_max = float(rng.uniform(60, 300))
    p.w = float(rng.uniform(1, 100))
    p.mass = float(rng.uniform(9, 16))
    for i in range(13):
        p.q_weights[i] = float(p.q_weights[i] * rng.uniform(0.3, 3.0))
    for i in range(12):
        p.r_weights[i] = float(10 ** rng.uniform(-6.5, -4))
    p.drop_ang_vel = int(t % 2)
    hs = float(rng.choice([0.005, 0.01, 0.02]))
    if N * hs > 0.3:
        hs = 0.015
    p.h, p.h_ref = hs, hs
    return p


def test_lane_kernel_randomised_parameter_sets(pkg, lib, oracle, monkeypatch):
    """Friction, force limit, weights (per-axis R included), mass, knot spacing, horizon and the angular-velocity quirk at
    random (the draw of test_gpu_parity.py::test_randomised_parameter_sets_match_oracle): the lane kernel follows the oracle
    for every set; so it does with zero velocity weights (S6 = M'PM stays positive definite through the position and
    attitude weights) and with an iteration cap that truncates every solve."""
    _forced(monkeypatch, 4)
    rng = np.random.default_rng(77)
    for t in range(8):
        p = _random_params(pkg, lib, rng, t)
        rec = pkg.random_go1_trot_states(192, config_id=40 + t)
        s = pkg.Solver(p, 192, device=0, lib=lib)
        f, info = s.solve(rec)
        s.close()
        fo, io = oracle.solve(p, rec, threads=8)
        assert (io["status"] == 0).all() and (info["status"] == 0).all(), (t, np.unique(info["status"]))
        assert np.abs(f - fo).max() < 1e-6, t
    rec = pkg.random_go1_trot_states(128, config_id=2)
    p = pkg.default_params(10, pkg.MODE_CONVERGED, lib)
    for i in (7, 8, 9, 10, 11, 12):
        p.q_weights[i] = 0.0
    s = pkg.Solver(p, 128, device=0, lib=lib)
    f, info = s.solve(rec)
    fo, io = oracle.solve(p, rec, threads=8)
    assert (info["status"] == 0).all() and (io["status"] == 0).all() and np.abs(f - fo).max() < 1e-6
    p = pkg.default_params(10, pkg.MODE_CONVERGED, lib)
    p.iterations_max = 6
    s.set_params(p)
    f, info = s.solve(rec)
    s.close()
    fo, io = oracle.solve(p, rec, threads=8)
    assert (info["status"] == pkg.MAX_ITER).all() and (io["status"] == pkg.MAX_ITER).all()
    assert (info["iterations"] == 6).all() and np.abs(f - fo).max() < 1e-6        # the truncated iterates agree as well


def test_lane_kernel_outside_the_envelope(pkg, lib, oracle, monkeypatch):
    """20 ms knots x N = 20 on strongly tilted states: a third of the instances fail on either side.  The lane
    kernel must fail at the same RATE, report every failure through the status word, return FINITE forces (a non-finite
    trial step is never applied) and agree with the oracle wherever both converge."""
    _forced(monkeypatch, 4)
    p = pkg.default_params(20, pkg.MODE_CONVERGED, lib)
    p.h, p.h_ref = 0.02, 0.02
    rec = pkg.random_go1_trot_states(256, config_id=41)
    s = pkg.Solver(p, 256, device=0, lib=lib)
    f, info = s.solve(rec)
    s.close()
    fo, io = oracle.solve(p, rec, threads=8)
    ok_g, ok_o = info["status"] == 0, io["status"] == 0
    both = ok_g & ok_o
    assert set(np.unique(info["status"])) <= {pkg.OK, pkg.MAX_ITER, pkg.NOT_PD}
    assert abs(ok_g.mean() - ok_o.mean()) < 0.08 and both.mean() > 0.45
    assert np.abs(f - fo)[both].max() < 1e-5
    assert np.isfinite(f).all()


def test_closed_loop_with_the_lane_kernel(pkg, lib, monkeypatch):
    """The device-resident closed loop in its per-tick form (more than 2048 robots) takes its solves from launch_solve, i.e.
    from the lane kernel when that is selected: its workspace must exist and its parameters be uploaded BEFORE the tick
    sequence is captured into a graph.  Same robots through the lane kernel and through the wave-per-instance kernels:
    every solve converges in both, the states agree to the solvers' agreement (a closed loop amplifies 1e-10 N only
    mildly over 40 ticks)."""
    lp = pkg.default_loop_params(lib)
    rng = np.random.default_rng(17)
    B = 3072
    cmds = np.zeros((B, 7))
    cmds[:, 0] = rng.uniform(-0.5, 0.5, B); cmds[:, 1] = rng.uniform(-0.2, 0.2, B); cmds[:, 2] = rng.uniform(0.26, 0.32, B)
    cmds[:, 5] = rng.uniform(-0.5, 0.5, B); cmds[:, 6] = (rng.random(B) < 0.9).astype(float)
    cmds[cmds[:, 6] == 0, :2] = 0.0
    cmds[cmds[:, 6] == 0, 5] = 0.0
    stand = cmds.copy(); stand[:, 6] = 0.0
    st0 = pkg.loop_states(stand, lp, height=0.3, yaw=rng.uniform(-3, 3, B), lib=lib)
    out = {}
    for v in (4, 0):
        _forced(monkeypatch, v)
        s = pkg.Solver(pkg.default_params(10, pkg.MODE_CONVERGED, lib), B, device=0, lib=lib)
        st = s.loop_run(st0, 6, lp)
        st["movement_mode"] = cmds[:, 6]
        st = s.loop_run(st, 40, lp)
        st = s.loop_run(st, 3, lp)          # a second call on the same handle: buffers and parameter slot are reused
        s.close()
        out[v] = st
        assert (st["status"] == 0).all() and np.isfinite(st["pos_world"]).all()
    assert np.array_equal(out[4]["tick"], out[0]["tick"])
    assert np.array_equal(out[4]["contacts"], out[0]["contacts"])
    dp = np.abs(out[4]["pos_world"] - out[0]["pos_world"]).max()
    dv = np.abs(out[4]["lin_vel_world"] - out[0]["lin_vel_world"]).max() if "lin_vel_world" in out[0].dtype.names else 0.0
    print(f"lane vs wave closed loop, {B} robots, 49 ticks: max position difference {dp:.2e} m, velocity {dv:.2e}")
    assert dp < 1e-7


@pytest.mark.parametrize("model", ["quat", "convex"])
def test_reference_mode_closed_loop_with_the_lane_kernel(pkg, lib, monkeypatch, model):
    """The closed loop in the reference's OWN solver mode at Monte-Carlo scale: every tick's solve on qmpc_lane_ref_kernel
    (workspace and parameters set up before the tick sequence is captured), against the same robots on the wave-per-instance
    reference kernels.  Truncated iterates of two kernel families differ by up to 1e-5 N on a few instances (the lane kernel's
    single-precision gains), which a closed loop carries along: gait state exact, positions to 1e-5 m over 46 ticks."""
    lp = pkg.default_loop_params(lib)
    rng = np.random.default_rng(23)
    B = 3072
    cmds = np.zeros((B, 7))
    cmds[:, 0] = rng.uniform(-0.5, 0.5, B); cmds[:, 1] = rng.uniform(-0.2, 0.2, B); cmds[:, 2] = rng.uniform(0.26, 0.32, B)
    cmds[:, 5] = rng.uniform(-0.5, 0.5, B); cmds[:, 6] = (rng.random(B) < 0.9).astype(float)
    cmds[cmds[:, 6] == 0, :2] = 0.0
    cmds[cmds[:, 6] == 0, 5] = 0.0
    stand = cmds.copy(); stand[:, 6] = 0.0
    st0 = pkg.loop_states(stand, lp, height=0.3, yaw=rng.uniform(-3, 3, B), lib=lib)
    p = (pkg.default_convex_params if model == "convex" else pkg.default_params)(10, pkg.MODE_REFERENCE, lib)
    out = {}
    for v in (4, 0):
        _forced(monkeypatch, v)
        monkeypatch.setenv("QMPC_LANE_REF_MIN", str(1 << 30))       # variant 0: the wave kernels at this size
        s = pkg.Solver(p, B, device=0, lib=lib)
        st = s.loop_run(st0, 6, lp)
        st["movement_mode"] = cmds[:, 6]
        st = s.loop_run(st, 40, lp)
        if v == 4:
            assert pkg.KERNEL_FAMILY[s.query(pkg.QUERY_KERNEL_FOR_BATCH, B)] == "lane"
        s.close()
        out[v] = st
        assert np.isfinite(st["pos_world"]).all() and (st["tick"] == 46).all()
    assert np.array_equal(out[4]["contacts"], out[0]["contacts"])
    same = (out[4]["status"] == out[0]["status"]) & (out[4]["iterations"] == out[0]["iterations"])
    dp = np.abs(out[4]["pos_world"] - out[0]["pos_world"]).max(axis=1)
    print(f"reference-mode closed loop ({model}), lane vs wave kernels, {B} robots, 46 ticks: position difference median {np.median(dp):.1e} m, "
          f"worst {dp.max():.1e} m; last tick's status / iteration words equal on {100 * same.mean():.2f} %")
    assert dp.max() < 1e-5 and same.mean() >= 0.99
    assert (np.abs(out[4]["pos_world"][:, 2] - cmds[:, 2]) < 0.08).all()


def test_closed_loop_at_the_automatic_switch_over(pkg, lib):
    """24576 robots: beyond the batch size from which qmpc_solve* picks the lane kernel by itself; the loop must do the same."""
    lp = pkg.default_loop_params(lib)
    B = 24576
    cmds = np.zeros((B, 7)); cmds[:, 0] = 0.3; cmds[:, 2] = 0.3
    st0 = pkg.loop_states(cmds, lp, height=0.3, lib=lib)
    s = pkg.Solver(pkg.default_params(10, pkg.MODE_CONVERGED, lib), B, device=0, lib=lib)
    st = s.loop_run(st0, 4, lp)
    st["movement_mode"] = 1.0
    st = s.loop_run(st, 12, lp)
    s.close()
    assert (st["status"] == 0).all() and (st["tick"] == 16).all()
    assert np.abs(st["pos_world"] - st["pos_world"][0]).max() < 1e-9      # identical robots, identical states


@pytest.mark.parametrize("N,warm", [(10, 0), (20, 0), (10, 1), (20, 1)])
def test_closed_loop_hand_off_of_the_lane_kernels_stragglers(pkg, lib, monkeypatch, N, warm):
    """Cold-started closed loop beyond the switch-over: the lane kernel stops at `11 + N/10` iterations per tick and the wave
    kernel continues the robots that are left, inside the captured tick.  20480 robots with different commands, the default cap
    and a cap of 8 (below the 10 iterations an in-gait solve needs: nearly every robot is handed over at every tick) against
    the pure lane kernel (QMPC_LANE_CAP_LOOP=0): same statuses, iteration counts, forces within 1e-9 N and positions within
    1e-9 m after 30 ticks; two runs bit-identical, and the run in two halves equal to the run in one piece."""
    # warm = 1: the warm-started loop (cap 8; the records carry the rows' initial slack residuals), low cap 4
    lp = pkg.default_loop_params(lib)
    lp.warm_start = float(warm)
    B = 28672 if (warm and N == 10) else 20480          # (the warm-started loop takes the lane kernel from 18432 robots on at N <= 12, 20480 beyond)
    env, low = ("QMPC_LANE_CAP_WARM", "4") if warm else ("QMPC_LANE_CAP_LOOP", "8")
    rng = np.random.default_rng(5)
    cmds = np.zeros((B, 7))
    cmds[:, 0] = rng.uniform(-0.5, 0.5, B); cmds[:, 1] = rng.uniform(-0.2, 0.2, B); cmds[:, 2] = rng.uniform(0.26, 0.32, B)
    cmds[:, 5] = rng.uniform(-0.5, 0.5, B)
    st0 = pkg.loop_states(cmds, lp, height=0.3, yaw=rng.uniform(-3, 3, B), lib=lib)
    out = {}
    for cap in ("0", low, None):
        if cap is None:
            monkeypatch.delenv(env, raising=False)
        else:
            monkeypatch.setenv(env, cap)
        prm = pkg.default_params(N, pkg.MODE_CONVERGED, lib)
        if warm:
            prm.ipm_mu0 = 1e-6
        s = pkg.Solver(prm, B, device=0, lib=lib)
        st = s.loop_run(st0, 3, lp)
        st["movement_mode"] = 1.0
        a = s.loop_run(st, 30, lp)
        if cap != "0":
            b = s.loop_run(st, 30, lp)
            assert a.tobytes() == b.tobytes()
            if not warm:          # (the first tick of a warm-started call is a cold one: split runs differ by design)
                h = s.loop_run(s.loop_run(st, 13, lp), 17, lp)
                assert h.tobytes() == a.tobytes()
        s.close()
        out[cap] = a
    p = out["0"]
    assert (p["status"] == 0).all() and (p["iterations"] > (4 if warm else 8)).mean() > 0.9
    for cap in (low, None):
        q = out[cap]
        assert np.array_equal(p["status"], q["status"]) and (p["iterations"] == q["iterations"]).mean() > 0.999
        assert np.abs(p["forces_body"] - q["forces_body"]).max() < 1e-9 and np.abs(p["pos_world"] - q["pos_world"]).max() < 1e-9
        print(f"N={N} cap {cap}: forces within {np.abs(p['forces_body'] - q['forces_body']).max():.1e} N of the pure lane kernel's")


@pytest.mark.parametrize("N,mu0", [(10, 0.0), (10, 1e-6), (20, 1e-6)])
def test_lane_kernel_warm_start_matches_oracle(pkg, lib, oracle, monkeypatch, N, mu0):
    """qmpc_solve_warm on the lane kernel: previous solutions shifted by one knot, changed contact sets, instances without
    a previous solution in the same launch (u_init = None starts everyone cold), trajectory output -- against the oracle's
    restatement of the same start and against the wave-per-instance kernels."""
    B = 192
    rec = pkg.random_go1_trot_states(B, config_id=2)
    rec2 = rec.copy()
    rec2["lin_vel_body"] += 0.02
    rec2["contacts"][::7] = rec2["contacts"][::7][:, ::-1]
    rec2["contacts"][3::11] = 1.0
    res = {}
    for v in (4, 0):
        _forced(monkeypatch, v)
        p = pkg.default_params(N, pkg.MODE_CONVERGED, lib)
        if mu0:
            p.ipm_mu0 = mu0
        s = pkg.Solver(p, B, device=0, lib=lib)
        f0, i0, tu0 = s.solve_warm(rec, None)
        f1, i1, tu1 = s.solve_warm(rec2, tu0)
        s.close()
        res[v] = (f0, i0, tu0, f1, i1, tu1)
    po = oracle.default_params(N, 0)
    if mu0:
        po.ipm_mu0 = mu0
    fo0, io0, tuo0 = oracle.solve_warm(po, rec, None)
    fo1, io1, tuo1 = oracle.solve_warm(po, rec2, res[4][2])
    f0, i0, tu0, f1, i1, tu1 = res[4]
    assert (i0["status"] == 0).all() and (i1["status"] == 0).all()
    assert np.abs(f0 - fo0).max() < 1e-6 and np.abs(tu0.reshape(B, -1) - tuo0.reshape(B, -1)).max() < 1e-5
    assert np.abs(f1 - fo1).max() < 1e-6, np.abs(f1 - fo1).max()
    assert np.abs(tu1.reshape(B, -1) - tuo1.reshape(B, -1)).max() < 1e-5
    di = np.abs(i1["iterations"].astype(int) - io1["iterations"].astype(int))
    assert (di == 0).mean() >= 0.9 and (di <= 1).mean() >= 0.95, np.bincount(di)
    assert np.abs(f1 - res[0][3]).max() < 1e-6 and np.abs(tu1.reshape(B, -1) - res[0][5].reshape(B, -1)).max() < 1e-5
    assert np.abs(tu1.reshape(B, N, 12)[np.repeat(rec2["contacts"] == 0, 3, axis=1)[:, None, :].repeat(N, 1)]).max() == 0.0
    print(f"N={N} mu0={mu0}: lane warm vs oracle {np.abs(f1 - fo1).max():.2e} N, vs wave kernel {np.abs(f1 - res[0][3]).max():.2e} N; "
          f"iterations equal on {(di == 0).mean():.3f}")


def test_warm_started_closed_loop_with_the_lane_kernel(pkg, lib, monkeypatch):
    """The warm-started loop (qmpc_loop_params.warm_start, low initial barrier) on the lane kernel against the wave kernels:
    same robots, same ticks, the first tick of each call cold, a failed previous solve ignored (check_prev)."""
    lp = pkg.default_loop_params(lib)
    lp.warm_start = 1.0
    rng = np.random.default_rng(23)
    B = 3072
    cmds = np.zeros((B, 7))
    cmds[:, 0] = rng.uniform(-0.5, 0.5, B); cmds[:, 1] = rng.uniform(-0.2, 0.2, B); cmds[:, 2] = rng.uniform(0.26, 0.32, B)
    cmds[:, 5] = rng.uniform(-0.5, 0.5, B); cmds[:, 6] = (rng.random(B) < 0.9).astype(float)
    cmds[cmds[:, 6] == 0, :2] = 0.0
    cmds[cmds[:, 6] == 0, 5] = 0.0
    stand = cmds.copy(); stand[:, 6] = 0.0
    st0 = pkg.loop_states(stand, lp, height=0.3, yaw=rng.uniform(-3, 3, B), lib=lib)
    out = {}
    for v in (4, 0):
        _forced(monkeypatch, v)
        p = pkg.default_params(10, pkg.MODE_CONVERGED, lib)
        p.ipm_mu0 = 1e-6
        p.drop_ang_vel = 0
        s = pkg.Solver(p, B, device=0, lib=lib)
        st = s.loop_run(st0, 6, lp)
        st["movement_mode"] = cmds[:, 6]
        st = s.loop_run(st, 60, lp)
        s.close()
        out[v] = st
        assert (st["status"] == 0).all() and np.isfinite(st["pos_world"]).all()
    dp = np.abs(out[4]["pos_world"] - out[0]["pos_world"]).max()
    print(f"warm-started loop, lane vs wave, {B} robots, 66 ticks: max position difference {dp:.2e} m; "
          f"mean iterations {out[4]['iterations'].mean():.2f} / {out[0]['iterations'].mean():.2f}")
    assert dp < 1e-7 and abs(out[4]["iterations"].mean() - out[0]["iterations"].mean()) < 0.2


@pytest.mark.parametrize("N,B", [(20, 320), (10, 192)])
def test_lane_kernel_convex_model_matches_oracle(pkg, lib, oracle, monkeypatch, N, B):
    """ConvexMpc's problem (Euler-angle model, world-frame forces) on the lane kernel: the same transition shape as the
    quaternion model's error state, the inertia at the midpoint yaw in the per-point map -- against the oracle and the
    wave-per-instance kernel, with a rejected record and a record without contacts in the batch."""
    rec = pkg.random_go1_convex_states(B, config_id=12)
    rec["contacts"][5] = 0.0
    rec["euler"][9, 1] = np.nan
    res = {}
    for v in (4, 0):
        _forced(monkeypatch, v)
        s = pkg.Solver(pkg.default_convex_params(N, pkg.MODE_CONVERGED, lib), B, device=0, lib=lib)
        res[v] = s.convex_solve(rec)
        s.close()
    f, info = res[4]
    fo, io = oracle.convex_solve(oracle.default_convex_params(N, 0), rec, threads=8)
    assert np.array_equal(info["status"], io["status"]) and np.array_equal(info["status"], res[0][1]["status"])
    assert info["status"][5] == pkg.NO_CONTACT and info["status"][9] == pkg.NAN_INPUT
    ok = info["status"] == 0
    assert ok.sum() == B - 2
    assert np.abs(f - fo).max() < 1e-6 and np.abs(f - res[0][0]).max() < 1e-6
    di = np.abs(info["iterations"].astype(int) - io["iterations"].astype(int))
    assert (di == 0).mean() >= 0.95 and (di <= 1).mean() >= 0.97, np.bincount(di)
    assert np.abs(info["cost"][ok] - io["cost"][ok]).max() < 1e-9 * max(1.0, np.abs(io["cost"][ok]).max())
    assert np.abs(f[np.repeat(rec["contacts"] == 0, 3, axis=1)]).max() == 0.0
    print(f"convex N={N}: lane vs oracle {np.abs(f - fo).max():.2e} N, vs wave kernel {np.abs(f - res[0][0]).max():.2e} N, "
          f"iterations equal on {(di == 0).mean():.3f}")


def test_lane_kernel_8_point_reference_mode(pkg, lib, oracle, monkeypatch):
    """The 8-contact-point model in the reference's solver mode on the lane kernel (qmpc_lane_ref_kernel<8>): status words and
    iteration counts identical to the oracle's and to the wave-per-instance reference kernels', forces of the truncated
    iterates within 1e-6 N on >= 95 % of the instances."""
    N, B = 16, 768
    p = pkg.default_biped8_params(N, pkg.MODE_REFERENCE, lib)
    rec = pkg.random_biped8_states(B, config_id=5)
    rec["contacts"][5] = 0.0
    out = {}
    for v in (4, 0):
        _forced(monkeypatch, v)
        monkeypatch.setenv("QMPC_LANE_REF_MIN", str(1 << 30))
        s = pkg.Solver(p, B, device=0, lib=lib)
        out[v] = s.solve8(rec)
        if v == 4:
            assert pkg.KERNEL_FAMILY[s.query(pkg.QUERY_LAST_KERNEL)] == "lane"
        s.close()
    fl, il = out[4]
    fw, iw = out[0]
    fo, io = oracle.solve8(oracle.default_biped8_params(N, 1), rec, threads=8)
    assert np.array_equal(il["status"], io["status"]) and np.array_equal(il["iterations"], io["iterations"])
    assert np.array_equal(il["status"], iw["status"]) and np.array_equal(il["iterations"], iw["iterations"])
    assert il["status"][5] == pkg.NO_CONTACT
    d = np.abs(fl - fo).max(axis=1)
    dw = np.abs(fl - fw).max(axis=1)
    print(f"lane kernel, 8-point model, reference mode N={N} B={B}: vs oracle within 1e-6 N on {100 * (d < 1e-6).mean():.1f} % (median "
          f"{np.median(d):.1e}, worst {d.max():.1e}); vs wave kernels {100 * (dw < 1e-6).mean():.1f} %; status counts "
          f"{np.bincount(il['status'], minlength=6).tolist()}")
    assert (d < 1e-6).mean() >= 0.999 and (dw < 1e-6).mean() >= 0.999
    assert (il["iterations"] <= 10).all() and np.isfinite(fl).all()
    assert (fl.reshape(-1, 8, 3)[rec["contacts"] == 0] == 0).all()
    monkeypatch.delenv("QMPC_LANE_REF_MIN")
    s = pkg.Solver(p, 65536, device=0, lib=lib)
    assert s.kernel_for_batch(65536) == "lane" and s.kernel_for_batch(32768) == "wform_ws" and s.kernel_for_batch(64) == "wform_lds"
    s.close()


@pytest.mark.gpu
@pytest.mark.parametrize("N,B", [(20, 1024), (10, 2048)])
def test_lane_kernel_convex_reference_mode(pkg, lib, oracle, monkeypatch, N, B):
    """ConvexMpc's OWN solver mode (five AL-iLQR iterations, ConvexMpc.cpp:36-38) on the lane kernel
    (qmpc_lane_ref_kernel<4, MD_CONVEX>): against the oracle on every instance -- identical status words and iteration
    counts, forces of the truncated iterates within 1e-6 N on >= 95 % -- and against the wrench-form reference kernels; a
    record without contacts and a rejected record keep their status words; trajectories come back; the library reports the
    lane family for this mode at Monte-Carlo sizes."""
    p = pkg.default_convex_params(N, pkg.MODE_REFERENCE, lib)
    rec = pkg.random_go1_convex_states(B, config_id=12)
    rec["contacts"][5] = 0.0
    rec["euler"][9, 1] = np.nan
    out = {}
    for v in (4, 0):
        _forced(monkeypatch, v)
        monkeypatch.setenv("QMPC_LANE_REF_MIN", str(1 << 30))       # variant 0: the wave kernels at this size
        s = pkg.Solver(p, B, device=0, lib=lib)
        out[v] = s.convex_solve(rec)
        if v == 4:
            f2, info2 = s.convex_solve(rec)
            assert np.array_equal(out[4][0], f2) and np.array_equal(out[4][1]["iterations"], info2["iterations"])
            assert pkg.KERNEL_FAMILY[s.query(pkg.QUERY_LAST_KERNEL)] == "lane"
            ft, it, tu, tx = s.convex_solve(rec, want_traj=True)        # trajectories from the lane kernel too
            assert pkg.KERNEL_FAMILY[s.query(pkg.QUERY_LAST_KERNEL)] == "lane"
            assert np.array_equal(ft, f2) and np.array_equal(tu[:, 0, :], f2) and np.isfinite(tx).all()
            out["traj"] = (tu, tx)
        else:
            out["traj0"] = s.convex_solve(rec, want_traj=True)[2:]
        s.close()
    monkeypatch.delenv("QMPC_LANE_REF_MIN")
    _forced(monkeypatch, 0)
    s = pkg.Solver(p, 65536, device=0, lib=lib)
    assert s.kernel_for_batch(65536) == "lane" and s.kernel_for_batch(1024) != "lane"
    s.close()
    fl, il = out[4]
    fw, iw = out[0]
    fo, io = oracle.convex_solve(oracle.default_convex_params(N, 1), rec, threads=8)
    # state trajectories of the two kernel families (the AL passes keep their feedback gains in double precision since the end of
    # round 5: with the packed single-precision gains later knots of the lane kernel's trajectory were 1e-5 N / 2e-6 state
    # units from the oracle's at N = 20)
    dt = np.abs(out["traj"][1] - out["traj0"][1]).reshape(B, -1).max(axis=1)
    print(f"lane vs wave kernels, ConvexMpc reference mode N={N}: state trajectories median {np.median(dt):.1e}, worst {dt.max():.1e}")
    assert (dt < 1e-6).mean() >= 0.995 and np.median(dt) < 1e-9
    assert np.array_equal(il["status"], io["status"]) and np.array_equal(il["iterations"], io["iterations"])
    assert np.array_equal(il["status"], iw["status"]) and np.array_equal(il["iterations"], iw["iterations"])
    assert il["status"][5] == pkg.NO_CONTACT and il["status"][9] == pkg.NAN_INPUT
    d = np.abs(fl - fo).max(axis=1)
    dw = np.abs(fl - fw).max(axis=1)
    print(f"lane kernel, ConvexMpc reference mode N={N} B={B}: vs oracle within 1e-6 N on {100 * (d < 1e-6).mean():.1f} % (median "
          f"{np.median(d):.1e}, worst {d.max():.1e}); vs wave kernels {100 * (dw < 1e-6).mean():.1f} %; status counts "
          f"{np.bincount(il['status'], minlength=6).tolist()}")
    assert (d < 1e-6).mean() >= 0.99 and (dw < 1e-6).mean() >= 0.999
    assert (il["iterations"] <= 5).all() and np.isfinite(fl).all()
    assert (fl.reshape(-1, 4, 3)[rec["contacts"] == 0] == 0).all()
    solved = io["status"] <= 1
    assert np.abs(il["cost"][solved] - io["cost"][solved]).max() < 1e-6 * max(1.0, np.abs(io["cost"][solved]).max())


def test_convex_closed_loop_with_the_lane_kernel(pkg, lib, monkeypatch):
    """ConvexMpc's tick in the device-resident loop (per-tick form) with its solves on the lane kernel, cold and
    warm-started, against the wave-per-instance kernels."""
    rng = np.random.default_rng(29)
    B = 3072
    cmds = np.zeros((B, 7))
    cmds[:, 0] = 0.6 * rng.uniform(-0.5, 0.5, B); cmds[:, 1] = rng.uniform(-0.2, 0.2, B); cmds[:, 2] = rng.uniform(0.26, 0.32, B)
    cmds[:, 5] = rng.uniform(-0.5, 0.5, B); cmds[:, 6] = (rng.random(B) < 0.9).astype(float)
    cmds[cmds[:, 6] == 0, :2] = 0.0
    cmds[cmds[:, 6] == 0, 5] = 0.0
    stand = cmds.copy(); stand[:, 6] = 0.0
    for warm in (0.0, 1.0):
        lp = pkg.default_loop_params(lib)
        lp.warm_start = warm
        st0 = pkg.loop_states(stand, lp, height=0.3, yaw=rng.uniform(-3, 3, B), lib=lib)
        out = {}
        for v in (4, 0):
            _forced(monkeypatch, v)
            p = pkg.default_convex_params(10, pkg.MODE_CONVERGED, lib)
            if warm:
                p.ipm_mu0 = 1e-6
            s = pkg.Solver(p, B, device=0, lib=lib)
            st = s.loop_run(st0, 6, lp)
            st["movement_mode"] = cmds[:, 6]
            st = s.loop_run(st, 40, lp)
            s.close()
            out[v] = st
            assert (st["status"] == 0).all() and np.isfinite(st["pos_world"]).all()
        dp = np.abs(out[4]["pos_world"] - out[0]["pos_world"]).max()
        print(f"ConvexMpc loop, warm_start={warm}: lane vs wave max position difference {dp:.2e} m; "
              f"mean iterations {out[4]['iterations'].mean():.2f} / {out[0]['iterations'].mean():.2f}")
        assert dp < 1e-7 and np.array_equal(out[4]["contacts"], out[0]["contacts"])


def test_lane_kernel_warm_and_convex_beyond_the_resident_lanes(pkg, lib, oracle):
    """70000 instances (more than the 65536 resident lanes: a second round of some wavefronts) through the automatic
    selection: warm-started QuatMpc solves and ConvexMpc solves, each against the oracle on a sample and for
    batch-position independence (a shard solved alone gives the same bits)."""
    B, N = 70000, 10
    rec = pkg.random_go1_trot_states(B, config_id=4)
    p = pkg.default_params(N, pkg.MODE_CONVERGED, lib)
    s = pkg.Solver(p, B, device=0, lib=lib)
    f0, i0, tu0 = s.solve_warm(rec, None)
    rec2 = rec.copy()
    rec2["lin_vel_body"] += 0.01
    f1, i1, tu1 = s.solve_warm(rec2, tu0)
    assert (i0["status"] == 0).all() and (i1["status"] == 0).all()
    sub = np.arange(0, B, B // 48)[:48]
    fo, io, tuo = oracle.solve_warm(oracle.default_params(N, 0), rec2[sub], tu0[sub])
    assert np.abs(f1[sub] - fo).max() < 1e-6 and np.abs(tu1[sub].reshape(48, -1) - tuo.reshape(48, -1)).max() < 1e-5
    fs, _, _ = s.solve_warm(rec2[40000:], tu0[40000:])
    assert np.array_equal(fs, f1[40000:])
    s.close()
    recc = pkg.random_go1_convex_states(B, config_id=12)
    sc = pkg.Solver(pkg.default_convex_params(N, pkg.MODE_CONVERGED, lib), B, device=0, lib=lib)
    fc, ic = sc.convex_solve(recc)
    assert (ic["status"] == 0).all()
    fco, _ = oracle.convex_solve(oracle.default_convex_params(N, 0), recc[sub], threads=8)
    assert np.abs(fc[sub] - fco).max() < 1e-6
    fcs, _ = sc.convex_solve(recc[40000:])
    assert np.array_equal(fcs, fc[40000:])
    sc.close()


@pytest.mark.parametrize("model", ["quat", "convex", "biped8"])
def test_lane_kernel_trajectory_outputs(pkg, lib, oracle, monkeypatch, model):
    """qmpc_solve*_traj on the lane kernel: input and state trajectories against the oracle's and the wave kernels'."""
    B = 96
    gen, dp, N = {"quat": (pkg.random_go1_trot_states, pkg.default_params, 10),
                  "convex": (pkg.random_go1_convex_states, pkg.default_convex_params, 20),
                  "biped8": (pkg.random_biped8_states, pkg.default_biped8_params, 16)}[model]
    rec = gen(B, config_id={"quat": 2, "convex": 12, "biped8": 5}[model])
    res = {}
    for v in (4, 0):
        _forced(monkeypatch, v)
        s = pkg.Solver(dp(N, pkg.MODE_CONVERGED, lib), B, device=0, lib=lib)
        call = {"quat": s.solve, "convex": s.convex_solve, "biped8": s.solve8}[model]
        res[v] = call(rec, want_traj=True)
        s.close()
    osolve, odp = {"quat": (oracle.solve, oracle.default_params), "convex": (oracle.convex_solve, oracle.default_convex_params),
                   "biped8": (oracle.solve8, oracle.default_biped8_params)}[model]
    fo, io, tuo, txo = osolve(odp(N, 0), rec, threads=8, want_traj=True)
    f, info, tu, tx = res[4]
    assert (info["status"] == 0).all()
    assert np.abs(f - fo).max() < 1e-6
    assert np.abs(tu.reshape(B, -1) - tuo.reshape(B, -1)).max() < 1e-5 and np.abs(tx.reshape(B, -1) - txo.reshape(B, -1)).max() < 1e-8
    assert np.abs(tu.reshape(B, -1) - res[0][2].reshape(B, -1)).max() < 1e-5
    assert np.abs(tx.reshape(B, -1) - res[0][3].reshape(B, -1)).max() < 1e-8
    assert np.array_equal(tu.reshape(B, N, -1)[:, 0, :], f)


def test_lane_kernel_convex_and_warm_randomised_parameter_sets(pkg, lib, oracle, monkeypatch):
    """Random friction, force limit, weights, mass, inertia, knot spacing and horizon for ConvexMpc's problem on the lane
    kernel, and the same draw of QuatMpc parameters for warm-started solves (previous solution of slightly different
    states): the lane kernel follows the oracle for every set."""
    _forced(monkeypatch, 4)
    rng = np.random.default_rng(91)
    for t in range(6):
        N = int(rng.choice([6, 10, 20, 30]))
        p = pkg.default_convex_params(N, pkg.MODE_CONVERGED, lib)
        p.mu = float(rng.uniform(0.3, 1.0))
        p.fz_max = float(rng.uniform(80, 300))
        p.mass = float(rng.uniform(9, 16))
        for i in range(12):
            p.q_weights[i] = float(p.q_weights[i] * rng.uniform(0.3, 3.0))
            p.r_weights[i] = float(10 ** rng.uniform(-6.5, -4.5))
        for i in (0, 4, 8):
            p.inertia[i] = float(p.inertia[i] * rng.uniform(0.6, 1.6))
        hs = float(rng.choice([0.005, 0.01]))
        p.h, p.h_ref = hs, hs
        rec = pkg.random_go1_convex_states(128, config_id=60 + t)
        s = pkg.Solver(p, 128, device=0, lib=lib)
        f, info = s.convex_solve(rec)
        s.close()
        fo, io = oracle.convex_solve(p, rec, threads=8)
        assert np.array_equal(info["status"], io["status"]), (t, np.unique(info["status"]), np.unique(io["status"]))
        ok = info["status"] == 0
        assert ok.mean() > 0.9 and np.abs(f[ok] - fo[ok]).max() < 1e-6, (t, np.abs(f[ok] - fo[ok]).max())
        di = np.abs(info["iterations"][ok].astype(int) - io["iterations"][ok].astype(int))
        assert (di <= 1).mean() >= 0.95, (t, np.bincount(di))
    for t in range(4):
        p = _random_params(pkg, lib, rng, t)
        p.ipm_mu0 = float(rng.choice([1e-2, 1e-4, 1e-6]))
        rec = pkg.random_go1_trot_states(128, config_id=70 + t)
        rec2 = rec.copy()
        rec2["lin_vel_body"] += rng.normal(0, 0.02, rec2["lin_vel_body"].shape)
        s = pkg.Solver(p, 128, device=0, lib=lib)
        _, _, tu = s.solve_warm(rec, None)
        f, info, tu1 = s.solve_warm(rec2, tu)
        s.close()
        fo, io, tuo = oracle.solve_warm(p, rec2, tu)
        assert np.array_equal(info["status"], io["status"]), t
        ok = info["status"] == 0
        assert ok.mean() > 0.9 and np.abs(f[ok] - fo[ok]).max() < 1e-6, (t, np.abs(f[ok] - fo[ok]).max())
        di = np.abs(info["iterations"][ok].astype(int) - io["iterations"][ok].astype(int))
        assert (di <= 1).mean() >= 0.95, (t, np.bincount(di))


@pytest.mark.parametrize("N,B", [(10, 30000), (20, 17000)])
def test_reference_mode_lane_pairs_return_the_plain_forms_bits(pkg, lib, oracle, monkeypatch, N, B):
    """Round 6: reference-mode batches that fill half of every wavefront run the trial sweeps and the AL backward pass as lane
    PAIRS (a trial of the sweep per partner lane, a point of the diagonal pair per lane).  A shard solved that way returns the
    bits of its block of a full-wavefront launch (the plain form), status and iteration words included -- which took explicit
    fma chains in the per-point block: the compiler's choice of which product a sum fuses differed between the instantiations by
    an ulp of the increment.  And the words are the oracle's."""
    _forced(monkeypatch, 4)
    p = pkg.default_params(N, pkg.MODE_REFERENCE, lib)
    rec = pkg.random_go1_trot_states(70000, config_id=3 if N == 20 else 2)
    s = pkg.Solver(p, 70000, device=0, lib=lib)
    f, info = s.solve(rec[:B])                     # 32 instances per wavefront: pairs
    ff, fi = s.solve(rec)                          # full wavefronts: the plain form
    s.close()
    assert np.array_equal(f, ff[:B]) and np.array_equal(info["status"], fi["status"][:B]) and \
        np.array_equal(info["iterations"], fi["iterations"][:B])
    idx = np.arange(0, B, B // 96)
    fo, io = oracle.solve(p, rec[idx], threads=8)
    assert np.array_equal(io["status"], info["status"][idx]) and np.array_equal(io["iterations"], info["iterations"][idx])
    assert np.abs(fo - f[idx]).max() < 1e-6


@pytest.mark.parametrize("N,B", [(10, 17001), (20, 15003)])
def test_lane_pairs_return_the_plain_forms_bits(pkg, lib, oracle, monkeypatch, N, B):
    """Round 6: in pair mode the apply pass, the per-point blocks and step 5 of the backward pass are split between the partner
    lanes (the last by blocks of the cost-to-go matrix, with the partner's z swapped in), the per-instance constants and the knot's
    state travel through the LDS halves the pair mode leaves unused (global_load_lds).  A shard solved that way -- its last
    wavefront partly filled -- returns the bits of its block of a full-wavefront launch, iteration words included; the forces are
    the oracle's."""
    _forced(monkeypatch, 4)      # the pure lane kernel: no hand-off between the two forms
    p = pkg.default_params(N, 0, lib)
    rec = pkg.random_go1_trot_states(70000, config_id=3 if N == 20 else 4)
    s = pkg.Solver(p, 70000, device=0, lib=lib)
    f, info = s.solve(rec[:B])                     # 32 instances per wavefront: pairs
    ff, fi = s.solve(rec)                          # full wavefronts: the plain form
    s.close()
    assert (info["status"] == 0).all()
    assert np.array_equal(f, ff[:B]) and np.array_equal(info["iterations"], fi["iterations"][:B])
    idx = np.arange(0, B, B // 64)
    fo, io = oracle.solve(p, rec[idx], threads=8)
    assert np.abs(fo - f[idx]).max() < 1e-8
    assert (io["iterations"] == info["iterations"][idx]).mean() >= 0.98
