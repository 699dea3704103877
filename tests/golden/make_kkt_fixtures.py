#!/usr/bin/env python3
"""Writes tests/golden/kkt_fixtures.npz: an algorithm-INDEPENDENT certificate for the regimes no reference golden
covers (tilted attitudes, trot contacts, active friction / force limits -- every instance the benchmark times).

For 64 seeded instances each of BASELINE config 2 (Go1, N=10), config 3 (Go1, N=20), ConvexMpc N=20 and the
8-contact-point model N=16 it stores the oracle's primal-dual point (U, lambda) -- multipliers through the oracle-only
call qo_solve_one_dual -- and, for a handful of instances per case, the solution found by a solver that shares
nothing with the oracle: tests/kkt_independent.py's primal active-set Newton method on finite-difference Hessians of
a torch-autograd gradient (plus scipy SLSQP's best cost).  tests/test_kkt_certificate.py then re-evaluates
stationarity / complementarity / feasibility of the stored points with kkt_independent (no oracle code involved),
re-derives the multipliers by non-negative least squares, and holds today's oracle -- and, on the GPU, the HIP
path -- to the stored points.

BUILD CONTAINER ONLY (imports the oracle; takes ~15 min):  python tests/golden/make_kkt_fixtures.py"""
import sys
import time
from pathlib import Path

import numpy as np

REPO = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(REPO))
sys.path.insert(0, str(REPO / "tests"))
from conftest import load_pkg  # noqa: E402

pkg = load_pkg()
from oracle import pyoracle as po  # noqa: E402
import kkt_independent as K  # noqa: E402

# name, generator, params, model, problem class, horizon, config_id, instances, independently solved instances
CASES = (
    ("quat_n10", "random_go1_trot_states", "default_params", "quat", "QuatProblem", 10, 2, 64, 8),
    ("quat_n20", "random_go1_trot_states", "default_params", "quat", "QuatProblem", 20, 3, 64, 2),
    ("convex_n20", "random_go1_convex_states", "default_convex_params", "convex", "ConvexProblem", 20, 13, 64, 2),
    ("biped8_n16", "random_biped8_states", "default_biped8_params", "biped8", "QuatProblem", 16, 5, 64, 2),
)

if __name__ == "__main__":
    out = {}
    for name, gen, dp, model, cls, N, cfg, n, n_ind in CASES:
        rec = getattr(pkg, gen)(n, config_id=cfg)
        par = getattr(po, dp)(N, 0)
        U, LAM, IT = [], [], []
        for i in range(n):
            tu, _, lam, _, info = po.solve_dual(par, rec[i:i + 1], model)
            assert info["status"] == 0, (name, i)
            U.append(tu); LAM.append(lam); IT.append(info["iterations"])
        out[name + "_U"] = np.array(U)
        out[name + "_lam"] = np.array(LAM)
        out[name + "_iterations"] = np.array(IT, dtype=np.int32)
        Ui, cost_gap = [], []
        for i in range(n_ind):
            prob = getattr(K, cls)(par, rec[i])
            t = time.time()
            Ua, W, _, it = K.active_set_newton(prob)
            d = np.abs(Ua - U[i]).max()
            print(f"{name}[{i}] active-set Newton: {it} iterations, |W| = {len(W)}, max|U - U_oracle| = {d:.2e} N "
                  f"({time.time() - t:.0f} s)", flush=True)
            Ui.append(Ua)
            if name == "quat_n10":
                _, res = K.scipy_solve(prob)
                gap = res.fun - prob.value_and_grad(U[i])[0]
                cost_gap.append(gap)
                print(f"          scipy SLSQP: status {res.status}, cost - cost_oracle = {gap:.2e}", flush=True)
        out[name + "_U_independent"] = np.array(Ui)
        if cost_gap:
            out[name + "_slsqp_cost_gap"] = np.array(cost_gap)
    np.savez_compressed(Path(__file__).parent / "kkt_fixtures.npz", **out)
    print({k: v.shape for k, v in out.items()})
