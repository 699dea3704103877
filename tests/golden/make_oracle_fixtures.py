#!/usr/bin/env python3
"""Writes tests/golden/oracle_regimes.npz: forces of the CPU oracle (converged mode) on small seeded samples of the
regimes no reference golden covers (random attitudes, trot contacts, active friction / force limits,
N = 10 and 20, YAML weights; the ConvexMpc model; 8 contact points).  The fixture freezes the oracle's answers so
that a later change to oracle/ (or to the state generators) cannot go unnoticed; tests/test_oracle_golden.py checks
the oracle against it on the CPU and tests/test_gpu_parity.py checks the HIP path against it on the GPU.
Run from the repo root:  python tests/golden/make_oracle_fixtures.py"""
import importlib
import sys
from pathlib import Path

import numpy as np

REPO = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(REPO))
sys.path.insert(0, str(REPO / "tests"))
from conftest import load_pkg  # noqa: E402

pkg = load_pkg()
from oracle import pyoracle as po  # noqa: E402

CASES = (   # name, generator, params, solver, horizon, config_id, instances
    ("quat_n10", "random_go1_trot_states", "default_params", "solve", 10, 2, 48),
    ("quat_n20", "random_go1_trot_states", "default_params", "solve", 20, 3, 32),
    ("convex_n20", "random_go1_convex_states", "default_convex_params", "convex_solve", 20, 13, 32),
    ("biped8_n16", "random_biped8_states", "default_biped8_params", "solve8", 16, 5, 24),
)

if __name__ == "__main__":
    out = {}
    for name, gen, dp, solve, N, cfg, n in CASES:
        rec = getattr(pkg, gen)(n, config_id=cfg)
        f, info = getattr(po, solve)(getattr(po, dp)(N, 0), rec, threads=4)
        assert (info["status"] == 0).all(), name
        out[name + "_forces"] = f
        out[name + "_iterations"] = info["iterations"].astype(np.int32)
    np.savez_compressed(Path(__file__).parent / "oracle_regimes.npz", **out)
    print({k: v.shape for k, v in out.items()})
