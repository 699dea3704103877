"""pytest configuration: markers, package loading, shared fixtures."""
from __future__ import annotations

import importlib.util
import sys
from pathlib import Path

import numpy as np
import pytest

REPO = Path(__file__).resolve().parent.parent
if str(REPO) not in sys.path:
    sys.path.insert(0, str(REPO))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_pkg():
    """Import quaternion-mpc_amd/ (hyphen in the name) as `quaternion_mpc_amd`."""
    name = "quaternion_mpc_amd"
    if name in sys.modules:
        return sys.modules[name]
    spec = importlib.util.spec_from_file_location(
        name, REPO / "quaternion-mpc_amd" / "__init__.py",
        submodule_search_locations=[str(REPO / "quaternion-mpc_amd")])
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod


@pytest.fixture(scope="session")
def pkg():
    return load_pkg()


@pytest.fixture(scope="session")
def oracle():
    """The CPU oracle (test infrastructure).  Built on demand with gcc."""
    from oracle import pyoracle

    pyoracle.lib()
    return pyoracle


TRUNK_INERTIA = (0.0168128557, 0.063009565, 0.0716547275)


def golden_problem(pkg, params, which: str):
    """Parameters + input record of the reference's two golden generators.

    which == "stand": legged_ctrl/src/test/test_altro/TestAltroQuatMpc.cpp:36-113
    which == "trot" : legged_ctrl/src/test/test_altro/TestAltroTrotQuatMpc.cpp:32-95
    """
    p = params
    for i in range(9):
        p.inertia[i] = 0.0
    for a in range(3):
        p.inertia[4 * a] = (12.84 / 5.204) * TRUNK_INERTIA[a]
    p.mass = 12.84
    p.fz_max = 200.0
    if which == "stand":
        q = [1, 1, 1, 0, 0, 0, 0, 2, 2, 2, 1, 1, 1]
        p.w, p.mu = 1.0, 0.6
        feet = np.array([[0.2104, 0.13, -0.325], [0.2104, -0.13, -0.325],
                         [-0.1658, 0.13, -0.325], [-0.1658, -0.13, -0.325]])
        rec = pkg.go1_stand_input(feet)
        cols = list(range(12))
    else:
        q = [1, 1, 1, 0, 0, 0, 0, 10, 10, 10, 10, 10, 10]
        p.w, p.mu = 10.0, 0.7
        # the generator has 2 feet (FL, RR); the two swing legs carry no force
        feet = np.array([[0.17, 0.13, -0.3], [0.2, -0.14, -0.3], [-0.2, 0.14, -0.3], [-0.17, -0.13, -0.3]])
        rec = pkg.go1_stand_input(feet)
        rec["contacts"][0] = [1, 0, 0, 1]
        rec["acc_ref_body"][0] = [0.5, 0.0, 0.0]   # x_ref = (t^2/4, 0,0 | 1,0,0,0 | t/2, 0,0 | 0)
        cols = [0, 1, 2, 9, 10, 11]
    for i in range(13):
        p.q_weights[i] = q[i]
    return p, rec, cols
