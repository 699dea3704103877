"""CPU suite: the C-ABI library loads, exports every symbol include/qmpc.h
declares, agrees on record sizes, and FAILS LOUDLY without a GPU."""
import ctypes as C
import re
from pathlib import Path

import numpy as np
import pytest

REPO = Path(__file__).resolve().parent.parent


@pytest.fixture(scope="module")
def lib(pkg):
    import __graft_entry__ as g

    g.build_hip()
    return pkg.load_library()


def test_exports_every_declared_symbol(lib, pkg):
    header = (REPO / "include" / "qmpc.h").read_text()
    declared = set(re.findall(r"\b(qmpc_[a-z_0-9]+)\s*\(", header))
    assert declared, "no declarations parsed"
    assert declared == set(pkg.EXPORTED_SYMBOLS)
    for name in declared:
        assert hasattr(lib, name), name


def test_record_sizes(lib, pkg):
    assert lib.qmpc_sizeof_input() == 384 == pkg.INPUT_DTYPE.itemsize
    assert lib.qmpc_sizeof_info() == 40 == pkg.INFO_DTYPE.itemsize
    assert lib.qmpc_sizeof_params() == C.sizeof(pkg.Params)


def test_default_params_agree_with_oracle(lib, pkg, oracle):
    for N in (10, 20):
        a = pkg.default_params(N, pkg.MODE_CONVERGED, lib)
        b = oracle.default_params(N, 0)
        assert bytes(a) == bytes(b)
    a = pkg.default_params(20, pkg.MODE_REFERENCE, lib)
    b = oracle.default_params(20, 1)
    for f, _ in pkg.Params._fields_:
        va, vb = getattr(a, f), getattr(b, f)
        if hasattr(va, "__len__"):
            assert list(va) == list(vb), f
        elif f not in ("tol_step", "ipm_mu_final", "ipm_sigma", "ipm_sigma_fast", "ipm_tau", "tol_feasibility",
                       "penalty_initial"):
            assert va == vb, f


def test_status_strings(lib):
    assert lib.qmpc_status_string(0) == b"ok"
    assert b"no CPU fallback" in lib.qmpc_status_string(17)
    assert lib.qmpc_version().startswith(b"qmpc-hip")


def test_bad_arguments_are_rejected(lib, pkg):
    h = C.c_void_p()
    assert lib.qmpc_create(None, 4, 0, C.byref(h)) == pkg.BAD_ARGUMENT
    p = pkg.default_params(10, 0, lib)
    assert lib.qmpc_create(C.byref(p), 0, 0, C.byref(h)) == pkg.BAD_ARGUMENT
    p.horizon = 99
    assert lib.qmpc_create(C.byref(p), 4, 0, C.byref(h)) == pkg.BAD_ARGUMENT


def test_round5_entry_points_reject_null_handles(lib, pkg):
    """qmpc_prepare / qmpc_query without a handle (no device needed): BAD_ARGUMENT, nothing dereferenced; qmpc_host_free(NULL)
    is a no-op; qmpc_host_alloc(0) returns NULL."""
    v = C.c_int64(7)
    assert lib.qmpc_prepare(None, 16) == pkg.BAD_ARGUMENT
    assert lib.qmpc_query(None, pkg.QUERY_KERNEL_FOR_BATCH, 16, C.byref(v)) == pkg.BAD_ARGUMENT and v.value == 7
    lib.qmpc_host_free(None)
    assert not lib.qmpc_host_alloc(0)


def test_fails_loudly_without_gpu(lib, pkg):
    """No silent CPU path: without a HIP device creation must return NO_DEVICE."""
    import torch

    if torch.cuda.is_available():
        pytest.skip("a GPU is visible here")
    p = pkg.default_params(10, 0, lib)
    with pytest.raises(pkg.QmpcError) as e:
        pkg.Solver(p, 4, device=0, lib=lib)
    assert e.value.code == pkg.NO_DEVICE
    out = np.zeros(192)
    assert lib.qmpc_selftest_mtm(0, out.ctypes.data, out.ctypes.data, out.ctypes.data) == pkg.NO_DEVICE


def test_product_does_not_touch_the_oracle():
    """The shipped path may not import / link anything under oracle/."""
    for f in (REPO / "quaternion-mpc_amd").rglob("*"):
        if f.suffix in (".py", ".hip", ".h", ".hpp", ".cpp"):
            txt = f.read_text()
            assert "pyoracle" not in txt and "qo_" not in txt and "libqmpc_oracle" not in txt, f
