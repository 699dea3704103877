"""quaternion-mpc_amd -- MI355X-native batched quaternion-MPC inner loop.

Python is plumbing only (tests, bench, multi-GPU launch): the product is the
C-ABI shared library ``csrc/libqmpc_hip.so`` declared in ``include/qmpc.h`` and
the C++ host class in ``host/``.  This module mirrors the C records with ctypes
and loads the library.  There is NO CPU fallback: if the HIP library is
missing or no GPU is visible, calls fail loudly.

The directory name contains a hyphen, so load it with
``importlib`` (see ``tests/conftest.py`` / ``__graft_entry__.py``) under the
module name ``quaternion_mpc_amd``.
"""
from __future__ import annotations

import ctypes as C
import os
from pathlib import Path

import numpy as np

PKG_DIR = Path(__file__).resolve().parent
REPO_DIR = PKG_DIR.parent
LIB_PATH = PKG_DIR / "csrc" / "libqmpc_hip.so"

NX, NE, NU, NLEG, NC = 13, 12, 12, 4, 24

# status codes (include/qmpc.h)
OK, MAX_ITER, NO_CONTACT, NAN_INPUT, LINESEARCH_FAIL, NOT_PD = 0, 1, 2, 3, 4, 5
BAD_ARGUMENT, NO_DEVICE, HIP_ERROR, BATCH_TOO_LARGE, UNSUPPORTED = 16, 17, 18, 19, 20
MODE_CONVERGED, MODE_REFERENCE = 0, 1
MODEL_QUAT, MODEL_CONVEX, MODEL_QUAT8 = 0, 1, 2


class Params(C.Structure):
    """struct qmpc_params (include/qmpc.h)."""

    _fields_ = [
        ("horizon", C.c_int32),
        ("h", C.c_float),
        ("h_ref", C.c_double),
        ("mass", C.c_double),
        ("inertia", C.c_double * 9),
        ("q_weights", C.c_double * 13),
        ("r_weights", C.c_double * 12),
        ("w", C.c_double),
        ("mu", C.c_double),
        ("fz_max", C.c_double),
        ("mode", C.c_int32),
        ("iterations_max", C.c_int32),
        ("penalty_initial", C.c_double),
        ("penalty_scaling", C.c_double),
        ("penalty_max", C.c_double),
        ("tol_stationarity", C.c_double),
        ("tol_feasibility", C.c_double),
        ("tol_cost_intermediate", C.c_double),
        ("tol_step", C.c_double),
        ("ipm_mu0", C.c_double),
        ("ipm_mu_final", C.c_double),
        ("ipm_sigma", C.c_double),
        ("ipm_sigma_fast", C.c_double),
        ("ipm_tau", C.c_double),
        ("linesearch_max", C.c_int32),
        ("drop_ang_vel", C.c_int32),
        ("model", C.c_int32),
    ]

    def copy(self) -> "Params":
        out = Params()
        C.memmove(C.byref(out), C.byref(self), C.sizeof(Params))
        return out


class LegGeometry(C.Structure):
    """struct qmpc_leg_geometry: the reference's rho_fix / rho_opt per leg (BaseInterface.cpp:10-34)."""

    _fields_ = [("rho_fix", (C.c_double * 5) * 4), ("rho_opt", (C.c_double * 3) * 4)]


# struct qmpc_input as a numpy structured dtype: 48 doubles, 384 B
INPUT_DTYPE = np.dtype(
    [
        ("quat", "<f8", (4,)),
        ("rot", "<f8", (9,)),
        ("lin_vel_body", "<f8", (3,)),
        ("ang_vel_body", "<f8", (3,)),
        ("foot_pos_body", "<f8", (12,)),
        ("contacts", "<f8", (4,)),
        ("pos_ref_body", "<f8", (3,)),
        ("vel_ref_body", "<f8", (3,)),
        ("acc_ref_body", "<f8", (3,)),
        ("quat_d", "<f8", (4,)),
    ],
    align=False,
)
assert INPUT_DTYPE.itemsize == 48 * 8

# struct qmpc_input8 (8 contact points, BASELINE config 5): 64 doubles, 512 B
INPUT8_DTYPE = np.dtype(
    [
        ("quat", "<f8", (4,)),
        ("rot", "<f8", (9,)),
        ("lin_vel_body", "<f8", (3,)),
        ("ang_vel_body", "<f8", (3,)),
        ("foot_pos_body", "<f8", (24,)),
        ("contacts", "<f8", (8,)),
        ("pos_ref_body", "<f8", (3,)),
        ("vel_ref_body", "<f8", (3,)),
        ("acc_ref_body", "<f8", (3,)),
        ("quat_d", "<f8", (4,)),
    ],
    align=False,
)
assert INPUT8_DTYPE.itemsize == 64 * 8

# struct qmpc_convex_input (ConvexMpc.cpp:81-198): 48 doubles, 384 B
CONVEX_INPUT_DTYPE = np.dtype(
    [
        ("euler", "<f8", (3,)),
        ("pos_world", "<f8", (3,)),
        ("ang_vel_world", "<f8", (3,)),
        ("lin_vel_world", "<f8", (3,)),
        ("foot_pos_abs_com", "<f8", (12,)),
        ("contacts", "<f8", (4,)),
        ("pos_d_world", "<f8", (3,)),
        ("lin_vel_d_world", "<f8", (3,)),
        ("yaw_rate_d", "<f8"),
        ("reserved", "<f8", (13,)),
    ],
    align=False,
)
assert CONVEX_INPUT_DTYPE.itemsize == 48 * 8

# ---- device-resident closed loop (include/qmpc.h: qmpc_loop_*), 820 doubles per instance ----
LOOP_WINDOW = 100
LOOP_FILTER_DTYPE = np.dtype([("ring", "<f8", (LOOP_WINDOW,)), ("head", "<f8"), ("count", "<f8"), ("sum", "<f8"),
                              ("correction", "<f8")], align=False)
LOOP_LEG_DTYPE = np.dtype([("gait_phase", "<f8"), ("state", "<f8"), ("pattern_index", "<f8"), ("prev_pattern_index", "<f8"),
                           ("start_time", "<f8"), ("end_time", "<f8"), ("not_first_call", "<f8"),
                           ("swing_start", "<f8", (3,)), ("swing_end", "<f8", (3,)), ("swing_extend", "<f8", (3,)),
                           ("fsm_pos", "<f8", (3,)), ("fsm_vel", "<f8", (3,)), ("fsm_acc", "<f8", (3,)),
                           ("terrain_height", "<f8")], align=False)
LOOP_STATE_DTYPE = np.dtype([
    ("pos_world", "<f8", (3,)), ("quat", "<f8", (4,)), ("lin_vel_world", "<f8", (3,)), ("ang_vel_body", "<f8", (3,)),
    ("foot_pos_world", "<f8", (12,)),
    ("joy", "<f8", (6,)), ("movement_mode", "<f8"), ("sin_ang_vel", "<f8"), ("attitude_traj_count", "<f8"),
    ("pos_d_world", "<f8", (3,)), ("pos_d_init", "<f8"), ("quat_d", "<f8", (4,)), ("lin_vel_d_rel", "<f8", (3,)),
    ("vel_filter", LOOP_FILTER_DTYPE, (3,)), ("pos_filter", LOOP_FILTER_DTYPE, (3,)),
    ("leg", LOOP_LEG_DTYPE, (4,)),
    ("contacts", "<f8", (4,)), ("gait_counter", "<f8", (4,)), ("forces_body", "<f8", (12,)), ("grf_world", "<f8", (12,)),
    ("foot_target_world", "<f8", (12,)), ("status", "<f8"), ("iterations", "<f8"), ("tick", "<f8")], align=False)
assert LOOP_STATE_DTYPE.itemsize == 820 * 8


# BaseInterface::tau_ctrl_update records (include/qmpc.h: qmpc_joint_feedback / qmpc_joint_command)
JOINT_FEEDBACK_DTYPE = np.dtype([
    ("joint_pos", "<f8", (12,)), ("joint_vel", "<f8", (12,)), ("torso_pos_world", "<f8", (3,)), ("torso_quat", "<f8", (4,)),
    ("torso_lin_vel_world", "<f8", (3,)), ("foot_pos_target_world", "<f8", (12,)), ("foot_vel_target_world", "<f8", (12,)),
    ("forces_body", "<f8", (12,)), ("plan_contacts", "<f8", (4,)), ("movement_mode", "<f8")])
JOINT_COMMAND_DTYPE = np.dtype([("joint_ang_tgt", "<f8", (12,)), ("joint_vel_tgt", "<f8", (12,)),
                                ("joint_tau_tgt", "<f8", (12,))])
assert JOINT_FEEDBACK_DTYPE.itemsize == 75 * 8 and JOINT_COMMAND_DTYPE.itemsize == 36 * 8


class LoopParams(C.Structure):
    """struct qmpc_loop_params."""

    _fields_ = [("gait_freq", C.c_double), ("default_foot_pos_rel", C.c_double * 12), ("dt", C.c_double),
                ("contact_height", C.c_double), ("warm_start", C.c_double)]


# struct qmpc_info: 2 x int32 + 4 doubles = 40 B
INFO_DTYPE = np.dtype(
    [
        ("status", "<i4"),
        ("iterations", "<i4"),
        ("cost", "<f8"),
        ("max_violation", "<f8"),
        ("last_step", "<f8"),
        ("penalty", "<f8"),
    ],
    align=False,
)
assert INFO_DTYPE.itemsize == 40


def _ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


class QmpcError(RuntimeError):
    def __init__(self, code: int, what: str):
        super().__init__(f"{what}: qmpc_status {code}")
        self.code = code


def load_library(path: os.PathLike | None = None) -> C.CDLL:
    """dlopen the HIP library.  Raises (never falls back) when it is absent."""
    # QMPC_LIB: a diagnostic build of the same library (tools/.prof/, phase counters compiled in)
    p = Path(path) if path else Path(os.environ.get("QMPC_LIB") or LIB_PATH)
    # torch bundles its own libamdhip64.so.7; whichever HIP runtime is loaded
    # first serves the whole process.  Let torch (our device-memory / stream /
    # RCCL plumbing) load its runtime first so both sides share one.
    try:
        import torch

        if torch.cuda.is_available():
            torch.cuda.init()
    except ImportError:
        pass
    if not p.exists():
        raise FileNotFoundError(
            f"{p} not built: run `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950).  There is no CPU fallback."
        )
    lib = C.CDLL(str(p))
    vp, i32 = C.c_void_p, C.c_int32
    lib.qmpc_default_params.argtypes = [C.POINTER(Params), i32, i32]
    lib.qmpc_default_params.restype = None
    lib.qmpc_create.argtypes = [C.POINTER(Params), i32, i32, C.POINTER(vp)]
    lib.qmpc_create.restype = i32
    lib.qmpc_set_params.argtypes = [vp, C.POINTER(Params)]
    lib.qmpc_set_params.restype = i32
    lib.qmpc_destroy.argtypes = [vp]
    lib.qmpc_destroy.restype = None
    lib.qmpc_solve.argtypes = [vp, i32, vp, vp, vp]
    lib.qmpc_solve.restype = i32
    lib.qmpc_solve_traj.argtypes = [vp, i32, vp, vp, vp, vp, vp]
    lib.qmpc_solve_traj.restype = i32
    lib.qmpc_solve_device.argtypes = [vp, i32, vp, vp, vp, vp]
    lib.qmpc_solve_device.restype = i32
    lib.qmpc_wait.argtypes = [vp]
    lib.qmpc_wait.restype = i32
    lib.qmpc_solve_async.argtypes = [vp, i32, vp, vp, vp]
    lib.qmpc_solve_async.restype = i32
    lib.qmpc_host_alloc.argtypes = [C.c_size_t]
    lib.qmpc_host_alloc.restype = vp
    lib.qmpc_host_free.argtypes = [vp]
    lib.qmpc_host_free.restype = None
    lib.qmpc_prepare.argtypes = [vp, i32]
    lib.qmpc_prepare.restype = i32
    lib.qmpc_query.argtypes = [vp, i32, C.c_int64, C.POINTER(C.c_int64)]
    lib.qmpc_query.restype = i32
    lib.qmpc_gather.argtypes = [vp, vp, vp, C.c_int64, vp, vp]
    lib.qmpc_gather.restype = i32
    lib.qmpc_last_kernel_ms.argtypes = [vp, C.POINTER(C.c_float)]
    lib.qmpc_last_kernel_ms.restype = i32
    lib.qmpc_linearize.argtypes = [vp, i32, vp, vp, vp, vp]
    lib.qmpc_linearize.restype = i32
    lib.qmpc_debug_profile.argtypes = [vp, i32, vp, vp]
    lib.qmpc_debug_profile.restype = i32
    lib.qmpc_selftest_lanes.argtypes = [i32, vp, vp]
    lib.qmpc_selftest_lanes.restype = i32
    lib.qmpc_selftest_mtm.argtypes = [i32, vp, vp, vp]
    lib.qmpc_selftest_mtm.restype = i32
    lib.qmpc_status_string.argtypes = [i32]
    lib.qmpc_status_string.restype = C.c_char_p
    lib.qmpc_version.argtypes = []
    lib.qmpc_version.restype = C.c_char_p
    lib.qmpc_default_convex_params.argtypes = [C.POINTER(Params), i32, i32]
    lib.qmpc_default_convex_params.restype = None
    lib.qmpc_convex_solve.argtypes = [vp, i32, vp, vp, vp]
    lib.qmpc_convex_solve.restype = i32
    lib.qmpc_convex_solve_traj.argtypes = [vp, i32, vp, vp, vp, vp, vp]
    lib.qmpc_convex_solve_traj.restype = i32
    lib.qmpc_convex_solve_device.argtypes = [vp, i32, vp, vp, vp, vp]
    lib.qmpc_convex_solve_device.restype = i32
    lib.qmpc_convex_linearize.argtypes = [vp, i32, vp, vp, vp, vp]
    lib.qmpc_convex_linearize.restype = i32
    lib.qmpc_default_go1_geometry.argtypes = [C.POINTER(LegGeometry)]
    lib.qmpc_default_go1_geometry.restype = None
    lib.qmpc_leg_kinematics.argtypes = [vp, C.POINTER(LegGeometry), i32, vp, vp, vp]
    lib.qmpc_leg_kinematics.restype = i32
    lib.qmpc_torque_map.argtypes = [vp, C.POINTER(LegGeometry), i32, vp, vp, vp, i32, vp]
    lib.qmpc_torque_map.restype = i32
    lib.qmpc_torque_map_device.argtypes = [vp, C.POINTER(LegGeometry), i32, vp, vp, vp, i32, vp, vp]
    lib.qmpc_torque_map_device.restype = i32
    lib.qmpc_leg_inverse_kinematics.argtypes = [vp, C.POINTER(LegGeometry), i32, vp, vp, vp]
    lib.qmpc_leg_inverse_kinematics.restype = i32
    lib.qmpc_joint_commands.argtypes = [vp, C.POINTER(LegGeometry), i32, vp, vp]
    lib.qmpc_joint_commands.restype = i32
    lib.qmpc_joint_commands_device.argtypes = [vp, C.POINTER(LegGeometry), i32, vp, vp, vp]
    lib.qmpc_joint_commands_device.restype = i32
    lib.qmpc_loop_joint_commands_device.argtypes = [vp, C.POINTER(LegGeometry), i32, vp, vp, vp, vp, vp]
    lib.qmpc_loop_joint_commands_device.restype = i32
    lib.qmpc_loop_run_joint_device.argtypes = [vp, C.POINTER(LoopParams), C.POINTER(LegGeometry), i32, vp, vp, i32, vp, vp, vp]
    lib.qmpc_loop_run_joint_device.restype = i32
    lib.qmpc_loop_joint_commands.argtypes = [vp, C.POINTER(LegGeometry), i32, vp, vp, vp, vp]
    lib.qmpc_loop_joint_commands.restype = i32
    lib.qmpc_solve_warm.argtypes = [vp, i32, vp, vp, vp, vp, vp]
    lib.qmpc_solve_warm.restype = i32
    lib.qmpc_solve_warm_device.argtypes = [vp, i32, vp, vp, vp, vp, vp, vp]
    lib.qmpc_solve_warm_device.restype = i32
    lib.qmpc_loop_joint_init.argtypes = [vp, i32]
    lib.qmpc_loop_joint_init.restype = None
    lib.qmpc_default_biped8_params.argtypes = [C.POINTER(Params), i32, i32]
    lib.qmpc_default_biped8_params.restype = None
    lib.qmpc_solve8.argtypes = [vp, i32, vp, vp, vp]
    lib.qmpc_solve8.restype = i32
    lib.qmpc_solve8_traj.argtypes = [vp, i32, vp, vp, vp, vp, vp]
    lib.qmpc_solve8_traj.restype = i32
    lib.qmpc_solve8_device.argtypes = [vp, i32, vp, vp, vp, vp]
    lib.qmpc_solve8_device.restype = i32
    lib.qmpc_default_loop_params.argtypes = [C.POINTER(LoopParams)]
    lib.qmpc_default_loop_params.restype = None
    lib.qmpc_loop_state_init.argtypes = [vp, C.POINTER(LoopParams), vp, C.c_double, C.c_double, C.c_double]
    lib.qmpc_loop_state_init.restype = None
    lib.qmpc_loop_run.argtypes = [vp, C.POINTER(LoopParams), i32, vp, i32, vp, vp]
    lib.qmpc_loop_run.restype = i32
    lib.qmpc_loop_run_device.argtypes = [vp, C.POINTER(LoopParams), i32, vp, i32, vp, vp, vp]
    lib.qmpc_loop_run_device.restype = i32
    for name in ("qmpc_sizeof_input", "qmpc_sizeof_params", "qmpc_sizeof_info", "qmpc_sizeof_convex_input",
                 "qmpc_sizeof_input8", "qmpc_sizeof_loop_state"):
        getattr(lib, name).argtypes = []
        getattr(lib, name).restype = i32
    if lib.qmpc_sizeof_input() != INPUT_DTYPE.itemsize:
        raise RuntimeError("qmpc_input ABI size mismatch")
    if lib.qmpc_sizeof_convex_input() != CONVEX_INPUT_DTYPE.itemsize:
        raise RuntimeError("qmpc_convex_input ABI size mismatch")
    if lib.qmpc_sizeof_input8() != INPUT8_DTYPE.itemsize:
        raise RuntimeError("qmpc_input8 ABI size mismatch")
    if lib.qmpc_sizeof_params() != C.sizeof(Params):
        raise RuntimeError("qmpc_params ABI size mismatch")
    if lib.qmpc_sizeof_info() != INFO_DTYPE.itemsize:
        raise RuntimeError("qmpc_info ABI size mismatch")
    if lib.qmpc_sizeof_loop_state() != LOOP_STATE_DTYPE.itemsize:
        raise RuntimeError("qmpc_loop_state ABI size mismatch")
    return lib


EXPORTED_SYMBOLS = (
    "qmpc_default_params",
    "qmpc_create",
    "qmpc_set_params",
    "qmpc_destroy",
    "qmpc_solve",
    "qmpc_solve_traj",
    "qmpc_solve_device",
    "qmpc_wait",
    "qmpc_solve_async",
    "qmpc_gather",
    "qmpc_last_kernel_ms",
    "qmpc_linearize",
    "qmpc_selftest_mtm",
    "qmpc_selftest_lanes",
    "qmpc_debug_profile",
    "qmpc_status_string",
    "qmpc_version",
    "qmpc_sizeof_input",
    "qmpc_sizeof_params",
    "qmpc_sizeof_info",
    "qmpc_default_convex_params",
    "qmpc_convex_solve",
    "qmpc_convex_solve_traj",
    "qmpc_convex_solve_device",
    "qmpc_convex_linearize",
    "qmpc_sizeof_convex_input",
    "qmpc_default_biped8_params",
    "qmpc_solve8",
    "qmpc_solve8_traj",
    "qmpc_solve8_device",
    "qmpc_sizeof_input8",
    "qmpc_default_go1_geometry",
    "qmpc_leg_kinematics",
    "qmpc_torque_map",
    "qmpc_torque_map_device",
    "qmpc_default_loop_params",
    "qmpc_loop_state_init",
    "qmpc_loop_run",
    "qmpc_loop_run_device",
    "qmpc_sizeof_loop_state",
    "qmpc_leg_inverse_kinematics",
    "qmpc_joint_commands",
    "qmpc_joint_commands_device",
    "qmpc_loop_joint_commands_device",
    "qmpc_loop_joint_init",
    "qmpc_loop_run_joint_device",
    "qmpc_loop_joint_commands",
    "qmpc_solve_warm",
    "qmpc_solve_warm_device",
    "qmpc_host_alloc",
    "qmpc_host_free",
    "qmpc_prepare",
    "qmpc_query",
)

# enum qmpc_query_what / qmpc_kernel_family (include/qmpc.h)
QUERY_HANDOFF_ACTIVE, QUERY_HANDOFF_ALLOC_FAILED, QUERY_KERNEL_FOR_BATCH, QUERY_LAST_KERNEL, QUERY_LANE_CAP, \
    QUERY_DEVICE_BYTES, QUERY_ZERO_COPY = 1, 2, 3, 4, 5, 6, 7
KERNEL_FAMILY = {0: "none", 1: "wform_lds", 2: "wform_ws", 3: "dense_lds", 4: "dense_ws", 5: "lane", 6: "lane_handoff"}


def default_params(horizon: int = 10, mode: int = MODE_CONVERGED, lib: C.CDLL | None = None) -> Params:
    lib = lib or load_library()
    p = Params()
    lib.qmpc_default_params(C.byref(p), horizon, mode)
    return p


def default_convex_params(horizon: int = 20, mode: int = MODE_CONVERGED, lib: C.CDLL | None = None) -> Params:
    """ConvexMpc values of gazebo_go1_convex_mpc.yaml (params.model = MODEL_CONVEX)."""
    lib = lib or load_library()
    p = Params()
    lib.qmpc_default_convex_params(C.byref(p), horizon, mode)
    return p


def default_biped8_params(horizon: int = 16, mode: int = MODE_CONVERGED, lib: C.CDLL | None = None) -> Params:
    """Synthetic 8-contact-point biped of BASELINE config 5 (params.model = MODEL_QUAT8)."""
    lib = lib or load_library()
    p = Params()
    lib.qmpc_default_biped8_params(C.byref(p), horizon, mode)
    return p


def default_loop_params(lib: C.CDLL | None = None) -> LoopParams:
    lib = lib or load_library()
    lp = LoopParams()
    lib.qmpc_default_loop_params(C.byref(lp))
    return lp


def loop_states(commands, lp: LoopParams | None = None, height: float = 0.3, yaw=0.0, lib: C.CDLL | None = None) -> np.ndarray:
    """qmpc_loop_state records of robots standing at `height` over their default footholds.
    commands: [B][7] = joy.{velx, vely, body_height, roll_rate, pitch_rate, yaw_rate}, movement_mode."""
    lib = lib or load_library()
    lp = lp or default_loop_params(lib)
    commands = np.atleast_2d(np.asarray(commands, dtype=np.float64))
    yaws = np.broadcast_to(np.asarray(yaw, dtype=np.float64), (commands.shape[0],))
    out = np.zeros(commands.shape[0], dtype=LOOP_STATE_DTYPE)
    for i, c in enumerate(commands):
        joy = np.ascontiguousarray(c[:6])
        lib.qmpc_loop_state_init(C.c_void_p(out[i:i + 1].ctypes.data), C.byref(lp), _ptr(joy), float(c[6]), float(height),
                                 float(yaws[i]))
    return out


class _PinnedBlock:
    """Owner of one qmpc_host_alloc allocation.  numpy arrays made from it (np.asarray and any view) keep it alive through
    their base chain; the memory is returned (qmpc_host_free) when the last of them is gone -- independent of any Solver."""

    def __init__(self, lib, nbytes: int):
        self._lib = lib
        self.ptr = lib.qmpc_host_alloc(nbytes)
        if not self.ptr:
            raise MemoryError("qmpc_host_alloc")
        self.__array_interface__ = {"shape": (nbytes,), "typestr": "|u1", "data": (int(self.ptr), False), "version": 3}

    def __del__(self):
        try:                      # the library may already be gone at interpreter shutdown
            if self.ptr:
                self._lib.qmpc_host_free(self.ptr)
                self.ptr = None
        except Exception:
            pass


class Solver:
    """Thin RAII wrapper over a qmpc_handle (one per GPU, single caller)."""

    def loop_run(self, states: np.ndarray, ticks: int, lp: LoopParams | None = None, trace: bool = False):
        """`ticks` ticks of the device-resident closed loop (front end + solve + plant per tick, state in HBM).
        Returns the final states (and, with trace, forces [ticks][B][12] and contacts [ticks][B][4])."""
        lp = lp or default_loop_params(self.lib)
        st = np.ascontiguousarray(states, dtype=LOOP_STATE_DTYPE).copy()
        B = st.shape[0]
        tf = np.zeros((ticks, B, 12)) if trace else None
        tc = np.zeros((ticks, B, 4)) if trace else None
        rc = self.lib.qmpc_loop_run(self._h, C.byref(lp), B, _ptr(st), int(ticks), _ptr(tf), _ptr(tc))
        if rc != OK:
            raise QmpcError(rc, "qmpc_loop_run")
        return (st, tf, tc) if trace else st

    def loop_run_device(self, batch: int, d_states: int, ticks: int, lp: LoopParams | None = None, d_trace_forces: int = 0,
                        d_trace_contacts: int = 0, stream: int = 0):
        lp = lp or default_loop_params(self.lib)
        rc = self.lib.qmpc_loop_run_device(self._h, C.byref(lp), int(batch), C.c_void_p(d_states), int(ticks),
                                           C.c_void_p(d_trace_forces) if d_trace_forces else None,
                                           C.c_void_p(d_trace_contacts) if d_trace_contacts else None,
                                           C.c_void_p(stream) if stream else None)
        if rc != OK:
            raise QmpcError(rc, "qmpc_loop_run_device")

    def __init__(self, params: Params, max_batch: int, device: int = 0, lib: C.CDLL | None = None):
        self.lib = lib or load_library()
        self.params = params.copy()
        self.max_batch = int(max_batch)
        self._h = C.c_void_p()
        st = self.lib.qmpc_create(C.byref(self.params), self.max_batch, device, C.byref(self._h))
        if st != OK:
            self._h = C.c_void_p()
            raise QmpcError(st, "qmpc_create")

    def close(self):
        h = getattr(self, "_h", None)
        if h is not None and h.value:
            self._h = None
            self.lib.qmpc_destroy(h)

    def __del__(self):
        try:                      # module globals may already be gone at interpreter shutdown
            self.close()
        except Exception:
            pass

    def set_params(self, params: Params):
        self.params = params.copy()
        st = self.lib.qmpc_set_params(self._h, C.byref(self.params))
        if st != OK:
            raise QmpcError(st, "qmpc_set_params")

    def solve(self, inputs: np.ndarray, want_traj: bool = False):
        """Host buffers in, host buffers out (H2D + kernel + D2H, synchronous)."""
        inputs = np.ascontiguousarray(inputs, dtype=INPUT_DTYPE)
        B, N = inputs.shape[0], self.params.horizon
        forces = np.zeros((B, NU), dtype=np.float64)
        info = np.zeros(B, dtype=INFO_DTYPE)
        if want_traj:
            tu = np.zeros((B, N, NU))
            tx = np.zeros((B, N + 1, NX))
            st = self.lib.qmpc_solve_traj(self._h, B, _ptr(inputs), _ptr(forces), _ptr(info), _ptr(tu), _ptr(tx))
            if st != OK:
                raise QmpcError(st, "qmpc_solve_traj")
            return forces, info, tu, tx
        st = self.lib.qmpc_solve(self._h, B, _ptr(inputs), _ptr(forces), _ptr(info))
        if st != OK:
            raise QmpcError(st, "qmpc_solve")
        return forces, info

    def prepare(self, batch: int | None = None):
        """Allocate now what solves of up to `batch` instances need later (qmpc_prepare)."""
        st = self.lib.qmpc_prepare(self._h, int(self.max_batch if batch is None else batch))
        if st != OK:
            raise QmpcError(st, "qmpc_prepare")

    def query(self, what: int, arg: int = 0) -> int:
        v = C.c_int64()
        st = self.lib.qmpc_query(self._h, int(what), int(arg), C.byref(v))
        if st != OK:
            raise QmpcError(st, "qmpc_query")
        return int(v.value)

    def kernel_for_batch(self, batch: int) -> str:
        return KERNEL_FAMILY[self.query(QUERY_KERNEL_FOR_BATCH, batch)]

    def pinned(self, shape, dtype=np.float64) -> np.ndarray:
        """Array in pinned, device-addressable host memory (qmpc_host_alloc).  The allocation lives as long as the array
        (or any view of it) does -- not as long as the solver: it is freed when the last reference goes away."""
        dt = np.dtype(dtype)
        count = int(np.prod(shape))
        block = _PinnedBlock(self.lib, max(count * dt.itemsize, 1))
        return np.asarray(block)[:count * dt.itemsize].view(dt).reshape(shape)      # base chain -> block

    def _check_out(self, name: str, a: np.ndarray, dtype, rows: int, cols: int | None):
        if not isinstance(a, np.ndarray) or not a.flags.c_contiguous or not a.flags.writeable:
            raise ValueError(f"{name}: a writable C-contiguous numpy array is required")
        if a.dtype != np.dtype(dtype):
            raise ValueError(f"{name}: dtype {a.dtype}, expected {np.dtype(dtype)}")
        need = rows * (cols if cols else 1)
        if a.size < need or (cols and a.ndim == 2 and a.shape[1] != cols):
            raise ValueError(f"{name}: shape {a.shape} cannot hold [{rows}" + (f", {cols}]" if cols else "]"))

    def solve_into(self, inputs: np.ndarray, forces: np.ndarray, info: np.ndarray | None = None):
        """qmpc_solve on caller-owned buffers (no allocation, no conversion): the call a C host makes.  The buffers are
        checked (layout, element type, room for the batch): the kernel writes through raw pointers."""
        if not isinstance(inputs, np.ndarray) or not inputs.flags.c_contiguous:
            raise ValueError("inputs: a C-contiguous numpy array is required")
        B = inputs.shape[0]
        if inputs.nbytes != B * INPUT_DTYPE.itemsize:
            raise ValueError(f"inputs: {inputs.nbytes} bytes for {B} records of {INPUT_DTYPE.itemsize} bytes")
        self._check_out("forces", forces, np.float64, B, NU)
        if info is not None:
            self._check_out("info", info, INFO_DTYPE, B, None)
        st = self.lib.qmpc_solve(self._h, B, _ptr(inputs), _ptr(forces), _ptr(info) if info is not None else None)
        if st != OK:
            raise QmpcError(st, "qmpc_solve")

    def solve_warm(self, inputs: np.ndarray, u_init: np.ndarray | None = None):
        """Warm-started solve: (forces [B,12], info, traj_u [B,N,12]); u_init = a previous traj_u (None: cold)."""
        inputs = np.ascontiguousarray(inputs, dtype=INPUT_DTYPE)
        B, N = inputs.shape[0], self.params.horizon
        forces = np.zeros((B, NU))
        info = np.zeros(B, dtype=INFO_DTYPE)
        tu = np.zeros((B, N, NU))
        ui = None if u_init is None else np.ascontiguousarray(u_init, dtype=np.float64).reshape(B, N, NU)
        st = self.lib.qmpc_solve_warm(self._h, B, _ptr(inputs), _ptr(ui), _ptr(forces), _ptr(info), _ptr(tu))
        if st != OK:
            raise QmpcError(st, "qmpc_solve_warm")
        return forces, info, tu

    def solve_device(self, batch: int, d_in: int, d_forces: int, d_info: int, stream: int = 0):
        """Device pointers (ints), stream-ordered, no synchronisation."""
        st = self.lib.qmpc_solve_device(self._h, int(batch), C.c_void_p(d_in), C.c_void_p(d_forces),
                                        C.c_void_p(d_info) if d_info else None,
                                        C.c_void_p(stream) if stream else None)
        if st != OK:
            raise QmpcError(st, "qmpc_solve_device")

    def solve_async(self, inputs: np.ndarray, forces: np.ndarray, info: np.ndarray = None):
        """Host buffers, non-blocking (qmpc_solve_async); the arrays must stay alive until wait()."""
        assert inputs.dtype == INPUT_DTYPE and inputs.flags.c_contiguous and forces.flags.c_contiguous
        st = self.lib.qmpc_solve_async(self._h, inputs.shape[0], _ptr(inputs), _ptr(forces),
                                       _ptr(info) if info is not None else None)
        if st != OK:
            raise QmpcError(st, "qmpc_solve_async")

    def gather(self, nccl_comm: int, d_local: int, count: int, d_all: int, stream: int = 0):
        """ncclAllGather of `count` doubles per rank (qmpc_gather); nccl_comm is the caller's ncclComm_t."""
        st = self.lib.qmpc_gather(self._h, C.c_void_p(nccl_comm), C.c_void_p(d_local), int(count),
                                  C.c_void_p(d_all), C.c_void_p(stream) if stream else None)
        if st != OK:
            raise QmpcError(st, "qmpc_gather")

    def wait(self):
        st = self.lib.qmpc_wait(self._h)
        if st != OK:
            raise QmpcError(st, "qmpc_wait")

    def last_kernel_ms(self) -> float:
        ms = C.c_float()
        st = self.lib.qmpc_last_kernel_ms(self._h, C.byref(ms))
        if st != OK:
            raise QmpcError(st, "qmpc_last_kernel_ms")
        return float(ms.value)

    # ---- 8 contact points (handle created with params.model = MODEL_QUAT8) ----
    def solve8(self, inputs: np.ndarray, want_traj: bool = False):
        inputs = np.ascontiguousarray(inputs, dtype=INPUT8_DTYPE)
        B, N = inputs.shape[0], self.params.horizon
        forces = np.zeros((B, 24), dtype=np.float64)
        info = np.zeros(B, dtype=INFO_DTYPE)
        if want_traj:
            tu = np.zeros((B, N, 24))
            tx = np.zeros((B, N + 1, NX))
            st = self.lib.qmpc_solve8_traj(self._h, B, _ptr(inputs), _ptr(forces), _ptr(info), _ptr(tu), _ptr(tx))
            if st != OK:
                raise QmpcError(st, "qmpc_solve8_traj")
            return forces, info, tu, tx
        st = self.lib.qmpc_solve8(self._h, B, _ptr(inputs), _ptr(forces), _ptr(info))
        if st != OK:
            raise QmpcError(st, "qmpc_solve8")
        return forces, info

    def solve8_device(self, batch: int, d_in: int, d_forces: int, d_info: int, stream: int = 0):
        st = self.lib.qmpc_solve8_device(self._h, int(batch), C.c_void_p(d_in), C.c_void_p(d_forces),
                                         C.c_void_p(d_info) if d_info else None,
                                         C.c_void_p(stream) if stream else None)
        if st != OK:
            raise QmpcError(st, "qmpc_solve8_device")

    # ---- ConvexMpc model (handle created with params.model = MODEL_CONVEX) ----
    def convex_solve(self, inputs: np.ndarray, want_traj: bool = False):
        """World-frame forces of legged::ConvexMpc::grf_update's problem, host buffers."""
        inputs = np.ascontiguousarray(inputs, dtype=CONVEX_INPUT_DTYPE)
        B, N = inputs.shape[0], self.params.horizon
        forces = np.zeros((B, NU), dtype=np.float64)
        info = np.zeros(B, dtype=INFO_DTYPE)
        if want_traj:
            tu = np.zeros((B, N, NU))
            tx = np.zeros((B, N + 1, 12))
            st = self.lib.qmpc_convex_solve_traj(self._h, B, _ptr(inputs), _ptr(forces), _ptr(info), _ptr(tu), _ptr(tx))
            if st != OK:
                raise QmpcError(st, "qmpc_convex_solve_traj")
            return forces, info, tu, tx
        st = self.lib.qmpc_convex_solve(self._h, B, _ptr(inputs), _ptr(forces), _ptr(info))
        if st != OK:
            raise QmpcError(st, "qmpc_convex_solve")
        return forces, info

    def convex_solve_device(self, batch: int, d_in: int, d_forces: int, d_info: int, stream: int = 0):
        st = self.lib.qmpc_convex_solve_device(self._h, int(batch), C.c_void_p(d_in), C.c_void_p(d_forces),
                                               C.c_void_p(d_info) if d_info else None,
                                               C.c_void_p(stream) if stream else None)
        if st != OK:
            raise QmpcError(st, "qmpc_convex_solve_device")

    def convex_linearize(self, inputs: np.ndarray):
        inputs = np.ascontiguousarray(inputs, dtype=CONVEX_INPUT_DTYPE)
        B, N = inputs.shape[0], self.params.horizon
        A = np.zeros((B, N, 12, 12))
        Bm = np.zeros((B, N, 12, 12))
        X = np.zeros((B, N + 1, 12))
        st = self.lib.qmpc_convex_linearize(self._h, B, _ptr(inputs), _ptr(A), _ptr(Bm), _ptr(X))
        if st != OK:
            raise QmpcError(st, "qmpc_convex_linearize")
        return A, Bm, X

    # ---- force -> joint torque consumer (BaseInterface::tau_ctrl_update) ----
    def default_go1_geometry(self) -> LegGeometry:
        g = LegGeometry()
        self.lib.qmpc_default_go1_geometry(C.byref(g))
        return g

    def leg_kinematics(self, geom: LegGeometry, joint_pos: np.ndarray):
        q = np.ascontiguousarray(joint_pos, dtype=np.float64).reshape(-1, 12)
        p = np.zeros((len(q), 12))
        J = np.zeros((len(q), 4, 9))
        st = self.lib.qmpc_leg_kinematics(self._h, C.byref(geom), len(q), _ptr(q), _ptr(p), _ptr(J))
        if st != OK:
            raise QmpcError(st, "qmpc_leg_kinematics")
        return p, J

    def torque_map(self, geom: LegGeometry, joint_pos, forces_body, contacts=None, walking: bool = True):
        q = np.ascontiguousarray(joint_pos, dtype=np.float64).reshape(-1, 12)
        f = np.ascontiguousarray(forces_body, dtype=np.float64).reshape(-1, 12)
        c = None if contacts is None else np.ascontiguousarray(contacts, dtype=np.float64).reshape(-1, 4)
        tau = np.zeros((len(q), 12))
        st = self.lib.qmpc_torque_map(self._h, C.byref(geom), len(q), _ptr(q), _ptr(f), _ptr(c), int(bool(walking)), _ptr(tau))
        if st != OK:
            raise QmpcError(st, "qmpc_torque_map")
        return tau

    def torque_map_device(self, geom: LegGeometry, batch: int, d_joint_pos: int, d_forces: int, d_contacts: int,
                          walking: bool, d_tau: int, stream: int = 0):
        st = self.lib.qmpc_torque_map_device(self._h, C.byref(geom), int(batch), C.c_void_p(d_joint_pos),
                                             C.c_void_p(d_forces), C.c_void_p(d_contacts) if d_contacts else None,
                                             int(bool(walking)), C.c_void_p(d_tau),
                                             C.c_void_p(stream) if stream else None)
        if st != OK:
            raise QmpcError(st, "qmpc_torque_map_device")

    def leg_inverse_kinematics(self, geom: LegGeometry, foot_pos_body, cur_joint_pos):
        """A1Kinematics::inv_kin for every (instance, leg); NaN where the foot is out of reach."""
        p = np.ascontiguousarray(foot_pos_body, dtype=np.float64).reshape(-1, 12)
        c = np.ascontiguousarray(cur_joint_pos, dtype=np.float64).reshape(-1, 12)
        q = np.zeros((len(p), 12))
        st = self.lib.qmpc_leg_inverse_kinematics(self._h, C.byref(geom), len(p), _ptr(p), _ptr(c), _ptr(q))
        if st != OK:
            raise QmpcError(st, "qmpc_leg_inverse_kinematics")
        return q

    def joint_commands(self, geom: LegGeometry, feedback: np.ndarray) -> np.ndarray:
        """BaseInterface::tau_ctrl_update for a batch of JOINT_FEEDBACK_DTYPE records -> JOINT_COMMAND_DTYPE."""
        fb = np.ascontiguousarray(feedback, dtype=JOINT_FEEDBACK_DTYPE)
        cmd = np.zeros(len(fb), dtype=JOINT_COMMAND_DTYPE)
        st = self.lib.qmpc_joint_commands(self._h, C.byref(geom), len(fb), _ptr(fb), _ptr(cmd))
        if st != OK:
            raise QmpcError(st, "qmpc_joint_commands")
        return cmd

    def joint_commands_device(self, geom: LegGeometry, batch: int, d_fb: int, d_cmd: int, stream: int = 0):
        st = self.lib.qmpc_joint_commands_device(self._h, C.byref(geom), int(batch), C.c_void_p(d_fb), C.c_void_p(d_cmd),
                                                 C.c_void_p(stream) if stream else None)
        if st != OK:
            raise QmpcError(st, "qmpc_joint_commands_device")

    def loop_joint_commands_device(self, geom: LegGeometry, batch: int, d_states: int, d_joint_pos: int, d_fb: int,
                                   d_cmd: int, stream: int = 0):
        """Joint-level feedback (d_fb may be 0) and commands of the robots of a closed loop, from their states."""
        st = self.lib.qmpc_loop_joint_commands_device(self._h, C.byref(geom), int(batch), C.c_void_p(d_states),
                                                      C.c_void_p(d_joint_pos), C.c_void_p(d_fb) if d_fb else None,
                                                      C.c_void_p(d_cmd), C.c_void_p(stream) if stream else None)
        if st != OK:
            raise QmpcError(st, "qmpc_loop_joint_commands_device")

    def loop_joint_commands(self, geom: LegGeometry, states: np.ndarray, joint_pos: np.ndarray | None = None):
        """Host-buffer form: (joint_pos [B][12] updated, feedback records, command records) of the robots' states."""
        st = np.ascontiguousarray(states, dtype=LOOP_STATE_DTYPE)
        B = len(st)
        jp = np.zeros((B, 12))
        if joint_pos is None:
            self.lib.qmpc_loop_joint_init(_ptr(jp), B)
        else:
            jp[:] = np.asarray(joint_pos, dtype=np.float64).reshape(B, 12)
        fb = np.zeros(B, dtype=JOINT_FEEDBACK_DTYPE)
        cmd = np.zeros(B, dtype=JOINT_COMMAND_DTYPE)
        rc = self.lib.qmpc_loop_joint_commands(self._h, C.byref(geom), B, _ptr(st), _ptr(jp), _ptr(fb), _ptr(cmd))
        if rc != OK:
            raise QmpcError(rc, "qmpc_loop_joint_commands")
        return jp, fb, cmd

    def loop_run_joint_device(self, geom: LegGeometry, batch: int, d_states: int, d_joint_pos: int, ticks: int,
                              lp: LoopParams | None = None, d_cmd: int = 0, d_trace_cmd: int = 0, stream: int = 0):
        """`ticks` ticks of the device-resident loop, each closed by the joint-level kernel (inside the captured graph)."""
        lp = lp or default_loop_params(self.lib)
        st = self.lib.qmpc_loop_run_joint_device(self._h, C.byref(lp), C.byref(geom), int(batch), C.c_void_p(d_states),
                                                 C.c_void_p(d_joint_pos), int(ticks), C.c_void_p(d_cmd) if d_cmd else None,
                                                 C.c_void_p(d_trace_cmd) if d_trace_cmd else None,
                                                 C.c_void_p(stream) if stream else None)
        if st != OK:
            raise QmpcError(st, "qmpc_loop_run_joint_device")

    def phase_profile(self, inputs: np.ndarray) -> np.ndarray:
        inputs = np.ascontiguousarray(inputs, dtype=INPUT_DTYPE)
        out = np.zeros((inputs.shape[0], 16), dtype=np.int64)
        st = self.lib.qmpc_debug_profile(self._h, inputs.shape[0], _ptr(inputs), _ptr(out))
        if st != OK:
            raise QmpcError(st, "qmpc_debug_profile")
        return out

    def linearize(self, inputs: np.ndarray):
        inputs = np.ascontiguousarray(inputs, dtype=INPUT_DTYPE)
        B, N = inputs.shape[0], self.params.horizon
        A = np.zeros((B, N, NE, NE))
        Bm = np.zeros((B, N, NE, NU))
        X = np.zeros((B, N + 1, NX))
        st = self.lib.qmpc_linearize(self._h, B, _ptr(inputs), _ptr(A), _ptr(Bm), _ptr(X))
        if st != OK:
            raise QmpcError(st, "qmpc_linearize")
        return A, Bm, X


from .scenarios import (go1_stand_input, quat_to_rot, random_go1_trot_states,  # noqa: E402,F401
                        random_go1_convex_states, random_biped8_states)
from .sharding import StepPipeline, gather_forces, shard_range, solve_sharded  # noqa: E402,F401
