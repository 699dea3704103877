"""Synthetic LeggedState records for the quaternion-MPC hot path.

The generator is the one fixed in SURVEY.md section 8(d): a counter-based RNG
(SplitMix64; seed ``0x5EED0000 + config_id``, stream = instance index) so that
the CPU oracle and the GPU see the very same arrays, on any machine, at any
batch size (instance i does not depend on the batch it is drawn in).

Field meaning follows ``struct qmpc_input`` in ``include/qmpc.h`` (which in turn
cites the LeggedState fields read by QuatMpc.cpp:109-276).
"""
from __future__ import annotations

import numpy as np

_GOLDEN = np.uint64(0x9E3779B97F4A7C15)
_M1 = np.uint64(0xBF58476D1CE4E5B9)
_M2 = np.uint64(0x94D049BB133111EB)


def _mix(z: np.ndarray) -> np.ndarray:
    """SplitMix64 finaliser on a uint64 array (wrap-around arithmetic)."""
    with np.errstate(over="ignore"):
        z = (z ^ (z >> np.uint64(30))) * _M1
        z = (z ^ (z >> np.uint64(27))) * _M2
        return z ^ (z >> np.uint64(31))


def _uniform(seed: int, index: np.ndarray, ndraw: int) -> np.ndarray:
    """[len(index), ndraw] doubles in [0,1): draw j of stream i."""
    with np.errstate(over="ignore"):
        key = _mix(np.uint64(seed) + (index.astype(np.uint64) + np.uint64(1)) * _GOLDEN)
        ctr = (np.arange(1, ndraw + 1, dtype=np.uint64) * _GOLDEN)[None, :]
        bits = _mix(key[:, None] + ctr)
    return (bits >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)


def _normal(u1: np.ndarray, u2: np.ndarray) -> np.ndarray:
    return np.sqrt(-2.0 * np.log(1.0 - u1)) * np.cos(2.0 * np.pi * u2)


def quat_mul(a: np.ndarray, b: np.ndarray) -> np.ndarray:
    aw, ax, ay, az = (a[..., i] for i in range(4))
    bw, bx, by, bz = (b[..., i] for i in range(4))
    return np.stack(
        [
            aw * bw - ax * bx - ay * by - az * bz,
            aw * bx + ax * bw + ay * bz - az * by,
            aw * by - ax * bz + ay * bw + az * bx,
            aw * bz + ax * by - ay * bx + az * bw,
        ],
        axis=-1,
    )


def quat_to_rot(q: np.ndarray) -> np.ndarray:
    """Body->world rotation matrix of unit quaternion (w,x,y,z); [...,9] row-major
    (what Eigen's toRotationMatrix gives at BaseInterface.cpp:196)."""
    w, x, y, z = (q[..., i] for i in range(4))
    R = np.stack(
        [
            1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y),
            2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x),
            2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y),
        ],
        axis=-1,
    )
    return R


def _axis_angle(axis: np.ndarray, ang: np.ndarray) -> np.ndarray:
    h = 0.5 * ang
    return np.concatenate([np.cos(h)[..., None], np.sin(h)[..., None] * axis], axis=-1)


def _input_dtype():
    from . import INPUT_DTYPE

    return INPUT_DTYPE


NOMINAL_FEET = np.array(  # yaml default footholds, gazebo_go1_quat_mpc.yaml:15-33
    [[0.20, 0.14, -0.30], [0.20, -0.14, -0.30], [-0.20, 0.14, -0.30], [-0.20, -0.14, -0.30]]
)


def random_go1_trot_states(batch: int, config_id: int = 2, first: int = 0, tilt_max: float = 0.5,
                           vel_sigma: float = 0.3) -> np.ndarray:
    """`batch` records, instance indices first .. first+batch-1."""
    seed = 0x5EED0000 + int(config_id)
    idx = np.arange(first, first + batch, dtype=np.uint64)
    u = _uniform(seed, idx, 48)
    c = iter(range(48))
    nx = lambda: u[:, next(c)]  # noqa: E731
    rec = np.zeros(batch, dtype=_input_dtype())

    yaw = (2.0 * nx() - 1.0) * np.pi
    tilt_dir = 2.0 * np.pi * nx()
    tilt = tilt_max * nx()
    zaxis = np.zeros((batch, 3)); zaxis[:, 2] = 1.0
    haxis = np.stack([np.cos(tilt_dir), np.sin(tilt_dir), np.zeros(batch)], axis=-1)
    q_yaw = _axis_angle(zaxis, yaw)
    q = quat_mul(q_yaw, _axis_angle(haxis, tilt))
    q /= np.linalg.norm(q, axis=-1, keepdims=True)
    rec["quat"] = q
    rec["rot"] = quat_to_rot(q)
    rec["lin_vel_body"] = np.stack([vel_sigma * _normal(nx(), nx()) for _ in range(3)], axis=-1)
    rec["ang_vel_body"] = np.stack([0.5 * _normal(nx(), nx()) for _ in range(3)], axis=-1)
    feet = np.tile(NOMINAL_FEET[None], (batch, 1, 1))
    for leg in range(4):
        for ax in range(3):
            feet[:, leg, ax] += 0.1 * nx() - 0.05
    rec["foot_pos_body"] = feet.reshape(batch, 12)  # [3*leg + axis] = Eigen 3x4 col-major
    g = nx()
    contacts = np.ones((batch, 4))
    a = g < 0.4            # FL + RR stance
    b = (g >= 0.4) & (g < 0.8)  # FR + RL stance
    contacts[a] = [1, 0, 0, 1]
    contacts[b] = [0, 1, 1, 0]
    rec["contacts"] = contacts
    rec["pos_ref_body"] = np.stack([0.02 * _normal(nx(), nx()) for _ in range(3)], axis=-1)
    rec["vel_ref_body"] = np.stack([nx() - 0.5, 0.2 * nx() - 0.1, np.zeros(batch)], axis=-1)
    small = np.stack([0.05 * _normal(nx(), nx()) for _ in range(2)] + [np.zeros(batch)], axis=-1)
    ang = np.linalg.norm(small, axis=-1)
    ax_ = np.where(ang[:, None] > 0, small / np.maximum(ang, 1e-300)[:, None], haxis)
    qd = quat_mul(q_yaw, _axis_angle(ax_, ang))
    qd /= np.linalg.norm(qd, axis=-1, keepdims=True)
    rec["quat_d"] = qd
    return rec


def go1_stand_input(feet: np.ndarray | None = None) -> np.ndarray:
    """Single stand-pose record (BASELINE config 0): identity attitude, rest,
    all four feet in contact, references at rest."""
    rec = np.zeros(1, dtype=_input_dtype())
    rec["quat"][0] = [1, 0, 0, 0]
    rec["rot"][0] = np.eye(3).reshape(9)
    f = NOMINAL_FEET if feet is None else np.asarray(feet, dtype=float)
    rec["foot_pos_body"][0] = f.reshape(12)
    rec["contacts"][0] = 1.0
    rec["quat_d"][0] = [1, 0, 0, 0]
    return rec


def random_go1_convex_states(batch: int, config_id: int = 12, first: int = 0, vel_sigma: float = 0.3) -> np.ndarray:
    """`batch` records of ``struct qmpc_convex_input`` (legged::ConvexMpc::grf_update's inputs,
    ConvexMpc.cpp:81-198), same counter-based generator: yaw U(-pi,pi), roll/pitch N(0,0.1^2),
    world-frame velocities N(0, vel_sigma^2) / N(0,0.5^2), footholds = Rz(yaw)(nominal + U(-.05,.05)^3),
    contacts 40/40/20 % as for the quaternion path, yaw-rate command U(-0.5,0.5)."""
    from . import CONVEX_INPUT_DTYPE

    seed = 0x5EED0000 + int(config_id)
    idx = np.arange(first, first + batch, dtype=np.uint64)
    u = _uniform(seed, idx, 48)
    c = iter(range(48))
    nx = lambda: u[:, next(c)]  # noqa: E731
    rec = np.zeros(batch, dtype=CONVEX_INPUT_DTYPE)
    yaw = (2.0 * nx() - 1.0) * np.pi
    roll = 0.1 * _normal(nx(), nx())
    pitch = 0.1 * _normal(nx(), nx())
    rec["euler"] = np.stack([roll, pitch, yaw], axis=-1)
    pos = np.stack([0.05 * _normal(nx(), nx()), 0.05 * _normal(nx(), nx()), 0.3 + 0.02 * _normal(nx(), nx())], axis=-1)
    rec["pos_world"] = pos
    rec["ang_vel_world"] = np.stack([0.5 * _normal(nx(), nx()) for _ in range(3)], axis=-1)
    rec["lin_vel_world"] = np.stack([vel_sigma * _normal(nx(), nx()) for _ in range(3)], axis=-1)
    cy, sy = np.cos(yaw), np.sin(yaw)
    feet = np.tile(NOMINAL_FEET[None], (batch, 1, 1))
    for leg in range(4):
        for ax in range(3):
            feet[:, leg, ax] += 0.1 * nx() - 0.05
    fw = np.empty_like(feet)
    fw[:, :, 0] = cy[:, None] * feet[:, :, 0] - sy[:, None] * feet[:, :, 1]
    fw[:, :, 1] = sy[:, None] * feet[:, :, 0] + cy[:, None] * feet[:, :, 1]
    fw[:, :, 2] = feet[:, :, 2]
    rec["foot_pos_abs_com"] = fw.reshape(batch, 12)
    g = nx()
    contacts = np.ones((batch, 4))
    contacts[g < 0.4] = [1, 0, 0, 1]
    contacts[(g >= 0.4) & (g < 0.8)] = [0, 1, 1, 0]
    rec["contacts"] = contacts
    rec["pos_d_world"] = np.stack([pos[:, 0], pos[:, 1], np.full(batch, 0.3)], axis=-1)
    vx, vy = nx() - 0.5, 0.2 * nx() - 0.1
    rec["lin_vel_d_world"] = np.stack([cy * vx - sy * vy, sy * vx + cy * vy, np.zeros(batch)], axis=-1)
    rec["yaw_rate_d"] = nx() - 0.5
    return rec


def random_biped8_states(batch: int, config_id: int = 5, first: int = 0, tilt_max: float = 0.3,
                         vel_sigma: float = 0.3) -> np.ndarray:
    """`batch` records of ``struct qmpc_input8`` for BASELINE config 5: a SYNTHETIC biped with two
    0.2 x 0.1 m feet, 4 corner contact points each (points 0-3 left foot, 4-7 right foot), hip height
    0.85 m.  The humanoid branch is not in the reference checkout, so this footprint is this
    repository's choice; same counter-based generator as the Go1 states.  Support:
    40 % both feet, 30 % left only, 30 % right only."""
    from . import INPUT8_DTYPE

    seed = 0x5EED0000 + int(config_id)
    idx = np.arange(first, first + batch, dtype=np.uint64)
    u = _uniform(seed, idx, 64)
    c = iter(range(64))
    nx = lambda: u[:, next(c)]  # noqa: E731
    rec = np.zeros(batch, dtype=INPUT8_DTYPE)
    yaw = (2.0 * nx() - 1.0) * np.pi
    tilt_dir = 2.0 * np.pi * nx()
    tilt = tilt_max * nx()
    zaxis = np.zeros((batch, 3)); zaxis[:, 2] = 1.0
    haxis = np.stack([np.cos(tilt_dir), np.sin(tilt_dir), np.zeros(batch)], axis=-1)
    q_yaw = _axis_angle(zaxis, yaw)
    q = quat_mul(q_yaw, _axis_angle(haxis, tilt))
    q /= np.linalg.norm(q, axis=-1, keepdims=True)
    rec["quat"] = q
    rec["rot"] = quat_to_rot(q)
    rec["lin_vel_body"] = np.stack([vel_sigma * _normal(nx(), nx()) for _ in range(3)], axis=-1)
    rec["ang_vel_body"] = np.stack([0.5 * _normal(nx(), nx()) for _ in range(3)], axis=-1)
    feet = np.zeros((batch, 8, 3))
    corners = np.array([[0.1, 0.05], [0.1, -0.05], [-0.1, 0.05], [-0.1, -0.05]])
    for foot, ysign in ((0, 1.0), (1, -1.0)):
        centre = np.stack([0.06 * nx() - 0.03, ysign * 0.1 + 0.06 * nx() - 0.03, -0.85 + 0.06 * nx() - 0.03], axis=-1)
        for k in range(4):
            feet[:, 4 * foot + k, 0] = centre[:, 0] + corners[k, 0]
            feet[:, 4 * foot + k, 1] = centre[:, 1] + corners[k, 1]
            feet[:, 4 * foot + k, 2] = centre[:, 2]
    rec["foot_pos_body"] = feet.reshape(batch, 24)
    g = nx()
    contacts = np.ones((batch, 8))
    contacts[(g >= 0.4) & (g < 0.7), 4:] = 0.0      # left foot only
    contacts[g >= 0.7, :4] = 0.0                    # right foot only
    rec["contacts"] = contacts
    rec["pos_ref_body"] = np.stack([0.02 * _normal(nx(), nx()) for _ in range(3)], axis=-1)
    rec["vel_ref_body"] = np.stack([nx() - 0.5, 0.2 * nx() - 0.1, np.zeros(batch)], axis=-1)
    small = np.stack([0.05 * _normal(nx(), nx()) for _ in range(2)] + [np.zeros(batch)], axis=-1)
    ang = np.linalg.norm(small, axis=-1)
    ax_ = np.where(ang[:, None] > 0, small / np.maximum(ang, 1e-300)[:, None], haxis)
    qd = quat_mul(q_yaw, _axis_angle(ax_, ang))
    qd /= np.linalg.norm(qd, axis=-1, keepdims=True)
    rec["quat_d"] = qd
    return rec
