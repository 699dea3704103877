// qmpc_loop_math.h -- arithmetic shared by the device-resident closed loop (qmpc_loop.hip) and its host
// reference driver (host/ClosedLoopHost.h): attitude conversions and the single-rigid-body PLANT that closes the
// loop around the MPC.  The plant is this repository's (the reference closes its loop through Gazebo / the robot);
// everything on the controller side of it mirrors the reference (qmpc_loop.hip cites the lines).
#pragma once

#include <math.h>

#if defined(__HIPCC__)
#define QMPC_HD __host__ __device__ inline
#else
#define QMPC_HD inline
#endif
// no fused multiply-adds on the device side either (g++ does not fuse on the host): both sides then differ only
// through their math libraries' atan2 / sin / cos
#if defined(__clang__)
#define QMPC_NO_CONTRACT _Pragma("clang fp contract(off)")
#else
#define QMPC_NO_CONTRACT
#endif

namespace qmpc_loop {

// body -> world rotation of q = (w, x, y, z); row-major 3x3 (Eigen's Quaterniond::toRotationMatrix)
QMPC_HD void quat_to_rot(const double* q, double* R) {
  QMPC_NO_CONTRACT
  const double w = q[0], x = q[1], y = q[2], z = q[3];
  const double tx = 2.0 * x, ty = 2.0 * y, tz = 2.0 * z;
  const double twx = tx * w, twy = ty * w, twz = tz * w;
  const double txx = tx * x, txy = ty * x, txz = tz * x;
  const double tyy = ty * y, tyz = tz * y, tzz = tz * z;
  R[0] = 1.0 - (tyy + tzz); R[1] = txy - twz;         R[2] = txz + twy;
  R[3] = txy + twz;         R[4] = 1.0 - (txx + tzz); R[5] = tyz - twx;
  R[6] = txz - twy;         R[7] = tyz + twx;         R[8] = 1.0 - (txx + tyy);
}

// roll, pitch, yaw of q = (w, x, y, z): Utils::quat_to_euler (legged_ctrl/src/utils/Utils.cpp:7-33), fbk.torso_euler
QMPC_HD void quat_to_euler(const double* q, double* e) {
  QMPC_NO_CONTRACT
  const double w = q[0], x = q[1], y = q[2], z = q[3];
  const double yy = y * y;
  e[0] = atan2(2.0 * (w * x + y * z), 1.0 - 2.0 * (x * x + yy));
  double t2 = 2.0 * (w * y - z * x);
  t2 = t2 > 1.0 ? 1.0 : t2;
  t2 = t2 < -1.0 ? -1.0 : t2;
  e[1] = asin(t2);
  e[2] = atan2(2.0 * (w * z + x * y), 1.0 - 2.0 * (yy + z * z));
}

// yaw-only rotation next to the full one (fbk.torso_rot_mat_z beside fbk.torso_rot_mat)
QMPC_HD void rot_to_rot_z(const double* R, double* Rz) {
  const double yaw = atan2(R[3], R[0]);
  const double c = cos(yaw), s = sin(yaw);
  Rz[0] = c;   Rz[1] = -s;  Rz[2] = 0.0;
  Rz[3] = s;   Rz[4] = c;   Rz[5] = 0.0;
  Rz[6] = 0.0; Rz[7] = 0.0; Rz[8] = 1.0;
}

// Plant: one explicit-midpoint step of a free rigid body under the body-frame foot forces u (3 per leg; swing legs
// carry zero force), world-frame position / velocity, body-frame angular velocity, no gyroscopic term (as the
// controller's own model, AltroUtils.cpp:389-391), feet fixed in the world during the step:
//   p' = v,  v' = R(q) sum(u)/m + g,  q' = 1/2 G(q) w,  w' = Iinv sum(r_l x u_l),  r_l = R(q)'(foot_l - p)
// state x = [p(3) q(4) v(3) w(3)]; the quaternion is re-normalised after the step.
QMPC_HD void plant_rate(const double* x, const double* u, const double* feet_world, int nleg, double mass,
                        const double* Iinv, double* xd) {
  QMPC_NO_CONTRACT
  double R[9];
  quat_to_rot(&x[3], R);
  double F[3] = {0.0, 0.0, 0.0}, tau[3] = {0.0, 0.0, 0.0};
  for (int l = 0; l < nleg; ++l) {
    const double d[3] = {feet_world[3 * l] - x[0], feet_world[3 * l + 1] - x[1], feet_world[3 * l + 2] - x[2]};
    const double r[3] = {R[0] * d[0] + R[3] * d[1] + R[6] * d[2], R[1] * d[0] + R[4] * d[1] + R[7] * d[2],
                         R[2] * d[0] + R[5] * d[1] + R[8] * d[2]};
    const double* f = &u[3 * l];
    F[0] += f[0]; F[1] += f[1]; F[2] += f[2];
    tau[0] += r[1] * f[2] - r[2] * f[1];
    tau[1] += r[2] * f[0] - r[0] * f[2];
    tau[2] += r[0] * f[1] - r[1] * f[0];
  }
  xd[0] = x[7]; xd[1] = x[8]; xd[2] = x[9];
  const double s = x[3], qx = x[4], qy = x[5], qz = x[6], wx = x[10], wy = x[11], wz = x[12];
  xd[3] = 0.5 * (-qx * wx - qy * wy - qz * wz);        // 1/2 G(q) w, QuaternionUtils.cpp:30-52
  xd[4] = 0.5 * (s * wx - qz * wy + qy * wz);
  xd[5] = 0.5 * (qz * wx + s * wy - qx * wz);
  xd[6] = 0.5 * (-qy * wx + qx * wy + s * wz);
  for (int a = 0; a < 3; ++a) {
    xd[7 + a] = (R[3 * a] * F[0] + R[3 * a + 1] * F[1] + R[3 * a + 2] * F[2]) / mass;
    xd[10 + a] = Iinv[3 * a] * tau[0] + Iinv[3 * a + 1] * tau[1] + Iinv[3 * a + 2] * tau[2];
  }
  xd[9] += -9.81;
}

QMPC_HD void plant_step(double* x, const double* u, const double* feet_world, int nleg, double mass,
                        const double* Iinv, double dt) {
  QMPC_NO_CONTRACT
  double k1[13], xm[13], k2[13];
  plant_rate(x, u, feet_world, nleg, mass, Iinv, k1);
  for (int i = 0; i < 13; ++i) xm[i] = x[i] + 0.5 * dt * k1[i];
  plant_rate(xm, u, feet_world, nleg, mass, Iinv, k2);
  for (int i = 0; i < 13; ++i) x[i] = x[i] + dt * k2[i];
  const double n = sqrt(x[3] * x[3] + x[4] * x[4] + x[5] * x[5] + x[6] * x[6]);
  for (int i = 3; i < 7; ++i) x[i] = x[i] / n;
}

// Condition matrix of the swing-foot quintic p(t) = sum_k a_k t^k (QuinticCurve::get_foot_swing_target,
// Utils.cpp:236-293): rows p(0), p(T), p'(0), p'(T), p(T/2), p'(T/2).  Generated, with the reference's FLOAT
// evaluation order so that every entry carries the same rounding as upstream's hand-written expressions: a power is
// the left-to-right product  (c * T) * T * ...  started from the derivative factor c, and the half-time rows divide
// that product by 2^j afterwards.
QMPC_HD void swing_condition_matrix(float T, double C[6][6]) {
  QMPC_NO_CONTRACT
  for (int r = 0; r < 6; ++r)
    for (int k = 0; k < 6; ++k) C[r][k] = 0.0;
  C[0][0] = 1.0;                                   // p(0)
  C[2][1] = 1.0;                                   // p'(0)
  for (int k = 0; k < 6; ++k) {
    float pw = 1.0f;                               // T^k
    for (int j = 0; j < k; ++j) pw = pw * T;
    C[1][k] = pw;                                  // p(T)
    C[4][k] = (k == 0) ? 1.0f : pw / (float)(1 << k);             // p(T/2) = T^k / 2^k
    if (k >= 1) {
      float dv = (float)k;                         // k T^(k-1), as (k * T) * T * ...
      for (int j = 0; j < k - 1; ++j) dv = dv * T;
      C[3][k] = (k == 1) ? 1.0f : dv;              // p'(T)
      C[5][k] = (k == 1) ? 1.0f : dv / (float)(1 << (k - 1));     // p'(T/2) = k T^(k-1) / 2^(k-1)
    }
  }
}

// 3x3 inverse by cofactors (row-major)
QMPC_HD void inv3(const double* A, double* B) {
  QMPC_NO_CONTRACT
  const double c00 = A[4] * A[8] - A[5] * A[7], c01 = A[5] * A[6] - A[3] * A[8], c02 = A[3] * A[7] - A[4] * A[6];
  const double id = 1.0 / (A[0] * c00 + A[1] * c01 + A[2] * c02);
  B[0] = c00 * id; B[1] = (A[2] * A[7] - A[1] * A[8]) * id; B[2] = (A[1] * A[5] - A[2] * A[4]) * id;
  B[3] = c01 * id; B[4] = (A[0] * A[8] - A[2] * A[6]) * id; B[5] = (A[2] * A[3] - A[0] * A[5]) * id;
  B[6] = c02 * id; B[7] = (A[1] * A[6] - A[0] * A[7]) * id; B[8] = (A[0] * A[4] - A[1] * A[3]) * id;
}

}  // namespace qmpc_loop
