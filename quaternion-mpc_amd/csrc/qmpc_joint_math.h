// qmpc_joint_math.h -- the low-level command of a tick, shared by the device kernels (qmpc_joint.hip) and the host
// mirror (host/JointCommandsHip.h): leg Jacobian, closed-form inverse kinematics and the per-leg part of
// BaseInterface::tau_ctrl_update (legged_ctrl/src/interfaces/BaseInterface.cpp:343-408).
//   A1Kinematics::jac       legged_ctrl/src/utils/A1Kinematics.cpp:15-19 (closed form :86-128)
//   A1Kinematics::inv_kin   A1Kinematics.cpp:335-459, single-precision atan2 approximation :291-312
// Host and device run the same expressions; they differ only through sin / cos / acos / sqrt of their math libraries.
#pragma once

#include "qmpc_loop_math.h"

namespace qmpc_joint {

struct LegPlane {            // the leg seen from the hip: extension L / fore-aft X, calf-only parts, hip link D
  double s0, c0, L, X, L2, X2, D;
};
QMPC_HD LegPlane leg_plane(const double* q, const double* rho_opt, const double* rho_fix) {
  QMPC_NO_CONTRACT
  LegPlane t;
  double s1, c1, s12, c12;
  sincos(q[0], &t.s0, &t.c0);
  sincos(q[1], &s1, &c1);
  sincos(q[1] + q[2], &s12, &c12);
  const double lce = rho_fix[4] - rho_opt[2];
  t.L2 = lce * c12 + rho_opt[0] * s12;
  t.X2 = -lce * s12 + rho_opt[0] * c12;
  t.L = rho_fix[3] * c1 + t.L2;
  t.X = -rho_fix[3] * s1 + t.X2;
  t.D = rho_fix[2] + rho_opt[1];
  return t;
}
// J column-major: J[3j+i] = d p_i / d q_j
QMPC_HD void leg_jacobian(const LegPlane& k, double* J) {
  QMPC_NO_CONTRACT
  J[0] = 0.0;    J[1] = -k.D * k.s0 + k.L * k.c0; J[2] = k.D * k.c0 + k.L * k.s0;
  J[3] = -k.L;   J[4] = k.X * k.s0;               J[5] = -k.X * k.c0;
  J[6] = -k.L2;  J[7] = k.X2 * k.s0;              J[8] = -k.X2 * k.c0;
}

// atan2 in single precision: odd degree-11 polynomial on [-1, 1] (five fused Horner steps), reciprocal identity for
// |y| > |x|, half-plane shift for x < 0
QMPC_HD float atan2_single(double yd, double xd) {
  QMPC_NO_CONTRACT
  const float y = (float)yd, x = (float)xd;
  const bool steep = fabsf(x) < fabsf(y);
  const float t = steep ? x / y : y / x;
  const float t2 = t * t;
  float h = fmaf(t2, -0.01172120f, 0.05265332f);
  h = fmaf(t2, h, -0.11643287f);
  h = fmaf(t2, h, 0.19354346f);
  h = fmaf(t2, h, -0.33262347f);
  h = fmaf(t2, h, 0.99997726f);
  float r = t * h;
  const float kPi = 3.14159265358979323846f, kHalfPi = 1.57079632679489661923f;
  if (steep) r = (t >= 0.0f ? kHalfPi : -kHalfPi) - r;
  if (x < 0.0f) r = (y >= 0.0f ? kPi : -kPi) + r;
  return r;
}

// q = (hip, thigh, calf) that puts the foot at p (body frame); cur_hip picks between the two hip solutions.
// Sums of two approximations are formed in single precision and sums with multiples of pi in double, as the
// expression types of the reference make them.  Out of reach: NaN.
QMPC_HD void leg_inverse(const double* p, double cur_hip, const double* rho_fix, double* q) {
  QMPC_NO_CONTRACT
  const double kPi = 3.14159265358979323846;
  const double oy = rho_fix[1], d = rho_fix[2], lt = rho_fix[3], lc = rho_fix[4];
  const double xs = p[0] - rho_fix[0], ys = p[1] - oy, zf = p[2];
  double L = sqrt(zf * zf + ys * ys - d * d);
  double hip = 0.0, mirror = 0.0;
  const double sd = oy > 0 ? d : -d;                      // |hip link| for the Go1 geometry
  if (oy > 0) {                                           // left legs
    const float aL = atan2_single(L, sd);
    if (zf > 0) {                                         // foot above the hip axis
      if (ys > 0) hip = (double)(atan2_single(zf, ys) - aL);
      else if (ys == 0) hip = kPi / 2 - (double)aL;
      else hip = (kPi - (double)atan2_single(zf, -ys)) - (double)aL;
      mirror = (double)(atan2_single(zf, ys) + aL);
    } else if (zf < 0) {
      if (ys > 0) hip = (double)(atan2_single(zf, ys) + aL);
      else if (ys == 0) hip = -kPi / 2 + (double)aL;
      else hip = (-kPi - (double)atan2_single(zf, -ys)) + (double)aL;
      mirror = (double)(atan2_single(zf, ys) - aL);
    } else {
      hip = (double)aL;
      mirror = (double)(-aL);
    }
  } else if (zf > 0) {                                    // right legs
    const float aL = atan2_single(L, sd), an = atan2_single(zf, -ys);
    if (ys < 0) hip = (double)(-an + aL);
    else if (ys == 0) hip = -kPi / 2 + (double)aL;
    else hip = (-kPi + (double)atan2_single(zf, ys)) + (double)aL;
    mirror = (double)(-aL - an);
  } else if (zf < 0) {
    const float aL = atan2_single(L, sd), an = atan2_single(-zf, -ys);
    if (ys < 0) hip = (double)(an - aL);
    else if (ys == 0) hip = -kPi / 2 - (double)aL;
    else hip = (kPi - (double)atan2_single(-zf, ys)) - (double)aL;
    mirror = (double)(an + aL);
  }                                                       // right leg with zf == 0: both stay 0, as in the reference
  if (!(fabs(hip - cur_hip) < fabs(mirror - cur_hip))) hip = mirror;

  const double cb = (lt * lt + lc * lc - xs * xs - L * L) / (2 * lt * lc);     // cosine of the interior knee angle
  const double beta = fabs(cb + 1) < 0.001 ? kPi : fabs(cb - 1) < 0.001 ? 0.0 : acos(cb);
  const double calf = beta - kPi;
  if (zf > d * sin(hip)) L = -L;
  double sk, ck;
  sincos(-calf, &sk, &ck);
  double thigh = (double)atan2_single(-xs, L) + (double)atan2_single(lc * sk, lt + lc * ck);
  if (thigh < -60 * kPi / 180) thigh += 2 * kPi;
  else if (thigh > 240 * kPi / 180) thigh -= 2 * kPi;
  q[0] = hip; q[1] = thigh; q[2] = calf;
}

// x = J^-1 b, J column-major, elimination with row pivoting on the largest entry of the column (PartialPivLU)
QMPC_HD void solve3(const double* J, const double* b, double* x) {
  QMPC_NO_CONTRACT
  double a[3][4] = {{J[0], J[3], J[6], b[0]}, {J[1], J[4], J[7], b[1]}, {J[2], J[5], J[8], b[2]}};
  for (int c = 0; c < 3; ++c) {
    int piv = c;
    for (int i = c + 1; i < 3; ++i)
      if (fabs(a[i][c]) > fabs(a[piv][c])) piv = i;
    for (int j = 0; j < 4; ++j) {          // branch-free row exchange keeps `a` in registers on the device
      const double u = a[c][j], v = piv == 1 ? a[1][j] : piv == 2 ? a[2][j] : a[0][j];
      a[c][j] = v;
      if (piv == 1) a[1][j] = u; else if (piv == 2) a[2][j] = u; else a[0][j] = u;
    }
    for (int i = c + 1; i < 3; ++i) {
      const double f = a[i][c] / a[c][c];
      for (int j = c; j < 4; ++j) a[i][j] -= f * a[c][j];
    }
  }
  x[2] = a[2][3] / a[2][2];
  x[1] = (a[1][3] - a[1][2] * x[2]) / a[1][1];
  x[0] = ((a[0][3] - a[0][1] * x[1]) - a[0][2] * x[2]) / a[0][0];
}

// One leg of tau_ctrl_update.  R row-major body -> world; q / qd the measured joint angles / velocities of the leg.
QMPC_HD void leg_command(const double* rho_opt, const double* rho_fix, const double* R, const double* torso_pos,
                         const double* torso_vel, const double* q, const double* qd, const double* foot_pos_tgt,
                         const double* foot_vel_tgt, const double* force_body, bool planned_contact, bool walking,
                         double* ang_tgt, double* vel_tgt, double* tau_tgt) {
  QMPC_NO_CONTRACT
  const LegPlane k = leg_plane(q, rho_opt, rho_fix);
  double J[9];
  leg_jacobian(k, J);
  double tau[3];
  for (int j = 0; j < 3; ++j)
    tau[j] = -(J[3 * j] * force_body[0] + J[3 * j + 1] * force_body[1] + J[3 * j + 2] * force_body[2]);
  if (!walking) {                                           // BaseInterface.cpp:400-403
    for (int j = 0; j < 3; ++j) { tau_tgt[j] = tau[j]; ang_tgt[j] = q[j]; vel_tgt[j] = qd[j]; }
    return;
  }
  const double dp[3] = {foot_pos_tgt[0] - torso_pos[0], foot_pos_tgt[1] - torso_pos[1], foot_pos_tgt[2] - torso_pos[2]};
  const double dv[3] = {foot_vel_tgt[0] - torso_vel[0], foot_vel_tgt[1] - torso_vel[1], foot_vel_tgt[2] - torso_vel[2]};
  double pb[3], vb[3], qt[3], qv[3];
  for (int i = 0; i < 3; ++i) {                             // R' v
    pb[i] = R[i] * dp[0] + R[3 + i] * dp[1] + R[6 + i] * dp[2];
    vb[i] = R[i] * dv[0] + R[3 + i] * dv[1] + R[6 + i] * dv[2];
  }
  leg_inverse(pb, q[0], rho_fix, qt);                       // :349-355
  solve3(J, vb, qv);                                        // :358-364
  const bool bad_q = (qt[0] != qt[0]) || (qt[1] != qt[1]) || (qt[2] != qt[2]);
  const bool bad_v = (qv[0] != qv[0]) || (qv[1] != qv[1]) || (qv[2] != qv[2]);
  for (int j = 0; j < 3; ++j) {
    ang_tgt[j] = bad_q ? q[j] : qt[j];
    vel_tgt[j] = bad_v ? qd[j] : qv[j];
    tau_tgt[j] = planned_contact ? tau[j] : 0.0;            // :367-371
  }
}

}  // namespace qmpc_joint
