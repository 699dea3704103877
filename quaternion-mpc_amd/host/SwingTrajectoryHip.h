// SwingTrajectoryHip.h -- the step before the force path: swing-foot
// trajectory and Raibert foothold targets, restated on the host.
//   QuinticCurve::get_foot_swing_target   legged_ctrl/src/utils/Utils.cpp:236-293
//   Raibert heuristic + foot targets      legged_ctrl/src/interfaces/BaseInterface.cpp:266-288
// Neither is on the force path (the MPC only sees the contact flags); they fill the foot targets the
// low-level joint controller tracks (BaseInterface.cpp:348-364).
#pragma once

#include <cmath>

#include "../csrc/qmpc_loop_math.h"

namespace legged {

// Per axis the reference fits a quintic a0..a5 through six conditions by inverting a 6x6 matrix
// (Utils.cpp:238-244,262,275,288): p(0), p(T), p'(0), p'(T), p(T/2), p'(T/2).  The matrix entries
// are formed in FLOAT (t and T are float arguments) and the solve is in double; here the system is
// solved by Gaussian elimination with partial pivoting on the same entries.
class QuinticCurveHip {
 public:
  // out: pos(3) vel(3) acc(3)
  void get_foot_swing_target(float t, float T, const double start[3], const double fin[3], double out[9]) {
    double C[6][6];
    qmpc_loop::swing_condition_matrix(T, C);      // p(0), p(T), p'(0), p'(T), p(T/2), p'(T/2); float entries
    const double dx = fin[0] - start[0], dy = fin[1] - start[1];
    const double k = 1.26 / T;                                   // Utils.cpp:247
    const double v_xy_mid = k * std::sqrt(dx * dx + dy * dy);
    const double theta = std::atan2(std::fabs(dy), std::fabs(dx));
    const double v_x_mid = (dx >= 0 ? 1 : -1) * v_xy_mid * std::cos(theta);
    const double v_y_mid = (dy >= 0 ? 1 : -1) * v_xy_mid * std::sin(theta);
    // conditions per axis: p0, pT, v0, vT, p(T/2), v(T/2)   (:255-261, :268-274, :281-287)
    const double con[3][6] = {{start[0], fin[0], 0.0, 0.0, (start[0] + fin[0]) / 2, v_x_mid},
                              {start[1], fin[1], 0.0, 0.0, (start[1] + fin[1]) / 2, v_y_mid},
                              {start[2], fin[2], 0.1, -0.1, 0.1, 0.0}};
    double a[3][6];
    solve3(C, con, a);
    const double td = t;
    // the polynomial is evaluated left to right with DOUBLE coefficients times the float t (Utils.cpp:263-265: `a_z(2) * t * t`),
    // i.e. every power of t is formed in double; only the entries of the condition matrix are float products of T
    for (int ax = 0; ax < 3; ++ax) {
      const double* c = a[ax];
      out[ax] = c[0] + c[1] * td + c[2] * td * td + c[3] * td * td * td + c[4] * td * td * td * td + c[5] * td * td * td * td * td;
      out[3 + ax] = c[1] + 2 * c[2] * td + 3 * c[3] * td * td + 4 * c[4] * td * td * td + 5 * c[5] * td * td * td * td;
      out[6 + ax] = 2 * c[2] + 6 * c[3] * td + 12 * c[4] * td * td + 20 * c[5] * td * td * td;
    }
  }

 private:
  // three right-hand sides, one factorisation (partial pivoting)
  static void solve3(double C[6][6], const double rhs[3][6], double x[3][6]) {
    double b[6][3];
    for (int i = 0; i < 6; ++i)
      for (int r = 0; r < 3; ++r) b[i][r] = rhs[r][i];
    for (int col = 0; col < 6; ++col) {
      int piv = col;
      for (int i = col + 1; i < 6; ++i)
        if (std::fabs(C[i][col]) > std::fabs(C[piv][col])) piv = i;
      if (piv != col) {
        for (int j = 0; j < 6; ++j) { const double tmp = C[col][j]; C[col][j] = C[piv][j]; C[piv][j] = tmp; }
        for (int r = 0; r < 3; ++r) { const double tmp = b[col][r]; b[col][r] = b[piv][r]; b[piv][r] = tmp; }
      }
      for (int i = col + 1; i < 6; ++i) {
        const double f = C[i][col] / C[col][col];
        for (int j = col; j < 6; ++j) C[i][j] -= f * C[col][j];
        for (int r = 0; r < 3; ++r) b[i][r] -= f * b[col][r];
      }
    }
    for (int r = 0; r < 3; ++r)
      for (int i = 5; i >= 0; --i) {
        double s = b[i][r];
        for (int j = i + 1; j < 6; ++j) s -= C[i][j] * x[r][j];
        x[r][i] = s / C[i][i];
      }
  }
};

constexpr double kFootDeltaXLimit = 0.5;   // LeggedParams.h:21
constexpr double kFootDeltaYLimit = 0.3;   // LeggedParams.h:22

// Raibert heuristic and the three foot-target frames (BaseInterface.cpp:266-288).  Works on any State with
// the reference's field names.
template <class State>
void raibert_foot_targets(State& s) {
  for (int r = 0; r < 3; ++r)   // torso_lin_vel_rel = R_z' v_world
    s.fbk.torso_lin_vel_rel[r] = s.fbk.torso_rot_mat_z(0, r) * s.fbk.torso_lin_vel_world[0] +
                                 s.fbk.torso_rot_mat_z(1, r) * s.fbk.torso_lin_vel_world[1] +
                                 s.fbk.torso_rot_mat_z(2, r) * s.fbk.torso_lin_vel_world[2];
  const double k = std::sqrt(std::fabs(s.fbk.torso_pos_world[2]) / 9.81);
  double d[3] = {0.0, 0.0, 0.0};
  d[0] = k * (s.fbk.torso_lin_vel_rel[0] - s.ctrl.torso_lin_vel_d_rel[0]) +
         (1.0 / s.param.gait_freq) / 2.0 * s.ctrl.torso_lin_vel_d_rel[0];
  if (d[0] < -kFootDeltaXLimit) d[0] = -kFootDeltaXLimit;
  if (d[0] > kFootDeltaXLimit) d[0] = kFootDeltaXLimit;
  d[1] = k * (s.fbk.torso_lin_vel_rel[1] - s.ctrl.torso_lin_vel_d_rel[1]) +
         (1.0 / s.param.gait_freq) / 2.0 * s.ctrl.torso_lin_vel_d_rel[1];
  if (d[1] < -kFootDeltaYLimit) d[1] = -kFootDeltaYLimit;
  if (d[1] > kFootDeltaYLimit) d[1] = kFootDeltaYLimit;
  double dabs[3];
  for (int r = 0; r < 3; ++r)
    dabs[r] = s.fbk.torso_rot_mat_z(r, 0) * d[0] + s.fbk.torso_rot_mat_z(r, 1) * d[1] + s.fbk.torso_rot_mat_z(r, 2) * d[2];
  for (int i = 0; i < 4; ++i) {
    double abs_[3];
    for (int r = 0; r < 3; ++r)   // foot_pos_target_abs = R_z default_foot_pos_rel
      abs_[r] = s.fbk.torso_rot_mat_z(r, 0) * s.param.default_foot_pos_rel(0, i) +
                s.fbk.torso_rot_mat_z(r, 1) * s.param.default_foot_pos_rel(1, i) +
                s.fbk.torso_rot_mat_z(r, 2) * s.param.default_foot_pos_rel(2, i);
    abs_[0] += dabs[0];
    abs_[1] += dabs[1];
    for (int r = 0; r < 3; ++r) {
      s.ctrl.foot_pos_target_abs(r, i) = abs_[r];
      s.ctrl.foot_pos_target_rel(r, i) = s.fbk.torso_rot_mat(0, r) * abs_[0] + s.fbk.torso_rot_mat(1, r) * abs_[1] +
                                         s.fbk.torso_rot_mat(2, r) * abs_[2];
      s.ctrl.foot_pos_target_world(r, i) = abs_[r] + s.fbk.torso_pos_world[r];
    }
  }
}

}  // namespace legged
