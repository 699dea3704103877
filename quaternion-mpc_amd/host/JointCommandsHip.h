// JointCommandsHip.h -- host mirror of the consumer of the MPC outputs, BaseInterface::tau_ctrl_update
// (legged_ctrl/src/interfaces/BaseInterface.cpp:343-408), written against the accessors of LeggedState like the MPC
// classes, so it compiles against the reference's Eigen-based state and against LeggedStateLite.  The per-leg
// arithmetic is csrc/qmpc_joint_math.h, the code the device kernels (csrc/qmpc_joint.hip) run.
//   reads   fbk.joint_pos / joint_vel / torso_rot_mat / torso_pos_world / torso_lin_vel_world,
//           ctrl.optimized_state[6+3i] / optimized_input[3i], [12+3i] / plan_contacts / movement_mode
//   writes  fbk.jac_foot (BaseInterface.cpp:209-212), ctrl.joint_ang_tgt / joint_vel_tgt / joint_tau_tgt
#pragma once

#include "../../include/qmpc.h"
#include "../csrc/qmpc_joint_math.h"

namespace legged {

template <class State>
class JointCommandsHipT {
 public:
  JointCommandsHipT() {                           // BaseInterface.cpp:10-34 (rho_fix / rho_opt lists)
    const double sx[4] = {1, 1, -1, -1}, sy[4] = {1, -1, 1, -1};
    for (int l = 0; l < 4; ++l) {
      geom.rho_fix[l][0] = sx[l] * 0.1881; geom.rho_fix[l][1] = sy[l] * 0.04675; geom.rho_fix[l][2] = sy[l] * 0.0812;
      geom.rho_fix[l][3] = 0.213; geom.rho_fix[l][4] = 0.213;
      for (int a = 0; a < 3; ++a) geom.rho_opt[l][a] = 0.0;
    }
  }
  qmpc_leg_geometry geom;

  bool tau_ctrl_update(State& s) const {
    double R[9], pos[3], vel[3];
    for (int r = 0; r < 3; ++r) {
      for (int c = 0; c < 3; ++c) R[3 * r + c] = s.fbk.torso_rot_mat(r, c);
      pos[r] = s.fbk.torso_pos_world(r);
      vel[r] = s.fbk.torso_lin_vel_world(r);
    }
    for (int l = 0; l < 4; ++l) {
      double q[3], qd[3], pt[3], vt[3], f[3], ang[3], vl[3], tau[3], J[9];
      for (int j = 0; j < 3; ++j) {
        q[j] = s.fbk.joint_pos(3 * l + j);
        qd[j] = s.fbk.joint_vel(3 * l + j);
        pt[j] = s.ctrl.optimized_state(6 + 3 * l + j);
        vt[j] = s.ctrl.optimized_input(12 + 3 * l + j);
        f[j] = s.ctrl.optimized_input(3 * l + j);
      }
      qmpc_joint::leg_jacobian(qmpc_joint::leg_plane(q, geom.rho_opt[l], geom.rho_fix[l]), J);
      for (int c = 0; c < 3; ++c)
        for (int r = 0; r < 3; ++r) s.fbk.jac_foot(r, 3 * l + c) = J[3 * c + r];
      qmpc_joint::leg_command(geom.rho_opt[l], geom.rho_fix[l], R, pos, vel, q, qd, pt, vt, f, s.ctrl.plan_contacts[l],
                              s.ctrl.movement_mode > 0, ang, vl, tau);
      for (int j = 0; j < 3; ++j) {
        s.ctrl.joint_ang_tgt(3 * l + j) = ang[j];
        s.ctrl.joint_vel_tgt(3 * l + j) = vl[j];
        s.ctrl.joint_tau_tgt(3 * l + j) = tau[j];
      }
    }
    return true;
  }
};

}  // namespace legged
