// qmpc_joint.hip -- the low-level command of a tick on the device (completed):
// BaseInterface::tau_ctrl_update (legged_ctrl/src/interfaces/BaseInterface.cpp:343-408) for a batch of robots --
// joint angle targets through the closed-form inverse kinematics (A1Kinematics.cpp:335-459), joint velocity targets
// through J^-1, joint torques -J'f -- from the outputs of the MPC tick.  The arithmetic lives in qmpc_joint_math.h
// (shared with the host mirror host/JointCommandsHip.h).
//
// These passes are HBM streaming: 75 doubles in and 36 out per robot (888 B), ~300 flops per leg.  A block owns 64
// consecutive robots; their records move between HBM and LDS as contiguous 8-byte-per-lane accesses (a wave
// instruction covers 512 contiguous bytes), one thread per (robot, leg) -- wave w of the block takes leg w of the 64
// robots -- picks its fields out of LDS.
#pragma once

#include "qmpc_joint_math.h"

namespace qmpc {

static_assert(sizeof(qmpc_joint_feedback) == 75 * sizeof(double), "feedback record");
static_assert(sizeof(qmpc_joint_command) == 36 * sizeof(double), "command record");

constexpr int kJointFb = 75, kJointCmd = 36, kJointTile = 64;

#ifndef QMPC_FUSED_TU
__global__ __launch_bounds__(256) void qmpc_joint_cmd_kernel(LegGeom G, const qmpc_joint_feedback* __restrict__ fb,
                                                             qmpc_joint_command* __restrict__ cmd, int batch) {
  __shared__ double rec[kJointTile * kJointFb];          // 38.4 KB: the feedback records, then the commands
  const int tid = threadIdx.x;
  const size_t r0 = (size_t)blockIdx.x * kJointTile;
  const int n = (int)(((size_t)batch - r0 < (size_t)kJointTile) ? ((size_t)batch - r0) : (size_t)kJointTile);
  const double* src = reinterpret_cast<const double*>(fb + r0);
  for (int i = tid; i < n * kJointFb; i += 256) rec[i] = src[i];
  __syncthreads();
  const int r = tid & 63, l = tid >> 6;     // a wave works on ONE leg index: the left / right branches of the inverse
                                            // kinematics do not diverge inside it
  double ang[3] = {0.0, 0.0, 0.0}, vel[3] = {0.0, 0.0, 0.0}, tau[3] = {0.0, 0.0, 0.0};
  if (r < n) {
    const qmpc_joint_feedback& f = *reinterpret_cast<const qmpc_joint_feedback*>(&rec[r * kJointFb]);
    double R[9];
    qmpc_loop::quat_to_rot(f.torso_quat, R);
    qmpc_joint::leg_command(G.rho_opt[l], G.rho_fix[l], R, f.torso_pos_world, f.torso_lin_vel_world, &f.joint_pos[3 * l],
                            &f.joint_vel[3 * l], &f.foot_pos_target_world[3 * l], &f.foot_vel_target_world[3 * l],
                            &f.forces_body[3 * l], f.plan_contacts[l] != 0.0, f.movement_mode > 0.0, ang, vel, tau);
  }
  __syncthreads();                                       // every record has been read: the tile is reused
  if (r < n) {
    double* o = &rec[r * kJointCmd];
    for (int j = 0; j < 3; ++j) { o[3 * l + j] = ang[j]; o[12 + 3 * l + j] = vel[j]; o[24 + 3 * l + j] = tau[j]; }
  }
  __syncthreads();
  double* dst = reinterpret_cast<double*>(cmd + r0);
  for (int i = tid; i < n * kJointCmd; i += 256) dst[i] = rec[i];
}

// A1Kinematics::inv_kin for every (robot, leg): consecutive threads touch consecutive 24-byte triples
__global__ __launch_bounds__(256) void qmpc_leg_inverse_kernel(LegGeom G, const double* __restrict__ foot_pos_body,
                                                               const double* __restrict__ cur_joint_pos,
                                                               double* __restrict__ joint_pos, int batch) {
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (size_t)batch * 4) return;
  const double p[3] = {foot_pos_body[3 * t], foot_pos_body[3 * t + 1], foot_pos_body[3 * t + 2]};
  double q[3];
  qmpc_joint::leg_inverse(p, cur_joint_pos[3 * t], G.rho_fix[t & 3], q);
  joint_pos[3 * t] = q[0]; joint_pos[3 * t + 1] = q[1]; joint_pos[3 * t + 2] = q[2];
}
#endif

// Joint-level feedback and commands of the robots of a closed loop, from their states after a tick.  The plant's legs
// are massless: the measured joint angles are the inverse kinematics of the plant's foot positions (hip branch: the
// angle of the previous tick; a foot out of reach keeps the previous angles), the joint velocities J^-1 (R'(foot
// velocity - torso velocity) - w x foot_body) with swing feet moving at their FSM target velocity and stance feet at rest.
// One thread per (robot, leg); the loop state is an 820-double record, of which a leg reads ~45.  `cmd` receives the
// commands of this call, `trace` (with the loop's tick counter `row`) one row per tick; either may be null.
// leg l of robot s: joint_pos_io / fb_out / cmd / trace point at THIS robot's records (fb_out, cmd, trace may be null)
__device__ inline void loop_joint_leg(const LegGeom& G, const qmpc_loop_state& s, const int l, double* __restrict__ joint_pos_io,
                                      qmpc_joint_feedback* __restrict__ fb_out, qmpc_joint_command* __restrict__ cmd,
                                      qmpc_joint_command* __restrict__ trace) {
  double R[9];
  qmpc_loop::quat_to_rot(s.quat, R);
  const bool walking = s.movement_mode > 0.0;
  const bool swing = s.movement_mode != 0.0 && s.contacts[l] == 0.0;
  double q[3], qd[3], qprev[3];
  {
#pragma clang fp contract(off)
    const double d[3] = {s.foot_pos_world[3 * l] - s.pos_world[0], s.foot_pos_world[3 * l + 1] - s.pos_world[1],
                         s.foot_pos_world[3 * l + 2] - s.pos_world[2]};
    const double fv[3] = {(swing ? s.leg[l].fsm_vel[0] : 0.0) - s.lin_vel_world[0],
                          (swing ? s.leg[l].fsm_vel[1] : 0.0) - s.lin_vel_world[1],
                          (swing ? s.leg[l].fsm_vel[2] : 0.0) - s.lin_vel_world[2]};
    double pb[3], vb[3];
    for (int a = 0; a < 3; ++a) {
      pb[a] = R[a] * d[0] + R[3 + a] * d[1] + R[6 + a] * d[2];
      vb[a] = R[a] * fv[0] + R[3 + a] * fv[1] + R[6 + a] * fv[2];
    }
    // foot_world = p + R foot_body  =>  d/dt foot_body = R'(v_foot - v_torso) - w x foot_body   (the relation the reference
    // uses the other way round: BaseInterface.cpp:229-231)
    {
      const double* w = s.ang_vel_body;
      const double wx[3] = {w[1] * pb[2] - w[2] * pb[1], w[2] * pb[0] - w[0] * pb[2], w[0] * pb[1] - w[1] * pb[0]};
      for (int a = 0; a < 3; ++a) vb[a] -= wx[a];
    }
    for (int a = 0; a < 3; ++a) qprev[a] = joint_pos_io[3 * l + a];
    qmpc_joint::leg_inverse(pb, qprev[0], G.rho_fix[l], q);
    if ((q[0] != q[0]) || (q[1] != q[1]) || (q[2] != q[2]))
      for (int a = 0; a < 3; ++a) q[a] = qprev[a];
    const qmpc_joint::LegPlane k = qmpc_joint::leg_plane(q, G.rho_opt[l], G.rho_fix[l]);
    double J[9];
    qmpc_joint::leg_jacobian(k, J);
    qmpc_joint::solve3(J, vb, qd);
  }
  double ang[3], vel[3], tau[3];
  qmpc_joint::leg_command(G.rho_opt[l], G.rho_fix[l], R, s.pos_world, s.lin_vel_world, q, qd, &s.foot_target_world[3 * l],
                          s.leg[l].fsm_vel, &s.forces_body[3 * l], s.contacts[l] != 0.0, walking, ang, vel, tau);
  for (int j = 0; j < 3; ++j) joint_pos_io[3 * l + j] = q[j];
  qmpc_joint_command* outs[2] = {cmd, trace};
  for (int w = 0; w < 2; ++w)
    if (outs[w])
      for (int j = 0; j < 3; ++j) {
        outs[w]->joint_ang_tgt[3 * l + j] = ang[j];
        outs[w]->joint_vel_tgt[3 * l + j] = vel[j];
        outs[w]->joint_tau_tgt[3 * l + j] = tau[j];
      }
  if (fb_out) {
    qmpc_joint_feedback& f = *fb_out;
    for (int j = 0; j < 3; ++j) {
      f.joint_pos[3 * l + j] = q[j];
      f.joint_vel[3 * l + j] = qd[j];
      f.foot_pos_target_world[3 * l + j] = s.foot_target_world[3 * l + j];
      f.foot_vel_target_world[3 * l + j] = s.leg[l].fsm_vel[j];
      f.forces_body[3 * l + j] = s.forces_body[3 * l + j];
    }
    f.plan_contacts[l] = s.contacts[l];
    if (l == 0) {
      for (int a = 0; a < 3; ++a) { f.torso_pos_world[a] = s.pos_world[a]; f.torso_lin_vel_world[a] = s.lin_vel_world[a]; }
      for (int a = 0; a < 4; ++a) f.torso_quat[a] = s.quat[a];
      f.movement_mode = s.movement_mode;
    }
  }
}

#ifndef QMPC_FUSED_TU
__global__ __launch_bounds__(256) void qmpc_loop_joint_kernel(LegGeom G, const qmpc_loop_state* __restrict__ st,
                                                              double* __restrict__ joint_pos_io,
                                                              qmpc_joint_feedback* __restrict__ fb_out,
                                                              qmpc_joint_command* __restrict__ cmd,
                                                              qmpc_joint_command* __restrict__ trace,
                                                              const int* __restrict__ row, int batch) {
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (size_t)batch * 4) return;
  const size_t i = t >> 2;
  loop_joint_leg(G, st[i], (int)(t & 3), joint_pos_io + 12 * i, fb_out ? fb_out + i : nullptr, cmd ? cmd + i : nullptr,
                 trace ? trace + ((size_t)(*row) * batch + i) : nullptr);
}
#endif

}  // namespace qmpc
