// ClosedLoopHost.h -- CPU reference of the device-resident closed loop (csrc/qmpc_loop.hip):
// the SAME tick, built from the host classes that mirror the reference (QuatMpcHipT: goal_update / foot_update /
// grf_update, LeggedContactFSMHip, MovingWindowFilterHip, raibert_foot_targets) plus the shared plant of
// csrc/qmpc_loop_math.h.  One instance, one B = 1 solve per tick through the C ABI -- i.e. exactly how the drop-in
// runs inside the reference's mpc_thread -- so that the batched device loop can be checked against it tick for tick.
//
//   tick:  feedback from the plant state  (R, R_z, foot_pos_body, contact flags; BaseInterface::fbk_update)
//          raibert_foot_targets           (BaseInterface.cpp:266-288)
//          mpc.update(state)              (QuatMpc.cpp:57-66)
//          plant step under ctrl.optimized_input[0:12]; swing feet go to their FSM targets
#pragma once

#include "../csrc/qmpc_loop_math.h"
#include <type_traits>

#include "ConvexMpcHip.h"
#include "JointCommandsHip.h"
#include "QuatMpcHip.h"
#include "SwingTrajectoryHip.h"

namespace legged {

// what the test harness drives, whichever controller sits inside
template <class State>
class ClosedLoopHostBase {
 public:
  virtual ~ClosedLoopHostBase() {}
  virtual bool tick() = 0;
  virtual void export_state(qmpc_loop_state* o) const = 0;
  virtual void joint_commands(double* joint_pos_io, qmpc_joint_feedback* fb, qmpc_joint_command* cmd) = 0;
  virtual qmpc_status device_status() const = 0;
  virtual void set_warm_start(bool on) = 0;
  State state;
};

// Mpc: QuatMpcHipT<State> (default) or ConvexMpcHipT<State> -- the two controllers share goal_update / foot_update /
// grf_update, leg_FSM and the outputs the plant consumes (ctrl.optimized_input[0:12], body frame)
template <class State, class Mpc = QuatMpcHipT<State>>
class ClosedLoopHostT : public ClosedLoopHostBase<State> {
 public:
  using ClosedLoopHostBase<State>::state;
  static constexpr bool kConvex = std::is_same<Mpc, ConvexMpcHipT<State>>::value;
  ClosedLoopHostT(const QmpcApi& api, const qmpc_loop_params& lp, const qmpc_loop_state& init, int horizon, int device,
                  int mode = QMPC_MODE_CONVERGED, int drop_ang_vel = 1)
      : lp_(lp) {
    if (kConvex) {                     // gazebo_go1_convex_mpc.yaml: 5 ms, the Euler-angle state's weights, mu 0.6, fz_max 200
      state.param.mpc_update_period = 5.0;
      const double q[13] = {3.0, 3.0, 3.0, 1.0, 1.0, 20.0, 0.0, 0.0, 3.0, 2.0, 3.0, 2.0, 0.0};
      for (int i = 0; i < 13; ++i) state.param.q_weights[i] = q[i];
      state.param.mu = 0.6;
      state.param.fz_max = 200.0;
    }
    state.param.mpc_horizon = horizon;
    state.param.gait_freq = lp.gait_freq;
    for (int l = 0; l < NUM_LEG; ++l)
      for (int a = 0; a < 3; ++a) state.param.default_foot_pos_rel(a, l) = lp.default_foot_pos_rel[3 * l + a];
    for (int a = 0; a < 3; ++a) { x_[a] = init.pos_world[a]; x_[7 + a] = init.lin_vel_world[a]; x_[10 + a] = init.ang_vel_body[a]; }
    for (int a = 0; a < 4; ++a) x_[3 + a] = init.quat[a];
    for (int a = 0; a < 12; ++a) feet_[a] = init.foot_pos_world[a];
    state.joy.velx = init.joy[0]; state.joy.vely = init.joy[1]; state.joy.body_height = init.joy[2];
    state.joy.roll_rate = init.joy[3]; state.joy.pitch_rate = init.joy[4]; state.joy.yaw_rate = init.joy[5];
    state.ctrl.movement_mode = init.movement_mode;
    state.joy.sin_ang_vel = init.sin_ang_vel != 0.0;
    state.ctrl.torso_quat_d.w() = init.quat_d[0]; state.ctrl.torso_quat_d.x() = init.quat_d[1];
    state.ctrl.torso_quat_d.y() = init.quat_d[2]; state.ctrl.torso_quat_d.z() = init.quat_d[3];
    for (int a = 0; a < 3; ++a) state.ctrl.torso_lin_vel_d_rel[a] = init.lin_vel_d_rel[a];
    refresh_feedback();
    state.estimator_init = true;
    make_mpc(api, device, mode, drop_ang_vel);            // QuatMpc takes torso_pos_d_world from the feedback (QuatMpc.cpp:13-20)
    if (kConvex) { state.joy.body_x = init.pos_d_world[0]; state.joy.body_y = init.pos_d_world[1]; }
    qmpc_loop::inv3(mpc->params().inertia, Iinv_);          // the plant of the device loop uses the handle's mass / inertia
  }
  ~ClosedLoopHostT() override { delete mpc; }
  qmpc_status device_status() const override { return mpc->last_status(); }
  void set_warm_start(bool on) override { warm(on); }

  // what BaseInterface::fbk_update derives from the estimator for the fields the tick reads
  void refresh_feedback() {
    double R[9], Rz[9];
    qmpc_loop::quat_to_rot(&x_[3], R);
    qmpc_loop::rot_to_rot_z(R, Rz);
    state.fbk.torso_quat.w() = x_[3]; state.fbk.torso_quat.x() = x_[4]; state.fbk.torso_quat.y() = x_[5]; state.fbk.torso_quat.z() = x_[6];
    for (int r = 0; r < 3; ++r)
      for (int c = 0; c < 3; ++c) { state.fbk.torso_rot_mat(r, c) = R[3 * r + c]; state.fbk.torso_rot_mat_z(r, c) = Rz[3 * r + c]; }
    for (int a = 0; a < 3; ++a) {
      state.fbk.torso_pos_world[a] = x_[a];
      state.fbk.torso_lin_vel_world[a] = x_[7 + a];
      state.fbk.torso_ang_vel_body[a] = x_[10 + a];
    }
    for (int l = 0; l < NUM_LEG; ++l) {
      const double d[3] = {feet_[3 * l] - x_[0], feet_[3 * l + 1] - x_[1], feet_[3 * l + 2] - x_[2]};
      for (int a = 0; a < 3; ++a) {
        state.fbk.foot_pos_world(a, l) = feet_[3 * l + a];
        state.fbk.foot_pos_body(a, l) = R[a] * d[0] + R[3 + a] * d[1] + R[6 + a] * d[2];
      }
      state.fbk.foot_contact_flag[l] = (feet_[3 * l + 2] <= lp_.contact_height) ? 1.0 : 0.0;
    }
    // what ConvexMpc reads on top (BaseInterface.cpp:197-199,217-223)
    double e[3];
    qmpc_loop::quat_to_euler(&x_[3], e);
    for (int a = 0; a < 3; ++a) {
      state.fbk.torso_euler[a] = e[a];
      state.fbk.torso_ang_vel_world[a] = R[3 * a] * x_[10] + R[3 * a + 1] * x_[11] + R[3 * a + 2] * x_[12];
    }
    for (int l = 0; l < NUM_LEG; ++l)
      for (int a = 0; a < 3; ++a)
        state.fbk.foot_pos_abs_com(a, l) = R[3 * a] * state.fbk.foot_pos_body(0, l) + R[3 * a + 1] * state.fbk.foot_pos_body(1, l) +
                                           R[3 * a + 2] * state.fbk.foot_pos_body(2, l);
  }

  bool tick() override {
    refresh_feedback();
    raibert_foot_targets(state);
    mpc->goal_update(state);
    mpc->foot_update(state);
    const bool ok = mpc->grf_update(state);
    double u[12];
    for (int a = 0; a < 12; ++a) u[a] = state.ctrl.optimized_input[a];
    qmpc_loop::plant_step(x_, u, feet_, NUM_LEG, mpc->params().mass, Iinv_, lp_.dt);
    if (state.ctrl.movement_mode != 0)
      for (int l = 0; l < NUM_LEG; ++l)
        if (!state.ctrl.plan_contacts[l])
          for (int a = 0; a < 3; ++a) feet_[3 * l + a] = mpc->leg_FSM[l].FSM_foot_pos_target_world[a];
    ticks_ += 1;
    return ok;
  }

  // Joint level of the tick just made (the plant has massless legs): measured angles = inverse kinematics of the
  // plant's feet (hip branch: joint_pos_io, the angles of the previous call; out of reach keeps them), measured
  // velocities = J^-1 R'(foot velocity - torso velocity), swing feet moving at their FSM target velocity; then
  // BaseInterface::tau_ctrl_update on that feedback.  Fills the records the device entry point produces.
  void joint_commands(double* joint_pos_io, qmpc_joint_feedback* fb, qmpc_joint_command* cmd) override {
    refresh_feedback();
    double R[9];
    qmpc_loop::quat_to_rot(&x_[3], R);
    for (int l = 0; l < NUM_LEG; ++l) {
      const bool swing = state.ctrl.movement_mode != 0 && !state.ctrl.plan_contacts[l];
      double pb[3], fv[3], vb[3], q[3], qd[3], J[9];
      for (int a = 0; a < 3; ++a) {
        pb[a] = state.fbk.foot_pos_body(a, l);
        fv[a] = (swing ? mpc->leg_FSM[l].FSM_foot_vel_target_world[a] : 0.0) - x_[7 + a];
      }
      for (int a = 0; a < 3; ++a) vb[a] = R[a] * fv[0] + R[3 + a] * fv[1] + R[6 + a] * fv[2];
      {   // d/dt foot_body = R'(v_foot - v_torso) - w x foot_body   (BaseInterface.cpp:229-231 read the other way round)
        const double* w = &x_[10];
        const double wx[3] = {w[1] * pb[2] - w[2] * pb[1], w[2] * pb[0] - w[0] * pb[2], w[0] * pb[1] - w[1] * pb[0]};
        for (int a = 0; a < 3; ++a) vb[a] -= wx[a];
      }
      qmpc_joint::leg_inverse(pb, joint_pos_io[3 * l], joints_.geom.rho_fix[l], q);
      if ((q[0] != q[0]) || (q[1] != q[1]) || (q[2] != q[2]))
        for (int a = 0; a < 3; ++a) q[a] = joint_pos_io[3 * l + a];
      qmpc_joint::leg_jacobian(qmpc_joint::leg_plane(q, joints_.geom.rho_opt[l], joints_.geom.rho_fix[l]), J);
      qmpc_joint::solve3(J, vb, qd);
      for (int a = 0; a < 3; ++a) {
        joint_pos_io[3 * l + a] = q[a];
        state.fbk.joint_pos(3 * l + a) = q[a];
        state.fbk.joint_vel(3 * l + a) = qd[a];
      }
    }
    joints_.tau_ctrl_update(state);
    for (int a = 0; a < 12; ++a) {
      fb->joint_pos[a] = state.fbk.joint_pos(a);
      fb->joint_vel[a] = state.fbk.joint_vel(a);
      fb->foot_pos_target_world[a] = state.ctrl.optimized_state(6 + a);
      fb->foot_vel_target_world[a] = state.ctrl.optimized_input(12 + a);
      fb->forces_body[a] = state.ctrl.optimized_input(a);
      cmd->joint_ang_tgt[a] = state.ctrl.joint_ang_tgt(a);
      cmd->joint_vel_tgt[a] = state.ctrl.joint_vel_tgt(a);
      cmd->joint_tau_tgt[a] = state.ctrl.joint_tau_tgt(a);
    }
    for (int a = 0; a < 3; ++a) { fb->torso_pos_world[a] = x_[a]; fb->torso_lin_vel_world[a] = x_[7 + a]; }
    for (int a = 0; a < 4; ++a) { fb->torso_quat[a] = x_[3 + a]; fb->plan_contacts[a] = state.ctrl.plan_contacts[a] ? 1.0 : 0.0; }
    fb->movement_mode = state.ctrl.movement_mode;
  }

  // the state in the device loop's record layout (filter internals are private to the host class: left zero)
  void export_state(qmpc_loop_state* o) const override {
    std::memset(o, 0, sizeof *o);
    for (int a = 0; a < 3; ++a) { o->pos_world[a] = x_[a]; o->lin_vel_world[a] = x_[7 + a]; o->ang_vel_body[a] = x_[10 + a]; }
    for (int a = 0; a < 4; ++a) o->quat[a] = x_[3 + a];
    for (int a = 0; a < 12; ++a) o->foot_pos_world[a] = feet_[a];
    o->joy[0] = state.joy.velx; o->joy[1] = state.joy.vely; o->joy[2] = state.joy.body_height;
    o->joy[3] = state.joy.roll_rate; o->joy[4] = state.joy.pitch_rate; o->joy[5] = state.joy.yaw_rate;
    o->movement_mode = state.ctrl.movement_mode;
    o->sin_ang_vel = state.joy.sin_ang_vel ? 1.0 : 0.0;
    o->attitude_traj_count = sweep_count();
    for (int a = 0; a < 3; ++a) { o->pos_d_world[a] = state.ctrl.torso_pos_d_world[a]; o->lin_vel_d_rel[a] = state.ctrl.torso_lin_vel_d_rel[a]; }
    o->pos_d_init = 1.0;
    o->quat_d[0] = state.ctrl.torso_quat_d.w(); o->quat_d[1] = state.ctrl.torso_quat_d.x();
    o->quat_d[2] = state.ctrl.torso_quat_d.y(); o->quat_d[3] = state.ctrl.torso_quat_d.z();
    for (int l = 0; l < NUM_LEG; ++l) {
      const LeggedContactFSMHip& F = mpc->leg_FSM[l];
      qmpc_loop_leg& L = o->leg[l];
      L.gait_phase = F.phase();
      L.state = (double)F.get_contact_state();
      L.pattern_index = F.pattern_index();
      L.prev_pattern_index = F.prev_pattern_index();
      L.start_time = F.state_start_time();
      L.end_time = F.state_end_time();
      L.not_first_call = F.first_call_done() ? 1.0 : 0.0;
      L.terrain_height = F.terrain_height;
      for (int a = 0; a < 3; ++a) {
        L.swing_start[a] = F.swing_start()[a];
        L.swing_end[a] = F.swing_end()[a];
        L.fsm_pos[a] = F.FSM_foot_pos_target_world[a];
        L.fsm_vel[a] = F.FSM_foot_vel_target_world[a];
        L.fsm_acc[a] = F.FSM_foot_acc_target_world[a];
        o->foot_target_world[3 * l + a] = state.ctrl.optimized_state[6 + 3 * l + a];
      }
      o->contacts[l] = state.ctrl.plan_contacts[l] ? 1.0 : 0.0;
      o->gait_counter[l] = state.ctrl.gait_counter[l];
    }
    for (int a = 0; a < 12; ++a) { o->forces_body[a] = state.ctrl.optimized_input[a]; o->grf_world[a] = state.ctrl.mpc_grf_world[a]; }
    o->status = (double)mpc->last_info().status;
    o->iterations = (double)mpc->last_info().iterations;
    o->tick = (double)ticks_;
  }

  Mpc* mpc = nullptr;

 private:
  template <class M = Mpc>
  typename std::enable_if<std::is_same<M, ConvexMpcHipT<State>>::value>::type make_mpc(const QmpcApi& api, int device, int mode, int) {
    mpc = new Mpc(state, api, device, mode);
  }
  template <class M = Mpc>
  typename std::enable_if<!std::is_same<M, ConvexMpcHipT<State>>::value>::type make_mpc(const QmpcApi& api, int device, int mode,
                                                                                       int drop_ang_vel) {
    mpc = new Mpc(state, api, device, mode, drop_ang_vel);
  }
  template <class M = Mpc>
  typename std::enable_if<std::is_same<M, ConvexMpcHipT<State>>::value>::type warm(bool) {}
  template <class M = Mpc>
  typename std::enable_if<!std::is_same<M, ConvexMpcHipT<State>>::value>::type warm(bool on) { mpc->set_warm_start(on); }
  template <class M = Mpc>
  typename std::enable_if<std::is_same<M, ConvexMpcHipT<State>>::value, double>::type sweep_count() const { return 0.0; }
  template <class M = Mpc>
  typename std::enable_if<!std::is_same<M, ConvexMpcHipT<State>>::value, double>::type sweep_count() const {
    return mpc->attitude_sweep_count();
  }
  qmpc_loop_params lp_;
  JointCommandsHipT<State> joints_;
  double x_[13];
  double feet_[12];
  double Iinv_[9];
  long ticks_ = 0;
};

}  // namespace legged
