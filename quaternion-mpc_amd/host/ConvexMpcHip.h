// ConvexMpcHip.h -- host-side drop-in for legged::ConvexMpc
// (legged_ctrl/include/mpc/ConvexMpc.h:8-37, src/mpc/ConvexMpc.cpp), SURVEY.md 8f rank 1.
//
// Same virtual surface as legged::LeggedMpc (LeggedMpc.h:21-28).  goal_update and
// foot_update restate the reference's host logic (ConvexMpc.cpp:51-79, :200-222);
// grf_update packs the LeggedState fields the reference reads (ConvexMpc.cpp:92-167)
// into one qmpc_convex_input and replaces the ALTRO set-up / Solve() / GetInput(0)
// block (ConvexMpc.cpp:84-187) by ONE call into the C ABI (include/qmpc.h).
// foot_update also publishes the FSM's swing-foot targets (:216-220; SwingTrajectoryHip.h).
#pragma once

#include <chrono>
#include <cstdio>

#include "../../include/qmpc.h"
#include "LeggedContactFSMHip.h"
#include "QuatMpcHip.h"

namespace legged {

template <class State>
class ConvexMpcHipT : public LeggedMpcHipT<State> {
 public:
  // mode: QMPC_MODE_CONVERGED, or QMPC_MODE_REFERENCE = the reference's own solver settings (five AL-iLQR iterations,
  // ConvexMpc.cpp:36-38)
  ConvexMpcHipT(State& state, const QmpcApi& api, int device = 0, int mode = QMPC_MODE_CONVERGED) : api_(api) {   // ConvexMpc.cpp:5-39
    h = state.param.mpc_update_period;   // [ms]
    horizon = state.param.mpc_horizon;
    for (int i = 0; i < NUM_LEG; ++i) leg_FSM[i].reset_params(state.param.gait_freq, i);
    api_.default_convex_params(&params_, horizon, mode);
    params_.h = static_cast<float>(h / 1000.0);    // SetTimeStep(h / 1000.0), float in the callbacks
    params_.h_ref = h / 1000.0;
    // the model's mass and inertia are literals upstream (AltroUtils.cpp:239,270-272): the defaults
    // carry them; u_ref uses param.robot_mass (ConvexMpc.cpp:107), the same 12.84
    params_.mass = state.param.robot_mass;
    for (int i = 0; i < 12; ++i) {
      params_.q_weights[i] = state.param.q_weights(i);
      params_.r_weights[i] = state.param.r_weights(i);
    }
    params_.mu = state.param.mu;
    params_.fz_max = state.param.fz_max;
    last_status_ = api_.create ? api_.create(&params_, 1, device, &handle_) : QMPC_NO_DEVICE;
  }
  ~ConvexMpcHipT() override {
    if (handle_ && api_.destroy) api_.destroy(handle_);
  }

  bool update(State& state) override {   // ConvexMpc.cpp:41-49
    goal_update(state);
    foot_update(state);
    grf_update(state);
    return true;
  }

  bool goal_update(State& state) override {   // ConvexMpc.cpp:51-79
    if (state.estimator_init == false) return true;
    state.ctrl.torso_pos_d_world[0] = state.joy.body_x;
    state.ctrl.torso_pos_d_world[1] = state.joy.body_y;
    state.ctrl.torso_pos_d_world[2] = state.joy.body_height;
    if (state.ctrl.torso_lin_vel_d_rel[0] < state.joy.velx) {
      state.ctrl.torso_lin_vel_d_rel[0] += 1.0 * h / 1000.0;
    } else if (state.ctrl.torso_lin_vel_d_rel[0] > state.joy.velx) {
      state.ctrl.torso_lin_vel_d_rel[0] -= 1.0 * h / 1000.0;
    }
    state.ctrl.torso_lin_vel_d_rel[1] = state.joy.vely;
    state.ctrl.torso_lin_vel_d_rel[2] = 0.0;
    for (int r = 0; r < 3; ++r)
      state.ctrl.torso_lin_vel_d_world[r] = state.fbk.torso_rot_mat_z(r, 0) * state.ctrl.torso_lin_vel_d_rel[0] +
                                            state.fbk.torso_rot_mat_z(r, 1) * state.ctrl.torso_lin_vel_d_rel[1] +
                                            state.fbk.torso_rot_mat_z(r, 2) * state.ctrl.torso_lin_vel_d_rel[2];
    state.ctrl.torso_ang_vel_d_body[2] = state.joy.yaw_rate;
    return true;
  }

  bool foot_update(State& state) override {   // ConvexMpc.cpp:200-222 (schedule part)
    if (state.ctrl.movement_mode == 0) {
      for (int i = 0; i < NUM_LEG; ++i) {
        leg_FSM[i].reset();
        state.ctrl.plan_contacts[i] = true;
      }
    } else {
      for (int i = 0; i < NUM_LEG; ++i) {
        const double cur[3] = {state.fbk.foot_pos_world(0, i), state.fbk.foot_pos_world(1, i), state.fbk.foot_pos_world(2, i)};
        const double tgt[3] = {state.ctrl.foot_pos_target_world(0, i), state.ctrl.foot_pos_target_world(1, i),
                               state.ctrl.foot_pos_target_world(2, i)};
        state.ctrl.gait_counter[i] = leg_FSM[i].update(h / 1000.0, state.param.gait_freq, cur, tgt,
                                                       static_cast<bool>(state.fbk.foot_contact_flag[i]));
      }
      for (int i = 0; i < NUM_LEG; ++i) state.ctrl.plan_contacts[i] = leg_FSM[i].get_contact_state();
    }
    for (int i = 0; i < NUM_LEG; ++i)   // FSM foot targets for the joint-level controller (:216-220)
      for (int a = 0; a < 3; ++a) {
        state.ctrl.optimized_state[6 + 3 * i + a] = leg_FSM[i].FSM_foot_pos_target_world[a];
        state.ctrl.optimized_input[12 + 3 * i + a] = leg_FSM[i].FSM_foot_vel_target_world[a];
        state.ctrl.optimized_input[24 + 3 * i + a] = leg_FSM[i].FSM_foot_acc_target_world[a];
      }
    return true;
  }

  void pack_input(const State& state, qmpc_convex_input* in) const {   // ConvexMpc.cpp:92-118,156-167
    for (int a = 0; a < 3; ++a) {
      in->euler[a] = state.fbk.torso_euler[a];
      in->pos_world[a] = state.fbk.torso_pos_world[a];
      in->ang_vel_world[a] = state.fbk.torso_ang_vel_world[a];
      in->lin_vel_world[a] = state.fbk.torso_lin_vel_world[a];
      in->pos_d_world[a] = state.ctrl.torso_pos_d_world[a];
      in->lin_vel_d_world[a] = state.ctrl.torso_lin_vel_d_world[a];
    }
    for (int l = 0; l < NUM_LEG; ++l) {
      for (int a = 0; a < 3; ++a) in->foot_pos_abs_com[3 * l + a] = state.fbk.foot_pos_abs_com(a, l);
      in->contacts[l] = state.ctrl.plan_contacts[l] ? 1.0 : 0.0;
    }
    in->yaw_rate_d = state.ctrl.torso_ang_vel_d_body[2];
    for (int i = 0; i < 13; ++i) in->reserved[i] = 0.0;
  }

  bool grf_update(State& state) override {   // ConvexMpc.cpp:81-198
    const auto t_start = std::chrono::high_resolution_clock::now();
    qmpc_convex_input in;
    pack_input(state, &in);
    double u[12] = {0};
    qmpc_info info;
    info.status = QMPC_NO_DEVICE;
    last_status_ = (handle_ && api_.convex_solve) ? api_.convex_solve(handle_, 1, &in, u, &info) : QMPC_NO_DEVICE;
    last_info_ = info;
    const auto t_end = std::chrono::high_resolution_clock::now();
    solve_ms_ = std::chrono::duration<double, std::milli>(t_end - t_start).count();   // t_total, :180-183
    if (last_status_ != QMPC_OK) {
      std::fprintf(stderr, "ConvexMpcHip::grf_update: qmpc_convex_solve failed with status %d\n", (int)last_status_);
      return false;
    }
    // (in its own solver mode a failed line search leaves the last accepted iterate, which upstream applies: ConvexMpc.cpp:186-190
    // reads the inputs whatever Solve() returned)
    if (info.status != QMPC_OK && info.status != QMPC_MAX_ITER &&
        !(params_.mode == QMPC_MODE_REFERENCE && info.status == QMPC_LINESEARCH_FAIL)) {   // zero forces / broken iterate: keep the previous ones
      std::fprintf(stderr, "ConvexMpcHip::grf_update: instance status %d, previous forces kept\n", (int)info.status);
      return false;
    }
    for (int i = 0; i < NUM_LEG; ++i)   // optimized_input = R' u_i  (:188-190)
      for (int r = 0; r < 3; ++r)
        state.ctrl.optimized_input[3 * i + r] = state.fbk.torso_rot_mat(0, r) * u[3 * i] +
                                                state.fbk.torso_rot_mat(1, r) * u[3 * i + 1] +
                                                state.fbk.torso_rot_mat(2, r) * u[3 * i + 2];
    for (int a = 0; a < 3; ++a) {       // :192-193
      state.ctrl.optimized_state[a] = state.ctrl.torso_pos_d_world[a];
      state.ctrl.optimized_state[3 + a] = state.ctrl.torso_euler_d[a];
    }
    return true;
  }

  bool terrain_update(State&) override { return true; }   // ConvexMpc.cpp:224-226

  qmpc_status last_status() const { return last_status_; }
  const qmpc_info& last_info() const { return last_info_; }
  const qmpc_params& params() const { return params_; }
  double solve_ms() const { return solve_ms_; }
  LeggedContactFSMHip leg_FSM[NUM_LEG];

 private:
  QmpcApi api_;
  qmpc_handle* handle_ = nullptr;
  qmpc_params params_;
  qmpc_status last_status_ = QMPC_NO_DEVICE;
  qmpc_info last_info_{};
  double solve_ms_ = 0.0;
  double h = 5.0;
  int horizon = 20;
};

}  // namespace legged
