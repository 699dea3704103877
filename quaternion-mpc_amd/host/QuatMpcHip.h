// QuatMpcHip.h -- host-side drop-in for legged::QuatMpc
// (legged_ctrl/include/mpc/QuatMpc.h:8-41, src/mpc/QuatMpc.cpp).
//
// Same virtual surface as the reference's seam, the abstract class
// legged::LeggedMpc (legged_ctrl/include/mpc/LeggedMpc.h:21-28):
//     bool update / goal_update / grf_update / foot_update / terrain_update(State&)
// goal_update and foot_update restate the reference's host logic
// (QuatMpc.cpp:68-107, :278-305); grf_update builds the reference trajectory
// inputs exactly as QuatMpc.cpp:112-176 does and then replaces the ALTRO set-up /
// Solve() / GetInput(0) block (QuatMpc.cpp:217-265) by ONE call into the C ABI
// (include/qmpc.h, batch = 1, blocking like the reference's mpc_thread).
//
// State is any type with the reference's field names and Eigen-style element
// access: the reference's own legged::LeggedState inside the ROS controller, or
// legged::LeggedStateLite (LeggedStateLite.h) in the ROS-free harness.
#pragma once

#include <chrono>
#include <cmath>
#include <cstdio>
#include <vector>

#include "../../include/qmpc.h"
#include "LeggedContactFSMHip.h"
#include "MovingWindowFilter.h"

namespace legged {

// The C-ABI entry points the class needs.  Filled by the caller (direct linking
// against libqmpc_hip.so, or dlopen as host_shim.cpp does).
struct QmpcApi {
  void (*default_params)(qmpc_params*, int32_t, int32_t) = nullptr;
  qmpc_status (*create)(const qmpc_params*, int32_t, int32_t, qmpc_handle**) = nullptr;
  qmpc_status (*solve)(qmpc_handle*, int32_t, const qmpc_input*, double*, qmpc_info*) = nullptr;
  void (*destroy)(qmpc_handle*) = nullptr;
  // optional: the warm-started solve (QuatMpcHipT::set_warm_start)
  qmpc_status (*solve_warm)(qmpc_handle*, int32_t, const qmpc_input*, const double*, double*, qmpc_info*, double*) = nullptr;
  // ConvexMpc entry points (only ConvexMpcHipT needs them)
  void (*default_convex_params)(qmpc_params*, int32_t, int32_t) = nullptr;
  qmpc_status (*convex_solve)(qmpc_handle*, int32_t, const qmpc_convex_input*, double*, qmpc_info*) = nullptr;
};

template <class State>
class LeggedMpcHipT {   // LeggedMpc.h:21-28
 public:
  virtual ~LeggedMpcHipT() {}
  virtual bool update(State&) { return true; }
  virtual bool goal_update(State&) { return true; }
  virtual bool grf_update(State&) { return true; }
  virtual bool foot_update(State&) { return true; }
  virtual bool terrain_update(State&) { return true; }
};

template <class State>
class QuatMpcHipT : public LeggedMpcHipT<State> {
 public:
  // mode: QMPC_MODE_CONVERGED (default: the KKT point of the problem the reference poses) or QMPC_MODE_REFERENCE
  // (the reference's own solver mode: AL-iLQR capped at 10 iterations, QuatMpc.cpp:21-26 -- the iterate the robot
  // would have applied upstream)
  // drop_ang_vel = 1 (default) reproduces the reference bit for bit: its x_init never receives fbk.torso_ang_vel_body (a
  // `;` ends the comma initialiser one line early, QuatMpc.cpp:242-245), so the MPC plans from zero angular velocity at
  // every tick.  On the robot the legs damp the body; on an ideal rigid-body plant that leaves the attitude loop without
  // a rate term (robots lose balance after 6-9 s).  0 feeds the measured angular velocity, as evidently meant.
  QuatMpcHipT(State& state, const QmpcApi& api, int device = 0, int mode = QMPC_MODE_CONVERGED, int drop_ang_vel = 1)
      : api_(api), mode_(mode) {   // QuatMpc.cpp:8-55
    for (int i = 0; i < 3; ++i) {
      torso_lin_vel_d_body_filter[i] = MovingWindowFilterHip(100);
      torso_pos_d_body_filter[i] = MovingWindowFilterHip(100);
      state.ctrl.torso_pos_d_world[i] = state.fbk.torso_pos_world[i];
    }
    const double nrm = std::sqrt(state.ctrl.torso_pos_d_world[0] * state.ctrl.torso_pos_d_world[0] +
                                 state.ctrl.torso_pos_d_world[1] * state.ctrl.torso_pos_d_world[1] +
                                 state.ctrl.torso_pos_d_world[2] * state.ctrl.torso_pos_d_world[2]);
    torso_pos_d_world_init = !(nrm < 0.001);
    h = state.param.mpc_update_period;   // [ms]
    horizon = state.param.mpc_horizon;
    for (int i = 0; i < NUM_LEG; ++i) leg_FSM[i].reset_params(state.param.gait_freq, i);
    attitude_traj_count = 0;
    // solver parameters = the fields grf_update passes to ALTRO (QuatMpc.cpp:182,218-229)
    api_.default_params(&params_, horizon, mode_);
    params_.h = static_cast<float>(h / 1000.0);
    params_.h_ref = h / 1000.0;
    params_.mass = state.param.robot_mass;
    for (int r = 0; r < 3; ++r)
      for (int c = 0; c < 3; ++c) params_.inertia[3 * r + c] = 1.2 * state.param.trunk_inertia(r, c);
    for (int i = 0; i < 13; ++i) params_.q_weights[i] = state.param.q_weights(i);
    for (int i = 0; i < 12; ++i) params_.r_weights[i] = state.param.r_weights(i);
    params_.w = state.param.w;
    params_.mu = state.param.mu;
    params_.fz_max = state.param.fz_max;
    params_.drop_ang_vel = drop_ang_vel;
    last_status_ = api_.create ? api_.create(&params_, 1, device, &handle_) : QMPC_NO_DEVICE;
  }
  ~QuatMpcHipT() override {
    if (handle_ && api_.destroy) api_.destroy(handle_);
  }

  bool update(State& state) override {   // QuatMpc.cpp:57-66
    goal_update(state);
    foot_update(state);
    grf_update(state);
    return true;
  }

  bool goal_update(State& state) override {   // QuatMpc.cpp:68-107
    if (state.estimator_init == false) return true;
    if (torso_pos_d_world_init == false) {
      for (int i = 0; i < 3; ++i) state.ctrl.torso_pos_d_world[i] = state.fbk.torso_pos_world[i];
      torso_pos_d_world_init = true;
    }
    state.ctrl.torso_lin_vel_d_rel[0] = state.joy.velx;
    state.ctrl.torso_lin_vel_d_rel[1] = state.joy.vely;
    state.ctrl.torso_lin_vel_d_rel[2] = 0.0;
    double vw[3], vb[3];
    for (int r = 0; r < 3; ++r) {   // torso_rot_mat_z * v_rel
      vw[r] = state.fbk.torso_rot_mat_z(r, 0) * state.ctrl.torso_lin_vel_d_rel[0] +
              state.fbk.torso_rot_mat_z(r, 1) * state.ctrl.torso_lin_vel_d_rel[1] +
              state.fbk.torso_rot_mat_z(r, 2) * state.ctrl.torso_lin_vel_d_rel[2];
    }
    for (int r = 0; r < 3; ++r) state.ctrl.torso_lin_vel_d_world[r] = vw[r];
    for (int r = 0; r < 3; ++r) {   // torso_rot_mat^T * v_world
      vb[r] = state.fbk.torso_rot_mat(0, r) * vw[0] + state.fbk.torso_rot_mat(1, r) * vw[1] +
              state.fbk.torso_rot_mat(2, r) * vw[2];
      state.ctrl.torso_lin_vel_d_body[r] = vb[r];
    }
    for (int i = 0; i < 3; ++i)
      torso_lin_vel_d_body_filtered[i] = torso_lin_vel_d_body_filter[i].CalculateAverage(vb[i]);
    state.ctrl.torso_ang_vel_d_body[0] = state.joy.roll_rate;
    state.ctrl.torso_ang_vel_d_body[1] = state.joy.pitch_rate;
    state.ctrl.torso_ang_vel_d_body[2] = state.joy.yaw_rate;
    // the 5 ms constant is hard-wired upstream, independent of mpc_update_period (:97-98)
    state.ctrl.torso_pos_d_world[0] += state.ctrl.torso_lin_vel_d_world[0] * 5.0 / 1000.0;
    state.ctrl.torso_pos_d_world[1] += state.ctrl.torso_lin_vel_d_world[1] * 5.0 / 1000.0;
    state.ctrl.torso_pos_d_world[2] = state.joy.body_height;
    double dp[3], pb[3];
    for (int i = 0; i < 3; ++i) dp[i] = state.ctrl.torso_pos_d_world[i] - state.fbk.torso_pos_world[i];
    for (int r = 0; r < 3; ++r) {
      pb[r] = state.fbk.torso_rot_mat(0, r) * dp[0] + state.fbk.torso_rot_mat(1, r) * dp[1] +
              state.fbk.torso_rot_mat(2, r) * dp[2];
      state.ctrl.torso_pos_d_body[r] = pb[r];
    }
    for (int i = 0; i < 3; ++i) torso_pos_d_body_filtered[i] = torso_pos_d_body_filter[i].CalculateAverage(pb[i]);
    return true;
  }

  bool foot_update(State& state) override {   // QuatMpc.cpp:278-305
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
        state.ctrl.gait_counter[i] = leg_FSM[i].update(5.0 / 1000.0, state.param.gait_freq, cur, tgt,
                                                       static_cast<bool>(state.fbk.foot_contact_flag[i]));
      }
      for (int i = 0; i < NUM_LEG; ++i) state.ctrl.plan_contacts[i] = leg_FSM[i].get_contact_state();
    }
    return true;
  }

  // Fills the C-ABI record from the state; includes the in/out update of
  // ctrl.torso_quat_d (QuatMpc.cpp:128-137) and the sinusoid test (:140-146).
  void pack_input(State& state, qmpc_input* in) {
    // torso_quat_d += 0.5 G(quat_d) w_d * 5 ms ; normalise   (:128-137)
    double qd[4] = {state.ctrl.torso_quat_d.w(), state.ctrl.torso_quat_d.x(), state.ctrl.torso_quat_d.y(),
                    state.ctrl.torso_quat_d.z()};
    const double wx = state.ctrl.torso_ang_vel_d_body[0], wy = state.ctrl.torso_ang_vel_d_body[1],
                 wz = state.ctrl.torso_ang_vel_d_body[2];
    // G(q) w  (QuaternionUtils.cpp:30-52)
    const double gw[4] = {-qd[1] * wx - qd[2] * wy - qd[3] * wz, qd[0] * wx - qd[3] * wy + qd[2] * wz,
                          qd[3] * wx + qd[0] * wy - qd[1] * wz, -qd[2] * wx + qd[1] * wy + qd[0] * wz};
    for (int i = 0; i < 4; ++i) qd[i] += 0.5 * gw[i] * 5.0 / 1000.0;
    const double n = std::sqrt(qd[0] * qd[0] + qd[1] * qd[1] + qd[2] * qd[2] + qd[3] * qd[3]);
    for (int i = 0; i < 4; ++i) qd[i] = qd[i] / n;
    state.ctrl.torso_quat_d.w() = qd[0];
    state.ctrl.torso_quat_d.x() = qd[1];
    state.ctrl.torso_quat_d.y() = qd[2];
    state.ctrl.torso_quat_d.z() = qd[3];
    if (state.joy.sin_ang_vel) {   // :140-146 (3.14 is the reference's literal)
      const double e = 3.14 / 8 * std::sin(2 * 3.14 / 900 * attitude_traj_count);
      state.ctrl.torso_euler_d[0] = e;
      state.ctrl.torso_euler_d[1] = e;
      state.ctrl.torso_euler_d[2] = e;
      attitude_traj_count += 1;
      // Utils::euler_to_quat (Utils.cpp:75-99)
      const double r = e / 2, p = e / 2, y = e / 2;
      const double cy = std::cos(y), sy = std::sin(y), cp = std::cos(p), sp = std::sin(p), cr = std::cos(r),
                   sr = std::sin(r);
      state.ctrl.torso_quat_d.w() = cy * cp * cr + sy * sp * sr;
      state.ctrl.torso_quat_d.x() = cy * cp * sr - sy * sp * cr;
      state.ctrl.torso_quat_d.y() = cy * sp * cr + sy * cp * sr;
      state.ctrl.torso_quat_d.z() = sy * cp * cr - cy * sp * sr;
    }
    in->quat[0] = state.fbk.torso_quat.w();
    in->quat[1] = state.fbk.torso_quat.x();
    in->quat[2] = state.fbk.torso_quat.y();
    in->quat[3] = state.fbk.torso_quat.z();
    for (int r = 0; r < 3; ++r)
      for (int c = 0; c < 3; ++c) in->rot[3 * r + c] = state.fbk.torso_rot_mat(r, c);
    // torso_lin_vel_body = R^T torso_lin_vel_world, written back like :231
    for (int r = 0; r < 3; ++r) {
      const double v = state.fbk.torso_rot_mat(0, r) * state.fbk.torso_lin_vel_world[0] +
                       state.fbk.torso_rot_mat(1, r) * state.fbk.torso_lin_vel_world[1] +
                       state.fbk.torso_rot_mat(2, r) * state.fbk.torso_lin_vel_world[2];
      state.fbk.torso_lin_vel_body[r] = v;
      in->lin_vel_body[r] = v;
      in->ang_vel_body[r] = state.fbk.torso_ang_vel_body[r];
    }
    for (int l = 0; l < NUM_LEG; ++l) {
      for (int a = 0; a < 3; ++a) in->foot_pos_body[3 * l + a] = state.fbk.foot_pos_body(a, l);
      in->contacts[l] = state.ctrl.plan_contacts[l] ? 1.0 : 0.0;
    }
    for (int a = 0; a < 3; ++a) {
      in->pos_ref_body[a] = torso_pos_d_body_filtered[a];
      in->vel_ref_body[a] = torso_lin_vel_d_body_filtered[a];
      in->acc_ref_body[a] = 0.0;
    }
    in->quat_d[0] = state.ctrl.torso_quat_d.w();
    in->quat_d[1] = state.ctrl.torso_quat_d.x();
    in->quat_d[2] = state.ctrl.torso_quat_d.y();
    in->quat_d[3] = state.ctrl.torso_quat_d.z();
  }

  bool grf_update(State& state) override {   // QuatMpc.cpp:109-276
    const auto t_start = std::chrono::high_resolution_clock::now();
    qmpc_input in;
    pack_input(state, &in);
    double u[12] = {0};
    qmpc_info info;
    info.status = QMPC_NO_DEVICE;
    if (warm_start_ && api_.solve_warm && handle_) {
      // start from last tick's solution (shifted by a knot inside the library) instead of u_ref: same KKT point, about
      // half the iterations with a low params.ipm_mu0; not what the reference does (QuatMpc.cpp:253), hence opt-in
      if (u_prev_.size() != (size_t)(12 * horizon)) u_prev_.assign((size_t)(12 * horizon), 0.0);
      last_status_ = api_.solve_warm(handle_, 1, &in, have_prev_ ? u_prev_.data() : nullptr, u, &info, u_prev_.data());
      have_prev_ = last_status_ == QMPC_OK && (info.status == QMPC_OK || info.status == QMPC_MAX_ITER);
    } else {
      last_status_ = handle_ ? api_.solve(handle_, 1, &in, u, &info) : QMPC_NO_DEVICE;
    }
    last_info_ = info;
    const auto t_end = std::chrono::high_resolution_clock::now();
    state.fbk.mpc_time = std::chrono::duration<double, std::milli>(t_end - t_start).count();   // :257-261
    // The FSM foot targets (:270-272) do not depend on the solve: BaseInterface.cpp:349,358 reads them for every
    // leg while movement_mode > 0 (inverse kinematics of the swing feet), so they are published whatever the
    // solver says -- the reference writes them unconditionally after Solve() (status ignored, :256).
    for (int i = 0; i < NUM_LEG; ++i)
      for (int a = 0; a < 3; ++a) {
        state.ctrl.optimized_state[6 + 3 * i + a] = leg_FSM[i].FSM_foot_pos_target_world[a];
        state.ctrl.optimized_input[12 + 3 * i + a] = leg_FSM[i].FSM_foot_vel_target_world[a];
        state.ctrl.optimized_input[24 + 3 * i + a] = leg_FSM[i].FSM_foot_acc_target_world[a];
      }
    if (last_status_ != QMPC_OK) {
      // fail loudly: the reference ignores SolveStatus (:256); we keep the previous forces
      std::fprintf(stderr, "QuatMpcHip::grf_update: qmpc_solve failed with status %d\n", (int)last_status_);
      return false;
    }
    // Per-instance status: QMPC_MAX_ITER still carries a usable iterate (it is what the reference itself applies,
    // its solver being capped at 10 iterations); every other non-OK word means the forces are zeros or a broken
    // iterate (NAN_INPUT, NO_CONTACT, LINESEARCH_FAIL, NOT_PD): keep the previous forces and say so.
    // (in reference mode a failed line search also leaves the last accepted iterate, which upstream would apply)
    if (info.status != QMPC_OK && info.status != QMPC_MAX_ITER &&
        !(mode_ == QMPC_MODE_REFERENCE && info.status == QMPC_LINESEARCH_FAIL)) {
      std::fprintf(stderr, "QuatMpcHip::grf_update: instance status %d, previous forces kept\n", (int)info.status);
      return false;
    }
    for (int i = 0; i < NUM_LEG; ++i) {   // :267-269
      for (int r = 0; r < 3; ++r) {
        state.ctrl.mpc_grf_world[3 * i + r] = state.fbk.torso_rot_mat(r, 0) * u[3 * i] +
                                              state.fbk.torso_rot_mat(r, 1) * u[3 * i + 1] +
                                              state.fbk.torso_rot_mat(r, 2) * u[3 * i + 2];
        state.ctrl.optimized_input[3 * i + r] = u[3 * i + r];
      }
    }
    return true;
  }

  bool terrain_update(State&) override { return true; }   // commented out upstream (:307-338)

  qmpc_status last_status() const { return last_status_; }
  // warm-started solves from now on (needs QmpcApi::solve_warm); false drops the kept solution
  void set_warm_start(bool on) { warm_start_ = on; have_prev_ = false; }
  double attitude_sweep_count() const { return attitude_traj_count; }   // ticks spent in the sin_ang_vel test mode
  const qmpc_info& last_info() const { return last_info_; }
  const qmpc_params& params() const { return params_; }
  LeggedContactFSMHip leg_FSM[NUM_LEG];

 private:
  QmpcApi api_;
  int mode_ = QMPC_MODE_CONVERGED;
  qmpc_handle* handle_ = nullptr;
  qmpc_params params_;
  qmpc_status last_status_ = QMPC_NO_DEVICE;
  qmpc_info last_info_{};
  bool torso_pos_d_world_init = false;
  MovingWindowFilterHip torso_lin_vel_d_body_filter[3];
  MovingWindowFilterHip torso_pos_d_body_filter[3];
  double torso_lin_vel_d_body_filtered[3] = {0, 0, 0};
  double torso_pos_d_body_filtered[3] = {0, 0, 0};
  double attitude_traj_count = 0;
  bool warm_start_ = false, have_prev_ = false;
  std::vector<double> u_prev_;
  double h = 10.0;
  int horizon = 20;
};

}  // namespace legged
