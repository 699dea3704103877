// LeggedLoggerHip.h -- the payload of the reference's /debug topics without ROS.
// legged::LeggedLogger::publish_state (legged_ctrl/include/utils/LeggedLogger.hpp:48-106) fills four
// messages from a LeggedState:
//   /debug/torso_odom     nav_msgs/Odometry        fbk pose + body twist            (:52-69)
//   /debug/torso_odom_d   nav_msgs/Odometry        desired pose + twist (ctrl)      (:72-89)
//   /debug/mpc_grf        sensor_msgs/JointState   name {FL,FR,RL,RR}, position = planned contact (0/1),
//                                                  velocity = 0, effort = |world-frame GRF of the leg|  (:36-47,92-97)
//   /debug/mpc_time       std_msgs/Float64         fbk.mpc_time                     (:100)
// debug_records() produces exactly those numbers in plain structs, field for field, so a ROS node can copy
// them into the message types and existing plots / bags keep their meaning; debug_grf_batch() does the
// mpc_grf part for a batch of solves (world-frame forces as written by QuatMpcHipT::grf_update).
// Written against the accessors of the call surface only (LeggedStateLite.h), so it compiles against the
// reference's Eigen-based LeggedState as well.
#pragma once

#include <cmath>
#include <cstdint>

namespace legged {

struct OdomRecord {          // the fields of nav_msgs/Odometry the logger sets
  double position[3];        // pose.pose.position x y z
  double orientation[4];     // pose.pose.orientation  w x y z  (the order the logger assigns them in)
  double linear[3];          // twist.twist.linear
  double angular[3];         // twist.twist.angular
};

struct MpcGrfRecord {        // sensor_msgs/JointState with 4 entries
  const char* name[4];
  double position[4];        // plan_contacts, 0 or 1
  double velocity[4];        // never written by the reference: zeros
  double effort[4];          // norm of the leg's world-frame force
};

struct DebugRecords {
  OdomRecord torso_odom, torso_odom_d;
  MpcGrfRecord mpc_grf;
  double mpc_time;
};

template <class State>
inline void debug_records(const State& state, DebugRecords& out) {
  for (int a = 0; a < 3; ++a) {
    out.torso_odom.position[a] = state.fbk.torso_pos_world(a);
    out.torso_odom.linear[a] = state.fbk.torso_lin_vel_body(a);
    out.torso_odom.angular[a] = state.fbk.torso_ang_vel_body(a);
    out.torso_odom_d.position[a] = state.ctrl.torso_pos_d_body(a);
    out.torso_odom_d.linear[a] = state.ctrl.torso_lin_vel_d_body(a);
    out.torso_odom_d.angular[a] = state.ctrl.torso_ang_vel_d_body(a);
  }
  out.torso_odom.orientation[0] = state.fbk.torso_quat.w();
  out.torso_odom.orientation[1] = state.fbk.torso_quat.x();
  out.torso_odom.orientation[2] = state.fbk.torso_quat.y();
  out.torso_odom.orientation[3] = state.fbk.torso_quat.z();
  out.torso_odom_d.orientation[0] = state.ctrl.torso_quat_d.w();
  out.torso_odom_d.orientation[1] = state.ctrl.torso_quat_d.x();
  out.torso_odom_d.orientation[2] = state.ctrl.torso_quat_d.y();
  out.torso_odom_d.orientation[3] = state.ctrl.torso_quat_d.z();
  static const char* const kNames[4] = {"FL", "FR", "RL", "RR"};
  for (int i = 0; i < 4; ++i) {
    out.mpc_grf.name[i] = kNames[i];
    out.mpc_grf.position[i] = state.ctrl.plan_contacts[i];
    out.mpc_grf.velocity[i] = 0.0;
    const double fx = state.ctrl.mpc_grf_world(3 * i), fy = state.ctrl.mpc_grf_world(3 * i + 1),
                 fz = state.ctrl.mpc_grf_world(3 * i + 2);
    out.mpc_grf.effort[i] = std::sqrt(fx * fx + fy * fy + fz * fz);
  }
  out.mpc_time = state.fbk.mpc_time;
}

// mpc_grf rows for a batch: forces_world [batch][12], contacts [batch][4] (the qmpc_input field) ->
// position / effort [batch][4]
inline void debug_grf_batch(int32_t batch, const double* forces_world, const double* contacts, double* position,
                            double* effort) {
  for (int32_t b = 0; b < batch; ++b)
    for (int i = 0; i < 4; ++i) {
      const double* f = forces_world + 12 * (size_t)b + 3 * i;
      position[4 * (size_t)b + i] = contacts[4 * (size_t)b + i] != 0.0 ? 1.0 : 0.0;
      effort[4 * (size_t)b + i] = std::sqrt(f[0] * f[0] + f[1] * f[1] + f[2] * f[2]);
    }
}

}  // namespace legged
