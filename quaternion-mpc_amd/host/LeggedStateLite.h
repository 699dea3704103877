// LeggedStateLite.h -- ROS-free, Eigen-free mirror of the call-surface types of
// zixinz990/quaternion-mpc: legged::LeggedState = {ctrl, fbk, joy, param,
// estimator_init} (legged_ctrl/include/LeggedState.h:20-261), restricted to the
// fields the quaternion-MPC tick reads or writes, with the
// SAME field names and the same element accessors Eigen offers ((i), (r,c),
// [i], .w()/.x()/.y()/.z(), .setZero()).  QuatMpcHipT<State> (QuatMpcHip.h) is
// written against those accessors only, so it compiles unchanged against the
// reference's own Eigen-based LeggedState inside the ROS controller and against
// this mirror in the ROS-free harness and tests.
#pragma once

#include <cmath>
#include <cstring>

#define NUM_LEG 4   // legged_ctrl/include/LeggedParams.h:9-12
#define LEG_DOF 3
#define NUM_DOF 12

namespace legged {
namespace lite {

// fixed-size column-major matrix (Eigen's default storage order)
template <int R, int C>
struct Mat {
  double d[R * C];
  Mat() { setZero(); }
  void setZero() { std::memset(d, 0, sizeof d); }
  double& operator()(int r, int c) { return d[r + R * c]; }
  const double& operator()(int r, int c) const { return d[r + R * c]; }
  double& operator()(int i) { return d[i]; }
  const double& operator()(int i) const { return d[i]; }
  double& operator[](int i) { return d[i]; }
  const double& operator[](int i) const { return d[i]; }
  double* data() { return d; }
  const double* data() const { return d; }
  double norm() const {
    double s = 0.0;
    for (int i = 0; i < R * C; ++i) s += d[i] * d[i];
    return std::sqrt(s);
  }
};
using Vector3d = Mat<3, 1>;
using Vector4d = Mat<4, 1>;
using Matrix3d = Mat<3, 3>;

struct Quaterniond {
  double w_ = 1.0, x_ = 0.0, y_ = 0.0, z_ = 0.0;
  double& w() { return w_; }
  double& x() { return x_; }
  double& y() { return y_; }
  double& z() { return z_; }
  const double& w() const { return w_; }
  const double& x() const { return x_; }
  const double& y() const { return y_; }
  const double& z() const { return z_; }
  void setIdentity() { w_ = 1.0; x_ = y_ = z_ = 0.0; }
};

}  // namespace lite

struct LeggedFeedbackLite {   // LeggedState.h:20-77
  lite::Vector3d torso_pos_world;
  lite::Quaterniond torso_quat;
  lite::Matrix3d torso_rot_mat;
  lite::Matrix3d torso_rot_mat_z;
  lite::Vector3d torso_lin_vel_world;
  lite::Vector3d torso_lin_vel_body;
  lite::Vector3d torso_lin_vel_rel;            // BaseInterface.cpp:267
  lite::Vector3d torso_ang_vel_body;
  lite::Vector4d foot_contact_flag;
  lite::Mat<LEG_DOF, NUM_LEG> foot_pos_body;
  lite::Mat<LEG_DOF, NUM_LEG> foot_pos_world;
  // read by ConvexMpc only (ConvexMpc.cpp:115-118,156-167)
  lite::Vector3d torso_euler;
  lite::Vector3d torso_ang_vel_world;
  lite::Mat<LEG_DOF, NUM_LEG> foot_pos_abs_com;
  // read by BaseInterface::tau_ctrl_update (BaseInterface.cpp:343-408; JointCommandsHip.h)
  lite::Mat<NUM_DOF, 1> joint_pos;
  lite::Mat<NUM_DOF, 1> joint_vel;
  lite::Mat<LEG_DOF, NUM_DOF> jac_foot;        // LeggedState.h: 3 x 12, one 3 x 3 block per leg
  double mpc_time = 0.0;
};

struct LeggedCtrlLite {       // LeggedState.h:79-125
  lite::Vector4d gait_counter;
  lite::Vector3d torso_pos_d_world;
  lite::Vector3d torso_pos_d_body;
  lite::Vector3d torso_euler_d;
  lite::Quaterniond torso_quat_d;
  lite::Vector3d torso_lin_vel_d_body;
  lite::Vector3d torso_lin_vel_d_rel;
  lite::Vector3d torso_lin_vel_d_world;
  lite::Vector3d torso_ang_vel_d_body;
  lite::Mat<3, NUM_LEG> foot_pos_target_world;
  lite::Mat<3, NUM_LEG> foot_pos_target_abs;   // LeggedState.h:102-104
  lite::Mat<3, NUM_LEG> foot_pos_target_rel;
  bool plan_contacts[NUM_LEG] = {true, true, true, true};
  lite::Mat<6 + 3 * NUM_LEG, 1> optimized_state;
  lite::Mat<9 * NUM_LEG, 1> optimized_input;
  lite::Mat<3 * NUM_LEG, 1> mpc_grf_world;
  lite::Mat<NUM_DOF, 1> joint_ang_tgt;         // written by tau_ctrl_update
  lite::Mat<NUM_DOF, 1> joint_vel_tgt;
  lite::Mat<NUM_DOF, 1> joint_tau_tgt;
  double movement_mode = 0;
};

struct LeggedJoyCmdLite {     // LeggedState.h:127-158
  double velx = 0.0, vely = 0.0, velz = 0.0;
  double pitch_rate = 0.0, roll_rate = 0.0, yaw_rate = 0.0;
  double body_height = 0.05;
  double body_x = 0.0, body_y = 0.0;   // ConvexMpc.cpp:57-58
  bool sin_ang_vel = false;
};

struct LeggedParamLite {      // LeggedState.h:160-244, values of gazebo_go1_quat_mpc.yaml
  double gait_freq = 2.2;
  double mpc_update_period = 10.0;  // [ms]
  int mpc_horizon = 20;
  double w = 50.0;
  lite::Mat<13, 1> q_weights;
  lite::Mat<12, 1> r_weights;
  double robot_mass = 12.84;
  lite::Matrix3d trunk_inertia;
  double mu = 0.7;
  double fz_max = 100.0;
  lite::Mat<3, NUM_LEG> default_foot_pos_rel;  // LeggedState.h:172, yaml default_foot_pos_*
  LeggedParamLite() {
    const double f[4][3] = {{0.20, 0.14, -0.3}, {0.20, -0.14, -0.3}, {-0.20, 0.14, -0.3}, {-0.20, -0.14, -0.3}};
    for (int l = 0; l < NUM_LEG; ++l)
      for (int a = 0; a < 3; ++a) default_foot_pos_rel(a, l) = f[l][a];
    const double q[13] = {2.5, 2.5, 10.0, 0, 0, 0, 0, 0.1, 0.1, 0.1, 0.15, 0.15, 0.15};
    for (int i = 0; i < 13; ++i) q_weights[i] = q[i];
    for (int i = 0; i < 12; ++i) r_weights[i] = 0.000001;
    trunk_inertia(0, 0) = 0.0168128557;
    trunk_inertia(1, 1) = 0.063009565;
    trunk_inertia(2, 2) = 0.0716547275;
  }
};

struct LeggedStateLite {      // LeggedState.h:246-261
  LeggedCtrlLite ctrl;
  LeggedFeedbackLite fbk;
  LeggedJoyCmdLite joy;
  LeggedParamLite param;
  bool estimator_init = false;
};

}  // namespace legged
