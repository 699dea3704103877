// qmpc_loop.hip -- device-resident closed loop around the MPC solve: the step before the
// path (goal, gait FSM, swing quintic, Raibert footholds, record packing), the path (the solve) and a
// single-rigid-body plant, with every robot's state kept in HBM (qmpc_loop_state, include/qmpc.h).  Two launch forms
// of the same per-robot functions (loop_front_one / loop_front_convex_one, the solve body, loop_post_one):
//   per tick   front kernel -> qmpc_solve_kernel -> post kernel on one stream (one thread per robot in the front / post
//              kernels: a few hundred flops of branchy scalar logic per tick, against ~10^6 for the solve);
//   persistent qmpc_loop_fused_kernel, at the end of this file: a wave owns one robot for all ticks.
//
// Reference arithmetic mirrored here (legged_ctrl/):
//   src/mpc/QuatMpc.cpp:68-107        goal_update              (host twin: host/QuatMpcHip.h)
//   src/mpc/QuatMpc.cpp:278-305       foot_update
//   src/utils/LeggedContactFSM.cpp:11-86,208-246,261-270   gait FSM        (host twin: host/LeggedContactFSMHip.h)
//   src/utils/Utils.cpp:236-293       swing-foot quintic       (host twin: host/SwingTrajectoryHip.h)
//   src/interfaces/BaseInterface.cpp:266-288  Raibert foothold targets
//   include/utils/MovingWindowFilter.hpp:14-63  Neumaier moving average (host twin: host/MovingWindowFilter.h)
//   src/mpc/QuatMpc.cpp:112-176,231-246,263-273  record packing, outputs
//   src/mpc/ConvexMpc.cpp:51-79,92-118,156-167,186-196,200-222   the sibling controller's tick (host twin: host/ConvexMpcHip.h)
// Floating-point contraction is switched off in the controller-side functions: the contact schedule must be
// bit-exact against the host classes (compare-and-add state machine), and g++ does not fuse on the host.
#pragma once

#include "qmpc_device.h"      // DevParams; included after qmpc_kernels.hip by qmpc_hip.hip
#include "qmpc_loop_math.h"

#include <type_traits>

// The controller-side functions of a tick (front end, back end): branchy scalar code on one lane.  They are CALLED, not
// inlined -- in the persistent kernel their temporaries and the 820-double state record competed with the solve body for
// the 256 registers of the two-waves-per-SIMD instantiations (140-244 spilled VGPRs, rounds 2-4); as functions they get
// their own allocation (one copy per occupancy: template parameter OCC, so that the calling kernel's register limit
// carries over) and the save / restore happens once per tick.  The per-tick kernels call the same functions: the two launch
// forms are bit-identical only when both compile the SAME function bodies (inlined into one and called from the other
// they differed in the last bit, round 5).
#define QMPC_LOOP_FN __device__ __attribute__((noinline))
namespace qmpc {

__device__ inline double loop_filter(qmpc_loop_filter& f, double v) {   // MovingWindowFilter.hpp:28-62
#pragma clang fp contract(off)
  auto neumaier = [&](double a) {
#pragma clang fp contract(off)
    const double ns = f.sum + a;
    if (fabs(f.sum) >= fabs(a)) f.correction += (f.sum - ns) + a;
    else f.correction += (a - ns) + f.sum;
    f.sum = ns;
  };
  int head = (int)f.head, count = (int)f.count;
  if (count == QMPC_LOOP_WINDOW) neumaier(-f.ring[head]);
  else ++count;
  neumaier(v);
  f.ring[head] = v;
  head = (head + 1) % QMPC_LOOP_WINDOW;
  f.head = (double)head;
  f.count = (double)count;
  return (f.sum + f.correction) / (double)QMPC_LOOP_WINDOW;
}

// QuinticCurve::get_foot_swing_target (Utils.cpp:236-293): matrix entries in FLOAT (products of the float argument T), powers of t in double (t, T are float
// arguments upstream), the 6x6 solve in double (Gaussian elimination with partial pivoting, three right-hand sides)
__device__ inline void loop_swing_target(float t, float T, const double* start, const double* fin, double* out) {
#pragma clang fp contract(off)
  double Cm[6][6];
  qmpc_loop::swing_condition_matrix(T, Cm);
  const double dx = fin[0] - start[0], dy = fin[1] - start[1];
  const double k = 1.26 / T;
  const double v_xy_mid = k * sqrt(dx * dx + dy * dy);
  const double theta = atan2(fabs(dy), fabs(dx));
  const double v_x_mid = (dx >= 0 ? 1 : -1) * v_xy_mid * cos(theta);
  const double v_y_mid = (dy >= 0 ? 1 : -1) * v_xy_mid * sin(theta);
  double b[6][3] = {{start[0], start[1], start[2]}, {fin[0], fin[1], fin[2]}, {0.0, 0.0, 0.1}, {0.0, 0.0, -0.1},
                    {(start[0] + fin[0]) / 2, (start[1] + fin[1]) / 2, 0.1}, {v_x_mid, v_y_mid, 0.0}};
  for (int col = 0; col < 6; ++col) {
    int piv = col;
    for (int i = col + 1; i < 6; ++i)
      if (fabs(Cm[i][col]) > fabs(Cm[piv][col])) piv = i;
    if (piv != col) {
      for (int j = 0; j < 6; ++j) { const double tmp = Cm[col][j]; Cm[col][j] = Cm[piv][j]; Cm[piv][j] = tmp; }
      for (int r = 0; r < 3; ++r) { const double tmp = b[col][r]; b[col][r] = b[piv][r]; b[piv][r] = tmp; }
    }
    for (int i = col + 1; i < 6; ++i) {
      const double f = Cm[i][col] / Cm[col][col];
      for (int j = col; j < 6; ++j) Cm[i][j] -= f * Cm[col][j];
      for (int r = 0; r < 3; ++r) b[i][r] -= f * b[col][r];
    }
  }
  const double td = t;
  // the polynomial is evaluated left to right with DOUBLE coefficients times the float t (Utils.cpp:263-265: `a_z(2) * t * t`),
  // i.e. every power of t is formed in double; only the entries of the condition matrix are float products of T
  for (int ax = 0; ax < 3; ++ax) {
    double c[6];
    for (int i = 5; i >= 0; --i) {
      double s = b[i][ax];
      for (int j = i + 1; j < 6; ++j) s -= Cm[i][j] * c[j];
      c[i] = s / Cm[i][i];
    }
    out[ax] = c[0] + c[1] * td + c[2] * td * td + c[3] * td * td * td + c[4] * td * td * td * td + c[5] * td * td * td * td * td;
    out[3 + ax] = c[1] + 2 * c[2] * td + 3 * c[3] * td * td + 4 * c[4] * td * td * td + 5 * c[5] * td * td * td * td;
    out[6 + ax] = 2 * c[2] + 6 * c[3] * td + 12 * c[4] * td * td + 20 * c[5] * td * td * td;
  }
}

// trot pattern of set_default_gait_pattern (LeggedContactFSM.cpp:87-108): legs 0,3 = [STANCE, SWING], legs 1,2 =
// [SWING, STANCE]; switch times 0.5, 1.0
__device__ __forceinline__ double loop_pattern(int leg, int idx) {
  const bool first_stance = (leg == 0 || leg == 3);
  return (idx == 0) == first_stance ? 1.0 : 0.0;
}
__device__ __forceinline__ double loop_switch_time(int idx) { return idx == 0 ? 0.5 : 1.0; }

__device__ inline void loop_fsm_common_enter(qmpc_loop_leg& L) {   // :208-223
#pragma clang fp contract(off)
  const int prev = (int)L.pattern_index;
  const int idx = (prev + 1) % 2;
  L.prev_pattern_index = (double)prev;
  L.pattern_index = (double)idx;
  if (idx < prev) L.gait_phase -= 1.0;
  L.start_time = L.gait_phase;
  L.end_time = loop_switch_time(idx);
}
__device__ inline double loop_fsm_percent(const qmpc_loop_leg& L) {   // :261-270
#pragma clang fp contract(off)
  double percent = (L.gait_phase - L.start_time) / (L.end_time - L.start_time);
  if (percent < 0.0) percent = 0.0;
  else if (percent > 1.0) percent = 1.0;
  return percent;
}
__device__ inline void loop_fsm_reset(qmpc_loop_leg& L, int leg) {   // :11-31
  L.gait_phase = 0.0;
  L.pattern_index = 0.0;
  L.prev_pattern_index = 1.0;
  L.start_time = 0.0;
  L.end_time = loop_switch_time(0);
  if (L.state == 0.0) {
    for (int a = 0; a < 3; ++a) { L.fsm_pos[a] = L.swing_end[a]; L.fsm_vel[a] = 0.0; }
  }
  L.state = loop_pattern(leg, 0);
  L.not_first_call = 0.0;
}
__device__ inline double loop_fsm_update(qmpc_loop_leg& L, double dt, double gait_freq, const double* cur,
                                         const double* tgt, bool flag) {   // :33-78
#pragma clang fp contract(off)
  if (L.not_first_call == 0.0) {
    for (int a = 0; a < 3; ++a) {
      L.swing_start[a] = cur[a];
      L.swing_end[a] = tgt[a];
      L.fsm_pos[a] = tgt[a];
      L.fsm_vel[a] = 0.0;
    }
    L.not_first_call = 1.0;
  }
  L.gait_phase += gait_freq * dt;
  if (L.state == 1.0) {
    if (L.gait_phase >= L.end_time) {
      L.terrain_height = cur[2];                        // stance_exit
      loop_fsm_common_enter(L);                         // swing_enter
      for (int a = 0; a < 3; ++a) { L.swing_start[a] = cur[a]; L.swing_extend[a] = 0.0; }
      L.state = 0.0;
    }
  } else {
    if ((loop_fsm_percent(L) > 0.9 && flag) || loop_fsm_percent(L) >= 1.0) {
      L.state = 1.0;
      loop_fsm_common_enter(L);                         // stance_enter
      for (int a = 0; a < 3; ++a) { L.fsm_pos[a] = cur[a]; L.fsm_vel[a] = 0.0; }
    }
  }
  if (L.state == 0.0) {                                 // swing_update
    const double t = loop_fsm_percent(L);
    double fin[3], out[9];
    for (int a = 0; a < 3; ++a) fin[a] = tgt[a] + L.swing_extend[a];
    loop_swing_target((float)(0.5 * t / gait_freq), (float)(0.5 / gait_freq), L.swing_start, fin, out);
    for (int a = 0; a < 3; ++a) { L.fsm_pos[a] = out[a]; L.fsm_vel[a] = out[3 + a]; L.fsm_acc[a] = out[6 + a]; }
  }
  return L.gait_phase;
}

// ---- front end of one tick: feedback, Raibert, goal_update, foot_update, record ---------------
// feedback the controller reads (BaseInterface::fbk_update): R, R_z, foot_pos_body, contact flags
__device__ inline void loop_feedback(const qmpc_loop_params& LP, const qmpc_loop_state& s, double* R, double* Rz,
                                     double* foot_body, double* flag) {
#pragma clang fp contract(off)
  qmpc_loop::quat_to_rot(s.quat, R);
  qmpc_loop::rot_to_rot_z(R, Rz);
  for (int l = 0; l < 4; ++l) {
    const double d[3] = {s.foot_pos_world[3 * l] - s.pos_world[0], s.foot_pos_world[3 * l + 1] - s.pos_world[1],
                         s.foot_pos_world[3 * l + 2] - s.pos_world[2]};
    for (int a = 0; a < 3; ++a) foot_body[3 * l + a] = R[a] * d[0] + R[3 + a] * d[1] + R[6 + a] * d[2];
    flag[l] = (s.foot_pos_world[3 * l + 2] <= LP.contact_height) ? 1.0 : 0.0;
  }
}
// Raibert foothold targets (BaseInterface.cpp:266-288), with last tick's ctrl.torso_lin_vel_d_rel
__device__ inline void loop_raibert(const qmpc_loop_params& LP, const qmpc_loop_state& s, const double* Rz, double* tgt_world) {
#pragma clang fp contract(off)
  double vrel[3];
  for (int r = 0; r < 3; ++r) vrel[r] = Rz[r] * s.lin_vel_world[0] + Rz[3 + r] * s.lin_vel_world[1] + Rz[6 + r] * s.lin_vel_world[2];
  const double k = sqrt(fabs(s.pos_world[2]) / 9.81);
  double d[3] = {0.0, 0.0, 0.0};
  d[0] = k * (vrel[0] - s.lin_vel_d_rel[0]) + (1.0 / LP.gait_freq) / 2.0 * s.lin_vel_d_rel[0];
  if (d[0] < -0.5) d[0] = -0.5;
  if (d[0] > 0.5) d[0] = 0.5;
  d[1] = k * (vrel[1] - s.lin_vel_d_rel[1]) + (1.0 / LP.gait_freq) / 2.0 * s.lin_vel_d_rel[1];
  if (d[1] < -0.3) d[1] = -0.3;
  if (d[1] > 0.3) d[1] = 0.3;
  double dabs[3];
  for (int r = 0; r < 3; ++r) dabs[r] = Rz[3 * r] * d[0] + Rz[3 * r + 1] * d[1] + Rz[3 * r + 2] * d[2];
  for (int l = 0; l < 4; ++l) {
    double ab[3];
    for (int r = 0; r < 3; ++r)
      ab[r] = Rz[3 * r] * LP.default_foot_pos_rel[3 * l] + Rz[3 * r + 1] * LP.default_foot_pos_rel[3 * l + 1] +
              Rz[3 * r + 2] * LP.default_foot_pos_rel[3 * l + 2];
    ab[0] += dabs[0];
    ab[1] += dabs[1];
    for (int r = 0; r < 3; ++r) tgt_world[3 * l + r] = ab[r] + s.pos_world[r];
  }
}
// foot_update (QuatMpc.cpp:278-305, ConvexMpc.cpp:200-222) and the published foot targets (QuatMpc.cpp:270)
__device__ inline void loop_foot_update(const qmpc_loop_params& LP, qmpc_loop_state& s, const double* tgt_world, const double* flag) {
#pragma clang fp contract(off)
  if (s.movement_mode == 0.0) {
    for (int l = 0; l < 4; ++l) {
      loop_fsm_reset(s.leg[l], l);
      s.contacts[l] = 1.0;
    }
  } else {
    for (int l = 0; l < 4; ++l)
      s.gait_counter[l] = loop_fsm_update(s.leg[l], 5.0 / 1000.0, LP.gait_freq, &s.foot_pos_world[3 * l], &tgt_world[3 * l],
                                          flag[l] != 0.0);
    for (int l = 0; l < 4; ++l) s.contacts[l] = s.leg[l].state;
  }
  for (int l = 0; l < 4; ++l)
    for (int a = 0; a < 3; ++a) s.foot_target_world[3 * l + a] = s.leg[l].fsm_pos[a];
}

template <int OCC = 1>
QMPC_LOOP_FN void loop_front_one(const qmpc_loop_params& LP, qmpc_loop_state& s, qmpc_input& in) {
#pragma clang fp contract(off)
  double R[9], Rz[9], foot_body[12], flag[4], tgt_world[12];
  loop_feedback(LP, s, R, Rz, foot_body, flag);
  loop_raibert(LP, s, Rz, tgt_world);
  // goal_update (QuatMpc.cpp:68-107)
  double vel_f[3], pos_f[3], wd[3];
  {
    if (s.pos_d_init == 0.0) {
      for (int a = 0; a < 3; ++a) s.pos_d_world[a] = s.pos_world[a];
      s.pos_d_init = 1.0;
    }
    s.lin_vel_d_rel[0] = s.joy[0];
    s.lin_vel_d_rel[1] = s.joy[1];
    s.lin_vel_d_rel[2] = 0.0;
    double vw[3], vb[3];
    for (int r = 0; r < 3; ++r)
      vw[r] = Rz[3 * r] * s.lin_vel_d_rel[0] + Rz[3 * r + 1] * s.lin_vel_d_rel[1] + Rz[3 * r + 2] * s.lin_vel_d_rel[2];
    for (int r = 0; r < 3; ++r) vb[r] = R[r] * vw[0] + R[3 + r] * vw[1] + R[6 + r] * vw[2];
    for (int a = 0; a < 3; ++a) vel_f[a] = loop_filter(s.vel_filter[a], vb[a]);
    wd[0] = s.joy[3]; wd[1] = s.joy[4]; wd[2] = s.joy[5];
    s.pos_d_world[0] += vw[0] * 5.0 / 1000.0;
    s.pos_d_world[1] += vw[1] * 5.0 / 1000.0;
    s.pos_d_world[2] = s.joy[2];
    double dp[3], pb[3];
    for (int a = 0; a < 3; ++a) dp[a] = s.pos_d_world[a] - s.pos_world[a];
    for (int r = 0; r < 3; ++r) pb[r] = R[r] * dp[0] + R[3 + r] * dp[1] + R[6 + r] * dp[2];
    for (int a = 0; a < 3; ++a) pos_f[a] = loop_filter(s.pos_filter[a], pb[a]);
  }
  loop_foot_update(LP, s, tgt_world, flag);
  // record (QuatMpc.cpp:112-176,231-246): torso_quat_d += 1/2 G(quat_d) w_d 5 ms, normalised
  {
    double* qd = s.quat_d;
    const double gw[4] = {-qd[1] * wd[0] - qd[2] * wd[1] - qd[3] * wd[2], qd[0] * wd[0] - qd[3] * wd[1] + qd[2] * wd[2],
                          qd[3] * wd[0] + qd[0] * wd[1] - qd[1] * wd[2], -qd[2] * wd[0] + qd[1] * wd[1] + qd[0] * wd[2]};
    for (int a = 0; a < 4; ++a) qd[a] += 0.5 * gw[a] * 5.0 / 1000.0;
    const double n = sqrt(qd[0] * qd[0] + qd[1] * qd[1] + qd[2] * qd[2] + qd[3] * qd[3]);
    for (int a = 0; a < 4; ++a) qd[a] = qd[a] / n;
    if (s.sin_ang_vel != 0.0) {      // the attitude-sweep test mode (QuatMpc.cpp:138-146; 3.14 is the reference's literal)
      const double e = 3.14 / 8 * sin(2 * 3.14 / 900 * s.attitude_traj_count);
      s.attitude_traj_count += 1.0;
      const double hr = e / 2;       // Utils::euler_to_quat (Utils.cpp:75-99) of (e, e, e)
      const double c = cos(hr), sn = sin(hr);
      qd[0] = c * c * c + sn * sn * sn;
      qd[1] = c * c * sn - sn * sn * c;
      qd[2] = c * sn * c + sn * c * sn;
      qd[3] = sn * c * c - c * sn * sn;
    }
  }
  for (int a = 0; a < 4; ++a) { in.quat[a] = s.quat[a]; in.quat_d[a] = s.quat_d[a]; in.contacts[a] = (s.contacts[a] != 0.0) ? 1.0 : 0.0; }
  for (int a = 0; a < 9; ++a) in.rot[a] = R[a];
  for (int r = 0; r < 3; ++r) {
    in.lin_vel_body[r] = R[r] * s.lin_vel_world[0] + R[3 + r] * s.lin_vel_world[1] + R[6 + r] * s.lin_vel_world[2];
    in.ang_vel_body[r] = s.ang_vel_body[r];
    in.pos_ref_body[r] = pos_f[r];
    in.vel_ref_body[r] = vel_f[r];
    in.acc_ref_body[r] = 0.0;
  }
  for (int a = 0; a < 12; ++a) in.foot_pos_body[a] = foot_body[a];
}
// ---- the same for the sibling controller, ConvexMpc (ConvexMpc.cpp:41-79,92-118,156-167,200-222) -------------------
// feedback: torso_euler (Utils::quat_to_euler), torso_ang_vel_world = R w, foot_pos_abs_com = R foot_pos_body
// (BaseInterface.cpp:197-199,217-223); goal_update: the desired position is the joystick's (body_x, body_y, height) --
// kept in pos_d_world[0:2], which this controller never integrates -- and velx ramps at 1 m/s^2; the gait FSM, the
// Raibert targets and the published foot targets are those of QuatMpc.
template <int OCC = 1>
QMPC_LOOP_FN void loop_front_convex_one(const qmpc_loop_params& LP, qmpc_loop_state& s, qmpc_convex_input& in) {
#pragma clang fp contract(off)
  double R[9], Rz[9], foot_body[12], flag[4], tgt_world[12];
  loop_feedback(LP, s, R, Rz, foot_body, flag);
  loop_raibert(LP, s, Rz, tgt_world);
  // goal_update (ConvexMpc.cpp:51-79)
  s.pos_d_world[2] = s.joy[2];
  if (s.lin_vel_d_rel[0] < s.joy[0]) s.lin_vel_d_rel[0] += 1.0 * 5.0 / 1000.0;
  else if (s.lin_vel_d_rel[0] > s.joy[0]) s.lin_vel_d_rel[0] -= 1.0 * 5.0 / 1000.0;
  s.lin_vel_d_rel[1] = s.joy[1];
  s.lin_vel_d_rel[2] = 0.0;
  double vw[3];
  for (int r = 0; r < 3; ++r)
    vw[r] = Rz[3 * r] * s.lin_vel_d_rel[0] + Rz[3 * r + 1] * s.lin_vel_d_rel[1] + Rz[3 * r + 2] * s.lin_vel_d_rel[2];
  loop_foot_update(LP, s, tgt_world, flag);
  // record (ConvexMpc.cpp:92-118,156-167)
  qmpc_loop::quat_to_euler(s.quat, in.euler);
  for (int r = 0; r < 3; ++r) {
    in.pos_world[r] = s.pos_world[r];
    in.ang_vel_world[r] = R[3 * r] * s.ang_vel_body[0] + R[3 * r + 1] * s.ang_vel_body[1] + R[3 * r + 2] * s.ang_vel_body[2];
    in.lin_vel_world[r] = s.lin_vel_world[r];
    in.pos_d_world[r] = s.pos_d_world[r];
    in.lin_vel_d_world[r] = vw[r];
  }
  for (int l = 0; l < 4; ++l) {
    for (int r = 0; r < 3; ++r)
      in.foot_pos_abs_com[3 * l + r] = R[3 * r] * foot_body[3 * l] + R[3 * r + 1] * foot_body[3 * l + 1] + R[3 * r + 2] * foot_body[3 * l + 2];
    in.contacts[l] = (s.contacts[l] != 0.0) ? 1.0 : 0.0;
  }
  in.yaw_rate_d = s.joy[5];
  for (int a = 0; a < 13; ++a) in.reserved[a] = 0.0;
}

#ifndef QMPC_FUSED_TU
__global__ __launch_bounds__(64) void qmpc_loop_front_convex_kernel(qmpc_loop_params LP, qmpc_loop_state* __restrict__ st,
                                                                    qmpc_convex_input* __restrict__ rec, int* __restrict__ row,
                                                                    int batch) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i == 0 && row) *row += 1;
  if (i >= batch) return;
  loop_front_convex_one(LP, st[i], rec[i]);
}
__global__ __launch_bounds__(64) void qmpc_loop_front_kernel(qmpc_loop_params LP, qmpc_loop_state* __restrict__ st,
                                                             qmpc_input* __restrict__ rec, int* __restrict__ row,
                                                             int batch) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i == 0 && row) *row += 1;                         // trace row of this tick (stream order: after the last post)
  if (i >= batch) return;
  loop_front_one(LP, st[i], rec[i]);
}
#endif

// ---- back end of one tick: outputs (QuatMpc.cpp:263-273), plant step, swing feet -----------------
// forces: the 12 forces of this robot's solve; trace_f / trace_c: this robot's slots in the trace row of this tick (or null)
// WORLD: the solve returns WORLD-frame forces (ConvexMpc); optimized_input = R' u (ConvexMpc.cpp:188-190)
template <bool WORLD = false, int OCC = 1>
QMPC_LOOP_FN void loop_post_one(const DevParams& P, const qmpc_loop_params& LP, qmpc_loop_state& s,
                                     const double* __restrict__ forces, const qmpc_info& inf, double* __restrict__ trace_f,
                                     double* __restrict__ trace_c) {
#pragma clang fp contract(off)
  const int status = inf.status;
  s.status = (double)status;
  s.iterations = (double)inf.iterations;
  const bool accepted = status == QMPC_OK || status == QMPC_MAX_ITER ||   // otherwise the previous forces stay (host classes);
                        (P.mode == QMPC_MODE_REFERENCE && status == QMPC_LINESEARCH_FAIL);   // the reference applies its last iterate
  double R[9];
  qmpc_loop::quat_to_rot(s.quat, R);
  if (WORLD) {
    if (accepted)
      for (int l = 0; l < 4; ++l)
        for (int r = 0; r < 3; ++r) {
          s.grf_world[3 * l + r] = forces[3 * l + r];
          s.forces_body[3 * l + r] = R[r] * forces[3 * l] + R[3 + r] * forces[3 * l + 1] + R[6 + r] * forces[3 * l + 2];
        }
  } else {
    if (accepted)
      for (int a = 0; a < 12; ++a) s.forces_body[a] = forces[a];
    for (int l = 0; l < 4; ++l)
      for (int r = 0; r < 3; ++r)
        s.grf_world[3 * l + r] = R[3 * r] * s.forces_body[3 * l] + R[3 * r + 1] * s.forces_body[3 * l + 1] +
                                 R[3 * r + 2] * s.forces_body[3 * l + 2];
  }
  if (trace_f) for (int a = 0; a < 12; ++a) trace_f[a] = s.forces_body[a];
  if (trace_c) for (int a = 0; a < 4; ++a) trace_c[a] = s.contacts[a];
  // plant: rigid body under the applied forces, feet fixed during the step
  double x[13];
  for (int a = 0; a < 3; ++a) { x[a] = s.pos_world[a]; x[7 + a] = s.lin_vel_world[a]; x[10 + a] = s.ang_vel_body[a]; }
  for (int a = 0; a < 4; ++a) x[3 + a] = s.quat[a];
  qmpc_loop::plant_step(x, s.forces_body, s.foot_pos_world, 4, P.mass, P.Iinv, LP.dt);
  for (int a = 0; a < 3; ++a) { s.pos_world[a] = x[a]; s.lin_vel_world[a] = x[7 + a]; s.ang_vel_body[a] = x[10 + a]; }
  for (int a = 0; a < 4; ++a) s.quat[a] = x[3 + a];
  // swing feet track their FSM target perfectly; stance feet stay where they are
  if (s.movement_mode != 0.0)
    for (int l = 0; l < 4; ++l)
      if (s.contacts[l] == 0.0)
        for (int a = 0; a < 3; ++a) s.foot_pos_world[3 * l + a] = s.leg[l].fsm_pos[a];
  s.tick += 1.0;
}
#ifndef QMPC_FUSED_TU
template <bool WORLD>
__global__ __launch_bounds__(64) void qmpc_loop_post_kernel(DevParams P, qmpc_loop_params LP, qmpc_loop_state* __restrict__ st,
                                                            const double* __restrict__ forces,
                                                            const qmpc_info* __restrict__ info, double* __restrict__ trace_f,
                                                            double* __restrict__ trace_c, const int* __restrict__ row,
                                                            int batch) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= batch) return;
  const size_t slot = (trace_f || trace_c) ? (size_t)(*row) * batch + i : 0;
  loop_post_one<WORLD>(P, LP, st[i], forces + 12 * (size_t)i, info[i], trace_f ? trace_f + 12 * slot : nullptr,
                       trace_c ? trace_c + 4 * slot : nullptr);
}
#endif

#ifdef QMPC_FUSED_TU      // compiled in its own translation unit (qmpc_loop_fused.hip)
}  // namespace qmpc
#include "qmpc_wform.h"     // VAR == 3: the solve in the wrench form (QuatMpc's problem, everything in LDS)
namespace qmpc {
// ---- the whole loop in ONE launch: a persistent wave per robot ------------------------------------------------
// The robots of a batch do not interact, so nothing forces them to advance in lock-step: with one launch sequence per
// tick every tick lasts as long as its slowest solve (mean / max of the iteration counts ~0.63 at 1024 robots), and
// that tail is paid `ticks` times.  Here a wave owns one robot for all ticks -- front end (lane 0), solve (the wave,
// solve_body of qmpc_kernels.hip), back end (lane 0) -- and the tails of different robots average out instead of
// adding up.  Same arithmetic in the same order as the per-tick kernels: bit-identical states and traces.
// It lives in a translation unit of its own because a second user of the solve body in THIS one changes the
// compiler's inlining of the solve kernel (measured: 2 % slower contract workload).
struct FusedJoint {          // the joint level closing every tick (JOINT instantiations)
  LegGeom G;
  double* joint_pos;
  qmpc_joint_command* cmd;
  qmpc_joint_command* trace;
};
// VAR: 0 / 1 / 2 as in qmpc_solve_kernel; 3 / 5: the wrench-form body (qmpc_wform_body.inc; QuatMpc's problem, converged
// mode) with everything in LDS / with its gains in the workspace
// REF: the reference's own solver mode (AL-iLQR, <= 10 iterations; one wave per SIMD like qmpc_ref_kernel, VAR 0 / 1)
// CONVEX: the sibling controller's problem (ConvexModel; its own solver mode on the wrench-form bodies 3 / 5 only)
template <int VAR, bool JOINT, bool REF, bool CONVEX = false>
__global__ __launch_bounds__(64, (REF && VAR != 5) ? 1 : QMPC_SOLVE_WAVES(QuatModel, VAR)) void qmpc_loop_fused_kernel(
    DevParams P, qmpc_loop_params LP, qmpc_loop_state* __restrict__ st, qmpc_input* __restrict__ rec,
    double* __restrict__ forces, qmpc_info* __restrict__ info, double* __restrict__ trace_f, double* __restrict__ trace_c,
    int ticks, int batch, double* __restrict__ gws, FusedJoint JL) {
  extern __shared__ __attribute__((aligned(16))) double sm[];
  const int b = blockIdx.x;
  if (b >= batch) return;
  const int lane = threadIdx.x;
  typedef typename std::conditional<CONVEX, ConvexModel, QuatModel>::type MD;
  constexpr bool PROF = false;
  // the called front / back end: one copy per occupancy, so that the register limit of the calling kernel carries over
  constexpr int OCC = (REF && VAR != 5) ? 1 : QMPC_SOLVE_WAVES(QuatModel, VAR);
  const qmpc_input* in_ = rec;
  double *traj_u = nullptr, *traj_x = nullptr;
  long long* prof_out = nullptr;
  bool prev_ok = false;
  for (int t = 0; t < ticks; ++t) {
    if (lane == 0) {
      if (CONVEX) loop_front_convex_one<OCC>(LP, st[b], reinterpret_cast<qmpc_convex_input*>(rec)[b]);
      else loop_front_one<OCC>(LP, st[b], rec[b]);
    }
    __syncthreads();                      // the record (global memory) is visible to the wave
    if constexpr (REF && (VAR == 3 || VAR == 5) && CONVEX) {
      [&]() {                             // ConvexMpc's own solver mode (five AL-iLQR iterations) on the wrench-form algebra
        constexpr int WVAR = VAR;
        const int wslot = b;
#define QMPC_WMODEL WM_CONVEX
#include "qmpc_wform_ref_body.inc"
#undef QMPC_WMODEL
      }();
    } else if constexpr (REF && (VAR == 3 || VAR == 5)) {
      [&]() {                             // the reference's solver mode on the wrench-form algebra
        constexpr int WVAR = VAR;
        const int wslot = b;
#include "qmpc_wform_ref_body.inc"
      }();
    } else if constexpr (REF) {
      [&]() {                             // `return` in the body (rejected input) ends this tick's solve only
#include "qmpc_ref_body.inc"
      }();
    } else if constexpr ((VAR == 3 || VAR == 5 || VAR == 6) && CONVEX) {
      [&]() {                             // ConvexMpc's problem on the wrench-form body (round 5)
        const int warm_t = (LP.warm_start != 0.0 && prev_ok) ? t : 0;
        constexpr int WVAR = VAR;
        const int wslot = b;
        constexpr const double* resume = nullptr;
#define QMPC_WMODEL WM_CONVEX
#include "qmpc_wform_body.inc"
#undef QMPC_WMODEL
      }();
    } else if constexpr (VAR == 3 || VAR == 5 || VAR == 6) {
      [&]() {
        const int warm_t = (LP.warm_start != 0.0 && prev_ok) ? t : 0;   // t > 0 and the last solve left a usable U in LDS
        constexpr int WVAR = VAR;
        const int wslot = b;
        constexpr const double* resume = nullptr;
#include "qmpc_wform_body.inc"
      }();
    } else {
      [&]() {
        const int warm_t = (LP.warm_start != 0.0 && prev_ok) ? t : 0;   // t > 0 and the last solve left a usable U in LDS
#include "qmpc_solve_body.inc"
      }();
    }
    __syncthreads();
    prev_ok = info[b].status == QMPC_OK || info[b].status == QMPC_MAX_ITER;   // uniform: every lane reads the same word
    if (lane == 0) {
      const size_t slot = (size_t)t * batch + b;
      loop_post_one<CONVEX, OCC>(P, LP, st[b], forces + 12 * (size_t)b, info[b], trace_f ? trace_f + 12 * slot : nullptr,
                                 trace_c ? trace_c + 4 * slot : nullptr);
    }
    __syncthreads();
    if (JOINT) {                          // the joint level of the tick (qmpc_joint.hip): one lane per leg
      if (lane < 4)
        loop_joint_leg(JL.G, st[b], lane, JL.joint_pos + 12 * (size_t)b, nullptr, JL.cmd ? JL.cmd + b : nullptr,
                       JL.trace ? JL.trace + ((size_t)t * batch + b) : nullptr);
      __syncthreads();
    }
  }
}


// ---- a single warm-started solve (qmpc_solve_warm*): the drop-in class's per-tick call ---------------------------------
// Same body; the start is u_init (the caller's previous solution of the same robot, e.g. last tick's traj_u; it may be the
// buffer traj_u is written to) shifted by one knot instead of u_ref.  u_init == nullptr: a plain cold solve.  In this translation unit for the reason given above.
template <int VAR, bool CONVEX = false>
__global__ __launch_bounds__(64, QMPC_SOLVE_WAVES(QuatModel, VAR)) void qmpc_solve_warm_kernel(
    DevParams P, const qmpc_input* __restrict__ in_, const double* u_init, double* __restrict__ forces,
    qmpc_info* __restrict__ info, double* traj_u, int batch, double* __restrict__ gws, int check_prev) {
  extern __shared__ __attribute__((aligned(16))) double sm[];
  const int b = blockIdx.x;
  if (b >= batch) return;
  const int lane = threadIdx.x;
  typedef typename std::conditional<CONVEX, ConvexModel, QuatModel>::type MD;
  constexpr bool PROF = false;
  double* traj_x = nullptr;
  long long* prof_out = nullptr;
  // check_prev (the closed loop's per-tick form): info[b] still holds the status of the robot's previous solve; a failed
  // one left no usable solution behind, so this one starts cold -- the rule of the persistent kernel
  const bool usable = u_init && (!check_prev || info[b].status == QMPC_OK || info[b].status == QMPC_MAX_ITER);
  const int warm_t = usable ? 1 : 0;
  if (usable) {
    LayoutW LWw;
    const Layout Lw = (VAR == 3 || VAR == 5 || VAR == 6) ? make_layout_w(P.N, &LWw, VAR == 5 || VAR == 6, VAR == 6)
                                                         : make_layout(P.N, VAR == 1 || VAR == 2, MD::NL, VAR == 2);
    for (int i = lane; i < P.N * 12; i += kWave) sm[Lw.U + i] = u_init[(size_t)b * P.N * 12 + i];
    __syncthreads();
  }
  if constexpr ((VAR == 3 || VAR == 5 || VAR == 6) && CONVEX) {
    constexpr int WVAR = VAR;
    const int wslot = b;
    constexpr const double* resume = nullptr;
#define QMPC_WMODEL WM_CONVEX
#include "qmpc_wform_body.inc"
#undef QMPC_WMODEL
  } else if constexpr (VAR == 3 || VAR == 5 || VAR == 6) {
    constexpr int WVAR = VAR;
    const int wslot = b;
    constexpr const double* resume = nullptr;
#include "qmpc_wform_body.inc"
  } else {
#include "qmpc_solve_body.inc"
  }
}
#endif

}  // namespace qmpc
