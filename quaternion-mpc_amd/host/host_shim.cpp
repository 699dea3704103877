// host_shim.cpp -- C entry points over the C++ host classes so the Python test
// suite (ctypes) can drive QuatMpcHipT<LeggedStateLite>, LeggedContactFSMHip and
// MovingWindowFilterHip.  Builds to host/libqmpc_host.so with plain g++; the HIP
// library is dlopen'ed at run time, so this file has no HIP dependency.
#include <dlfcn.h>

#include <cstdint>
#include <cstdio>
#include <cstring>

#include "LeggedStateLite.h"
#include "QuatMpcHip.h"
#include "ConvexMpcHip.h"
#include "ClosedLoopHost.h"
#include "LeggedLoggerHip.h"

using legged::LeggedStateLite;
using Mpc = legged::QuatMpcHipT<LeggedStateLite>;
using CvxMpc = legged::ConvexMpcHipT<LeggedStateLite>;

namespace {
struct Harness {
  LeggedStateLite state;
  Mpc* mpc = nullptr;
  CvxMpc* cvx = nullptr;
  void* dl = nullptr;
};

// a stub that records that no device library was given (host-logic tests on CPU)
void stub_default_params(qmpc_params* p, int32_t horizon, int32_t mode) {
  std::memset(p, 0, sizeof *p);
  p->horizon = horizon;
  p->mode = mode;
}

// Test double of the C ABI (CPU suite only): a "device" whose answer the test scripts, so that the host class's
// handling of call-level and per-instance status words can be exercised without a GPU.
struct FakeDevice { double forces[12]; int32_t call_status; int32_t inst_status; int calls; } g_fake = {{0}, 0, 0, 0};
qmpc_status fake_create(const qmpc_params*, int32_t, int32_t, qmpc_handle** out) {
  *out = reinterpret_cast<qmpc_handle*>(&g_fake);
  return QMPC_OK;
}
qmpc_status fake_solve(qmpc_handle*, int32_t batch, const qmpc_input*, double* forces, qmpc_info* info) {
  g_fake.calls += 1;
  for (int b = 0; b < batch; ++b) {
    for (int i = 0; i < 12; ++i) forces[12 * b + i] = g_fake.inst_status == QMPC_NAN_INPUT ? 0.0 : g_fake.forces[i];
    if (info) { std::memset(&info[b], 0, sizeof(qmpc_info)); info[b].status = g_fake.inst_status; }
  }
  return static_cast<qmpc_status>(g_fake.call_status);
}
void fake_destroy(qmpc_handle*) {}

// binds the C-ABI entry points from libqmpc_hip.so (or leaves the stubs for a device-less harness)
bool bind_api(Harness* h, const char* lib_path, legged::QmpcApi& api);
}  // namespace

extern "C" {

// ConvexMpc harness: gazebo_go1_convex_mpc.yaml values (5 ms, Q of the Euler-angle state, mu 0.6, fz_max 200)
void* qh_convex_create(const char* lib_path, int horizon) {
  Harness* h = new Harness();
  h->state.param.mpc_horizon = horizon;
  h->state.param.mpc_update_period = 5.0;
  const double q[13] = {3.0, 3.0, 3.0, 1.0, 1.0, 20.0, 0.0, 0.0, 3.0, 2.0, 3.0, 2.0, 0.0};
  for (int i = 0; i < 13; ++i) h->state.param.q_weights[i] = q[i];
  h->state.param.mu = 0.6;
  h->state.param.fz_max = 200.0;
  legged::QmpcApi api;
  if (!bind_api(h, lib_path, api)) { delete h; return nullptr; }
  h->state.fbk.torso_rot_mat(0, 0) = h->state.fbk.torso_rot_mat(1, 1) = h->state.fbk.torso_rot_mat(2, 2) = 1.0;
  h->state.fbk.torso_rot_mat_z = h->state.fbk.torso_rot_mat;
  h->cvx = new CvxMpc(h->state, api, 0);
  return h;
}
int qh_convex_device_status(void* p) { return (int)static_cast<Harness*>(p)->cvx->last_status(); }
// extra feedback ConvexMpc reads: euler(3) ang_vel_world(3) foot_pos_abs_com(12, [3*leg+axis]) = 18 doubles
void qh_convex_set_feedback(void* p, const double* f) {
  LeggedStateLite& s = static_cast<Harness*>(p)->state;
  for (int i = 0; i < 3; ++i) { s.fbk.torso_euler[i] = f[i]; s.fbk.torso_ang_vel_world[i] = f[3 + i]; }
  for (int l = 0; l < 4; ++l)
    for (int a = 0; a < 3; ++a) s.fbk.foot_pos_abs_com(a, l) = f[6 + 3 * l + a];
}
void qh_convex_set_body_xy(void* p, double x, double y) {
  LeggedStateLite& s = static_cast<Harness*>(p)->state;
  s.joy.body_x = x; s.joy.body_y = y;
}
void qh_convex_goal_update(void* p) { Harness* h = static_cast<Harness*>(p); h->cvx->goal_update(h->state); }
void qh_convex_foot_update(void* p) { Harness* h = static_cast<Harness*>(p); h->cvx->foot_update(h->state); }
int qh_convex_grf_update(void* p) { Harness* h = static_cast<Harness*>(p); return h->cvx->grf_update(h->state) ? 1 : 0; }
int qh_convex_update(void* p) { Harness* h = static_cast<Harness*>(p); return h->cvx->update(h->state) ? 1 : 0; }
void qh_convex_pack_input(void* p, qmpc_convex_input* out) { Harness* h = static_cast<Harness*>(p); h->cvx->pack_input(h->state, out); }
// out: lin_vel_d_rel(3) lin_vel_d_world(3) pos_d_world(3) yaw_rate_d(1) optimized_state[0:6] = 16 doubles
void qh_convex_get_goal(void* p, double* o) {
  LeggedStateLite& s = static_cast<Harness*>(p)->state;
  for (int i = 0; i < 3; ++i) {
    o[i] = s.ctrl.torso_lin_vel_d_rel[i]; o[3 + i] = s.ctrl.torso_lin_vel_d_world[i]; o[6 + i] = s.ctrl.torso_pos_d_world[i];
  }
  o[9] = s.ctrl.torso_ang_vel_d_body[2];
  for (int i = 0; i < 6; ++i) o[10 + i] = s.ctrl.optimized_state[i];
}

// lib_path: path of libqmpc_hip.so, or NULL/"" for a harness without a device
void* qh_create(const char* lib_path, int horizon) {
  Harness* h = new Harness();
  h->state.param.mpc_horizon = horizon;
  legged::QmpcApi api;
  if (!bind_api(h, lib_path, api)) { delete h; return nullptr; }
  h->state.fbk.torso_rot_mat(0, 0) = h->state.fbk.torso_rot_mat(1, 1) = h->state.fbk.torso_rot_mat(2, 2) = 1.0;
  h->state.fbk.torso_rot_mat_z = h->state.fbk.torso_rot_mat;
  h->mpc = new Mpc(h->state, api, 0);
  return h;
}

}  // extern "C"

namespace {
bool bind_api(Harness* h, const char* lib_path, legged::QmpcApi& api) {
  api.default_params = stub_default_params;
  api.default_convex_params = stub_default_params;
  if (lib_path && lib_path[0]) {
    h->dl = dlopen(lib_path, RTLD_NOW | RTLD_LOCAL);
    if (!h->dl) {
      std::fprintf(stderr, "qh_create: dlopen(%s) failed: %s\n", lib_path, dlerror());
      return false;
    }
    api.default_params = reinterpret_cast<decltype(api.default_params)>(dlsym(h->dl, "qmpc_default_params"));
    api.create = reinterpret_cast<decltype(api.create)>(dlsym(h->dl, "qmpc_create"));
    api.solve = reinterpret_cast<decltype(api.solve)>(dlsym(h->dl, "qmpc_solve"));
    api.destroy = reinterpret_cast<decltype(api.destroy)>(dlsym(h->dl, "qmpc_destroy"));
    api.solve_warm = reinterpret_cast<decltype(api.solve_warm)>(dlsym(h->dl, "qmpc_solve_warm"));
    api.default_convex_params =
        reinterpret_cast<decltype(api.default_convex_params)>(dlsym(h->dl, "qmpc_default_convex_params"));
    api.convex_solve = reinterpret_cast<decltype(api.convex_solve)>(dlsym(h->dl, "qmpc_convex_solve"));
    if (!api.default_params || !api.create || !api.solve || !api.destroy || !api.default_convex_params ||
        !api.convex_solve) {
      std::fprintf(stderr, "qh_create: missing qmpc_* symbols in %s\n", lib_path);
      return false;
    }
  }
  return true;
}
}  // namespace

extern "C" {

// harness over the scripted test double above
void* qh_create_fake(int horizon) {
  Harness* h = new Harness();
  h->state.param.mpc_horizon = horizon;
  legged::QmpcApi api;
  api.default_params = stub_default_params;
  api.create = fake_create;
  api.solve = fake_solve;
  api.destroy = fake_destroy;
  h->state.fbk.torso_rot_mat(0, 0) = h->state.fbk.torso_rot_mat(1, 1) = h->state.fbk.torso_rot_mat(2, 2) = 1.0;
  h->state.fbk.torso_rot_mat_z = h->state.fbk.torso_rot_mat;
  h->mpc = new Mpc(h->state, api, 0);
  return h;
}
void qh_fake_script(const double* forces12, int call_status, int inst_status) {
  for (int i = 0; i < 12; ++i) g_fake.forces[i] = forces12[i];
  g_fake.call_status = call_status;
  g_fake.inst_status = inst_status;
}

// ---- closed loop on the host: the parity reference of qmpc_loop_run (one instance) ----------------
struct LoopHarness {
  Harness base;                      // owns the dlopen handle
  legged::ClosedLoopHostBase<LeggedStateLite>* loop = nullptr;
};
// lib_path NULL / "": the scripted test double (qh_fake_script) stands in for the device
void* qh_loop_create_opts(const char* lib_path, int horizon, int mode, int drop_ang_vel, const qmpc_loop_params* lp,
                          const qmpc_loop_state* init) {
  LoopHarness* h = new LoopHarness();
  legged::QmpcApi api;
  if (lib_path && lib_path[0]) {
    if (!bind_api(&h->base, lib_path, api)) { delete h; return nullptr; }
  } else {
    api.default_params = stub_default_params;
    api.create = fake_create;
    api.solve = fake_solve;
    api.destroy = fake_destroy;
  }
  h->loop = new legged::ClosedLoopHostT<LeggedStateLite>(api, *lp, *init, horizon, 0, mode, drop_ang_vel);
  return h;
}
// the sibling controller in the same loop: ConvexMpcHipT (gazebo_go1_convex_mpc.yaml values)
void* qh_loop_create_convex_mode(const char* lib_path, int horizon, int mode, const qmpc_loop_params* lp, const qmpc_loop_state* init) {
  LoopHarness* h = new LoopHarness();
  legged::QmpcApi api;
  if (!(lib_path && lib_path[0]) || !bind_api(&h->base, lib_path, api)) { delete h; return nullptr; }
  h->loop = new legged::ClosedLoopHostT<LeggedStateLite, legged::ConvexMpcHipT<LeggedStateLite>>(api, *lp, *init, horizon, 0, mode);
  return h;
}
void* qh_loop_create_convex(const char* lib_path, int horizon, const qmpc_loop_params* lp, const qmpc_loop_state* init) {
  return qh_loop_create_convex_mode(lib_path, horizon, QMPC_MODE_CONVERGED, lp, init);
}
void* qh_loop_create_mode(const char* lib_path, int horizon, int mode, const qmpc_loop_params* lp, const qmpc_loop_state* init) {
  return qh_loop_create_opts(lib_path, horizon, mode, 1, lp, init);
}
void* qh_loop_create(const char* lib_path, int horizon, const qmpc_loop_params* lp, const qmpc_loop_state* init) {
  return qh_loop_create_mode(lib_path, horizon, QMPC_MODE_CONVERGED, lp, init);
}
int qh_loop_device_status(void* p) { return (int)static_cast<LoopHarness*>(p)->loop->device_status(); }
int qh_loop_tick(void* p) { return static_cast<LoopHarness*>(p)->loop->tick() ? 1 : 0; }
// joy.{velx, vely, body_height, roll_rate, pitch_rate, yaw_rate} and ctrl.movement_mode from the next tick on
void qh_loop_set_command(void* p, const double* joy, double movement_mode) {
  LeggedStateLite& s = static_cast<LoopHarness*>(p)->loop->state;
  s.joy.velx = joy[0]; s.joy.vely = joy[1]; s.joy.body_height = joy[2];
  s.joy.roll_rate = joy[3]; s.joy.pitch_rate = joy[4]; s.joy.yaw_rate = joy[5];
  s.ctrl.movement_mode = movement_mode;
}
// joint level of the tick just made: joint_pos_io [12] in/out, one feedback and one command record out
void qh_loop_joint(void* p, double* joint_pos_io, qmpc_joint_feedback* fb, qmpc_joint_command* cmd) {
  static_cast<LoopHarness*>(p)->loop->joint_commands(joint_pos_io, fb, cmd);
}
// BaseInterface::tau_ctrl_update on plain records (the host mirror of qmpc_joint_commands), batch instances
void qh_joint_commands(int batch, const qmpc_joint_feedback* fb, qmpc_joint_command* cmd) {
  legged::JointCommandsHipT<LeggedStateLite> jc;
  for (int b = 0; b < batch; ++b) {
    LeggedStateLite s;
    double R[9];
    qmpc_loop::quat_to_rot(fb[b].torso_quat, R);
    for (int r = 0; r < 3; ++r) {
      for (int c = 0; c < 3; ++c) s.fbk.torso_rot_mat(r, c) = R[3 * r + c];
      s.fbk.torso_pos_world(r) = fb[b].torso_pos_world[r];
      s.fbk.torso_lin_vel_world(r) = fb[b].torso_lin_vel_world[r];
    }
    for (int a = 0; a < 12; ++a) {
      s.fbk.joint_pos(a) = fb[b].joint_pos[a];
      s.fbk.joint_vel(a) = fb[b].joint_vel[a];
      s.ctrl.optimized_state(6 + a) = fb[b].foot_pos_target_world[a];
      s.ctrl.optimized_input(12 + a) = fb[b].foot_vel_target_world[a];
      s.ctrl.optimized_input(a) = fb[b].forces_body[a];
    }
    for (int l = 0; l < 4; ++l) s.ctrl.plan_contacts[l] = fb[b].plan_contacts[l] != 0.0;
    s.ctrl.movement_mode = fb[b].movement_mode;
    jc.tau_ctrl_update(s);
    for (int a = 0; a < 12; ++a) {
      cmd[b].joint_ang_tgt[a] = s.ctrl.joint_ang_tgt(a);
      cmd[b].joint_vel_tgt[a] = s.ctrl.joint_vel_tgt(a);
      cmd[b].joint_tau_tgt[a] = s.ctrl.joint_tau_tgt(a);
    }
  }
}
// A1Kinematics::inv_kin through the shared arithmetic, host side: [batch][12] each
void qh_leg_inverse(int batch, const double* foot_pos_body, const double* cur_joint_pos, double* joint_pos) {
  legged::JointCommandsHipT<LeggedStateLite> jc;
  for (size_t t = 0; t < (size_t)batch * 4; ++t)
    qmpc_joint::leg_inverse(&foot_pos_body[3 * t], cur_joint_pos[3 * t], jc.geom.rho_fix[t & 3], &joint_pos[3 * t]);
}
void qh_loop_set_warm_start(void* p, int on) { static_cast<LoopHarness*>(p)->loop->set_warm_start(on != 0); }
void qh_loop_set_sin_ang_vel(void* p, int on) { static_cast<LoopHarness*>(p)->loop->state.joy.sin_ang_vel = on != 0; }
void qh_loop_export(void* p, qmpc_loop_state* out) { static_cast<LoopHarness*>(p)->loop->export_state(out); }
void qh_loop_destroy(void* p) {
  LoopHarness* h = static_cast<LoopHarness*>(p);
  if (!h) return;
  delete h->loop;
  delete h;
}

void qh_destroy(void* p) {
  Harness* h = static_cast<Harness*>(p);
  if (!h) return;
  delete h->mpc;
  delete h->cvx;
  // the HIP library stays loaded: unloading a library with live device code objects is not safe
  delete h;
}

int qh_device_status(void* p) { return (int)static_cast<Harness*>(p)->mpc->last_status(); }

// fbk: quat(4) rot(9, row-major) pos_world(3) lin_vel_world(3) ang_vel_body(3)
//      foot_pos_body(12, [3*leg+axis]) foot_contact_flag(4)            = 38 doubles
void qh_set_feedback(void* p, const double* f) {
  LeggedStateLite& s = static_cast<Harness*>(p)->state;
  s.fbk.torso_quat.w() = f[0]; s.fbk.torso_quat.x() = f[1]; s.fbk.torso_quat.y() = f[2]; s.fbk.torso_quat.z() = f[3];
  for (int r = 0; r < 3; ++r)
    for (int c = 0; c < 3; ++c) s.fbk.torso_rot_mat(r, c) = f[4 + 3 * r + c];
  // yaw-only rotation (BaseInterface.cpp keeps torso_rot_mat_z next to torso_rot_mat)
  const double yaw = std::atan2(s.fbk.torso_rot_mat(1, 0), s.fbk.torso_rot_mat(0, 0));
  s.fbk.torso_rot_mat_z.setZero();
  s.fbk.torso_rot_mat_z(0, 0) = std::cos(yaw); s.fbk.torso_rot_mat_z(0, 1) = -std::sin(yaw);
  s.fbk.torso_rot_mat_z(1, 0) = std::sin(yaw); s.fbk.torso_rot_mat_z(1, 1) = std::cos(yaw);
  s.fbk.torso_rot_mat_z(2, 2) = 1.0;
  for (int i = 0; i < 3; ++i) {
    s.fbk.torso_pos_world[i] = f[13 + i];
    s.fbk.torso_lin_vel_world[i] = f[16 + i];
    s.fbk.torso_ang_vel_body[i] = f[19 + i];
  }
  for (int l = 0; l < 4; ++l) {
    for (int a = 0; a < 3; ++a) s.fbk.foot_pos_body(a, l) = f[22 + 3 * l + a];
    s.fbk.foot_contact_flag[l] = f[34 + l];
  }
  s.estimator_init = true;
}

// joy: velx vely body_height roll_rate pitch_rate yaw_rate ; movement_mode
void qh_set_command(void* p, const double* j, double movement_mode) {
  LeggedStateLite& s = static_cast<Harness*>(p)->state;
  s.joy.velx = j[0]; s.joy.vely = j[1]; s.joy.body_height = j[2];
  s.joy.roll_rate = j[3]; s.joy.pitch_rate = j[4]; s.joy.yaw_rate = j[5];
  s.ctrl.movement_mode = movement_mode;
}

void qh_goal_update(void* p) { Harness* h = static_cast<Harness*>(p); h->mpc->goal_update(h->state); }
void qh_foot_update(void* p) { Harness* h = static_cast<Harness*>(p); h->mpc->foot_update(h->state); }
int qh_grf_update(void* p) { Harness* h = static_cast<Harness*>(p); return h->mpc->grf_update(h->state) ? 1 : 0; }
int qh_update(void* p) { Harness* h = static_cast<Harness*>(p); return h->mpc->update(h->state) ? 1 : 0; }

// the record grf_update would hand to the C ABI (includes the quat_d in/out update)
void qh_pack_input(void* p, qmpc_input* out) { Harness* h = static_cast<Harness*>(p); h->mpc->pack_input(h->state, out); }

// out: plan_contacts(4) gait_counter(4) optimized_input[0:12] mpc_grf_world(12) quat_d(4)
//      pos_d_world(3) mpc_time(1)                                        = 40 doubles
void qh_get_outputs(void* p, double* o) {
  LeggedStateLite& s = static_cast<Harness*>(p)->state;
  for (int i = 0; i < 4; ++i) { o[i] = s.ctrl.plan_contacts[i] ? 1.0 : 0.0; o[4 + i] = s.ctrl.gait_counter[i]; }
  for (int i = 0; i < 12; ++i) { o[8 + i] = s.ctrl.optimized_input[i]; o[20 + i] = s.ctrl.mpc_grf_world[i]; }
  o[32] = s.ctrl.torso_quat_d.w(); o[33] = s.ctrl.torso_quat_d.x(); o[34] = s.ctrl.torso_quat_d.y(); o[35] = s.ctrl.torso_quat_d.z();
  for (int i = 0; i < 3; ++i) o[36 + i] = s.ctrl.torso_pos_d_world[i];
  o[39] = s.fbk.mpc_time;
}

// what the low-level thread reads besides the forces (BaseInterface.cpp:349,358; QuatMpc.cpp:270-272):
// out = optimized_state[6:18] optimized_input[12:24] optimized_input[24:36], then the FSM members themselves
// (pos, vel, acc per leg, [3*leg+axis])                                 = 72 doubles
void qh_get_foot_targets(void* p, double* o) {
  Harness* h = static_cast<Harness*>(p);
  LeggedStateLite& s = h->state;
  for (int i = 0; i < 12; ++i) {
    o[i] = s.ctrl.optimized_state[6 + i];
    o[12 + i] = s.ctrl.optimized_input[12 + i];
    o[24 + i] = s.ctrl.optimized_input[24 + i];
  }
  for (int l = 0; l < 4; ++l)
    for (int a = 0; a < 3; ++a) {
      o[36 + 3 * l + a] = h->mpc->leg_FSM[l].FSM_foot_pos_target_world[a];
      o[48 + 3 * l + a] = h->mpc->leg_FSM[l].FSM_foot_vel_target_world[a];
      o[60 + 3 * l + a] = h->mpc->leg_FSM[l].FSM_foot_acc_target_world[a];
    }
}
// fbk.foot_pos_world and ctrl.foot_pos_target_world ([3*leg+axis]), the FSM's inputs (QuatMpc.cpp:291-294)
void qh_set_foot_world(void* p, const double* cur12, const double* tgt12) {
  LeggedStateLite& s = static_cast<Harness*>(p)->state;
  for (int l = 0; l < 4; ++l)
    for (int a = 0; a < 3; ++a) {
      s.fbk.foot_pos_world(a, l) = cur12[3 * l + a];
      s.ctrl.foot_pos_target_world(a, l) = tgt12[3 * l + a];
    }
}

// ---- stand-alone pieces ---------------------------------------------------------
// contact schedule: ticks x 4 flags in (any non-zero = contact), ticks x 4 contacts
// and phases out.  mode[t] is ctrl.movement_mode at tick t.
void qh_fsm_run(double gait_freq, int ticks, const double* mode, const double* flags, int32_t* contacts, double* phases) {
  legged::LeggedContactFSMHip fsm[4];
  for (int i = 0; i < 4; ++i) fsm[i].reset_params(gait_freq, i);
  for (int t = 0; t < ticks; ++t) {
    if (mode[t] == 0) {
      for (int i = 0; i < 4; ++i) { fsm[i].reset(); contacts[4 * t + i] = 1; phases[4 * t + i] = fsm[i].phase(); }
    } else {
      for (int i = 0; i < 4; ++i)
        phases[4 * t + i] = fsm[i].update(5.0 / 1000.0, gait_freq, static_cast<bool>(flags[4 * t + i]));
      for (int i = 0; i < 4; ++i) contacts[4 * t + i] = (int32_t)fsm[i].get_contact_state();
    }
  }
}

// swing-foot quintic (Utils.cpp:236-293): out = pos(3) vel(3) acc(3)
void qh_swing_target(float t, float T, const double* start, const double* fin, double* out) {
  legged::QuinticCurveHip q;
  q.get_foot_swing_target(t, T, start, fin, out);
}

// full FSM update over `ticks` ticks for one leg: cur / tgt are [ticks][3], flags [ticks];
// outputs contacts [ticks], phases [ticks], targets [ticks][9] (pos, vel, acc)
void qh_fsm_leg_run(int leg, double gait_freq, double dt, int ticks, const double* cur, const double* tgt,
                    const double* flags, int32_t* contacts, double* phases, double* targets) {
  legged::LeggedContactFSMHip fsm;
  fsm.reset_params(gait_freq, leg);
  fsm.reset();
  for (int t = 0; t < ticks; ++t) {
    phases[t] = fsm.update(dt, gait_freq, cur + 3 * t, tgt + 3 * t, static_cast<bool>(flags[t]));
    contacts[t] = (int32_t)fsm.get_contact_state();
    for (int a = 0; a < 3; ++a) {
      targets[9 * t + a] = fsm.FSM_foot_pos_target_world[a];
      targets[9 * t + 3 + a] = fsm.FSM_foot_vel_target_world[a];
      targets[9 * t + 6 + a] = fsm.FSM_foot_acc_target_world[a];
    }
  }
}

// Raibert foothold targets (BaseInterface.cpp:266-288) on the harness state; vel_d_rel(3) is
// ctrl.torso_lin_vel_d_rel; out = foot_pos_target_abs(12) rel(12) world(12), [3*leg+axis]
void qh_raibert(void* p, const double* vel_d_rel, double* out) {
  LeggedStateLite& s = static_cast<Harness*>(p)->state;
  for (int i = 0; i < 3; ++i) s.ctrl.torso_lin_vel_d_rel[i] = vel_d_rel[i];
  legged::raibert_foot_targets(s);
  for (int l = 0; l < 4; ++l)
    for (int a = 0; a < 3; ++a) {
      out[3 * l + a] = s.ctrl.foot_pos_target_abs(a, l);
      out[12 + 3 * l + a] = s.ctrl.foot_pos_target_rel(a, l);
      out[24 + 3 * l + a] = s.ctrl.foot_pos_target_world(a, l);
    }
}

// /debug topic payloads (LeggedLogger.hpp:48-106) of the harness state, flattened:
// out[0:13] torso_odom (pos3, quat wxyz, lin3, ang3), out[13:26] torso_odom_d, out[26:30] mpc_grf.position,
// out[30:34] .velocity, out[34:38] .effort, out[38] mpc_time; names -> 4 x 3 chars (NUL terminated)
void qh_debug_records(void* p, double* out, char* names) {
  const LeggedStateLite& s = static_cast<Harness*>(p)->state;
  legged::DebugRecords r;
  legged::debug_records(s, r);
  const legged::OdomRecord* od[2] = {&r.torso_odom, &r.torso_odom_d};
  for (int k = 0; k < 2; ++k) {
    double* o = out + 13 * k;
    for (int a = 0; a < 3; ++a) { o[a] = od[k]->position[a]; o[7 + a] = od[k]->linear[a]; o[10 + a] = od[k]->angular[a]; }
    for (int a = 0; a < 4; ++a) o[3 + a] = od[k]->orientation[a];
  }
  for (int i = 0; i < 4; ++i) {
    out[26 + i] = r.mpc_grf.position[i];
    out[30 + i] = r.mpc_grf.velocity[i];
    out[34 + i] = r.mpc_grf.effort[i];
    std::snprintf(names + 3 * i, 3, "%s", r.mpc_grf.name[i]);
  }
  out[38] = r.mpc_time;
}

// test hook: the controller outputs the logger reads
void qh_set_mpc_outputs(void* p, const double* grf_world, const double* contacts, double mpc_time) {
  LeggedStateLite& s = static_cast<Harness*>(p)->state;
  for (int i = 0; i < 12; ++i) s.ctrl.mpc_grf_world[i] = grf_world[i];
  for (int i = 0; i < 4; ++i) s.ctrl.plan_contacts[i] = contacts[i] != 0.0;
  s.fbk.mpc_time = mpc_time;
}

void qh_debug_grf_batch(int32_t batch, const double* forces_world, const double* contacts, double* position,
                        double* effort) {
  legged::debug_grf_batch(batch, forces_world, contacts, position, effort);
}

void qh_filter_run(int window, int n, const double* in, double* out) {
  legged::MovingWindowFilterHip f(window);
  for (int i = 0; i < n; ++i) out[i] = f.CalculateAverage(in[i]);
}

}  // extern "C"
