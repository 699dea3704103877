/* closed_loop.c -- the device-resident closed loop from plain C (include/qmpc.h, qmpc_loop_*): a squad of Go1 robots
 * stands up, then trots under individual velocity / turn commands; every tick (goal update, gait FSM, swing quintic,
 * Raibert footholds, record packing -> MPC solve -> rigid-body plant) runs on the GPU with the robots' state in HBM.
 *   gcc -O2 -I include examples/closed_loop.c -o examples/closed_loop quaternion-mpc_amd/csrc/libqmpc_hip.so \
 *       -Wl,-rpath,'$ORIGIN/../quaternion-mpc_amd/csrc' -lm
 * usage: closed_loop [robots=64] [ticks=400] [warm=0]     (one tick = 5 ms; warm=1: every solve starts from the previous
 *        tick's solution and a low initial barrier -- the same forces in about half the iterations)
 * Prints where the robots ended up; exits non-zero if a robot fell, a solve failed, or there is no GPU (there is no
 * CPU fallback: qmpc_create then returns QMPC_NO_DEVICE). */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>

#include "qmpc.h"

int main(int argc, char** argv) {
  const int robots = (argc > 1) ? atoi(argv[1]) : 64;
  const int ticks = (argc > 2) ? atoi(argv[2]) : 400;
  const int warm = (argc > 3) ? atoi(argv[3]) : 0;
  qmpc_params p;
  qmpc_default_params(&p, /*horizon=*/10, QMPC_MODE_CONVERGED);
  p.drop_ang_vel = 0;   /* the MPC sees the body's angular velocity: the reference's x_init leaves it out (QuatMpc.cpp:242-245),
                           which an ideal rigid-body plant without leg damping does not forgive for long */
  if (warm) p.ipm_mu0 = 1e-6;   /* goes with lp.warm_start below */
  qmpc_handle* h = NULL;
  qmpc_status st = qmpc_create(&p, robots, /*device=*/0, &h);
  if (st != QMPC_OK) {
    fprintf(stderr, "qmpc_create: %s\n", qmpc_status_string(st));
    return 2;
  }
  qmpc_loop_params lp;
  qmpc_default_loop_params(&lp);
  lp.warm_start = warm ? 1.0 : 0.0;
  qmpc_loop_state* s = calloc((size_t)robots, sizeof *s);
  for (int i = 0; i < robots; ++i) {
    /* joy: velx, vely, body_height, roll_rate, pitch_rate, yaw_rate -- a fan of headings and speeds */
    const double joy[6] = {0.1 + 0.4 * i / (double)robots, 0.0, 0.30, 0.0, 0.0, (i % 3 - 1) * 0.2};
    qmpc_loop_state_init(&s[i], &lp, joy, /*movement_mode=*/0.0, /*height=*/0.30, /*yaw=*/6.28 * i / (double)robots);
  }
  st = qmpc_loop_run(h, &lp, robots, s, /*ticks=*/10, NULL, NULL);        /* stand: the gait FSM resets */
  for (int i = 0; i < robots && st == QMPC_OK; ++i) s[i].movement_mode = 1.0;   /* the state is plain data between runs */
  if (st == QMPC_OK) st = qmpc_loop_run(h, &lp, robots, s, ticks, NULL, NULL);
  if (st != QMPC_OK) {
    fprintf(stderr, "qmpc_loop_run: %s\n", qmpc_status_string(st));
    return 3;
  }
  int bad = 0;
  double far = 0.0;
  for (int i = 0; i < robots; ++i) {
    const double d = hypot(s[i].pos_world[0], s[i].pos_world[1]);
    if (d > far) far = d;
    if (!(s[i].pos_world[2] > 0.2 && s[i].pos_world[2] < 0.4) || s[i].status != 0.0) ++bad;
    if (i < 4 || i == robots - 1)
      printf("robot %3d: tick %.0f, position (%.3f, %.3f, %.3f), contacts %d%d%d%d, solver status %.0f (%.0f iterations)\n", i,
             s[i].tick, s[i].pos_world[0], s[i].pos_world[1], s[i].pos_world[2], (int)s[i].contacts[0], (int)s[i].contacts[1],
             (int)s[i].contacts[2], (int)s[i].contacts[3], s[i].status, s[i].iterations);
  }
  /* the tick down to the motors (BaseInterface::tau_ctrl_update): measured joint angles by inverse kinematics of the
   * plant's feet, joint angle / velocity targets of the swing feet, joint torques -J'f of the stance feet */
  qmpc_leg_geometry geom;
  qmpc_default_go1_geometry(&geom);
  double* joint_pos = malloc(sizeof(double) * 12 * (size_t)robots);
  qmpc_joint_command* cmd = calloc((size_t)robots, sizeof *cmd);
  qmpc_loop_joint_init(joint_pos, robots);
  st = qmpc_loop_joint_commands(h, &geom, robots, s, joint_pos, NULL, cmd);
  if (st != QMPC_OK) {
    fprintf(stderr, "qmpc_loop_joint_commands: %s\n", qmpc_status_string(st));
    return 5;
  }
  for (int l = 0; l < 4; ++l)
    printf("robot   0 leg %d (%s): joint angles (%.3f, %.3f, %.3f) rad, torque command (%.2f, %.2f, %.2f) N m\n", l,
           s[0].contacts[l] != 0.0 ? "stance" : "swing ", joint_pos[3 * l], joint_pos[3 * l + 1], joint_pos[3 * l + 2],
           cmd[0].joint_tau_tgt[3 * l], cmd[0].joint_tau_tgt[3 * l + 1], cmd[0].joint_tau_tgt[3 * l + 2]);
  free(joint_pos);
  free(cmd);
  printf("%d robots x %d ticks (%.1f s of robot time): farthest walked %.3f m, %d not upright / not converged\n", robots, ticks,
         ticks * lp.dt, far, bad);
  qmpc_destroy(h);
  free(s);
  return bad ? 4 : 0;
}
