/* solve_batch.c -- the C ABI from plain C: one stand-pose Go1 state and a small batch of perturbed copies through
 * qmpc_solve (host buffers, blocking), the call that replaces legged::QuatMpc::grf_update's solver block
 * (legged_ctrl/src/mpc/QuatMpc.cpp:217-265).
 *   gcc -O2 -I include examples/solve_batch.c -o examples/solve_batch quaternion-mpc_amd/csrc/libqmpc_hip.so \
 *       -Wl,-rpath,'$ORIGIN/../quaternion-mpc_amd/csrc' -lm
 * Prints the 12 body-frame foot forces of instance 0 (stand: fz = m g / 4 per leg up to the moment balance) and the
 * status / iteration counts; exits non-zero on any failure (there is no CPU fallback: without a GPU qmpc_create
 * returns QMPC_NO_DEVICE). */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "qmpc.h"

int main(int argc, char** argv) {
  const int batch = (argc > 1) ? atoi(argv[1]) : 8;
  qmpc_params p;
  qmpc_default_params(&p, /*horizon=*/10, QMPC_MODE_CONVERGED);
  qmpc_handle* h = NULL;
  qmpc_status st = qmpc_create(&p, batch, /*device=*/0, &h);
  if (st != QMPC_OK) {
    fprintf(stderr, "qmpc_create: %s\n", qmpc_status_string(st));
    return 2;
  }
  qmpc_input* in = calloc((size_t)batch, sizeof *in);
  double* forces = calloc((size_t)batch * 12, sizeof *forces);
  qmpc_info* info = calloc((size_t)batch, sizeof *info);
  const double feet[4][3] = {{0.20, 0.14, -0.30}, {0.20, -0.14, -0.30}, {-0.20, 0.14, -0.30}, {-0.20, -0.14, -0.30}};
  for (int b = 0; b < batch; ++b) {
    qmpc_input* r = &in[b];
    const double yaw = 0.05 * b;                       /* instance 0 is the exact stand pose */
    r->quat[0] = cos(yaw / 2); r->quat[3] = sin(yaw / 2);
    r->rot[0] = cos(yaw); r->rot[1] = -sin(yaw); r->rot[3] = sin(yaw); r->rot[4] = cos(yaw); r->rot[8] = 1.0;
    for (int l = 0; l < 4; ++l) {
      for (int a = 0; a < 3; ++a) r->foot_pos_body[3 * l + a] = feet[l][a];
      r->contacts[l] = 1.0;
    }
    r->lin_vel_body[0] = 0.02 * b;
    memcpy(r->quat_d, r->quat, sizeof r->quat);
  }
  st = qmpc_solve(h, batch, in, forces, info);
  if (st != QMPC_OK) {
    fprintf(stderr, "qmpc_solve: %s\n", qmpc_status_string(st));
    return 3;
  }
  double fz = 0.0;
  printf("instance 0 forces (body frame, N):");
  for (int j = 0; j < 12; ++j) {
    printf(" %.6f", forces[j]);
    if (j % 3 == 2) fz += forces[j];
  }
  printf("\nsum fz = %.9f (m g = %.9f)\n", fz, p.mass * 9.81);
  int bad = 0;
  for (int b = 0; b < batch; ++b) {
    printf("instance %d: status %d (%s), %d iterations, |dU| %.2e\n", b, info[b].status, qmpc_status_string(info[b].status),
           info[b].iterations, info[b].last_step);
    bad += info[b].status != QMPC_OK;
  }
  qmpc_destroy(h);
  free(in); free(forces); free(info);
  return bad ? 4 : (fabs(fz - p.mass * 9.81) < 1e-3 ? 0 : 5);
}
