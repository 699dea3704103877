/*
 * qo_legkin.c -- see qo_legkin.h.  TEST INFRASTRUCTURE ONLY.
 *
 * The reference evaluates machine-generated closed forms (A1Kinematics.cpp:38-128).
 * Restated geometrically: with c = rho_opt, (ox, oy, d, lt, lc) = rho_fix,
 *   L  = lt cos q1 + (lc - c2) cos(q1+q2) + c0 sin(q1+q2)   (leg extension along the hip's -z)
 *   X  = -lt sin q1 - (lc - c2) sin(q1+q2) + c0 cos(q1+q2)  (fore-aft offset)
 *   p  = ( ox + X,  oy + (d+c1) cos q0 + L sin q0,  (d+c1) sin q0 - L cos q0 )
 * and dL/dq1 = X, dX/dq1 = -L, dL/dq2 = X2, dX/dq2 = -L2 with the calf-only parts L2, X2.
 */
#include "qo_legkin.h"

#include <math.h>
#include <string.h>

void qo_default_go1_geometry(qmpc_leg_geometry* g) {
  memset(g, 0, sizeof *g);
  const double sx[4] = {1, 1, -1, -1}, sy[4] = {1, -1, 1, -1};
  for (int l = 0; l < 4; ++l) { /* BaseInterface.cpp:12-26; LeggedParams.h:14-15 */
    g->rho_fix[l][0] = sx[l] * 0.1881;
    g->rho_fix[l][1] = sy[l] * 0.04675;
    g->rho_fix[l][2] = sy[l] * 0.0812;
    g->rho_fix[l][3] = 0.213;
    g->rho_fix[l][4] = 0.213;
  }
}

typedef struct leg_terms { double s0, c0, L, X, L2, X2, D; } leg_terms;

static void terms(const double q[3], const double c[3], const double r[5], leg_terms* t) {
  t->s0 = sin(q[0]); t->c0 = cos(q[0]);
  const double s1 = sin(q[1]), c1 = cos(q[1]);
  const double s12 = sin(q[1] + q[2]), c12 = cos(q[1] + q[2]);
  const double lce = r[4] - c[2];
  t->L2 = lce * c12 + c[0] * s12;
  t->X2 = -lce * s12 + c[0] * c12;
  t->L = r[3] * c1 + t->L2;
  t->X = -r[3] * s1 + t->X2;
  t->D = r[2] + c[1];
}

void qo_leg_fk(const double q[3], const double c[3], const double r[5], double p[3]) {
  leg_terms t;
  terms(q, c, r, &t);
  p[0] = r[0] + t.X;
  p[1] = r[1] + t.D * t.c0 + t.L * t.s0;
  p[2] = t.D * t.s0 - t.L * t.c0;
}

void qo_leg_jac(const double q[3], const double c[3], const double r[5], double J[9]) {
  leg_terms t;
  terms(q, c, r, &t);
  J[0] = 0.0;       J[1] = -t.D * t.s0 + t.L * t.c0; J[2] = t.D * t.c0 + t.L * t.s0;
  J[3] = -t.L;      J[4] = t.X * t.s0;               J[5] = -t.X * t.c0;
  J[6] = -t.L2;     J[7] = t.X2 * t.s0;              J[8] = -t.X2 * t.c0;
}

void qo_leg_kinematics(const qmpc_leg_geometry* g, int32_t batch, const double* joint_pos,
                       double* foot_pos_body, double* jac) {
  for (int b = 0; b < batch; ++b)
    for (int l = 0; l < 4; ++l) {
      const double* q = &joint_pos[12 * (size_t)b + 3 * l];
      if (foot_pos_body) qo_leg_fk(q, g->rho_opt[l], g->rho_fix[l], &foot_pos_body[12 * (size_t)b + 3 * l]);
      if (jac) qo_leg_jac(q, g->rho_opt[l], g->rho_fix[l], &jac[36 * (size_t)b + 9 * l]);
    }
}

/* BaseInterface.cpp:366-370 (walking) and :401 (standing) */
void qo_torque_map(const qmpc_leg_geometry* g, int32_t batch, const double* joint_pos,
                   const double* forces_body, const double* contacts, int32_t walking, double* tau) {
  for (int b = 0; b < batch; ++b)
    for (int l = 0; l < 4; ++l) {
      double* t = &tau[12 * (size_t)b + 3 * l];
      const int stance = !contacts || contacts[4 * (size_t)b + l] != 0.0;
      if (walking && !stance) { t[0] = t[1] = t[2] = 0.0; continue; }
      double J[9];
      qo_leg_jac(&joint_pos[12 * (size_t)b + 3 * l], g->rho_opt[l], g->rho_fix[l], J);
      const double* f = &forces_body[12 * (size_t)b + 3 * l];
      for (int j = 0; j < 3; ++j) t[j] = -(J[3 * j] * f[0] + J[3 * j + 1] * f[1] + J[3 * j + 2] * f[2]);
    }
}
