/*
 * qo_kat.c -- the reference's generic AL-iLQR known-answer tests, restated as
 * problems for the oracle's solver scheme (qo_altro.c).  TEST INFRASTRUCTURE
 * ONLY.  These pin the solver *scheme* (iteration counts, saturation values);
 * the expected numbers live in the reference's own tests:
 *   legged_ctrl/src/test/test_altro/TestDoubleIntegrator.cpp:170-256 (goal
 *     constraint, GetIterations()==3), :258-375 (control bounds, u0 = -1,
 *     GetIterations()==5)
 *   legged_ctrl/src/test/test_altro/TestPendulum.cpp:13-43 (midpoint KAT),
 *     :45-115 (swing-up, xN_expected, <= 10 iterations)
 *   legged_ctrl/src/test/test_altro/TestDoubleIntegrator.cpp:69-168 (unconstrained,
 *     3 iterations: closer to the goal than x0 but farther than 1e-3)
 *   legged_ctrl/src/test/test_altro/TestPendulum.cpp:117-203 (goal constrained,
 *     |xN - xf| < 1e-4 in <= 10 iterations)
 *   legged_ctrl/src/test/test_altro/AltroTestUtils.cpp:45-82 (pendulum model)
 */
#include <math.h>
#include <string.h>

#include "qo_altro.h"
#include "qo_srbd.h"

/* ---- double integrator (TestDoubleIntegrator.cpp:11-34), dim = 2 ---------- */
static void di_dyn(void* ctx, int k, double* xn, const double* x, const double* u, float h) {
  (void)ctx; (void)k;
  const double b = h * h / 2; /* float arithmetic, as upstream */
  for (int i = 0; i < 2; ++i) {
    xn[i] = x[i] + x[i + 2] * h + u[i] * b;
    xn[i + 2] = x[i + 2] + u[i] * h;
  }
}
static void di_jac(void* ctx, int k, double* J, const double* x, const double* u, float h) {
  (void)ctx; (void)k; (void)x; (void)u;
  memset(J, 0, sizeof(double) * 4 * 6);
  const double b = h * h / 2;
  for (int i = 0; i < 2; ++i) {
    J[i + 4 * i] = 1.0;
    J[(i + 2) + 4 * (i + 2)] = 1.0;
    J[i + 4 * (i + 2)] = h;
    J[i + 4 * (4 + i)] = b;
    J[(i + 2) + 4 * (4 + i)] = h;
  }
}
static void goal_con(void* ctx, int k, double* c, const double* x, const double* u) {
  (void)ctx; (void)k; (void)u;
  for (int i = 0; i < 4; ++i) c[i] = x[i]; /* xf = 0 */
}
static void goal_jac(void* ctx, int k, double* J, const double* x, const double* u) {
  (void)ctx; (void)k; (void)x; (void)u;
  for (int i = 0; i < 4; ++i) J[i + 4 * i] = 1.0; /* 4 x 6 col-major */
}
static void ubnd_con(void* ctx, int k, double* c, const double* x, const double* u) {
  (void)ctx; (void)k; (void)x;
  for (int i = 0; i < 2; ++i) {
    c[i] = u[i] - 1.0;
    c[i + 2] = -1.0 - u[i];
  }
}
static void ubnd_jac(void* ctx, int k, double* J, const double* x, const double* u) {
  (void)ctx; (void)k; (void)x; (void)u;
  for (int i = 0; i < 2; ++i) {
    J[i + 4 * (4 + i)] = 1.0;
    J[(i + 2) + 4 * (4 + i)] = -1.0;
  }
}

static void di_problem(qo_problem* p, double x00, double x01) {
  memset(p, 0, sizeof *p);
  p->n = 4; p->m = 2; p->N = 10;
  const float tf = 5.0f;
  p->h = tf / (float)10.0; /* const float h = tf / static_cast<double>(num_horizon) */
  p->dyn = di_dyn; p->jac = di_jac;
  for (int k = 0; k <= 10; ++k) {
    for (int i = 0; i < 4; ++i) p->Q[k][i] = 1.0;
    for (int j = 0; j < 2; ++j) p->R[k][j] = 1e-2;
  }
  p->x0[0] = x00; p->x0[1] = x01;
}

/* which: 0 = SolveGoalConstraint, 1 = ControlBounds, 2 = SolverUnconstrained (iterations_max = 3).
 * out[0]=iterations, out[1]=status, out[2]=|x_N|, out[3..4]=u_0 */
int qo_kat_double_integrator(int which, double* out, int verbose) {
  static qo_problem p;
  qo_options o;
  qo_default_options(&o, QO_MODE_REFERENCE);
  o.verbose = verbose;
  o.penalty_scaling = 100.0;
  if (which == 2) {
    di_problem(&p, 1.0, 2.0);
    p.ncon = 0;
    o.iterations_max = 3;   /* TestDoubleIntegrator.cpp:129 */
  } else if (which == 0) {
    di_problem(&p, 1.0, 2.0);
    p.ncon = 1;
    p.con[0] = (qo_constraint){QO_EQUALITY, 4, 10, 11, goal_con, goal_jac, NULL, NULL};
  } else {
    di_problem(&p, 2.0, 2.0);
    o.penalty_initial = 100.0;
    p.ncon = 2;
    p.con[0] = (qo_constraint){QO_EQUALITY, 4, 10, 11, goal_con, goal_jac, NULL, NULL};
    p.con[1] = (qo_constraint){QO_INEQUALITY, 4, 0, 10, ubnd_con, ubnd_jac, NULL, NULL};
  }
  double X[11 * 4], U[10 * 2];
  memset(U, 0, sizeof U);
  qo_result r;
  qo_altro_solve(&p, &o, X, U, &r);
  out[0] = r.iterations;
  out[1] = r.status;
  double s = 0;
  for (int i = 0; i < 4; ++i) s += X[40 + i] * X[40 + i];
  out[2] = sqrt(s);
  out[3] = U[0];
  out[4] = U[1];
  return r.status;
}

/* ---- pendulum (AltroTestUtils.cpp:39-82) ---------------------------------- */
static void pend_f(void* ctx, double* xd, const double* x, const double* u) {
  (void)ctx;
  const double l = 0.5, g = 9.81, b = 0.1, m = 1.0 * l * l;
  xd[0] = x[1];
  xd[1] = u[0] / m - g * sin(x[0]) / l - b * x[1] / m;
}
static void pend_df(void* ctx, double* J, const double* x, const double* u) {
  (void)ctx; (void)u;
  const double l = 0.5, g = 9.81, b = 0.1, m = 1.0 * l * l;
  J[0] = 0.0;                 /* domega/dtheta */
  J[1] = -g * cos(x[0]) / l;  /* dalpha/dtheta */
  J[2] = 1.0;
  J[3] = -b / m;
  J[4] = 0.0;
  J[5] = 1 / m;
}
static void pend_dyn(void* ctx, int k, double* xn, const double* x, const double* u, float h) {
  (void)k;
  qo_midpoint_dynamics(2, 1, pend_f, ctx, xn, x, u, h);
}
static void pend_jac(void* ctx, int k, double* J, const double* x, const double* u, float h) {
  (void)k;
  qo_midpoint_jacobian(2, 1, pend_f, pend_df, ctx, J, x, u, h);
}

/* TestPendulum.cpp:13-43: xn(2) and J(2x3 col-major) at x=(0.1,-0.4), u=1.34, h=0.05f */
void qo_kat_pendulum_midpoint(double* xn, double* J) {
  const double x[2] = {0.1, -0.4}, u[1] = {1.34};
  const float h = 0.05f;
  pend_dyn(NULL, 0, xn, x, u, h);
  pend_jac(NULL, 0, J, x, u, h);
}

/* TestPendulum.cpp:45-115: out[0]=iterations, out[1]=status, out[2..3]=x_N */
int qo_kat_pendulum_swingup(double* out, int verbose) {
  static qo_problem p;
  memset(&p, 0, sizeof p);
  p.n = 2; p.m = 1; p.N = 50;
  const float tf = 3.0f;
  p.h = tf / (float)50.0;
  p.dyn = pend_dyn; p.jac = pend_jac;
  for (int k = 0; k <= 50; ++k) {
    for (int i = 0; i < 2; ++i) p.Q[k][i] = (k == 50) ? 1.0 : 1e-2;
    p.R[k][0] = 1e-3;
    p.xref[k][0] = M_PI;
  }
  qo_options o;
  qo_default_options(&o, QO_MODE_REFERENCE);
  o.iterations_max = 20;
  o.verbose = verbose;
  double X[51 * 2], U[50];
  for (int k = 0; k < 50; ++k) U[k] = 0.1;
  qo_result r;
  qo_altro_solve(&p, &o, X, U, &r);
  out[0] = r.iterations;
  out[1] = r.status;
  out[2] = X[100];
  out[3] = X[101];
  return r.status;
}

/* TestPendulum.cpp:117-203: N = 20, tf = 2, terminal equality x_N = (pi, 0).
 * out[0]=iterations, out[1]=status, out[2]=|x_N - xf| */
static void pend_goal_con(void* ctx, int k, double* c, const double* x, const double* u) {
  (void)ctx; (void)k; (void)u;
  c[0] = M_PI - x[0];
  c[1] = 0.0 - x[1];
}
static void pend_goal_jac(void* ctx, int k, double* J, const double* x, const double* u) {
  (void)ctx; (void)k; (void)x; (void)u;
  J[0 + 2 * 0] = -1.0; /* 2 x 3 col-major, -I on the state block */
  J[1 + 2 * 1] = -1.0;
}
int qo_kat_pendulum_goal(double* out, int verbose) {
  static qo_problem p;
  memset(&p, 0, sizeof p);
  p.n = 2; p.m = 1; p.N = 20;
  const float tf = 2.0f;
  p.h = tf / (float)20.0;
  p.dyn = pend_dyn; p.jac = pend_jac;
  for (int k = 0; k <= 20; ++k) {
    for (int i = 0; i < 2; ++i) p.Q[k][i] = (k == 20) ? 1.0 : 1e-2;
    p.R[k][0] = 1e-3;
    p.xref[k][0] = M_PI;
  }
  p.ncon = 1;
  p.con[0] = (qo_constraint){QO_EQUALITY, 2, 20, 21, pend_goal_con, pend_goal_jac, NULL, NULL};
  qo_options o;
  qo_default_options(&o, QO_MODE_REFERENCE);
  o.iterations_max = 100;
  o.verbose = verbose;
  double X[21 * 2], U[20];
  for (int k = 0; k < 20; ++k) U[k] = 0.1;
  qo_result r;
  qo_altro_solve(&p, &o, X, U, &r);
  out[0] = r.iterations;
  out[1] = r.status;
  out[2] = hypot(X[40] - M_PI, X[41]);
  return r.status;
}
