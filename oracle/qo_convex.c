/*
 * qo_convex.c -- CPU restatement of legged::ConvexMpc::grf_update's problem
 * (legged_ctrl/src/mpc/ConvexMpc.cpp:81-198) and of the Euler-angle single-rigid-
 * body model it hands to the solver (legged_ctrl/src/utils/AltroUtils.cpp:224-359),
 * on top of the restated solver scheme (qo_altro.c).  TEST INFRASTRUCTURE ONLY.
 * PARITY UNPINNED (see qo_convex.h).
 */
#include "qo_convex.h"

#include <math.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>

#include "qo_altro.h"
#include "qo_linalg.h"
#include "qo_srbd.h"

/* I_world^-1 = Rz(yaw) diag(1/Ixx,1/Iyy,1/Izz) Rz(yaw)'  -- the closed form of
 * (Rz I Rz')^-1 at AltroUtils.cpp:274-288 (the reference inverts numerically). */
static void inertia_world_inv(const qo_convex_model* m, double cy, double sy, double W[9]) {
  const double a = 1.0 / m->inertia_diag[0], b = 1.0 / m->inertia_diag[1], c = 1.0 / m->inertia_diag[2];
  W[0] = cy * cy * a + sy * sy * b;
  W[1] = cy * sy * (a - b);
  W[2] = 0.0;
  W[3] = W[1];
  W[4] = sy * sy * a + cy * cy * b;
  W[5] = 0.0;
  W[6] = 0.0; W[7] = 0.0; W[8] = c;
}

void qo_ct_srb_dynamics(const qo_convex_model* m, double* xd, const double* x, const double* u) {
  const double sy = sin(x[2]), cy = cos(x[2]);
  /* Ac x: rpy rate = [[c,s,0],[-s,c,0],[0,0,1]] ang_vel  (AltroUtils.cpp:256-264) */
  xd[0] = cy * x[6] + sy * x[7];
  xd[1] = -sy * x[6] + cy * x[7];
  xd[2] = x[8];
  xd[3] = x[9]; xd[4] = x[10]; xd[5] = x[11];
  /* Bc u: I_world^-1 skew(r_i) u_i and u_i / m  (AltroUtils.cpp:284-288) */
  double tau[3] = {0, 0, 0}, F[3] = {0, 0, 0};
  for (int i = 0; i < 4; ++i) {
    if (m->contacts[i] == 0.0) continue; /* swing leg: force pinned to 0 by its cone rows */
    const double* r = &m->foot_pos[3 * i];
    const double* f = &u[3 * i];
    tau[0] += r[1] * f[2] - r[2] * f[1];
    tau[1] += r[2] * f[0] - r[0] * f[2];
    tau[2] += r[0] * f[1] - r[1] * f[0];
    F[0] += f[0]; F[1] += f[1]; F[2] += f[2];
  }
  double W[9];
  inertia_world_inv(m, cy, sy, W);
  for (int a = 0; a < 3; ++a) {
    xd[6 + a] = W[3 * a] * tau[0] + W[3 * a + 1] * tau[1] + W[3 * a + 2] * tau[2];
    xd[9 + a] = F[a] / m->mass;
  }
  xd[11] += -9.81; /* g_vec, AltroUtils.cpp:233 */
}

void qo_ct_srb_jacobian(const qo_convex_model* m, double* J, const double* x, const double* u) {
  (void)u;
  const int n = 12;
  memset(J, 0, sizeof(double) * 12 * 24);
  const double sy = sin(x[2]), cy = cos(x[2]);
  /* AltroUtils.cpp:354-357 */
  J[0 + n * 2] = x[7] * cy - x[6] * sy;
  J[1 + n * 2] = -x[6] * cy - x[7] * sy;
  J[0 + n * 6] = cy;  J[0 + n * 7] = sy;
  J[1 + n * 6] = -sy; J[1 + n * 7] = cy;
  J[2 + n * 8] = 1.0;
  for (int a = 0; a < 3; ++a) J[(3 + a) + n * (9 + a)] = 1.0;
  double W[9], S[9];
  inertia_world_inv(m, cy, sy, W);
  for (int i = 0; i < 4; ++i) {
    if (m->contacts[i] == 0.0) continue;
    qo_skew(&m->foot_pos[3 * i], S);
    for (int r = 0; r < 3; ++r)
      for (int c = 0; c < 3; ++c) {
        double s = 0.0;
        for (int t = 0; t < 3; ++t) s += W[3 * r + t] * S[3 * t + c];
        J[(6 + r) + n * (12 + 3 * i + c)] = s;
      }
    for (int a = 0; a < 3; ++a) J[(9 + a) + n * (12 + 3 * i + a)] = 1.0 / m->mass;
  }
}

static void ct_dyn_cb(void* ctx, double* xd, const double* x, const double* u) {
  qo_ct_srb_dynamics((const qo_convex_model*)ctx, xd, x, u);
}
static void ct_jac_cb(void* ctx, double* jac, const double* x, const double* u) {
  qo_ct_srb_jacobian((const qo_convex_model*)ctx, jac, x, u);
}
void qo_convex_discrete_dynamics(const qo_convex_model* m, double* xn, const double* x,
                                 const double* u, float h) {
  qo_midpoint_dynamics(12, 12, ct_dyn_cb, (void*)m, xn, x, u, h);
}
void qo_convex_discrete_jacobian(const qo_convex_model* m, double* jac, const double* x,
                                 const double* u, float h) {
  qo_midpoint_jacobian(12, 12, ct_dyn_cb, ct_jac_cb, (void*)m, jac, x, u, h);
}

/* gazebo_go1_convex_mpc.yaml:35-73; ConvexMpc.cpp:36-38 */
void qo_default_convex_params(qmpc_params* p, int32_t horizon, int32_t mode) {
  memset(p, 0, sizeof *p);
  p->model = QMPC_MODEL_CONVEX;
  p->horizon = horizon;
  p->h = (float)(5.0 / 1000.0);   /* mpc_update_period: 5.0 [ms] */
  p->h_ref = 5.0 / 1000.0;
  p->mass = 12.84;                /* hard-coded in the model, AltroUtils.cpp:239 */
  const double trunk[3] = {0.0168128557, 0.063009565, 0.0716547275}; /* AltroUtils.cpp:270-272 */
  for (int a = 0; a < 3; ++a) p->inertia[4 * a] = trunk[a];
  const double q[12] = {3.0, 3.0, 3.0, 1.0, 1.0, 20.0, 0.0, 0.0, 3.0, 2.0, 3.0, 2.0};
  memcpy(p->q_weights, q, sizeof q);
  for (int j = 0; j < 12; ++j) p->r_weights[j] = 0.000001;
  p->mu = 0.6;
  p->fz_max = 200.0;
  p->mode = mode;
  p->linesearch_max = 10;
  qo_options o;
  qo_default_options(&o, mode);
  p->iterations_max = o.iterations_max;
  p->penalty_initial = o.penalty_initial;
  p->penalty_scaling = o.penalty_scaling;
  p->penalty_max = o.penalty_max;
  p->tol_stationarity = o.tol_stationarity;
  p->tol_feasibility = o.tol_feasibility;
  p->tol_cost_intermediate = o.tol_cost_intermediate;
  p->tol_step = o.tol_step;
  p->ipm_mu0 = o.ipm_mu0;
  p->ipm_mu_final = o.ipm_mu_final;
  p->ipm_sigma = o.ipm_sigma;
  p->ipm_sigma_fast = o.ipm_sigma_fast;
  p->ipm_tau = o.ipm_tau;
  if (mode == QMPC_MODE_REFERENCE) p->iterations_max = 5; /* ConvexMpc.cpp:37 */
}

void qo_convex_build_reference(const qmpc_params* p, const qmpc_convex_input* in, double* xref,
                               double* uref) {
  int nc = 0;
  for (int i = 0; i < 4; ++i) if (in->contacts[i] != 0.0) nc++;
  memset(uref, 0, sizeof(double) * 12);
  for (int i = 0; i < 4; ++i) /* ConvexMpc.cpp:107-110 */
    uref[3 * i + 2] = p->mass * 9.81 / (double)nc * in->contacts[i];
  const double h_ms = p->h_ref * 1000.0;
  for (int k = 0; k <= p->horizon; ++k) { /* ConvexMpc.cpp:95-106 */
    double* xr = &xref[12 * k];
    memset(xr, 0, sizeof(double) * 12);
    xr[2] = in->euler[2] + in->yaw_rate_d * h_ms / 1000.0 * k;
    xr[3] = in->pos_d_world[0]; xr[4] = in->pos_d_world[1]; xr[5] = in->pos_d_world[2];
    xr[8] = in->yaw_rate_d;
    xr[9] = in->lin_vel_d_world[0]; xr[10] = in->lin_vel_d_world[1];
  }
}

typedef struct cvx_ctx {
  qo_convex_model model;
  double mu, fz_max;
  double contacts[4];
  double row_enable[24];
} cvx_ctx;

static const double kIdentity[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};

static void dyn_cb(void* ctx, int k, double* xn, const double* x, const double* u, float h) {
  (void)k;
  qo_convex_discrete_dynamics(&((cvx_ctx*)ctx)->model, xn, x, u, h);
}
static void jac_cb(void* ctx, int k, double* jac, const double* x, const double* u, float h) {
  (void)k;
  qo_convex_discrete_jacobian(&((cvx_ctx*)ctx)->model, jac, x, u, h);
}
/* ConvexMpc.cpp:126-136: the pyramid acts on the world-frame forces directly */
static void cone_con(void* ctx, int k, double* c, const double* x, const double* u) {
  (void)k; (void)x;
  const cvx_ctx* m = (const cvx_ctx*)ctx;
  qo_cone_eval(m->mu, m->fz_max, kIdentity, m->contacts, u, c);
}
/* ConvexMpc.cpp:14-32: 24 x 24 col-major, columns 12.. are the inputs */
static void cone_jac(void* ctx, int k, double* jac, const double* x, const double* u) {
  (void)k; (void)x; (void)u;
  const cvx_ctx* m = (const cvx_ctx*)ctx;
  double C[18];
  qo_cone_block(m->mu, kIdentity, C);
  for (int i = 0; i < 4; ++i) {
    if (m->contacts[i] == 0.0) continue;
    for (int r = 0; r < 6; ++r)
      for (int c = 0; c < 3; ++c) jac[(6 * i + r) + 24 * (12 + 3 * i + c)] = C[3 * r + c];
  }
}

static void setup_problem(const qmpc_params* p, const qmpc_convex_input* in, cvx_ctx* ctx,
                          qo_problem* prob) {
  const int N = p->horizon;
  memset(ctx, 0, sizeof *ctx);
  memcpy(ctx->model.foot_pos, in->foot_pos_abs_com, sizeof in->foot_pos_abs_com);
  for (int a = 0; a < 3; ++a) ctx->model.inertia_diag[a] = p->inertia[4 * a];
  ctx->model.mass = p->mass;
  ctx->mu = p->mu;
  ctx->fz_max = p->fz_max;
  for (int i = 0; i < 4; ++i) ctx->contacts[i] = (in->contacts[i] != 0.0) ? 1.0 : 0.0;
  memcpy(ctx->model.contacts, ctx->contacts, sizeof ctx->contacts);
  memset(prob->con, 0, sizeof prob->con);
  memset(prob->x0, 0, sizeof prob->x0);
  prob->n = 12; prob->m = 12; prob->N = N;
  prob->use_quaternion = 0;
  prob->quat_start_index = 0;
  prob->h = p->h;
  prob->dyn = dyn_cb; prob->jac = jac_cb; prob->dyn_ctx = ctx;
  double xref[(QMPC_MAX_HORIZON + 1) * 12], uref[12];
  qo_convex_build_reference(p, in, xref, uref);
  for (int k = 0; k <= N; ++k) { /* SetLQRCost, ConvexMpc.cpp:143-147 */
    memset(prob->Q[k], 0, sizeof prob->Q[k]);
    memcpy(prob->Q[k], p->q_weights, sizeof(double) * 12);
    memcpy(prob->R[k], p->r_weights, sizeof(double) * 12);
    memset(prob->xref[k], 0, sizeof prob->xref[k]);
    memcpy(prob->xref[k], &xref[12 * k], sizeof(double) * 12);
    memcpy(prob->uref[k], uref, sizeof(double) * 12);
    prob->w[k] = 0.0;
  }
  /* SetConstraint(..., 0, horizon + 1) (ConvexMpc.cpp:149-150) also names the terminal
   * knot, whose input does not enter the dynamics; the rows that matter are k = 0..N-1 */
  prob->ncon = 1;
  prob->con[0].type = QO_INEQUALITY;
  prob->con[0].p = 24;
  prob->con[0].k_start = 0;
  prob->con[0].k_stop = N;
  prob->con[0].con = cone_con;
  prob->con[0].jac = cone_jac;
  prob->con[0].ctx = ctx;
  for (int i = 0; i < 24; ++i) ctx->row_enable[i] = ctx->contacts[i / 6];
  prob->con[0].row_enable = ctx->row_enable;
  for (int a = 0; a < 3; ++a) { /* x_init, ConvexMpc.cpp:156-167 */
    prob->x0[a] = in->euler[a];
    prob->x0[3 + a] = in->pos_world[a];
    prob->x0[6 + a] = in->ang_vel_world[a];
    prob->x0[9 + a] = in->lin_vel_world[a];
  }
}

static void options_from_params(const qmpc_params* p, qo_options* o, int verbose) {
  memset(o, 0, sizeof *o);
  o->mode = p->mode;
  o->iterations_max = p->iterations_max;
  o->penalty_initial = p->penalty_initial;
  o->penalty_scaling = p->penalty_scaling;
  o->penalty_max = p->penalty_max;
  o->tol_stationarity = p->tol_stationarity;
  o->tol_feasibility = p->tol_feasibility;
  o->tol_cost_intermediate = p->tol_cost_intermediate;
  o->tol_step = p->tol_step;
  o->linesearch_max = p->linesearch_max;
  o->verbose = verbose;
  o->ipm_iterations_max = (p->mode == QMPC_MODE_CONVERGED) ? p->iterations_max : 0;
  o->ipm_mu0 = p->ipm_mu0;
  o->ipm_mu_final = p->ipm_mu_final;
  o->ipm_sigma = p->ipm_sigma;
  o->ipm_sigma_fast = p->ipm_sigma_fast;
  o->ipm_tau = p->ipm_tau;
}

static int input_is_finite(const qmpc_convex_input* in) {
  const double* v = (const double*)in;
  for (size_t i = 0; i < sizeof(qmpc_convex_input) / sizeof(double); ++i)
    if (!isfinite(v[i])) return 0;
  return 1;
}

/* set by qo_convex_solve_one_dual around its call: where the multipliers / slacks go */
static __thread double* tl_dual = NULL;
static __thread double* tl_slack = NULL;

int qo_convex_solve_one(const qmpc_params* p, const qmpc_convex_input* in, double* forces,
                        qmpc_info* info, double* traj_u, double* traj_x, int verbose) {
  const int N = p->horizon;
  qmpc_info inf;
  memset(&inf, 0, sizeof inf);
  memset(forces, 0, sizeof(double) * 12);
  int nc = 0;
  for (int i = 0; i < 4; ++i) if (in->contacts[i] != 0.0) nc++;
  if (!input_is_finite(in)) inf.status = QMPC_NAN_INPUT;
  else if (nc == 0) inf.status = QMPC_NO_CONTACT;
  if (inf.status != QMPC_OK) {
    if (info) *info = inf;
    if (traj_u) memset(traj_u, 0, sizeof(double) * N * 12);
    if (traj_x) memset(traj_x, 0, sizeof(double) * (N + 1) * 12);
    return inf.status;
  }
  static __thread cvx_ctx* ctx = NULL;
  static __thread qo_problem* prob = NULL;
  if (!ctx) {
    ctx = (cvx_ctx*)malloc(sizeof(cvx_ctx));
    prob = (qo_problem*)malloc(sizeof(qo_problem));
  }
  setup_problem(p, in, ctx, prob);
  qo_options o;
  options_from_params(p, &o, verbose);
  o.dual_out = tl_dual;
  o.slack_out = tl_slack;
  double X[(QMPC_MAX_HORIZON + 1) * 12], U[QMPC_MAX_HORIZON * 12];
  /* SetInput(u_ref) on all knots (ConvexMpc.cpp:176) */
  for (int k = 0; k < N; ++k) memcpy(&U[12 * k], prob->uref[0], sizeof(double) * 12);
  qo_result r;
  qo_altro_solve(prob, &o, X, U, &r);
  memcpy(forces, U, sizeof(double) * 12); /* GetInput(u, 0), ConvexMpc.cpp:186-187 */
  inf.status = r.status;
  inf.iterations = r.iterations;
  inf.cost = r.cost;
  inf.max_violation = r.max_violation;
  inf.last_step = r.last_step;
  inf.penalty = r.penalty;
  if (info) *info = inf;
  if (traj_u) memcpy(traj_u, U, sizeof(double) * N * 12);
  if (traj_x) memcpy(traj_x, X, sizeof(double) * (N + 1) * 12);
  return inf.status;
}

/* As qo_convex_solve_one, plus multipliers and slacks of the cone rows, [N][24] (certificate fixtures only) */
int qo_convex_solve_one_dual(const qmpc_params* p, const qmpc_convex_input* in, double* forces, qmpc_info* info,
                             double* traj_u, double* traj_x, double* dual, double* slack) {
  tl_dual = dual;
  tl_slack = slack;
  const int st = qo_convex_solve_one(p, in, forces, info, traj_u, traj_x, 0);
  tl_dual = NULL;
  tl_slack = NULL;
  return st;
}

typedef struct batch_job {
  const qmpc_params* p;
  const qmpc_convex_input* in;
  double* forces;
  qmpc_info* info;
  double* traj_u;
  double* traj_x;
  int begin, end;
} batch_job;

static void* batch_worker(void* arg) {
  batch_job* j = (batch_job*)arg;
  const int N = j->p->horizon;
  for (int b = j->begin; b < j->end; ++b)
    qo_convex_solve_one(j->p, &j->in[b], &j->forces[12 * b], j->info ? &j->info[b] : NULL,
                        j->traj_u ? &j->traj_u[(size_t)b * N * 12] : NULL,
                        j->traj_x ? &j->traj_x[(size_t)b * (N + 1) * 12] : NULL, 0);
  return NULL;
}

int qo_convex_solve_batch(const qmpc_params* p, int32_t batch, const qmpc_convex_input* in,
                          double* forces, qmpc_info* info, double* traj_u, double* traj_x,
                          int32_t threads) {
  if (threads < 1) threads = 1;
  if (threads > batch) threads = batch > 0 ? batch : 1;
  batch_job* jobs = (batch_job*)calloc((size_t)threads, sizeof(batch_job));
  pthread_t* tid = (pthread_t*)calloc((size_t)threads, sizeof(pthread_t));
  for (int t = 0; t < threads; ++t) {
    jobs[t].p = p; jobs[t].in = in; jobs[t].forces = forces; jobs[t].info = info;
    jobs[t].traj_u = traj_u; jobs[t].traj_x = traj_x;
    jobs[t].begin = (int)((long long)batch * t / threads);
    jobs[t].end = (int)((long long)batch * (t + 1) / threads);
  }
  if (threads == 1) {
    batch_worker(&jobs[0]);
  } else {
    for (int t = 0; t < threads; ++t) pthread_create(&tid[t], NULL, batch_worker, &jobs[t]);
    for (int t = 0; t < threads; ++t) pthread_join(tid[t], NULL);
  }
  free(jobs);
  free(tid);
  return 0;
}

int qo_convex_linearize(const qmpc_params* p, int32_t batch, const qmpc_convex_input* in,
                        double* A, double* B, double* X) {
  const int N = p->horizon;
  cvx_ctx* ctx = (cvx_ctx*)malloc(sizeof(cvx_ctx));
  qo_problem* prob = (qo_problem*)malloc(sizeof(qo_problem));
  for (int b = 0; b < batch; ++b) {
    setup_problem(p, &in[b], ctx, prob);
    double* Xb = &X[(size_t)b * (N + 1) * 12];
    memcpy(Xb, prob->x0, sizeof(double) * 12);
    double jac[12 * 24];
    for (int k = 0; k < N; ++k)
      qo_convex_discrete_dynamics(&ctx->model, &Xb[12 * (k + 1)], &Xb[12 * k], prob->uref[0], p->h);
    for (int k = 0; k < N; ++k) {
      qo_convex_discrete_jacobian(&ctx->model, jac, &Xb[12 * k], prob->uref[0], p->h);
      double* Ak = &A[((size_t)b * N + k) * 144];
      double* Bk = &B[((size_t)b * N + k) * 144];
      for (int r = 0; r < 12; ++r)
        for (int c = 0; c < 12; ++c) {
          Ak[12 * r + c] = jac[r + 12 * c];
          Bk[12 * r + c] = jac[r + 12 * (12 + c)];
        }
    }
  }
  free(ctx);
  free(prob);
  return 0;
}

void qo_convex_step(const qmpc_params* p, const qmpc_convex_input* in, const double* x,
                    const double* u, double* xn) {
  qo_convex_model m;
  memcpy(m.foot_pos, in->foot_pos_abs_com, sizeof m.foot_pos);
  for (int a = 0; a < 3; ++a) m.inertia_diag[a] = p->inertia[4 * a];
  m.mass = p->mass;
  for (int i = 0; i < 4; ++i) m.contacts[i] = (in->contacts[i] != 0.0) ? 1.0 : 0.0;
  qo_convex_discrete_dynamics(&m, xn, x, u, p->h);
}
