/*
 * qo_quatmpc.c -- CPU restatement of legged::QuatMpc::grf_update's problem
 * construction and solve (legged_ctrl/src/mpc/QuatMpc.cpp:109-276) on top of
 * the restated model (qo_srbd.c) and solver scheme (qo_altro.c).
 * TEST INFRASTRUCTURE ONLY: it checks the HIP path and is timed as the
 * cpu_baseline ("port") in bench.py; the product never calls it.
 */
#include "qo_quatmpc.h"

#include <math.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>

#include "qo_altro.h"
#include "qo_linalg.h"
#include "qo_srbd.h"

/* legged_ctrl/config/gazebo_go1_quat_mpc.yaml:36-75,115-122; QuatMpc.cpp:21-26,182 */
void qo_default_params(qmpc_params* p, int32_t horizon, int32_t mode) {
  memset(p, 0, sizeof *p);
  p->horizon = horizon;
  p->h = (float)(10.0 / 1000.0);  /* mpc_update_period: 10.0 [ms] -> float seconds */
  p->h_ref = 10.0 / 1000.0;
  p->mass = 12.84;
  const double trunk[3] = {0.0168128557, 0.063009565, 0.0716547275};
  for (int a = 0; a < 3; ++a) p->inertia[4 * a] = 1.2 * trunk[a]; /* QuatMpc.cpp:182 */
  const double q[13] = {2.5, 2.5, 10.0, 0, 0, 0, 0, 0.1, 0.1, 0.1, 0.15, 0.15, 0.15};
  memcpy(p->q_weights, q, sizeof q);
  for (int j = 0; j < 12; ++j) p->r_weights[j] = 0.000001;
  p->w = 50.0;
  p->mu = 0.7;
  p->fz_max = 100.0;
  p->mode = mode;
  p->linesearch_max = 10;
  p->drop_ang_vel = 1;
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
  if (mode == QMPC_MODE_REFERENCE) {
    p->iterations_max = 10;    /* QuatMpc.cpp:22 */
    p->penalty_scaling = 20.0; /* QuatMpc.cpp:26 */
  }
}

typedef struct mpc_ctx {
  qo_srbd_model model;
  double mu, fz_max;
  double CR[18];
  double row_enable[48];
} mpc_ctx;

/* the fields of qmpc_input / qmpc_input8, by pointer */
typedef struct in_view {
  int nleg;
  const double *quat, *rot, *lin_vel_body, *ang_vel_body, *foot, *contacts, *pos_ref, *vel_ref, *acc_ref, *quat_d;
} in_view;
static in_view view4(const qmpc_input* in) {
  in_view v = {4, in->quat, in->rot, in->lin_vel_body, in->ang_vel_body, in->foot_pos_body, in->contacts,
               in->pos_ref_body, in->vel_ref_body, in->acc_ref_body, in->quat_d};
  return v;
}
static in_view view8(const qmpc_input8* in) {
  in_view v = {8, in->quat, in->rot, in->lin_vel_body, in->ang_vel_body, in->foot_pos_body, in->contacts,
               in->pos_ref_body, in->vel_ref_body, in->acc_ref_body, in->quat_d};
  return v;
}

static void dyn_cb(void* ctx, int k, double* xn, const double* x, const double* u, float h) {
  (void)k;
  qo_srbd_discrete_dynamics(&((mpc_ctx*)ctx)->model, xn, x, u, h);
}
static void jac_cb(void* ctx, int k, double* jac, const double* x, const double* u, float h) {
  (void)k;
  qo_srbd_discrete_jacobian(&((mpc_ctx*)ctx)->model, jac, x, u, h);
}
/* QuatMpc.cpp:194-205 */
static void cone_con(void* ctx, int k, double* c, const double* x, const double* u) {
  (void)k; (void)x;
  const mpc_ctx* m = (const mpc_ctx*)ctx;
  qo_cone_eval_n(m->model.nleg, m->mu, m->fz_max, m->model.rot, m->model.contacts, u, c);
}
/* QuatMpc.cpp:207-215: 24 x 24 col-major, block C_mat*R at rows 6i, cols 12+3i.
 * Swing-leg blocks are left zero (their forces are pinned to 0). */
static void cone_jac(void* ctx, int k, double* jac, const double* x, const double* u) {
  (void)k; (void)x; (void)u;
  const mpc_ctx* m = (const mpc_ctx*)ctx;
  const int rows = 6 * m->model.nleg;
  for (int i = 0; i < m->model.nleg; ++i) {
    if (m->model.contacts[i] == 0.0) continue;
    for (int r = 0; r < 6; ++r)
      for (int c = 0; c < 3; ++c) jac[(6 * i + r) + rows * (12 + 3 * i + c)] = m->CR[3 * r + c];
  }
}

static void build_reference_v(const qmpc_params* p, const in_view* in, double* xref, double* uref) {
  /* QuatMpc.cpp:118-125 */
  int nc = 0;
  for (int i = 0; i < in->nleg; ++i) if (in->contacts[i] != 0.0) nc++;
  memset(uref, 0, sizeof(double) * 3 * in->nleg);
  for (int i = 0; i < in->nleg; ++i)
    uref[3 * i + 2] = in->contacts[i] * p->mass * 9.81 / (double)nc;
  /* QuatMpc.cpp:148-176 (h there is in ms: i*h/1000.0, left-to-right) */
  const double h_ms = p->h_ref * 1000.0;
  for (int k = 0; k <= p->horizon; ++k) {
    double* xr = &xref[13 * k];
    memset(xr, 0, sizeof(double) * 13);
    const double t = (double)k * p->h_ref;
    xr[0] = in->pos_ref[0] + in->vel_ref[0] * k * h_ms / 1000.0 + 0.5 * in->acc_ref[0] * t * t;
    xr[1] = in->pos_ref[1] + in->vel_ref[1] * k * h_ms / 1000.0 + 0.5 * in->acc_ref[1] * t * t;
    xr[2] = in->pos_ref[2] + 0.5 * in->acc_ref[2] * t * t;
    for (int a = 0; a < 4; ++a) xr[3 + a] = in->quat_d[a];
    for (int a = 0; a < 3; ++a) xr[7 + a] = in->vel_ref[a] + in->acc_ref[a] * t;
  }
}
void qo_build_reference(const qmpc_params* p, const qmpc_input* in, double* xref, double* uref) {
  const in_view v = view4(in);
  build_reference_v(p, &v, xref, uref);
}

static int record_is_finite(const void* in, size_t bytes) {
  const double* v = (const double*)in;
  for (size_t i = 0; i < bytes / sizeof(double); ++i)
    if (!isfinite(v[i])) return 0;
  return 1;
}

static void setup_problem(const qmpc_params* p, const in_view* in, mpc_ctx* ctx, qo_problem* prob) {
  const int N = p->horizon, nl = in->nleg, m = 3 * nl;
  memset(ctx, 0, sizeof *ctx);
  ctx->model.nleg = nl;
  memcpy(ctx->model.foot_pos_body, in->foot, sizeof(double) * 3 * nl);
  memcpy(ctx->model.inertia, p->inertia, sizeof p->inertia);
  ctx->model.mass = p->mass;
  memcpy(ctx->model.rot, in->rot, sizeof(double) * 9);
  for (int i = 0; i < nl; ++i) ctx->model.contacts[i] = (in->contacts[i] != 0.0) ? 1.0 : 0.0;
  qo_srbd_prepare(&ctx->model);
  ctx->mu = p->mu;
  ctx->fz_max = p->fz_max;
  qo_cone_block(p->mu, in->rot, ctx->CR);

  memset(prob->con, 0, sizeof prob->con);
  memset(prob->x0, 0, sizeof prob->x0);
  prob->n = 13; prob->m = m; prob->N = N;
  prob->use_quaternion = 1;       /* QuatMpc.cpp:24 */
  prob->quat_start_index = 3;     /* QuatMpc.cpp:25 */
  prob->h = p->h;
  prob->dyn = dyn_cb; prob->jac = jac_cb; prob->dyn_ctx = ctx;
  double xref[(QMPC_MAX_HORIZON + 1) * 13], uref[24];
  build_reference_v(p, in, xref, uref);
  for (int k = 0; k <= N; ++k) {
    memcpy(prob->Q[k], p->q_weights, sizeof(double) * 13);
    for (int j = 0; j < m; ++j) prob->R[k][j] = p->r_weights[j % 12];
    memcpy(prob->xref[k], &xref[13 * k], sizeof(double) * 13);
    memcpy(prob->uref[k], uref, sizeof(double) * m);
    prob->w[k] = p->w;
  }
  /* SetConstraint(..., 24, INEQUALITY, "friction cone", 0, horizon): knots 0..N-1 */
  prob->ncon = 1;
  prob->con[0].type = QO_INEQUALITY;
  prob->con[0].p = 6 * nl;
  prob->con[0].k_start = 0;
  prob->con[0].k_stop = N;
  prob->con[0].con = cone_con;
  prob->con[0].jac = cone_jac;
  prob->con[0].ctx = ctx;
  for (int i = 0; i < 6 * nl; ++i) ctx->row_enable[i] = ctx->model.contacts[i / 6];
  prob->con[0].row_enable = ctx->row_enable;
  /* x_init, QuatMpc.cpp:231-246 (angular velocity dropped by the ';' at :242) */
  prob->x0[3] = in->quat[0]; prob->x0[4] = in->quat[1];
  prob->x0[5] = in->quat[2]; prob->x0[6] = in->quat[3];
  for (int a = 0; a < 3; ++a) {
    prob->x0[7 + a] = in->lin_vel_body[a];
    prob->x0[10 + a] = p->drop_ang_vel ? 0.0 : in->ang_vel_body[a];
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

static __thread const double* warm_u_init = NULL;   /* qo_solve_one_warm: the previous solution [N][m], or NULL */

static int solve_one_v(const qmpc_params* p, const in_view* in, const void* rec, size_t rec_bytes,
                       double* forces, qmpc_info* info, double* traj_u, double* traj_x, int verbose,
                       double* dual, double* slack) {
  const int N = p->horizon, m = 3 * in->nleg;
  qmpc_info inf;
  memset(&inf, 0, sizeof inf);
  memset(forces, 0, sizeof(double) * m);
  int nc = 0;
  for (int i = 0; i < in->nleg; ++i) if (in->contacts[i] != 0.0) nc++;
  if (!record_is_finite(rec, rec_bytes)) inf.status = QMPC_NAN_INPUT;
  else if (nc == 0) inf.status = QMPC_NO_CONTACT;
  if (inf.status != QMPC_OK) {
    if (info) *info = inf;
    if (traj_u) memset(traj_u, 0, sizeof(double) * N * m);
    if (traj_x) memset(traj_x, 0, sizeof(double) * (N + 1) * 13);
    return inf.status;
  }
  static __thread mpc_ctx* ctx = NULL;     /* per-thread, reused */
  static __thread qo_problem* prob = NULL;
  if (!ctx) {
    ctx = (mpc_ctx*)malloc(sizeof(mpc_ctx));
    prob = (qo_problem*)malloc(sizeof(qo_problem));
  }
  setup_problem(p, in, ctx, prob);
  qo_options o;
  options_from_params(p, &o, verbose);
  o.dual_out = dual;
  o.slack_out = slack;
  double X[(QMPC_MAX_HORIZON + 1) * 13], U[QMPC_MAX_HORIZON * 24];
  /* initial guess: SetInput(u_ref) on all knots (QuatMpc.cpp:253); the state
   * guess x_ref (:250-252) is overwritten by the solver's initial rollout */
  for (int k = 0; k < N; ++k) memcpy(&U[m * k], prob->uref[0], sizeof(double) * m);
  if (warm_u_init) {
    /* warm start (qmpc_solve_warm, qmpc_loop_params.warm_start): the previous solution shifted by one knot, the last knot
     * repeated; swing legs are pinned to 0, a leg that has just landed (previous force exactly 0) starts from u_ref */
    for (int j = 0; j < m; ++j) {
      const int stance = in->contacts[j / 3] != 0.0;
      for (int k = 0; k < N; ++k) {
        const double prev = warm_u_init[m * ((k + 1 < N) ? k + 1 : k) + j];
        U[m * k + j] = !stance ? 0.0 : (prev != 0.0 ? prev : prob->uref[0][j]);
      }
    }
  }
  qo_result r;
  qo_altro_solve(prob, &o, X, U, &r);
  memcpy(forces, U, sizeof(double) * m); /* GetInput(u, 0), QuatMpc.cpp:264-265 */
  inf.status = r.status;
  inf.iterations = r.iterations;
  inf.cost = r.cost;
  inf.max_violation = r.max_violation;
  inf.last_step = r.last_step;
  inf.penalty = r.penalty;
  if (info) *info = inf;
  if (traj_u) memcpy(traj_u, U, sizeof(double) * N * m);
  if (traj_x) memcpy(traj_x, X, sizeof(double) * (N + 1) * 13);
  return inf.status;
}

int qo_solve_one(const qmpc_params* p, const qmpc_input* in, double* forces, qmpc_info* info,
                 double* traj_u, double* traj_x, int verbose) {
  const in_view v = view4(in);
  return solve_one_v(p, &v, in, sizeof *in, forces, info, traj_u, traj_x, verbose, NULL, NULL);
}
int qo_solve8_one(const qmpc_params* p, const qmpc_input8* in, double* forces, qmpc_info* info,
                  double* traj_u, double* traj_x, int verbose) {
  const in_view v = view8(in);
  return solve_one_v(p, &v, in, sizeof *in, forces, info, traj_u, traj_x, verbose, NULL, NULL);
}

/* One instance started from u_init [N][12] (NULL: from u_ref, i.e. qo_solve_one) */
int qo_solve_one_warm(const qmpc_params* p, const qmpc_input* in, const double* u_init, double* forces, qmpc_info* info,
                      double* traj_u) {
  const in_view v = view4(in);
  warm_u_init = u_init;
  const int st = solve_one_v(p, &v, in, sizeof *in, forces, info, traj_u, NULL, 0, NULL, NULL);
  warm_u_init = NULL;
  return st;
}

/* One instance with the multipliers and slacks of the cone rows, [N][6 nleg] each (certificate fixtures only) */
int qo_solve_one_dual(const qmpc_params* p, const qmpc_input* in, double* forces, qmpc_info* info,
                      double* traj_u, double* traj_x, double* dual, double* slack) {
  const in_view v = view4(in);
  return solve_one_v(p, &v, in, sizeof *in, forces, info, traj_u, traj_x, 0, dual, slack);
}
int qo_solve8_one_dual(const qmpc_params* p, const qmpc_input8* in, double* forces, qmpc_info* info,
                       double* traj_u, double* traj_x, double* dual, double* slack) {
  const in_view v = view8(in);
  return solve_one_v(p, &v, in, sizeof *in, forces, info, traj_u, traj_x, 0, dual, slack);
}

typedef struct batch_job {
  const qmpc_params* p;
  const qmpc_input* in;
  const qmpc_input8* in8;
  double* forces;
  qmpc_info* info;
  double* traj_u;
  double* traj_x;
  int begin, end;
} batch_job;

static void* batch_worker(void* arg) {
  batch_job* j = (batch_job*)arg;
  const int N = j->p->horizon;
  for (int b = j->begin; b < j->end; ++b) {
    if (j->in8)
      qo_solve8_one(j->p, &j->in8[b], &j->forces[24 * (size_t)b], j->info ? &j->info[b] : NULL,
                    j->traj_u ? &j->traj_u[(size_t)b * N * 24] : NULL,
                    j->traj_x ? &j->traj_x[(size_t)b * (N + 1) * 13] : NULL, 0);
    else
      qo_solve_one(j->p, &j->in[b], &j->forces[12 * (size_t)b], j->info ? &j->info[b] : NULL,
                   j->traj_u ? &j->traj_u[(size_t)b * N * 12] : NULL,
                   j->traj_x ? &j->traj_x[(size_t)b * (N + 1) * 13] : NULL, 0);
  }
  return NULL;
}

static int solve_batch_any(const qmpc_params* p, int32_t batch, const qmpc_input* in, const qmpc_input8* in8,
                           double* forces, qmpc_info* info, double* traj_u, double* traj_x, int32_t threads) {
  if (threads < 1) threads = 1;
  if (threads > batch) threads = batch > 0 ? batch : 1;
  batch_job* jobs = (batch_job*)calloc((size_t)threads, sizeof(batch_job));
  pthread_t* tid = (pthread_t*)calloc((size_t)threads, sizeof(pthread_t));
  for (int t = 0; t < threads; ++t) {
    jobs[t].p = p; jobs[t].in = in; jobs[t].in8 = in8; jobs[t].forces = forces; jobs[t].info = info;
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

int qo_solve_batch(const qmpc_params* p, int32_t batch, const qmpc_input* in, double* forces,
                   qmpc_info* info, double* traj_u, double* traj_x, int32_t threads) {
  return solve_batch_any(p, batch, in, NULL, forces, info, traj_u, traj_x, threads);
}
int qo_solve8_batch(const qmpc_params* p, int32_t batch, const qmpc_input8* in, double* forces,
                    qmpc_info* info, double* traj_u, double* traj_x, int32_t threads) {
  return solve_batch_any(p, batch, NULL, in, forces, info, traj_u, traj_x, threads);
}

/* BASELINE.json config 5: SYNTHETIC 30 kg biped with two 0.2 x 0.1 m feet (4 corner
 * contact points each); the humanoid branch itself is not in the reference checkout,
 * so nothing upstream pins these values. */
void qo_default_biped8_params(qmpc_params* p, int32_t horizon, int32_t mode) {
  qo_default_params(p, horizon, mode);
  p->model = QMPC_MODEL_QUAT8;
  p->mass = 30.0;
  memset(p->inertia, 0, sizeof p->inertia);
  p->inertia[0] = 1.2; p->inertia[4] = 1.0; p->inertia[8] = 0.3;
  p->fz_max = 250.0;
}

int qo_linearize(const qmpc_params* p, int32_t batch, const qmpc_input* in, double* Abar,
                 double* Bbar, double* X) {
  const int N = p->horizon;
  mpc_ctx* ctx = (mpc_ctx*)malloc(sizeof(mpc_ctx));
  qo_problem* prob = (qo_problem*)malloc(sizeof(qo_problem));
  for (int b = 0; b < batch; ++b) {
    const in_view v = view4(&in[b]);
    setup_problem(p, &v, ctx, prob);
    double* Xb = &X[(size_t)b * (N + 1) * 13];
    memcpy(Xb, prob->x0, sizeof(double) * 13);
    double jac[13 * 25];
    for (int k = 0; k < N; ++k)
      qo_srbd_discrete_dynamics(&ctx->model, &Xb[13 * (k + 1)], &Xb[13 * k], prob->uref[0], p->h);
    for (int k = 0; k < N; ++k) {
      qo_srbd_discrete_jacobian(&ctx->model, jac, &Xb[13 * k], prob->uref[0], p->h);
      qo_srbd_project(jac, &Xb[13 * k], &Xb[13 * (k + 1)],
                      &Abar[((size_t)b * N + k) * 144], &Bbar[((size_t)b * N + k) * 144]);
    }
  }
  free(ctx);
  free(prob);
  return 0;
}
