/*
 * qo_altro.h -- CPU restatement of the AL-iLQR scheme of the reference's
 * external solver (zixinz990/altro @ b47202ff, fork of bjack205/altro "ALTRO-C",
 * pinned by legged_ctrl/CMakeLists.txt:34-40; NOT vendored in the reference
 * tree and not available offline).  TEST INFRASTRUCTURE ONLY.
 *
 * The scheme is restated from the published algorithm (Howell, Jackson,
 * Manchester, "ALTRO", IROS 2019; Jackson et al., "Planning with Attitude",
 * RA-L 2021 for the quaternion error state) and pinned on the reference's own
 * call sites and known-answer tests:
 *   - API conventions  : legged_ctrl/src/mpc/QuatMpc.cpp:218-256
 *   - generic AL KATs   : legged_ctrl/src/test/test_altro/TestDoubleIntegrator.cpp
 *                         (GetIterations()==3 at :255, ==5 at :374)
 *   - nonlinear iLQR KAT: legged_ctrl/src/test/test_altro/TestPendulum.cpp:111-114
 * See SURVEY.md Appendix B for the outer/inner logic that reproduces those
 * iteration counts.
 */
#ifndef QO_ALTRO_H_
#define QO_ALTRO_H_

#ifdef __cplusplus
extern "C" {
#endif

#define QO_MAXN 16   /* state dim     */
#define QO_MAXM 24   /* input dim     */
#define QO_MAXP 48   /* rows per constraint */
#define QO_MAXH 64   /* horizon       */
#define QO_MAXCON 4

enum { QO_EQUALITY = 0, QO_INEQUALITY = 1,
       QO_SOC = 2 /* c(x,u) = (v, t) in the second-order cone |v| <= t; reference (AL) mode only,
                     at most QO_SOC_MAXP rows */ };
#define QO_SOC_MAXP 8
enum { QO_MODE_CONVERGED = 0, QO_MODE_REFERENCE = 1 };
enum {
  QO_STATUS_OK = 0,
  QO_STATUS_MAX_ITER = 1,
  QO_STATUS_LINESEARCH_FAIL = 4,
  QO_STATUS_NOT_PD = 5
};

/* altro::ExplicitDynamicsFunction / Jacobian (float h!), AltroUtils.cpp:9-10,78-79 */
typedef void (*qo_dyn_fn)(void* ctx, int k, double* xn, const double* x, const double* u, float h);
typedef void (*qo_jac_fn)(void* ctx, int k, double* jac /* n x (n+m) col-major */,
                          const double* x, const double* u, float h);
/* altro::ConstraintFunction / Jacobian; jac is p x (ne+m) col-major, i.e. in
 * ERROR-state coordinates when use_quaternion (QuatMpc.cpp:210-214 maps 24x24) */
typedef void (*qo_con_fn)(void* ctx, int k, double* c, const double* x, const double* u);
typedef void (*qo_conjac_fn)(void* ctx, int k, double* jac, const double* x, const double* u);

typedef struct qo_constraint {
  int type;           /* QO_EQUALITY / QO_INEQUALITY                          */
  int p;              /* rows                                                  */
  int k_start, k_stop;/* knots [k_start, k_stop)  (k_stop exclusive) */
  qo_con_fn con;
  qo_conjac_fn jac;
  void* ctx;
  const double* row_enable; /* optional p flags; 0 = row identically inactive
                               (swing-leg cone rows: forces pinned to 0)      */
} qo_constraint;

typedef struct qo_options {
  int mode;
  int iterations_max;
  double penalty_initial, penalty_scaling, penalty_max;
  double tol_stationarity, tol_feasibility, tol_cost_intermediate;
  double tol_step;
  int linesearch_max;
  /* 0: backtracking on the sufficient-decrease test (use_backtracking_linesearch = true: what every caller on the path
   * selects, QuatMpc.cpp:23, ConvexMpc.cpp:38, TestBicycle.cpp:154); 1: the interpolating strong-Wolfe search that is the
   * upstream default (bracketing + cubic-interpolation zoom on merit and merit slope), RECALLED -- the fork is not in the
   * reference tree -- and used by the generic known-answer tests only                                                    */
  int linesearch_cubic;
  int verbose;
  /* converged mode: primal-dual interior point on the same Riccati core,
   * run until barrier <= ipm_mu_final, |c+s| <= tol_feasibility and the last
   * input step <= tol_step (the reference's own AL scheme stalls on states
   * whose optimum sits on many cone faces)                     */
  int ipm_iterations_max;
  double ipm_mu0;          /* initial barrier: s0 = max(-c,1), lambda0 = mu0/s0   */
  double ipm_mu_final;     /* hand over to the AL polish below this barrier   */
  double ipm_sigma;        /* centering parameter                              */
  double ipm_sigma_fast;   /* centering once full steps are being taken        */
  double ipm_tau;          /* fraction to the boundary                         */
  /* optional exports of the converged mode (NULL = off): multipliers and slacks of the inequality rows, laid out
   * [k = 0..N-1][constraint ci in order][row 0..p-1]; rows that are switched off (row_enable == 0) report 0 */
  double* dual_out;
  double* slack_out;
  /* optional warm start of the reference (AL) mode across solves (NULL = multipliers 0, penalty_initial): in/out
   * multipliers [k = 0..N][constraint ci][QO_MAXP] and the penalty; an MPC loop that re-solves a shifted problem
   * (TestBicycle.cpp:165-200) keeps both between calls */
  double* dual_io;
  double* penalty_io;
} qo_options;

typedef struct qo_problem {
  int n, m, N;
  int use_quaternion, quat_start_index;   /* AltroOptions, QuatMpc.cpp:24-25   */
  float h;                                 /* SetTimeStep, QuatMpc.cpp:224      */
  qo_dyn_fn dyn;
  qo_jac_fn jac;
  void* dyn_ctx;
  /* SetQuaternionCost / SetLQRCost per knot k = 0..N (QuatMpc.cpp:226-228):
   *   0.5 (x-xr)'Q(x-xr) + w (1 - |qr'q|) + [k<N] 0.5 (u-ur)'R(u-ur)          */
  double Q[QO_MAXH + 1][QO_MAXN];
  double R[QO_MAXH + 1][QO_MAXM];
  double xref[QO_MAXH + 1][QO_MAXN];
  double uref[QO_MAXH + 1][QO_MAXM];
  double w[QO_MAXH + 1];
  int ncon;
  qo_constraint con[QO_MAXCON];
  double x0[QO_MAXN];
} qo_problem;

typedef struct qo_result {
  int status;
  int iterations;
  double cost;            /* un-augmented objective at the solution            */
  double max_violation;
  double stationarity;
  double last_step;
  double penalty;
  int dual_updates;
  int linesearch_halvings;
  int ipm_iterations;
  double ipm_mu;
} qo_result;

void qo_default_options(qo_options* o, int mode);

/* Solve.  X: (N+1) x n, U: N x m, row-major.  On entry U holds the initial
 * input guess (SetInput); X[0] is overwritten by x0 and the rest by the
 * rollout (ALTRO's initial open-loop rollout).  Returns result.status. */
int qo_altro_solve(const qo_problem* prob, const qo_options* opts, double* X, double* U,
                   qo_result* res);

#ifdef __cplusplus
}
#endif
#endif
