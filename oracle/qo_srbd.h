/*
 * qo_srbd.h -- CPU restatement of the quaternion single-rigid-body model of
 * zixinz990/quaternion-mpc.  TEST INFRASTRUCTURE ONLY.
 */
#ifndef QO_SRBD_H_
#define QO_SRBD_H_

#ifdef __cplusplus
extern "C" {
#endif

/* Per-instance constants of the model (what the reference's closures capture
 * by reference at QuatMpc.cpp:184-189). */
typedef struct qo_srbd_model {
  int nleg;                 /* contact points: 4 (Go1) or 8 (synthetic biped); 0 means 4 */
  double foot_pos_body[24]; /* 3 x nleg col-major (Eigen layout)              */
  double inertia[9];        /* row-major                                     */
  double inertia_inv[9];    /* filled by qo_srbd_prepare                     */
  double mass;
  double rot[9];            /* torso_rot_mat body->world, row-major          */
  double contacts[8];       /* 1 = stance; swing-leg forces are pinned to 0  */
  double g_body[3];         /* R' * (0,0,-9.81), filled by qo_srbd_prepare   */
  double moment_gravity[3]; /* c x (5.204 g_body), filled by prepare         */
} qo_srbd_model;

void qo_srbd_prepare(qo_srbd_model* m);

/* Utils::skew (Utils.cpp:101-105), row-major 3x3 */
void qo_skew(const double v[3], double S[9]);
/* QuaternionUtils::L (QuaternionUtils.cpp:30-37), row-major 4x4 */
void qo_quat_L(const double q[4], double L[16]);
/* QuaternionUtils::G = L(q) H (QuaternionUtils.cpp:48-52), row-major 4x3 */
void qo_quat_G(const double q[4], double G[12]);
/* cayley_map / inv_cayley_map / quat_mult / quat_conj (QuaternionUtils.cpp:10-28) */
void qo_cayley_map(const double phi[3], double q[4]);
void qo_inv_cayley_map(const double q[4], double phi[3]);
void qo_quat_mult(const double a[4], const double b[4], double out[4]);
void qo_quat_conj(const double q[4], double out[4]);

/* QuadrupedModel::ct_srb_quat_dynamics (AltroUtils.cpp:363-392) */
void qo_ct_srb_quat_dynamics(const qo_srbd_model* m, double* x_dot, const double* x,
                             const double* u);
/* QuadrupedModel::ct_srb_quat_jacobian (AltroUtils.cpp:395-439); jac is the
 * reference's buffer: 13 x 25 COLUMN-major, [d/dx | d/du]. */
void qo_ct_srb_quat_jacobian(const qo_srbd_model* m, double* jac, const double* x,
                             const double* u);

/* Generic continuous-time callbacks + explicit midpoint (AltroUtils.cpp:9-22,
 * 78-110).  `h` is float exactly as in altro::ExplicitDynamicsFunction. */
typedef void (*qo_ct_dyn_fn)(void* ctx, double* x_dot, const double* x, const double* u);
typedef void (*qo_ct_jac_fn)(void* ctx, double* jac /* n x (n+m) col-major */, const double* x,
                             const double* u);
void qo_midpoint_dynamics(int n, int m, qo_ct_dyn_fn f, void* ctx, double* xn, const double* x,
                          const double* u, float h);
void qo_midpoint_jacobian(int n, int m, qo_ct_dyn_fn f, qo_ct_jac_fn df, void* ctx, double* jac,
                          const double* x, const double* u, float h);

/* SRBD-specialised wrappers used by the MPC oracle. */
void qo_srbd_discrete_dynamics(const qo_srbd_model* m, double* xn, const double* x,
                               const double* u, float h);
void qo_srbd_discrete_jacobian(const qo_srbd_model* m, double* jac /* 13 x (13 + 3 nleg) col-major */,
                               const double* x, const double* u, float h);

/* Attitude Jacobian E(x) = blkdiag(I3, G(q), I3, I3), 13 x 12 row-major
 * (pattern at AltroUtils.cpp:153-157). */
void qo_srbd_error_jacobian(const double* x, double* E);
/* Abar = E(xn)' A E(x), Bbar = E(xn)' B  (AltroUtils.cpp:167-168); A|B taken
 * from the 13x25 col-major jac; outputs 12x12 row-major. */
void qo_srbd_project(const double* jac, const double* x, const double* xn, double* Abar,
                     double* Bbar);

/* Friction-cone closures (QuatMpc.cpp:47-52,194-215): c(24), and the 6x3 block
 * C_mat * R shared by all legs (row-major). */
void qo_cone_block(double mu, const double rot[9], double CR[18]);
void qo_cone_eval(double mu, double fz_max, const double rot[9], const double contacts[4],
                  const double* u, double* c);
/* same for n contact points (6 n rows) */
void qo_cone_eval_n(int n, double mu, double fz_max, const double rot[9], const double* contacts,
                    const double* u, double* c);

#ifdef __cplusplus
}
#endif
#endif
