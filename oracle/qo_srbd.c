/*
 * qo_srbd.c -- CPU restatement (plain C, fp64) of the quaternion SRBD model on
 * the quaternion-MPC hot path.  TEST INFRASTRUCTURE ONLY: it is the checker for
 * the HIP path, never the thing measured or shipped.
 *
 * Each function cites the reference file:line (relative to the upstream repo
 * zixinz990/quaternion-mpc, legged_ctrl/) whose arithmetic it restates.
 * Pinned by tests/test_oracle_golden.py against the reference's own golden
 * trajectories (tests/golden/quat_mpc_test.json, trot_quat_mpc_test.json).
 */
#include "qo_srbd.h"

#include <math.h>
#include <string.h>

#include "qo_linalg.h"

/* src/utils/Utils.cpp:101-105 */
void qo_skew(const double v[3], double S[9]) {
  S[0] = 0.0;   S[1] = -v[2]; S[2] = v[1];
  S[3] = v[2];  S[4] = 0.0;   S[5] = -v[0];
  S[6] = -v[1]; S[7] = v[0];  S[8] = 0.0;
}

/* src/utils/QuaternionUtils.cpp:30-37:  L = [[s, -v'],[v, s I + skew(v)]] */
void qo_quat_L(const double q[4], double L[16]) {
  const double s = q[0], x = q[1], y = q[2], z = q[3];
  L[0] = s;  L[1] = -x;  L[2] = -y;  L[3] = -z;
  L[4] = x;  L[5] = s;   L[6] = -z;  L[7] = y;
  L[8] = y;  L[9] = z;   L[10] = s;  L[11] = -x;
  L[12] = z; L[13] = -y; L[14] = x;  L[15] = s;
}

/* src/utils/QuaternionUtils.cpp:48-52:  G = L(q) [0; I3]  (4x3) */
void qo_quat_G(const double q[4], double G[12]) {
  const double s = q[0], x = q[1], y = q[2], z = q[3];
  G[0] = -x; G[1] = -y;  G[2] = -z;
  G[3] = s;  G[4] = -z;  G[5] = y;
  G[6] = z;  G[7] = s;   G[8] = -x;
  G[9] = -y; G[10] = x;  G[11] = s;
}

/* src/utils/QuaternionUtils.cpp:10-14 */
void qo_cayley_map(const double phi[3], double q[4]) {
  const double nrm = sqrt(phi[0] * phi[0] + phi[1] * phi[1] + phi[2] * phi[2]);
  const double sc = 1.0 / sqrt(1.0 + nrm * nrm);
  q[0] = sc; q[1] = sc * phi[0]; q[2] = sc * phi[1]; q[3] = sc * phi[2];
}
/* src/utils/QuaternionUtils.cpp:16-18 */
void qo_inv_cayley_map(const double q[4], double phi[3]) {
  phi[0] = q[1] / q[0]; phi[1] = q[2] / q[0]; phi[2] = q[3] / q[0];
}
/* src/utils/QuaternionUtils.cpp:20-22 */
void qo_quat_mult(const double a[4], const double b[4], double out[4]) {
  double L[16];
  qo_quat_L(a, L);
  double r[4];
  qo_mv(4, 4, L, 4, b, r);
  memcpy(out, r, sizeof r);
}
/* src/utils/QuaternionUtils.cpp:24-28 */
void qo_quat_conj(const double q[4], double out[4]) {
  out[0] = q[0]; out[1] = -q[1]; out[2] = -q[2]; out[3] = -q[3];
}

static void inv3(const double A[9], double Ai[9]) {
  /* Eigen's fixed-size 3x3 inverse() is the cofactor formula. */
  const double c00 = A[4] * A[8] - A[5] * A[7];
  const double c01 = A[5] * A[6] - A[3] * A[8];
  const double c02 = A[3] * A[7] - A[4] * A[6];
  const double det = A[0] * c00 + A[1] * c01 + A[2] * c02;
  const double id = 1.0 / det;
  Ai[0] = c00 * id;
  Ai[1] = (A[2] * A[7] - A[1] * A[8]) * id;
  Ai[2] = (A[1] * A[5] - A[2] * A[4]) * id;
  Ai[3] = c01 * id;
  Ai[4] = (A[0] * A[8] - A[2] * A[6]) * id;
  Ai[5] = (A[2] * A[3] - A[0] * A[5]) * id;
  Ai[6] = c02 * id;
  Ai[7] = (A[1] * A[6] - A[0] * A[7]) * id;
  Ai[8] = (A[0] * A[4] - A[1] * A[3]) * id;
}

void qo_srbd_prepare(qo_srbd_model* m) {
  inv3(m->inertia, m->inertia_inv);
  /* AltroUtils.cpp:368-374: g_body = R' (0,0,-9.81); moment = c x (5.204 g_body) */
  const double gw[3] = {0.0, 0.0, -9.81};
  qo_mtv(3, 3, m->rot, 3, gw, m->g_body);
  const double com[3] = {0.0223, 0.002, -0.0005};
  const double fg[3] = {5.204 * m->g_body[0], 5.204 * m->g_body[1], 5.204 * m->g_body[2]};
  m->moment_gravity[0] = com[1] * fg[2] - com[2] * fg[1];
  m->moment_gravity[1] = com[2] * fg[0] - com[0] * fg[2];
  m->moment_gravity[2] = com[0] * fg[1] - com[1] * fg[0];
}

/* src/utils/AltroUtils.cpp:363-392.  Swing-leg forces are multiplied by the
 * contact flag (the reference forces them to ~0 through the cone rows; SURVEY
 * A.6) so that the pinned-to-zero convention holds for any u handed in. */
void qo_ct_srb_quat_dynamics(const qo_srbd_model* m, double* x_dot, const double* x,
                             const double* u) {
  double fsum[3] = {0, 0, 0};
  double mom[3] = {m->moment_gravity[0], m->moment_gravity[1], m->moment_gravity[2]};
  const int nl = m->nleg ? m->nleg : 4;
  for (int i = 0; i < nl; ++i) {
    const double* r = &m->foot_pos_body[3 * i];
    const double c = m->contacts[i];
    const double f[3] = {c * u[3 * i], c * u[3 * i + 1], c * u[3 * i + 2]};
    fsum[0] += f[0]; fsum[1] += f[1]; fsum[2] += f[2];
    mom[0] += r[1] * f[2] - r[2] * f[1];
    mom[1] += r[2] * f[0] - r[0] * f[2];
    mom[2] += r[0] * f[1] - r[1] * f[0];
  }
  /* p_dot = v */
  x_dot[0] = x[7]; x_dot[1] = x[8]; x_dot[2] = x[9];
  /* q_dot = 0.5 G(q) w */
  double G[12];
  qo_quat_G(&x[3], G);
  for (int r = 0; r < 4; ++r)
    x_dot[3 + r] = 0.5 * (G[3 * r] * x[10] + G[3 * r + 1] * x[11] + G[3 * r + 2] * x[12]);
  /* v_dot = sum f / m + g_body */
  for (int a = 0; a < 3; ++a) x_dot[7 + a] = fsum[a] / m->mass + m->g_body[a];
  /* w_dot = I^-1 moment (gyroscopic term commented out upstream, :390) */
  qo_mv(3, 3, m->inertia_inv, 3, mom, &x_dot[10]);
}

/* src/utils/AltroUtils.cpp:395-439; jac 13x25 col-major: J(r,c) = jac[r + 13 c] */
void qo_ct_srb_quat_jacobian(const qo_srbd_model* m, double* jac, const double* x,
                             const double* u) {
  (void)u;
  const int nl = m->nleg ? m->nleg : 4;
  memset(jac, 0, sizeof(double) * 13 * (13 + 3 * nl));
#define J(r, c) jac[(r) + 13 * (c)]
  /* dp_dot/dv */
  J(0, 7) = 1.0; J(1, 8) = 1.0; J(2, 9) = 1.0;
  const double wx = x[10], wy = x[11], wz = x[12];
  /* dq_dot/dq: 0.5 [[0,-w'],[w,-skew(w)]] */
  J(3, 4) = -0.5 * wx; J(3, 5) = -0.5 * wy; J(3, 6) = -0.5 * wz;
  J(4, 3) = 0.5 * wx;  J(5, 3) = 0.5 * wy;  J(6, 3) = 0.5 * wz;
  J(4, 5) = 0.5 * wz;  J(4, 6) = -0.5 * wy;
  J(5, 4) = -0.5 * wz; J(5, 6) = 0.5 * wx;
  J(6, 4) = 0.5 * wy;  J(6, 5) = -0.5 * wx;
  /* dq_dot/dw = 0.5 G(q) */
  J(3, 10) = -0.5 * x[4]; J(3, 11) = -0.5 * x[5]; J(3, 12) = -0.5 * x[6];
  J(4, 10) = 0.5 * x[3];  J(4, 11) = -0.5 * x[6]; J(4, 12) = 0.5 * x[5];
  J(5, 10) = 0.5 * x[6];  J(5, 11) = 0.5 * x[3];  J(5, 12) = -0.5 * x[4];
  J(6, 10) = -0.5 * x[5]; J(6, 11) = 0.5 * x[4];  J(6, 12) = 0.5 * x[3];
  /* d/du: dv_dot/df_i = I/m ; dw_dot/df_i = I^-1 skew(r_i) */
  for (int i = 0; i < nl; ++i) {
    const double c = m->contacts[i];
    if (c == 0.0) continue; /* swing leg pinned: zero column block */
    double S[9], IS[9];
    qo_skew(&m->foot_pos_body[3 * i], S);
    qo_mm(3, 3, 3, m->inertia_inv, 3, S, 3, IS, 3);
    for (int a = 0; a < 3; ++a) {
      J(7 + a, 13 + 3 * i + a) = (1.0 / m->mass);
      for (int b = 0; b < 3; ++b) J(10 + a, 13 + 3 * i + b) = IS[3 * a + b];
    }
  }
#undef J
}

/* src/utils/AltroUtils.cpp:9-22 (float h; h/2 is a float division upstream, exact) */
void qo_midpoint_dynamics(int n, int m, qo_ct_dyn_fn f, void* ctx, double* xn, const double* x,
                          const double* u, float h) {
  (void)m;
  double xm[32];
  f(ctx, xm, x, u);
  const double hh = (double)(h / 2);
  for (int i = 0; i < n; ++i) xm[i] = xm[i] * hh + x[i];
  f(ctx, xn, xm, u);
  const double hd = (double)h;
  for (int i = 0; i < n; ++i) xn[i] = x[i] + hd * xn[i];
}

/* src/utils/AltroUtils.cpp:78-110:
 *   J_x = I + h Am (I + h/2 A),  J_u = h (Am h/2 B + Bm)                     */
void qo_midpoint_jacobian(int n, int m, qo_ct_dyn_fn f, qo_ct_jac_fn df, void* ctx, double* jac,
                          const double* x, const double* u, float h) {
  double xm[32];
  double J0[32 * 64], Jm[32 * 64]; /* n x (n+m) col-major */
  const double hd = (double)h, hh = (double)(h / 2);
  f(ctx, xm, x, u);
  for (int i = 0; i < n; ++i) xm[i] = x[i] + hh * xm[i];
  df(ctx, J0, x, u);
  df(ctx, Jm, xm, u);
  /* A-part: I + h * Am * (I + hh*A) */
  for (int c = 0; c < n; ++c)
    for (int r = 0; r < n; ++r) {
      double s = 0.0;
      for (int t = 0; t < n; ++t) {
        const double inner = (t == c ? 1.0 : 0.0) + hh * J0[t + n * c];
        s += (hd * Jm[r + n * t]) * inner;
      }
      jac[r + n * c] = (r == c ? 1.0 : 0.0) + s;
    }
  /* B-part: h * (Am * hh * B + Bm) */
  for (int c = 0; c < m; ++c)
    for (int r = 0; r < n; ++r) {
      double s = 0.0;
      for (int t = 0; t < n; ++t) s += (Jm[r + n * t] * hh) * J0[t + n * (n + c)];
      jac[r + n * (n + c)] = hd * (s + Jm[r + n * (n + c)]);
    }
}

static void srbd_f(void* ctx, double* xd, const double* x, const double* u) {
  qo_ct_srb_quat_dynamics((const qo_srbd_model*)ctx, xd, x, u);
}
static void srbd_df(void* ctx, double* J, const double* x, const double* u) {
  qo_ct_srb_quat_jacobian((const qo_srbd_model*)ctx, J, x, u);
}
void qo_srbd_discrete_dynamics(const qo_srbd_model* m, double* xn, const double* x,
                               const double* u, float h) {
  qo_midpoint_dynamics(13, 3 * (m->nleg ? m->nleg : 4), srbd_f, (void*)m, xn, x, u, h);
}
void qo_srbd_discrete_jacobian(const qo_srbd_model* m, double* jac, const double* x,
                               const double* u, float h) {
  qo_midpoint_jacobian(13, 3 * (m->nleg ? m->nleg : 4), srbd_f, srbd_df, (void*)m, jac, x, u, h);
}

/* E(x) = blkdiag(I3, G(q), I3, I3): 13x12 row-major (AltroUtils.cpp:153-157) */
void qo_srbd_error_jacobian(const double* x, double* E) {
  memset(E, 0, sizeof(double) * 13 * 12);
  for (int a = 0; a < 3; ++a) {
    E[a * 12 + a] = 1.0;
    E[(7 + a) * 12 + 6 + a] = 1.0;
    E[(10 + a) * 12 + 9 + a] = 1.0;
  }
  double G[12];
  qo_quat_G(&x[3], G);
  for (int r = 0; r < 4; ++r)
    for (int c = 0; c < 3; ++c) E[(3 + r) * 12 + 3 + c] = G[3 * r + c];
}

/* AltroUtils.cpp:167-168 */
void qo_srbd_project(const double* jac, const double* x, const double* xn, double* Abar,
                     double* Bbar) {
  double E[13 * 12], En[13 * 12];
  qo_srbd_error_jacobian(x, E);
  qo_srbd_error_jacobian(xn, En);
  double A[13 * 13], B[13 * 12];
  for (int r = 0; r < 13; ++r) {
    for (int c = 0; c < 13; ++c) A[r * 13 + c] = jac[r + 13 * c];
    for (int c = 0; c < 12; ++c) B[r * 12 + c] = jac[r + 13 * (13 + c)];
  }
  double AE[13 * 12];
  qo_mm(13, 13, 12, A, 13, E, 12, AE, 12);
  qo_mtm(12, 13, 12, En, 12, AE, 12, Abar, 12);
  qo_mtm(12, 13, 12, En, 12, B, 12, Bbar, 12);
}

/* QuatMpc.cpp:47-52 (C_mat) times torso_rot_mat (:203,213): 6x3 row-major */
void qo_cone_block(double mu, const double rot[9], double CR[18]) {
  const double C[18] = {1, 0, -mu, -1, 0, -mu, 0, 1, -mu, 0, -1, -mu, 0, 0, 1, 0, 0, -1};
  qo_mm(6, 3, 3, C, 3, rot, 3, CR, 3);
}

/* QuatMpc.cpp:194-205: c_i = C_mat R u_i + (0,0,0,0,-fz_max*contact_i,0) */
void qo_cone_eval_n(int n, double mu, double fz_max, const double rot[9], const double* contacts,
                    const double* u, double* c) {
  double CR[18];
  qo_cone_block(mu, rot, CR);
  for (int i = 0; i < n; ++i) {
    qo_mv(6, 3, CR, 3, &u[3 * i], &c[6 * i]);
    c[6 * i + 4] += -fz_max * contacts[i];
  }
}
void qo_cone_eval(double mu, double fz_max, const double rot[9], const double contacts[4],
                  const double* u, double* c) {
  qo_cone_eval_n(4, mu, fz_max, rot, contacts, u, c);
}
