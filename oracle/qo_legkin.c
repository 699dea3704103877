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

/* ---- inverse kinematics --------------------------------------------------------------------------------------
 * A1Kinematics.cpp:291-312: atan on [-1, 1] as an odd degree-11 polynomial in SINGLE precision (fused Horner
 * steps), extended to atan2 by the reciprocal identity and the half-plane shift.  The arguments arrive as doubles
 * and are narrowed at the call, the result is a float. */
static float atan_unit(float x) {
  static const float c[6] = {0.99997726f, -0.33262347f, 0.19354346f, -0.11643287f, 0.05265332f, -0.01172120f};
  const float x2 = x * x;
  float acc = c[5];
  for (int i = 4; i >= 0; --i) acc = fmaf(x2, acc, c[i]);
  return x * acc;
}
static float atan2_sp(double yd, double xd) {
  const float y = (float)yd, x = (float)xd;
  const float pi = (float)M_PI, half_pi = (float)M_PI_2;
  const int steep = fabsf(x) < fabsf(y);
  const float ratio = steep ? x / y : y / x;
  float r = atan_unit(ratio);
  if (steep) r = (ratio >= 0.0f ? half_pi : -half_pi) - r;
  if (x < 0.0f) r = (y >= 0.0f ? pi : -pi) + r;
  return r;
}

/* A1Kinematics.cpp:335-459.  The hip angle t1 comes from the y-z plane: the foot sits at distance d (hip link) and
 * L (leg extension) from the hip axis, so atan2(Zf, Yf') and atan2(L, d) add or subtract depending on the side of
 * the leg (sign of oy) and on the quadrant; the mirrored solution is the candidate, and the one nearer to the
 * current hip angle wins.  Mixed float / double arithmetic follows the reference's expression types: a sum of two
 * approximations is formed in single precision, a sum with a multiple of M_PI in double.  (`abs` in the reference
 * is the double overload: Eigen's SSE headers pull <stdlib.h> into the translation unit.) */
void qo_leg_inv_kin(const double p[3], const double cur_q[3], const double r[5], double q[3]) {
  const double oy = r[1], d = r[2], lt = r[3], lc = r[4];
  const double xs = p[0] - r[0], ys = p[1] - oy, zf = p[2];
  double L = sqrt(zf * zf + ys * ys - d * d);
  double t1 = 0.0, alt = 0.0;
  if (oy > 0) {                                           /* left legs */
    const float aL = atan2_sp(L, d);
    if (zf > 0) {                                         /* foot above the hip axis */
      if (ys > 0) t1 = (double)(atan2_sp(zf, ys) - aL);
      else if (ys == 0) t1 = M_PI / 2 - (double)aL;
      else t1 = (M_PI - (double)atan2_sp(zf, -ys)) - (double)aL;
      alt = (double)(atan2_sp(zf, ys) + aL);
    } else if (zf < 0) {
      if (ys > 0) t1 = (double)(atan2_sp(zf, ys) + aL);
      else if (ys == 0) t1 = -M_PI / 2 + (double)aL;
      else t1 = (-M_PI - (double)atan2_sp(zf, -ys)) + (double)aL;
      alt = (double)(atan2_sp(zf, ys) - aL);
    } else {
      t1 = (double)aL;
      alt = (double)(-aL);
    }
  } else {                                                /* right legs: d < 0 */
    if (zf > 0) {
      const float aL = atan2_sp(L, -d);
      if (ys < 0) t1 = (double)(-atan2_sp(zf, -ys) + aL);
      else if (ys == 0) t1 = -M_PI / 2 + (double)aL;
      else t1 = (-M_PI + (double)atan2_sp(zf, ys)) + (double)aL;
      alt = (double)(-aL - atan2_sp(zf, -ys));
    } else if (zf < 0) {
      const float aL = atan2_sp(L, -d);
      if (ys < 0) t1 = (double)(atan2_sp(-zf, -ys) - aL);
      else if (ys == 0) t1 = -M_PI / 2 - (double)aL;
      else t1 = (M_PI - (double)atan2_sp(-zf, ys)) - (double)aL;
      alt = (double)(atan2_sp(-zf, -ys) + aL);
    }                                                     /* zf == 0: both stay 0 in the reference */
  }
  if (!(fabs(t1 - cur_q[0]) < fabs(alt - cur_q[0]))) t1 = alt;

  /* knee from the law of cosines in the leg plane, clamped near the straight / folded leg (:421-430) */
  const double cb = (lt * lt + lc * lc - xs * xs - L * L) / (2 * lt * lc);
  double beta;
  if (fabs(cb + 1) < 0.001) beta = M_PI;
  else if (fabs(cb - 1) < 0.001) beta = 0;
  else beta = acos(cb);
  const double t3 = beta - M_PI;

  /* thigh: direction to the foot in the leg plane + the interior angle at the hip (:434-450) */
  if (zf > d * sin(t1)) L = -L;
  const double gamma = (double)atan2_sp(-xs, L);
  const double alpha = (double)atan2_sp(lc * sin(-t3), lt + lc * cos(-t3));
  double t2 = gamma + alpha;
  if (t2 < -60 * M_PI / 180) t2 += 2 * M_PI;
  else if (t2 > 240 * M_PI / 180) t2 -= 2 * M_PI;
  q[0] = t1; q[1] = t2; q[2] = t3;
}

void qo_leg_inverse_kinematics(const qmpc_leg_geometry* g, int32_t batch, const double* foot_pos_body,
                               const double* cur_joint_pos, double* joint_pos) {
  for (size_t t = 0; t < 4 * (size_t)batch; ++t)
    qo_leg_inv_kin(&foot_pos_body[3 * t], &cur_joint_pos[3 * t], g->rho_fix[t & 3], &joint_pos[3 * t]);
}

/* x = A^-1 b for a COLUMN-major 3x3 by elimination with row pivoting (Eigen's PartialPivLU, which `jac.lu()` is:
 * the largest |entry| of the column becomes the pivot) */
static void lu_solve3(const double Acm[9], const double b[3], double x[3]) {
  double a[3][4];
  for (int i = 0; i < 3; ++i) { for (int j = 0; j < 3; ++j) a[i][j] = Acm[3 * j + i]; a[i][3] = b[i]; }
  for (int c = 0; c < 3; ++c) {
    int piv = c;
    for (int i = c + 1; i < 3; ++i) if (fabs(a[i][c]) > fabs(a[piv][c])) piv = i;
    if (piv != c) for (int j = 0; j < 4; ++j) { const double t = a[c][j]; a[c][j] = a[piv][j]; a[piv][j] = t; }
    for (int i = c + 1; i < 3; ++i) {
      const double f = a[i][c] / a[c][c];
      for (int j = c; j < 4; ++j) a[i][j] -= f * a[c][j];
    }
  }
  for (int i = 2; i >= 0; --i) {
    double s = a[i][3];
    for (int j = i + 1; j < 3; ++j) s -= a[i][j] * x[j];
    x[i] = s / a[i][i];
  }
}

static void quat_rot(const double q[4], double R[9] /* row-major */) {
  const double w = q[0], x = q[1], y = q[2], z = q[3];
  R[0] = 1 - 2 * (y * y + z * z); R[1] = 2 * (x * y - w * z);     R[2] = 2 * (x * z + w * y);
  R[3] = 2 * (x * y + w * z);     R[4] = 1 - 2 * (x * x + z * z); R[5] = 2 * (y * z - w * x);
  R[6] = 2 * (x * z - w * y);     R[7] = 2 * (y * z + w * x);     R[8] = 1 - 2 * (x * x + y * y);
}

/* BaseInterface.cpp:343-408 */
void qo_joint_commands(const qmpc_leg_geometry* g, int32_t batch, const qmpc_joint_feedback* fb,
                       qmpc_joint_command* cmd) {
  for (int b = 0; b < batch; ++b) {
    const qmpc_joint_feedback* f = &fb[b];
    qmpc_joint_command* o = &cmd[b];
    double R[9];
    quat_rot(f->torso_quat, R);
    for (int l = 0; l < 4; ++l) {
      const double* qj = &f->joint_pos[3 * l];
      double J[9];
      qo_leg_jac(qj, g->rho_opt[l], g->rho_fix[l], J);
      const double* fl = &f->forces_body[3 * l];
      double tau[3];
      for (int j = 0; j < 3; ++j) tau[j] = -(J[3 * j] * fl[0] + J[3 * j + 1] * fl[1] + J[3 * j + 2] * fl[2]);
      if (f->movement_mode > 0) {
        double dp[3], dv[3], pb[3], vb[3], qt[3], qd[3];
        for (int i = 0; i < 3; ++i) {
          dp[i] = f->foot_pos_target_world[3 * l + i] - f->torso_pos_world[i];
          dv[i] = f->foot_vel_target_world[3 * l + i] - f->torso_lin_vel_world[i];
        }
        for (int i = 0; i < 3; ++i) {                      /* R' v */
          pb[i] = R[i] * dp[0] + R[3 + i] * dp[1] + R[6 + i] * dp[2];
          vb[i] = R[i] * dv[0] + R[3 + i] * dv[1] + R[6 + i] * dv[2];
        }
        qo_leg_inv_kin(pb, qj, g->rho_fix[l], qt);
        const int bad_q = isnan(qt[0]) || isnan(qt[1]) || isnan(qt[2]);
        lu_solve3(J, vb, qd);
        const int bad_v = isnan(qd[0]) || isnan(qd[1]) || isnan(qd[2]);
        for (int j = 0; j < 3; ++j) {
          o->joint_ang_tgt[3 * l + j] = bad_q ? qj[j] : qt[j];
          o->joint_vel_tgt[3 * l + j] = bad_v ? f->joint_vel[3 * l + j] : qd[j];
          o->joint_tau_tgt[3 * l + j] = (f->plan_contacts[l] != 0.0) ? tau[j] : 0.0;
        }
      } else {
        for (int j = 0; j < 3; ++j) {
          o->joint_tau_tgt[3 * l + j] = tau[j];
          o->joint_ang_tgt[3 * l + j] = qj[j];
          o->joint_vel_tgt[3 * l + j] = f->joint_vel[3 * l + j];
        }
      }
    }
  }
}
