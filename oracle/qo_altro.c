/*
 * qo_altro.c -- CPU restatement of the reference's AL-iLQR solver scheme
 * (external dependency zixinz990/altro @ b47202ff; see qo_altro.h header for
 * provenance and for how it is pinned).  TEST INFRASTRUCTURE ONLY.
 *
 * Scheme:
 *   lambda <- 0, rho <- penalty_initial; U <- initial inputs; X <- rollout
 *   repeat:
 *     expansions at (X,U): cost gradient/Hessian in error-state coordinates,
 *       discrete dynamics Jacobians projected with E(x), AL terms of every
 *       constraint:  z = lambda + rho c,  z+ = Proj(z),  active = z > 0
 *     backward Riccati pass -> K_k, d_k
 *     forward pass with backtracking on the AL merit function
 *     stationarity / feasibility / step size at the NEW trajectory
 *     convergence test; dual + penalty update
 */
#include "qo_altro.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "qo_linalg.h"
#include "qo_srbd.h" /* qo_quat_G */

#define NE_MAX QO_MAXN
#define M_MAX QO_MAXM

typedef struct knot_ws {
  double A[NE_MAX * NE_MAX];   /* ne x ne row-major (projected)        */
  double B[NE_MAX * M_MAX];    /* ne x m                                */
  double lx[NE_MAX];           /* error-state cost gradient             */
  double lu[M_MAX];
  double lxx[NE_MAX * NE_MAX];
  double luu[M_MAX];           /* diagonal                              */
  double K[M_MAX * NE_MAX];    /* m x ne                                */
  double d[M_MAX];
  /* constraints */
  double c[QO_MAXCON][QO_MAXP];
  double Jx[QO_MAXCON][QO_MAXP * NE_MAX];  /* p x ne row-major */
  double Ju[QO_MAXCON][QO_MAXP * M_MAX];   /* p x m  row-major */
  double lam[QO_MAXCON][QO_MAXP];
  double s[QO_MAXCON][QO_MAXP];    /* interior-point slacks             */
  double rc[QO_MAXCON][QO_MAXP];   /* slack residual c(u) + s, tracked analytically */
  double kap[QO_MAXCON][QO_MAXP];  /* 1 on rows showing the weakly-active signature       */
  double ds[QO_MAXCON][QO_MAXP];
  double dlam[QO_MAXCON][QO_MAXP];
  /* second-order-cone blocks: raw Jacobians (Jx / Ju of such a block hold rows rotated into the eigenbasis of
   * the projection Jacobian, see soc_refresh), multiplier and curvature weights in that basis */
  double socJx[QO_MAXCON][QO_SOC_MAXP * NE_MAX];
  double socJu[QO_MAXCON][QO_SOC_MAXP * M_MAX];
  double soc_zp[QO_MAXCON][QO_SOC_MAXP];
  double soc_w[QO_MAXCON][QO_SOC_MAXP];
} knot_ws;

typedef struct solver_ws {
  const qo_problem* prob;
  int n, ne, m, N;
  knot_ws* kn;        /* N+1 */
  double* X;          /* (N+1) x n, current */
  double* U;          /* N x m              */
  double* Xc;         /* candidate          */
  double* Uc;
  double* dU;         /* candidate input increment alpha d + K dx, kept as computed */
  double rho;
  double dV1, dV2;
  int ipm;            /* 1: phase-1 interior-point weights, 0: AL weights */
  double ipm_target;  /* sigma * mu */
} solver_ws;

static int con_active_at(const qo_constraint* c, int k) { return k >= c->k_start && k < c->k_stop; }
static int row_on(const qo_constraint* c, int i) { return !c->row_enable || c->row_enable[i] != 0.0; }

/* E(x): n x ne row-major.  Identity except the quaternion rows. */
static void error_jacobian(const qo_problem* p, const double* x, double* E, int ne) {
  const int n = p->n;
  memset(E, 0, sizeof(double) * n * ne);
  if (!p->use_quaternion) {
    for (int i = 0; i < n; ++i) E[i * ne + i] = 1.0;
    return;
  }
  const int qi = p->quat_start_index;
  for (int i = 0; i < qi; ++i) E[i * ne + i] = 1.0;
  double G[12];
  qo_quat_G(&x[qi], G);
  for (int r = 0; r < 4; ++r)
    for (int c = 0; c < 3; ++c) E[(qi + r) * ne + qi + c] = G[3 * r + c];
  for (int i = qi + 4; i < n; ++i) E[i * ne + (i - 1)] = 1.0;
}

/* dx = x1 (-) x0 in error-state coordinates; attitude part through the inverse
 * Cayley map of q0^-1 * q1  (QuaternionUtils.cpp:16-18). */
static void state_diff(const qo_problem* p, const double* x1, const double* x0, double* dx) {
  const int n = p->n;
  if (!p->use_quaternion) {
    for (int i = 0; i < n; ++i) dx[i] = x1[i] - x0[i];
    return;
  }
  const int qi = p->quat_start_index;
  for (int i = 0; i < qi; ++i) dx[i] = x1[i] - x0[i];
  double G[12];
  qo_quat_G(&x0[qi], G);
  const double* q1 = &x1[qi];
  const double* q0 = &x0[qi];
  const double sc = q0[0] * q1[0] + q0[1] * q1[1] + q0[2] * q1[2] + q0[3] * q1[3];
  for (int c = 0; c < 3; ++c) {
    double s = 0.0;
    for (int r = 0; r < 4; ++r) s += G[3 * r + c] * q1[r];
    dx[qi + c] = s / sc;
  }
  for (int i = qi + 4; i < n; ++i) dx[i - 1] = x1[i] - x0[i];
}

/* stage / terminal cost (un-augmented) */
static double knot_cost(const qo_problem* p, int k, const double* x, const double* u) {
  double J = 0.0;
  for (int i = 0; i < p->n; ++i) {
    const double e = x[i] - p->xref[k][i];
    J += 0.5 * p->Q[k][i] * e * e;
  }
  if (p->use_quaternion) {
    const int qi = p->quat_start_index;
    const double dq = qo_dot(4, &p->xref[k][qi], &x[qi]);
    J += p->w[k] * (1.0 - fabs(dq));
  }
  if (k < p->N) {
    for (int j = 0; j < p->m; ++j) {
      const double e = u[j] - p->uref[k][j];
      J += 0.5 * p->R[k][j] * e * e;
    }
  }
  return J;
}


/* ---- second-order cone K = {(v, t): |v| <= t} (self-dual) ------------------------------------
 * Constraint c(x,u) in K.  AL term (|Proj_K(lam - rho c)|^2 - |lam|^2) / (2 rho), gradient -J' Proj_K(lam - rho c),
 * Gauss-Newton curvature rho J' dProj J, dual update lam <- Proj_K(lam - rho c)  (conic augmented Lagrangian of
 * Jackson, Howell et al., "ALTRO-C").  dProj is symmetric with eigenvalues 1 on (w,1)/sqrt2, (1 + t/a)/2 on the
 * directions (e,0) with e orthogonal to w = v/|v| (a = |v|), and 0 on (w,-1)/sqrt2 when the point is outside both the
 * cone and its polar; the identity inside the cone, zero inside the polar.  soc_refresh rotates the block's Jacobian
 * rows into that eigenbasis, so the per-row machinery (multiplier zp, weight act) applies unchanged. */
static void soc_project(int p, const double* z, double* out) {
  const int nv = p - 1;
  double a = 0.0;
  for (int i = 0; i < nv; ++i) a += z[i] * z[i];
  a = sqrt(a);
  const double t = z[nv];
  if (a <= t) { memcpy(out, z, sizeof(double) * p); return; }
  if (a <= -t) { memset(out, 0, sizeof(double) * p); return; }
  const double sc = 0.5 * (a + t);
  for (int i = 0; i < nv; ++i) out[i] = sc * z[i] / a;
  out[nv] = sc;
}

static void soc_refresh(solver_ws* ws) {
  const qo_problem* pr = ws->prob;
  const int ne = ws->ne, m = ws->m;
  for (int ci = 0; ci < pr->ncon; ++ci) {
    const qo_constraint* cn = &pr->con[ci];
    if (cn->type != QO_SOC) continue;
    const int p = cn->p, nv = p - 1;
    for (int k = 0; k <= ws->N; ++k) {
      if (!con_active_at(cn, k)) continue;
      knot_ws* kw = &ws->kn[k];
      double z[QO_SOC_MAXP], pz[QO_SOC_MAXP], Q[QO_SOC_MAXP * QO_SOC_MAXP], ev[QO_SOC_MAXP];
      for (int i = 0; i < p; ++i) z[i] = kw->lam[ci][i] - ws->rho * kw->c[ci][i];
      soc_project(p, z, pz);
      double a = 0.0;
      for (int i = 0; i < nv; ++i) a += z[i] * z[i];
      a = sqrt(a);
      const double t = z[nv];
      /* eigenbasis (columns of Q) and eigenvalues of dProj at z */
      memset(Q, 0, sizeof Q);
      if (a <= t || a <= -t || a == 0.0) {
        for (int i = 0; i < p; ++i) { Q[i * p + i] = 1.0; ev[i] = (a <= t) ? 1.0 : 0.0; }
      } else {
        double w[QO_SOC_MAXP];
        for (int i = 0; i < nv; ++i) w[i] = z[i] / a;
        const double r2 = sqrt(0.5);
        for (int i = 0; i < nv; ++i) { Q[i * p + 0] = r2 * w[i]; Q[i * p + (p - 1)] = r2 * w[i]; }
        Q[nv * p + 0] = r2;  Q[nv * p + (p - 1)] = -r2;
        ev[0] = 1.0; ev[p - 1] = 0.0;
        /* Householder reflection mapping e_1 to w: its other columns span the complement of w */
        double hv[QO_SOC_MAXP], hn = 0.0;
        for (int i = 0; i < nv; ++i) { hv[i] = ((i == 0) ? 1.0 : 0.0) - w[i]; hn += hv[i] * hv[i]; }
        for (int c = 1; c < nv; ++c) {
          for (int i = 0; i < nv; ++i) {
            const double hic = ((i == c) ? 1.0 : 0.0) - ((hn > 0.0) ? 2.0 * hv[i] * hv[c] / hn : 0.0);
            Q[i * p + c] = hic;
          }
          ev[c] = 0.5 * (1.0 + t / a);
        }
      }
      /* rows in the eigenbasis: J_eff = Q' J ; multiplier zp = Q' (-Proj(z)) ; weight rho * eigenvalue */
      for (int r = 0; r < p; ++r) {
        for (int c = 0; c < ne; ++c) {
          double sx = 0.0;
          for (int i = 0; i < p; ++i) sx += Q[i * p + r] * kw->socJx[ci][i * ne + c];
          kw->Jx[ci][r * ne + c] = sx;
        }
        for (int c = 0; c < m; ++c) {
          double su = 0.0;
          for (int i = 0; i < p; ++i) su += Q[i * p + r] * kw->socJu[ci][i * m + c];
          kw->Ju[ci][r * m + c] = su;
        }
        double sz = 0.0;
        for (int i = 0; i < p; ++i) sz += Q[i * p + r] * pz[i];
        kw->soc_zp[ci][r] = -sz;
        kw->soc_w[ci][r] = ws->rho * ev[r];
      }
    }
  }
}

/* AL merit term of one constraint block:  (|Proj(lam + rho c)|^2 - |lam|^2) / (2 rho) */
static double al_term(const qo_constraint* cn, int type, int pr, const double* c,
                      const double* lam, double rho) {
  double s = 0.0;
  if (type == QO_SOC) {
    double z[QO_SOC_MAXP], pz[QO_SOC_MAXP];
    for (int i = 0; i < pr; ++i) z[i] = lam[i] - rho * c[i];
    soc_project(pr, z, pz);
    for (int i = 0; i < pr; ++i) s += pz[i] * pz[i] - lam[i] * lam[i];
    return s / (2.0 * rho);
  }
  for (int i = 0; i < pr; ++i) {
    if (!row_on(cn, i)) continue;
    double z = lam[i] + rho * c[i];
    if (type == QO_INEQUALITY && z < 0.0) z = 0.0;
    s += z * z - lam[i] * lam[i];
  }
  return s / (2.0 * rho);
}

static double total_cost(const solver_ws* ws, const double* X, const double* U, double* plain,
                         double* viol) {
  const qo_problem* p = ws->prob;
  double J = 0.0, Jal = 0.0, v = 0.0;
  double c[QO_MAXP];
  for (int k = 0; k <= ws->N; ++k) {
    const double* x = &X[k * ws->n];
    const double* u = (k < ws->N) ? &U[k * ws->m] : NULL;
    J += knot_cost(p, k, x, u);
    for (int ci = 0; ci < p->ncon; ++ci) {
      const qo_constraint* cn = &p->con[ci];
      if (!con_active_at(cn, k)) continue;
      cn->con(cn->ctx, k, c, x, u);
      Jal += al_term(cn, cn->type, cn->p, c, ws->kn[k].lam[ci], ws->rho);
      if (cn->type == QO_SOC) {   /* distance of c from the cone */
        double pc[QO_SOC_MAXP];
        soc_project(cn->p, c, pc);
        for (int i = 0; i < cn->p; ++i) v = fmax(v, fabs(c[i] - pc[i]));
        continue;
      }
      for (int i = 0; i < cn->p; ++i) {
        if (!row_on(cn, i)) continue;
        const double vi = (cn->type == QO_INEQUALITY) ? fmax(c[i], 0.0) : fabs(c[i]);
        if (vi > v) v = vi;
      }
    }
  }
  if (plain) *plain = J;
  if (viol) *viol = v;
  return J + Jal;
}

/* expansions at (X,U): dynamics Jacobians, cost, constraints */
static void expansions(solver_ws* ws) {
  const qo_problem* p = ws->prob;
  const int n = ws->n, ne = ws->ne, m = ws->m, N = ws->N;
  double E[QO_MAXN * NE_MAX], En[QO_MAXN * NE_MAX];
  double jac[QO_MAXN * (QO_MAXN + QO_MAXM)];
  double tmp[QO_MAXN * NE_MAX];
  double cj[QO_MAXP * (NE_MAX + M_MAX)];
  for (int k = 0; k <= N; ++k) {
    knot_ws* kw = &ws->kn[k];
    const double* x = &ws->X[k * n];
    const double* u = (k < N) ? &ws->U[k * m] : NULL;
    error_jacobian(p, x, E, ne);
    /* --- cost --- */
    double lxf[QO_MAXN];
    for (int i = 0; i < n; ++i) lxf[i] = p->Q[k][i] * (x[i] - p->xref[k][i]);
    double quat_hess = 0.0;
    if (p->use_quaternion) {
      const int qi = p->quat_start_index;
      const double dq = qo_dot(4, &p->xref[k][qi], &x[qi]);
      const double s = (dq >= 0.0) ? 1.0 : -1.0;
      for (int r = 0; r < 4; ++r) lxf[qi + r] += -s * p->w[k] * p->xref[k][qi + r];
      quat_hess = -qo_dot(4, &x[qi], &lxf[qi]);
    }
    qo_mtv(n, ne, E, ne, lxf, kw->lx);
    /* lxx = E' diag(Q) E (+ quat_hess I3 on the attitude block) */
    for (int i = 0; i < n; ++i)
      for (int j = 0; j < ne; ++j) tmp[i * ne + j] = p->Q[k][i] * E[i * ne + j];
    qo_mtm(ne, n, ne, E, ne, tmp, ne, kw->lxx, ne);
    if (p->use_quaternion) {
      const int qi = p->quat_start_index;
      for (int a = 0; a < 3; ++a) kw->lxx[(qi + a) * ne + qi + a] += quat_hess;
    }
    if (k < N) {
      for (int j = 0; j < m; ++j) {
        kw->lu[j] = p->R[k][j] * (u[j] - p->uref[k][j]);
        kw->luu[j] = p->R[k][j];
      }
      /* --- dynamics --- */
      const double* xn = &ws->X[(k + 1) * n];
      error_jacobian(p, xn, En, ne);
      p->jac(p->dyn_ctx, k, jac, x, u, p->h);
      /* A = En' Jx E ; B = En' Ju  (AltroUtils.cpp:167-168) */
      double JxE[QO_MAXN * NE_MAX];
      for (int r = 0; r < n; ++r)
        for (int c = 0; c < ne; ++c) {
          double s = 0.0;
          for (int t = 0; t < n; ++t) s += jac[r + n * t] * E[t * ne + c];
          JxE[r * ne + c] = s;
        }
      qo_mtm(ne, n, ne, En, ne, JxE, ne, kw->A, ne);
      for (int r = 0; r < ne; ++r)
        for (int c = 0; c < m; ++c) {
          double s = 0.0;
          for (int t = 0; t < n; ++t) s += En[t * ne + r] * jac[t + n * (n + c)];
          kw->B[r * m + c] = s;
        }
    }
    /* --- constraints --- */
    for (int ci = 0; ci < p->ncon; ++ci) {
      const qo_constraint* cn = &p->con[ci];
      if (!con_active_at(cn, k)) continue;
      cn->con(cn->ctx, k, kw->c[ci], x, u);
      memset(cj, 0, sizeof(double) * cn->p * (ne + m));
      cn->jac(cn->ctx, k, cj, x, u);
      for (int r = 0; r < cn->p; ++r) {
        for (int c = 0; c < ne; ++c) kw->Jx[ci][r * ne + c] = cj[r + cn->p * c];
        for (int c = 0; c < m; ++c) kw->Ju[ci][r * m + c] = cj[r + cn->p * (ne + c)];
      }
      if (cn->type == QO_SOC) {
        memcpy(kw->socJx[ci], kw->Jx[ci], sizeof(double) * cn->p * ne);
        memcpy(kw->socJu[ci], kw->Ju[ci], sizeof(double) * cn->p * m);
      }
    }
  }
}

/* Per-row gradient weight zp (multiplies J' into Qx/Qu) and Hessian weight act
 * (multiplies J'J) of constraint ci at knot k.
 *   AL  : zp = Proj(lambda + rho c),            act = rho * [z > 0]
 *   IPM : zp = sigma mu / s + (lambda/s)(c + s), act = lambda / s   (inequalities) */
static void al_multiplier(const solver_ws* ws, int k, int ci, double* zp, double* act) {
  const qo_constraint* cn = &ws->prob->con[ci];
  const knot_ws* kw = &ws->kn[k];
  if (cn->type == QO_SOC) {       /* rotated rows, see soc_refresh */
    for (int i = 0; i < cn->p; ++i) { zp[i] = kw->soc_zp[ci][i]; act[i] = kw->soc_w[ci][i]; }
    return;
  }
  for (int i = 0; i < cn->p; ++i) {
    if (!row_on(cn, i)) { zp[i] = 0.0; act[i] = 0.0; continue; }
    if (ws->ipm && cn->type == QO_INEQUALITY) {
      const double sig = kw->lam[ci][i] / kw->s[ci][i];
      zp[i] = ws->ipm_target / kw->s[ci][i] - kw->kap[ci][i] * kw->lam[ci][i] + sig * kw->rc[ci][i];
      act[i] = sig;
      continue;
    }
    double z = kw->lam[ci][i] + ws->rho * kw->c[ci][i];
    double a = ws->rho;
    if (cn->type == QO_INEQUALITY && !(z > 0.0)) { z = 0.0; a = 0.0; }
    zp[i] = z;
    act[i] = a;
  }
}


/* Orthonormal T (m x m, columns t_j): modified Gram-Schmidt over the rows of Ju
 * with non-zero Hessian weight, heaviest first, completed with the unit vectors. */
static void build_rotation(const solver_ws* ws, int k, double hw[QO_MAXCON][QO_MAXP], double* T) {
  const qo_problem* p = ws->prob;
  const int m = ws->m;
  const knot_ws* kw = &ws->kn[k];
  int nrow = 0;
  int rci[QO_MAXCON * QO_MAXP], rri[QO_MAXCON * QO_MAXP];
  for (int ci = 0; ci < p->ncon; ++ci) {
    if (!con_active_at(&p->con[ci], k)) continue;
    for (int r = 0; r < p->con[ci].p; ++r)
      if (hw[ci][r] > 0.0) { rci[nrow] = ci; rri[nrow] = r; nrow++; }
  }
  /* insertion sort by decreasing weight (stable) */
  for (int a = 1; a < nrow; ++a) {
    const int c0 = rci[a], r0 = rri[a];
    const double w0 = hw[c0][r0];
    int b = a - 1;
    while (b >= 0 && hw[rci[b]][rri[b]] < w0) { rci[b + 1] = rci[b]; rri[b + 1] = rri[b]; --b; }
    rci[b + 1] = c0; rri[b + 1] = r0;
  }
  int ncol = 0;
  double v[M_MAX];
  for (int cand = 0; cand < nrow + m && ncol < m; ++cand) {
    double nrm0 = 0.0;
    if (cand < nrow) {
      const double* row = &kw->Ju[rci[cand]][rri[cand] * m];
      for (int j = 0; j < m; ++j) { v[j] = row[j]; nrm0 += row[j] * row[j]; }
    } else {
      for (int j = 0; j < m; ++j) v[j] = 0.0;
      v[cand - nrow] = 1.0;
      nrm0 = 1.0;
    }
    if (nrm0 == 0.0) continue;
    for (int pass = 0; pass < 2; ++pass)
      for (int c = 0; c < ncol; ++c) {
        double dp = 0.0;
        for (int j = 0; j < m; ++j) dp += T[j * m + c] * v[j];
        for (int j = 0; j < m; ++j) v[j] -= dp * T[j * m + c];
      }
    double nrm = 0.0;
    for (int j = 0; j < m; ++j) nrm += v[j] * v[j];
    if (nrm <= 1e-12 * nrm0) continue; /* (anti)parallel to the span so far */
    nrm = sqrt(nrm);
    for (int j = 0; j < m; ++j) T[j * m + ncol] = v[j] / nrm;
    ncol++;
  }
}

static int backward_pass(solver_ws* ws) {
  const qo_problem* p = ws->prob;
  soc_refresh(ws);
  const int ne = ws->ne, m = ws->m, N = ws->N;
  double P[NE_MAX * NE_MAX], pv[NE_MAX];
  double zp[QO_MAXP], act[QO_MAXP];
  /* terminal */
  {
    knot_ws* kw = &ws->kn[N];
    memcpy(P, kw->lxx, sizeof(double) * ne * ne);
    memcpy(pv, kw->lx, sizeof(double) * ne);
    for (int ci = 0; ci < p->ncon; ++ci) {
      const qo_constraint* cn = &p->con[ci];
      if (!con_active_at(cn, N)) continue;
      al_multiplier(ws, N, ci, zp, act);
      for (int r = 0; r < cn->p; ++r) {
        const double* jr = &kw->Jx[ci][r * ne];
        for (int a = 0; a < ne; ++a) {
          pv[a] += jr[a] * zp[r];
          if (act[r] != 0.0)
            for (int b = 0; b < ne; ++b) P[a * ne + b] += act[r] * jr[a] * jr[b];
        }
      }
    }
  }
  ws->dV1 = 0.0;
  ws->dV2 = 0.0;
  double PA[NE_MAX * NE_MAX], PB[NE_MAX * M_MAX];
  double Qxx[NE_MAX * NE_MAX], Qux[M_MAX * NE_MAX], Quu[M_MAX * M_MAX], Qx[NE_MAX], Qu[M_MAX];
  double L[M_MAX * M_MAX], QuuK[M_MAX * NE_MAX], tmpv[M_MAX];
  double T[M_MAX * M_MAX], QuuR[M_MAX * M_MAX], QuxR[M_MAX * NE_MAX], QuR[M_MAX];
  double Kt[M_MAX * NE_MAX], dt[M_MAX];
  double hw[QO_MAXCON][QO_MAXP];
  for (int k = N - 1; k >= 0; --k) {
    knot_ws* kw = &ws->kn[k];
    memset(hw, 0, sizeof hw);
    qo_mm(ne, ne, ne, P, ne, kw->A, ne, PA, ne);
    qo_mm(ne, ne, m, P, ne, kw->B, m, PB, m);
    qo_mtm(ne, ne, ne, kw->A, ne, PA, ne, Qxx, ne);
    qo_mtm(m, ne, ne, kw->B, m, PA, ne, Qux, ne);
    qo_mtm(m, ne, m, kw->B, m, PB, m, Quu, m);
    qo_mtv(ne, ne, kw->A, ne, pv, Qx);
    qo_mtv(ne, m, kw->B, m, pv, Qu);
    for (int a = 0; a < ne; ++a) {
      Qx[a] += kw->lx[a];
      for (int b = 0; b < ne; ++b) Qxx[a * ne + b] += kw->lxx[a * ne + b];
    }
    for (int j = 0; j < m; ++j) {
      Qu[j] += kw->lu[j];
      Quu[j * m + j] += kw->luu[j];
    }
    for (int ci = 0; ci < p->ncon; ++ci) {
      const qo_constraint* cn = &p->con[ci];
      if (!con_active_at(cn, k)) continue;
      al_multiplier(ws, k, ci, zp, act);
      for (int r = 0; r < cn->p; ++r) {
        const double* jx = &kw->Jx[ci][r * ne];
        const double* ju = &kw->Ju[ci][r * m];
        for (int a = 0; a < ne; ++a) Qx[a] += jx[a] * zp[r];
        for (int j = 0; j < m; ++j) Qu[j] += ju[j] * zp[r];
        hw[ci][r] = act[r];
        if (act[r] != 0.0) {
          const double rho = act[r];
          for (int a = 0; a < ne; ++a)
            for (int b = 0; b < ne; ++b) Qxx[a * ne + b] += rho * jx[a] * jx[b];
          for (int j = 0; j < m; ++j)
            for (int b = 0; b < ne; ++b) Qux[j * ne + b] += rho * ju[j] * jx[b];
          /* the Ju'Ju term is assembled in rotated coordinates below */
        }
      }
    }
    /* gains: K = -Quu^-1 Qux, d = -Quu^-1 Qu, solved in ROTATED input
     * coordinates u = T ut.  T is orthonormal with its leading columns spanning
     * the heavily weighted constraint rows (Gram-Schmidt in order of decreasing
     * weight), so the huge interior-point / penalty terms w_i a_i a_i' land on
     * the diagonal of the rotated Quu and the Cholesky factorisation keeps the
     * tiny R-curvature of the remaining directions (no cancellation). */
    build_rotation(ws, k, hw, T);
    /* Quu_rot = T' Quu0 T + sum_i w_i (T'a_i)(T'a_i)' */
    qo_mm(m, m, m, Quu, m, T, m, QuuK /* scratch m x m */, m);
    qo_mtm(m, m, m, T, m, QuuK, m, QuuR, m);
    for (int ci = 0; ci < p->ncon; ++ci) {
      const qo_constraint* cn = &p->con[ci];
      if (!con_active_at(cn, k)) continue;
      for (int r = 0; r < cn->p; ++r) {
        const double wr = hw[ci][r];
        if (wr == 0.0) continue;
        double at[M_MAX];
        qo_mtv(m, m, T, m, &kw->Ju[ci][r * m], at);
        for (int j = 0; j < m; ++j)
          for (int i = 0; i < m; ++i) QuuR[j * m + i] += wr * at[j] * at[i];
      }
    }
    for (int a = 0; a < m; ++a)
      for (int b = a + 1; b < m; ++b) {
        const double sv = 0.5 * (QuuR[a * m + b] + QuuR[b * m + a]);
        QuuR[a * m + b] = sv;
        QuuR[b * m + a] = sv;
      }
    qo_mtm(m, m, ne, T, m, Qux, ne, QuxR, ne);
    qo_mtv(m, m, T, m, Qu, QuR);
    memcpy(L, QuuR, sizeof(double) * m * m);
    if (qo_chol(m, L, m) != 0) return QO_STATUS_NOT_PD;
    for (int j = 0; j < m; ++j) {
      for (int b = 0; b < ne; ++b) Kt[j * ne + b] = -QuxR[j * ne + b];
      dt[j] = -QuR[j];
    }
    qo_chol_solve(m, L, m, Kt, ne, ne);
    qo_chol_solve(m, L, m, dt, 1, 1);
    qo_mm(m, m, ne, T, m, Kt, ne, kw->K, ne);
    qo_mv(m, m, T, m, dt, kw->d);
    /* cost-to-go (rotated coordinates):
     *   P = Qxx + Kt'QuuR Kt + Kt'QuxR + QuxR'Kt ; p = Qx + Kt'QuuR dt + Kt'QuR + QuxR'dt */
    qo_mm(m, m, ne, QuuR, m, Kt, ne, QuuK, ne);
    for (int a = 0; a < ne; ++a)
      for (int b = 0; b < ne; ++b) {
        double s = Qxx[a * ne + b];
        for (int j = 0; j < m; ++j)
          s += Kt[j * ne + a] * QuuK[j * ne + b] + Kt[j * ne + a] * QuxR[j * ne + b] +
               QuxR[j * ne + a] * Kt[j * ne + b];
        P[a * ne + b] = s;
      }
    /* symmetrise */
    for (int a = 0; a < ne; ++a)
      for (int b = a + 1; b < ne; ++b) {
        const double s = 0.5 * (P[a * ne + b] + P[b * ne + a]);
        P[a * ne + b] = s;
        P[b * ne + a] = s;
      }
    qo_mv(m, m, QuuR, m, dt, tmpv);
    for (int a = 0; a < ne; ++a) {
      double s = Qx[a];
      for (int j = 0; j < m; ++j)
        s += Kt[j * ne + a] * (tmpv[j] + QuR[j]) + QuxR[j * ne + a] * dt[j];
      pv[a] = s;
    }
    ws->dV1 += qo_dot(m, dt, QuR);
    ws->dV2 += 0.5 * qo_dot(m, dt, tmpv);
  }
  return QO_STATUS_OK;
}

static void rollout_open_loop(solver_ws* ws) {
  const qo_problem* p = ws->prob;
  memcpy(ws->X, p->x0, sizeof(double) * ws->n);
  for (int k = 0; k < ws->N; ++k)
    p->dyn(p->dyn_ctx, k, &ws->X[(k + 1) * ws->n], &ws->X[k * ws->n], &ws->U[k * ws->m], p->h);
}

/* closed-loop rollout with step length alpha into (Xc,Uc) */
static void rollout_closed_loop(solver_ws* ws, double alpha) {
  const qo_problem* p = ws->prob;
  const int n = ws->n, ne = ws->ne, m = ws->m;
  double dx[NE_MAX];
  memcpy(ws->Xc, p->x0, sizeof(double) * n);
  for (int k = 0; k < ws->N; ++k) {
    const knot_ws* kw = &ws->kn[k];
    state_diff(p, &ws->Xc[k * n], &ws->X[k * n], dx);
    for (int j = 0; j < m; ++j) {
      double s = alpha * kw->d[j];
      for (int b = 0; b < ne; ++b) s += kw->K[j * ne + b] * dx[b];
      ws->dU[k * m + j] = s;
      ws->Uc[k * m + j] = ws->U[k * m + j] + s;
    }
    p->dyn(p->dyn_ctx, k, &ws->Xc[(k + 1) * n], &ws->Xc[k * n], &ws->Uc[k * m], p->h);
  }
}

/* |grad_U L_A|_inf at the current trajectory through the costate recursion */
static double stationarity(const solver_ws* ws) {
  const qo_problem* p = ws->prob;
  const int ne = ws->ne, m = ws->m, N = ws->N;
  double y[NE_MAX], yn[NE_MAX], gu[M_MAX], zp[QO_MAXP], act[QO_MAXP];
  const knot_ws* kw = &ws->kn[N];
  memcpy(y, kw->lx, sizeof(double) * ne);
  for (int ci = 0; ci < p->ncon; ++ci) {
    if (!con_active_at(&p->con[ci], N)) continue;
    al_multiplier(ws, N, ci, zp, act);
    for (int r = 0; r < p->con[ci].p; ++r)
      for (int a = 0; a < ne; ++a) y[a] += kw->Jx[ci][r * ne + a] * zp[r];
  }
  double g = 0.0;
  for (int k = N - 1; k >= 0; --k) {
    kw = &ws->kn[k];
    qo_mtv(ne, m, kw->B, m, y, gu);
    qo_mtv(ne, ne, kw->A, ne, y, yn);
    for (int j = 0; j < m; ++j) gu[j] += kw->lu[j];
    for (int a = 0; a < ne; ++a) yn[a] += kw->lx[a];
    for (int ci = 0; ci < p->ncon; ++ci) {
      if (!con_active_at(&p->con[ci], k)) continue;
      al_multiplier(ws, k, ci, zp, act);
      for (int r = 0; r < p->con[ci].p; ++r) {
        for (int j = 0; j < m; ++j) gu[j] += kw->Ju[ci][r * m + j] * zp[r];
        for (int a = 0; a < ne; ++a) yn[a] += kw->Jx[ci][r * ne + a] * zp[r];
      }
    }
    for (int j = 0; j < m; ++j) g = fmax(g, fabs(gu[j]));
    memcpy(y, yn, sizeof(double) * ne);
  }
  return g;
}

/* lambda <- Proj(lambda + rho c); returns max |delta lambda| */
static double dual_update(solver_ws* ws) {
  const qo_problem* p = ws->prob;
  double dl = 0.0;
  for (int k = 0; k <= ws->N; ++k)
    for (int ci = 0; ci < p->ncon; ++ci) {
      const qo_constraint* cn = &p->con[ci];
      if (!con_active_at(cn, k)) continue;
      knot_ws* kw = &ws->kn[k];
      if (cn->type == QO_SOC) {
        double z[QO_SOC_MAXP], pz[QO_SOC_MAXP];
        for (int i = 0; i < cn->p; ++i) z[i] = kw->lam[ci][i] - ws->rho * kw->c[ci][i];
        soc_project(cn->p, z, pz);
        for (int i = 0; i < cn->p; ++i) { dl = fmax(dl, fabs(pz[i] - kw->lam[ci][i])); kw->lam[ci][i] = pz[i]; }
        continue;
      }
      for (int i = 0; i < cn->p; ++i) {
        if (!row_on(cn, i)) continue;
        double z = kw->lam[ci][i] + ws->rho * kw->c[ci][i];
        if (cn->type == QO_INEQUALITY && z < 0.0) z = 0.0;
        dl = fmax(dl, fabs(z - kw->lam[ci][i]));
        kw->lam[ci][i] = z;
      }
    }
  return dl;
}


/* ---- the interpolating line search (upstream default, use_backtracking_linesearch = false) ------------------------------
 * RECALLED from upstream ALTRO-C (the reference pins the fork zixinz990/altro@b47202ff, CMakeLists.txt:34-40, which is not in
 * the tree): the merit phi(alpha) is the AL cost of the closed-loop rollout at step length alpha, its slope dphi(alpha) the
 * gradient of that cost at the CANDIDATE trajectory contracted with d(x,u)/d(alpha) of the rollout,
 *     dx_0/da = 0,   du_k/da = d_k + K_k dx_k/da,   dx_{k+1}/da = A_k dx_k/da + B_k du_k/da   (A, B at the candidate),
 *     dphi = sum_k (lx_k + Jx_k' z_k)' dx_k/da + (lu_k + Ju_k' z_k)' du_k/da   (+ the terminal knot),  z = Proj(lam + rho c).
 * Search: start at alpha = 1; accept when sufficient decrease (c1 = 1e-4) AND |dphi| <= c2 |dphi(0)| (c2 = 0.9) hold; grow
 * the step (x 1.5, up to 2) while the merit still falls steeply; once a minimum is bracketed, zoom with the minimiser of the
 * cubic through both end points' values and slopes (bisection when it leaves the bracket).  Only the generic known-answer
 * tests reach it: every caller on the path selects the backtracking search. */
static double merit_and_slope(solver_ws* ws, double alpha, double* plain, double* viol, double* dphi) {
  const qo_problem* p = ws->prob;
  const int ne = ws->ne, m = ws->m, N = ws->N;
  rollout_closed_loop(ws, alpha);
  const double phi = total_cost(ws, ws->Xc, ws->Uc, plain, viol);
  if (!dphi) return phi;
  if (!isfinite(phi)) { *dphi = 0.0; return phi; }
  /* expansions at the candidate: install it as the nominal trajectory for the call, then put the nominal back (the gains
   * K, d are not touched; the stale expansions are recomputed at the accepted point by the caller) */
  double* Xn = ws->X; double* Un = ws->U;
  ws->X = ws->Xc; ws->U = ws->Uc;
  expansions(ws);
  soc_refresh(ws);
  double dx[NE_MAX], dxn[NE_MAX], du[M_MAX], zp[QO_MAXP], act[QO_MAXP];
  memset(dx, 0, sizeof dx);
  double slope = 0.0;
  for (int k = 0; k <= N; ++k) {
    const knot_ws* kw = &ws->kn[k];
    if (k < N)
      for (int j = 0; j < m; ++j) {
        double sv = kw->d[j];
        for (int b = 0; b < ne; ++b) sv += kw->K[j * ne + b] * dx[b];
        du[j] = sv;
      }
    for (int a = 0; a < ne; ++a) slope += kw->lx[a] * dx[a];
    if (k < N) for (int j = 0; j < m; ++j) slope += kw->lu[j] * du[j];
    for (int ci = 0; ci < p->ncon; ++ci) {
      if (!con_active_at(&p->con[ci], k)) continue;
      al_multiplier(ws, k, ci, zp, act);
      for (int r = 0; r < p->con[ci].p; ++r) {
        double jd = 0.0;
        for (int a = 0; a < ne; ++a) jd += kw->Jx[ci][r * ne + a] * dx[a];
        if (k < N) for (int j = 0; j < m; ++j) jd += kw->Ju[ci][r * m + j] * du[j];
        slope += zp[r] * jd;
      }
    }
    if (k < N) {
      for (int a = 0; a < ne; ++a) {
        double sv = 0.0;
        for (int b = 0; b < ne; ++b) sv += kw->A[a * ne + b] * dx[b];
        for (int j = 0; j < m; ++j) sv += kw->B[a * m + j] * du[j];
        dxn[a] = sv;
      }
      memcpy(dx, dxn, sizeof(double) * ne);
    }
  }
  ws->X = Xn; ws->U = Un;
  *dphi = slope;
  return phi;
}

/* minimiser of the cubic through (a, fa, ga) and (b, fb, gb) (Nocedal & Wright, eq. 3.59); NaN when it does not exist */
static double cubic_min(double a, double fa, double ga, double b, double fb, double gb) {
  const double d1 = ga + gb - 3.0 * (fa - fb) / (a - b);
  const double rad = d1 * d1 - ga * gb;
  if (!(rad >= 0.0)) return NAN;
  const double d2 = ((b > a) ? 1.0 : -1.0) * sqrt(rad);
  return b - (b - a) * (gb + d2 - d1) / (gb - ga + 2.0 * d2);
}

/* returns 1 and the accepted step (candidate in Xc, Uc, dU) or 0; *evals counts merit evaluations */
static int linesearch_cubic(solver_ws* ws, const qo_options* o, double phi0, double dphi0, double* alpha_out, double* Jn,
                            double* Jn_plain, double* vn, int* evals) {
  const double c1 = 1e-4, c2 = 0.9, alpha_max = 2.0, beta_increase = 1.5, min_interval = 1e-6;
  const int max_iters = 25;
  if (!(dphi0 < 0.0)) return 0;       /* not a descent direction */
  double a_lo = 0.0, f_lo = phi0, g_lo = dphi0, a_hi = 0.0, f_hi = 0.0, g_hi = 0.0;
  double alpha = 1.0, a_prev = 0.0, f_prev = phi0, g_prev = dphi0;
  int bracketed = 0, hit_max = 0;
  double f = 0.0, g = 0.0, pl = 0.0, vi = 0.0;
  for (int it = 0; it < max_iters; ++it) {
    f = merit_and_slope(ws, alpha, &pl, &vi, &g);
    ++*evals;
    if (o->verbose > 1) fprintf(stderr, "   ls(cubic) bracket alpha=%.6f phi=%.15e dphi=%.3e (phi0=%.15e dphi0=%.3e)\n", alpha, f, g, phi0, dphi0);
    const int decrease = isfinite(f) && f <= phi0 + c1 * alpha * dphi0;
    if (!decrease || (it > 0 && f >= f_prev)) {
      a_lo = a_prev; f_lo = f_prev; g_lo = g_prev; a_hi = alpha; f_hi = f; g_hi = g; bracketed = 1;
      break;
    }
    if (fabs(g) <= c2 * fabs(dphi0)) { *alpha_out = alpha; *Jn = f; *Jn_plain = pl; *vn = vi; return 1; }
    if (g >= 0.0) {
      a_lo = alpha; f_lo = f; g_lo = g; a_hi = a_prev; f_hi = f_prev; g_hi = g_prev; bracketed = 1;
      break;
    }
    if (hit_max) { *alpha_out = alpha; *Jn = f; *Jn_plain = pl; *vn = vi; return 1; }
    a_prev = alpha; f_prev = f; g_prev = g;
    alpha = fmin(alpha * beta_increase, alpha_max);
    hit_max = alpha >= alpha_max;
  }
  if (!bracketed) return 0;
  for (int it = 0; it < max_iters; ++it) {
    double a = isfinite(f_hi) ? cubic_min(a_lo, f_lo, g_lo, a_hi, f_hi, g_hi) : NAN;
    const double lo = fmin(a_lo, a_hi), hi = fmax(a_lo, a_hi);
    if (!isfinite(a) || a <= lo || a >= hi) a = 0.5 * (a_lo + a_hi);
    f = merit_and_slope(ws, a, &pl, &vi, &g);
    ++*evals;
    if (o->verbose > 1) fprintf(stderr, "   ls(cubic) zoom [%.6f, %.6f] alpha=%.6f phi=%.15e dphi=%.3e\n", a_lo, a_hi, a, f, g);
    const int decrease = isfinite(f) && f <= phi0 + c1 * a * dphi0;
    if (!decrease || f >= f_lo) {
      a_hi = a; f_hi = f; g_hi = g;
    } else {
      if (fabs(g) <= c2 * fabs(dphi0)) { *alpha_out = a; *Jn = f; *Jn_plain = pl; *vn = vi; return 1; }
      if (g * (a_hi - a_lo) >= 0.0) { a_hi = a_lo; f_hi = f_lo; g_hi = g_lo; }
      a_lo = a; f_lo = f; g_lo = g;
    }
    if (fabs(a_hi - a_lo) < min_interval) {
      /* window too small: upstream returns the best point with sufficient decrease; the candidate buffers must hold it */
      if (a_lo > 0.0) {
        f = merit_and_slope(ws, a_lo, &pl, &vi, NULL);
        *alpha_out = a_lo; *Jn = f; *Jn_plain = pl; *vn = vi;
        return 1;
      }
      return 0;
    }
  }
  return 0;
}

/* ---- converged mode: primal-dual interior point on the Riccati core ----------
 * (The reference has no such mode: its AL-iLQR scheme, restated above, stalls on
 * states whose optimum sits on many cone faces.)  Newton steps on the perturbed
 * KKT system of the SAME NLP, reusing the backward pass with the interior-point
 * weights of al_multiplier().
 *
 * Slack bookkeeping.  The cone rows are linear in u, so with the input
 * increment dU of a rollout  c(U + dU) = c(U) + Ju dU exactly.  The slack
 * residual rc = c(u) + s is carried as its own variable and updated with the
 * (small, accurately known) increments, rc <- rc + Ju dU + ds, never
 * recomputed from c(u) (~100 N, rounding ~1e-14 N): that keeps the RELATIVE
 * accuracy of slacks far below 1e-14 N, which the multipliers of weakly
 * active rows need. */
static int ipm_init(solver_ws* ws, double mu0) {
  const qo_problem* p = ws->prob;
  int rows = 0;
  for (int k = 0; k <= ws->N; ++k)
    for (int ci = 0; ci < p->ncon; ++ci) {
      const qo_constraint* cn = &p->con[ci];
      if (!con_active_at(cn, k) || cn->type != QO_INEQUALITY) continue;
      knot_ws* kw = &ws->kn[k];
      for (int i = 0; i < cn->p; ++i) {
        if (!row_on(cn, i)) continue;
        kw->s[ci][i] = fmax(-kw->c[ci][i], 1.0);
        kw->rc[ci][i] = kw->c[ci][i] + kw->s[ci][i];
        kw->lam[ci][i] = mu0 / kw->s[ci][i];
        rows++;
      }
    }
  return rows;
}

static double ipm_mu(const solver_ws* ws, double* resid) {
  const qo_problem* p = ws->prob;
  double sum = 0.0, r = 0.0;
  int rows = 0;
  for (int k = 0; k <= ws->N; ++k)
    for (int ci = 0; ci < p->ncon; ++ci) {
      const qo_constraint* cn = &p->con[ci];
      if (!con_active_at(cn, k) || cn->type != QO_INEQUALITY) continue;
      const knot_ws* kw = &ws->kn[k];
      for (int i = 0; i < cn->p; ++i) {
        if (!row_on(cn, i)) continue;
        sum += kw->s[ci][i] * kw->lam[ci][i];
        r = fmax(r, fabs(kw->rc[ci][i]));
        rows++;
      }
    }
  if (resid) *resid = r;
  return rows ? sum / rows : 0.0;
}

/* Trial directions from the closed-loop rollout at alpha = 1 (increment dU):
 *   ds = -(Ju dU + rc),  dlam = (target - s lam - lam ds)/s,
 * and the fraction-to-the-boundary step lengths. */
static void ipm_directions(solver_ws* ws, double tau, double* alpha_p, double* alpha_d) {
  const qo_problem* p = ws->prob;
  const int m = ws->m, N = ws->N;
  double ap = 1.0, ad = 1.0;
  for (int k = 0; k < N; ++k) {
    knot_ws* kw = &ws->kn[k];
    const double* du = &ws->dU[k * m];
    for (int ci = 0; ci < p->ncon; ++ci) {
      const qo_constraint* cn = &p->con[ci];
      if (!con_active_at(cn, k) || cn->type != QO_INEQUALITY) continue;
      for (int i = 0; i < cn->p; ++i) {
        if (!row_on(cn, i)) continue;
        double jd = 0.0;
        for (int j = 0; j < m; ++j) jd += kw->Ju[ci][i * m + j] * du[j];
        const double sv = kw->s[ci][i], lv = kw->lam[ci][i];
        const double dsv = -(jd + kw->rc[ci][i]);
        const double dlv = (ws->ipm_target - (1.0 + kw->kap[ci][i]) * sv * lv - lv * dsv) / sv;
        kw->ds[ci][i] = dsv;
        kw->dlam[ci][i] = dlv;
        if (dsv < 0.0) ap = fmin(ap, -tau * sv / dsv);
        if (dlv < 0.0) ad = fmin(ad, -tau * lv / dlv);
      }
    }
  }
  *alpha_p = ap;
  *alpha_d = ad;
}

/* Apply the step.  A shortened primal step scales the trial increment,
 * dU <- alpha_p dU (the cone rows are linear in u, so s + alpha_p ds stays in the
 * interior exactly and rc <- (1 - alpha_p) rc); a full step zeroes rc exactly. */
static void ipm_apply(solver_ws* ws, double alpha_p, double alpha_d) {
  const qo_problem* p = ws->prob;
  for (int k = 0; k <= ws->N; ++k)
    for (int ci = 0; ci < p->ncon; ++ci) {
      const qo_constraint* cn = &p->con[ci];
      if (!con_active_at(cn, k) || cn->type != QO_INEQUALITY) continue;
      knot_ws* kw = &ws->kn[k];
      for (int i = 0; i < cn->p; ++i) {
        if (!row_on(cn, i)) continue;
        const double s0 = kw->s[ci][i], l0 = kw->lam[ci][i];
        kw->s[ci][i] += alpha_p * kw->ds[ci][i];
        kw->rc[ci][i] = (alpha_p >= 1.0) ? 0.0 : (1.0 - alpha_p) * kw->rc[ci][i];
        kw->lam[ci][i] += alpha_d * kw->dlam[ci][i];
        /* Tapia indicators: a weakly active row halves BOTH s and lambda on a full
         * Newton step (regular rows send one of the two ratios to ~1, the other to ~sigma) */
        const double s1 = kw->s[ci][i], l1 = kw->lam[ci][i];   /* ratios vs 0.4 / 0.6 without dividing */
        const int sig = (alpha_p >= 0.99 && alpha_d >= 0.99 && s1 < 0.6 * s0 && l1 < 0.6 * l0 &&
                         (kw->kap[ci][i] != 0.0 || (s1 > 0.4 * s0 && l1 > 0.4 * l0)));
        kw->kap[ci][i] = sig ? 1.0 : 0.0;
      }
    }
}

static const double* p_x0(const solver_ws* ws) { return ws->prob->x0; }

static int ipm_phase(solver_ws* ws, const qo_options* o, qo_result* r) {
  const int n = ws->n, m = ws->m, N = ws->N;
  ipm_init(ws, o->ipm_mu0);
  int it;
  r->last_step = 1e300;
  double last_ap = 0.0, last_ad = 0.0;
  int status = QO_STATUS_MAX_ITER;
  for (it = 1; it <= o->ipm_iterations_max + 1; ++it) {
    double resid;
    const double mu = ipm_mu(ws, &resid);
    r->ipm_mu = mu;
    if (mu <= o->ipm_mu_final && resid <= o->tol_feasibility && r->last_step <= o->tol_step) {
      status = QO_STATUS_OK;
      break;
    }
    if (it > o->ipm_iterations_max) break;
    ws->ipm = 1;
    {
      /* centering: sigma until full steps are taken, then the fast value; short
       * steps (jamming near the boundary) call for more centering */
      double sg = o->ipm_sigma;
      const double amin = fmin(last_ap, last_ad);
      if (it > 1 && amin >= 0.99) sg = o->ipm_sigma_fast;
      else if (it > 1 && amin < 0.2) sg = fmax(sg, 0.8);
      else if (it > 1 && amin < 0.5) sg = fmax(sg, 0.5);
      ws->ipm_target = sg * mu;
    }
    const int bp = backward_pass(ws);
    if (bp != QO_STATUS_OK) { ws->ipm = 0; status = bp; break; }
    double ap, ad;
    rollout_closed_loop(ws, 1.0);                        /* trial step */
    ipm_directions(ws, o->ipm_tau, &ap, &ad);
    last_ap = ap; last_ad = ad;
    ws->ipm = 0;
    ipm_apply(ws, ap, ad);
    /* convergence is judged on the FULL Newton step (the trial increment) */
    double step = 0.0;
    for (int i = 0; i < N * m; ++i) step = fmax(step, fabs(ws->dU[i]));
    if (ap < 1.0) {   /* shortened primal step: scaled increment, open-loop states */
      for (int i = 0; i < N * m; ++i) {
        ws->dU[i] *= ap;
        ws->Uc[i] = ws->U[i] + ws->dU[i];
      }
      memcpy(ws->Xc, p_x0(ws), sizeof(double) * n);
      for (int k = 0; k < N; ++k)
        ws->prob->dyn(ws->prob->dyn_ctx, k, &ws->Xc[(k + 1) * n], &ws->Xc[k * n], &ws->Uc[k * m], ws->prob->h);
    }
    memcpy(ws->U, ws->Uc, sizeof(double) * N * m);
    memcpy(ws->X, ws->Xc, sizeof(double) * (N + 1) * n);
    expansions(ws);
    r->ipm_iterations = it;
    r->last_step = step;
    if (o->verbose)
      fprintf(stderr, "ipm %2d  mu=%.3e  resid=%.3e  ap=%.4f ad=%.4f  step=%.3e\n", it, mu, resid,
              ap, ad, step);
  }
  return status;
}

void qo_default_options(qo_options* o, int mode) {
  memset(o, 0, sizeof *o);
  o->mode = mode;
  o->penalty_initial = 1.0;
  o->penalty_max = 1e8;
  o->tol_stationarity = 1e-4;
  o->tol_feasibility = 1e-4;
  o->tol_cost_intermediate = 1e-4;
  o->linesearch_max = 10;
  if (mode == QO_MODE_REFERENCE) {
    /* upstream ALTRO-C defaults; QuatMpc overrides iterations_max / scaling */
    o->iterations_max = 200;
    o->penalty_scaling = 10.0;
    o->tol_step = 0.0;
  } else {
    o->iterations_max = 120;
    o->penalty_scaling = 10.0;
    o->tol_feasibility = 1e-8;
    o->tol_step = 1e-8;
    o->ipm_iterations_max = 120;
    o->ipm_mu_final = 1e-12;
    o->ipm_sigma = 0.2;
    o->ipm_sigma_fast = 0.01;
    o->ipm_mu0 = 0.01;
    o->ipm_tau = 0.995;
  }
}

int qo_altro_solve(const qo_problem* prob, const qo_options* opts, double* X, double* U,
                   qo_result* res) {
  solver_ws ws;
  memset(&ws, 0, sizeof ws);
  ws.prob = prob;
  ws.n = prob->n;
  ws.m = prob->m;
  ws.N = prob->N;
  ws.ne = prob->n - (prob->use_quaternion ? 1 : 0);
  const int n = ws.n, m = ws.m, N = ws.N;
  /* per-thread scratch, allocated once and reused by every solve on this thread */
  static __thread knot_ws* tl_kn = NULL;
  static __thread double* tl_buf = NULL;
  if (!tl_kn) {
    tl_kn = (knot_ws*)malloc(sizeof(knot_ws) * (QO_MAXH + 1));
    tl_buf = (double*)malloc(sizeof(double) * ((QO_MAXH + 1) * QO_MAXN + 2 * QO_MAXH * QO_MAXM + 8));
  }
  ws.kn = tl_kn;
  for (int k = 0; k <= N; ++k) {   /* only the state that persists across iterations */
    memset(ws.kn[k].lam, 0, sizeof ws.kn[k].lam);
    memset(ws.kn[k].kap, 0, sizeof ws.kn[k].kap);
    memset(ws.kn[k].s, 0, sizeof ws.kn[k].s);
    memset(ws.kn[k].rc, 0, sizeof ws.kn[k].rc);
  }
  ws.X = X;
  ws.U = U;
  ws.Xc = tl_buf;
  ws.Uc = ws.Xc + (QO_MAXH + 1) * QO_MAXN;
  ws.dU = ws.Uc + QO_MAXH * QO_MAXM + 4;
  ws.rho = opts->penalty_initial;
  if (opts->penalty_io && *opts->penalty_io > 0.0) ws.rho = *opts->penalty_io;
  if (opts->dual_io)
    for (int k = 0; k <= N; ++k)
      for (int ci = 0; ci < prob->ncon; ++ci)
        memcpy(ws.kn[k].lam[ci], &opts->dual_io[((size_t)k * QO_MAXCON + ci) * QO_MAXP], sizeof(double) * QO_MAXP);

  qo_result r;
  memset(&r, 0, sizeof r);
  r.status = QO_STATUS_MAX_ITER;

  rollout_open_loop(&ws);
  expansions(&ws);
  if (opts->mode == QO_MODE_CONVERGED && opts->ipm_iterations_max > 0) {
    const int st = ipm_phase(&ws, opts, &r);
    if (st != QO_STATUS_OK) r.status = st;
    {
      /* converged mode ends here: the interior-point iterate IS the answer */
      double Jp, vv;
      total_cost(&ws, ws.X, ws.U, &Jp, &vv);
      r.iterations = r.ipm_iterations;
      r.cost = Jp;
      r.max_violation = vv;
      r.status = st;
      r.penalty = r.ipm_mu;
      if (res) *res = r;
      if (opts->dual_out || opts->slack_out) {
        size_t o = 0;
        for (int k = 0; k < N; ++k)
          for (int ci = 0; ci < prob->ncon; ++ci) {
            const qo_constraint* cn = &prob->con[ci];
            for (int i = 0; i < cn->p; ++i, ++o) {
              const int on = con_active_at(cn, k) && cn->type == QO_INEQUALITY && row_on(cn, i);
              if (opts->dual_out) opts->dual_out[o] = on ? ws.kn[k].lam[ci][i] : 0.0;
              if (opts->slack_out) opts->slack_out[o] = on ? ws.kn[k].s[ci][i] : 0.0;
            }
          }
      }
      return r.status;
    }
  }
  double Jplain, viol;
  double J = total_cost(&ws, ws.X, ws.U, &Jplain, &viol);
  int iter = 0;
  for (iter = 1; iter <= opts->iterations_max; ++iter) {
    const int bp = backward_pass(&ws);
    if (bp != QO_STATUS_OK) { r.status = bp; --iter; break; }
    /* forward pass: backtracking line search on the AL merit */
    double alpha = 1.0, Jn = J, Jn_plain = Jplain, vn = viol;
    int accepted = 0;
    if (opts->linesearch_cubic) {
      int evals = 0;
      accepted = linesearch_cubic(&ws, opts, J, ws.dV1, &alpha, &Jn, &Jn_plain, &vn, &evals);
      r.linesearch_halvings += evals - 1;
    } else
    for (int ls = 0; ls <= opts->linesearch_max; ++ls) {
      rollout_closed_loop(&ws, alpha);
      Jn = total_cost(&ws, ws.Xc, ws.Uc, &Jn_plain, &vn);
      const double expected = alpha * ws.dV1; /* directional derivative * alpha (< 0) */
      const double slack = 1e-12 * fmax(1.0, fabs(J));
      if (opts->verbose > 1)
        fprintf(stderr, "   ls alpha=%.3e J=%.15e Jn=%.15e expected=%.3e dV2=%.3e\n", alpha, J, Jn, expected, ws.dV2);
      if (isfinite(Jn) && Jn - J <= 1e-4 * expected + slack) { accepted = 1; break; }
      alpha *= 0.5;
      r.linesearch_halvings++;
    }
    if (!accepted) { r.status = QO_STATUS_LINESEARCH_FAIL; --iter; break; }
    double step = 0.0;
    for (int i = 0; i < N * m; ++i) step = fmax(step, fabs(ws.Uc[i] - ws.U[i]));
    memcpy(ws.X, ws.Xc, sizeof(double) * (N + 1) * n);
    memcpy(ws.U, ws.Uc, sizeof(double) * N * m);
    const double dJ = J - Jn;
    J = Jn; Jplain = Jn_plain; viol = vn;
    expansions(&ws);
    r.last_step = step;
    if (opts->verbose)
      fprintf(stderr, "iter %2d  J=%.12e  dJ=%.3e  alpha=%.4f  step=%.3e  viol=%.3e  rho=%.1e\n",
              iter, J, dJ, alpha, step, viol, ws.rho);
    if (opts->mode == QO_MODE_REFERENCE) {
      soc_refresh(&ws);   /* rotated rows / multipliers at the accepted point */
      const double stat = stationarity(&ws);
      r.stationarity = stat;
      if (opts->verbose) fprintf(stderr, "         stationarity=%.3e\n", stat);
      if (stat < opts->tol_stationarity && viol < opts->tol_feasibility) {
        r.status = QO_STATUS_OK;
        break;
      }
      if (stat < opts->tol_stationarity || fabs(dJ) < opts->tol_cost_intermediate) {
        dual_update(&ws);
        ws.rho = fmin(ws.rho * opts->penalty_scaling, opts->penalty_max);
        r.dual_updates++;
        J = total_cost(&ws, ws.X, ws.U, &Jplain, &viol);
      }
    }
  }
  if (opts->penalty_io) *opts->penalty_io = ws.rho;
  if (opts->dual_io)
    for (int k = 0; k <= N; ++k)
      for (int ci = 0; ci < prob->ncon; ++ci)
        memcpy(&opts->dual_io[((size_t)k * QO_MAXCON + ci) * QO_MAXP], ws.kn[k].lam[ci], sizeof(double) * QO_MAXP);
  if (iter > opts->iterations_max) iter = opts->iterations_max;
  r.iterations = iter;
  r.cost = Jplain;
  r.max_violation = viol;
  r.penalty = ws.rho;
  if (res) *res = r;

  return r.status;
}
