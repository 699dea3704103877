// qmpc_params_dev.h -- DevParams: the kernels' by-value copy of qmpc_params plus derived constants.
// No HIP dependency: the lane-per-instance core (qmpc_lane_core.h) is also compiled by g++ for its CPU numerics
// test, which needs the same structure.
#pragma once

#include <cmath>
#include <cstring>

#include "../../include/qmpc.h"

namespace qmpc {

// Device copy of qmpc_params plus derived constants (host fills it).
struct DevParams {
  int N;
  int mode;
  int iterations_max;
  int drop_ang_vel;
  double h;        // (double)(float h)
  double hh;       // (double)(h/2) with h float
  double h_ref;
  double mass;
  double inv_mass;
  double Iinv[9];
  double Q[13];
  double R[12];
  double w;
  double mu;
  double fz_max;
  double tol_feas, tol_step, mu0, mu_final, sigma, sigma_fast, tau;
  // reference mode (AL-iLQR, QuatMpc.cpp:21-26 + upstream ALTRO defaults)
  double penalty_initial, penalty_scaling, penalty_max, tol_stat, tol_cost_int;
  int linesearch_max;
};

// qmpc_params -> DevParams; QMPC_OK or QMPC_BAD_ARGUMENT
inline int fill_dev_params(const qmpc_params* p, DevParams* d) {
  if (!p || p->horizon < 1 || p->horizon > QMPC_MAX_HORIZON) return QMPC_BAD_ARGUMENT;
  if (p->mode != QMPC_MODE_CONVERGED && p->mode != QMPC_MODE_REFERENCE) return QMPC_BAD_ARGUMENT;
  if (p->mode == QMPC_MODE_REFERENCE && !(p->penalty_initial > 0.0 && p->penalty_scaling >= 1.0)) return QMPC_BAD_ARGUMENT;
  if (p->model != QMPC_MODEL_QUAT && p->model != QMPC_MODEL_CONVEX && p->model != QMPC_MODEL_QUAT8)
    return QMPC_BAD_ARGUMENT;
  if (!(p->mass > 0.0) || !(p->h > 0.0f)) return QMPC_BAD_ARGUMENT;
  if (p->mode == QMPC_MODE_CONVERGED && !(p->ipm_mu0 > 0.0)) return QMPC_BAD_ARGUMENT;
  std::memset(d, 0, sizeof *d);
  d->N = p->horizon;
  d->mode = p->mode;
  d->iterations_max = p->iterations_max;
  d->drop_ang_vel = p->drop_ang_vel;
  d->h = (double)p->h;
  d->hh = (double)(p->h / 2);  // float division, as `h / 2` in AltroUtils.cpp:16,94
  d->h_ref = p->h_ref;
  d->mass = p->mass;
  d->inv_mass = 1.0 / p->mass;
  // cofactor inverse of the 3x3 inertia (Eigen's fixed-size inverse(), AltroUtils.cpp:391)
  const double* A = p->inertia;
  const double c00 = A[4] * A[8] - A[5] * A[7], c01 = A[5] * A[6] - A[3] * A[8], c02 = A[3] * A[7] - A[4] * A[6];
  const double det = A[0] * c00 + A[1] * c01 + A[2] * c02;
  if (!(std::fabs(det) > 0.0)) return QMPC_BAD_ARGUMENT;
  const double id = 1.0 / det;
  d->Iinv[0] = c00 * id; d->Iinv[1] = (A[2] * A[7] - A[1] * A[8]) * id; d->Iinv[2] = (A[1] * A[5] - A[2] * A[4]) * id;
  d->Iinv[3] = c01 * id; d->Iinv[4] = (A[0] * A[8] - A[2] * A[6]) * id; d->Iinv[5] = (A[2] * A[3] - A[0] * A[5]) * id;
  d->Iinv[6] = c02 * id; d->Iinv[7] = (A[1] * A[6] - A[0] * A[7]) * id; d->Iinv[8] = (A[0] * A[4] - A[1] * A[3]) * id;
  std::memcpy(d->Q, p->q_weights, sizeof d->Q);
  std::memcpy(d->R, p->r_weights, sizeof d->R);
  for (int j = 0; j < 12; ++j) if (!(d->R[j] > 0.0)) return QMPC_BAD_ARGUMENT;
  d->w = p->w;
  d->mu = p->mu;
  d->fz_max = p->fz_max;
  d->tol_feas = p->tol_feasibility;
  d->tol_step = p->tol_step;
  d->mu0 = p->ipm_mu0;
  d->mu_final = p->ipm_mu_final;
  d->sigma = p->ipm_sigma;
  d->sigma_fast = p->ipm_sigma_fast;
  d->tau = p->ipm_tau;
  d->penalty_initial = p->penalty_initial;
  d->penalty_scaling = p->penalty_scaling;
  d->penalty_max = p->penalty_max;
  d->tol_stat = p->tol_stationarity;
  d->tol_cost_int = p->tol_cost_intermediate;
  d->linesearch_max = p->linesearch_max;
  return QMPC_OK;
}

}  // namespace qmpc
