// qmpc_ref.hip -- QMPC_MODE_REFERENCE on the device: the reference's OWN operating mode, i.e. the AL-iLQR scheme of
// its external solver with QuatMpc's settings (iterations_max = 10, penalty_scaling = 20, backtracking line search,
// status ignored; legged_ctrl/src/mpc/QuatMpc.cpp:21-26,256), as SURVEY.md Appendix B states it (the CPU checker under oracle/ restates the same scheme):
//
//   lambda <- 0, rho <- penalty_initial; U <- u_ref; X <- rollout; J <- AL merit
//   repeat iter = 1 .. iterations_max:
//     backward Riccati pass with AL weights  w_i = rho [lambda_i + rho c_i > 0],  g_i = max(lambda_i + rho c_i, 0)
//     forward pass u = u + alpha d + K dx, backtracking on alpha until the AL merit decreases (Armijo, 1e-4)
//     stationarity |grad_U L_A|_inf at the NEW trajectory (costate recursion), feasibility, dJ
//     converged: stationarity < tol and feasibility < tol
//     if stationarity < tol or |dJ| < tol_cost_intermediate:  lambda <- max(lambda + rho c, 0);  rho <- rho * scaling
//
// It shares the whole solver core with the converged mode (qmpc_kernels.hip): set-up, expansions, rotation pre-pass
// (AL weights), register-resident MFMA backward pass with the rotated Gauss-Jordan stage solve, closed-loop rollout.
// The result is the reference-style TRUNCATED iterate: what the robot would have applied.  Slot use: LAM = lambda,
// RC = c(U), S = candidate inputs (scratch), DS / DLAM unused.
#pragma once

#include "qmpc_device.h"      // included after qmpc_kernels.hip by qmpc_hip.hip
#include "qmpc_wform.h"       // stationarity_w: the row-parallel costate sweep (four-point QuatMpc)

namespace qmpc {

// c(U) of every cone row into the RC slot; returns this lane's share of the violation max(c, 0)
template <class D>
__device__ inline double ref_cone_refresh(const DevParams& P, const Layout& L, double* sm, double* sl, unsigned conmask,
                                          int lane) {
  double v = 0.0;
  for (int i = lane; i < P.N * D::NC; i += kWave) {
    const double c = cone_value<D>(P, L, sm, i);
    sl[L.RC + i] = c;
    if (conmask & (1u << ((i % D::NC) / 6))) v = fmax(v, fmax(c, 0.0));
  }
  return v;
}

// AL merit of a trajectory: plain objective + sum_rows (max(lambda + rho c, 0)^2 - lambda^2) / (2 rho).
// CAND: the candidate (Xc, U + dU; its inputs are first written to the S slot); else the current (X, U).
template <class MD, bool CAND>
__device__ inline double ref_merit(const DevParams& P, const Layout& L, double* sm, double* sl, double rho,
                                   unsigned conmask, int lane, double* plain, double* viol) {
  typedef typename MD::D D;
  const int N = P.N;
  const double* cst = sm + L.cst;
  const double* cr = cst + D::C_CR;
  double* Uc = sl + L.S;
  if (CAND) {
    for (int i = lane; i < N * D::NU; i += kWave) Uc[i] = sm[L.U + i] + sm[L.dU + i];
    QSYNC();
  }
  const double* Xp = sm + (CAND ? L.Xc : L.X);
  const double* Up = CAND ? Uc : sm + L.U;
  double J = 0.0;
  if (lane <= N) J = MD::knot_cost(P, sm + L.refp, sm + L.uref, lane, Xp + 13 * lane, (lane < N) ? Up + D::NU * lane : nullptr);
  double al = 0.0, v = 0.0;
  for (int idx = lane; idx < N * D::NC; idx += kWave) {
    const int k = idx / D::NC, row = idx - D::NC * k, l = row / 6, i = row - 6 * l;
    if (!(conmask & (1u << l))) continue;
    const double* u = Up + D::NU * k + 3 * l;
    double c = cr[3 * i] * u[0] + cr[3 * i + 1] * u[1] + cr[3 * i + 2] * u[2];
    if (i == 4) c += -P.fz_max * cst[D::C_CON + l];
    const double lam = sl[L.LAM + idx];
    double z = lam + rho * c;
    if (z < 0.0) z = 0.0;
    al += z * z - lam * lam;
    v = fmax(v, fmax(c, 0.0));
  }
  J = wave_sum(J);
  al = wave_sum(al);
  if (plain) *plain = J;
  if (viol) *viol = wave_max(v);
  return J + al / (2.0 * rho);
}

// |grad_U L_A|_inf at (X, U) through the costate recursion  y_k = lx_k + Abar_k' y_{k+1},
// gu_k = R (u_k - u_ref) + Bbar_k' y_{k+1} + sum_i zp_i a_i   (costate recursion of SURVEY.md Appendix B)
template <class MD>
__device__ inline double ref_stationarity(const DevParams& P, const Layout& L, double* sm, const double* sl, double rho,
                                          unsigned conmask, int lane) {
  typedef typename MD::D D;
  const int N = P.N;
  const double* cst = sm + L.cst;
  const double* bw0 = sm + L.bw0;
  const double* cr = cst + D::C_CR;
  double* y = sm + L.tile;                 // 12 doubles of scratch (all-LDS layout: the MFMA tile; else the set-up alias)
  double yv = (lane < 12) ? sm[L.XT + kXT * N + 9 + lane] : 0.0;
  double g = 0.0;
  for (int k = N - 1; k >= 0; --k) {
    QSYNC();
    if (lane < 12) y[lane] = yv;
    QSYNC();
    const double* AB = sm + L.AB + kAB * k;
    if (lane < D::NU) {
      const int j = lane, l = j / 3, a = j - 3 * l;
      double gu = 0.0;
      if (conmask & (1u << l)) {
        for (int r = 0; r < 12; ++r) gu += MD::b_elem(P, cst, bw0, AB, r, j) * y[r];
        gu += P.R[j % 12] * (sm[L.U + D::NU * k + j] - sm[L.uref + j]);
        for (int i = 0; i < 6; ++i) {
          const int idx = D::NC * k + 6 * l + i;
          const double z = sl[L.LAM + idx] + rho * sl[L.RC + idx];
          if (z > 0.0) gu += z * cr[3 * i + a];
        }
      }
      g = fmax(g, fabs(gu));
    }
    if (lane < 12) {
      double s = sm[L.XT + kXT * k + 9 + lane];
      for (int r = 0; r < 12; ++r) s += MD::a_elem(P, cst, bw0, AB, r, lane) * y[r];
      yv = s;
    }
  }
  return wave_max(g);
}

template <class MD, int VAR>
__global__ __launch_bounds__(64, 1) void qmpc_ref_kernel(   // one wave per SIMD: the line-search loop keeps ~400 registers live
    DevParams P, const qmpc_input* __restrict__ in_, double* __restrict__ forces, qmpc_info* __restrict__ info,
    double* __restrict__ traj_u, double* __restrict__ traj_x, int batch, double* __restrict__ gws) {
  extern __shared__ __attribute__((aligned(16))) double sm[];
  const int b = blockIdx.x;
  if (b >= batch) return;
  const int lane = threadIdx.x;
#include "qmpc_ref_body.inc"
}

}  // namespace qmpc
