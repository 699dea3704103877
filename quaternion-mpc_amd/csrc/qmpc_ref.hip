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
  typedef typename MD::D D;
  constexpr int NU = D::NU, NC = D::NC;
  extern __shared__ __attribute__((aligned(16))) double sm[];
  const int b = blockIdx.x;
  if (b >= batch) return;
  const int lane = threadIdx.x;
  const int N = P.N;
  constexpr bool KDG = VAR >= 1;
  constexpr bool LEAN = KDG || MD::NL != 4;
  const Layout L = make_layout(N, KDG, MD::NL, false);
  const size_t slice = (size_t)N * (D::KD + D::ROT);
  double* KD = KDG ? gws + (size_t)b * slice : sm + L.KD;
  double* ROT = KDG ? KD + N * D::KD : sm + L.ROT;
  double* sl = sm;
  const void* in = reinterpret_cast<const double*>(in_) + (size_t)b * ((MD::NX == 13) ? D::REC : 48);
  int status = QMPC_OK;
  Prof<false> prof;
  setup_instance<MD>(P, L, sm, in, lane, &status);
  if (status != QMPC_OK) {
    if (lane < NU) forces[NU * (size_t)b + lane] = 0.0;
    if (lane == 0 && info) {
      qmpc_info r = {status, 0, 0.0, 0.0, 0.0, 0.0};
      info[b] = r;
    }
    if (traj_u) for (int i = lane; i < N * NU; i += kWave) traj_u[(size_t)b * N * NU + i] = 0.0;
    if (traj_x) for (int i = lane; i < (N + 1) * MD::NX; i += kWave) traj_x[(size_t)b * (N + 1) * MD::NX + i] = 0.0;
    return;
  }
  unsigned conmask = 0;
  for (int l = 0; l < MD::NL; ++l) conmask |= (sm[L.cst + D::C_CON + l] != 0.0) ? (1u << l) : 0u;
  conmask = __builtin_amdgcn_readfirstlane(conmask);
  for (int i = lane; i < N * NU; i += kWave) sm[L.U + i] = sm[L.uref + (i % NU)];
  for (int i = lane; i < N * NC; i += kWave) sl[L.LAM + i] = 0.0;
  QSYNC();
  rollout_open<MD, LEAN>(P, L, sm, lane);
  expansions<MD>(P, L, sm, lane);
  ref_cone_refresh<D>(P, L, sm, sl, conmask, lane);
  QSYNC();
  double rho = P.penalty_initial;
  double Jplain = 0.0, viol = 0.0;
  double J = ref_merit<MD, false>(P, L, sm, sl, rho, conmask, lane, &Jplain, &viol);
  int iter = 0;
  double last_step = 0.0;
  status = QMPC_MAX_ITER;
  for (iter = 1; iter <= P.iterations_max; ++iter) {
    rotation_prepass<D, true>(P, L, sm, sl, ROT, rho, lane);
    if (KDG) __syncthreads();
    double dV1 = 0.0;
    if (backward_pass<MD, false, (!KDG || QMPC_PIPE_ALL), (D::TU > 1)>(P, L, sm, KD, ROT, lane, conmask, prof, &dV1)) {
      status = QMPC_NOT_PD;
      --iter;
      break;
    }
    if (KDG) __syncthreads();
    // forward pass: backtracking line search on the AL merit
    double alpha = 1.0, Jn = J, Jn_plain = Jplain, vn = viol;
    bool accepted = false;
    for (int ls = 0; ls <= P.linesearch_max; ++ls) {
      rollout_closed<MD, !KDG, QMPC_PF_K, LEAN, false>(P, L, sm, KD, ROT, alpha, lane, prof);
      Jn = ref_merit<MD, true>(P, L, sm, sl, rho, conmask, lane, &Jn_plain, &vn);
      const double expected = alpha * dV1;
      const double slack = 1e-12 * fmax(1.0, fabs(J));
      if (isfinite(Jn) && Jn - J <= 1e-4 * expected + slack) { accepted = true; break; }
      alpha *= 0.5;
    }
    if (!accepted) {
      status = QMPC_LINESEARCH_FAIL;
      --iter;
      break;
    }
    double step = 0.0;
    for (int i = lane; i < N * NU; i += kWave) {
      step = fmax(step, fabs(sm[L.dU + i]));
      sm[L.U + i] += sm[L.dU + i];
    }
    for (int i = lane; i < (N + 1) * 13; i += kWave) sm[L.X + i] = sm[L.Xc + i];
    last_step = wave_max(step);
    QSYNC();
    const double dJ = J - Jn;
    J = Jn; Jplain = Jn_plain; viol = vn;
    expansions<MD>(P, L, sm, lane);
    ref_cone_refresh<D>(P, L, sm, sl, conmask, lane);
    QSYNC();
    const double stat = ref_stationarity<MD>(P, L, sm, sl, rho, conmask, lane);
    if (stat < P.tol_stat && viol < P.tol_feas) {
      status = QMPC_OK;
      break;
    }
    if (stat < P.tol_stat || fabs(dJ) < P.tol_cost_int) {
      for (int i = lane; i < N * NC; i += kWave) {       // lambda <- max(lambda + rho c, 0)
        const double z = sl[L.LAM + i] + rho * sl[L.RC + i];
        sl[L.LAM + i] = (z > 0.0) ? z : 0.0;
      }
      rho = fmin(rho * P.penalty_scaling, P.penalty_max);
      QSYNC();
      J = ref_merit<MD, false>(P, L, sm, sl, rho, conmask, lane, &Jplain, &viol);
    }
  }
  if (iter > P.iterations_max) iter = P.iterations_max;
  if (lane < NU) forces[NU * (size_t)b + lane] = sm[L.U + lane];
  if (traj_u) for (int i = lane; i < N * NU; i += kWave) traj_u[(size_t)b * N * NU + i] = sm[L.U + i];
  if (traj_x)
    for (int i = lane; i < (N + 1) * MD::NX; i += kWave) {
      const int k = i / MD::NX, j = i - MD::NX * k;
      traj_x[(size_t)b * (N + 1) * MD::NX + i] = sm[L.X + 13 * k + j];
    }
  if (info && lane == 0) {
    qmpc_info r = {status, iter, Jplain, viol, last_step, rho};
    info[b] = r;
  }
}

}  // namespace qmpc
