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

// ---- line search of the four-point QuatMpc problem on the dense kernels, THREE trial step lengths per rollout --------------
// The closed-loop rollout (rollout_closed, qmpc_kernels.hip) keeps the 16 lanes 4 l + a busy and lets every lane advance
// the same state; the trials of one line search are independent of each other.  Each 16-lane row of the wavefront therefore
// rolls its own step length alpha 2^-g out (rows 0..2; row 3 repeats row 2): the inputs are broadcast inside the row
// (row_newbcast), the state cost of the row's trajectory is summed on the way, increments and states are left in
//   group 0: dU, Xc      group 1: DLAM (12 N), S (13 N)      group 2: XT (12 N), DS (13 N)
// -- the reference mode uses none of S / DS / DLAM during the search (the input weights of stationarity_w sit in the last 12
// entries of DLAM), and the cost expansions in XT are dead between the backward pass and the expansions at the accepted point.
// The same idea as rollout_trials_w (qmpc_wform.h), which has four groups because its wrench-space records are smaller.
__device__ __forceinline__ int trial_du_slot(const Layout& L, int g) { return g == 0 ? L.dU : (g == 1 ? L.DLAM : L.XT); }
__device__ __forceinline__ int trial_x_slot(const Layout& L, int g) { return g == 0 ? L.Xc + 13 : (g == 1 ? L.S : L.DS); }

template <class MD, bool LEAN>
__device__ inline double rollout_trials_dense(const DevParams& P, const Layout& L, double* sm, const double* KD,
                                              const double* ROT, double alpha_g, int lane) {
  typedef typename MD::D D;
  static_assert(D::NU == 12, "one 16-lane row per trial: four contact points");
  const int N = P.N;
  const double* cst = sm + L.cst;
  typename MD::template RegsT<LEAN> M;
  M.load(cst, sm + L.bw0);
  const int r = lane & 15, g = (lane >> 4) < 2 ? (lane >> 4) : 2;
  const bool ulane = (r & 3) < 3;
  const int ql = ulane ? (r >> 2) : 0, qa = ulane ? (r & 3) : 0;
  const int uj = 3 * ql + qa;
  double* dug = sm + trial_du_slot(L, g);
  double* xg = sm + trial_x_slot(L, g);
  double xc[13], xn[13];
#pragma unroll
  for (int i = 0; i < 13; ++i) xc[i] = cst[D::C_X0 + i];
  if (lane == 15)
#pragma unroll
    for (int i = 0; i < 13; ++i) sm[L.Xc + i] = xc[i];
  double Jx = MD::knot_cost(P, sm + L.refp, sm + L.uref, 0, xc, nullptr);
  for (int k = 0; k < N; ++k) {
    RollLoads cur;
    roll_load<D, true>(L, sm, KD, ROT, k, uj, ql, qa, cur);
    double dx[12];
    MD::state_diff(cur.xo, xc, dx);
    const double* kd = cur.kd;
    const double p0 = alpha_g * kd[12] + kd[0] * dx[0] + kd[1] * dx[1] + kd[2] * dx[2];
    const double p1 = kd[3] * dx[3] + kd[4] * dx[4] + kd[5] * dx[5];
    const double p2 = kd[6] * dx[6] + kd[7] * dx[7] + kd[8] * dx[8];
    const double p3 = kd[9] * dx[9] + kd[10] * dx[10] + kd[11] * dx[11];
    const double s = (p0 + p1) + (p2 + p3);
    const double s0 = dpp_mov<0x00>(s), s1 = dpp_mov<0x55>(s), s2 = dpp_mov<0xAA>(s);   // quad_perm broadcasts
    const double inc = cur.T[0] * s0 + cur.T[1] * s1 + cur.T[2] * s2;
    if (ulane) dug[D::NU * k + uj] = inc;
    const double unew = cur.uo + inc;
    double un[12];      // input 3 l + a from lane 4 l + a of the row (row_newbcast)
    un[0] = dpp_mov<0x150>(unew); un[1] = dpp_mov<0x151>(unew); un[2] = dpp_mov<0x152>(unew);
    un[3] = dpp_mov<0x154>(unew); un[4] = dpp_mov<0x155>(unew); un[5] = dpp_mov<0x156>(unew);
    un[6] = dpp_mov<0x158>(unew); un[7] = dpp_mov<0x159>(unew); un[8] = dpp_mov<0x15A>(unew);
    un[9] = dpp_mov<0x15C>(unew); un[10] = dpp_mov<0x15D>(unew); un[11] = dpp_mov<0x15E>(unew);
    MD::template step<LEAN>(P, M, xc, un, xn);
#pragma unroll
    for (int i = 0; i < 13; ++i) xc[i] = xn[i];
    if (r == 15)
#pragma unroll
      for (int i = 0; i < 13; ++i) xg[13 * k + i] = xn[i];
    Jx += MD::knot_cost(P, sm + L.refp, sm + L.uref, k + 1, xc, nullptr);
  }
  QSYNC();
  return Jx;
}

// the inputs' share of the three trials' merit, one lane per (knot, contact point): u = U + dU_g, input cost,
// augmented-Lagrangian terms and violation of the point's cone rows (the arithmetic of ref_merit).  Per-lane partial sums:
// Ju = input cost, mer = Ju + (AL terms) / (2 rho), vi = violation.
template <class D>
__device__ inline void trial_inputs_dense(const DevParams& P, const Layout& L, const double* sm, const double* sl,
                                          const double* Rl, double rho, int lane, double Ju[3], double mer[3], double vi[3]) {
  const int N = P.N;
  const double* cst = sm + L.cst;
  const double* cr = cst + D::C_CR;
  double al[3];
#pragma unroll
  for (int g = 0; g < 3; ++g) { Ju[g] = 0.0; al[g] = 0.0; vi[g] = 0.0; }
  for (int q = lane; q < 4 * N; q += kWave) {
    const int k = q >> 2, l = q & 3;
    const bool stance = cst[D::C_CON + l] != 0.0;
    double u0[3], ur[3], Rw[3], lam[6];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      u0[a] = sm[L.U + D::NU * k + 3 * l + a];
      ur[a] = sm[L.uref + 3 * l + a];
      Rw[a] = Rl[3 * l + a];
    }
#pragma unroll
    for (int i = 0; i < 6; ++i) lam[i] = sl[L.LAM + D::NC * k + 6 * l + i];
    const double fzc = -P.fz_max * cst[D::C_CON + l];
#pragma unroll
    for (int g = 0; g < 3; ++g) {
      const double* du = sm + trial_du_slot(L, g) + D::NU * k + 3 * l;
      double u[3];
#pragma unroll
      for (int a = 0; a < 3; ++a) u[a] = u0[a] + du[a];
#pragma unroll
      for (int a = 0; a < 3; ++a) { const double e = u[a] - ur[a]; Ju[g] += 0.5 * Rw[a] * e * e; }
      if (stance) {
#pragma unroll
        for (int i = 0; i < 6; ++i) {
          double c = cr[3 * i] * u[0] + cr[3 * i + 1] * u[1] + cr[3 * i + 2] * u[2];
          if (i == 4) c += fzc;
          double zz = lam[i] + rho * c;
          if (zz < 0.0) zz = 0.0;
          al[g] += zz * zz - lam[i] * lam[i];
          vi[g] = fmax(vi[g], fmax(c, 0.0));
        }
      }
    }
  }
  const double i2r = 1.0 / (2.0 * rho);
#pragma unroll
  for (int g = 0; g < 3; ++g) mer[g] = Ju[g] + al[g] * i2r;
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
