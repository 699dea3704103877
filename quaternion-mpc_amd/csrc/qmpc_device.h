// qmpc_device.h -- device-side building blocks of the batched MPC inner loop for gfx950 (MI355X).
// One 64-lane wavefront owns one MPC instance; its working set lives in LDS (the gains and, for long
// horizons and large batches, the slack arrays move to an L2-resident workspace: make_layout()).
//
// Contents: DevParams, the LDS layout, FP64 MFMA helpers on [12][16] tiles, cross-lane primitives (DPP,
// v_permlane*_swap, v_readlane), wave reductions, and the model policies QuatModelT<NL> / ConvexModel.
//
// Reference arithmetic being accelerated (zixinz990/quaternion-mpc, legged_ctrl/):
//   src/utils/AltroUtils.cpp:363-439   ct_srb_quat_dynamics / _jacobian
//   src/utils/AltroUtils.cpp:224-359   ct_srb_dynamics / _jacobian (ConvexMpc's model)
//   src/utils/AltroUtils.cpp:9-22,78-110 explicit midpoint + chain rule (float h)
//   src/utils/QuaternionUtils.cpp:30-52 L(q), G(q)
//   src/mpc/QuatMpc.cpp:109-276, src/mpc/ConvexMpc.cpp:81-198   problem construction, cone rows, output
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/qmpc.h"
#include "qmpc_params_dev.h"

namespace qmpc {

constexpr int kWave = 64;
constexpr int LD = 16;          // leading dimension of every LDS matrix: [12][16]
constexpr int MAT = 12 * LD;    // 192 doubles = 3 MFMA fragments of 64 lanes

typedef double d4 __attribute__((ext_vector_type(4)));

// ---- per-instance LDS layout (offsets in doubles) ---------------------------
struct Layout {
  int cst, bw0, refp, uref, X, U, Xc, dU, S, LAM, DS, DLAM, RC, AB, XT, KD, ROT, tile, total;
  int CV;      // wrench-form layouts of ConvexMpc's problem (make_layout_w): 8 doubles per knot, Iw(yaw_m)^-1 and Iw(yaw_m)
};

// Per-knot record sizes
constexpr int kAB = 27;    // Aphiphi(9) Aphiw(9) W(9)
constexpr int kXT = 21;    // lxx attitude block (9), lx (12)
// gains [K | d] (13 per input row) and rotation blocks (21 per contact point: T(9), Dblk(9), gq(3)) are sized
// by Dim<NL>::KD / ::ROT below

// Sizes and cst[] slots for NL contact points (4: Go1, the reference; 8: the synthetic biped of
// BASELINE config 5).  Inputs come in tiles of 12 (= 4 contact points = one [12][16] MFMA tile).
template <int NL>
struct Dim {
  static constexpr int NLEG = NL, NU = 3 * NL, NC = 6 * NL, TU = NL / 4;
  static constexpr int KD = 13 * NU, ROT = 21 * NL;        // per-knot gains / rotation blocks
  static constexpr int C_FOOT = 0, C_GB = 3 * NL, C_WD0 = C_GB + 3, C_CR = C_WD0 + 3, C_CON = C_CR + 18,
                       C_X0 = C_CON + NL;
  static constexpr int REC = 32 + 4 * NL;                  // doubles per input record (48 / 64)
  static constexpr int R_FOOT = 19, R_CON = 19 + 3 * NL, R_POS = 19 + 4 * NL, R_QD = R_POS + 9;
};

// kd_global: the per-knot gains KD and rotation blocks ROT (the two largest
// arrays) live in an HBM/L2-resident workspace instead of LDS; N=20 then fits 4
// instances per CU (36 KB) instead of 2 (75 KB), N=10 fits 8 (19 KB).
// sl_global: the five slack / multiplier arrays (lane-parallel, coalesced accesses only) live in the
// workspace as well; offsets S..RC are then relative to that slice.  N=20 drops from 36 KB to 17 KB of LDS
// (two waves per SIMD instead of one).
__host__ __device__ inline Layout make_layout(int N, bool kd_global = false, int nl = 4, bool sl_global = false) {
  Layout L;
  int o = 0;
  auto take = [&](int n) { int r = o; o += n; return r; };
  const int nu = 3 * nl, nc = 6 * nl;
  L.cst = take(nl == 4 ? 64 : 80);
  L.bw0 = take(3 * nu);
  L.refp = take(13);
  L.uref = take(nu);
  L.X = take((N + 1) * 13);
  L.U = take(N * nu);
  L.Xc = take((N + 1) * 13);
  L.dU = take(N * nu);   // candidate input increment alpha d + K dx (kept as computed)
  if (sl_global) {
    L.S = 0; L.LAM = N * nc; L.DS = 2 * N * nc; L.DLAM = 3 * N * nc; L.RC = 4 * N * nc;
  } else {
    L.S = take(N * nc);
    L.LAM = take(N * nc);
    L.DS = take(N * nc);
    L.DLAM = take(N * nc);
    L.RC = take(N * nc);   // slack residual c(u) + s, tracked analytically
  }
  L.AB = take(N * kAB);
  L.XT = take((N + 1) * kXT);
  if (kd_global) {
    L.KD = -1;
    L.ROT = -1;
    // set-up scratch (one record, <= 64 doubles) aliases arrays that are not yet initialised: the slack
    // array, or X..U..Xc (contiguous) when the slacks are in the workspace
    L.tile = sl_global ? L.X : L.S;
  } else {
    L.KD = take(N * 13 * nu);
    L.ROT = take(N * 21 * nl);
    o = (o + 1) & ~1;
    L.tile = take(MAT);
  }
  L.total = (o + 1) & ~1;
  return L;
}

// cst[] slots
enum {
  C_FOOT = 0,    // 12: foot_pos_body[3*leg+axis]
  C_GB = 12,     // 3 : R' (0,0,-9.81)
  C_WD0 = 15,    // 3 : Iinv * (c x 5.204 g_body)
  C_CR = 18,     // 18: C_mat * R   (6x3 row-major)
  C_CON = 36,    // 4 : contacts (0/1)
  C_X0 = 40,     // 13: initial state
};

#define QSYNC() __syncthreads()

// ---- FP64 MFMA on [12][16] LDS tiles ----------------------------------------
// v_mfma_f64_16x16x4_f64: A operand lane l = A[i=l&15][k=l>>4], B operand lane l
// = B[k=l>>4][j=l&15], D reg r lane l = D[row=4r+(l>>4)][col=l&15].  With tiles
// stored row-major [12][16], fragment kk of a tile is simply tile[64*kk + lane],
// for the B operand AND (meaning the transpose) for the A operand.
//   C = X' * Y  (12x16 result rows 0..11; rows 12..15 of the MFMA output dropped)
__device__ __forceinline__ void mtm_load(const double* X, const double* Y, int lane, d4& acc) {
  const double x0 = X[lane], x1 = X[64 + lane], x2 = X[128 + lane];
  const double y0 = Y[lane], y1 = Y[64 + lane], y2 = Y[128 + lane];
  acc = __builtin_amdgcn_mfma_f64_16x16x4f64(x0, y0, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f64_16x16x4f64(x1, y1, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f64_16x16x4f64(x2, y2, acc, 0, 0, 0);
}
// register-resident fragments: acc += X' * Y
__device__ __forceinline__ d4 mtm3(const double X[3], const double Y[3], d4 acc) {
  acc = __builtin_amdgcn_mfma_f64_16x16x4f64(X[0], Y[0], acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f64_16x16x4f64(X[1], Y[1], acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f64_16x16x4f64(X[2], Y[2], acc, 0, 0, 0);
  return acc;
}
__device__ __forceinline__ void mtm(double* C, const double* X, const double* Y, int lane) {
  d4 acc = {0.0, 0.0, 0.0, 0.0};
  mtm_load(X, Y, lane, acc);
  C[lane] = acc[0];
  C[64 + lane] = acc[1];
  C[128 + lane] = acc[2];
}

// ---- cross-lane primitives (no LDS): DPP inside a 16-lane row, gfx950
// v_permlane16_swap / v_permlane32_swap across rows -----------------------------
typedef unsigned u2v __attribute__((ext_vector_type(2)));

template <int CTRL>
__device__ __forceinline__ double dpp_mov(double x) {
  int lo = __double2loint(x), hi = __double2hiint(x);
  // every lane of these controls has a valid source and the masks are full, so the `old` operand is dead:
  // bound_ctrl lets the compiler drop its initialisation (two v_mov + a hazard nop per DPP pair)
  lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xF, 0xF, true);
  hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xF, 0xF, true);
  return __hiloint2double(hi, lo);
}
// v_permlane16_swap(x,x): [0] = rows (r0,r0,r2,r2), [1] = rows (r1,r1,r3,r3)
__device__ __forceinline__ void swap16(double x, double& even, double& odd) {
  const u2v lo = __builtin_amdgcn_permlane16_swap((unsigned)__double2loint(x), (unsigned)__double2loint(x), false, false);
  const u2v hi = __builtin_amdgcn_permlane16_swap((unsigned)__double2hiint(x), (unsigned)__double2hiint(x), false, false);
  even = __hiloint2double((int)hi[0], (int)lo[0]);
  odd = __hiloint2double((int)hi[1], (int)lo[1]);
}
// v_permlane32_swap(x,x): [0] = (lower half, lower half), [1] = (upper half, upper half)
__device__ __forceinline__ void swap32(double x, double& lower, double& upper) {
  const u2v lo = __builtin_amdgcn_permlane32_swap((unsigned)__double2loint(x), (unsigned)__double2loint(x), false, false);
  const u2v hi = __builtin_amdgcn_permlane32_swap((unsigned)__double2hiint(x), (unsigned)__double2hiint(x), false, false);
  lower = __hiloint2double((int)hi[0], (int)lo[0]);
  upper = __hiloint2double((int)hi[1], (int)lo[1]);
}
// value of 16-lane row G, same column, in all four rows
template <int G>
__device__ __forceinline__ double rowgroup_bcast(double x) {
  double ev, od, lo, up;
  swap16(x, ev, od);
  swap32((G & 1) ? od : ev, lo, up);
  return (G & 2) ? up : lo;
}

// ---- wave reductions (64 lanes): butterfly with quad_perm / row mirrors, then
// the two permlane swaps; every lane ends with the result -----------------------
struct OpMin { __device__ __forceinline__ double operator()(double a, double b) const { return fmin(a, b); } };
struct OpMax { __device__ __forceinline__ double operator()(double a, double b) const { return fmax(a, b); } };
struct OpSum { __device__ __forceinline__ double operator()(double a, double b) const { return a + b; } };
template <class Op>
__device__ __forceinline__ double wave_reduce(double v, Op op) {
  v = op(v, dpp_mov<0xB1>(v));    // quad_perm [1,0,3,2]
  v = op(v, dpp_mov<0x4E>(v));    // quad_perm [2,3,0,1]
  v = op(v, dpp_mov<0x141>(v));   // row_half_mirror
  v = op(v, dpp_mov<0x140>(v));   // row_mirror
  double a, b;
  swap16(v, a, b);
  v = op(a, b);
  swap32(v, a, b);
  return op(a, b);
}
__device__ __forceinline__ double wave_min(double v) { return wave_reduce(v, OpMin()); }
__device__ __forceinline__ double wave_max(double v) { return wave_reduce(v, OpMax()); }
__device__ __forceinline__ double wave_sum(double v) { return wave_reduce(v, OpSum()); }

// 1/x: v_rcp_f64 + two Newton steps (instead of the ~12-instruction IEEE sequence)
__device__ __forceinline__ double fast_rcp(double x) {
  double r = __builtin_amdgcn_rcp(x);
  r = fma(fma(-x, r, 1.0), r, r);
  r = fma(fma(-x, r, 1.0), r, r);
  return r;
}

// 1/sqrt(x): v_rsq_f64 + two Newton steps (replaces an IEEE sqrt followed by three IEEE divisions)
__device__ __forceinline__ double fast_rsqrt(double x) {
  double r = __builtin_amdgcn_rsq(x);
  r = r * fma(-0.5 * x * r, r, 1.5);
  r = r * fma(-0.5 * x * r, r, 1.5);
  return r;
}

// ---- cross-lane moves inside the MFMA fragment layout --------------------------
// lane = 16*g + c holds rows {g, 4+g, 8+g} of column c.
// value of lane (g, J) of the same 16-lane row (DPP row_newbcast, no LDS)
template <int J>
__device__ __forceinline__ double row_bcast(double x) {
  int lo = __double2loint(x), hi = __double2hiint(x);
  lo = __builtin_amdgcn_update_dpp(0, lo, 0x150 + J, 0xF, 0xF, true);
  hi = __builtin_amdgcn_update_dpp(0, hi, 0x150 + J, 0xF, 0xF, true);
  return __hiloint2double(hi, lo);
}
// wave-uniform copy of lane l's value (v_readlane -> SGPRs)
__device__ __forceinline__ double read_lane(double x, int l) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(x), l);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(x), l);
  return __hiloint2double(hi, lo);
}

// ---- quaternion helpers (QuaternionUtils.cpp:30-52) --------------------------
// G(q) (4x3): rows (-x,-y,-z), (s,-z,y), (z,s,-x), (-y,x,s)
__device__ __forceinline__ void quat_G(const double* q, double G[12]) {
  const double s = q[0], x = q[1], y = q[2], z = q[3];
  G[0] = -x; G[1] = -y;  G[2] = -z;
  G[3] = s;  G[4] = -z;  G[5] = y;
  G[6] = z;  G[7] = s;   G[8] = -x;
  G[9] = -y; G[10] = x;  G[11] = s;
}
// Omega(w) (4x4) = [[0,-w'],[w,-skew(w)]]  (AltroUtils.cpp:408-410, without the 0.5)
__device__ __forceinline__ void quat_Omega(const double* w, double O[16]) {
  const double x = w[0], y = w[1], z = w[2];
  O[0] = 0;  O[1] = -x; O[2] = -y;  O[3] = -z;
  O[4] = x;  O[5] = 0;  O[6] = z;   O[7] = -y;
  O[8] = y;  O[9] = -z; O[10] = 0;  O[11] = x;
  O[12] = z; O[13] = y; O[14] = -x; O[15] = 0;
}

// Model constants of one instance held in registers by the rollouts.
//   LEAN: the masked contact points c_l r_l (3 NL numbers); the torque is formed with cross products, as the
//         reference does (AltroUtils.cpp:376-391).  Used where registers are the occupancy limit (workspace
//         variants, 8 contact points).
//   else: Bw0 = Iinv skew(r_l) c_l (9 NL numbers): one fused dot product per angular axis, ~1 % faster when a
//         wave has the SIMD's registers to itself (variant 0).
template <int NL, bool LEAN>
struct ModelRegsT {
  double gb[3], wd0[3], rm[LEAN ? 3 * NL : 9 * NL];
  __device__ __forceinline__ void load(const double* cst, const double* bw0) {
    typedef Dim<NL> D;
#pragma unroll
    for (int i = 0; i < 3; ++i) { gb[i] = cst[D::C_GB + i]; wd0[i] = cst[D::C_WD0 + i]; }
    if (LEAN) {
#pragma unroll
      for (int i = 0; i < 3 * NL; ++i) rm[i] = cst[D::C_CON + i / 3] * cst[D::C_FOOT + i];
    } else {
#pragma unroll
      for (int i = 0; i < 9 * NL; ++i) rm[i] = bw0[i];
    }
  }
};

// Explicit-midpoint step of the quaternion SRBD (AltroUtils.cpp:9-22 applied to
// :363-392).  vdot and wdot do not depend on the state, so both midpoint
// evaluations share them.  x, xn: 13 doubles in registers.  Swing-point inputs are
// identically zero in every rollout (zero reference, zero gains), so the force sum needs no mask.
template <int NL, bool LEAN>
__device__ __forceinline__ void srbd_step(const DevParams& P, const ModelRegsT<NL, LEAN>& M, const double* x,
                                          const double* u, double* xn) {
  double F[3], vd[3], wd[3];
  {
    // independent partial sums: the wave has no other work to hide the add chains
    double f0[2] = {0, 0}, f1[2] = {0, 0}, f2[2] = {0, 0};
#pragma unroll
    for (int l = 0; l < NL; ++l) { f0[l & 1] += u[3 * l]; f1[l & 1] += u[3 * l + 1]; f2[l & 1] += u[3 * l + 2]; }
    F[0] = f0[0] + f0[1]; F[1] = f1[0] + f1[1]; F[2] = f2[0] + f2[1];
    if (LEAN) {
      double t0[2] = {0, 0}, t1[2] = {0, 0}, t2[2] = {0, 0};
#pragma unroll
      for (int l = 0; l < NL; ++l) {
        const double* r = &M.rm[3 * l];
        const double* f = &u[3 * l];
        t0[l & 1] += r[1] * f[2] - r[2] * f[1];
        t1[l & 1] += r[2] * f[0] - r[0] * f[2];
        t2[l & 1] += r[0] * f[1] - r[1] * f[0];
      }
      const double tau[3] = {t0[0] + t0[1], t1[0] + t1[1], t2[0] + t2[1]};
#pragma unroll
      for (int a = 0; a < 3; ++a)
        wd[a] = M.wd0[a] + (P.Iinv[3 * a] * tau[0] + P.Iinv[3 * a + 1] * tau[1] + P.Iinv[3 * a + 2] * tau[2]);
    } else {
#pragma unroll
      for (int a = 0; a < 3; ++a) {
        const double* b = &M.rm[3 * NL * a];
        double s[4] = {M.wd0[a], 0.0, 0.0, 0.0};
#pragma unroll
        for (int j = 0; j < 3 * NL; ++j) s[(j / 3) & 3] += b[j] * u[j];
        wd[a] = (s[0] + s[1]) + (s[2] + s[3]);
      }
    }
  }
#pragma unroll
  for (int a = 0; a < 3; ++a) vd[a] = F[a] * P.inv_mass + M.gb[a];
  // midpoint state
  double G[12];
  quat_G(&x[3], G);
  double qm[4], wm[3];
#pragma unroll
  for (int r = 0; r < 4; ++r)
    qm[r] = x[3 + r] + P.hh * (0.5 * (G[3 * r] * x[10] + G[3 * r + 1] * x[11] + G[3 * r + 2] * x[12]));
#pragma unroll
  for (int a = 0; a < 3; ++a) wm[a] = x[10 + a] + P.hh * wd[a];
  quat_G(qm, G);
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    xn[a] = x[a] + P.h * (x[7 + a] + P.hh * vd[a]);
    xn[7 + a] = x[7 + a] + P.h * vd[a];
    xn[10 + a] = x[10 + a] + P.h * wd[a];
  }
#pragma unroll
  for (int r = 0; r < 4; ++r)
    xn[3 + r] = x[3 + r] + P.h * (0.5 * (G[3 * r] * wm[0] + G[3 * r + 1] * wm[1] + G[3 * r + 2] * wm[2]));
}

// Reference state of knot k (QuatMpc.cpp:148-176), from refp = pos(3) vel(3) acc(3) quat_d(4)
__device__ __forceinline__ void xref_at(const DevParams& P, const double* refp, int k, double* xr) {
  const double t = (double)k * P.h_ref;
  const double h_ms = P.h_ref * 1000.0;
  xr[0] = refp[0] + refp[3] * k * h_ms / 1000.0 + 0.5 * refp[6] * t * t;
  xr[1] = refp[1] + refp[4] * k * h_ms / 1000.0 + 0.5 * refp[7] * t * t;
  xr[2] = refp[2] + 0.5 * refp[8] * t * t;
  xr[3] = refp[9]; xr[4] = refp[10]; xr[5] = refp[11]; xr[6] = refp[12];
  xr[7] = refp[3] + refp[6] * t; xr[8] = refp[4] + refp[7] * t; xr[9] = refp[5] + refp[8] * t;
  xr[10] = 0.0; xr[11] = 0.0; xr[12] = 0.0;
}

// Per-knot expansion, executed by ONE lane per knot (knots are independent):
//  - compact error-state Jacobian blocks AB[27] = {Aphiphi(9), Aphiw(9), W(9)}
//    of Abar = E(x+)' A E(x), Bbar = E(x+)' B   (AltroUtils.cpp:78-110,153-168):
//      Abar = [[I,0,hI,0],[0,Aphiphi,0,Aphiw],[0,0,I,0],[0,0,0,I]]
//      Bbar = [ (h^2/2m) c_i I ; (h/4) W (h Bw0) ; (h/m) c_i I ; h Bw0 ]
//  - cost gradient lx(12) and attitude Hessian block lxx(9) in error coordinates
//    (w (1 - |qref'q|) term)
template <int NL>
__device__ inline void expand_knot(const DevParams& P, const double* cst, const double* bw0,
                                   const double* refp, int k, const double* x, const double* u,
                                   const double* xn, double* AB, double* lx, double* lxx) {
  const int N = P.N;
  if (k < N) {
    double wd[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      double s = cst[Dim<NL>::C_WD0 + a];
      for (int j = 0; j < 3 * NL; ++j) s += bw0[3 * NL * a + j] * u[j];
      wd[a] = s;
    }
    double G0[12], Gm[12], Gn[12], O0[16], Om[16];
    quat_G(&x[3], G0);
    double qm[4], wm[3];
#pragma unroll
    for (int r = 0; r < 4; ++r)
      qm[r] = x[3 + r] + P.hh * (0.5 * (G0[3 * r] * x[10] + G0[3 * r + 1] * x[11] + G0[3 * r + 2] * x[12]));
#pragma unroll
    for (int a = 0; a < 3; ++a) wm[a] = x[10 + a] + P.hh * wd[a];
    quat_G(qm, Gm);
    quat_G(&xn[3], Gn);
    quat_Omega(&x[10], O0);
    quat_Omega(wm, Om);
    // Aqq = I + (h/2) Om (I + (h/4) O0)   (4x4)
    double M1[16], Aqq[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) M1[i] = ((i % 5 == 0) ? 1.0 : 0.0) + 0.5 * P.hh * O0[i];
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        double s = 0.0;
#pragma unroll
        for (int t = 0; t < 4; ++t) s += Om[4 * r + t] * M1[4 * t + c];
        Aqq[4 * r + c] = ((r == c) ? 1.0 : 0.0) + P.hh * s;
      }
    // Aqw = (h/2) (Om (h/4) G0 + Gm)   (4x3)
    double Aqw[12];
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        double s = 0.0;
#pragma unroll
        for (int t = 0; t < 4; ++t) s += Om[4 * r + t] * (0.5 * P.hh * G0[3 * t + c]);
        Aqw[3 * r + c] = P.hh * (s + Gm[3 * r + c]);
      }
    // Aphiphi = Gn' Aqq G0 ; Aphiw = Gn' Aqw ; W = Gn' Gm
    double AG[12];
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        double s = 0.0;
#pragma unroll
        for (int t = 0; t < 4; ++t) s += Aqq[4 * r + t] * G0[3 * t + c];
        AG[3 * r + c] = s;
      }
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        double s1 = 0.0, s2 = 0.0, s3 = 0.0;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          s1 += Gn[3 * t + r] * AG[3 * t + c];
          s2 += Gn[3 * t + r] * Aqw[3 * t + c];
          s3 += Gn[3 * t + r] * Gm[3 * t + c];
        }
        AB[3 * r + c] = s1;
        AB[9 + 3 * r + c] = s2;
        AB[18 + 3 * r + c] = s3;
      }
  }
  // ---- cost expansion at knot k (k = 0..N) ----
  double xr[13];
  xref_at(P, refp, k, xr);
  double lxf[13];
#pragma unroll
  for (int i = 0; i < 13; ++i) lxf[i] = P.Q[i] * (x[i] - xr[i]);
  const double dq = xr[3] * x[3] + xr[4] * x[4] + xr[5] * x[5] + xr[6] * x[6];
  const double sg = (dq >= 0.0) ? 1.0 : -1.0;
#pragma unroll
  for (int r = 0; r < 4; ++r) lxf[3 + r] += -sg * P.w * xr[3 + r];
  const double qh = -(x[3] * lxf[3] + x[4] * lxf[4] + x[5] * lxf[5] + x[6] * lxf[6]);
  double G[12];
  quat_G(&x[3], G);
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    lx[a] = lxf[a];
    lx[6 + a] = lxf[7 + a];
    lx[9 + a] = lxf[10 + a];
    lx[3 + a] = G[a] * lxf[3] + G[3 + a] * lxf[4] + G[6 + a] * lxf[5] + G[9 + a] * lxf[6];
  }
#pragma unroll
  for (int a = 0; a < 3; ++a)
#pragma unroll
    for (int b = 0; b < 3; ++b) {
      double s = (a == b) ? qh : 0.0;
#pragma unroll
      for (int t = 0; t < 4; ++t) s += G[3 * t + a] * P.Q[3 + t] * G[3 * t + b];
      lxx[3 * a + b] = s;
    }
}

// The same expansion split over FOUR lanes per knot (the wave executes a quarter of the instructions):
// part c = 0, 1, 2 produces column c of the three 3x3 blocks -- with A_qq = I + (h/2) Om (I + (h/4) O0),
//   Aphiphi[:,c] = Gn' (g + (h/2) Om (g + (h/4) O0 g)),  g = G0[:,c]
//   Aphiw[:,c]   = Gn' (h/2) (Om (h/4) g + Gm[:,c]),     W[:,c] = Gn' Gm[:,c]
// (Om v, O0 v are the quaternion-rate products with 12 nonzeros) -- and part 3 the cost expansion.
// Omega(w) v for a 4-vector v
__device__ __forceinline__ void omega_mul(const double* w, const double* v, double* o) {
  o[0] = -w[0] * v[1] - w[1] * v[2] - w[2] * v[3];
  o[1] = w[0] * v[0] + w[2] * v[2] - w[1] * v[3];
  o[2] = w[1] * v[0] - w[2] * v[1] + w[0] * v[3];
  o[3] = w[2] * v[0] + w[1] * v[1] - w[0] * v[2];
}
template <int NL>
__device__ inline void expand_knot_part(const DevParams& P, const double* cst, const double* bw0,
                                        const double* refp, int k, int part, const double* x, const double* u,
                                        const double* xn, double* ABk, double* XTk) {
  if (part < 3) {
    if (k >= P.N) return;
    const int c = part;
    double wd[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      double s = cst[Dim<NL>::C_WD0 + a];
      for (int j = 0; j < 3 * NL; ++j) s += bw0[3 * NL * a + j] * u[j];
      wd[a] = s;
    }
    double G0[12], Gm[12], Gn[12];
    quat_G(&x[3], G0);
    double qm[4], wm[3];
#pragma unroll
    for (int r = 0; r < 4; ++r)
      qm[r] = x[3 + r] + P.hh * (0.5 * (G0[3 * r] * x[10] + G0[3 * r + 1] * x[11] + G0[3 * r + 2] * x[12]));
#pragma unroll
    for (int a = 0; a < 3; ++a) wm[a] = x[10 + a] + P.hh * wd[a];
    quat_G(qm, Gm);
    quat_G(&xn[3], Gn);
    double g[4], gm[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {      // column c of G0 / Gm (c is lane-dependent: selects, no indexing)
      g[r] = (c == 0) ? G0[3 * r] : (c == 1 ? G0[3 * r + 1] : G0[3 * r + 2]);
      gm[r] = (c == 0) ? Gm[3 * r] : (c == 1 ? Gm[3 * r + 1] : Gm[3 * r + 2]);
    }
    double t0[4], t1[4], t2[4], ag[4], aw[4];
    omega_mul(&x[10], g, t0);                                        // O0 g
#pragma unroll
    for (int r = 0; r < 4; ++r) t1[r] = g[r] + (0.5 * P.hh) * t0[r];  // (I + (h/4) O0) g
    omega_mul(wm, t1, t2);                                           // Om (.)
    omega_mul(wm, g, t0);                                            // Om g
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      ag[r] = g[r] + P.hh * t2[r];
      aw[r] = P.hh * ((0.5 * P.hh) * t0[r] + gm[r]);
    }
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      const double s1 = Gn[r] * ag[0] + Gn[3 + r] * ag[1] + Gn[6 + r] * ag[2] + Gn[9 + r] * ag[3];
      const double s2 = Gn[r] * aw[0] + Gn[3 + r] * aw[1] + Gn[6 + r] * aw[2] + Gn[9 + r] * aw[3];
      const double s3 = Gn[r] * gm[0] + Gn[3 + r] * gm[1] + Gn[6 + r] * gm[2] + Gn[9 + r] * gm[3];
      ABk[3 * r + c] = s1;
      ABk[9 + 3 * r + c] = s2;
      ABk[18 + 3 * r + c] = s3;
    }
    return;
  }
  // ---- part 3: cost expansion at knot k (k = 0..N), as in expand_knot ----
  double xr[13];
  xref_at(P, refp, k, xr);
  double lxf[13];
#pragma unroll
  for (int i = 0; i < 13; ++i) lxf[i] = P.Q[i] * (x[i] - xr[i]);
  const double dq = xr[3] * x[3] + xr[4] * x[4] + xr[5] * x[5] + xr[6] * x[6];
  const double sg = (dq >= 0.0) ? 1.0 : -1.0;
#pragma unroll
  for (int r = 0; r < 4; ++r) lxf[3 + r] += -sg * P.w * xr[3 + r];
  const double qh = -(x[3] * lxf[3] + x[4] * lxf[4] + x[5] * lxf[5] + x[6] * lxf[6]);
  double G[12];
  quat_G(&x[3], G);
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    XTk[9 + a] = lxf[a];
    XTk[9 + 6 + a] = lxf[7 + a];
    XTk[9 + 9 + a] = lxf[10 + a];
    XTk[9 + 3 + a] = G[a] * lxf[3] + G[3 + a] * lxf[4] + G[6 + a] * lxf[5] + G[9 + a] * lxf[6];
  }
#pragma unroll
  for (int a = 0; a < 3; ++a)
#pragma unroll
    for (int b = 0; b < 3; ++b) {
      double s = (a == b) ? qh : 0.0;
#pragma unroll
      for (int t = 0; t < 4; ++t) s += G[3 * t + a] * P.Q[3 + t] * G[3 * t + b];
      XTk[3 * a + b] = s;
    }
}

// Element (r,c) of the dense 12x12 Abar from the compact blocks.
__device__ __forceinline__ double abar_elem(const DevParams& P, const double* AB, int r, int c) {
  if (r < 3) return ((r == c) ? 1.0 : 0.0) + ((c == r + 6) ? P.h : 0.0);
  if (r < 6) {
    if (c >= 3 && c < 6) return AB[3 * (r - 3) + (c - 3)];
    if (c >= 9) return AB[9 + 3 * (r - 3) + (c - 9)];
    return 0.0;
  }
  return (r == c) ? 1.0 : 0.0;
}
// Element (r, 3l+a) of the dense 12 x 3NL Bbar (unrotated); bw0 is 3 x 3NL row-major.
template <int NL>
__device__ __forceinline__ double bbar_elem(const DevParams& P, const double* cst, const double* bw0,
                                            const double* AB, int r, int col) {
  constexpr int NU = 3 * NL;
  const int l = col / 3, a = col - 3 * l;
  const double cl = cst[Dim<NL>::C_CON + l];
  if (r < 3) return (r == a) ? cl * (P.h * (P.hh * (1.0 / P.mass))) : 0.0;
  if (r < 6) {
    const double* W = AB + 18 + 3 * (r - 3);
    return (0.5 * P.hh) * (W[0] * (P.h * bw0[col]) + W[1] * (P.h * bw0[NU + col]) +
                            W[2] * (P.h * bw0[2 * NU + col]));
  }
  if (r < 9) return (r - 6 == a) ? cl * (P.h * (1.0 / P.mass)) : 0.0;
  return P.h * bw0[NU * (r - 9) + col];
}


// =============================================================================
// Model policies.  The solver core (Riccati backward pass in MFMA fragments,
// stage solve, rollouts, interior-point bookkeeping) is shared; a policy supplies
// what differs between the reference's two controllers:
//   QuatModel   -- legged::QuatMpc   (QuatMpc.cpp:109-276, quaternion SRBD, error state)
//   ConvexModel -- legged::ConvexMpc (ConvexMpc.cpp:81-198, Euler-angle SRBD)
// Both keep a 13-double state slot per knot in LDS (the convex model uses 12).
// =============================================================================

// per-lane cost-Hessian pattern of the backward pass: Qxx[e] += qadd[e] + XT[xoff[e]]
struct CostPattern {
  double qadd[3];
  int xoff[3];
};

template <int NL_>
struct QuatModelT {
  static constexpr int NX = 13;
  static constexpr int NL = NL_;
  static constexpr int EXPAND_PARTS = 4;     // lanes per knot in the expansions
  typedef Dim<NL_> D;
  template <bool LEAN> using RegsT = ModelRegsT<NL_, LEAN>;

  template <bool LEAN>
  static __device__ __forceinline__ void step(const DevParams& P, const RegsT<LEAN>& M, const double* x,
                                              const double* u, double* xn) {
    srbd_step<NL_, LEAN>(P, M, x, u, xn);
  }
  // dx = xc (-) xo : inverse Cayley map of xo.q^-1 * xc.q (QuaternionUtils.cpp:16-18)
  static __device__ __forceinline__ void state_diff(const double* xo, const double* xc, double* dx) {
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      dx[a] = xc[a] - xo[a];
      dx[6 + a] = xc[7 + a] - xo[7 + a];
      dx[9 + a] = xc[10 + a] - xo[10 + a];
    }
    double G[12];
    quat_G(&xo[3], G);
    const double isc = fast_rcp(xo[3] * xc[3] + xo[4] * xc[4] + xo[5] * xc[5] + xo[6] * xc[6]);
#pragma unroll
    for (int a = 0; a < 3; ++a)
      dx[3 + a] = (G[a] * xc[3] + G[3 + a] * xc[4] + G[6 + a] * xc[5] + G[9 + a] * xc[6]) * isc;
  }
  static __device__ __forceinline__ void expand(const DevParams& P, const double* cst, const double* bw0,
                                                const double* refp, int k, const double* x, const double* u,
                                                const double* xn, double* AB, double* lx, double* lxx) {
    expand_knot<NL_>(P, cst, bw0, refp, k, x, u, xn, AB, lx, lxx);
  }
  static __device__ __forceinline__ void expand_part(const DevParams& P, const double* cst, const double* bw0,
                                                     const double* refp, int k, int part, const double* x,
                                                     const double* u, const double* xn, double* ABk, double* XTk) {
    expand_knot_part<NL_>(P, cst, bw0, refp, k, part, x, u, xn, ABk, XTk);
  }
  // un-augmented objective of knot k (u == nullptr at the terminal knot)
  static __device__ __forceinline__ double knot_cost(const DevParams& P, const double* refp, const double* uref,
                                                     int k, const double* x, const double* u) {
    double xr[13];
    xref_at(P, refp, k, xr);
    double J = 0.0;
    for (int i = 0; i < 13; ++i) { const double e = x[i] - xr[i]; J += 0.5 * P.Q[i] * e * e; }
    const double dq = xr[3] * x[3] + xr[4] * x[4] + xr[5] * x[5] + xr[6] * x[6];
    J += P.w * (1.0 - fabs(dq));
    if (u)
      for (int j = 0; j < D::NU; ++j) { const double e = u[j] - uref[j]; J += 0.5 * P.R[j % 12] * e * e; }
    return J;
  }

  // operand patterns of the backward pass for fragment rows r_e = 4e + g, column c of input tile t.
  // LEANOPS (workspace variants, two waves per SIMD, 256 registers): the nine h Bw0 entries of the lane's contact
  // point are re-read from LDS for every attitude row instead of living in 18 registers for the whole pass.
  template <bool LEANOPS>
  struct OperandsT {
    double Ac[3], Bc[D::TU][3][3], hbw[LEANOPS ? 1 : D::TU][LEANOPS ? 1 : 3][LEANOPS ? 1 : 3];
    const double* bwl;          // LEANOPS: &bw0[3 * lc]
    int aoff[3];
    bool phi[3];
    int g;
    __device__ __forceinline__ void init(const DevParams& P, const double* cst, const double* bw0, int lane,
                                         CostPattern& cp) {
      const int c = lane & 15;
      g = lane >> 4;
      const bool cval = c < 12;
      const int lc = cval ? c / 3 : 0;
      bwl = bw0 + 3 * lc;
      if (!LEANOPS) {
#pragma unroll
        for (int t = 0; t < D::TU; ++t)
#pragma unroll
          for (int j = 0; j < 3; ++j)
#pragma unroll
            for (int a = 0; a < 3; ++a) hbw[LEANOPS ? 0 : t][LEANOPS ? 0 : j][LEANOPS ? 0 : a] = cval ? P.h * bw0[D::NU * j + 12 * t + 3 * lc + a] : 0.0;
      }
#pragma unroll
      for (int e = 0; e < 3; ++e) {
        const int r = 4 * e + g;
        phi[e] = (r >= 3 && r < 6);
        Ac[e] = cval ? ((((r == c) && !phi[e]) ? 1.0 : 0.0) + ((r < 3 && c == r + 6) ? P.h : 0.0)) : 0.0;
        aoff[e] = -1;
        if (phi[e] && c >= 3 && c < 6) aoff[e] = 3 * (r - 3) + (c - 3);
        if (phi[e] && c >= 9 && c < 12) aoff[e] = 9 + 3 * (r - 3) + (c - 9);
        cp.qadd[e] = (cval && r == c && !phi[e]) ? P.Q[(r < 3) ? r : r + 1] : 0.0;
        cp.xoff[e] = -1;
        if (phi[e] && c >= 3 && c < 6) cp.xoff[e] = 3 * (r - 3) + (c - 3);
        if (c == 12) cp.xoff[e] = 9 + r;
#pragma unroll
        for (int t = 0; t < D::TU; ++t) {
          const double conl = cval ? cst[D::C_CON + 4 * t + lc] : 0.0;
#pragma unroll
          for (int a = 0; a < 3; ++a) {
            double v = 0.0;
            if (cval) {
              if (r < 3) v = (r == a) ? conl * (P.h * (P.hh * (1.0 / P.mass))) : 0.0;
              else if (r >= 6 && r < 9) v = (r - 6 == a) ? conl * (P.h * (1.0 / P.mass)) : 0.0;
              else if (r >= 9) v = P.h * bw0[D::NU * (r - 9) + 12 * t + 3 * lc + a];
            }
            Bc[t][e][a] = v;
          }
        }
      }
    }
    // Abar and rotated Bbar*T (tc[t] = column bc of the frame T of the lane's leg in tile t) straight into
    // fragments, one fragment row e at a time.  Branch-free (unconditional loads at clamped offsets, then
    // selects): the rows are scheduled between the MFMAs of the backward pass.
    // first half of row e: A entry and the attitude-row candidates of the unrotated B entries
    __device__ __forceinline__ void row_a(const DevParams& P, const double* ABk, int e, double Afo[3],
                                          double pr[][3]) const {
      const double av = ABk[(aoff[e] >= 0) ? aoff[e] : 0];
      Afo[e] = (aoff[e] >= 0) ? av : Ac[e];
      const double* W = ABk + 18 + (phi[e] ? 3 * (4 * e + g - 3) : 0);
      const double w0 = W[0], w1 = W[1], w2 = W[2];
#pragma unroll
      for (int t = 0; t < D::TU; ++t) {
        if (LEANOPS) {      // the lanes without a column read contact point 0; their frame column tc is zero
          const double* b = bwl + 12 * t;
#pragma unroll
          for (int a = 0; a < 3; ++a)
            pr[t][a] = (0.5 * P.hh) * (w0 * (P.h * b[a]) + w1 * (P.h * b[D::NU + a]) + w2 * (P.h * b[2 * D::NU + a]));
        } else {
          pr[t][0] = (0.5 * P.hh) * (w0 * hbw[t][0][0] + w1 * hbw[t][1][0] + w2 * hbw[t][2][0]);
          pr[t][1] = (0.5 * P.hh) * (w0 * hbw[t][0][1] + w1 * hbw[t][1][1] + w2 * hbw[t][2][1]);
          pr[t][2] = (0.5 * P.hh) * (w0 * hbw[t][0][2] + w1 * hbw[t][1][2] + w2 * hbw[t][2][2]);
        }
      }
    }
    // second half: select and rotate into the frame column
    __device__ __forceinline__ void row_b(const double tc[][3], int e, const double pr[][3], double Bfo[][3]) const {
#pragma unroll
      for (int t = 0; t < D::TU; ++t) {
        const double b0 = phi[e] ? pr[t][0] : Bc[t][e][0], b1 = phi[e] ? pr[t][1] : Bc[t][e][1],
                     b2 = phi[e] ? pr[t][2] : Bc[t][e][2];
        Bfo[t][e] = b0 * tc[t][0] + b1 * tc[t][1] + b2 * tc[t][2];
      }
    }
    __device__ __forceinline__ void build_row(const DevParams& P, const double* ABk, const double tc[][3], int e,
                                              double Afo[3], double Bfo[][3]) const {
      double pr[D::TU][3];
      row_a(P, ABk, e, Afo, pr);
      row_b(tc, e, pr, Bfo);
    }
    __device__ __forceinline__ void build(const DevParams& P, const double* ABk, const double tc[][3],
                                          double Afo[3], double Bfo[][3]) const {
#pragma unroll
      for (int e = 0; e < 3; ++e) build_row(P, ABk, tc, e, Afo, Bfo);
    }
  };
  typedef OperandsT<false> Operands;

  static __device__ __forceinline__ double a_elem(const DevParams& P, const double* cst, const double* bw0,
                                                  const double* AB, int r, int c) {
    return abar_elem(P, AB, r, c);
  }
  static __device__ __forceinline__ double b_elem(const DevParams& P, const double* cst, const double* bw0,
                                                  const double* AB, int r, int c) {
    return bbar_elem<NL_>(P, cst, bw0, AB, r, c);
  }
};
typedef QuatModelT<4> QuatModel;
typedef QuatModelT<8> Quat8Model;   // BASELINE config 5: the same problem with 8 contact points (synthetic biped)

// ---- legged::ConvexMpc's model ------------------------------------------------
// state x = [roll pitch yaw, pos(3), ang_vel_world(3), lin_vel_world(3)], inputs =
// world-frame foot forces.  Continuous dynamics (AltroUtils.cpp:224-293):
//   d(rpy) = Rz(yaw)' w,  d(pos) = v,  d(w) = Iw(yaw)^-1 sum_l r_l x u_l,  d(v) = sum u_l/m + g
// with Iw(yaw)^-1 = Rz diag(1/I) Rz'.  Its Jacobian (AltroUtils.cpp:295-359) omits
// d(Iw^-1)/d(yaw); the midpoint chain rule (AltroUtils.cpp:78-110) then gives
//   A = I + h Am + (h h/2) jm e8',   B = h ((h/2) Am B0 + Bm)
// whose only state-dependent entries are the per-knot record CV_* below.
enum { CV_JM0 = 0, CV_JM1, CV_CM, CV_SM, CV_M00, CV_M01, CV_M10, CV_M11, CV_W00, CV_W01, CV_W11, CV_COUNT };
// reference record refp[]: yaw0, yaw_rate_d, pos_d(3), vx_d, vy_d
enum { CR_YAW = 0, CR_RATE, CR_POS, CR_VX = 5, CR_VY = 6 };

struct ConvexRegs {
  double con[4], sk[36];      // contacts; masked skew rows: tau[a] = sum_j sk[12a+j] u[j]
  __device__ __forceinline__ void load(const double* cst, const double* bw0) {
#pragma unroll
    for (int i = 0; i < 4; ++i) con[i] = cst[C_CON + i];
#pragma unroll
    for (int i = 0; i < 36; ++i) sk[i] = bw0[i];
  }
};

struct ConvexModel {
  static constexpr int NX = 12;
  static constexpr int NL = 4;
  static constexpr int EXPAND_PARTS = 1;     // one lane per knot (two sincos dominate; nothing to split)
  typedef Dim<4> D;
  typedef ConvexRegs Regs;
  template <bool LEAN> using RegsT = ConvexRegs;

  // Iw(yaw)^-1 tau with c = cos(yaw), s = sin(yaw)
  static __device__ __forceinline__ void winv(const DevParams& P, double c, double s, double& w00, double& w01,
                                              double& w11) {
    const double a = P.Iinv[0], b = P.Iinv[4];
    w00 = c * c * a + s * s * b;
    w01 = c * s * (a - b);
    w11 = s * s * a + c * c * b;
  }
  static __device__ __forceinline__ void torque_force(const Regs& M, const double* u, double* tau, double* F) {
    F[0] = F[1] = F[2] = 0.0;
#pragma unroll
    for (int l = 0; l < 4; ++l) {
      const double c = M.con[l];
      F[0] += c * u[3 * l]; F[1] += c * u[3 * l + 1]; F[2] += c * u[3 * l + 2];
    }
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      const double* b = &M.sk[12 * a];
      const double s0 = b[0] * u[0] + b[1] * u[1] + b[2] * u[2];
      const double s1 = b[3] * u[3] + b[4] * u[4] + b[5] * u[5];
      const double s2 = b[6] * u[6] + b[7] * u[7] + b[8] * u[8];
      const double s3 = b[9] * u[9] + b[10] * u[10] + b[11] * u[11];
      tau[a] = (s0 + s1) + (s2 + s3);
    }
  }
  // explicit midpoint (AltroUtils.cpp:9-22) of ct_srb_dynamics
  template <bool LEAN>
  static __device__ __forceinline__ void step(const DevParams& P, const Regs& M, const double* x,
                                              const double* u, double* xn) {
    double tau[3], F[3];
    torque_force(M, u, tau, F);
    double s0, c0;
    sincos(x[2], &s0, &c0);
    double w00, w01, w11;
    winv(P, c0, s0, w00, w01, w11);
    const double vd[3] = {F[0] * P.inv_mass, F[1] * P.inv_mass, F[2] * P.inv_mass - 9.81};
    const double wd0[3] = {w00 * tau[0] + w01 * tau[1], w01 * tau[0] + w11 * tau[1], P.Iinv[8] * tau[2]};
    const double yawm = x[2] + P.hh * x[8];
    const double wm[3] = {x[6] + P.hh * wd0[0], x[7] + P.hh * wd0[1], x[8] + P.hh * wd0[2]};
    double sm_, cm_;
    sincos(yawm, &sm_, &cm_);
    winv(P, cm_, sm_, w00, w01, w11);
    const double wdm[3] = {w00 * tau[0] + w01 * tau[1], w01 * tau[0] + w11 * tau[1], P.Iinv[8] * tau[2]};
    xn[0] = x[0] + P.h * (cm_ * wm[0] + sm_ * wm[1]);
    xn[1] = x[1] + P.h * (-sm_ * wm[0] + cm_ * wm[1]);
    xn[2] = x[2] + P.h * wm[2];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      xn[3 + a] = x[3 + a] + P.h * (x[9 + a] + P.hh * vd[a]);
      xn[6 + a] = x[6 + a] + P.h * wdm[a];
      xn[9 + a] = x[9 + a] + P.h * vd[a];
    }
    xn[12] = 0.0;
  }
  static __device__ __forceinline__ void state_diff(const double* xo, const double* xc, double* dx) {
#pragma unroll
    for (int i = 0; i < 12; ++i) dx[i] = xc[i] - xo[i];
  }
  // reference state of knot k (ConvexMpc.cpp:95-106)
  static __device__ __forceinline__ void xref(const DevParams& P, const double* refp, int k, double* xr) {
    const double h_ms = P.h_ref * 1000.0;
#pragma unroll
    for (int i = 0; i < 12; ++i) xr[i] = 0.0;
    xr[2] = refp[CR_YAW] + refp[CR_RATE] * h_ms / 1000.0 * k;
    xr[3] = refp[CR_POS]; xr[4] = refp[CR_POS + 1]; xr[5] = refp[CR_POS + 2];
    xr[8] = refp[CR_RATE];
    xr[9] = refp[CR_VX]; xr[10] = refp[CR_VY];
  }
  static __device__ __forceinline__ void expand_part(const DevParams&, const double*, const double*, const double*,
                                                     int, int, const double*, const double*, const double*,
                                                     double*, double*) {}
  static __device__ inline void expand(const DevParams& P, const double* cst, const double* bw0,
                                       const double* refp, int k, const double* x, const double* u,
                                       const double* xn, double* AB, double* lx, double* lxx) {
    if (k < P.N) {
      double tau[3];
#pragma unroll
      for (int a = 0; a < 3; ++a) {
        double s = 0.0;
        for (int j = 0; j < 12; ++j) s += bw0[12 * a + j] * u[j];
        tau[a] = s;
      }
      double s0, c0;
      sincos(x[2], &s0, &c0);
      double w00, w01, w11;
      winv(P, c0, s0, w00, w01, w11);
      const double wm0 = x[6] + P.hh * (w00 * tau[0] + w01 * tau[1]);
      const double wm1 = x[7] + P.hh * (w01 * tau[0] + w11 * tau[1]);
      double sm_, cm_;
      sincos(x[2] + P.hh * x[8], &sm_, &cm_);
      AB[CV_JM0] = wm1 * cm_ - wm0 * sm_;       // AltroUtils.cpp:354 at the midpoint
      AB[CV_JM1] = -wm0 * cm_ - wm1 * sm_;      // :355
      AB[CV_CM] = cm_;
      AB[CV_SM] = sm_;
      // M1 = Rz(yaw_m)' Iw(yaw)^-1  (upper-left 2x2; the rest is (0,0,1/Izz))
      AB[CV_M00] = cm_ * w00 + sm_ * w01;
      AB[CV_M01] = cm_ * w01 + sm_ * w11;
      AB[CV_M10] = -sm_ * w00 + cm_ * w01;
      AB[CV_M11] = -sm_ * w01 + cm_ * w11;
      winv(P, cm_, sm_, w00, w01, w11);
      AB[CV_W00] = w00; AB[CV_W01] = w01; AB[CV_W11] = w11;
#pragma unroll
      for (int i = CV_COUNT; i < kAB; ++i) AB[i] = 0.0;
    }
    double xr[12];
    xref(P, refp, k, xr);
#pragma unroll
    for (int i = 0; i < 12; ++i) lx[i] = P.Q[i] * (x[i] - xr[i]);
#pragma unroll
    for (int i = 0; i < 9; ++i) lxx[i] = 0.0;
  }
  static __device__ __forceinline__ double knot_cost(const DevParams& P, const double* refp, const double* uref,
                                                     int k, const double* x, const double* u) {
    double xr[12];
    xref(P, refp, k, xr);
    double J = 0.0;
    for (int i = 0; i < 12; ++i) { const double e = x[i] - xr[i]; J += 0.5 * P.Q[i] * e * e; }
    if (u)
      for (int j = 0; j < 12; ++j) { const double e = u[j] - uref[j]; J += 0.5 * P.R[j] * e * e; }
    return J;
  }

  // entries of A that vary per knot: (row, col, record slot, scale)
  struct Operands;
  template <bool LEANOPS> using OperandsT = Operands;
  struct Operands {
    double Ac[3], asc[3], Bc[3][3], sk[3][3];
    double m0s[3], m1s[3], mz[3];      // row of the 3x3 factor: m0s*AB[mo0] , m1s*AB[mo1], mz
    int aoff[3], mo0[3], mo1[3];
    bool var[3];
    __device__ __forceinline__ void init(const DevParams& P, const double* cst, const double* bw0, int lane,
                                         CostPattern& cp) {
      const int c = lane & 15, g = lane >> 4;
      const bool cval = c < 12;
      const int lc = cval ? c / 3 : 0;
      const double conl = cval ? cst[C_CON + lc] : 0.0;
      const double hhh = P.h * P.hh;
#pragma unroll
      for (int j = 0; j < 3; ++j)
#pragma unroll
        for (int a = 0; a < 3; ++a) sk[j][a] = cval ? bw0[12 * j + 3 * lc + a] : 0.0;
#pragma unroll
      for (int e = 0; e < 3; ++e) {
        const int r = 4 * e + g;
        double ac = 0.0;
        if (cval) {
          if (r == c) ac = 1.0;
          if (r >= 3 && r < 6 && c == r + 6) ac = P.h;
          if (r == 2 && c == 8) ac = P.h;
        }
        Ac[e] = ac;
        aoff[e] = -1; asc[e] = 0.0;
        if (r == 0 && c == 2) { aoff[e] = CV_JM0; asc[e] = P.h; }
        if (r == 1 && c == 2) { aoff[e] = CV_JM1; asc[e] = P.h; }
        if (r == 0 && c == 6) { aoff[e] = CV_CM; asc[e] = P.h; }
        if (r == 0 && c == 7) { aoff[e] = CV_SM; asc[e] = P.h; }
        if (r == 1 && c == 6) { aoff[e] = CV_SM; asc[e] = -P.h; }
        if (r == 1 && c == 7) { aoff[e] = CV_CM; asc[e] = P.h; }
        if (r == 0 && c == 8) { aoff[e] = CV_JM0; asc[e] = hhh; }
        if (r == 1 && c == 8) { aoff[e] = CV_JM1; asc[e] = hhh; }
        cp.qadd[e] = (cval && r == c) ? P.Q[r] : 0.0;
        cp.xoff[e] = (c == 12) ? 9 + r : -1;
        // B rows: 0..2 = h h/2 (Rz_m' Iw0^-1) S, 3..5 = h h/2 con/m, 6..8 = h Iw_m^-1 S, 9..11 = h con/m
        var[e] = cval && (r < 3 || (r >= 6 && r < 9));
        mo0[e] = 0; mo1[e] = 0; m0s[e] = 0.0; m1s[e] = 0.0; mz[e] = 0.0;
        if (r == 0) { mo0[e] = CV_M00; mo1[e] = CV_M01; m0s[e] = hhh; m1s[e] = hhh; }
        if (r == 1) { mo0[e] = CV_M10; mo1[e] = CV_M11; m0s[e] = hhh; m1s[e] = hhh; }
        if (r == 2) mz[e] = hhh * P.Iinv[8];
        if (r == 6) { mo0[e] = CV_W00; mo1[e] = CV_W01; m0s[e] = P.h; m1s[e] = P.h; }
        if (r == 7) { mo0[e] = CV_W01; mo1[e] = CV_W11; m0s[e] = P.h; m1s[e] = P.h; }
        if (r == 8) mz[e] = P.h * P.Iinv[8];
#pragma unroll
        for (int a = 0; a < 3; ++a) {
          double v = 0.0;
          if (cval) {
            if (r >= 3 && r < 6) v = (r - 3 == a) ? conl * (hhh * (1.0 / P.mass)) : 0.0;
            else if (r >= 9) v = (r - 9 == a) ? conl * (P.h * (1.0 / P.mass)) : 0.0;
          }
          Bc[e][a] = v;
        }
      }
    }
    __device__ __forceinline__ void row_a(const DevParams& P, const double* ABk, int e, double Afo[3],
                                          double pr[][3]) const {
      const double av = asc[e] * ABk[(aoff[e] >= 0) ? aoff[e] : 0];
      Afo[e] = (aoff[e] >= 0) ? av : Ac[e];
      const double m0 = m0s[e] * ABk[mo0[e]], m1 = m1s[e] * ABk[mo1[e]], m2 = mz[e];
      pr[0][0] = m0 * sk[0][0] + m1 * sk[1][0] + m2 * sk[2][0];
      pr[0][1] = m0 * sk[0][1] + m1 * sk[1][1] + m2 * sk[2][1];
      pr[0][2] = m0 * sk[0][2] + m1 * sk[1][2] + m2 * sk[2][2];
    }
    __device__ __forceinline__ void row_b(const double tc[][3], int e, const double pr[][3], double Bfo[][3]) const {
      const double b0 = var[e] ? pr[0][0] : Bc[e][0], b1 = var[e] ? pr[0][1] : Bc[e][1], b2 = var[e] ? pr[0][2] : Bc[e][2];
      Bfo[0][e] = b0 * tc[0][0] + b1 * tc[0][1] + b2 * tc[0][2];
    }
    __device__ __forceinline__ void build_row(const DevParams& P, const double* ABk, const double tc[][3], int e,
                                              double Afo[3], double Bfo[][3]) const {
      double pr[1][3];
      row_a(P, ABk, e, Afo, pr);
      row_b(tc, e, pr, Bfo);
    }
    __device__ __forceinline__ void build(const DevParams& P, const double* ABk, const double tc[][3],
                                          double Afo[3], double Bfo[][3]) const {
#pragma unroll
      for (int e = 0; e < 3; ++e) build_row(P, ABk, tc, e, Afo, Bfo);
    }
  };

  // dense discrete Jacobians (qmpc_convex_linearize)
  static __device__ __forceinline__ double a_elem(const DevParams& P, const double* cst, const double* bw0,
                                                  const double* AB, int r, int c) {
    const double hhh = P.h * P.hh;
    double v = (r == c) ? 1.0 : 0.0;
    if (r >= 3 && r < 6 && c == r + 6) v = P.h;
    if (r == 2 && c == 8) v = P.h;
    if (r == 0 && c == 2) v = P.h * AB[CV_JM0];
    if (r == 1 && c == 2) v = P.h * AB[CV_JM1];
    if (r == 0 && c == 6) v = P.h * AB[CV_CM];
    if (r == 0 && c == 7) v = P.h * AB[CV_SM];
    if (r == 1 && c == 6) v = -P.h * AB[CV_SM];
    if (r == 1 && c == 7) v = P.h * AB[CV_CM];
    if (r == 0 && c == 8) v = hhh * AB[CV_JM0];
    if (r == 1 && c == 8) v = hhh * AB[CV_JM1];
    return v;
  }
  static __device__ __forceinline__ double b_elem(const DevParams& P, const double* cst, const double* bw0,
                                                  const double* AB, int r, int col) {
    const int l = col / 3, a = col - 3 * l;
    const double hhh = P.h * P.hh;
    const double cl = cst[C_CON + l];
    const double s0 = bw0[col], s1 = bw0[12 + col], s2 = bw0[24 + col];
    if (r == 0) return hhh * AB[CV_M00] * s0 + hhh * AB[CV_M01] * s1;
    if (r == 1) return hhh * AB[CV_M10] * s0 + hhh * AB[CV_M11] * s1;
    if (r == 2) return hhh * P.Iinv[8] * s2;
    if (r < 6) return (r - 3 == a) ? cl * (hhh * (1.0 / P.mass)) : 0.0;
    if (r == 6) return P.h * AB[CV_W00] * s0 + P.h * AB[CV_W01] * s1;
    if (r == 7) return P.h * AB[CV_W01] * s0 + P.h * AB[CV_W11] * s1;
    if (r == 8) return P.h * P.Iinv[8] * s2;
    return (r - 9 == a) ? cl * (P.h * (1.0 / P.mass)) : 0.0;
  }
};

}  // namespace qmpc
