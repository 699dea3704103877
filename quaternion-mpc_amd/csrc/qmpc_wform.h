// qmpc_wform.h -- the wave-per-instance solve with the Newton
// system of a knot eliminated in the WRENCH form (round 4).  Same problem, same interior-point iteration and the same
// control flow as qmpc_solve_body.inc (reference: QuatMpc.cpp:109-276 poses the problem, AltroUtils.cpp:363-439 /
// :9-22,78-110 are the dynamics and their midpoint linearisation); what changes is the algebra of one Riccati step and
// of the closed-loop rollout, carried over from the lane-per-instance core (qmpc_lane_core.h) into the MFMA fragment
// layout of this kernel family.
//
// The 12 inputs of a knot act on the state only through the 6-dimensional wrench: Bbar_k = M_k Wr with
// Wr = [c_l I ; Bw0_l] (6 x 12, the same at every knot) and M_k = [[m1 I, 0], [0, Wt_k], [m2 I, 0], [0, h I]] (12 x 6),
// and the input Hessian is block diagonal apart from that coupling: Quu = D + Wr' S6 Wr, D = blkdiag(D_l),
// S6 = M'PM.  With V_l = [T_l ; Bw0_l T_l] (T_l the rotated frame of DESIGN.md section 2), G = sum_l V_l D_l^-1 V_l',
// r6 = sum_l V_l D_l^-1 g_l, Yp = M'[P | p] (6 x 13) and e = Abar dx the predicted state error, the wrench increment
// dw = sum_l V_l du_l solves
//        (I + G S6) [Xw | xw] = -(G Yp + [0 | r6])            dw = Xw e + xw,
// the costate of the contact points is  zeta = Yp_fb e + y0 + S6 dw = Xz e + xz  with  [Xz | xz] = Yp + S6 [Xw | xw],
// the inputs follow per contact point,  du_l = -D_l^-1 (V_l' zeta + g_l),  and the cost-to-go is
//        [P | p]_k = [lxx | lx] + Abar' ([P | p] + Yp_fb' [Xw | xw]) Abar_aug.
// The 6 x 6 system is solved in its symmetric positive definite form  W' = S6 (I + G S6) = S6 + S6 G S6,
// W' [Xw | xw] = -S6 (G Yp + [0 | r6]) =: -Q,  and  S6 G S6 = Q_fb M  is a column operation on Q.  Its right-hand side
// vanishes with the step (the bracket is the wrench-space residual), so the conditioning of W' (~1e7) perturbs the
// Newton direction by 1e-9 of its size and the fixed point not at all; the 1e14 : 1e-6 weight ratios of the cone rows
// stay inside the 3 x 3 blocks D_l, factorised per contact point in their rotated frames as before.
//
// What it buys at one wave per SIMD (the B = 1024 contract workload): per knot 13 FP64 MFMAs instead of 18 (the
// products against Abar and M that only mix COLUMNS are DPP column operations on the fragments, no matrix
// instruction), SIX Gauss-Jordan pivots on 6-row fragments instead of 6 (trot) / 12 (four stance legs) on 12-row
// ones, and the four 3 x 3 blocks leave the sequential recursion altogether (one lane per knot and contact point,
// before the pass).  The closed-loop rollout carries the 6-dimensional wrench instead of the 12 inputs: lane i < 6 owns
// a row of Xw, lanes 6..11 a row of Xz (their costates are parked for the input recovery, which runs lane-parallel
// over (knot, contact point) after the rollout), six values are broadcast per knot instead of twelve.
//
// Included after qmpc_kernels.hip by the translation units that use it: qmpc_wform.hip (the solve kernel) and
// qmpc_loop_fused.hip (the closed loop's persistent kernel and the warm-started solve); qmpc_wform_body.inc is the
// body of one solve, the counterpart of qmpc_solve_body.inc.
#pragma once

namespace qmpc {


// ---- LDS layout: that of the all-LDS variant, the set-up scratch aliased onto the slack array, plus per knot the
// wrench-space blocks G (21, symmetric) and r6 (6) of the contact points and the wrench Wr u_k (6) ------------------
struct LayoutW {
  int GK, WR, SLW;      // SLW: where the slack block [S | LAM | RC] starts inside the workspace slice (sl_global layouts)
};
constexpr int kGK = 27;
// workspace variant (kd_global): the gains KD, the per-point records ROT and the per-knot blocks GK live in a global
// workspace slice of the instance, [KD | ROT | GK], instead of LDS (19 KB of LDS at N=10: two waves per SIMD)
// NL contact points (4: Go1; 8: the synthetic biped of BASELINE config 5): the per-point arrays scale with NL, the gains
// [Xw | xw], [Xz | xz] (156 per knot) and the per-knot blocks G, r6 live in the 6-dimensional wrench space and do not
// sl_global (WVAR 6, round 5): the slack / multiplier / residual arrays S, LAM, RC live in the workspace slice too, behind
// [KD | ROT | GK]; such a layout is LEAN (no direction arrays, weakly-active flags in a register) like the eight-point one.
// N=20: 37 KB -> 18 KB of LDS, i.e. two waves per SIMD at the reference's own horizon.
template <int NL = 4>
__host__ __device__ inline size_t wform_slice(int N, bool sl_global = false) {
  return ((size_t)N * (13 * 12 + 21 * NL + kGK + (sl_global ? 18 * NL : 0)) + 1 + 1) & ~(size_t)1;
}
template <int NL = 4>
__host__ __device__ inline Layout make_layout_w(int N, LayoutW* W, bool kd_global = false, bool sl_global = false, bool cv = false,
                                                bool full = false) {      // full: the direction arrays even for eight points (reference mode)
  Layout L;
  int o = 0;
  auto take = [&](int n) { int r = o; o += n; return r; };
  constexpr int nu = 3 * NL, nc = 6 * NL, nl = NL;
  L.cst = take(NL == 4 ? 64 : 80);
  L.bw0 = take(3 * nu);
  L.refp = take(13);
  L.uref = take(nu);
  L.X = take((N + 1) * 13);
  L.U = take(N * nu);
  L.Xc = take((N + 1) * 13);
  L.dU = take(N * nu);
  const bool lean = ((NL == 8) && !full) || sl_global;
  if (sl_global) {      // offsets relative to the slack block of the workspace slice
    L.S = 0; L.LAM = N * nc; L.RC = 2 * N * nc;
  } else {
    L.S = take(N * nc);
    L.LAM = take(N * nc);
  }
  // lean layouts: the directions are not kept (apply_w recomputes them from dU, the lane kernel's rule) and the weakly-active
  // flags live in a register bit field -- 12 KB less for eight points at N=16, which is what lets four instances share a CU
  L.DS = lean ? -1 : take(N * nc);
  L.DLAM = lean ? -1 : take(N * nc);
  if (!sl_global) L.RC = take(N * nc);
  L.AB = take(N * kAB);
  L.XT = take((N + 1) * kXT);
  if (kd_global) {
    L.KD = 0; L.ROT = N * 13 * 12; W->GK = N * (13 * 12 + 21 * nl);      // offsets inside the workspace slice
    W->SLW = (N * (13 * 12 + 21 * nl + kGK) + 1 + 1) & ~1;
  } else {
    L.KD = take(N * 13 * 12);       // per knot [Xw | xw] (rows 0..5) and [Xz | xz] (rows 6..11), 13 entries per row
    L.ROT = take(N * 21 * nl);      // per (knot, point): T (9), l10 l20 l21, id0 id1 id2, gq (3), 3 spare (zeta, below)
    W->GK = take(N * kGK + 1);      // per knot G (21) r6 (6); one 0.0 behind the array (masked operand reads point at it)
  }
  W->WR = take(N * 6);
  L.CV = cv ? take(N * 8) : -1;      // ConvexMpc: per knot Winv_k (w00 w01 w11 wzz) and Iw_k (i00 i01 i11 izz)
  // set-up scratch (one record): the slack arrays are initialised after it; with the slacks in the workspace X (13 (N + 1)
  // doubles, written by the rollout that follows the set-up) takes the record
  L.tile = sl_global ? L.X : L.S;
  L.total = (o + 1) & ~1;
  return L;
}
// the costate zeta_k (6) of the trial rollout is parked in the spare slots of the knot's first two ROT records
template <int NL = 4>
__device__ __forceinline__ int zeta_slot(int k, int i) { return 21 * NL * k + 21 * (i / 3) + 18 + (i % 3); }

constexpr int S6I_(int i, int j) { return i * 6 - i * (i - 1) / 2 + (j - i); }
__host__ __device__ constexpr int S6I(int i, int j) { return i <= j ? S6I_(i, j) : S6I_(j, i); }      // also at run time

// ---- the sibling controller's problem on the same kernels (round 5; ConvexMpc.cpp:81-198, AltroUtils.cpp:224-359) ----------
// MD = WM_CONVEX: the Euler-angle model.  Its state stays in the reference's order [rpy, pos, w, v] in LDS (X, 13 doubles
// apart like the quaternion state); inside the recursion the blocks are ordered like the error state of the quaternion
// model, [p, phi, v, w] (cvperm), and the midpoint transition has the shape the backward pass is written for,
//   Abar = [[I,0,hI,0],[0,A1,0,A3],[0,0,I,0],[0,0,0,I]],  A1 = I + h [0 0 j0; 0 0 j1; 0 0 0],  A3 = h [c s hh j0; -s c hh j1; 0 0 1]
// (c, s, j at the midpoint yaw; the reference's Jacobian omits d Iw^-1 / d yaw, AltroUtils.cpp:295-359), and
// Bbar = M Wr_k with the wrench (F, t') where t' = Iw(yaw_m)^-1 sum r x u: the inertia at the MIDPOINT yaw goes into the
// per-point map knot by knot (Winv_k [r_l]x, formed where it is used from the knot's four numbers), M's attitude block is
// Wt = h hh Rz_m' Iw(yaw)^-1 Iw(yaw_m).  The cone rows act on the world-frame forces directly.  The rollout carries the RAW
// wrench (F, sum r x u): an increment dt' of the linearised torque is dt = Iw(yaw_m) dt' exactly.  This is the formulation
// of the lane kernel's MD_CONVEX passes (qmpc_lane_core.h), in the fragment layout.
// Per-knot numbers beyond A1 / A3 / W (AB record, same slots as the quaternion model): Winv_k = Iw(yaw_m)^-1 (w00 w01 w11 wzz)
// and Iw(yaw_m) (i00 i01 i11 izz) in an array of their own (Layout::CV, 8 per knot; in LDS in every variant).
enum { WM_QUAT = 0, WM_CONVEX = 1 };
__host__ __device__ constexpr int cvperm(int r) { return r < 3 ? r + 3 : (r < 6 ? r - 3 : (r < 9 ? r + 3 : r - 3)); }   // internal row -> state index
__device__ __forceinline__ void cv_step_w(const DevParams& P, const double* x, const double w[6], double* xn) {
  double s0, c0, w00, w01, w11;
  sincos(x[2], &s0, &c0);
  ConvexModel::winv(P, c0, s0, w00, w01, w11);
  const double vd[3] = {w[0] * P.inv_mass, w[1] * P.inv_mass, w[2] * P.inv_mass - 9.81};
  const double wd0[3] = {w00 * w[3] + w01 * w[4], w01 * w[3] + w11 * w[4], P.Iinv[8] * w[5]};
  const double yawm = x[2] + P.hh * x[8];
  const double wm[3] = {x[6] + P.hh * wd0[0], x[7] + P.hh * wd0[1], x[8] + P.hh * wd0[2]};
  double sm_, cm_;
  sincos(yawm, &sm_, &cm_);
  ConvexModel::winv(P, cm_, sm_, w00, w01, w11);
  const double wdm[3] = {w00 * w[3] + w01 * w[4], w01 * w[3] + w11 * w[4], P.Iinv[8] * w[5]};
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
// y = Winv_k t  (t a 3-vector of the wrench's torque part)
__device__ __forceinline__ void cv_winv_mul(const double* W, const double t[3], double y[3]) {
  y[0] = W[0] * t[0] + W[1] * t[1];
  y[1] = W[1] * t[0] + W[2] * t[1];
  y[2] = W[3] * t[2];
}

// ---- pre-pass over (knot, contact point), one lane each: barrier weights, rotated frame, L D L' of the 3 x 3 block,
// and the point's share of G = sum V D^-1 V', r6 = sum V D^-1 g (the arithmetic of rotation_prepass in
// qmpc_kernels.hip followed by leg_block / pass B step 1 of qmpc_lane_core.h).  The four points of a knot sit in one
// lane quad: their shares are summed with quad_perm moves and lane 0 of the quad stores the knot's 27 numbers. ----
// AL = true (reference mode, qmpc_wform_ref_body.inc): augmented-Lagrangian weights instead of barrier weights,
//   w_i = rho [lam_i + rho c_i > 0],  g_i = max(lam_i + rho c_i, 0)   (SURVEY.md Appendix B), c_i in the RC slot, `target` = rho
template <bool AL = false, int NL = 4, int MD = WM_QUAT, bool LEAN = (NL == 8)>
__device__ inline void prepass_w(const DevParams& P, const Layout& L, double* sm, const double* sl,
                                 double* ROT, double* GK, double target, int lane, unsigned kapbits = 0) {
  typedef Dim<NL> D;
  constexpr int LSH = (NL == 8) ? 3 : 2;
  static_assert(NL == 4 || NL == 8, "contact points per knot: one lane quad or two");
  const int N = P.N;
  const double* cst = sm + L.cst;
  double cr[18];
#pragma unroll
  for (int i = 0; i < 18; ++i) cr[i] = cst[D::C_CR + i];
  for (int q0 = 0; q0 < NL * N; q0 += kWave) {
    const int q = q0 + lane;
    const bool live = q < NL * N;
    const int k = live ? (q >> LSH) : 0, l = q & (NL - 1);
    double* out = ROT + D::ROT * k + 21 * l;
    double acc[kGK];
#pragma unroll
    for (int i = 0; i < kGK; ++i) acc[i] = 0.0;
    const bool stance = live && cst[D::C_CON + l] != 0.0;
    if (live && !stance) {
      out[0] = 1; out[1] = 0; out[2] = 0; out[3] = 0; out[4] = 1; out[5] = 0; out[6] = 0; out[7] = 0; out[8] = 1;
      out[9] = 0; out[10] = 0; out[11] = 0; out[12] = 1; out[13] = 1; out[14] = 1; out[15] = 0; out[16] = 0; out[17] = 0;
    }
    if (stance) {
      const int lr = (NL == 4) ? l : (l & 3);      // r_weights[j % 12] weights input j
      const double R0 = P.R[3 * lr], R1 = P.R[3 * lr + 1], R2 = P.R[3 * lr + 2];
      double w[6], gi[6];
#pragma unroll
      for (int i = 0; i < 6; ++i) {
        const double lam = sl[L.LAM + D::NC * k + 6 * l + i];
        const double rc = sl[L.RC + D::NC * k + 6 * l + i];
        if (AL) {
          const double z = lam + target * rc;
          w[i] = (z > 0.0) ? target : 0.0;
          gi[i] = (z > 0.0) ? z : 0.0;
        } else {
          const double s = sl[L.S + D::NC * k + 6 * l + i];
          // weakly-active flag (see ipm_apply); eight points: bit 6 j + i of the lane's j-th (knot, point)
          const double kap = LEAN ? (((kapbits >> (6 * (q0 >> 6) + i)) & 1u) ? 1.0 : 0.0) : sl[L.DS + D::NC * k + 6 * l + i];
          const double is = fast_rcp(s);
          w[i] = lam * is;
          gi[i] = (target + lam * rc) * is - kap * lam;
        }
      }
      // heaviest row i1, second heaviest non-(anti)parallel row i2 (rows 4,5 are antiparallel)
      int i1 = 0;
      double w1 = w[0];
#pragma unroll
      for (int i = 1; i < 6; ++i) if (w[i] > w1) { w1 = w[i]; i1 = i; }
      int i2 = -1;
      double w2 = -1.0;
#pragma unroll
      for (int i = 0; i < 6; ++i) {
        const bool skip = (i == i1) || ((i1 >= 4) && (i >= 4));
        if (!skip && w[i] > w2) { w2 = w[i]; i2 = i; }
      }
      double a1[3] = {0, 0, 0}, a2[3] = {0, 0, 0};
#pragma unroll
      for (int i = 0; i < 6; ++i)
#pragma unroll
        for (int a = 0; a < 3; ++a) {
          a1[a] = (i == i1) ? cr[3 * i + a] : a1[a];
          a2[a] = (i == i2) ? cr[3 * i + a] : a2[a];
        }
      const double in1 = fast_rsqrt(a1[0] * a1[0] + a1[1] * a1[1] + a1[2] * a1[2]);
      double q1[3], q2[3], q3[3];
#pragma unroll
      for (int a = 0; a < 3; ++a) q1[a] = a1[a] * in1;
#pragma unroll
      for (int pass = 0; pass < 2; ++pass) {
        const double dp = a2[0] * q1[0] + a2[1] * q1[1] + a2[2] * q1[2];
#pragma unroll
        for (int a = 0; a < 3; ++a) a2[a] -= dp * q1[a];
      }
      const double in2 = fast_rsqrt(a2[0] * a2[0] + a2[1] * a2[1] + a2[2] * a2[2]);
#pragma unroll
      for (int a = 0; a < 3; ++a) q2[a] = a2[a] * in2;
      q3[0] = q1[1] * q2[2] - q1[2] * q2[1];
      q3[1] = q1[2] * q2[0] - q1[0] * q2[2];
      q3[2] = q1[0] * q2[1] - q1[1] * q2[0];
      double T[9];  // T[3a+b] = (q_b)_a
#pragma unroll
      for (int a = 0; a < 3; ++a) { T[3 * a] = q1[a]; T[3 * a + 1] = q2[a]; T[3 * a + 2] = q3[a]; }
      const double Rl[3] = {R0, R1, R2};
      const double* u = sm + L.U + D::NU * k + 3 * l;
      const double* ur = sm + L.uref + 3 * l;
      const double ru[3] = {R0 * (u[0] - ur[0]), R1 * (u[1] - ur[1]), R2 * (u[2] - ur[2])};
      double gq[3];
      double d00, d01, d02, d11, d12, d22;
      {
        double tr[9];
#pragma unroll
        for (int i = 0; i < 9; ++i) tr[i] = T[i] * Rl[i / 3];
        d00 = tr[0] * T[0] + tr[3] * T[3] + tr[6] * T[6];
        d01 = tr[0] * T[1] + tr[3] * T[4] + tr[6] * T[7];
        d02 = tr[0] * T[2] + tr[3] * T[5] + tr[6] * T[8];
        d11 = tr[1] * T[1] + tr[4] * T[4] + tr[7] * T[7];
        d12 = tr[1] * T[2] + tr[4] * T[5] + tr[7] * T[8];
        d22 = tr[2] * T[2] + tr[5] * T[5] + tr[8] * T[8];
      }
#pragma unroll
      for (int a = 0; a < 3; ++a) gq[a] = T[a] * ru[0] + T[3 + a] * ru[1] + T[6 + a] * ru[2];
#pragma unroll
      for (int i = 0; i < 6; ++i) {
        double at[3];  // rotated row
#pragma unroll
        for (int b = 0; b < 3; ++b) at[b] = T[b] * cr[3 * i] + T[3 + b] * cr[3 * i + 1] + T[6 + b] * cr[3 * i + 2];
#pragma unroll
        for (int a = 0; a < 3; ++a) gq[a] += gi[i] * at[a];
        const double w0 = w[i] * at[0], w1_ = w[i] * at[1];
        d00 += w0 * at[0]; d01 += w0 * at[1]; d02 += w0 * at[2];
        d11 += w1_ * at[1]; d12 += w1_ * at[2];
        d22 += w[i] * at[2] * at[2];
      }
      // L D L' in the pivot order of the frame (heaviest direction first)
      const double id0 = fast_rcp(d00);
      const double l10 = d01 * id0, l20 = d02 * id0;
      const double e11 = d11 - l10 * d01;
      const double id1 = fast_rcp(e11);
      const double e21 = d12 - l20 * d01;
      const double l21 = e21 * id1;
      const double e22 = d22 - l20 * d02 - l21 * e21;
      const double id2 = fast_rcp(e22);
#pragma unroll
      for (int i = 0; i < 9; ++i) out[i] = T[i];
      out[9] = l10; out[10] = l20; out[11] = l21; out[12] = id0; out[13] = id1; out[14] = id2;
      out[15] = gq[0]; out[16] = gq[1]; out[17] = gq[2];
      // V = [T ; Bw0_l T] (6 x 3); vt_j = columns of V L^-T; G += sum_j id_j vt_j vt_j', r6 += sum_j vt_j id_j y_j, y = L^-1 gq
      double V[18];
#pragma unroll
      for (int i = 0; i < 9; ++i) V[i] = T[i];
      {
        const double* bw = sm + L.bw0 + 3 * l;     // Bw0_l[a][c] = bw[12 a + c]
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
          for (int b = 0; b < 3; ++b)
            V[9 + 3 * a + b] = bw[D::NU * a] * T[b] + bw[D::NU * a + 1] * T[3 + b] + bw[D::NU * a + 2] * T[6 + b];
        if (MD == WM_CONVEX) {      // bw is [r_l]x: the point map of knot k is Winv_k [r_l]x
          const double* Wk = sm + L.CV + 8 * k;
#pragma unroll
          for (int b = 0; b < 3; ++b) {
            const double t3[3] = {V[9 + b], V[12 + b], V[15 + b]};
            double y[3];
            cv_winv_mul(Wk, t3, y);
            V[9 + b] = y[0]; V[12 + b] = y[1]; V[15 + b] = y[2];
          }
        }
      }
      const double y0 = gq[0], y1 = gq[1] - l10 * y0, y2 = gq[2] - l20 * y0 - l21 * y1;
      const double z0 = id0 * y0, z1 = id1 * y1, z2 = id2 * y2;
      double v0[6], v1[6], v2[6];
#pragma unroll
      for (int i = 0; i < 6; ++i) {
        v0[i] = V[3 * i];
        v1[i] = V[3 * i + 1] - l10 * v0[i];
        v2[i] = V[3 * i + 2] - l20 * v0[i] - l21 * v1[i];
        acc[21 + i] = v0[i] * z0 + v1[i] * z1 + v2[i] * z2;
      }
#pragma unroll
      for (int i = 0; i < 6; ++i) {
        const double b0 = id0 * v0[i], b1 = id1 * v1[i], b2 = id2 * v2[i];
#pragma unroll
        for (int j = i; j < 6; ++j) acc[S6I(i, j)] = b0 * v0[j] + b1 * v1[j] + b2 * v2[j];
      }
    }
    // sum over the quad (the four contact points of the knot; eight: the neighbouring quad too)
#pragma unroll
    for (int i = 0; i < kGK; ++i) {
      double v = acc[i];
      v += dpp_mov<0xB1>(v);    // quad_perm [1,0,3,2]
      v += dpp_mov<0x4E>(v);    // quad_perm [2,3,0,1]
      if (NL == 8) v += dpp_mov<0x141>(v);      // row_half_mirror: lane i of a group of eight reads lane 7 - i (the other quad)
      acc[i] = v;
    }
    if (live && l == 0) {
      double* gk = GK + kGK * k;
#pragma unroll
      for (int i = 0; i < kGK; ++i) gk[i] = acc[i];
    }
  }
  QSYNC();
}

// diagnostics (PROF builds): a phase boundary pinned between the computation of (a, b) and their uses -- the values
// pass through one volatile asm before the clock is read and through another one after it
template <bool PROF>
__device__ __forceinline__ void tick_dep(Prof<PROF>& prof, int ph, double& a, double& b) {
  if (PROF) {
    asm volatile("s_nop 0" : "+v"(a), "+v"(b));
    prof.tick(ph);
    asm volatile("s_nop 0" : "+v"(a), "+v"(b));
  }
}

template <bool PROF>
__device__ __forceinline__ void tick_dep1(Prof<PROF>& prof, int ph, double& a) {
  if (PROF) {
    asm volatile("s_nop 0" : "+v"(a));
    prof.tick(ph);
    asm volatile("s_nop 0" : "+v"(a));
  }
}

// ---- pieces of the backward pass ------------------------------------------------------------------------------------
// acc += X' Y over fragment rows 0..7 (two k-steps): the wrench-space products (K = 6; rows 6, 7 of both operands are 0)
__device__ __forceinline__ d4 mtm2(const double X[2], const double Y[2], d4 acc) {
  acc = __builtin_amdgcn_mfma_f64_16x16x4f64(X[0], Y[0], acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f64_16x16x4f64(X[1], Y[1], acc, 0, 0, 0);
  return acc;
}
// per-lane multipliers of the two column operations (lane c = column c of the fragment)
struct ColOps {
  double m1c, m2c, keep, hsel;
  __device__ __forceinline__ void init(const DevParams& P, int c) {
    const double m1 = P.h * (P.hh * (1.0 / P.mass)), m2 = P.h * (1.0 / P.mass);
    m1c = (c < 3) ? m1 : 0.0;
    m2c = (c < 3) ? m2 : ((c < 6) ? P.h : 0.0);
    keep = (c >= 3 && c < 6) ? 0.0 : 1.0;
    hsel = (c >= 6 && c < 9) ? P.h : 0.0;
  }
};
// Lane patterns of the backward pass, computed ONCE per solve (they depend on the lane and the horizon only): fragment
// rows r_e = 4 e + g.  Every per-knot operand is one LDS read at a lane-constant index that steps back by the record
// size per knot; lanes outside a pattern read slots that hold 0.0 (fixed index, stride 0), so no select follows the
// loads.  The three column multipliers wt[t] / at[t] of a lane sit 3 doubles apart in the knot's record (immediate
// offsets of one index register); the Wt block of the M fragment and the A1 / A3 blocks of the N = Abar - I fragment
// ARE those multipliers on the lanes whose fragment row is 3 + t.
#ifndef QMPC_W_FUSED_ROWS
#define QMPC_W_FUSED_ROWS 1      // input recovery + directions, and the apply step, one lane per (knot, contact point)
#endif
constexpr int kZeroSlots = 54;      // cst[54..63] hold 0.0 (cst[] is used up to slot 52)
template <int NL> __host__ __device__ constexpr int kZeroSlotsT() { return NL == 4 ? kZeroSlots : 70; }      // 8 points: cst[] is used up to slot 68 of 80
struct BwPat {
  ColOps co;
  double Mc[3], Nc[2], qadd[3], sel[3];     // sel[t]: 1.0 where the lane's fragment row is 3 + t
  int ix_w, st_w, ix_a, st_a, ix_x[3], st_x[3], ix_g[2], st_g[2], kwo[2], xoffN[3];
  bool c12;
  template <int NL = 4, int MD = WM_QUAT>
  __device__ __forceinline__ void init(const DevParams& P, const Layout& L, int lane) {
    const int N = P.N;
    const int c = lane & 15, g = lane >> 4;
    co.init(P, c);
    const int zs = L.cst + kZeroSlotsT<NL>();
    const bool c35 = (c >= 3 && c < 6), c911 = (c >= 9 && c < 12);
    c12 = c == 12;
    const int abN = L.AB + kAB * (N - 1);
    ix_w = c35 ? abN + 18 + (c - 3) : zs;   st_w = c35 ? kAB : 0;
    ix_a = c35 ? abN + (c - 3) : (c911 ? abN + 9 + (c - 9) : zs);   st_a = (c35 || c911) ? kAB : 0;
    sel[0] = (g == 3) ? 1.0 : 0.0;      // row 3 + t lives in (e, g) = (0, 3), (1, 0), (1, 1)
    sel[1] = (g == 0) ? 1.0 : 0.0;
    sel[2] = (g == 1) ? 1.0 : 0.0;
    const double m1 = P.h * (P.hh * (1.0 / P.mass)), m2 = P.h * (1.0 / P.mass);
#pragma unroll
    for (int e = 0; e < 3; ++e) {
      const int r = 4 * e + g;
      const bool phi = (r >= 3 && r < 6);
      double v = 0.0;
      if (c < 6) {
        if (r < 3) v = (r == c) ? m1 : 0.0;
        else if (r >= 6 && r < 9) v = (r - 6 == c) ? m2 : 0.0;
        else if (r >= 9) v = (r - 9 == c - 3) ? P.h : 0.0;
      }
      Mc[e] = v;
      if (MD == WM_CONVEX) qadd[e] = (c < 12 && r == c) ? P.Q[cvperm(r < 12 ? r : 0)] : 0.0;      // LQR cost: constant diagonal Hessian
      else qadd[e] = (c < 12 && r == c && !phi) ? P.Q[(r < 3) ? r : r + 1] : 0.0;
      const bool xv = (MD == WM_CONVEX) ? c12 : ((phi && c35) || c12);
      const int xo = c12 ? 9 + r : 3 * (r - 3) + (c - 3);
      xoffN[e] = xv ? xo : -1;
      ix_x[e] = xv ? L.XT + kXT * (N - 1) + xo : zs;
      st_x[e] = xv ? kXT : 0;
    }
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      const int r = 4 * e + g;
      const bool phi = (r >= 3 && r < 6);
      // N = Abar - I, rows 0..7 (rows 6, 7 are zero)
      Nc[e] = (r < 3 && c == r + 6) ? P.h : ((phi && r == c) ? -1.0 : 0.0);
      // G (columns 0..5) and r6 (column 12) share a register: as the A operand of G Yp the column 12 only feeds the
      // unused output row 12
      const bool gv = r < 6 && (c < 6 || c12);
      const int go = c12 ? 21 + r : S6I(r < 6 ? r : 0, c < 6 ? c : 0);
      ix_g[e] = gv ? kGK * (N - 1) + go : kGK * N;      // relative to GK; GK[27 N] holds 0.0
      st_g[e] = gv ? kGK : 0;
      kwo[e] = (r < 6 && c < 13) ? 13 * r + c : -1;      // [Xw | xw] row r; [Xz | xz] row 6 + r is 78 doubles further
    }
  }
};
// acc += sum_t x[lane 3 + t of the row] * m[t]: three v_fmac_f64 with a DPP source (row_newbcast on src0 of the 64-bit
// VOP2 form, gfx90a+).  The leading s_nop covers the VALU-write -> DPP-read hazard, which the compiler does not track
// through inline asm.
#ifndef QMPC_COL_FUSED
#define QMPC_COL_FUSED 1
#endif
__device__ __forceinline__ double fma_bcast345(double acc, double x, const double m[3]) {
#if QMPC_COL_FUSED
  asm("s_nop 1\n\t"
      "v_fmac_f64_dpp %0, %1, %2 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\t"
      "v_fmac_f64_dpp %0, %1, %3 row_newbcast:4 row_mask:0xf bank_mask:0xf\n\t"
      "v_fmac_f64_dpp %0, %1, %4 row_newbcast:5 row_mask:0xf bank_mask:0xf"
      : "+v"(acc)
      : "v"(x), "v"(m[0]), "v"(m[1]), "v"(m[2]));
  return acc;
#else
  acc = fma(m[0], row_bcast<3>(x), acc);
  acc = fma(m[1], row_bcast<4>(x), acc);
  return fma(m[2], row_bcast<5>(x), acc);
#endif
}
// x M   (columns 0..5 of the result; M = [[m1 I, 0], [0, Wt], [m2 I, 0], [0, h I]]): lane c < 3 takes m1 x[c] + m2 x[c+6],
// lane 3 + b takes sum_t x[3+t] Wt[t][b] + h x[9+b]; wt[t] = Wt[t][c-3] on lanes 3..5, 0 elsewhere
__device__ __forceinline__ double times_M(double x, const ColOps& co, const double wt[3]) {
  double r = co.m1c * x;
  r = fma(co.m2c, dpp_mov<0x106>(x), r);       // row_shl:6: lane c reads lane c + 6
  return fma_bcast345(r, x, wt);
}
// x Abar_aug  (Abar = [[I,0,hI,0],[0,A1,0,A3],[0,0,I,0],[0,0,0,I]], the gradient column 12 untouched):
// at[t] = A1[t][c-3] on lanes 3..5, A3[t][c-9] on lanes 9..11, 0 elsewhere
__device__ __forceinline__ double times_Abar(double x, const ColOps& co, const double at[3]) {
  double r = co.keep * x;
  r = fma(co.hsel, dpp_mov<0x116>(x), r);      // row_shr:6: lane c reads lane c - 6
  return fma_bcast345(r, x, at);
}

// One Gauss-Jordan step on pivot J < 6 of the pair (M | Rr) held in two fragment registers (rows 0..3 and 4..7):
// row J is eliminated from every other row.  Returns -1 / pivot: the diagonal is divided out at the end, and a later step
// does not touch an earlier pivot (its column is already zero in the later pivot rows).
// PERM: the pivot row reaches the other row groups through v_permlane16/32_swap (VALU) instead of ds_bpermute (LDS)
#ifndef QMPC_GJ_PERM
#define QMPC_GJ_PERM 0
#endif
#ifndef QMPC_GJ_FUSED
#define QMPC_GJ_FUSED 0     // measured: 1.76 M against 1.82 M solves/s -- the 64-bit DPP form issues slower than mov + fma
#endif
template <int J, bool INV = false>
__device__ __forceinline__ double gj6_step(double M[2], double Rr[2], int c, int g, bool& pd, double* Iv = nullptr) {
  constexpr int ej = J >> 2, gj = J & 3;
  double mrow, rrow, irow = 0.0;
  if (QMPC_GJ_PERM) {
    mrow = rowgroup_bcast<gj>(M[ej]);
    rrow = rowgroup_bcast<gj>(Rr[ej]);
    if (INV) irow = rowgroup_bcast<gj>(Iv[ej]);
  } else {
    const int src = (gj << 4) | c;
    mrow = __shfl(M[ej], src);      // row J, same column, all row groups
    rrow = __shfl(Rr[ej], src);
    if (INV) irow = __shfl(Iv[ej], src);
  }
  const double piv = read_lane(M[ej], (gj << 4) | J);
  pd = pd && (piv > 0.0);          // false for a NaN pivot too (fmin / fmax would drop it silently)
  const double ninv = -fast_rcp(piv);
#if QMPC_GJ_FUSED
  // X[r][c] += X[r][J] * (-row_J[c] / piv) as ONE v_fmac_f64 with a DPP source per fragment register; the DPP row mask
  // (one bit per row group) leaves the pivot row itself untouched.  The right-hand side first: it reads the old column J.
  const double nm = ninv * mrow, nr = ninv * rrow;
  constexpr int m0 = (ej == 0) ? (0xf ^ (1 << gj)) : 0xf, m1 = (ej == 1) ? (0xf ^ (1 << gj)) : 0xf;
  asm("s_nop 1\n\t"
      "v_fmac_f64_dpp %2, %0, %5 row_newbcast:%6 row_mask:%7 bank_mask:0xf\n\t"
      "v_fmac_f64_dpp %3, %1, %5 row_newbcast:%6 row_mask:%8 bank_mask:0xf\n\t"
      "v_fmac_f64_dpp %0, %0, %4 row_newbcast:%6 row_mask:%7 bank_mask:0xf\n\t"
      "v_fmac_f64_dpp %1, %1, %4 row_newbcast:%6 row_mask:%8 bank_mask:0xf"
      : "+v"(M[0]), "+v"(M[1]), "+v"(Rr[0]), "+v"(Rr[1])
      : "v"(nm), "v"(nr), "n"(J), "n"(m0), "n"(m1));
  static_assert(!INV, "the fused form does not track the inverse");
#else
#pragma unroll
  for (int e = 0; e < 2; ++e) {
    const double col = row_bcast<J>(M[e]);
    const double f = ((e == ej) && (g == gj)) ? 0.0 : col * ninv;
    M[e] = fma(f, mrow, M[e]);
    Rr[e] = fma(f, rrow, Rr[e]);
    if (INV) Iv[e] = fma(f, irow, Iv[e]);
  }
#endif
  return ninv;
}

// Two Gauss-Jordan steps at once: the 2 x 2 pivot block B = [[a, b], [b, d]] on rows / columns (J0, J0 + 1), J0 = 0, 2, 4 (both
// rows live in the same fragment register, in adjacent row groups).  Every other row r loses its columns J0, J0 + 1:
//   row_r += f0 row_J0 + f1 row_J1,   (f0, f1) = -(M[r][J0], M[r][J0+1]) B^-1,   B^-1 = [[d, -b], [-b, a]] / det
// and the pivot rows themselves leave as -B^-1 [row_J0; row_J1] (the same formula with (M[r][J0], M[r][J0+1]) = e_1, e_2 and
// no old row): after the three steps the right-hand side holds X = -W'^-1 Q, the tracked inverse -W'^-1, nothing is
// divided out at the end.  One reciprocal (of det = a d - b^2 > 0 together with a > 0: Sylvester) and one round of
// cross-row-group broadcasts per PAIR of pivots: the serial chain of the stage solve has three links instead of six
// (round 5: stage solve 950 -> see profiles/r05_phase_cycles.txt).  Block Cholesky: as stable as the scalar pivots
// (the cancellation in det is the one the scalar Schur complement d - b^2 / a carries).
#ifndef QMPC_GJ_BLOCK2
#define QMPC_GJ_BLOCK2 1
#endif
template <int J0, bool INV = false>
__device__ __forceinline__ void gj6_pair_step(double M[2], double Rr[2], int c, int g, bool& pd, double* Iv = nullptr) {
  constexpr int ej = J0 >> 2, g0 = J0 & 3, g1 = g0 + 1;
  static_assert((J0 & 1) == 0 && J0 < 6, "pivot pairs (0,1), (2,3), (4,5)");
  const int s0 = (g0 << 4) | c, s1 = (g1 << 4) | c;
  const double m0 = __shfl(M[ej], s0), m1 = __shfl(M[ej], s1);      // rows J0, J0 + 1, same column, all row groups
  const double r0 = __shfl(Rr[ej], s0), r1 = __shfl(Rr[ej], s1);
  double i0 = 0.0, i1 = 0.0;
  if (INV) { i0 = __shfl(Iv[ej], s0); i1 = __shfl(Iv[ej], s1); }
  const double a = read_lane(M[ej], (g0 << 4) | J0), b = read_lane(M[ej], (g0 << 4) | (J0 + 1)),
               d = read_lane(M[ej], (g1 << 4) | (J0 + 1));
  const double det = fma(a, d, -(b * b));
  pd = pd && (a > 0.0) && (det > 0.0);      // false for NaNs too
  const double rd = fast_rcp(det);
  const double ia = a * rd, ib = b * rd, id = d * rd;
#pragma unroll
  for (int e = 0; e < 2; ++e) {
    double c0 = row_bcast<J0>(M[e]), c1 = row_bcast<J0 + 1>(M[e]);
    double bm = M[e], br = Rr[e], bi = INV ? Iv[e] : 0.0;
    if (e == ej) {
      const bool p0 = g == g0, p1 = g == g1;
      c0 = p0 ? 1.0 : (p1 ? 0.0 : c0);
      c1 = p1 ? 1.0 : (p0 ? 0.0 : c1);
      if (p0 || p1) { bm = 0.0; br = 0.0; bi = 0.0; }
    }
    const double f0 = fma(ib, c1, -(id * c0)), f1 = fma(ib, c0, -(ia * c1));
    M[e] = fma(f1, m1, fma(f0, m0, bm));
    Rr[e] = fma(f1, r1, fma(f0, r0, br));
    if (INV) Iv[e] = fma(f1, i1, fma(f0, i0, bi));
  }
}

// Riccati backward pass in the wrench form; writes per knot [Xw | xw] and [Xz | xz] (KD).  Returns nonzero when a pivot
// of W' is not positive (P lost positive definiteness: QMPC_NOT_PD).
// REFINE (reference mode, round 5): one step of iterative refinement on the 6 x 6 stage solve.  W' = S6 (I + G S6) carries
// cond(S6) twice; the residual is taken on the system that carries it ONCE, (I + G S6) X = -C (not symmetric, so never
// eliminated itself), and the correction goes back through the SPD elimination: W' dX = S6 r, with W'^-1 tracked by the same
// Gauss-Jordan steps (a third fragment pair that starts as the identity).  The converged mode does not need it -- its fixed
// point does not depend on the accuracy of a Newton direction -- but a TRUNCATED iterate keeps the rounding of its
// directions: without the refinement 65 % of the N=20 reference-mode forces were within 1e-6 N of the oracle's.
template <bool PROF, bool REFINE = false, int MD = WM_QUAT>
// y0out (optional, 6 per knot): y0_k = M' p_{k+1}, the wrench-space costate BEFORE the stage solve -- the rotated input
// gradient of contact point l is V_l' y0 + gq_l (the expected decrease of the reference mode's line search needs it)
__device__ inline int backward_pass_w(const DevParams& P, const Layout& L, const BwPat& bp, double* sm, double* KD,
                                      const double* GK, int lane, Prof<PROF>& prof, double* y0out = nullptr) {
  const int N = P.N;
  const int c = lane & 15, g = lane >> 4;
  const double wscale = (MD == WM_CONVEX) ? P.h * P.hh : (0.5 * P.hh) * P.h;        // Wt = (h^2 / 4) Gn'Gm;  ConvexMpc: h hh (Rz_m' Iw^-1 Iw_m)
  const ColOps& co = bp.co;
  const double* Mc = bp.Mc;
  const double* Nc = bp.Nc;
  const double* qadd = bp.qadd;
  const double* msel = bp.sel;
  const double* nsel = bp.sel;
  const int* xoffN = bp.xoffN;
  const int* kwo = bp.kwo;
  int ix_w = bp.ix_w, ix_a = bp.ix_a, ix_x[3] = {bp.ix_x[0], bp.ix_x[1], bp.ix_x[2]}, ix_g[2] = {bp.ix_g[0], bp.ix_g[1]};
  const bool rowok1 = kwo[1] >= 0;
  // ---- terminal cost-to-go  P_aug = [lxx_N | lx_N] ----
  double Pf[3];
  {
    const double* XTk = sm + L.XT + kXT * N;
#pragma unroll
    for (int e = 0; e < 3; ++e) Pf[e] = qadd[e] + ((xoffN[e] >= 0) ? XTk[xoffN[e]] : 0.0);
  }
  bool pd = true;
  const d4 z4 = {0.0, 0.0, 0.0, 0.0};
  // operands of a knot (independent of the cost-to-go): loaded one knot ahead, right after the first products are issued
  struct KnotOps { double wt[3], at[3], gr[2], xt[3]; };
  auto load_ops = [&](KnotOps& o) {       // the knot the running indices point at; then one knot back
#pragma unroll
    for (int t = 0; t < 3; ++t) {
      o.wt[t] = sm[ix_w + 3 * t]; o.at[t] = sm[ix_a + 3 * t]; o.xt[t] = sm[ix_x[t]];
      ix_x[t] -= bp.st_x[t];
    }
    ix_w -= bp.st_w; ix_a -= bp.st_a;
#pragma unroll
    for (int e = 0; e < 2; ++e) { o.gr[e] = GK[ix_g[e]]; ix_g[e] -= bp.st_g[e]; }
  };
  KnotOps ops, opn;
  load_ops(ops);
  for (int k = N - 1; k >= 0; --k) {
    double wt[3], Mf[3], Nf[2], R6f[2];
    const double* at = ops.at;
    const double* Gf = ops.gr;
    const double* xt = ops.xt;
#pragma unroll
    for (int t = 0; t < 3; ++t) wt[t] = wscale * ops.wt[t];
    Mf[0] = fma(msel[0], wt[0], Mc[0]);
    Mf[1] = fma(msel[1], wt[1], fma(msel[2], wt[2], Mc[1]));
    Mf[2] = Mc[2];
    Nf[0] = fma(nsel[0], at[0], Nc[0]);
    Nf[1] = fma(nsel[1], at[1], fma(nsel[2], at[2], Nc[1]));
    R6f[0] = bp.c12 ? ops.gr[0] : 0.0;
    R6f[1] = bp.c12 ? ops.gr[1] : 0.0;
    tick_dep(prof, PH_BUILD, Pf[0], Pf[1]);
    // ---- 1. Yp = M' [P | p]  (6 x 13) ----
    double Yp[2];
    {
      const d4 a = mtm3(Mf, Pf, z4);
      Yp[0] = a[0]; Yp[1] = a[1];
    }
    if (k > 0) load_ops(opn);
    if (y0out && bp.c12) {
      if (g < 4) y0out[6 * k + g] = Yp[0];
      if (g < 2) y0out[6 * k + 4 + g] = Yp[1];
    }
    tick_dep(prof, PH_DRAIN, Yp[0], Yp[1]);
    // ---- 2. S6 = Yp_fb M  (column operation) ----
    double S6[2];
    S6[0] = times_M(Yp[0], co, wt);
    S6[1] = times_M(Yp[1], co, wt);
    // ---- 3. C = G Yp + [0 | r6] ;  4. Q = S6 C,  W' = S6 + Q_fb M ----
    double Cf[2], Qf[2], Wm[2];
    {
      const d4 ini = {R6f[0], R6f[1], 0.0, 0.0};
      const d4 a = mtm2(Gf, Yp, ini);
      Cf[0] = a[0]; Cf[1] = a[1];
      const d4 b = mtm2(S6, Cf, z4);
      Qf[0] = b[0]; Qf[1] = b[1];
    }
    Wm[0] = S6[0] + times_M(Qf[0], co, wt);
    Wm[1] = S6[1] + times_M(Qf[1], co, wt);
    tick_dep(prof, PH_MFMA, Wm[0], Wm[1]);
    // ---- 5. W' X = -Q ----
    double Iv[2];
    if (REFINE) {      // the identity, rows 0..5: it leaves the elimination as diag * W'^-1
      Iv[0] = (c == g) ? 1.0 : 0.0;
      Iv[1] = (g < 2 && c == 4 + g) ? 1.0 : 0.0;
    }
    double Xf[2];
#if QMPC_GJ_BLOCK2
    gj6_pair_step<0, REFINE>(Wm, Qf, c, g, pd, Iv);
    gj6_pair_step<2, REFINE>(Wm, Qf, c, g, pd, Iv);
    gj6_pair_step<4, REFINE>(Wm, Qf, c, g, pd, Iv);
    const double nd0 = 1.0, nd1 = 1.0;      // (the pivot rows leave the block steps as -B^-1 row)
    Xf[0] = Qf[0];
    Xf[1] = rowok1 ? Qf[1] : 0.0;
#else
    const double n0 = gj6_step<0, REFINE>(Wm, Qf, c, g, pd, Iv);
    const double n1 = gj6_step<1, REFINE>(Wm, Qf, c, g, pd, Iv);
    const double n2 = gj6_step<2, REFINE>(Wm, Qf, c, g, pd, Iv);
    const double n3 = gj6_step<3, REFINE>(Wm, Qf, c, g, pd, Iv);
    const double n4 = gj6_step<4, REFINE>(Wm, Qf, c, g, pd, Iv);
    const double n5 = gj6_step<5, REFINE>(Wm, Qf, c, g, pd, Iv);
    const double nd0 = (g == 0) ? n0 : (g == 1 ? n1 : (g == 2 ? n2 : n3)), nd1 = (g == 0) ? n4 : n5;
    Xf[0] = Qf[0] * nd0;      // X = -diag^-1 Q
    Xf[1] = rowok1 ? Qf[1] * nd1 : 0.0;
#endif
    if (REFINE) {
      // -r = C + X + G (S6 X);  v = S6 (-r);  X <- X - W'^-1 v   (nWi = -W'^-1 = n_r Iv, rows 6, 7 zero like every operand here)
      double nWi[2], T1[2], nr[2], vv[2];
      nWi[0] = Iv[0] * nd0;
      nWi[1] = (g < 2) ? Iv[1] * nd1 : 0.0;
      {
        const d4 a = mtm2(S6, Xf, z4);
        T1[0] = a[0]; T1[1] = a[1];
      }
      {
        const d4 ini = {Cf[0] + Xf[0], Cf[1] + Xf[1], 0.0, 0.0};
        const d4 a = mtm2(Gf, T1, ini);
        nr[0] = a[0]; nr[1] = rowok1 ? a[1] : 0.0;
      }
      {
        const d4 a = mtm2(S6, nr, z4);
        vv[0] = a[0]; vv[1] = a[1];
      }
      {
        const d4 ini = {Xf[0], Xf[1], 0.0, 0.0};
        const d4 a = mtm2(nWi, vv, ini);
        Xf[0] = a[0];
        Xf[1] = rowok1 ? a[1] : 0.0;
      }
    }
    tick_dep(prof, PH_SOLVE, Xf[0], Xf[1]);
    // ---- 6. Pi = [P | p] + Yp_fb' X ;  7. gains [Xz | xz] = Yp + S6 X ----
    double Pi[3];
    {
      const d4 ini = {Pf[0], Pf[1], Pf[2], 0.0};
      const d4 a = mtm2(Yp, Xf, ini);
      Pi[0] = a[0]; Pi[1] = a[1]; Pi[2] = a[2];
    }
    double Zf[2];
    {
      const d4 ini = {Yp[0], Yp[1], 0.0, 0.0};
      const d4 a = mtm2(S6, Xf, ini);
      Zf[0] = a[0]; Zf[1] = a[1];
    }
    // ---- 8. [P | p]_k = [lxx | lx] + Abar' (Pi Abar_aug) ----
    double Uf[3];
#pragma unroll
    for (int e = 0; e < 3; ++e) Uf[e] = times_Abar(Pi[e], co, at);
    {
      const d4 ini = {Uf[0] + (qadd[0] + xt[0]), Uf[1] + (qadd[1] + xt[1]), Uf[2] + (qadd[2] + xt[2]), 0.0};
      const d4 a = mtm2(Nf, Uf, ini);
      Pf[0] = a[0]; Pf[1] = a[1]; Pf[2] = a[2];
    }
    {
      double* KDk = KD + 156 * k;
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        if (kwo[e] >= 0) { KDk[kwo[e]] = Xf[e]; KDk[kwo[e] + 78] = Zf[e]; }
      }
    }
    ops = opn;
    tick_dep(prof, PH_PUPD, Pf[0], Pf[1]);
  }
  return !pd;
}

// ---- explicit midpoint step given the wrench w = (sum of the forces, sum of Bw0_l u_l) (AltroUtils.cpp:9-22 on :363-392)
__device__ __forceinline__ void srbd_step_w(const DevParams& P, const double gb[3], const double wd0[3], const double* x,
                                            const double w[6], double* xn) {
  double vd[3], wd[3];
#pragma unroll
  for (int a = 0; a < 3; ++a) { vd[a] = w[a] * P.inv_mass + gb[a]; wd[a] = wd0[a] + w[3 + a]; }
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

// wrench of every knot from the inputs U (WITH_DU: U + dU), lane (k, i), i < 6
template <bool WITH_DU, int NL = 4>
__device__ inline void wrench_from_inputs(const DevParams& P, const Layout& L, const LayoutW& LW, double* sm, int lane) {
  const int N = P.N;
  constexpr int NU = 3 * NL;
  for (int q = lane; q < 6 * N; q += kWave) {
    const int k = q / 6, i = q - 6 * k;
    double u[NU];
#pragma unroll
    for (int j = 0; j < NU; ++j) u[j] = sm[L.U + NU * k + j] + (WITH_DU ? sm[L.dU + NU * k + j] : 0.0);
    const double* b = sm + L.bw0 + ((i >= 3) ? NU * (i - 3) : 0);
    if (NL == 4) {
      double s0 = 0.0, s1 = 0.0, f0 = 0.0, f1 = 0.0;
#pragma unroll
      for (int j = 0; j < 6; ++j) { s0 += b[j] * u[j]; s1 += b[6 + j] * u[6 + j]; }
#pragma unroll
      for (int a = 0; a < 3; ++a) {
        f0 += (i == a) ? u[a] + u[6 + a] : 0.0;
        f1 += (i == a) ? u[3 + a] + u[9 + a] : 0.0;
      }
      sm[LW.WR + q] = (i < 3) ? f0 + f1 : s0 + s1;
    } else {
      double sv[4] = {0.0, 0.0, 0.0, 0.0}, fv[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
      for (int j = 0; j < NU; ++j) sv[j / 6] += b[j] * u[j];
#pragma unroll
      for (int l = 0; l < NL; ++l)
#pragma unroll
        for (int a = 0; a < 3; ++a) fv[l >> 1] += (i == a) ? u[3 * l + a] : 0.0;
      sm[LW.WR + q] = (i < 3) ? (fv[0] + fv[1]) + (fv[2] + fv[3]) : (sv[0] + sv[1]) + (sv[2] + sv[3]);
    }
  }
  QSYNC();
}

// shortened primal step: scale the trial increment and re-roll the states open loop from the knots' wrenches
template <int NL = 4, int MD = WM_QUAT>
__device__ inline void rollout_scaled_w(const DevParams& P, const Layout& L, const LayoutW& LW, double* sm, double ap, int lane) {
  typedef Dim<NL> D;
  const int N = P.N;
  const double* cst = sm + L.cst;
  for (int i = lane; i < N * D::NU; i += kWave) sm[L.dU + i] *= ap;
  QSYNC();
  wrench_from_inputs<true, NL>(P, L, LW, sm, lane);
  double gb[3], wd0[3];
#pragma unroll
  for (int a = 0; a < 3; ++a) { gb[a] = cst[D::C_GB + a]; wd0[a] = cst[D::C_WD0 + a]; }
  double x[13], xn[13], w[6], wn[6];
#pragma unroll
  for (int i = 0; i < 13; ++i) x[i] = cst[D::C_X0 + i];
#pragma unroll
  for (int i = 0; i < 6; ++i) w[i] = sm[LW.WR + i];
  for (int k = 0; k < N; ++k) {
    const int kn = (k + 1 < N) ? k + 1 : k;
#pragma unroll
    for (int i = 0; i < 6; ++i) wn[i] = sm[LW.WR + 6 * kn + i];
    if (MD == WM_CONVEX) cv_step_w(P, x, w, xn);
    else srbd_step_w(P, gb, wd0, x, w, xn);
#pragma unroll
    for (int i = 0; i < 13; ++i) x[i] = xn[i];
#pragma unroll
    for (int i = 0; i < 6; ++i) w[i] = wn[i];
    if (lane == 0)
#pragma unroll
      for (int i = 0; i < 13; ++i) sm[L.Xc + 13 * (k + 1) + i] = xn[i];
  }
  QSYNC();
}

// ---- expansions at (X, U): four lanes per knot (three columns of the Jacobian blocks + the cost expansion) as in
// expand_knot_part (qmpc_device.h; AltroUtils.cpp:78-110,153-168), with the angular acceleration taken from the knot's
// wrench (WR) instead of being re-summed over the twelve inputs by every lane ------------------------------------------
template <int NL = 4, int MD = WM_QUAT>
__device__ inline void expansions_w(const DevParams& P, const Layout& L, const LayoutW& LW, double* sm, int lane) {
  typedef Dim<NL> D;
  const int N = P.N;
  const double* cst = sm + L.cst;
  if (MD == WM_CONVEX) {
    // one lane per knot (two sincos dominate): A1, A3, W = Rz_m' Iw(yaw)^-1 Iw(yaw_m) into the AB record, Winv_k / Iw_k and
    // the cost gradient (internal block order) into XT -- the arithmetic of ConvexModel::expand / cv_expansion
    for (int k = lane; k <= N; k += kWave) {
      double x[12];
#pragma unroll
      for (int i = 0; i < 12; ++i) x[i] = sm[L.X + 13 * k + i];
      double* ABk = sm + L.AB + kAB * k;
      double* XTk = sm + L.XT + kXT * k;
      if (k < N) {
        const double t0 = sm[LW.WR + 6 * k + 3], t1 = sm[LW.WR + 6 * k + 4];
        double s0, c0, w00, w01, w11;
        sincos(x[2], &s0, &c0);
        ConvexModel::winv(P, c0, s0, w00, w01, w11);
        const double wm0 = x[6] + P.hh * (w00 * t0 + w01 * t1);
        const double wm1 = x[7] + P.hh * (w01 * t0 + w11 * t1);
        double sm_, cm_;
        sincos(x[2] + P.hh * x[8], &sm_, &cm_);
        const double j0 = wm1 * cm_ - wm0 * sm_, j1 = -wm0 * cm_ - wm1 * sm_;
        const double m00 = cm_ * w00 + sm_ * w01, m01 = cm_ * w01 + sm_ * w11;
        const double m10 = -sm_ * w00 + cm_ * w01, m11 = -sm_ * w01 + cm_ * w11;
        ConvexModel::winv(P, cm_, sm_, w00, w01, w11);
        const double ixx = 1.0 / P.Iinv[0], iyy = 1.0 / P.Iinv[4], izz = 1.0 / P.Iinv[8];
        const double i00 = cm_ * cm_ * ixx + sm_ * sm_ * iyy, i01 = cm_ * sm_ * (ixx - iyy), i11 = sm_ * sm_ * ixx + cm_ * cm_ * iyy;
        ABk[0] = 1.0; ABk[1] = 0.0; ABk[2] = P.h * j0;
        ABk[3] = 0.0; ABk[4] = 1.0; ABk[5] = P.h * j1;
        ABk[6] = 0.0; ABk[7] = 0.0; ABk[8] = 1.0;
        ABk[9] = P.h * cm_;   ABk[10] = P.h * sm_; ABk[11] = P.h * P.hh * j0;
        ABk[12] = -P.h * sm_; ABk[13] = P.h * cm_; ABk[14] = P.h * P.hh * j1;
        ABk[15] = 0.0; ABk[16] = 0.0; ABk[17] = P.h;
        ABk[18] = m00 * i00 + m01 * i01; ABk[19] = m00 * i01 + m01 * i11; ABk[20] = 0.0;
        ABk[21] = m10 * i00 + m11 * i01; ABk[22] = m10 * i01 + m11 * i11; ABk[23] = 0.0;
        ABk[24] = 0.0; ABk[25] = 0.0; ABk[26] = 1.0;
        double* CVk = sm + L.CV + 8 * k;
        CVk[0] = w00; CVk[1] = w01; CVk[2] = w11; CVk[3] = P.Iinv[8];
        CVk[4] = i00; CVk[5] = i01; CVk[6] = i11; CVk[7] = izz;
      }
#pragma unroll
      for (int i = 0; i < 9; ++i) XTk[i] = 0.0;      // (no attitude block in this cost's Hessian)
      double xr[12];
      ConvexModel::xref(P, sm + L.refp, k, xr);
#pragma unroll
      for (int r = 0; r < 12; ++r) XTk[9 + r] = P.Q[cvperm(r)] * (x[cvperm(r)] - xr[cvperm(r)]);
    }
    QSYNC();
    return;
  }
  for (int q = lane; q < 4 * (N + 1); q += kWave) {
    const int k = q >> 2, part = q & 3;
    double x[13];
#pragma unroll
    for (int i = 0; i < 13; ++i) x[i] = sm[L.X + 13 * k + i];
    double* ABk = sm + L.AB + kAB * k;
    double* XTk = sm + L.XT + kXT * k;
    if (part < 3) {
      if (k >= N) continue;
      const int c = part;
      double xn[4], wd[3];
#pragma unroll
      for (int i = 0; i < 4; ++i) xn[i] = sm[L.X + 13 * (k + 1) + 3 + i];
#pragma unroll
      for (int a = 0; a < 3; ++a) wd[a] = cst[D::C_WD0 + a] + sm[LW.WR + 6 * k + 3 + a];
      double G0[12], Gm[12], Gn[12];
      quat_G(&x[3], G0);
      double qm[4], wm[3];
#pragma unroll
      for (int r = 0; r < 4; ++r)
        qm[r] = x[3 + r] + P.hh * (0.5 * (G0[3 * r] * x[10] + G0[3 * r + 1] * x[11] + G0[3 * r + 2] * x[12]));
#pragma unroll
      for (int a = 0; a < 3; ++a) wm[a] = x[10 + a] + P.hh * wd[a];
      quat_G(qm, Gm);
      quat_G(xn, Gn);
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
        ABk[3 * r + c] = Gn[r] * ag[0] + Gn[3 + r] * ag[1] + Gn[6 + r] * ag[2] + Gn[9 + r] * ag[3];
        ABk[9 + 3 * r + c] = Gn[r] * aw[0] + Gn[3 + r] * aw[1] + Gn[6 + r] * aw[2] + Gn[9 + r] * aw[3];
        ABk[18 + 3 * r + c] = Gn[r] * gm[0] + Gn[3 + r] * gm[1] + Gn[6 + r] * gm[2] + Gn[9 + r] * gm[3];
      }
    } else {
      // the cost expansion of knot k = 0..N (part 3 of expand_knot_part)
      double u0[1] = {0.0};
      expand_knot_part<NL>(P, cst, sm + L.bw0, sm + L.refp, k, 3, x, u0, x, ABk, XTk);
    }
  }
  QSYNC();
}

// ---- closed-loop trial rollout (alpha = 1) in the wrench space: lane i < 6 owns row i of [Xw | xw], lanes 6..11 row
// i - 6 of [Xz | xz]; the state is advanced by every lane redundantly, six wrench components are broadcast per knot ----
struct RollLoadsW {
  double xo[13], ab[18], kd[13], wk;
};
// ConvexMpc: the knot's raw wrench (all six, every lane advances the state) and Iw(yaw_m) of the linearisation
struct RollLoadsC {
  double wr[6], iw[4];
};
__device__ __forceinline__ void roll_load_c(const Layout& L, const LayoutW& LW, const double* sm, int k, RollLoadsC& r) {
#pragma unroll
  for (int i = 0; i < 6; ++i) r.wr[i] = sm[LW.WR + 6 * k + i];
#pragma unroll
  for (int i = 0; i < 4; ++i) r.iw[i] = sm[L.CV + 8 * k + 4 + i];
}
__device__ __forceinline__ void roll_load_w(const Layout& L, const LayoutW& LW, const double* sm, const double* KD, int k,
                                            int row, int wi, RollLoadsW& r) {
#pragma unroll
  for (int i = 0; i < 13; ++i) r.xo[i] = sm[L.X + 13 * k + i];
#pragma unroll
  for (int i = 0; i < 18; ++i) r.ab[i] = sm[L.AB + kAB * k + i];
  const double* kd = KD + 156 * k + 13 * row;
#pragma unroll
  for (int i = 0; i < 13; ++i) r.kd[i] = kd[i];
  r.wk = sm[LW.WR + 6 * k + wi];
}
// PF: the next knot's gains / old state / Jacobian blocks are loaded one knot ahead into a second register set (90 more
// registers: the one-wave-per-SIMD form); without it the loads sit at the top of the knot (two waves per SIMD hide them)
template <bool PROF, bool PF, int NL = 4, int MD = WM_QUAT>
__device__ inline void rollout_closed_w(const DevParams& P, const Layout& L, const LayoutW& LW, double* sm,
                                        const double* KD, double* ROT, int lane, Prof<PROF>& prof, double alpha = 1.0) {
  typedef Dim<NL> D;
  const int N = P.N;
  const double* cst = sm + L.cst;
  double gb[3], wd0[3];
#pragma unroll
  for (int a = 0; a < 3; ++a) { gb[a] = cst[D::C_GB + a]; wd0[a] = cst[D::C_WD0 + a]; }
  const int row = (lane < 12) ? lane : 0, wi = (lane < 6) ? lane : 0;
  const bool zlane = lane >= 6 && lane < 12;
  double xc[13], xn[13];
#pragma unroll
  for (int i = 0; i < 13; ++i) xc[i] = cst[D::C_X0 + i];
  if (lane == 0)
#pragma unroll
    for (int i = 0; i < 13; ++i) sm[L.Xc + i] = xc[i];
  // one knot: gains / old state / Jacobian blocks in `cur`, the next knot's loaded into `nxt` meanwhile (the loop body is
  // instantiated twice with the roles swapped: no register copies)
  auto knot = [&](int k, RollLoadsW& cur, RollLoadsW& nxt) {
    if (!PF) roll_load_w(L, LW, sm, KD, k, row, wi, cur);
    double dx[12], e[12];
    if (MD == WM_CONVEX) {
#pragma unroll
      for (int r = 0; r < 12; ++r) dx[r] = xc[cvperm(r)] - cur.xo[cvperm(r)];      // internal block order [p, phi, v, w]
    } else {
      QuatModel::state_diff(cur.xo, xc, dx);
    }
    // e = Abar dx
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      e[a] = dx[a] + P.h * dx[6 + a];
      e[3 + a] = (cur.ab[3 * a] * dx[3] + cur.ab[3 * a + 1] * dx[4] + cur.ab[3 * a + 2] * dx[5]) +
                 (cur.ab[9 + 3 * a] * dx[9] + cur.ab[9 + 3 * a + 1] * dx[10] + cur.ab[9 + 3 * a + 2] * dx[11]);
      e[6 + a] = dx[6 + a];
      e[9 + a] = dx[9 + a];
    }
    const double* kd = cur.kd;
    const double p0 = alpha * kd[12] + kd[0] * e[0] + kd[1] * e[1] + kd[2] * e[2];      // alpha scales the feed-forward part
    const double p1 = kd[3] * e[3] + kd[4] * e[4] + kd[5] * e[5];
    const double p2 = kd[6] * e[6] + kd[7] * e[7] + kd[8] * e[8];
    const double p3 = kd[9] * e[9] + kd[10] * e[10] + kd[11] * e[11];
    const double s = (p0 + p1) + (p2 + p3);
    if (zlane) ROT[zeta_slot<NL>(k, lane - 6)] = s;      // the costate of the contact points, for the input recovery
    double wn = (MD == WM_CONVEX) ? s : cur.wk + s;      // (ConvexMpc: the increment alone; the torque part changes frame below)
    tick_dep1(prof, PH_R_GAIN, wn);
    if (PF && k + 1 < N) roll_load_w(L, LW, sm, KD, k + 1, row, wi, nxt);      // one knot ahead
    double w[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) w[i] = read_lane(wn, i);
    tick_dep(prof, PH_R_BCAST, w[0], w[5]);
    if (MD == WM_CONVEX) {
      RollLoadsC rc;
      roll_load_c(L, LW, sm, k, rc);
      const double d3 = rc.iw[0] * w[3] + rc.iw[1] * w[4], d4 = rc.iw[1] * w[3] + rc.iw[2] * w[4], d5 = rc.iw[3] * w[5];
      w[0] += rc.wr[0]; w[1] += rc.wr[1]; w[2] += rc.wr[2];
      w[3] = rc.wr[3] + d3; w[4] = rc.wr[4] + d4; w[5] = rc.wr[5] + d5;
      cv_step_w(P, xc, w, xn);
    } else
    srbd_step_w(P, gb, wd0, xc, w, xn);
#pragma unroll
    for (int i = 0; i < 13; ++i) xc[i] = xn[i];
    if (lane == 0)
#pragma unroll
      for (int i = 0; i < 13; ++i) sm[L.Xc + 13 * (k + 1) + i] = xn[i];
    tick_dep(prof, PH_R_STEP, xc[3], xc[10]);
  };
  if (PF) {
    RollLoadsW ra, rb;
    roll_load_w(L, LW, sm, KD, 0, row, wi, ra);
    for (int k = 0; k < N; k += 2) {
      knot(k, ra, rb);
      if (k + 1 < N) knot(k + 1, rb, ra);
    }
  } else {
    RollLoadsW ra;
    for (int k = 0; k < N; ++k) knot(k, ra, ra);
  }
  QSYNC();
}

// ---- expected decrease of a full step, sum_k d_k' Qu_k (the Armijo test of the reference mode's line search): one lane
// per (knot, contact point); d_l = -D~_l^-1 (V_l' xz + gq_l) is the feed-forward part of the point's step, Qu_l = V_l' y0 + gq_l
// its gradient (y0 from the backward pass, xz = column 12 of [Xz | xz]) ---------------------------------------------------
template <int MD = WM_QUAT, int NL = 4>
__device__ inline double expected_decrease_w(const DevParams& P, const Layout& L, double* sm, const double* KD,
                                             const double* ROT, const double* y0, int lane) {
  typedef Dim<NL> D;
  constexpr int LSH = (NL == 8) ? 3 : 2;
  const int N = P.N;
  const double* cst = sm + L.cst;
  double part = 0.0;
  for (int q = lane; q < NL * N; q += kWave) {
    const int k = q >> LSH, l = q & (NL - 1);
    if (cst[D::C_CON + l] == 0.0) continue;
    const double* rec = ROT + D::ROT * k + 21 * l;
    double T[9], xz[6], yk[6];
#pragma unroll
    for (int i = 0; i < 9; ++i) T[i] = rec[i];
#pragma unroll
    for (int i = 0; i < 6; ++i) { xz[i] = KD[156 * k + 13 * (6 + i) + 12]; yk[i] = y0[6 * k + i]; }
    if (MD == WM_CONVEX) {      // Bw0_l' t = [r_l]x' (Winv_k t)
      double a[3], b[3];
      cv_winv_mul(sm + L.CV + 8 * k, xz + 3, a);
      cv_winv_mul(sm + L.CV + 8 * k, yk + 3, b);
#pragma unroll
      for (int i = 0; i < 3; ++i) { xz[3 + i] = a[i]; yk[3 + i] = b[i]; }
    }
    const double l10 = rec[9], l20 = rec[10], l21 = rec[11], id0 = rec[12], id1 = rec[13], id2 = rec[14];
    const double* bw = sm + L.bw0 + 3 * l;
    double fz[3], fy[3];
#pragma unroll
    for (int b = 0; b < 3; ++b) {
      fz[b] = xz[b] + (bw[b] * xz[3] + bw[D::NU + b] * xz[4] + bw[2 * D::NU + b] * xz[5]);
      fy[b] = yk[b] + (bw[b] * yk[3] + bw[D::NU + b] * yk[4] + bw[2 * D::NU + b] * yk[5]);
    }
    double t[3], qu[3];
#pragma unroll
    for (int b = 0; b < 3; ++b) {
      t[b] = rec[15 + b] + (T[b] * fz[0] + T[3 + b] * fz[1] + T[6 + b] * fz[2]);
      qu[b] = rec[15 + b] + (T[b] * fy[0] + T[3 + b] * fy[1] + T[6 + b] * fy[2]);
    }
    const double y0_ = t[0], y1_ = t[1] - l10 * y0_, y2_ = t[2] - l20 * y0_ - l21 * y1_;
    const double x2 = id2 * y2_;
    const double x1 = id1 * y1_ - l21 * x2;
    const double x0 = id0 * y0_ - l10 * x1 - l20 * x2;
    part -= x0 * qu[0] + x1 * qu[1] + x2 * qu[2];
  }
  return wave_sum(part);
}

// ---- input increments of the trial step, one lane per (knot, contact point):
//      du_l = -T_l D~_l^-1 (V_l' zeta + gq_l),   V_l' zeta = T_l' (zeta_f + Bw0_l' zeta_t) ----
// Returns nonzero (wave-uniform) when an increment is not finite: the trial step is then NOT applied (QMPC_NOT_PD, the
// rule of the lane kernel) -- the step-length reductions that follow drop NaNs silently.
template <int NL = 4, int MD = WM_QUAT>
__device__ inline int recover_inputs_w(const DevParams& P, const Layout& L, double* sm, const double* ROT, int lane,
                                       double alpha = 1.0, const double* zsrc = nullptr) {
  typedef Dim<NL> D;
  constexpr int LSH = (NL == 8) ? 3 : 2;
  const int N = P.N;
  const double* cst = sm + L.cst;
  bool bad = false;
  for (int q = lane; q < NL * N; q += kWave) {
    const int k = q >> LSH, l = q & (NL - 1);
    double du[3] = {0.0, 0.0, 0.0};
    if (cst[D::C_CON + l] != 0.0) {
      const double* rec = ROT + D::ROT * k + 21 * l;
      double T[9], z[6];
#pragma unroll
      for (int i = 0; i < 9; ++i) T[i] = rec[i];
#pragma unroll
      for (int i = 0; i < 6; ++i) z[i] = zsrc ? zsrc[6 * k + i] : ROT[zeta_slot<NL>(k, i)];
      if (MD == WM_CONVEX) {      // Bw0_l' z_t = [r_l]x' (Winv_k z_t)
        double y[3];
        cv_winv_mul(sm + L.CV + 8 * k, z + 3, y);
        z[3] = y[0]; z[4] = y[1]; z[5] = y[2];
      }
      const double l10 = rec[9], l20 = rec[10], l21 = rec[11], id0 = rec[12], id1 = rec[13], id2 = rec[14];
      const double* bw = sm + L.bw0 + 3 * l;
      double f[3];
#pragma unroll
      for (int b = 0; b < 3; ++b) f[b] = z[b] + (bw[b] * z[3] + bw[D::NU + b] * z[4] + bw[2 * D::NU + b] * z[5]);
      double t[3];
#pragma unroll
      for (int b = 0; b < 3; ++b) t[b] = alpha * rec[15 + b] + (T[b] * f[0] + T[3 + b] * f[1] + T[6 + b] * f[2]);
      const double y0 = t[0], y1 = t[1] - l10 * y0, y2 = t[2] - l20 * y0 - l21 * y1;
      const double x2 = id2 * y2;
      const double x1 = id1 * y1 - l21 * x2;
      const double x0 = id0 * y0 - l10 * x1 - l20 * x2;
#pragma unroll
      for (int a = 0; a < 3; ++a) du[a] = -(T[3 * a] * x0 + T[3 * a + 1] * x1 + T[3 * a + 2] * x2);
    }
    double* o = sm + L.dU + D::NU * k + 3 * l;
    o[0] = du[0]; o[1] = du[1]; o[2] = du[2];
    bad = bad || !(isfinite(du[0]) && isfinite(du[1]) && isfinite(du[2]));
  }
  QSYNC();
  return __any(bad ? 1 : 0);
}

// ---- converged mode, one lane per (knot, contact point): the input increments of the trial step (recover_inputs_w), then
// the slack / multiplier directions of the point's six rows and its share of the step lengths (ipm_directions of
// qmpc_kernels.hip: ds = -(a_i . dU_l + rc), dlam = (target - (1 + kappa) s lam - lam ds) / s, fraction to the boundary).
// The rows' loads are issued together, nothing branches, and the ratio tests keep the smallest s / (-ds) as a fraction
// (compared by cross-multiplication): one division per lane instead of two per row.  `kapbits` returns the weakly-active
// flags of the lane's rows (bit 6 j + i: row i of the lane's j-th (knot, point)), read from the DS slot before the
// directions overwrite it; apply_w consumes them.  Returns nonzero (wave-uniform) when an increment is not finite.
template <int NL = 4, int MD = WM_QUAT, bool LEAN = (NL == 8)>
__device__ inline int recover_directions_w(const DevParams& P, const Layout& L, double* sm, double* sl, const double* ROT,
                                           double target, int lane, double* alpha_p, double* alpha_d, double* full_step,
                                           unsigned& kapbits) {
  typedef Dim<NL> D;
  constexpr int LSH = (NL == 8) ? 3 : 2;
  const int N = P.N;
  const double* cst = sm + L.cst;
  const double* cr = cst + D::C_CR;
  bool bad = false;
  double pn = 1.0, pd = 0.0, dn = 1.0, dd = 0.0;      // smallest s / (-ds), lam / (-dlam) so far, as fractions (den 0: none)
  double stp = 0.0;
  unsigned bits = 0;
  int j = 0;
  for (int q = lane; q < NL * N; q += kWave, ++j) {
    const int k = q >> LSH, l = q & (NL - 1);
    const bool stance = cst[D::C_CON + l] != 0.0;
    const double* rec = ROT + D::ROT * k + 21 * l;
    const int i0 = D::NC * k + 6 * l;
    double T[9], z[6], sv[6], lv[6], kap[6], rc[6];
#pragma unroll
    for (int i = 0; i < 9; ++i) T[i] = rec[i];
#pragma unroll
    for (int i = 0; i < 6; ++i) z[i] = ROT[zeta_slot<NL>(k, i)];
    if (MD == WM_CONVEX) {
      double y[3];
      cv_winv_mul(sm + L.CV + 8 * k, z + 3, y);
      z[3] = y[0]; z[4] = y[1]; z[5] = y[2];
    }
    const double l10 = rec[9], l20 = rec[10], l21 = rec[11], id0 = rec[12], id1 = rec[13], id2 = rec[14];
    const double g0 = rec[15], g1 = rec[16], g2 = rec[17];
#pragma unroll
    for (int i = 0; i < 6; ++i) {
      sv[i] = sl[L.S + i0 + i]; lv[i] = sl[L.LAM + i0 + i]; rc[i] = sl[L.RC + i0 + i];
      kap[i] = LEAN ? (((kapbits >> (6 * j + i)) & 1u) ? 1.0 : 0.0) : sl[L.DS + i0 + i];
    }
    const double* bw = sm + L.bw0 + 3 * l;
    double du[3] = {0.0, 0.0, 0.0};
    if (stance) {
      double f[3];
#pragma unroll
      for (int b = 0; b < 3; ++b) f[b] = z[b] + (bw[b] * z[3] + bw[D::NU + b] * z[4] + bw[2 * D::NU + b] * z[5]);
      const double t0 = g0 + (T[0] * f[0] + T[3] * f[1] + T[6] * f[2]);
      const double t1 = g1 + (T[1] * f[0] + T[4] * f[1] + T[7] * f[2]);
      const double t2 = g2 + (T[2] * f[0] + T[5] * f[1] + T[8] * f[2]);
      const double y0 = t0, y1 = t1 - l10 * y0, y2 = t2 - l20 * y0 - l21 * y1;
      const double x2 = id2 * y2;
      const double x1 = id1 * y1 - l21 * x2;
      const double x0 = id0 * y0 - l10 * x1 - l20 * x2;
#pragma unroll
      for (int a = 0; a < 3; ++a) du[a] = -(T[3 * a] * x0 + T[3 * a + 1] * x1 + T[3 * a + 2] * x2);
    }
    double* o = sm + L.dU + D::NU * k + 3 * l;
    o[0] = du[0]; o[1] = du[1]; o[2] = du[2];
    bad = bad || !(isfinite(du[0]) && isfinite(du[1]) && isfinite(du[2]));
    stp = fmax(stp, fmax(fabs(du[0]), fmax(fabs(du[1]), fabs(du[2]))));
#pragma unroll
    for (int i = 0; i < 6; ++i) {
      const double jd = cr[3 * i] * du[0] + cr[3 * i + 1] * du[1] + cr[3 * i + 2] * du[2];
      double dsv = -(jd + rc[i]);
      double dlv = (target - (1.0 + kap[i]) * sv[i] * lv[i] - lv[i] * dsv) * fast_rcp(sv[i]);
      dsv = stance ? dsv : 0.0;
      dlv = stance ? dlv : 0.0;
      // sv / (-dsv) < pn / pd  <=>  sv pd < pn (-dsv)   (pd = 0: nothing yet, any candidate wins)
      const bool up = (dsv < 0.0) && (pd == 0.0 || sv[i] * pd < pn * (-dsv));
      pn = up ? sv[i] : pn; pd = up ? -dsv : pd;
      const bool ud = (dlv < 0.0) && (dd == 0.0 || lv[i] * dd < dn * (-dlv));
      dn = ud ? lv[i] : dn; dd = ud ? -dlv : dd;
      if (!LEAN) {
        sl[L.DS + i0 + i] = dsv;
        sl[L.DLAM + i0 + i] = dlv;
      }
      bits |= (stance && kap[i] != 0.0) ? (1u << (6 * j + i)) : 0u;
    }
  }
  const double ap = (pd > 0.0) ? fmin(1.0, P.tau * pn * fast_rcp(pd)) : 1.0;
  const double ad = (dd > 0.0) ? fmin(1.0, P.tau * dn * fast_rcp(dd)) : 1.0;
  *alpha_p = wave_min(ap);
  *alpha_d = wave_min(ad);
  *full_step = wave_max(stp);
  kapbits = bits;
  QSYNC();
  return __any(bad ? 1 : 0);
}

// apply the step to (s, rc, lambda) and leave the weakly-active (Tapia) flags in the DS slot -- ipm_apply of qmpc_kernels.hip
// with the rows of a contact point in one lane
// Eight points: the directions are recomputed here from the trial increment dU (still unscaled: the caller applies before it
// re-rolls a shortened step) and the new flags go back into `kapbits`; sm / target are only read in that form.
template <int NL = 4, bool LEAN = (NL == 8)>
__device__ inline void apply_w(const DevParams& P, const Layout& L, double* sl, double ap, double ad, unsigned conmask,
                               int lane, unsigned& kapbits, double& sl_part, double& rc_part, const double* sm = nullptr,
                               double target = 0.0) {
  typedef Dim<NL> D;
  constexpr int LSH = (NL == 8) ? 3 : 2;
  const int N = P.N;
  sl_part = 0.0;
  rc_part = 0.0;
  const bool full = (ap >= 0.99) && (ad >= 0.99);
  const double rcs = (ap >= 1.0) ? 0.0 : (1.0 - ap);
  int j = 0;
  for (int q = lane; q < NL * N; q += kWave, ++j) {
    const int k = q >> LSH, l = q & (NL - 1);
    if (!(conmask & (1u << l))) continue;
    const int i0 = D::NC * k + 6 * l;
    double s0[6], l0[6], ds[6], dl[6], rc[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) {
      s0[i] = sl[L.S + i0 + i]; l0[i] = sl[L.LAM + i0 + i]; rc[i] = sl[L.RC + i0 + i];
      if (!LEAN) { ds[i] = sl[L.DS + i0 + i]; dl[i] = sl[L.DLAM + i0 + i]; }
    }
    if (LEAN) {
      const double* cr = sm + L.cst + D::C_CR;
      const double* du = sm + L.dU + D::NU * k + 3 * l;
      const double du0 = du[0], du1 = du[1], du2 = du[2];
#pragma unroll
      for (int i = 0; i < 6; ++i) {
        const double kp = ((kapbits >> (6 * j + i)) & 1u) ? 1.0 : 0.0;
        const double jd = cr[3 * i] * du0 + cr[3 * i + 1] * du1 + cr[3 * i + 2] * du2;
        ds[i] = -(jd + rc[i]);
        dl[i] = (target - (1.0 + kp) * s0[i] * l0[i] - l0[i] * ds[i]) * fast_rcp(s0[i]);
      }
    }
    unsigned newbits = 0;
#pragma unroll
    for (int i = 0; i < 6; ++i) {
      const bool kap0 = (kapbits >> (6 * j + i)) & 1u;
      const double s1 = s0[i] + ap * ds[i];
      const double l1 = l0[i] + ad * dl[i];
      const double rc1 = (ap >= 1.0) ? 0.0 : rcs * rc[i];
      sl[L.S + i0 + i] = s1;
      sl[L.RC + i0 + i] = rc1;
      sl[L.LAM + i0 + i] = l1;
      sl_part += s1 * l1;
      rc_part = fmax(rc_part, fabs(rc1));
      const bool sig = full && (s1 < 0.6 * s0[i]) && (l1 < 0.6 * l0[i]) && (kap0 || ((s1 > 0.4 * s0[i]) && (l1 > 0.4 * l0[i])));
      if (LEAN) newbits |= sig ? (1u << (6 * j + i)) : 0u;
      else sl[L.DS + i0 + i] = sig ? 1.0 : 0.0;
    }
    if (LEAN) kapbits = (kapbits & ~(63u << (6 * j))) | newbits;
  }
}

// ---- reference mode: FOUR trial step lengths of the backtracking line search per rollout ------------------------------
// The closed-loop rollout of one trial keeps 12 of the 64 lanes busy (the rows of the wrench-space gains) and every lane
// advances the same state.  The trials of one line search are independent of each other, so each 16-lane row of the
// wavefront rolls its own step length alpha_g = alpha 2^-g out: rows 0..5 of a group own [Xw | xw], rows 6..11 [Xz | xz],
// the knot's six wrench components are broadcast inside the row (DPP row_newbcast), the state cost of the group's
// trajectory is summed on the way.  What a group leaves behind is its costates zeta (ZG, for the inputs) and its states:
// group 0 in Xc itself, groups 1..3 in the S, DLAM and XT slots -- the reference mode uses neither S nor DLAM, and the cost
// expansions in XT are dead between the backward pass and the expansions at the accepted point.
// Returns the state cost sum_k l_k(x_k) of the lane's group.
__device__ __forceinline__ int trial_states_slot(const Layout& L, int g) {      // knots 1..N of group g's trajectory
  return g == 0 ? L.Xc + 13 : (g == 1 ? L.S : (g == 2 ? L.DLAM : L.XT));
}
template <bool PF, int MD = WM_QUAT, int NL = 4>
__device__ inline double rollout_trials_w(const DevParams& P, const Layout& L, const LayoutW& LW, double* sm,
                                          const double* KD, double* ZG, double alpha_g, int lane) {
  typedef typename std::conditional<MD == WM_CONVEX, ConvexModel, typename std::conditional<NL == 8, Quat8Model, QuatModel>::type>::type TM;
  typedef Dim<NL> D;
  const int N = P.N;
  const double* cst = sm + L.cst;
  double gb[3], wd0[3];
#pragma unroll
  for (int a = 0; a < 3; ++a) { gb[a] = cst[D::C_GB + a]; wd0[a] = cst[D::C_WD0 + a]; }
  const int r = lane & 15, g = lane >> 4;
  const int row = (r < 12) ? r : 0, wi = (r < 6) ? r : 0;
  double* zg = ZG + 6 * N * g;
  double* xg = sm + trial_states_slot(L, g);
  double xc[13], xn[13];
#pragma unroll
  for (int i = 0; i < 13; ++i) xc[i] = cst[D::C_X0 + i];
  if (lane == 15)
#pragma unroll
    for (int i = 0; i < 13; ++i) sm[L.Xc + i] = xc[i];
  double Jx = TM::knot_cost(P, sm + L.refp, sm + L.uref, 0, xc, nullptr);
  // one knot; PF: the next knot's gains / old state / Jacobian blocks are loaded meanwhile into the other register set
  auto knot = [&](int k, RollLoadsW& cur, RollLoadsW& nxt) {
    if (!PF) roll_load_w(L, LW, sm, KD, k, row, wi, cur);
    double dx[12], e[12];
    if (MD == WM_CONVEX) {
#pragma unroll
      for (int rr = 0; rr < 12; ++rr) dx[rr] = xc[cvperm(rr)] - cur.xo[cvperm(rr)];
    } else {
      QuatModel::state_diff(cur.xo, xc, dx);
    }
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      e[a] = dx[a] + P.h * dx[6 + a];
      e[3 + a] = (cur.ab[3 * a] * dx[3] + cur.ab[3 * a + 1] * dx[4] + cur.ab[3 * a + 2] * dx[5]) +
                 (cur.ab[9 + 3 * a] * dx[9] + cur.ab[9 + 3 * a + 1] * dx[10] + cur.ab[9 + 3 * a + 2] * dx[11]);
      e[6 + a] = dx[6 + a];
      e[9 + a] = dx[9 + a];
    }
    const double* kd = cur.kd;
    const double p0 = alpha_g * kd[12] + kd[0] * e[0] + kd[1] * e[1] + kd[2] * e[2];
    const double p1 = kd[3] * e[3] + kd[4] * e[4] + kd[5] * e[5];
    const double p2 = kd[6] * e[6] + kd[7] * e[7] + kd[8] * e[8];
    const double p3 = kd[9] * e[9] + kd[10] * e[10] + kd[11] * e[11];
    const double s = (p0 + p1) + (p2 + p3);
    const double wn = (MD == WM_CONVEX) ? s : cur.wk + s;
    if (r >= 6 && r < 12) zg[6 * k + r - 6] = s;
    if (PF && k + 1 < N) roll_load_w(L, LW, sm, KD, k + 1, row, wi, nxt);
    double w[6];
    w[0] = dpp_mov<0x150>(wn); w[1] = dpp_mov<0x151>(wn); w[2] = dpp_mov<0x152>(wn);      // row_newbcast:0..5
    w[3] = dpp_mov<0x153>(wn); w[4] = dpp_mov<0x154>(wn); w[5] = dpp_mov<0x155>(wn);
    if (MD == WM_CONVEX) {
      RollLoadsC rc;
      roll_load_c(L, LW, sm, k, rc);
      const double d3 = rc.iw[0] * w[3] + rc.iw[1] * w[4], d4 = rc.iw[1] * w[3] + rc.iw[2] * w[4], d5 = rc.iw[3] * w[5];
      w[0] += rc.wr[0]; w[1] += rc.wr[1]; w[2] += rc.wr[2];
      w[3] = rc.wr[3] + d3; w[4] = rc.wr[4] + d4; w[5] = rc.wr[5] + d5;
      cv_step_w(P, xc, w, xn);
    } else
    srbd_step_w(P, gb, wd0, xc, w, xn);
#pragma unroll
    for (int i = 0; i < 13; ++i) xc[i] = xn[i];
    if (r == 15)
#pragma unroll
      for (int i = 0; i < 13; ++i) xg[13 * k + i] = xn[i];
    Jx += TM::knot_cost(P, sm + L.refp, sm + L.uref, k + 1, xc, nullptr);
  };
  if (PF) {
    RollLoadsW ra, rb;
    roll_load_w(L, LW, sm, KD, 0, row, wi, ra);
    for (int k = 0; k < N; k += 2) {
      knot(k, ra, rb);
      if (k + 1 < N) knot(k + 1, rb, ra);
    }
  } else {
    RollLoadsW ra;
    for (int k = 0; k < N; ++k) knot(k, ra, ra);
  }
  QSYNC();
  return Jx;
}

// the inputs' share of the four trials' merit, one lane per (knot, contact point): du_l(alpha_g) as in recover_inputs_w from
// the group's costates, u = U + du, then the input cost, the augmented-Lagrangian terms max(lambda + rho c, 0)^2 - lambda^2
// and the violation max(c, 0) of the point's cone rows (the arithmetic of ref_merit in qmpc_ref.hip).  Per-lane partial sums:
// Ju = input cost, mer = Ju + (augmented-Lagrangian terms) / (2 rho), vi = violation.
template <int MD = WM_QUAT, int NL = 4>
__device__ inline void trial_inputs_w(const DevParams& P, const Layout& L, const double* sm, const double* sl,
                                      const double* ROT, const double* ZG, const double* Rl, double alpha, double rho,
                                      int lane, double Ju[4], double mer[4], double vi[4]) {
  typedef Dim<NL> D;
  constexpr int LSH = (NL == 8) ? 3 : 2;
  const int N = P.N;
  const double* cst = sm + L.cst;
  const double* cr = cst + D::C_CR;
  double al[4];
#pragma unroll
  for (int g = 0; g < 4; ++g) { Ju[g] = 0.0; al[g] = 0.0; vi[g] = 0.0; }
  for (int q = lane; q < NL * N; q += kWave) {
    const int k = q >> LSH, l = q & (NL - 1);
    const bool stance = cst[D::C_CON + l] != 0.0;
    const double* rec = ROT + D::ROT * k + 21 * l;
    const double* bw = sm + L.bw0 + 3 * l;
    double u0[3], ur[3], Rw[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      u0[a] = sm[L.U + D::NU * k + 3 * l + a];
      ur[a] = sm[L.uref + 3 * l + a];
      Rw[a] = Rl[3 * l + a];      // input weights from LDS (a lane-dependent index into the kernel arguments is a waterfall loop)
    }
    double T[9], lam[6];
#pragma unroll
    for (int i = 0; i < 9; ++i) T[i] = stance ? rec[i] : 0.0;
#pragma unroll
    for (int i = 0; i < 6; ++i) lam[i] = sl[L.LAM + D::NC * k + 6 * l + i];
    const double l10 = rec[9], l20 = rec[10], l21 = rec[11], id0 = rec[12], id1 = rec[13], id2 = rec[14];
    const double fzc = -P.fz_max * cst[D::C_CON + l];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const double ag = alpha * (g == 0 ? 1.0 : (g == 1 ? 0.5 : (g == 2 ? 0.25 : 0.125)));
      double u[3] = {u0[0], u0[1], u0[2]};
      if (stance) {
        const double* zs = ZG + 6 * N * g + 6 * k;
        double z[6];
#pragma unroll
        for (int i = 0; i < 6; ++i) z[i] = zs[i];
        if (MD == WM_CONVEX) {
          double y[3];
          cv_winv_mul(sm + L.CV + 8 * k, z + 3, y);
          z[3] = y[0]; z[4] = y[1]; z[5] = y[2];
        }
        double f[3];
#pragma unroll
        for (int b = 0; b < 3; ++b) f[b] = z[b] + (bw[b] * z[3] + bw[D::NU + b] * z[4] + bw[2 * D::NU + b] * z[5]);
        double t[3];
#pragma unroll
        for (int b = 0; b < 3; ++b) t[b] = ag * rec[15 + b] + (T[b] * f[0] + T[3 + b] * f[1] + T[6 + b] * f[2]);
        const double y0 = t[0], y1 = t[1] - l10 * y0, y2 = t[2] - l20 * y0 - l21 * y1;
        const double x2 = id2 * y2;
        const double x1 = id1 * y1 - l21 * x2;
        const double x0 = id0 * y0 - l10 * x1 - l20 * x2;
#pragma unroll
        for (int a = 0; a < 3; ++a) u[a] = u0[a] + (-(T[3 * a] * x0 + T[3 * a + 1] * x1 + T[3 * a + 2] * x2));
      }
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
  for (int g = 0; g < 4; ++g) mer[g] = Ju[g] + al[g] * i2r;
}

// ---- reference mode: |grad_U L_A|_inf at (X, U) (ref_stationarity of qmpc_ref.hip) through the block structure of the
// transition.  Lane c < 12 of every 16-lane row owns component c of the costate y_k = lx_k + Abar_k' y_{k+1}:
//   position rows keep y,  attitude rows take A1' y_att,  velocity rows add h y_pos,  rate rows add A3' y_att
// -- three row broadcasts and one row shift per knot, the lane's three coefficients come from LDS through per-lane
// addresses (lanes without a term read a 0.0 slot).  Bbar_k' y_{k+1} = Wr' (M_k' y_{k+1}) leaves a 6-vector per knot in `my`
// (lanes 0..2 the force part, 3..5 the torque part); the gradient rows are then formed one lane per (knot, contact point):
//   gu = R (u - u_ref) + c_l mf + Bw0_l' mt + sum_i max(lambda_i + rho c_i, 0) a_i.
#ifndef QMPC_STAT_ATTR
#define QMPC_STAT_ATTR inline
#endif
template <int MD = WM_QUAT, int NL = 4>
__device__ QMPC_STAT_ATTR double stationarity_w(const DevParams& P, const Layout& L, double* sm, const double* sl, double* my,
                                        const double* Rl, double rho, unsigned conmask, int lane) {
  typedef Dim<NL> D;
  constexpr int LSH = (NL == 8) ? 3 : 2;
  const int N = P.N;
  const double* cst = sm + L.cst;
  const double* cr = cst + D::C_CR;
  const int c = lane & 15, tp = c / 3, j = c - 3 * tp;
  const int zero = L.cst + kZeroSlotsT<NL>();
  // per-lane operand addresses at knot 0 and their stride per knot
  const bool rot = (tp == 1 || tp == 3) && c < 12;
  const int ia = rot ? L.AB + (tp == 1 ? 0 : 9) + j : zero, sa = rot ? kAB : 0;          // A1(:, j) / A3(:, j): + 3 r
  const bool tq = tp == 1;
  const int iw = tq ? L.AB + 18 + j : zero, sw = tq ? kAB : 0;                              // W(:, j): + 3 r
  const int da = rot ? 3 : 0, dw = tq ? 3 : 0;
  const int il = (c < 12) ? L.XT + 9 + c : zero, sl_ = (c < 12) ? kXT : 0;
  const double self = (c < 12 && tp != 1) ? 1.0 : 0.0, hsh = (tp == 2 && c < 12) ? P.h : 0.0;
  const double cpf = P.h * (P.hh * P.inv_mass), cvf = P.h * P.inv_mass;
  const double mself = (tp == 0) ? cpf : 0.0, mshl = (tp == 0) ? cvf : (tq ? P.h : 0.0),
               mw = tq ? ((MD == WM_CONVEX) ? P.h * P.hh : P.h * (0.5 * P.hh)) : 0.0;
  double y = sm[il + sl_ * N];
  for (int k = N - 1; k >= 0; --k) {
    const double a0 = sm[ia + sa * k], a1 = sm[ia + sa * k + da], a2 = sm[ia + sa * k + 2 * da];
    const double w0 = sm[iw + sw * k], w1 = sm[iw + sw * k + dw], w2 = sm[iw + sw * k + 2 * dw];
    const double lx = sm[il + sl_ * k];
    const double b3 = dpp_mov<0x153>(y), b4 = dpp_mov<0x154>(y), b5 = dpp_mov<0x155>(y);      // row_newbcast:3..5
    const double up = dpp_mov<0x106>(y);                                                       // row_shl:6: lane c reads c + 6
    const double dn = dpp_mov<0x116>(y);                                                       // row_shr:6: lane c reads c - 6
    const double m = fma(mself, y, mshl * up) + mw * (w0 * b3 + w1 * b4 + w2 * b5);
    if (lane < 6) my[6 * k + lane] = m;
    y = lx + (self * y + hsh * dn + (a0 * b3 + a1 * b4 + a2 * b5));
  }
  QSYNC();
  double g = 0.0;
  for (int q = lane; q < NL * N; q += kWave) {
    const int k = q >> LSH, l = q & (NL - 1);
    if (!(conmask & (1u << l))) continue;
    double m[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) m[i] = my[6 * k + i];
    if (MD == WM_CONVEX) {
      double y[3];
      cv_winv_mul(sm + L.CV + 8 * k, m + 3, y);
      m[3] = y[0]; m[4] = y[1]; m[5] = y[2];
    }
    const double* bw = sm + L.bw0 + 3 * l;
    const double con = cst[D::C_CON + l];
    double zp[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) {
      const int idx = D::NC * k + 6 * l + i;
      zp[i] = fmax(sl[L.LAM + idx] + rho * sl[L.RC + idx], 0.0);
    }
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      const int jj = 3 * l + a;
      double gu = Rl[jj] * (sm[L.U + D::NU * k + jj] - sm[L.uref + jj]) + con * m[a] +
                  (bw[a] * m[3] + bw[D::NU + a] * m[4] + bw[2 * D::NU + a] * m[5]);
#pragma unroll
      for (int i = 0; i < 6; ++i) gu += zp[i] * cr[3 * i + a];
      g = fmax(g, fabs(gu));
    }
  }
  return wave_max(g);
}

}  // namespace qmpc
