// qmpc_kernels.hip -- gfx950 kernels of the batched MPC inner loop.
//
// One wavefront (64 lanes, one workgroup) owns one MPC instance.  The solve is the converged mode of
// include/qmpc.h: a primal-dual interior-point iteration whose Newton systems are solved with the
// iLQR/Riccati recursion over the horizon (12-dim error state, 3 NL inputs), i.e. the same backward /
// forward pass structure as the reference's external AL-iLQR solver (QuatMpc.cpp:218-256), with the cone
// rows (QuatMpc.cpp:194-215) handled by barrier weights instead of penalties.
//
// Everything that differs between the reference's two controllers and the 8-contact-point stand-in of
// BASELINE config 5 sits in a model policy (qmpc_device.h: QuatModelT<NL>, ConvexModel); this file is the
// shared core, templated on it:
//   setup_instance      record -> constants, reference, initial guess
//   expansions          one lane per knot: compact Jacobian record + cost expansion
//   rotation_prepass    one lane per (knot, contact point): frame T_l, rotated block, gradient
//   backward_pass       every 12x12 (+1 gradient column) matrix lives in REGISTERS in the FP64 MFMA fragment
//                       layout (lane 16g+c holds rows {g,4+g,8+g} of column c), so the v_mfma_f64_16x16x4_f64
//                       products of a knot chain without any LDS round trip; Gauss-Jordan stage solve with
//                       cross-lane moves (DPP row broadcast, ds_bpermute, v_readlane) in the same layout
//   rollout_*           nonlinear rollouts; rotated gains applied inside lane quads
//   ipm_directions / ipm_apply   slack / multiplier bookkeeping over the cone rows
//   qmpc_solve_kernel<MD, PROF, VAR>   VAR: where the gains and the slack arrays live (LDS / workspace)
//   qmpc_linearize_kernel, qmpc_leg_kernel, self-test kernels
#include "qmpc_device.h"

#ifndef QMPC_V2_WAVES
#define QMPC_V2_WAVES 2
#endif
#ifndef QMPC_PF_K
#define QMPC_PF_K true   // prefetch the next knot's gain row / frame in the rollout (all variants)
#endif
#ifndef QMPC_PIPE_ALL
#define QMPC_PIPE_ALL false  // pipelined operand build in the workspace variants: measured slower (register pressure)
#endif
#ifndef QMPC_LEANOPS
#define QMPC_LEANOPS true    // workspace variants: h Bw0 entries re-read from LDS in the operand build (register pressure)
#endif
#ifndef QMPC_NL8_WAVES
#define QMPC_NL8_WAVES 2
#endif

namespace qmpc {

// Optional phase-level cycle accounting (s_memtime), compiled in only for the
// diagnostic instantiation qmpc_solve_kernel<true>.
enum { PH_SETUP = 0, PH_EXPAND, PH_BUILD, PH_MFMA, PH_SOLVE, PH_PUPD, PH_DIRS, PH_ROLL, PH_MISC,
       PH_PREPASS, PH_R_GAIN, PH_R_BCAST, PH_R_STEP, PH_APPLY, PH_DRAIN, PH_COUNT };   // 9.. : finer split, diagnostics only
template <bool PROF>
struct Prof {
  long long t[PH_COUNT];
  long long last;
  __device__ __forceinline__ void start() {
    if (PROF) {
#pragma unroll
      for (int i = 0; i < PH_COUNT; ++i) t[i] = 0;
      last = clock64();
    }
  }
  __device__ __forceinline__ void tick(int ph) {
    if (PROF) {
      const long long now = clock64();
      t[ph] += now - last;
      last = now;
    }
  }
};

// ---- instance set-up: record -> LDS constants, reference, initial guess -----
// model-specific part: raw[48] (the record, in LDS scratch) -> cst, refp, uref, bw0
template <class MD>
__device__ inline void model_setup(const DevParams& P, const Layout& L, double* sm, const double* raw, int lane, int nc);

// QuatMpc record (qmpc_input / qmpc_input8): quat 0..3, rot 4..12, linvel 13..15, angvel 16..18,
//      foot 19.., contacts, posref, velref, accref, quat_d (offsets Dim<NL>::R_*)
template <int NL>
__device__ inline void quat_setup(const DevParams& P, const Layout& L, double* sm, const double* raw, int lane,
                                  int nc) {
  typedef Dim<NL> D;
  double* cst = sm + L.cst;
  if (lane < D::NU) cst[D::C_FOOT + lane] = raw[D::R_FOOT + lane];
  if (lane < NL) cst[D::C_CON + lane] = (raw[D::R_CON + lane] != 0.0) ? 1.0 : 0.0;
  if (lane < 3) {
    // g_body = R' (0,0,-9.81)  (AltroUtils.cpp:368-371)
    cst[D::C_GB + lane] = raw[4 + 6 + lane] * (-9.81);
  }
  if (lane < 13) {
    // x_init (QuatMpc.cpp:231-246; angular velocity dropped by the ';' at :242)
    double x0 = 0.0;
    if (lane >= 3 && lane < 7) x0 = raw[lane - 3];
    else if (lane >= 7 && lane < 10) x0 = raw[13 + lane - 7];
    else if (lane >= 10) x0 = P.drop_ang_vel ? 0.0 : raw[16 + lane - 10];
    cst[D::C_X0 + lane] = x0;
    // reference parameters: pos vel acc quat_d
    sm[L.refp + lane] = (lane < 9) ? raw[D::R_POS + lane] : raw[D::R_QD + lane - 9];
  }
  if (lane < D::NU) {
    // u_ref (QuatMpc.cpp:118-125)
    const int l = lane / 3, a = lane - 3 * l;
    sm[L.uref + lane] = (a == 2) ? raw[D::R_CON + l] * P.mass * 9.81 / (double)nc : 0.0;
  }
  if (lane < 18) {
    // C_mat * R (QuatMpc.cpp:47-52,203): rows (1,0,-mu),(-1,0,-mu),(0,1,-mu),(0,-1,-mu),(0,0,1),(0,0,-1)
    const int r = lane / 3, c = lane - 3 * r;
    const double C0 = (r == 0) ? 1.0 : (r == 1 ? -1.0 : 0.0);
    const double C1 = (r == 2) ? 1.0 : (r == 3 ? -1.0 : 0.0);
    const double C2 = (r < 4) ? -P.mu : (r == 4 ? 1.0 : -1.0);
    cst[D::C_CR + lane] = C0 * raw[4 + c] + C1 * raw[4 + 3 + c] + C2 * raw[4 + 6 + c];
  }
  QSYNC();
  if (lane < 3) {
    // wd0 = Iinv * (c x 5.204 g_body)  (AltroUtils.cpp:373-374,391)
    const double com[3] = {0.0223, 0.002, -0.0005};
    const double fg[3] = {5.204 * cst[D::C_GB], 5.204 * cst[D::C_GB + 1], 5.204 * cst[D::C_GB + 2]};
    const double mg[3] = {com[1] * fg[2] - com[2] * fg[1], com[2] * fg[0] - com[0] * fg[2],
                          com[0] * fg[1] - com[1] * fg[0]};
    cst[D::C_WD0 + lane] = P.Iinv[3 * lane] * mg[0] + P.Iinv[3 * lane + 1] * mg[1] + P.Iinv[3 * lane + 2] * mg[2];
  }
  for (int i = lane; i < 3 * D::NU; i += kWave) {
    // Bw0 = Iinv * skew(r_l) * contact_l  (AltroUtils.cpp:431-434), 3 x NU row-major
    const int a = i / D::NU, col = i - D::NU * a, l = col / 3, b = col - 3 * l;
    const double* r = cst + D::C_FOOT + 3 * l;
    // skew(r) column b: b=0:(0, r2, -r1)  b=1:(-r2, 0, r0)  b=2:(r1, -r0, 0)
    double s0, s1, s2;
    if (b == 0) { s0 = 0.0; s1 = r[2]; s2 = -r[1]; }
    else if (b == 1) { s0 = -r[2]; s1 = 0.0; s2 = r[0]; }
    else { s0 = r[1]; s1 = -r[0]; s2 = 0.0; }
    sm[L.bw0 + i] = cst[D::C_CON + l] * (P.Iinv[3 * a] * s0 + P.Iinv[3 * a + 1] * s1 + P.Iinv[3 * a + 2] * s2);
  }
  QSYNC();
}
template <>
__device__ inline void model_setup<QuatModel>(const DevParams& P, const Layout& L, double* sm, const double* raw,
                                              int lane, int nc) {
  quat_setup<4>(P, L, sm, raw, lane, nc);
}
template <>
__device__ inline void model_setup<Quat8Model>(const DevParams& P, const Layout& L, double* sm, const double* raw,
                                               int lane, int nc) {
  quat_setup<8>(P, L, sm, raw, lane, nc);
}

// ConvexMpc record (qmpc_convex_input): euler 0..2, pos 3..5, angvel 6..8, linvel 9..11, foot 12..23,
//      contacts 24..27, pos_d 28..30, lin_vel_d 31..33, yaw_rate_d 34
template <>
__device__ inline void model_setup<ConvexModel>(const DevParams& P, const Layout& L, double* sm, const double* raw,
                                                int lane, int nc) {
  double* cst = sm + L.cst;
  if (lane < 12) cst[C_FOOT + lane] = raw[12 + lane];
  if (lane < 4) cst[C_CON + lane] = (raw[24 + lane] != 0.0) ? 1.0 : 0.0;
  if (lane < 13) {
    cst[C_X0 + lane] = (lane < 12) ? raw[lane] : 0.0;      // x_init, ConvexMpc.cpp:156-167
    double rp = 0.0;                                       // yaw0, yaw_rate_d, pos_d(3), vx_d, vy_d
    if (lane == CR_YAW) rp = raw[2];
    else if (lane == CR_RATE) rp = raw[34];
    else if (lane >= CR_POS && lane < CR_POS + 3) rp = raw[28 + lane - CR_POS];
    else if (lane == CR_VX) rp = raw[31];
    else if (lane == CR_VY) rp = raw[32];
    sm[L.refp + lane] = rp;
  }
  if (lane < 12) {
    // u_ref (ConvexMpc.cpp:107-110)
    const int l = lane / 3, a = lane - 3 * l;
    sm[L.uref + lane] = (a == 2) ? P.mass * 9.81 / (double)nc * raw[24 + l] : 0.0;
  }
  if (lane < 18) {
    // the pyramid acts on the world-frame forces directly (ConvexMpc.cpp:126-136)
    const int r = lane / 3, c = lane - 3 * r;
    const double C0 = (r == 0) ? 1.0 : (r == 1 ? -1.0 : 0.0);
    const double C1 = (r == 2) ? 1.0 : (r == 3 ? -1.0 : 0.0);
    const double C2 = (r < 4) ? -P.mu : (r == 4 ? 1.0 : -1.0);
    cst[C_CR + lane] = (c == 0) ? C0 : (c == 1 ? C1 : C2);
  }
  QSYNC();
  if (lane < 36) {
    // S = skew(r_l) * contact_l (AltroUtils.cpp:284-286), 3x12 row-major: tau = S u
    const int a = lane / 12, col = lane - 12 * a, l = col / 3, b = col - 3 * l;
    const double* r = cst + C_FOOT + 3 * l;
    double s0, s1, s2;
    if (b == 0) { s0 = 0.0; s1 = r[2]; s2 = -r[1]; }
    else if (b == 1) { s0 = -r[2]; s1 = 0.0; s2 = r[0]; }
    else { s0 = r[1]; s1 = -r[0]; s2 = 0.0; }
    sm[L.bw0 + lane] = cst[C_CON + l] * (a == 0 ? s0 : (a == 1 ? s1 : s2));
  }
  QSYNC();
}

template <class MD>
__device__ inline void setup_instance(const DevParams& P, const Layout& L, double* sm,
                                      const void* in, int lane, int* status) {
  typedef typename MD::D D;
  constexpr bool quat = (MD::NX == 13);
  constexpr int REC = quat ? D::REC : 48;          // doubles in one record
  const double* rec = reinterpret_cast<const double*>(in);
  // one coalesced 8-byte-per-lane read of the record (48 or 64 doubles)
  const double v = (lane < REC) ? rec[lane] : 0.0;
  const unsigned long long bad = __ballot(lane < REC && !isfinite(v));
  double* raw = sm + L.tile;  // scratch
  if (lane < REC) raw[lane] = v;
  QSYNC();
  constexpr int con0 = quat ? D::R_CON : 24;       // contacts[] inside the record
  int nc = 0;
  for (int l = 0; l < MD::NL; ++l) nc += (raw[con0 + l] != 0.0) ? 1 : 0;
  *status = bad ? QMPC_NAN_INPUT : (nc == 0 ? QMPC_NO_CONTACT : QMPC_OK);
  if (*status != QMPC_OK) return;
  model_setup<MD>(P, L, sm, raw, lane, nc);
}

// open-loop rollout of U from x0 into X (every lane computes it redundantly;
// lane 0 publishes).  ALTRO's initial rollout.
template <class MD, bool LEAN>
__device__ inline void rollout_open(const DevParams& P, const Layout& L, double* sm, int lane) {
  typedef typename MD::D D;
  const double* cst = sm + L.cst;
  typename MD::template RegsT<LEAN> M;
  M.load(cst, sm + L.bw0);
  double x[13], xn[13], u[D::NU];
#pragma unroll
  for (int i = 0; i < 13; ++i) x[i] = cst[D::C_X0 + i];
  if (lane == 0)
#pragma unroll
    for (int i = 0; i < 13; ++i) sm[L.X + i] = x[i];
  for (int k = 0; k < P.N; ++k) {
#pragma unroll
    for (int j = 0; j < D::NU; ++j) u[j] = sm[L.U + D::NU * k + j];
    MD::template step<LEAN>(P, M, x, u, xn);
#pragma unroll
    for (int i = 0; i < 13; ++i) x[i] = xn[i];
    if (lane == 0)
#pragma unroll
      for (int i = 0; i < 13; ++i) sm[L.X + 13 * (k + 1) + i] = xn[i];
  }
  QSYNC();
}

// expansions at (X,U): one lane per knot, or EXPAND_PARTS lanes per knot (knots and parts are independent)
template <class MD>
__device__ inline void expansions(const DevParams& P, const Layout& L, double* sm, int lane) {
  typedef typename MD::D D;
  const int N = P.N;
  if (MD::EXPAND_PARTS == 4 && N <= 15) {     // one round of 4 (N + 1) lanes; longer horizons gain nothing from two
    const int part = lane & 3;
    {
      const int k = lane >> 2;
      if (k <= N) {
        double x[13], xn[13], u[D::NU];
#pragma unroll
        for (int i = 0; i < 13; ++i) {
          x[i] = sm[L.X + 13 * k + i];
          xn[i] = (k < N) ? sm[L.X + 13 * (k + 1) + i] : 0.0;
        }
#pragma unroll
        for (int j = 0; j < D::NU; ++j) u[j] = (k < N) ? sm[L.U + D::NU * k + j] : 0.0;
        MD::expand_part(P, sm + L.cst, sm + L.bw0, sm + L.refp, k, part, x, u, xn, sm + L.AB + kAB * k,
                        sm + L.XT + kXT * k);
      }
    }
    QSYNC();
    return;
  }
  if (lane <= N) {
    double x[13], xn[13], u[D::NU];
#pragma unroll
    for (int i = 0; i < 13; ++i) {
      x[i] = sm[L.X + 13 * lane + i];
      xn[i] = (lane < N) ? sm[L.X + 13 * (lane + 1) + i] : 0.0;
    }
#pragma unroll
    for (int j = 0; j < D::NU; ++j) u[j] = (lane < N) ? sm[L.U + D::NU * lane + j] : 0.0;
    double AB[27], lx[12], lxx[9];
    MD::expand(P, sm + L.cst, sm + L.bw0, sm + L.refp, lane, x, u, xn, AB, lx, lxx);
    if (lane < N)
#pragma unroll
      for (int i = 0; i < 27; ++i) sm[L.AB + kAB * lane + i] = AB[i];
#pragma unroll
    for (int i = 0; i < 9; ++i) sm[L.XT + kXT * lane + i] = lxx[i];
#pragma unroll
    for (int i = 0; i < 12; ++i) sm[L.XT + kXT * lane + 9 + i] = lx[i];
  }
  QSYNC();
}

// cone value of row idx = NC k + 6 l + i at the current inputs:
// c = C R u_l + b  (QuatMpc.cpp:194-205)
template <class D>
__device__ __forceinline__ double cone_value(const DevParams& P, const Layout& L, const double* sm, int idx) {
  const int k = idx / D::NC, row = idx - D::NC * k, l = row / 6, i = row - 6 * l;
  const double* cr = sm + L.cst + D::C_CR + 3 * i;
  const double* u = sm + L.U + D::NU * k + 3 * l;
  double c = cr[0] * u[0] + cr[1] * u[1] + cr[2] * u[2];
  if (i == 4) c += -P.fz_max * sm[L.cst + D::C_CON + l];
  return c;
}

// Pre-pass over all (knot, leg) pairs, one lane each: rotation T_l of the leg's
// input coordinates (q1 along the heaviest cone row, q2 the Gram-Schmidt
// complement of the second heaviest non-parallel row, q3 = q1 x q2; see
// DESIGN.md "rotated stage solve") and the blocks the backward pass adds in the
// rotated coordinates:
//   Dblk = T' R_l T + sum_i w_i (T'a_i)(T'a_i)',   w_i = lam_i / s_i
//   gq   = T' (R_l (u_l - uref_l)) + sum_i g_i (T'a_i),  g_i = target/s_i - kappa_i lam_i + w_i rc_i
// ROT record per leg: T (9, row-major [a][b]), Dblk (9), gq (3).
// AL = true (reference mode, qmpc_ref.hip): the rows carry augmented-Lagrangian weights instead of barrier weights,
//   w_i = rho [lam_i + rho c_i > 0],  g_i = max(lam_i + rho c_i, 0)   (SURVEY.md Appendix B),
// with c_i kept in the RC slot and `target` = rho.
template <class D, bool AL = false>
__device__ inline void rotation_prepass(const DevParams& P, const Layout& L, double* sm, const double* sl,
                                        double* ROT, double target, int lane) {
  const int N = P.N;
  const double* cst = sm + L.cst;
  double cr[18];
#pragma unroll
  for (int i = 0; i < 18; ++i) cr[i] = cst[D::C_CR + i];
  for (int q = lane; q < D::NLEG * N; q += kWave) {
    const int k = q / D::NLEG, l = q - D::NLEG * k;
    double* out = ROT + D::ROT * k + 21 * l;
    const int l4 = l & 3;     // R holds 12 weights: input j uses R[j % 12]
    const double R0 = P.R[3 * l4], R1 = P.R[3 * l4 + 1], R2 = P.R[3 * l4 + 2];
    if (cst[D::C_CON + l] == 0.0) {
      // swing leg: identity frame, block = R, zero gradient (forces pinned to 0)
      out[0] = 1; out[1] = 0; out[2] = 0; out[3] = 0; out[4] = 1; out[5] = 0; out[6] = 0; out[7] = 0; out[8] = 1;
      out[9] = R0; out[10] = 0; out[11] = 0; out[12] = 0; out[13] = R1; out[14] = 0; out[15] = 0; out[16] = 0; out[17] = R2;
      out[18] = 0; out[19] = 0; out[20] = 0;
      continue;
    }
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
        const double kap = sl[L.DS + D::NC * k + 6 * l + i];     // weakly-active flag (see ipm_apply)
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
    double Db[9], gq[3];
    const double Rl[3] = {R0, R1, R2};
    const double* u = sm + L.U + D::NU * k + 3 * l;
    const double* ur = sm + L.uref + 3 * l;
    const double ru[3] = {R0 * (u[0] - ur[0]), R1 * (u[1] - ur[1]), R2 * (u[2] - ur[2])};
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      gq[a] = T[a] * ru[0] + T[3 + a] * ru[1] + T[6 + a] * ru[2];
#pragma unroll
      for (int b = 0; b < 3; ++b)
        Db[3 * a + b] = T[a] * Rl[0] * T[b] + T[3 + a] * Rl[1] * T[3 + b] + T[6 + a] * Rl[2] * T[6 + b];
    }
#pragma unroll
    for (int i = 0; i < 6; ++i) {
      double at[3];  // rotated row: at[b] = sum_a T[a][b] a_i[a]
#pragma unroll
      for (int b = 0; b < 3; ++b) at[b] = T[b] * cr[3 * i] + T[3 + b] * cr[3 * i + 1] + T[6 + b] * cr[3 * i + 2];
#pragma unroll
      for (int a = 0; a < 3; ++a) {
        gq[a] += gi[i] * at[a];
#pragma unroll
        for (int b = 0; b < 3; ++b) Db[3 * a + b] += w[i] * at[a] * at[b];
      }
    }
#pragma unroll
    for (int i = 0; i < 9; ++i) { out[i] = T[i]; out[9 + i] = Db[i]; }
    out[18] = gq[0]; out[19] = gq[1]; out[20] = gq[2];
  }
  QSYNC();
}

// One Gauss-Jordan step on GLOBAL pivot J (input tile tj = J / 12, local index J % 12) of the
// pair (M | Rr) held in the fragment layout, M as TU x TU tiles and Rr as TU tiles: row J is
// eliminated from every OTHER row (the pivot row is left as it is; gj_finish divides by the
// diagonal at the end).
template <int J, int TU>
__device__ __forceinline__ void gj_step_mov(double M[][TU][3], double Rr[][3], int c, int g, double& minpiv) {
  constexpr int tj = J / 12, Jl = J % 12, ej = Jl >> 2, gj = Jl & 3;
  const int src = (gj << 4) | c;
  double mrow[TU];
#pragma unroll
  for (int t = 0; t < TU; ++t) mrow[t] = __shfl(M[tj][t][ej], src);   // row J, same column, all row groups
  const double rrow = __shfl(Rr[tj][ej], src);
  const double piv = read_lane(M[tj][tj][ej], (gj << 4) | Jl);
  minpiv = fmin(minpiv, piv);          // positivity is checked once per pass (NaN pivots poison the gains -> NOT_PD below)
  const double ninv = -fast_rcp(piv);
#pragma unroll
  for (int t = 0; t < TU; ++t)
#pragma unroll
    for (int e = 0; e < 3; ++e) {
      const double col = row_bcast<Jl>(M[t][tj][e]);
      const double f = ((t == tj) && (e == ej) && (g == gj)) ? 0.0 : col * ninv;
#pragma unroll
      for (int t2 = 0; t2 < TU; ++t2) M[t][t2][e] = fma(f, mrow[t2], M[t][t2][e]);
      Rr[t][e] = fma(f, rrow, Rr[t][e]);
    }
}
// The rank-one update is issued as v_fmac_f64 with a DPP source: gfx90a+ lets the 64-bit VOP2 ops take
// row_newbcast on src0, so  M[r][c] += M[r][J] * (-row_J[c] / piv)  is ONE instruction per fragment (no
// broadcast move, no multiplier register), and the DPP row mask (one bit per 16-lane row = one bit per row
// group g) leaves the pivot row itself untouched.  The leading s_nop covers the VALU-write -> DPP-read
// hazard, which the compiler does not track through inline asm.
#define QMPC_FMAC_DPP(dst, srcb, mul, JL, MASK) \
  "v_fmac_f64_dpp " dst ", " srcb ", " mul " row_newbcast:" JL " row_mask:" MASK " bank_mask:0xf\n"
template <int J, int TU>
__device__ __forceinline__ void gj_step_dpp64(double M[][TU][3], double Rr[][3], int c, int g, double& minpiv) {
  constexpr int tj = J / 12, Jl = J % 12, ej = Jl >> 2, gj = Jl & 3;
  const int src = (gj << 4) | c;
  double mrow[TU];
#pragma unroll
  for (int t = 0; t < TU; ++t) mrow[t] = __shfl(M[tj][t][ej], src);   // row J, same column, all row groups
  const double rrow = __shfl(Rr[tj][ej], src);
  const double piv = read_lane(M[tj][tj][ej], (gj << 4) | Jl);
  minpiv = fmin(minpiv, piv);          // positivity is checked once per pass (NaN pivots poison the gains -> NOT_PD below)
  const double ninv = -fast_rcp(piv);
  double nr[TU];
#pragma unroll
  for (int t = 0; t < TU; ++t) nr[t] = ninv * mrow[t];
  const double nrr = ninv * rrow;
  constexpr int pm = 0xf ^ (1 << gj);       // every row group but the pivot's
#pragma unroll
  for (int t = 0; t < TU; ++t) {
    if constexpr (TU == 1) {
      asm volatile(
          "s_nop 1\n"
          QMPC_FMAC_DPP("%3", "%0", "%7", "%8", "%9")
          QMPC_FMAC_DPP("%0", "%0", "%6", "%8", "%9")
          QMPC_FMAC_DPP("%4", "%1", "%7", "%8", "%10")
          QMPC_FMAC_DPP("%1", "%1", "%6", "%8", "%10")
          QMPC_FMAC_DPP("%5", "%2", "%7", "%8", "%11")
          QMPC_FMAC_DPP("%2", "%2", "%6", "%8", "%11")
          : "+v"(M[0][0][0]), "+v"(M[0][0][1]), "+v"(M[0][0][2]), "+v"(Rr[0][0]), "+v"(Rr[0][1]), "+v"(Rr[0][2])
          : "v"(nr[0]), "v"(nrr), "n"(Jl), "n"(ej == 0 ? pm : 0xf), "n"(ej == 1 ? pm : 0xf), "n"(ej == 2 ? pm : 0xf));
    } else {
      // source column: tile tj of this row tile; it is updated last (in place)
      double(&Ms)[3] = M[t][tj];
      double(&Mo)[3] = M[t][1 - tj];
      const bool pt = (t == tj);     // the row tile that holds the pivot row
      if (pt) {
        asm volatile(
            "s_nop 1\n"
            QMPC_FMAC_DPP("%6", "%0", "%11", "%12", "%13")
            QMPC_FMAC_DPP("%3", "%0", "%10", "%12", "%13")
            QMPC_FMAC_DPP("%0", "%0", "%9", "%12", "%13")
            QMPC_FMAC_DPP("%7", "%1", "%11", "%12", "%14")
            QMPC_FMAC_DPP("%4", "%1", "%10", "%12", "%14")
            QMPC_FMAC_DPP("%1", "%1", "%9", "%12", "%14")
            QMPC_FMAC_DPP("%8", "%2", "%11", "%12", "%15")
            QMPC_FMAC_DPP("%5", "%2", "%10", "%12", "%15")
            QMPC_FMAC_DPP("%2", "%2", "%9", "%12", "%15")
            : "+v"(Ms[0]), "+v"(Ms[1]), "+v"(Ms[2]), "+v"(Mo[0]), "+v"(Mo[1]), "+v"(Mo[2]), "+v"(Rr[t][0]),
              "+v"(Rr[t][1]), "+v"(Rr[t][2])
            : "v"(nr[tj]), "v"(nr[1 - tj]), "v"(nrr), "n"(Jl), "n"(ej == 0 ? pm : 0xf), "n"(ej == 1 ? pm : 0xf),
              "n"(ej == 2 ? pm : 0xf));
      } else {
        asm volatile(
            "s_nop 1\n"
            QMPC_FMAC_DPP("%6", "%0", "%11", "%12", "0xf")
            QMPC_FMAC_DPP("%3", "%0", "%10", "%12", "0xf")
            QMPC_FMAC_DPP("%0", "%0", "%9", "%12", "0xf")
            QMPC_FMAC_DPP("%7", "%1", "%11", "%12", "0xf")
            QMPC_FMAC_DPP("%4", "%1", "%10", "%12", "0xf")
            QMPC_FMAC_DPP("%1", "%1", "%9", "%12", "0xf")
            QMPC_FMAC_DPP("%8", "%2", "%11", "%12", "0xf")
            QMPC_FMAC_DPP("%5", "%2", "%10", "%12", "0xf")
            QMPC_FMAC_DPP("%2", "%2", "%9", "%12", "0xf")
            : "+v"(Ms[0]), "+v"(Ms[1]), "+v"(Ms[2]), "+v"(Mo[0]), "+v"(Mo[1]), "+v"(Mo[2]), "+v"(Rr[t][0]),
              "+v"(Rr[t][1]), "+v"(Rr[t][2])
            : "v"(nr[tj]), "v"(nr[1 - tj]), "v"(nrr), "n"(Jl));
      }
    }
  }
}
// DPP64: the fused form wins where registers / issue slots bound the kernel (slack arrays in the workspace at
// 2 waves/SIMD, two input tiles); at one wave per SIMD with one tile the separate broadcast moves are faster
template <int J, int TU, bool DPP64>
__device__ __forceinline__ void gj_step(double M[][TU][3], double Rr[][3], int c, int g, double& minpiv) {
  if constexpr (DPP64) gj_step_dpp64<J, TU>(M, Rr, c, g, minpiv);
  else gj_step_mov<J, TU>(M, Rr, c, g, minpiv);
}
// the three pivots of contact point LEG (a wave-uniform branch skips swing legs)
template <int LEG, int TU, bool DPP64>
__device__ __forceinline__ void gj_leg(double M[][TU][3], double Rr[][3], int c, int g, double& minpiv) {
  gj_step<3 * LEG, TU, DPP64>(M, Rr, c, g, minpiv);
  gj_step<3 * LEG + 1, TU, DPP64>(M, Rr, c, g, minpiv);
  gj_step<3 * LEG + 2, TU, DPP64>(M, Rr, c, g, minpiv);
}
// X = diag(M)^-1 Rr after all pivots (rows of skipped swing-leg pivots keep their
// own positive diagonal R and a zero right-hand side)
template <int TU>
__device__ __forceinline__ void gj_finish(const double M[][TU][3], double Rr[][3], int g) {
#pragma unroll
  for (int t = 0; t < TU; ++t)
#pragma unroll
    for (int e = 0; e < 3; ++e) {
      const double dg = __shfl(M[t][t][e], (g << 4) | (4 * e + g));   // M[r][r] lives in lane (g, r)
      Rr[t][e] *= fast_rcp(dg);
    }
}

// Riccati backward pass with interior-point weights; writes KD (rotated gains
// [Kt | dt], NU x 13 per knot).  Returns nonzero when a pivot is not positive.
// PIPE: build the next knot's operands during the stage solve (needs 12 more VGPRs)
template <class MD, bool PROF, bool PIPE, bool DPP64, bool LEANOPS = false>
// dV1 (optional): the expected decrease sum_k d_k' Qu_k of the line-search test (reference mode).
__device__ inline int backward_pass(const DevParams& P, const Layout& L, double* sm, double* KD,
                                    const double* ROT, int lane, unsigned conmask, Prof<PROF>& prof,
                                    double* dV1 = nullptr) {
  typedef typename MD::D D;
  constexpr int TU = D::TU;
  const int N = P.N;
  const double* cst = sm + L.cst;
  const double* bw0 = sm + L.bw0;
  const int c = lane & 15, g = lane >> 4;
  const bool cval = c < 12;
  const int lc = cval ? c / 3 : 0, bc = cval ? c - 3 * lc : 0;
  // ---- per-lane, per-fragment patterns (row r_e = 4e + g): the model's operand / cost
  //      patterns, and the rotated-block / gain offsets shared by both models ----
  typename MD::template OperandsT<LEANOPS> ops;
  CostPattern cp;
  ops.init(P, cst, bw0, lane, cp);
  int doff[3], goff[3], koff[3];
#pragma unroll
  for (int e = 0; e < 3; ++e) {
    const int r = 4 * e + g;
    const int lr = r / 3, ar = r - 3 * lr;
    doff[e] = (cval && lc == lr) ? 21 * lr + 9 + 3 * ar + bc : -1;
    goff[e] = (c == 12) ? 21 * lr + 18 + ar : -1;
    koff[e] = (c < 13) ? 13 * r + c : -1;
  }
  // ---- terminal cost-to-go  P_aug = [lxx_N | lx_N] ----
  double Pf[3];
  {
    const double* XTk = sm + L.XT + kXT * N;
#pragma unroll
    for (int e = 0; e < 3; ++e) Pf[e] = cp.qadd[e] + ((cp.xoff[e] >= 0) ? XTk[cp.xoff[e]] : 0.0);
  }
  double minpiv = 1e300;
  double dv_part = 0.0;       // this lane's share of dV1 (gradient column only)
  // operands of knot kk: Abar and rotated Bbar * T, straight into fragments.  They do not
  // depend on the cost-to-go, so the NEXT knot's operands are built while the
  // latency-bound stage solve of the current knot runs (software pipelining).
  // clamped offsets of the per-knot terms: loaded unconditionally (no exec-mask juggling between the
  // MFMAs) and selected afterwards
  int xo[3], go[3], dof[3];
#pragma unroll
  for (int e = 0; e < 3; ++e) {
    xo[e] = (cp.xoff[e] >= 0) ? cp.xoff[e] : 0;
    go[e] = (goff[e] >= 0) ? goff[e] : 0;
    dof[e] = (doff[e] >= 0) ? doff[e] : 0;
  }
  auto build_operands = [&](int kk, double Afo[3], double Bfo[][3]) {
    const double* ABk = sm + L.AB + kAB * kk;
    const double* ROTkk = ROT + D::ROT * kk;
    double tc[TU][3];
#pragma unroll
    for (int t = 0; t < TU; ++t) {
      const double* Tl = ROTkk + 21 * (4 * t + lc);     // lc = bc = 0 on the lanes without a column
      const double v0 = Tl[bc], v1 = Tl[3 + bc], v2 = Tl[6 + bc];
      tc[t][0] = cval ? v0 : 0.0; tc[t][1] = cval ? v1 : 0.0; tc[t][2] = cval ? v2 : 0.0;
    }
    ops.build(P, ABk, tc, Afo, Bfo);
  };
  double Af[3], Bf[TU][3], Afn[3], Bfn[TU][3];
  if (PIPE) build_operands(N - 1, Af, Bf);
  for (int k = N - 1; k >= 0; --k) {
    if (!PIPE) build_operands(k, Af, Bf);
    const double* ROTk = ROT + D::ROT * k;
    const double* XTk = sm + L.XT + kXT * k;
    prof.tick(PH_BUILD);
    // per-knot terms (independent of the products below)
    double xt[3], gt[TU][3], dt_[TU][3];
#pragma unroll
    for (int e = 0; e < 3; ++e) {
      xt[e] = XTk[xo[e]];
#pragma unroll
      for (int t = 0; t < TU; ++t) { gt[t][e] = ROTk[84 * t + go[e]]; dt_[t][e] = ROTk[84 * t + dof[e]]; }
    }
    // ---- T = P'A (col 12 <- p), S = P'B ; Qxx = A'T, Qux = B'T, Quu = B'S ----
    // The next knot's operand build is independent work placed between the products: its LDS reads are in
    // flight while the products run (the vector arithmetic itself does not overlap an FP64 MFMA of the same
    // wave: tools/microbench/mfma_bench.hip).
    const d4 z4 = {0.0, 0.0, 0.0, 0.0};
    d4 aT = z4, aS[TU];
    double Tf[3], Sf[TU][3];
    if (PIPE) {
      // T and S chains interleaved, one fragment row of the NEXT knot's operands between the products
      // (scheduling fences keep the compiler from clustering the MFMAs again)
      const int kn = k > 0 ? k - 1 : 0;
      const double* ROTn = ROT + D::ROT * kn;
      double tc[TU][3];
#pragma unroll
      for (int t = 0; t < TU; ++t) {
        aS[t] = z4;
        const double* Tl = ROTn + 21 * (4 * t + lc);
        const double v0 = Tl[bc], v1 = Tl[3 + bc], v2 = Tl[6 + bc];
        tc[t][0] = cval ? v0 : 0.0; tc[t][1] = cval ? v1 : 0.0; tc[t][2] = cval ? v2 : 0.0;
      }
      // Ordering pins (empty volatile asm statements keep their mutual order): each MFMA's A operand and the
      // inputs / outputs of each half row pass through one, which leaves  MFMA, half row, MFMA, half row, ...
      int abofs = kAB * kn;
#pragma unroll
      for (int kk = 0; kk < 3; ++kk) {
        double pr[TU][3];
        double pa = Pf[kk];
        asm volatile("" : "+v"(pa));
        aT = __builtin_amdgcn_mfma_f64_16x16x4f64(pa, Af[kk], aT, 0, 0, 0);
        asm volatile("" : "+v"(abofs));
        ops.row_a(P, sm + L.AB + abofs, kk, Afn, pr);
        asm volatile("" : "+v"(Afn[kk]));
#pragma unroll
        for (int t = 0; t < TU; ++t) {
          asm volatile("" : "+v"(pr[t][0]), "+v"(pr[t][1]), "+v"(pr[t][2]));
          double pb = Pf[kk];
          asm volatile("" : "+v"(pb));
          aS[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(pb, Bf[t][kk], aS[t], 0, 0, 0);
        }
#pragma unroll
        for (int t = 0; t < TU; ++t) asm volatile("" : "+v"(pr[t][0]), "+v"(pr[t][1]), "+v"(pr[t][2]));
        ops.row_b(tc, kk, pr, Bfn);
#pragma unroll
        for (int t = 0; t < TU; ++t) asm volatile("" : "+v"(Bfn[t][kk]));
      }
    } else {
      aT = mtm3(Pf, Af, z4);
#pragma unroll
      for (int t = 0; t < TU; ++t) aS[t] = mtm3(Pf, Bf[t], z4);
    }
#pragma unroll
    for (int t = 0; t < TU; ++t)
#pragma unroll
      for (int e = 0; e < 3; ++e) Sf[t][e] = aS[t][e];
#pragma unroll
    for (int e = 0; e < 3; ++e) Tf[e] = (c == 12) ? Pf[e] : aT[e];
    const d4 aXX = mtm3(Af, Tf, z4);
    double Qxx[3], Qux[TU][3], Quu[TU][TU][3], Rr[TU][3];
#pragma unroll
    for (int e = 0; e < 3; ++e) Qxx[e] = aXX[e] + cp.qadd[e] + ((cp.xoff[e] >= 0) ? xt[e] : 0.0);
#pragma unroll
    for (int t = 0; t < TU; ++t) {
      const d4 aUX = mtm3(Bf[t], Tf, z4);
#pragma unroll
      for (int e = 0; e < 3; ++e) {
        Qux[t][e] = aUX[e] + ((goff[e] >= 0) ? gt[t][e] : 0.0);
        Rr[t][e] = Qux[t][e];
      }
#pragma unroll
      for (int t2 = 0; t2 < TU; ++t2) {
        const d4 aUU = mtm3(Bf[t], Sf[t2], z4);
#pragma unroll
        for (int e = 0; e < 3; ++e)
          Quu[t][t2][e] = aUU[e] + ((t == t2 && doff[e] >= 0) ? dt_[t][e] : 0.0);
      }
    }
    prof.tick(PH_MFMA);
    if (PROF) {   // diagnostics: wait for the last product before the stage solve is timed
      double sink;
      asm volatile("v_mov_b64 %0, %1" : "=v"(sink) : "v"(Quu[TU - 1][TU - 1][2]));
      asm volatile("v_mov_b64 %0, %1" : "=v"(sink) : "v"(Rr[TU - 1][2]));
      prof.tick(PH_DRAIN);
    }
    // ---- stage solve: [Kt | dt] = -Quu^-1 [Qux | Qu]; swing-leg pivots are decoupled ----
    if (conmask & 1u) gj_leg<0, TU, DPP64>(Quu, Rr, c, g, minpiv);
    if (conmask & 2u) gj_leg<1, TU, DPP64>(Quu, Rr, c, g, minpiv);
    if (conmask & 4u) gj_leg<2, TU, DPP64>(Quu, Rr, c, g, minpiv);
    if (conmask & 8u) gj_leg<3, TU, DPP64>(Quu, Rr, c, g, minpiv);
    if (TU > 1) {
      if (conmask & 16u) gj_leg<(TU > 1 ? 4 : 0), TU, DPP64>(Quu, Rr, c, g, minpiv);
      if (conmask & 32u) gj_leg<(TU > 1 ? 5 : 0), TU, DPP64>(Quu, Rr, c, g, minpiv);
      if (conmask & 64u) gj_leg<(TU > 1 ? 6 : 0), TU, DPP64>(Quu, Rr, c, g, minpiv);
      if (conmask & 128u) gj_leg<(TU > 1 ? 7 : 0), TU, DPP64>(Quu, Rr, c, g, minpiv);
    }
    gj_finish<TU>(Quu, Rr, g);
    double Kf[TU][3];
#pragma unroll
    for (int t = 0; t < TU; ++t)
#pragma unroll
      for (int e = 0; e < 3; ++e) Kf[t][e] = -Rr[t][e];
    if (dV1) {                // column 12 of Qux is the rotated Qu, of Kf the rotated feed-forward
#pragma unroll
      for (int t = 0; t < TU; ++t)
#pragma unroll
        for (int e = 0; e < 3; ++e) dv_part += (c == 12) ? Kf[t][e] * Qux[t][e] : 0.0;
    }
    prof.tick(PH_SOLVE);
    // ---- cost-to-go: P_aug <- Qxx_aug + Qux_aug' [Kt | dt] ----
    {
      d4 acc = {Qxx[0], Qxx[1], Qxx[2], 0.0};
#pragma unroll
      for (int t = 0; t < TU; ++t) acc = mtm3(Qux[t], Kf[t], acc);
#pragma unroll
      for (int e = 0; e < 3; ++e) Pf[e] = acc[e];
    }
    // ---- store the ROTATED gains [Kt | dt] straight from the fragments; the rollouts
    //      apply T_l (3x3 per leg) to the input increments ----
    {
      double* KDk = KD + D::KD * k;
#pragma unroll
      for (int t = 0; t < TU; ++t)
#pragma unroll
        for (int e = 0; e < 3; ++e)
          if (koff[e] >= 0) KDk[156 * t + koff[e]] = Kf[t][e];
    }
    if (PIPE) {
#pragma unroll
      for (int e = 0; e < 3; ++e) Af[e] = Afn[e];
#pragma unroll
      for (int t = 0; t < TU; ++t)
#pragma unroll
        for (int e = 0; e < 3; ++e) Bf[t][e] = Bfn[t][e];
    }
    prof.tick(PH_PUPD);
  }
  if (dV1) *dV1 = wave_sum(dv_part);
  return !(minpiv > 0.0);   // also true for a NaN pivot
}

// nonlinear closed-loop rollout with step alpha from (X,U) into the candidate
// Xc and the input increment dU = T (alpha dt + Kt (x' (-) x)).  The gains are
// stored in the rotated input coordinates of the backward pass; lane 4l+a (a<3)
// owns input 3l+a, so the 3x3 rotation T_l is applied inside a lane quad with
// DPP quad_perm broadcasts.  The loads of knot k+1 (old state, gain row, T_l, old
// input) do not depend on the rollout state and are issued one knot ahead.
struct RollLoads {
  double xo[13], kd[13], T[3], uo;
};
template <class D, bool WITH_X>
__device__ __forceinline__ void roll_load(const Layout& L, const double* sm, const double* KD,
                                          const double* ROT, int k, int uj, int ql, int qa, RollLoads& r) {
  if (WITH_X)
#pragma unroll
    for (int i = 0; i < 13; ++i) r.xo[i] = sm[L.X + 13 * k + i];
  const double* kd = KD + D::KD * k + 13 * uj;
#pragma unroll
  for (int i = 0; i < 13; ++i) r.kd[i] = kd[i];
  const double* T = ROT + D::ROT * k + 21 * ql + 3 * qa;
  r.T[0] = T[0]; r.T[1] = T[1]; r.T[2] = T[2];
  r.uo = sm[L.U + D::NU * k + uj];
}
// PF_X: also prefetch the old state (LDS variant: registers to spare); the global-gains
// variant is register-bound (2 waves/SIMD) and prefetches only its high-latency gain row / T_l
template <class MD, bool PF_X, bool PF_K, bool LEAN, bool PROF>
__device__ inline void rollout_closed(const DevParams& P, const Layout& L, double* sm, const double* KD,
                                      const double* ROT, double alpha, int lane, Prof<PROF>& prof) {
  typedef typename MD::D D;
  const int N = P.N;
  const double* cst = sm + L.cst;
  typename MD::template RegsT<LEAN> M;
  M.load(cst, sm + L.bw0);
  const bool ulane = (lane < 4 * D::NLEG) && ((lane & 3) < 3);
  const int ql = ulane ? (lane >> 2) : 0, qa = ulane ? (lane & 3) : 0;   // contact point, axis of this lane
  const int uj = 3 * ql + qa;
  double xc[13], xn[13];
#pragma unroll
  for (int i = 0; i < 13; ++i) xc[i] = cst[D::C_X0 + i];
  if (lane == 0)
#pragma unroll
    for (int i = 0; i < 13; ++i) sm[L.Xc + i] = xc[i];
  RollLoads cur, nxt;
  if (PF_K) roll_load<D, PF_X>(L, sm, KD, ROT, 0, uj, ql, qa, cur);
  for (int k = 0; k < N; ++k) {
    if (!PF_K) roll_load<D, PF_X>(L, sm, KD, ROT, k, uj, ql, qa, cur);
    if (!PF_X)
#pragma unroll
      for (int i = 0; i < 13; ++i) cur.xo[i] = sm[L.X + 13 * k + i];
    // dx = xc (-) X_k in error-state coordinates
    double dx[12];
    MD::state_diff(cur.xo, xc, dx);
    double unew;
    {
      // rotated increment of input uj, then u-space increment through T_l
      const double* kd = cur.kd;
      const double p0 = alpha * kd[12] + kd[0] * dx[0] + kd[1] * dx[1] + kd[2] * dx[2];
      const double p1 = kd[3] * dx[3] + kd[4] * dx[4] + kd[5] * dx[5];
      const double p2 = kd[6] * dx[6] + kd[7] * dx[7] + kd[8] * dx[8];
      const double p3 = kd[9] * dx[9] + kd[10] * dx[10] + kd[11] * dx[11];
      const double s = (p0 + p1) + (p2 + p3);
      const double s0 = dpp_mov<0x00>(s), s1 = dpp_mov<0x55>(s), s2 = dpp_mov<0xAA>(s);   // quad_perm broadcasts
      const double inc = cur.T[0] * s0 + cur.T[1] * s1 + cur.T[2] * s2;
      if (ulane) sm[L.dU + D::NU * k + uj] = inc;               // the increment, as computed
      unew = cur.uo + inc;
    }
    prof.tick(PH_R_GAIN);
    if (PF_K && k + 1 < N) roll_load<D, PF_X>(L, sm, KD, ROT, k + 1, uj, ql, qa, nxt);   // one knot ahead
    // broadcast the new inputs from their owner lanes (4l+a) with v_readlane: no LDS round trip
    double un[D::NU];
#pragma unroll
    for (int j = 0; j < D::NU; ++j) un[j] = read_lane(unew, 4 * (j / 3) + (j % 3));
    prof.tick(PH_R_BCAST);
    MD::template step<LEAN>(P, M, xc, un, xn);
#pragma unroll
    for (int i = 0; i < 13; ++i) xc[i] = xn[i];
    if (lane == 0)
#pragma unroll
      for (int i = 0; i < 13; ++i) sm[L.Xc + 13 * (k + 1) + i] = xn[i];
    if (PF_K) cur = nxt;
    prof.tick(PH_R_STEP);
  }
  QSYNC();
}

// Slack / multiplier directions from the TRIAL rollout (alpha = 1).  The cone
// rows are linear in u, so with the rollout's input increment dU
//   ds = -(a_i . dU_l + rc),  rc = c(u) + s  (tracked analytically, see below),
//   dlam = (target - (1 + kappa) s lam - lam ds) / s,
// followed by the fraction-to-the-boundary step lengths.
template <class D>
__device__ inline void ipm_directions(const DevParams& P, const Layout& L, double* sm, double* sl, double target,
                                      int lane, double* alpha_p, double* alpha_d, double* full_step) {
  const int N = P.N;
  const double* cst = sm + L.cst;
  const double* cr = cst + D::C_CR;
  double ap = 1.0, ad = 1.0, stp = 0.0;
  for (int idx = lane; idx < N * D::NC; idx += kWave) {
    const int k = idx / D::NC, row = idx - D::NC * k, l = row / 6, i = row - 6 * l;
    double dsv = 0.0, dlv = 0.0;
    if (cst[D::C_CON + l] != 0.0) {
      const double* du = sm + L.dU + D::NU * k + 3 * l;
      const double d0 = du[0], d1 = du[1], d2 = du[2];
      const double jd = cr[3 * i] * d0 + cr[3 * i + 1] * d1 + cr[3 * i + 2] * d2;
      stp = fmax(stp, fmax(fabs(d0), fmax(fabs(d1), fabs(d2))));
      const double sv = sl[L.S + idx], lv = sl[L.LAM + idx];
      const double kap = sl[L.DS + idx];                     // flag left by the previous ipm_apply
      dsv = -(jd + sl[L.RC + idx]);
      dlv = (target - (1.0 + kap) * sv * lv - lv * dsv) * fast_rcp(sv);
      if (dsv < 0.0) ap = fmin(ap, -P.tau * sv * fast_rcp(dsv));
      if (dlv < 0.0) ad = fmin(ad, -P.tau * lv * fast_rcp(dlv));
    }
    sl[L.DS + idx] = dsv;
    sl[L.DLAM + idx] = dlv;
  }
  *alpha_p = wave_min(ap);
  *alpha_d = wave_min(ad);
  // convergence is judged on the FULL Newton step (the trial increment): swing inputs do not move and every
  // stance input is seen by the six rows of its contact point
  *full_step = wave_max(stp);
}

// Apply the step to (s, rc, lam).  The slack residual rc = c(u) + s is carried
// as its own variable, never recomputed from c(u) ~ 100 N, so slacks keep their
// RELATIVE accuracy far below 1e-14 N (weakly active rows need it).  The cone
// rows are linear in u: a shortened primal step scales the trial increment
// (dU <- alpha_p dU), hence s + alpha_p ds stays inside the interior exactly and
// rc <- (1 - alpha_p) rc; a full step zeroes rc exactly.
template <class D>
__device__ inline void ipm_apply(const DevParams& P, const Layout& L, double* sl, double ap, double ad,
                                 unsigned conmask, int lane, unsigned& kapbits, double& sl_part, double& rc_part) {
  const int N = P.N;
  unsigned newbits = 0;
  int j = 0;
  sl_part = 0.0;      // this lane's share of sum s*lambda and max |rc| at the NEW point: the next iteration's
  rc_part = 0.0;      // barrier parameter / residual come from these, without re-reading the arrays
  for (int idx = lane; idx < N * D::NC; idx += kWave, ++j) {
    const int l = (idx % D::NC) / 6;
    if (!(conmask & (1u << l))) continue;
    const double s0 = sl[L.S + idx], l0 = sl[L.LAM + idx];
    const bool kap0 = (kapbits >> j) & 1u;   // this lane owns row idx in every pass
    const double s1 = s0 + ap * sl[L.DS + idx];
    const double l1 = l0 + ad * sl[L.DLAM + idx];
    const double rc1 = (ap >= 1.0) ? 0.0 : (1.0 - ap) * sl[L.RC + idx];
    sl[L.S + idx] = s1;
    sl[L.RC + idx] = rc1;
    sl[L.LAM + idx] = l1;
    sl_part += s1 * l1;
    rc_part = fmax(rc_part, fabs(rc1));
    // Tapia indicators: a weakly active row halves BOTH s and lambda on a full Newton
    // step (regular rows send one ratio to ~1, the other to ~sigma).  Such rows get the
    // second-order complementarity right-hand side  target - 2 s lam  next iteration,
    // which removes their linear (ratio 1/2) convergence.  The flag lives in the DS slot.
    // ratios s1/s0, l1/l0 against 0.4 / 0.6, written without the divisions (s0, l0 > 0)
    const bool sig = (ap >= 0.99) && (ad >= 0.99) && (s1 < 0.6 * s0) && (l1 < 0.6 * l0) &&
                     (kap0 || ((s1 > 0.4 * s0) && (l1 > 0.4 * l0)));
    sl[L.DS + idx] = sig ? 1.0 : 0.0;        // read by the next rotation pre-pass / directions
    newbits |= sig ? (1u << j) : 0u;
  }
  kapbits = newbits;
}

// shortened primal step: scale the trial increment and re-roll the states open loop
template <class MD, bool LEAN>
__device__ inline void rollout_scaled(const DevParams& P, const Layout& L, double* sm, double ap, int lane) {
  typedef typename MD::D D;
  const int N = P.N;
  const double* cst = sm + L.cst;
  for (int i = lane; i < N * D::NU; i += kWave) sm[L.dU + i] *= ap;
  QSYNC();
  typename MD::template RegsT<LEAN> M;
  M.load(cst, sm + L.bw0);
  double x[13], xn[13], u[D::NU];
#pragma unroll
  for (int i = 0; i < 13; ++i) x[i] = cst[D::C_X0 + i];
  for (int k = 0; k < N; ++k) {
#pragma unroll
    for (int j = 0; j < D::NU; ++j) u[j] = sm[L.U + D::NU * k + j] + sm[L.dU + D::NU * k + j];
    MD::template step<LEAN>(P, M, x, u, xn);
#pragma unroll
    for (int i = 0; i < 13; ++i) x[i] = xn[i];
    if (lane == 0)
#pragma unroll
      for (int i = 0; i < 13; ++i) sm[L.Xc + 13 * (k + 1) + i] = xn[i];
  }
  QSYNC();
}

template <class MD>
__device__ inline double cost_plain(const DevParams& P, const Layout& L, double* sm, int lane) {
  double J = 0.0;
  if (lane <= P.N)
    J = MD::knot_cost(P, sm + L.refp, sm + L.uref, lane, sm + L.X + 13 * lane,
                      (lane < P.N) ? sm + L.U + MD::D::NU * lane : nullptr);
  return wave_sum(J);
}

// ---- the solve kernel ---------------------------------------------------------
// VAR 0: everything in LDS; 1: gains / rotation blocks in the global workspace gws (one slice per instance);
// 2: the slack / multiplier arrays there as well
#define QMPC_SOLVE_WAVES(MD, VAR) \
  ((VAR) == 0 || (VAR) == 3 ? 1 : (MD::NL != 4 ? ((VAR) == 2 ? QMPC_NL8_WAVES : 1) : ((VAR) == 2 ? QMPC_V2_WAVES : 2)))
template <class MD, bool PROF, int VAR>
__global__ __launch_bounds__(64, QMPC_SOLVE_WAVES(MD, VAR)) void qmpc_solve_kernel(
    DevParams P, const qmpc_input* __restrict__ in_, double* __restrict__ forces, qmpc_info* __restrict__ info,
    double* __restrict__ traj_u, double* __restrict__ traj_x, int batch, long long* __restrict__ prof_out,
    double* __restrict__ gws) {
  extern __shared__ __attribute__((aligned(16))) double sm[];
  const int b = blockIdx.x;
  if (b >= batch) return;
  const int lane = threadIdx.x;
  constexpr int warm_t = 0;            // a plain solve always starts from u_ref (QuatMpc.cpp:253)
#include "qmpc_solve_body.inc"
}

// ---- linearisation only (qmpc_linearize): rollout of U = u_ref + dense Abar/Bbar
template <class MD>
__global__ __launch_bounds__(64) void qmpc_linearize_kernel(DevParams P, const qmpc_input* __restrict__ in,
                                                            double* __restrict__ Abar,
                                                            double* __restrict__ Bbar,
                                                            double* __restrict__ Xout, int batch) {
  extern __shared__ __attribute__((aligned(16))) double sm[];
  const int b = blockIdx.x;
  if (b >= batch) return;
  const int lane = threadIdx.x;
  const int N = P.N;
  const Layout L = make_layout(N, true, MD::NL);
  int status = QMPC_OK;
  setup_instance<MD>(P, L, sm, in + b, lane, &status);
  if (status != QMPC_OK) {
    for (int i = lane; i < N * 144; i += kWave) { Abar[(size_t)b * N * 144 + i] = 0.0; Bbar[(size_t)b * N * 144 + i] = 0.0; }
    for (int i = lane; i < (N + 1) * MD::NX; i += kWave) Xout[(size_t)b * (N + 1) * MD::NX + i] = 0.0;
    return;
  }
  for (int i = lane; i < N * 12; i += kWave) sm[L.U + i] = sm[L.uref + (i % 12)];
  QSYNC();
  rollout_open<MD, true>(P, L, sm, lane);
  expansions<MD>(P, L, sm, lane);
  for (int i = lane; i < N * 144; i += kWave) {
    const int k = i / 144, e = i - 144 * k, r = e / 12, c = e - 12 * r;
    const double* AB = sm + L.AB + kAB * k;
    Abar[(size_t)b * N * 144 + i] = MD::a_elem(P, sm + L.cst, sm + L.bw0, AB, r, c);
    Bbar[(size_t)b * N * 144 + i] = MD::b_elem(P, sm + L.cst, sm + L.bw0, AB, r, c);
  }
  for (int i = lane; i < (N + 1) * MD::NX; i += kWave) {
    const int k = i / MD::NX, j = i - MD::NX * k;
    Xout[(size_t)b * (N + 1) * MD::NX + i] = sm[L.X + 13 * k + j];
  }
}

// ---- leg kinematics + force -> joint torque map ------------
// A1Kinematics::fk / ::jac (A1Kinematics.cpp:9-19, closed forms :38-128) and
// BaseInterface::tau_ctrl_update (BaseInterface.cpp:343-408: tau = -J' f, zero for swing
// legs while walking).  One thread per (instance, leg); consecutive threads touch
// consecutive 24-byte triples, so every load/store of a wave is one contiguous span:
// this pass is pure HBM streaming (40 doubles per instance).
struct LegGeom {
  double rho_fix[4][5];
  double rho_opt[4][3];
};
#ifndef QMPC_FUSED_TU    // the kernels below are not templates: one definition, in the first translation unit
struct LegTerms { double s0, c0, L, X, L2, X2, D; };
__device__ __forceinline__ LegTerms leg_terms(const double* q, const double* c, const double* r) {
  LegTerms t;
  double s1, c1, s12, c12;
  sincos(q[0], &t.s0, &t.c0);
  sincos(q[1], &s1, &c1);
  sincos(q[1] + q[2], &s12, &c12);
  const double lce = r[4] - c[2];
  t.L2 = lce * c12 + c[0] * s12;       // calf part of the leg extension
  t.X2 = -lce * s12 + c[0] * c12;      // calf part of the fore-aft offset
  t.L = r[3] * c1 + t.L2;
  t.X = -r[3] * s1 + t.X2;
  t.D = r[2] + c[1];
  return t;
}
// mode 0: foot_pos_body / jac outputs (either may be null); mode 1: tau = -J' f
__global__ __launch_bounds__(256) void qmpc_leg_kernel(LegGeom G, const double* __restrict__ joint_pos,
                                                       const double* __restrict__ forces,
                                                       const double* __restrict__ contacts, int walking,
                                                       double* __restrict__ out_p, double* __restrict__ out_J,
                                                       double* __restrict__ out_tau, int batch) {
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (size_t)batch * 4) return;
  const int l = (int)(t & 3);
  const double q[3] = {joint_pos[3 * t], joint_pos[3 * t + 1], joint_pos[3 * t + 2]};
  if (out_tau) {
    const bool stance = !contacts || contacts[t] != 0.0;
    if (walking && !stance) {
      out_tau[3 * t] = 0.0; out_tau[3 * t + 1] = 0.0; out_tau[3 * t + 2] = 0.0;
      return;
    }
  }
  const LegTerms k = leg_terms(q, G.rho_opt[l], G.rho_fix[l]);
  // J column-major: J[3j+i] = d p_i / d q_j
  const double J[9] = {0.0, -k.D * k.s0 + k.L * k.c0, k.D * k.c0 + k.L * k.s0,
                       -k.L, k.X * k.s0, -k.X * k.c0,
                       -k.L2, k.X2 * k.s0, -k.X2 * k.c0};
  if (out_p) {
    out_p[3 * t] = G.rho_fix[l][0] + k.X;
    out_p[3 * t + 1] = G.rho_fix[l][1] + k.D * k.c0 + k.L * k.s0;
    out_p[3 * t + 2] = k.D * k.s0 - k.L * k.c0;
  }
  if (out_J)
#pragma unroll
    for (int i = 0; i < 9; ++i) out_J[9 * t + i] = J[i];
  if (out_tau) {
    const double f[3] = {forces[3 * t], forces[3 * t + 1], forces[3 * t + 2]};
#pragma unroll
    for (int j = 0; j < 3; ++j) out_tau[3 * t + j] = -(J[3 * j] * f[0] + J[3 * j + 1] * f[1] + J[3 * j + 2] * f[2]);
  }
}

// Torque map as a streaming pass: a block owns 64 consecutive instances (768 doubles of joint angles, of forces
// and of torques).  The arrays move between HBM and LDS as contiguous 16-byte-per-lane accesses; the per-(instance,
// leg) triples are picked out of LDS, where the 24-byte stride costs nothing.  The pointers must be 16-byte
// aligned (the launcher falls back to qmpc_leg_kernel otherwise).
__global__ __launch_bounds__(256) void qmpc_tau_kernel(LegGeom G, const double* __restrict__ joint_pos,
                                                       const double* __restrict__ forces,
                                                       const double* __restrict__ contacts, int walking,
                                                       double* __restrict__ out_tau, int batch) {
  __shared__ __attribute__((aligned(16))) double sq[768], sf[768];
  const int tid = threadIdx.x;
  const size_t inst0 = (size_t)blockIdx.x * 64;
  const int n = (int)(((size_t)batch - inst0 < 64) ? ((size_t)batch - inst0) : 64);   // instances of this block
  const int pairs = 6 * n;                                                            // double2 per array
  const double2* q2 = reinterpret_cast<const double2*>(joint_pos + 12 * inst0);
  const double2* f2 = reinterpret_cast<const double2*>(forces + 12 * inst0);
  for (int i = tid; i < pairs; i += 256) {
    reinterpret_cast<double2*>(sq)[i] = q2[i];
    reinterpret_cast<double2*>(sf)[i] = f2[i];
  }
  const bool live = tid < 4 * n;
  const bool stance = live && (!contacts || contacts[4 * inst0 + tid] != 0.0);
  __syncthreads();
  double tau[3] = {0.0, 0.0, 0.0};
  if (live && !(walking && !stance)) {
    const int l = tid & 3;
    const double q[3] = {sq[3 * tid], sq[3 * tid + 1], sq[3 * tid + 2]};
    const double f[3] = {sf[3 * tid], sf[3 * tid + 1], sf[3 * tid + 2]};
    const LegTerms k = leg_terms(q, G.rho_opt[l], G.rho_fix[l]);
    const double J[9] = {0.0, -k.D * k.s0 + k.L * k.c0, k.D * k.c0 + k.L * k.s0,
                         -k.L, k.X * k.s0, -k.X * k.c0,
                         -k.L2, k.X2 * k.s0, -k.X2 * k.c0};
#pragma unroll
    for (int j = 0; j < 3; ++j) tau[j] = -(J[3 * j] * f[0] + J[3 * j + 1] * f[1] + J[3 * j + 2] * f[2]);
  }
  __syncthreads();            // every triple has been read: sq is reused for the torques
  if (live) { sq[3 * tid] = tau[0]; sq[3 * tid + 1] = tau[1]; sq[3 * tid + 2] = tau[2]; }
  __syncthreads();
  double2* t2 = reinterpret_cast<double2*>(out_tau + 12 * inst0);
  for (int i = tid; i < pairs; i += 256) t2[i] = reinterpret_cast<const double2*>(sq)[i];
}

// ---- MFMA layout self-test: C = X' * Y on [12][16] tiles -------------------------
__global__ __launch_bounds__(64) void qmpc_selftest_kernel(const double* __restrict__ X,
                                                           const double* __restrict__ Y,
                                                           double* __restrict__ C) {
  __shared__ __attribute__((aligned(16))) double t[3 * MAT];
  const int lane = threadIdx.x;
  for (int i = lane; i < MAT; i += kWave) { t[i] = X[i]; t[MAT + i] = Y[i]; }
  QSYNC();
  mtm(t + 2 * MAT, t, t + MAT, lane);
  QSYNC();
  for (int i = lane; i < MAT; i += kWave) C[i] = t[2 * MAT + i];
}

// ---- cross-lane primitive self-test: row-group broadcasts, wave reductions, quad broadcasts
__global__ __launch_bounds__(64) void qmpc_selftest_lanes_kernel(const double* __restrict__ in,
                                                                 double* __restrict__ out) {
  const int lane = threadIdx.x;
  const double x = in[lane];
  out[lane] = rowgroup_bcast<0>(x);
  out[64 + lane] = rowgroup_bcast<1>(x);
  out[128 + lane] = rowgroup_bcast<2>(x);
  out[192 + lane] = rowgroup_bcast<3>(x);
  out[256 + lane] = wave_sum(x);
  out[320 + lane] = wave_max(x);
  out[384 + lane] = wave_min(x);
  out[448 + lane] = row_bcast<5>(x);
  out[512 + lane] = dpp_mov<0x55>(x);   // quad_perm [1,1,1,1]
}

#endif  // QMPC_FUSED_TU

}  // namespace qmpc
