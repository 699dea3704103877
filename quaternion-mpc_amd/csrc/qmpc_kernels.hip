// qmpc_kernels.hip -- gfx950 kernels of the batched quaternion-MPC solve.
//
// One wavefront (64 lanes, one workgroup) owns one MPC instance.  The solve is
// the converged mode of include/qmpc.h: a primal-dual interior-point iteration
// whose Newton systems are solved with the iLQR/Riccati recursion over the
// horizon (12-dim error state, 12 inputs), i.e. the same backward/forward pass
// structure as the reference's external AL-iLQR solver (QuatMpc.cpp:218-256),
// with the cone rows (QuatMpc.cpp:194-215) handled by barrier weights instead of
// penalties.  All 12x12 products run on the FP64 matrix core
// (v_mfma_f64_16x16x4_f64) from [12][16] LDS tiles.
#include "qmpc_device.h"

namespace qmpc {

// Optional phase-level cycle accounting (s_memtime), compiled in only for the
// diagnostic instantiation qmpc_solve_kernel<true>.
enum { PH_SETUP = 0, PH_EXPAND, PH_BUILD, PH_MFMA, PH_SOLVE, PH_PUPD, PH_DIRS, PH_ROLL, PH_MISC, PH_COUNT };
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
__device__ inline void setup_instance(const DevParams& P, const Layout& L, double* sm,
                                      const qmpc_input* in, int lane, int* status) {
  double* cst = sm + L.cst;
  const double* rec = reinterpret_cast<const double*>(in);
  // one coalesced 8-byte-per-lane read of the 48-double record
  const double v = (lane < 48) ? rec[lane] : 0.0;
  const unsigned long long bad = __ballot(lane < 48 && !isfinite(v));
  double* raw = sm + L.Pm;  // scratch
  if (lane < 48) raw[lane] = v;
  QSYNC();
  // raw: quat 0..3, rot 4..12, linvel 13..15, angvel 16..18, foot 19..30,
  //      contacts 31..34, posref 35..37, velref 38..40, accref 41..43, quat_d 44..47
  int nc = 0;
  for (int l = 0; l < 4; ++l) nc += (raw[31 + l] != 0.0) ? 1 : 0;
  *status = bad ? QMPC_NAN_INPUT : (nc == 0 ? QMPC_NO_CONTACT : QMPC_OK);
  if (*status != QMPC_OK) return;
  if (lane < 12) cst[C_FOOT + lane] = raw[19 + lane];
  if (lane < 4) cst[C_CON + lane] = (raw[31 + lane] != 0.0) ? 1.0 : 0.0;
  if (lane < 3) {
    // g_body = R' (0,0,-9.81)  (AltroUtils.cpp:368-371)
    cst[C_GB + lane] = raw[4 + 6 + lane] * (-9.81);
  }
  if (lane < 13) {
    // x_init (QuatMpc.cpp:231-246; angular velocity dropped by the ';' at :242)
    double x0 = 0.0;
    if (lane >= 3 && lane < 7) x0 = raw[lane - 3];
    else if (lane >= 7 && lane < 10) x0 = raw[13 + lane - 7];
    else if (lane >= 10) x0 = P.drop_ang_vel ? 0.0 : raw[16 + lane - 10];
    cst[C_X0 + lane] = x0;
    // reference parameters: pos vel acc quat_d
    sm[L.refp + lane] = (lane < 9) ? raw[35 + lane] : raw[44 + lane - 9];
  }
  if (lane < 12) {
    // u_ref (QuatMpc.cpp:118-125)
    const int l = lane / 3, a = lane - 3 * l;
    sm[L.uref + lane] = (a == 2) ? raw[31 + l] * P.mass * 9.81 / (double)nc : 0.0;
  }
  if (lane < 18) {
    // C_mat * R (QuatMpc.cpp:47-52,203): rows (1,0,-mu),(-1,0,-mu),(0,1,-mu),(0,-1,-mu),(0,0,1),(0,0,-1)
    const int r = lane / 3, c = lane - 3 * r;
    const double C0 = (r == 0) ? 1.0 : (r == 1 ? -1.0 : 0.0);
    const double C1 = (r == 2) ? 1.0 : (r == 3 ? -1.0 : 0.0);
    const double C2 = (r < 4) ? -P.mu : (r == 4 ? 1.0 : -1.0);
    cst[C_CR + lane] = C0 * raw[4 + c] + C1 * raw[4 + 3 + c] + C2 * raw[4 + 6 + c];
  }
  QSYNC();
  if (lane < 3) {
    // wd0 = Iinv * (c x 5.204 g_body)  (AltroUtils.cpp:373-374,391)
    const double com[3] = {0.0223, 0.002, -0.0005};
    const double fg[3] = {5.204 * cst[C_GB], 5.204 * cst[C_GB + 1], 5.204 * cst[C_GB + 2]};
    const double mg[3] = {com[1] * fg[2] - com[2] * fg[1], com[2] * fg[0] - com[0] * fg[2],
                          com[0] * fg[1] - com[1] * fg[0]};
    cst[C_WD0 + lane] = P.Iinv[3 * lane] * mg[0] + P.Iinv[3 * lane + 1] * mg[1] + P.Iinv[3 * lane + 2] * mg[2];
  }
  if (lane < 36) {
    // Bw0 = Iinv * skew(r_l) * contact_l  (AltroUtils.cpp:431-434), 3x12 row-major
    const int a = lane / 12, col = lane - 12 * a, l = col / 3, b = col - 3 * l;
    const double* r = cst + C_FOOT + 3 * l;
    // skew(r) column b: b=0:(0, r2, -r1)  b=1:(-r2, 0, r0)  b=2:(r1, -r0, 0)
    double s0, s1, s2;
    if (b == 0) { s0 = 0.0; s1 = r[2]; s2 = -r[1]; }
    else if (b == 1) { s0 = -r[2]; s1 = 0.0; s2 = r[0]; }
    else { s0 = r[1]; s1 = -r[0]; s2 = 0.0; }
    sm[L.bw0 + lane] = cst[C_CON + l] * (P.Iinv[3 * a] * s0 + P.Iinv[3 * a + 1] * s1 + P.Iinv[3 * a + 2] * s2);
  }
  QSYNC();
}

// open-loop rollout of U from x0 into X (every lane computes it redundantly;
// lane 0 publishes).  ALTRO's initial rollout.
__device__ inline void rollout_open(const DevParams& P, const Layout& L, double* sm, int lane) {
  const double* cst = sm + L.cst;
  const double* bw0 = sm + L.bw0;
  double x[13], xn[13], u[12];
#pragma unroll
  for (int i = 0; i < 13; ++i) x[i] = cst[C_X0 + i];
  if (lane == 0)
#pragma unroll
    for (int i = 0; i < 13; ++i) sm[L.X + i] = x[i];
  for (int k = 0; k < P.N; ++k) {
#pragma unroll
    for (int j = 0; j < 12; ++j) u[j] = sm[L.U + 12 * k + j];
    srbd_step(P, cst, bw0, x, u, xn);
#pragma unroll
    for (int i = 0; i < 13; ++i) x[i] = xn[i];
    if (lane == 0)
#pragma unroll
      for (int i = 0; i < 13; ++i) sm[L.X + 13 * (k + 1) + i] = xn[i];
  }
  QSYNC();
}

// expansions at (X,U): one lane per knot + cone values by 64 lanes
__device__ inline void expansions(const DevParams& P, const Layout& L, double* sm, int lane) {
  const int N = P.N;
  if (lane <= N) {
    double x[13], xn[13], u[12];
#pragma unroll
    for (int i = 0; i < 13; ++i) {
      x[i] = sm[L.X + 13 * lane + i];
      xn[i] = (lane < N) ? sm[L.X + 13 * (lane + 1) + i] : 0.0;
    }
#pragma unroll
    for (int j = 0; j < 12; ++j) u[j] = (lane < N) ? sm[L.U + 12 * lane + j] : 0.0;
    double AB[27], lx[12], lxx[9];
    expand_knot(P, sm + L.cst, sm + L.bw0, sm + L.refp, lane, x, u, xn, AB, lx, lxx);
    if (lane < N)
#pragma unroll
      for (int i = 0; i < 27; ++i) sm[L.AB + 27 * lane + i] = AB[i];
#pragma unroll
    for (int i = 0; i < 12; ++i) sm[L.LX + 12 * lane + i] = lx[i];
#pragma unroll
    for (int i = 0; i < 9; ++i) sm[L.LXX + 9 * lane + i] = lxx[i];
  }
  // cone values c = C R u_l + b  (QuatMpc.cpp:194-205), N*24 rows
  const double* cr = sm + L.cst + C_CR;
  for (int idx = lane; idx < N * 24; idx += kWave) {
    const int k = idx / 24, row = idx - 24 * k, l = row / 6, i = row - 6 * l;
    const double* u = sm + L.U + 12 * k + 3 * l;
    double c = cr[3 * i] * u[0] + cr[3 * i + 1] * u[1] + cr[3 * i + 2] * u[2];
    if (i == 4) c += -P.fz_max * sm[L.cst + C_CON + l];
    sm[L.CV + idx] = c;
  }
  QSYNC();
}

// Per-leg rotation of the input coordinates for knot k (see DESIGN.md "rotated
// stage solve"): T_l = [q1 q2 q3] with q1 along the heaviest cone row, q2 the
// Gram-Schmidt complement of the second heaviest non-parallel row.  Lanes 0..3
// (one per leg) write rot[9*l + 3*a + b] = T_l[a][b], at[18*l + 3*i + b] =
// (T_l' a_i)[b], and lanes 0..23 write wts (Hessian weight) / gw (gradient weight).
__device__ inline void leg_rotations(const DevParams& P, const Layout& L, double* sm, int k,
                                     double target, int lane) {
  const double* cst = sm + L.cst;
  if (lane < 24) {
    const int l = lane / 6;
    const double on = cst[C_CON + l];
    const double s = sm[L.S + 24 * k + lane], lam = sm[L.LAM + 24 * k + lane];
    const double c = sm[L.CV + 24 * k + lane];
    const double w = (on != 0.0) ? lam / s : 0.0;
    sm[L.wts + lane] = w;
    sm[L.gw + lane] = (on != 0.0) ? (target / s + w * (c + s)) : 0.0;
  }
  QSYNC();
  if (lane < 4) {
    const int l = lane;
    double T[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    const double* cr = cst + C_CR;
    if (cst[C_CON + l] != 0.0) {
      const double* w = sm + L.wts + 6 * l;
      int i1 = 0;
      for (int i = 1; i < 6; ++i) if (w[i] > w[i1]) i1 = i;
      int i2 = -1;
      for (int i = 0; i < 6; ++i) {
        if (i == i1) continue;
        if ((i1 >= 4) && (i >= 4)) continue;  // rows 4,5 are antiparallel
        if (i2 < 0 || w[i] > w[i2]) i2 = i;
      }
      double q1[3], q2[3], q3[3];
      double n1 = sqrt(cr[3 * i1] * cr[3 * i1] + cr[3 * i1 + 1] * cr[3 * i1 + 1] + cr[3 * i1 + 2] * cr[3 * i1 + 2]);
      for (int a = 0; a < 3; ++a) q1[a] = cr[3 * i1 + a] / n1;
      double v[3] = {cr[3 * i2], cr[3 * i2 + 1], cr[3 * i2 + 2]};
      for (int pass = 0; pass < 2; ++pass) {
        const double dp = v[0] * q1[0] + v[1] * q1[1] + v[2] * q1[2];
        for (int a = 0; a < 3; ++a) v[a] -= dp * q1[a];
      }
      const double n2 = sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
      for (int a = 0; a < 3; ++a) q2[a] = v[a] / n2;
      q3[0] = q1[1] * q2[2] - q1[2] * q2[1];
      q3[1] = q1[2] * q2[0] - q1[0] * q2[2];
      q3[2] = q1[0] * q2[1] - q1[1] * q2[0];
      for (int a = 0; a < 3; ++a) { T[3 * a] = q1[a]; T[3 * a + 1] = q2[a]; T[3 * a + 2] = q3[a]; }
    }
    for (int i = 0; i < 9; ++i) sm[L.rot + 9 * l + i] = T[i];
    for (int i = 0; i < 6; ++i)
      for (int b = 0; b < 3; ++b)
        sm[L.at + 18 * l + 3 * i + b] = T[b] * cr[3 * i] + T[3 + b] * cr[3 * i + 1] + T[6 + b] * cr[3 * i + 2];
  }
  QSYNC();
}

// Solve  Sm * X = -Tm  for the 13 columns of Tm by Gaussian elimination without
// pivoting (Sm is SPD and, in the rotated coordinates, scaled-diagonally
// dominant); X -> Bm.  All three are [12][16] tiles.
__device__ inline int stage_solve(double* Sm, double* Tm, double* Bm, int lane) {
  int bad = 0;
#pragma unroll
  for (int j = 0; j < 11; ++j) {
    const int nr = 11 - j;         // rows below the pivot
    const int ncols = nr + 13;     // trailing matrix columns + 13 right-hand sides
    const double piv = Sm[j * LD + j];
    bad |= !(piv > 0.0);
    const double inv = 1.0 / piv;
    for (int idx = lane; idx < nr * ncols; idx += kWave) {
      const int i = j + 1 + idx / ncols;
      const int cc = idx - (i - j - 1) * ncols;
      const double f = Sm[i * LD + j] * inv;
      if (cc < nr) {
        const int c = j + 1 + cc;
        Sm[i * LD + c] -= f * Sm[j * LD + c];
      } else {
        const int c = cc - nr;
        Tm[i * LD + c] -= f * Tm[j * LD + c];
      }
    }
    QSYNC();
  }
  // back substitution, one lane per right-hand side
  if (lane < 13) {
    double xs[12];
#pragma unroll
    for (int i = 11; i >= 0; --i) {
      double s = Tm[i * LD + lane];
#pragma unroll
      for (int t = i + 1; t < 12; ++t) s -= Sm[i * LD + t] * xs[t];
      xs[i] = s / Sm[i * LD + i];
    }
#pragma unroll
    for (int i = 0; i < 12; ++i) Bm[i * LD + lane] = -xs[i];
  } else if (lane < 16) {
#pragma unroll
    for (int i = 0; i < 12; ++i) Bm[i * LD + lane] = 0.0;
  }
  bad |= !(Sm[11 * LD + 11] > 0.0);
  QSYNC();
  return bad;
}

// Riccati backward pass with interior-point weights; writes KD (unrotated gains
// [K | d], 12 x 13 per knot).  Returns nonzero when a pivot is not positive.
template <bool PROF>
__device__ inline int backward_pass(const DevParams& P, const Layout& L, double* sm, double target,
                                    int lane, Prof<PROF>& prof) {
  const int N = P.N;
  double* Pm = sm + L.Pm; double* Am = sm + L.Am; double* Bm = sm + L.Bm;
  double* Tm = sm + L.Tm; double* Sm = sm + L.Sm;
  const double* cst = sm + L.cst;
  const double* bw0 = sm + L.bw0;
  int notpd = 0;
  // terminal cost-to-go: P = lxx_N, p = lx_N (column 12)
  for (int idx = lane; idx < MAT; idx += kWave) {
    const int r = idx / LD, c = idx - LD * r;
    double v = 0.0;
    if (c < 12) {
      if (r >= 3 && r < 6 && c >= 3 && c < 6) v = sm[L.LXX + 9 * N + 3 * (r - 3) + (c - 3)];
      else if (r == c) v = P.Q[(r < 3) ? r : r + 1];
    } else if (c == 12) {
      v = sm[L.LX + 12 * N + r];
    }
    Pm[idx] = v;
  }
  QSYNC();
  for (int k = N - 1; k >= 0; --k) {
    const double* AB = sm + L.AB + 27 * k;
    leg_rotations(P, L, sm, k, target, lane);
    const double* rot = sm + L.rot;
    // dense Abar and rotated Bbar*T tiles
    for (int idx = lane; idx < MAT; idx += kWave) {
      const int r = idx / LD, c = idx - LD * r;
      double a = 0.0, b = 0.0;
      if (c < 12) {
        a = abar_elem(P, AB, r, c);
        const int l = c / 3, bb = c - 3 * l;
        const double* T = rot + 9 * l;
        b = bbar_elem(P, cst, bw0, AB, r, 3 * l) * T[bb] +
            bbar_elem(P, cst, bw0, AB, r, 3 * l + 1) * T[3 + bb] +
            bbar_elem(P, cst, bw0, AB, r, 3 * l + 2) * T[6 + bb];
      }
      Am[idx] = a;
      Bm[idx] = b;
    }
    QSYNC();
    prof.tick(PH_BUILD);
    // T = P'A (col 12 <- p), S = P'B
    {
      d4 accT = {0, 0, 0, 0}, accS = {0, 0, 0, 0};
      mtm_load(Pm, Am, lane, accT);
      mtm_load(Pm, Bm, lane, accS);
      const bool c12 = (lane & 15) == 12;
      Tm[lane] = c12 ? Pm[lane] : accT[0];
      Tm[64 + lane] = c12 ? Pm[64 + lane] : accT[1];
      Tm[128 + lane] = c12 ? Pm[128 + lane] : accT[2];
      Sm[lane] = accS[0]; Sm[64 + lane] = accS[1]; Sm[128 + lane] = accS[2];
    }
    QSYNC();
    // Qxx_aug = A'T_aug -> Am ; Qux_aug = B'T_aug -> Tm ; Quu = B'S -> Sm
    {
      d4 a1 = {0, 0, 0, 0}, a2 = {0, 0, 0, 0}, a3 = {0, 0, 0, 0};
      mtm_load(Am, Tm, lane, a1);
      mtm_load(Bm, Tm, lane, a2);
      mtm_load(Bm, Sm, lane, a3);
      QSYNC();
      Am[lane] = a1[0]; Am[64 + lane] = a1[1]; Am[128 + lane] = a1[2];
      Tm[lane] = a2[0]; Tm[64 + lane] = a2[1]; Tm[128 + lane] = a2[2];
      Sm[lane] = a3[0]; Sm[64 + lane] = a3[1]; Sm[128 + lane] = a3[2];
    }
    QSYNC();
    // stage cost + barrier terms
    for (int idx = lane; idx < MAT; idx += kWave) {
      const int r = idx / LD, c = idx - LD * r;
      if (c < 12) {
        // Qxx += lxx
        double v = 0.0;
        if (r >= 3 && r < 6 && c >= 3 && c < 6) v = sm[L.LXX + 9 * k + 3 * (r - 3) + (c - 3)];
        else if (r == c) v = P.Q[(r < 3) ? r : r + 1];
        Am[idx] += v;
        // Quu leg block += T' R T + sum_i w_i at_i at_i'
        const int l = r / 3, a = r - 3 * l;
        if (c / 3 == l) {
          const int b = c - 3 * l;
          const double* T = rot + 9 * l;
          double q = T[a] * P.R[3 * l] * T[b] + T[3 + a] * P.R[3 * l + 1] * T[3 + b] +
                     T[6 + a] * P.R[3 * l + 2] * T[6 + b];
          const double* at = sm + L.at + 18 * l;
          const double* w = sm + L.wts + 6 * l;
#pragma unroll
          for (int i = 0; i < 6; ++i) q += w[i] * at[3 * i + a] * at[3 * i + b];
          Sm[idx] += q;
        }
      } else if (c == 12) {
        // Qx += lx
        Am[idx] += sm[L.LX + 12 * k + r];
        // Qu (rotated) += T'(R (u - uref)) + sum_i g_i at_i
        const int l = r / 3, a = r - 3 * l;
        const double* T = rot + 9 * l;
        const double* u = sm + L.U + 12 * k + 3 * l;
        const double* ur = sm + L.uref + 3 * l;
        double q = T[a] * P.R[3 * l] * (u[0] - ur[0]) + T[3 + a] * P.R[3 * l + 1] * (u[1] - ur[1]) +
                   T[6 + a] * P.R[3 * l + 2] * (u[2] - ur[2]);
        const double* at = sm + L.at + 18 * l;
        const double* g = sm + L.gw + 6 * l;
#pragma unroll
        for (int i = 0; i < 6; ++i) q += g[i] * at[3 * i + a];
        Tm[idx] += q;
      }
    }
    QSYNC();
    // keep Qux_aug for the cost-to-go update (the solve destroys Tm): copy to Pm
    Pm[lane] = Tm[lane]; Pm[64 + lane] = Tm[64 + lane]; Pm[128 + lane] = Tm[128 + lane];
    QSYNC();
    prof.tick(PH_MFMA);
    notpd |= stage_solve(Sm, Tm, Bm, lane);   // Bm <- [Kt | dt] (rotated)
    prof.tick(PH_SOLVE);
    // unrotate and store the gains: KD[3l+a][c] = sum_b T_l[a][b] * Bm[3l+b][c]
    for (int idx = lane; idx < 12 * 13; idx += kWave) {
      const int r = idx / 13, c = idx - 13 * r, l = r / 3, a = r - 3 * l;
      const double* T = rot + 9 * l;
      sm[L.KD + 156 * k + idx] = T[3 * a] * Bm[(3 * l) * LD + c] + T[3 * a + 1] * Bm[(3 * l + 1) * LD + c] +
                                  T[3 * a + 2] * Bm[(3 * l + 2) * LD + c];
    }
    // P_aug <- Qxx_aug + Qux_aug' [Kt | dt]
    {
      d4 acc = {Am[lane], Am[64 + lane], Am[128 + lane], 0.0};
      mtm_load(Pm, Bm, lane, acc);
      QSYNC();
      Pm[lane] = acc[0]; Pm[64 + lane] = acc[1]; Pm[128 + lane] = acc[2];
    }
    QSYNC();
    prof.tick(PH_PUPD);
  }
  return notpd;
}

// linear forward sweep: du, ds, dlam and the fraction-to-the-boundary lengths
__device__ inline void ipm_directions(const DevParams& P, const Layout& L, double* sm, double target,
                                      int lane, double* alpha_p, double* alpha_d) {
  const int N = P.N;
  const double* cst = sm + L.cst;
  const double* bw0 = sm + L.bw0;
  double* dx = sm + L.dx;
  double* du = sm + L.du;
  if (lane < 12) dx[lane] = 0.0;
  QSYNC();
  double ap = 1.0, ad = 1.0;
  int cur = 0;
  for (int k = 0; k < N; ++k) {
    const double* dxc = dx + 12 * cur;
    double* dxn = dx + 12 * (cur ^ 1);
    if (lane < 12) {
      const double* kd = sm + L.KD + 156 * k + 13 * lane;
      double s = kd[12];
#pragma unroll
      for (int b = 0; b < 12; ++b) s += kd[b] * dxc[b];
      du[lane] = s;
    }
    QSYNC();
    if (lane < 24) {
      const int l = lane / 6, i = lane - 6 * l;
      if (cst[C_CON + l] != 0.0) {
        const double* cr = cst + C_CR + 3 * i;
        const double jd = cr[0] * du[3 * l] + cr[1] * du[3 * l + 1] + cr[2] * du[3 * l + 2];
        const double sv = sm[L.S + 24 * k + lane], lv = sm[L.LAM + 24 * k + lane];
        const double dsv = -(jd + sm[L.CV + 24 * k + lane] + sv);
        const double dlv = (target - sv * lv - lv * dsv) / sv;
        sm[L.DS + 24 * k + lane] = dsv;
        sm[L.DLAM + 24 * k + lane] = dlv;
        if (dsv < 0.0) ap = fmin(ap, -P.tau * sv / dsv);
        if (dlv < 0.0) ad = fmin(ad, -P.tau * lv / dlv);
      } else {
        sm[L.DS + 24 * k + lane] = 0.0;
        sm[L.DLAM + 24 * k + lane] = 0.0;
      }
    } else if (lane >= 32 && lane < 44) {
      // dx+ = Abar dx + Bbar du
      const int r = lane - 32;
      const double* AB = sm + L.AB + 27 * k;
      double s = 0.0;
#pragma unroll
      for (int c = 0; c < 12; ++c) s += abar_elem(P, AB, r, c) * dxc[c];
#pragma unroll
      for (int c = 0; c < 12; ++c) s += bbar_elem(P, cst, bw0, AB, r, c) * du[c];
      dxn[r] = s;
    }
    QSYNC();
    cur ^= 1;
  }
  *alpha_p = wave_min(ap);
  *alpha_d = wave_min(ad);
}

// nonlinear closed-loop rollout with step alpha, in place; returns |dU|_inf
__device__ inline double rollout_closed(const DevParams& P, const Layout& L, double* sm, double alpha,
                                        int lane) {
  const int N = P.N;
  const double* cst = sm + L.cst;
  const double* bw0 = sm + L.bw0;
  double* du = sm + L.du;  // holds the new input of the current knot
  double xc[13], xn[13];
#pragma unroll
  for (int i = 0; i < 13; ++i) xc[i] = cst[C_X0 + i];
  double step = 0.0;
  for (int k = 0; k < N; ++k) {
    // dx = xc (-) X_k : inverse Cayley map of q_k^-1 * qc (QuaternionUtils.cpp:16-18)
    double xo[13];
#pragma unroll
    for (int i = 0; i < 13; ++i) xo[i] = sm[L.X + 13 * k + i];
    double dx[12];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      dx[a] = xc[a] - xo[a];
      dx[6 + a] = xc[7 + a] - xo[7 + a];
      dx[9 + a] = xc[10 + a] - xo[10 + a];
    }
    {
      double G[12];
      quat_G(&xo[3], G);
      const double sc = xo[3] * xc[3] + xo[4] * xc[4] + xo[5] * xc[5] + xo[6] * xc[6];
#pragma unroll
      for (int a = 0; a < 3; ++a)
        dx[3 + a] = (G[a] * xc[3] + G[3 + a] * xc[4] + G[6 + a] * xc[5] + G[9 + a] * xc[6]) / sc;
    }
    if (lane < 12) {
      const double* kd = sm + L.KD + 156 * k + 13 * lane;
      const double uo = sm[L.U + 12 * k + lane];
      double s = uo + alpha * kd[12];
#pragma unroll
      for (int b = 0; b < 12; ++b) s += kd[b] * dx[b];
      du[lane] = s;
      step = fmax(step, fabs(s - uo));
    }
    QSYNC();
    double un[12];
#pragma unroll
    for (int j = 0; j < 12; ++j) un[j] = du[j];
    srbd_step(P, cst, bw0, xc, un, xn);
    // publish the new knot (old x_k, u_k are no longer needed)
    if (lane == 0)
#pragma unroll
      for (int i = 0; i < 13; ++i) sm[L.X + 13 * k + i] = xc[i];
    if (lane < 12) sm[L.U + 12 * k + lane] = du[lane];
#pragma unroll
    for (int i = 0; i < 13; ++i) xc[i] = xn[i];
    QSYNC();
  }
  if (lane == 0)
#pragma unroll
    for (int i = 0; i < 13; ++i) sm[L.X + 13 * N + i] = xc[i];
  QSYNC();
  return wave_max(step);
}

__device__ inline double cost_plain(const DevParams& P, const Layout& L, double* sm, int lane) {
  double J = 0.0;
  if (lane <= P.N) {
    double xr[13];
    xref_at(P, sm + L.refp, lane, xr);
    const double* x = sm + L.X + 13 * lane;
    for (int i = 0; i < 13; ++i) { const double e = x[i] - xr[i]; J += 0.5 * P.Q[i] * e * e; }
    const double dq = xr[3] * x[3] + xr[4] * x[4] + xr[5] * x[5] + xr[6] * x[6];
    J += P.w * (1.0 - fabs(dq));
    if (lane < P.N) {
      const double* u = sm + L.U + 12 * lane;
      for (int j = 0; j < 12; ++j) { const double e = u[j] - sm[L.uref + j]; J += 0.5 * P.R[j] * e * e; }
    }
  }
  return wave_sum(J);
}

// ---- the solve kernel ---------------------------------------------------------
template <bool PROF>
__global__ __launch_bounds__(64) void qmpc_solve_kernel(DevParams P, const qmpc_input* __restrict__ in,
                                                        double* __restrict__ forces,
                                                        qmpc_info* __restrict__ info,
                                                        double* __restrict__ traj_u,
                                                        double* __restrict__ traj_x, int batch,
                                                        long long* __restrict__ prof_out) {
  extern __shared__ __attribute__((aligned(16))) double sm[];
  const int b = blockIdx.x;
  if (b >= batch) return;
  const int lane = threadIdx.x;
  const int N = P.N;
  const Layout L = make_layout(N);
  int status = QMPC_OK;
  Prof<PROF> prof;
  prof.start();
  setup_instance(P, L, sm, in + b, lane, &status);
  if (status != QMPC_OK) {
    if (lane < 12) forces[12 * (size_t)b + lane] = 0.0;
    if (lane == 0 && info) {
      qmpc_info r = {status, 0, 0.0, 0.0, 0.0, 0.0};
      info[b] = r;
    }
    if (traj_u) for (int i = lane; i < N * 12; i += kWave) traj_u[(size_t)b * N * 12 + i] = 0.0;
    if (traj_x) for (int i = lane; i < (N + 1) * 13; i += kWave) traj_x[(size_t)b * (N + 1) * 13 + i] = 0.0;
    return;
  }
  // initial guess U = u_ref (QuatMpc.cpp:253), slacks and multipliers
  for (int i = lane; i < N * 12; i += kWave) sm[L.U + i] = sm[L.uref + (i % 12)];
  QSYNC();
  rollout_open(P, L, sm, lane);
  expansions(P, L, sm, lane);
  for (int i = lane; i < N * 24; i += kWave) {
    const double s0 = fmax(-sm[L.CV + i], 1.0);
    sm[L.S + i] = s0;
    sm[L.LAM + i] = 1.0 / s0;
  }
  QSYNC();
  prof.tick(PH_SETUP);
  int it = 0, iters = 0;
  double mu = 0.0, resid = 0.0, last_step = 1e300, last_ap = 0.0, last_ad = 0.0;
  status = QMPC_MAX_ITER;
  for (it = 1; it <= P.iterations_max + 1; ++it) {
    // barrier parameter and slack residual over the enabled rows
    double sl = 0.0, rs = 0.0, cnt = 0.0;
    for (int i = lane; i < N * 24; i += kWave) {
      const int l = (i % 24) / 6;
      if (sm[L.cst + C_CON + l] != 0.0) {
        sl += sm[L.S + i] * sm[L.LAM + i];
        rs = fmax(rs, fabs(sm[L.CV + i] + sm[L.S + i]));
        cnt += 1.0;
      }
    }
    mu = wave_sum(sl) / wave_sum(cnt);
    resid = wave_max(rs);
    if (mu <= P.mu_final && resid <= P.tol_feas && last_step <= P.tol_step) { status = QMPC_OK; break; }
    if (it > P.iterations_max) break;
    double sg = P.sigma;
    if (it > 1 && last_ap >= 0.999 && last_ad >= 0.999) sg = P.sigma_fast;
    const double target = sg * mu;
    prof.tick(PH_MISC);
    if (backward_pass<PROF>(P, L, sm, target, lane, prof)) { status = QMPC_NOT_PD; break; }
    double ap, ad;
    ipm_directions(P, L, sm, target, lane, &ap, &ad);
    last_ap = ap; last_ad = ad;
    prof.tick(PH_DIRS);
    last_step = rollout_closed(P, L, sm, ap, lane);
    prof.tick(PH_ROLL);
    for (int i = lane; i < N * 24; i += kWave) {
      sm[L.S + i] += ap * sm[L.DS + i];
      sm[L.LAM + i] += ad * sm[L.DLAM + i];
    }
    QSYNC();
    prof.tick(PH_MISC);
    expansions(P, L, sm, lane);
    prof.tick(PH_EXPAND);
    iters = it;
  }
  // outputs: GetInput(u, 0) (QuatMpc.cpp:264-265)
  if (lane < 12) forces[12 * (size_t)b + lane] = sm[L.U + lane];
  if (traj_u) for (int i = lane; i < N * 12; i += kWave) traj_u[(size_t)b * N * 12 + i] = sm[L.U + i];
  if (traj_x) for (int i = lane; i < (N + 1) * 13; i += kWave) traj_x[(size_t)b * (N + 1) * 13 + i] = sm[L.X + i];
  if (info) {
    const double J = cost_plain(P, L, sm, lane);
    double viol = 0.0;
    for (int i = lane; i < N * 24; i += kWave) {
      const int l = (i % 24) / 6;
      if (sm[L.cst + C_CON + l] != 0.0) viol = fmax(viol, fmax(sm[L.CV + i], 0.0));
    }
    viol = wave_max(viol);
    if (lane == 0) {
      qmpc_info r = {status, iters, J, viol, last_step, mu};
      info[b] = r;
    }
  }
  if (PROF && prof_out && lane == 0) {
    prof.tick(PH_MISC);
#pragma unroll
    for (int i = 0; i < PH_COUNT; ++i) prof_out[16 * (size_t)b + i] = prof.t[i];
    prof_out[16 * (size_t)b + 15] = iters;
  }
}

// ---- linearisation only (qmpc_linearize): rollout of U = u_ref + dense Abar/Bbar
__global__ __launch_bounds__(64) void qmpc_linearize_kernel(DevParams P, const qmpc_input* __restrict__ in,
                                                            double* __restrict__ Abar,
                                                            double* __restrict__ Bbar,
                                                            double* __restrict__ Xout, int batch) {
  extern __shared__ __attribute__((aligned(16))) double sm[];
  const int b = blockIdx.x;
  if (b >= batch) return;
  const int lane = threadIdx.x;
  const int N = P.N;
  const Layout L = make_layout(N);
  int status = QMPC_OK;
  setup_instance(P, L, sm, in + b, lane, &status);
  if (status != QMPC_OK) {
    for (int i = lane; i < N * 144; i += kWave) { Abar[(size_t)b * N * 144 + i] = 0.0; Bbar[(size_t)b * N * 144 + i] = 0.0; }
    for (int i = lane; i < (N + 1) * 13; i += kWave) Xout[(size_t)b * (N + 1) * 13 + i] = 0.0;
    return;
  }
  for (int i = lane; i < N * 12; i += kWave) sm[L.U + i] = sm[L.uref + (i % 12)];
  QSYNC();
  rollout_open(P, L, sm, lane);
  expansions(P, L, sm, lane);
  for (int i = lane; i < N * 144; i += kWave) {
    const int k = i / 144, e = i - 144 * k, r = e / 12, c = e - 12 * r;
    const double* AB = sm + L.AB + 27 * k;
    Abar[(size_t)b * N * 144 + i] = abar_elem(P, AB, r, c);
    Bbar[(size_t)b * N * 144 + i] = bbar_elem(P, sm + L.cst, sm + L.bw0, AB, r, c);
  }
  for (int i = lane; i < (N + 1) * 13; i += kWave) Xout[(size_t)b * (N + 1) * 13 + i] = sm[L.X + i];
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

}  // namespace qmpc
