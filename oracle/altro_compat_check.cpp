// altro_compat_check.cpp -- drives altro_compat.hpp the way the reference's solver tests drive the real
// ALTROSolver (legged_ctrl/src/test/test_altro/TestDoubleIntegrator.cpp:69-375, TestPendulum.cpp:117-203,
// TestAltroApi.cpp) and prints one "name key=value ..." line per case; tests/test_oracle_altro_api.py checks
// the numbers against the expectations those tests state (GetIterations()==3 / ==5, distances, saturation).
// TEST INFRASTRUCTURE ONLY.
#include <cmath>
#include <cstdio>
#include <vector>

#include "altro_compat.hpp"
#include "qo_srbd.h"

using namespace altro;

namespace {

constexpr int kDim = 2, kN = 4, kM = 2;

// discrete double integrator, dim 2: x = [p; v], x+ = [p + v h + u h^2/2; v + u h]
void di_step(double* xn, const double* x, const double* u, float h) {
  const double b = h * h / 2;
  for (int i = 0; i < kDim; ++i) {
    xn[i] = x[i] + x[i + kDim] * h + u[i] * b;
    xn[i + kDim] = x[i + kDim] + u[i] * h;
  }
}
void di_step_jac(double* J, const double*, const double*, float h) {   // 4 x 6 column-major
  for (int i = 0; i < kN * (kN + kM); ++i) J[i] = 0.0;
  const double b = h * h / 2;
  for (int i = 0; i < kDim; ++i) {
    J[i + kN * i] = 1.0;
    J[(i + kDim) + kN * (i + kDim)] = 1.0;
    J[i + kN * (i + kDim)] = h;
    J[i + kN * (kN + i)] = b;
    J[(i + kDim) + kN * (kN + i)] = h;
  }
}

struct DiSetup {
  int horizon = 10;
  float h = 5.0f / 10.0f;
  std::vector<double> Q = std::vector<double>(kN, 1.0), R = std::vector<double>(kM, 1e-2);
  std::vector<double> xf = std::vector<double>(kN, 0.0), uf = std::vector<double>(kM, 0.0);
};

int di_common(ALTROSolver& s, const DiSetup& d, const std::vector<double>& x0) {
  int bad = 0;
  bad += s.SetDimension(kN, kM, 0, LastIndex) != ErrorCodes::NoError;
  bad += s.SetTimeStep(d.h, 0, LastIndex) != ErrorCodes::NoError;
  bad += s.SetExplicitDynamics(di_step, di_step_jac, 0, LastIndex) != ErrorCodes::NoError;
  bad += s.SetLQRCost(kN, kM, d.Q.data(), d.R.data(), d.xf.data(), d.uf.data(), 0, LastIndex) != ErrorCodes::NoError;
  bad += s.SetInitialState(x0.data(), kN) != ErrorCodes::NoError;
  return bad;
}

double norm(const std::vector<double>& v) {
  double s = 0;
  for (double x : v) s += x * x;
  return std::sqrt(s);
}

void case_unconstrained() {
  DiSetup d;
  ALTROSolver s(d.horizon);
  const std::vector<double> x0 = {1.0, 2.0, 0.0, 0.0};
  int bad = di_common(s, d, x0);
  bad += s.Initialize() != ErrorCodes::NoError;
  std::vector<double> u0(kM, 0.0);
  s.SetState(x0.data(), kN, 0, LastIndex);
  s.SetInput(u0.data(), kM, 0, LastIndex);
  const double c0 = s.CalcCost();
  AltroOptions o;
  o.iterations_max = 3;
  s.SetOptions(o);
  const SolveStatus st = s.Solve();
  std::vector<double> xN(kN);
  s.GetState(xN.data(), d.horizon);
  std::printf("di_unconstrained bad=%d initialized=%d status=%d iterations=%d cost0=%.17g cost=%.17g dist=%.17g dist0=%.17g\n",
              bad, (int)s.IsInitialized(), (int)st, s.GetIterations(), c0, s.CalcCost(), norm(xN), norm(x0));
}

void case_goal() {
  DiSetup d;
  ALTROSolver s(d.horizon);
  const std::vector<double> x0 = {1.0, 2.0, 0.0, 0.0};
  int bad = di_common(s, d, x0);
  auto con = [](a_float* c, const a_float* x, const a_float*) { for (int i = 0; i < kN; ++i) c[i] = x[i]; };
  auto jac = [](a_float* J, const a_float*, const a_float*) {           // 4 x 6 column-major
    for (int i = 0; i < kN * (kN + kM); ++i) J[i] = 0.0;
    for (int i = 0; i < kN; ++i) J[i + kN * i] = 1.0;
  };
  bad += s.SetConstraint(con, jac, kN, ConstraintType::EQUALITY, "goal", d.horizon, 0, nullptr) != ErrorCodes::NoError;
  bad += s.Initialize() != ErrorCodes::NoError;
  std::vector<double> u0(kM, 0.0);
  s.SetState(x0.data(), kN, 0, LastIndex);
  s.SetInput(u0.data(), kM, 0, LastIndex);
  AltroOptions o;
  o.penalty_scaling = 100;
  s.SetOptions(o);
  const SolveStatus st = s.Solve();
  std::vector<double> xN(kN);
  s.GetState(xN.data(), d.horizon);
  std::printf("di_goal bad=%d status=%d iterations=%d dist=%.17g feas=%.17g\n", bad, (int)st, s.GetIterations(),
              norm(xN), s.GetPrimalFeasibility());
}

void case_bounds() {
  DiSetup d;
  ALTROSolver s(d.horizon);
  const std::vector<double> x0 = {2.0, 2.0, 0.0, 0.0};
  int bad = di_common(s, d, x0);
  const double ub = 1.0;
  auto goal = [](a_float* c, const a_float* x, const a_float*) { for (int i = 0; i < kN; ++i) c[i] = x[i]; };
  auto goal_j = [](a_float* J, const a_float*, const a_float*) {
    for (int i = 0; i < kN * (kN + kM); ++i) J[i] = 0.0;
    for (int i = 0; i < kN; ++i) J[i + kN * i] = 1.0;
  };
  auto bnd = [ub](a_float* c, const a_float*, const a_float* u) {
    for (int i = 0; i < kM; ++i) { c[i] = u[i] - ub; c[i + kM] = -ub - u[i]; }
  };
  auto bnd_j = [](a_float* J, const a_float*, const a_float*) {          // 4 x 6 column-major
    for (int i = 0; i < 2 * kM * (kN + kM); ++i) J[i] = 0.0;
    for (int i = 0; i < kM; ++i) { J[i + 4 * (kN + i)] = 1.0; J[(i + kM) + 4 * (kN + i)] = -1.0; }
  };
  std::vector<ConstraintIndex> idx;
  bad += s.SetConstraint(goal, goal_j, kN, ConstraintType::EQUALITY, "goal", d.horizon, 0, nullptr) != ErrorCodes::NoError;
  bad += s.SetConstraint(bnd, bnd_j, 2 * kM, ConstraintType::INEQUALITY, "bounds", 0, d.horizon, &idx) != ErrorCodes::NoError;
  bad += s.Initialize() != ErrorCodes::NoError;
  std::vector<double> u0(kM, 0.0);
  s.SetState(x0.data(), kN, 0, LastIndex);
  s.SetInput(u0.data(), kM, 0, LastIndex);
  AltroOptions o;
  o.penalty_scaling = 100;
  o.penalty_initial = 100;
  s.SetOptions(o);
  const SolveStatus st = s.Solve();
  std::vector<double> xN(kN), u(kM);
  s.GetState(xN.data(), d.horizon);
  s.GetInput(u.data(), 0);
  std::printf("di_bounds bad=%d status=%d iterations=%d dist=%.17g u0=%.17g u1=%.17g ncon_idx=%d\n", bad, (int)st,
              s.GetIterations(), norm(xN), u[0], u[1], (int)idx.size());
}

void case_pendulum_goal() {
  // simple pendulum, explicit midpoint; terminal equality x_N = (pi, 0), N = 20, tf = 2
  const double l = 0.5, g = 9.81, bf = 0.1, ml2 = 1.0 * l * l;
  auto f = [=](double* xd, const double* x, const double* u) {
    xd[0] = x[1];
    xd[1] = u[0] / ml2 - g * std::sin(x[0]) / l - bf * x[1] / ml2;
  };
  auto dyn = [=](double* xn, const double* x, const double* u, float h) {
    double k1[2], xm[2], k2[2];
    f(k1, x, u);
    for (int i = 0; i < 2; ++i) xm[i] = x[i] + 0.5 * h * k1[i];
    f(k2, xm, u);
    for (int i = 0; i < 2; ++i) xn[i] = x[i] + h * k2[i];
  };
  auto jacf = [=](double A[4], double B[2], const double* x) {        // column-major 2x2, 2x1
    A[0] = 0.0; A[1] = -g * std::cos(x[0]) / l; A[2] = 1.0; A[3] = -bf / ml2;
    B[0] = 0.0; B[1] = 1.0 / ml2;
  };
  auto jac = [=](double* J, const double* x, const double* u, float h) {   // 2 x 3 column-major
    double k1[2], xm[2], A1[4], B1[2], Am[4], Bm[2];
    f(k1, x, u);
    for (int i = 0; i < 2; ++i) xm[i] = x[i] + 0.5 * h * k1[i];
    jacf(A1, B1, x);
    jacf(Am, Bm, xm);
    // dx+/dx = I + h Am (I + h/2 A1),  dx+/du = h (Am h/2 B1 + Bm)
    double M[4] = {1.0 + 0.5 * h * A1[0], 0.5 * h * A1[1], 0.5 * h * A1[2], 1.0 + 0.5 * h * A1[3]};
    for (int c = 0; c < 2; ++c)
      for (int r = 0; r < 2; ++r)
        J[r + 2 * c] = (r == c ? 1.0 : 0.0) + h * (Am[r] * M[2 * c] + Am[r + 2] * M[2 * c + 1]);
    for (int r = 0; r < 2; ++r) J[r + 4] = h * (0.5 * h * (Am[r] * B1[0] + Am[r + 2] * B1[1]) + Bm[r]);
  };
  const int N = 20;
  ALTROSolver s(N);
  int bad = 0;
  bad += s.SetDimension(2, 1, 0, LastIndex) != ErrorCodes::NoError;
  bad += s.SetTimeStep(2.0f / 20.0f, 0, LastIndex) != ErrorCodes::NoError;
  bad += s.SetExplicitDynamics(dyn, jac, 0, LastIndex) != ErrorCodes::NoError;
  const double Q[2] = {1e-2, 1e-2}, Qf[2] = {1.0, 1.0}, R[1] = {1e-3}, xf[2] = {M_PI, 0.0}, uf[1] = {0.0}, x0[2] = {0.0, 0.0};
  bad += s.SetLQRCost(2, 1, Q, R, xf, uf, 0, N) != ErrorCodes::NoError;
  bad += s.SetLQRCost(2, 1, Qf, R, xf, uf, N, 0) != ErrorCodes::NoError;
  bad += s.SetInitialState(x0, 2) != ErrorCodes::NoError;
  auto con = [](a_float* c, const a_float* x, const a_float*) { c[0] = M_PI - x[0]; c[1] = -x[1]; };
  auto cj = [](a_float* J, const a_float*, const a_float*) {
    for (int i = 0; i < 6; ++i) J[i] = 0.0;
    J[0] = -1.0; J[3] = -1.0;
  };
  bad += s.SetConstraint(con, cj, 2, ConstraintType::EQUALITY, "goal", N, 0, nullptr) != ErrorCodes::NoError;
  bad += s.Initialize() != ErrorCodes::NoError;
  const double u0[1] = {0.1};
  s.SetInput(u0, 1, 0, LastIndex);
  AltroOptions o;
  o.iterations_max = 100;
  s.SetOptions(o);
  const SolveStatus st = s.Solve();
  double xN[2];
  s.GetState(xN, N);
  std::printf("pendulum_goal bad=%d status=%d iterations=%d dist=%.17g\n", bad, (int)st, s.GetIterations(),
              std::hypot(xN[0] - M_PI, xN[1]));
}

void case_api_errors() {
  ALTROSolver s(10);
  const double v[4] = {0, 0, 0, 0};
  const int e1 = (int)s.SetLQRCost(4, 2, v, v, v, v, 0, LastIndex);           // before SetDimension
  const int e2 = (int)s.SetState(v, 4, 0, LastIndex);                         // before Initialize
  s.SetDimension(4, 2, 0, LastIndex);
  const int e3 = (int)s.SetInitialState(v, 3);                                // wrong size
  const int e4 = (int)s.SetTimeStep(0.1f, 11, 0);                             // knot beyond the horizon
  auto c = [](a_float*, const a_float*, const a_float*) {};
  const int e5 = (int)s.SetConstraint(c, c, 9, ConstraintType::SECOND_ORDER_CONE, "soc", 0, 10, nullptr);   // > QO_SOC_MAXP rows
  const int e6 = (int)s.Initialize();                                         // no dynamics / time step yet
  std::printf("api_errors e1=%d e2=%d e3=%d e4=%d e5=%d e6=%d unsolved=%d\n", e1, e2, e3, e4, e5, e6, (int)s.Solve());
}

// The call pattern of legged::QuatMpc::grf_update (QuatMpc.cpp:179-265) on the stand-pose problem of the reference's
// golden generator (TestAltroQuatMpc.cpp:36-113): quaternion cost at every knot, one 24-row INEQUALITY block on
// [0, N), error-state Jacobians, initial guess X = x_ref, U = u_ref.  Prints u_0 (12) for the golden comparison.
void case_quatmpc_stand(bool tight) {
  const int n = 13, m = 12, N = 20;
  qo_srbd_model md;
  std::memset(&md, 0, sizeof md);
  md.nleg = 4;
  const double feet[4][3] = {{0.2104, 0.13, -0.325}, {0.2104, -0.13, -0.325}, {-0.1658, 0.13, -0.325}, {-0.1658, -0.13, -0.325}};
  const double trunk[3] = {0.0168128557, 0.063009565, 0.0716547275};
  for (int l = 0; l < 4; ++l) {
    for (int a = 0; a < 3; ++a) md.foot_pos_body[3 * l + a] = feet[l][a];
    md.contacts[l] = 1.0;
  }
  for (int a = 0; a < 3; ++a) { md.inertia[4 * a] = (12.84 / 5.204) * trunk[a]; md.rot[4 * a] = 1.0; }
  md.mass = 12.84;
  qo_srbd_prepare(&md);
  const double mu = 0.6, fz_max = 200.0;
  auto dyn = [&md](double* xn, const double* x, const double* u, float h) { qo_srbd_discrete_dynamics(&md, xn, x, u, h); };
  auto jac = [&md](double* J, const double* x, const double* u, float h) { qo_srbd_discrete_jacobian(&md, J, x, u, h); };
  auto cone = [&md, mu, fz_max](a_float* c, const a_float*, const a_float* u) { qo_cone_eval(mu, fz_max, md.rot, md.contacts, u, c); };
  auto cone_jac = [&md, mu](a_float* J, const a_float*, const a_float*) {       // 24 x (12 + 12) column-major
    double CR[18];
    qo_cone_block(mu, md.rot, CR);
    for (int i = 0; i < 24 * 24; ++i) J[i] = 0.0;
    for (int l = 0; l < 4; ++l)
      for (int r = 0; r < 6; ++r)
        for (int a = 0; a < 3; ++a) J[(6 * l + r) + 24 * (12 + 3 * l + a)] = CR[3 * r + a];
  };
  const double Q[13] = {1, 1, 1, 0, 0, 0, 0, 2, 2, 2, 1, 1, 1};
  double R[12], xref[13] = {0, 0, 0, 1, 0, 0, 0, 0, 0, 0, 0, 0, 0}, uref[12];
  for (int j = 0; j < 12; ++j) { R[j] = 1e-6; uref[j] = (j % 3 == 2) ? 12.84 * 9.81 / 4.0 : 0.0; }
  AltroOptions o;                        // QuatMpc.cpp:21-26
  o.iterations_max = 10;
  o.penalty_scaling = 20.0;
  o.use_quaternion = true;
  o.quat_start_index = 3;
  if (tight) { o.tol_stationarity = 1e-9; o.tol_cost_intermediate = 1e-12; o.iterations_max = 50; }
  ALTROSolver s(N);
  int bad = 0;
  s.SetOptions(o);
  bad += s.SetDimension(n, m) != ErrorCodes::NoError;
  bad += s.SetExplicitDynamics(dyn, jac) != ErrorCodes::NoError;
  bad += s.SetTimeStep(10.0 / 1000.0) != ErrorCodes::NoError;
  for (int k = 0; k <= N; ++k) bad += s.SetQuaternionCost(n, m, Q, R, 1.0, xref, uref, k, 0) != ErrorCodes::NoError;
  bad += s.SetConstraint(cone, cone_jac, 24, ConstraintType::INEQUALITY, "friction cone", 0, N) != ErrorCodes::NoError;
  bad += s.SetInitialState(xref, n) != ErrorCodes::NoError;
  bad += s.Initialize() != ErrorCodes::NoError;
  for (int k = 0; k <= N; ++k) s.SetState(xref, n, k);
  s.SetInput(uref, m);
  const SolveStatus st = s.Solve();
  double u0[12];
  s.GetInput(u0, 0);
  std::printf("%s bad=%d status=%d iterations=%d feas=%.17g", tight ? "quatmpc_stand_tight" : "quatmpc_stand", bad, (int)st, s.GetIterations(), s.GetPrimalFeasibility());
  for (int j = 0; j < 12; ++j) std::printf(" u%d=%.17g", j, u0[j]);
  std::printf("\n");
}

// TestDoubleIntegrator.cpp:377-492: goal constraint + |u| <= u_bnd as a second-order cone c = (u, u_bnd)
void case_soc() {
  DiSetup d;
  ALTROSolver s(d.horizon);
  const std::vector<double> x0 = {2.0, 2.0, 0.0, 0.0};
  int bad = di_common(s, d, x0);
  const double ub = 1.0;
  auto goal = [](a_float* c, const a_float* x, const a_float*) { for (int i = 0; i < kN; ++i) c[i] = x[i]; };
  auto goal_j = [](a_float* J, const a_float*, const a_float*) {
    for (int i = 0; i < kN * (kN + kM); ++i) J[i] = 0.0;
    for (int i = 0; i < kN; ++i) J[i + kN * i] = 1.0;
  };
  auto soc = [ub](a_float* c, const a_float*, const a_float* u) { c[0] = u[0]; c[1] = u[1]; c[2] = ub; };
  auto soc_j = [](a_float* J, const a_float*, const a_float*) {            // 3 x 6 column-major
    for (int i = 0; i < 3 * (kN + kM); ++i) J[i] = 0.0;
    for (int i = 0; i < kM; ++i) J[i + 3 * (kN + i)] = 1.0;
  };
  bad += s.SetConstraint(goal, goal_j, kN, ConstraintType::EQUALITY, "goal", d.horizon, 0, nullptr) != ErrorCodes::NoError;
  bad += s.SetConstraint(soc, soc_j, kM + 1, ConstraintType::SECOND_ORDER_CONE, "bounds", 0, d.horizon, nullptr) != ErrorCodes::NoError;
  bad += s.Initialize() != ErrorCodes::NoError;
  std::vector<double> u0(kM, 0.0);
  s.SetState(x0.data(), kN, 0, LastIndex);
  s.SetInput(u0.data(), kM, 0, LastIndex);
  AltroOptions o;
  o.penalty_initial = 1.0;
  o.penalty_scaling = 100;
  s.SetOptions(o);
  const SolveStatus st = s.Solve();
  std::vector<double> xN(kN), u(kM);
  s.GetState(xN.data(), d.horizon);
  s.GetInput(u.data(), 0);
  std::printf("di_soc bad=%d status=%d iterations=%d dist=%.17g unorm=%.17g feas=%.17g\n", bad, (int)st, s.GetIterations(),
              norm(xN), norm(u), s.GetPrimalFeasibility());
}

}  // namespace

int main() {
  case_unconstrained();
  case_goal();
  case_bounds();
  case_pendulum_goal();
  case_api_errors();
  case_soc();
  case_quatmpc_stand(false);
  case_quatmpc_stand(true);
  return 0;
}
