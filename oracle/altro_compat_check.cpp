// altro_compat_check.cpp -- drives altro_compat.hpp the way the reference's solver tests drive the real
// ALTROSolver (legged_ctrl/src/test/test_altro/TestDoubleIntegrator.cpp:69-375, TestPendulum.cpp:117-203,
// TestAltroApi.cpp) and prints one "name key=value ..." line per case; tests/test_oracle_altro_api.py checks
// the numbers against the expectations those tests state (GetIterations()==3 / ==5, distances, saturation).
// TEST INFRASTRUCTURE ONLY.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <string>
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

void case_goal(bool cubic = false) {
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
  o.use_backtracking_linesearch = !cubic;
  o.penalty_scaling = 100;
  s.SetOptions(o);
  const SolveStatus st = s.Solve();
  std::vector<double> xN(kN);
  s.GetState(xN.data(), d.horizon);
  std::printf("di_goal%s bad=%d status=%d iterations=%d dist=%.17g feas=%.17g\n", cubic ? "_cubic" : "", bad, (int)st, s.GetIterations(),
              norm(xN), s.GetPrimalFeasibility());
}

void case_bounds(bool cubic = false) {
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
  o.use_backtracking_linesearch = !cubic;
  o.penalty_scaling = 100;
  o.penalty_initial = 100;
  s.SetOptions(o);
  const SolveStatus st = s.Solve();
  std::vector<double> xN(kN), u(kM);
  s.GetState(xN.data(), d.horizon);
  s.GetInput(u.data(), 0);
  std::printf("di_bounds%s bad=%d status=%d iterations=%d dist=%.17g u0=%.17g u1=%.17g ncon_idx=%d\n", cubic ? "_cubic" : "", bad, (int)st,
              s.GetIterations(), norm(xN), u[0], u[1], (int)idx.size());
}

void case_pendulum_goal(bool cubic = false) {
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
  o.use_backtracking_linesearch = !cubic;
  o.iterations_max = 100;
  s.SetOptions(o);
  const SolveStatus st = s.Solve();
  double xN[2];
  s.GetState(xN, N);
  std::printf("pendulum_goal%s bad=%d status=%d iterations=%d dist=%.17g\n", cubic ? "_cubic" : "", bad, (int)st, s.GetIterations(),
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
  o.use_backtracking_linesearch = true;  // QuatMpc.cpp:23
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
void case_soc(bool cubic = false) {
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
  o.use_backtracking_linesearch = !cubic;
  o.penalty_initial = 1.0;
  o.penalty_scaling = 100;
  s.SetOptions(o);
  const SolveStatus st = s.Solve();
  std::vector<double> xN(kN), u(kM);
  s.GetState(xN.data(), d.horizon);
  s.GetInput(u.data(), 0);
  std::printf("di_soc%s bad=%d status=%d iterations=%d dist=%.17g unorm=%.17g feas=%.17g\n", cubic ? "_cubic" : "", bad, (int)st, s.GetIterations(),
              norm(xN), norm(u), s.GetPrimalFeasibility());
}

}  // namespace

// ---- TestBicycle.cpp: closed-loop nonlinear MPC of a kinematic bicycle along the "Scotty dog" reference ---------
// The reference commits BOTH the reference trajectory (scotty.json) and the output of that program
// (scotty_mpc.json: 200 closed-loop states / inputs, solver iterations and tracking error per step); copies of the
// two data files are in tests/golden/.  The loop below is TestBicycle.cpp:25-200 through the compat API.
struct Bicycle {             // AltroTestUtils.cpp:134-237, ReferenceFrame::CenterOfGravity
  double L = 2.7, lr = 1.5;
  void f(double* xd, const double* x, const double* u) const {
    const double v = u[0], th = x[2], de = x[3];
    const double beta = std::atan2(lr * de, L), om = v * std::cos(beta) * std::tan(de) / L;
    xd[0] = v * std::cos(th + beta); xd[1] = v * std::sin(th + beta); xd[2] = om; xd[3] = u[1];
  }
  void J(double* j, const double* x, const double* u) const {   // 4 x 6 column-major
    for (int i = 0; i < 24; ++i) j[i] = 0.0;
    const double v = u[0], th = x[2], de = x[3];
    const double by = lr * de, bx = L, beta = std::atan2(by, bx), db = bx / (bx * bx + by * by) * lr;
    const double dom_dde = v / L * (-std::sin(beta) * std::tan(de) * db + std::cos(beta) / (std::cos(de) * std::cos(de)));
    const double dom_dv = std::cos(beta) * std::tan(de) / L;
    const double st = std::sin(th + beta), ct = std::cos(th + beta);
    j[0 + 4 * 2] = v * -st; j[0 + 4 * 3] = v * -st * db; j[0 + 4 * 4] = ct;
    j[1 + 4 * 2] = v * ct;  j[1 + 4 * 3] = v * ct * db;  j[1 + 4 * 4] = st;
    j[2 + 4 * 3] = dom_dde; j[2 + 4 * 4] = dom_dv;
    j[3 + 4 * 5] = 1.0;
  }
};
// numbers of the array stored under "key" in a JSON file of nested numeric arrays
std::vector<double> json_numbers(const std::string& text, const std::string& key) {
  std::vector<double> out;
  size_t p = text.find("\"" + key + "\"");
  if (p == std::string::npos) return out;
  p = text.find(':', p) + 1;
  while (p < text.size() && (text[p] == ' ' || text[p] == '\n')) ++p;
  if (text[p] != '[') { out.push_back(std::strtod(text.c_str() + p, nullptr)); return out; }
  int depth = 0;
  for (; p < text.size(); ++p) {
    const char ch = text[p];
    if (ch == '[') ++depth;
    else if (ch == ']') { if (--depth == 0) break; }
    else if (ch == '-' || (ch >= '0' && ch <= '9')) {
      char* end = nullptr;
      out.push_back(std::strtod(text.c_str() + p, &end));
      p = (size_t)(end - text.c_str()) - 1;
    }
  }
  return out;
}
std::string slurp(const std::string& path) {
  std::string s;
  if (FILE* f = std::fopen(path.c_str(), "rb")) {
    char buf[65536];
    size_t n;
    while ((n = std::fread(buf, 1, sizeof buf, f)) > 0) s.append(buf, n);
    std::fclose(f);
  }
  return s;
}
void case_bicycle_mpc(const std::string& dir, int warm = 0, bool shift = false, bool verbose = false) {
  const std::string ref = slurp(dir + "/scotty.json"), gold = slurp(dir + "/scotty_mpc.json");
  const std::vector<double> xr = json_numbers(ref, "state_trajectory"), ur = json_numbers(ref, "input_trajectory");
  const std::vector<double> xg = json_numbers(gold, "state_trajectory"), ug = json_numbers(gold, "input_trajectory");
  const std::vector<double> ig = json_numbers(gold, "solve_iters"), eg = json_numbers(gold, "tracking_error");
  if (xr.size() < 4 * 240 || xg.size() != 4 * 201 || ug.size() != 2 * 200 || ig.size() != 200) {
    std::printf("bicycle_mpc bad=1\n");
    return;
  }
  const int n = 4, m = 2, N = 30, Nsim = 200;
  const float h = 0.1f;       // the golden's states reproduce under the midpoint rule with h = 0.1f to 0.0 (checked below)
  Bicycle car;
  auto dyn0 = [car](double* xd, const double* x, const double* u) { car.f(xd, x, u); };
  auto dyn = [dyn0](double* xn, const double* x, const double* u, float hh) {     // AltroUtils.cpp:9-22
    double xm[4], k[4];
    dyn0(k, x, u);
    for (int i = 0; i < 4; ++i) xm[i] = x[i] + hh / 2 * k[i];
    dyn0(k, xm, u);
    for (int i = 0; i < 4; ++i) xn[i] = x[i] + hh * k[i];
  };
  auto jac = [car, dyn0](double* Jo, const double* x, const double* u, float hh) {   // AltroUtils.cpp:78-110
    double k[4], xm[4], J0[24], Jm[24];
    dyn0(k, x, u);
    for (int i = 0; i < 4; ++i) xm[i] = x[i] + hh / 2 * k[i];
    car.J(J0, x, u);
    car.J(Jm, xm, u);
    for (int r = 0; r < 4; ++r) {
      for (int c = 0; c < 4; ++c) {       // I + h Am (I + h/2 A)
        double s = 0.0;
        for (int t = 0; t < 4; ++t) s += Jm[r + 4 * t] * ((t == c ? 1.0 : 0.0) + hh / 2 * J0[t + 4 * c]);
        Jo[r + 4 * c] = (r == c ? 1.0 : 0.0) + hh * s;
      }
      for (int c = 0; c < 2; ++c) {       // h (Am h/2 B + Bm)
        double s = 0.0;
        for (int t = 0; t < 4; ++t) s += Jm[r + 4 * t] * hh / 2 * J0[t + 4 * (4 + c)];
        Jo[r + 4 * (4 + c)] = hh * (s + Jm[r + 4 * (4 + c)]);
      }
    }
  };
  // the golden itself pins model + midpoint + float h: roll its inputs through dyn
  double pin = 0.0;
  for (int k = 0; k < Nsim; ++k) {
    double xn[4];
    dyn(xn, &xg[4 * k], &ug[2 * k], h);
    for (int i = 0; i < 4; ++i) pin = std::fmax(pin, std::fabs(xn[i] - xg[4 * (k + 1) + i]));
  }
  ALTROSolver solver(N);
  int bad = 0;
  bad += solver.SetDimension(n, m) != ErrorCodes::NoError;
  bad += solver.SetExplicitDynamics(dyn, jac) != ErrorCodes::NoError;
  bad += solver.SetTimeStep(h) != ErrorCodes::NoError;
  const std::vector<double> Qd(n, 1e-2), Rd(m, 1e-3);
  for (int k = 0; k <= N; ++k) bad += solver.SetLQRCost(n, m, Qd.data(), Rd.data(), &xr[4 * k], &ur[2 * k], k) != ErrorCodes::NoError;
  const double dmax = 60 * M_PI / 180.0;
  auto con = [dmax](a_float* c, const a_float* x, const a_float*) { c[0] = x[3] - dmax; c[1] = -dmax - x[3]; };
  auto cjac = [](a_float* j, const a_float*, const a_float*) { for (int i = 0; i < 12; ++i) j[i] = 0.0; j[0 + 2 * 3] = 1.0; j[1 + 2 * 3] = -1.0; };
  bad += solver.SetConstraint(con, cjac, 2, ConstraintType::INEQUALITY, "steering angle bound", 0, N + 1) != ErrorCodes::NoError;
  bad += solver.SetInitialState(&xr[0], n) != ErrorCodes::NoError;
  bad += solver.Initialize() != ErrorCodes::NoError;
  const double u0[2] = {ur[0], 0.0};
  bad += solver.SetInput(u0, m) != ErrorCodes::NoError;
  for (int k = 0; k <= N; ++k) solver.SetState(&xr[4 * k], n, k);
  AltroOptions opts;
  opts.iterations_max = 80;
  opts.use_backtracking_linesearch = true;      // TestBicycle.cpp:154
  solver.SetOptions(opts);
  solver.SetWarmStart(warm, shift);
  std::vector<double> xs(4 * (Nsim + 1), 0.0), us(2 * Nsim, 0.0);
  for (int i = 0; i < 4; ++i) xs[i] = xr[i];
  double xdiff = 0.0, udiff = 0.0, ediff = 0.0, emax = 0.0;
  int iters_sum = 0, iters_match = 0, iters_maxdiff = 0, fails = 0;
  for (int it = 0; it < Nsim; ++it) {
    const SolveStatus st = solver.Solve();
    fails += st != SolveStatus::Success;
    const int its = solver.GetIterations();
    iters_sum += its;
    iters_match += its == (int)ig[it];
    iters_maxdiff = std::max(iters_maxdiff, std::abs(its - (int)ig[it]));
    if (verbose) std::printf("  step %3d status %d iters %2d golden %2d\n", it, (int)st, its, (int)ig[it]);
    solver.GetInput(&us[2 * it], 0);
    dyn(&xs[4 * (it + 1)], &xs[4 * it], &us[2 * it], h);
    double e = 0.0;
    for (int i = 0; i < 4; ++i) { const double d = xs[4 * (it + 1) + i] - xr[4 * (it + 1) + i]; e += d * d; }
    e = std::sqrt(e);
    emax = std::fmax(emax, e);
    ediff = std::fmax(ediff, std::fabs(e - eg[it]));
    for (int i = 0; i < 4; ++i) xdiff = std::fmax(xdiff, std::fabs(xs[4 * (it + 1) + i] - xg[4 * (it + 1) + i]));
    for (int j = 0; j < 2; ++j) udiff = std::fmax(udiff, std::fabs(us[2 * it + j] - ug[2 * it + j]));
    for (int k = 0; k <= N; ++k) {       // TestBicycle.cpp:184-196
      const double* xk = &xr[4 * (k + it + 1)];
      double q[4];
      for (int i = 0; i < 4; ++i) q[i] = -Qd[i] * xk[i];
      solver.UpdateLinearCosts(q, nullptr, 0.0, k);
    }
    solver.SetInitialState(&xs[4 * (it + 1)], n);
    solver.ShiftTrajectory();
  }
  int gsum = 0;
  for (double v : ig) gsum += (int)v;
  std::printf("bicycle_mpc warm=%d shift=%d bad=%d pin=%.3e fails=%d iters_sum=%d golden_iters_sum=%d iters_match=%d iters_maxdiff=%d xdiff=%.6e udiff=%.6e "
              "terr_diff=%.6e terr_max=%.6e\n", warm, (int)shift, bad, pin, fails, iters_sum, gsum, iters_match, iters_maxdiff, xdiff, udiff, ediff, emax);
}

int main(int argc, char** argv) {
  if (argc > 1) {
    const int warm = argc > 2 ? std::atoi(argv[2]) : 0;
    case_bicycle_mpc(argv[1], warm, argc > 3 && std::atoi(argv[3]) != 0, argc > 4);
    return 0;
  }
  case_unconstrained();
  case_goal();
  case_bounds();
  case_pendulum_goal();
  case_api_errors();
  case_soc();
  case_quatmpc_stand(false);
  case_quatmpc_stand(true);
  // the generic KATs once more with upstream's default (interpolating) line search as recalled (qo_altro.c: linesearch_cubic)
  case_goal(true);
  case_bounds(true);
  case_pendulum_goal(true);
  case_soc(true);
  return 0;
}
