// lane_core_host.cpp -- TEST INFRASTRUCTURE: g++ build of the lane-per-instance solver core
// (quaternion-mpc_amd/csrc/qmpc_lane_core.h, the text hipcc compiles into qmpc_lane_kernel), one instance after the
// other with unit strides.  It exists so that the numerics of the wrench-form elimination can be checked against the
// oracle on a machine without a GPU (tests/test_lane_core_cpu.py); nothing in the product loads it.
#include <cstdlib>
#include <vector>

#include "../../quaternion-mpc_amd/csrc/qmpc_lane_core.h"

using namespace qmpc;
using namespace qmpc::lane;

template <int NL, int MD = MD_QUAT>
static int solve_all(const DevParams& P, int batch, const double* rec, double* forces, qmpc_info* info,
                     bool warm = false, const double* u_init = nullptr, double* traj_u = nullptr) {
  const WsOff O = make_wsoff<NL>(P.N, P.mode == QMPC_MODE_REFERENCE);
  std::vector<double> ws((size_t)O.total), pl((size_t)LDim<NL>::PLDS);
  for (int b = 0; b < batch; ++b) {
    Ctx c = {ws.data(), 8, 0, pl.data(), 8, 0};
    LaneK<NL> K;
    LaneState st;
    const size_t ts = (size_t)P.N * 3 * NL;
    lane_setup<NL, MD>(P, c, O, rec + (size_t)b * LDim<NL>::REC, K, st, warm, u_init ? u_init + b * ts : nullptr);
    if (st.active && P.mode == QMPC_MODE_REFERENCE) {
      lane_solve_ref<NL, MD>(P, c, O, K, st);
    } else if (st.active)
      while (lane_iteration<NL, MD>(P, c, O, K, st, warm)) {}
    lane_finish<NL, MD>(P, c, O, K, st, forces + (size_t)b * 3 * NL, info ? info + b : nullptr, traj_u ? traj_u + b * ts : nullptr);
  }
  return 0;
}

extern "C" int lane_host_solve(const qmpc_params* p, int batch, const double* rec, double* forces, qmpc_info* info) {
  DevParams P;
  const int st = fill_dev_params(p, &P);
  if (st != QMPC_OK) return st;
  if (p->model == QMPC_MODEL_QUAT8) return solve_all<8>(P, batch, rec, forces, info);
  if (p->model == QMPC_MODEL_QUAT) return solve_all<4>(P, batch, rec, forces, info);
  if (p->model == QMPC_MODEL_CONVEX) return solve_all<4, MD_CONVEX>(P, batch, rec, forces, info);
  return QMPC_BAD_ARGUMENT;
}

// warm-started solve: u_init [batch][N][12] (null: every instance starts cold, but the launch keeps per-row residuals like
// a warm one), traj_u out (may alias u_init)
extern "C" int lane_host_solve_warm(const qmpc_params* p, int batch, const double* rec, const double* u_init, double* forces,
                                    qmpc_info* info, double* traj_u) {
  DevParams P;
  const int st = fill_dev_params(p, &P);
  if (st != QMPC_OK) return st;
  if (p->model == QMPC_MODEL_QUAT) return solve_all<4>(P, batch, rec, forces, info, true, u_init, traj_u);
  return QMPC_BAD_ARGUMENT;
}
