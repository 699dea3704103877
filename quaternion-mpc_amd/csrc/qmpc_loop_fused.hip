// qmpc_loop_fused.hip -- second translation unit of libqmpc_hip.so: the persistent wave-per-robot kernel of the closed
// loop (qmpc_loop_fused_kernel, defined in qmpc_loop.hip) and its launcher.  The kernel re-uses the body of the solve
// kernel (qmpc_solve_body.inc); compiled next to qmpc_solve_kernel it perturbs that kernel's inlining and register
// allocation (contract workload 2 % slower), so it gets a code object of its own.  The shared sources are included
// under another namespace name: their non-template kernels would otherwise be defined twice at link time.
#define QMPC_FUSED_TU 1
#define qmpc qmpc_fused_tu
#include "qmpc_kernels.hip"
#include "qmpc_joint.hip"
#include "qmpc_ref.hip"
#include "qmpc_loop.hip"
#undef qmpc

#include <cstring>

using namespace qmpc_fused_tu;

// called from qmpc_hip.hip (declared there); hidden: not part of the C ABI
template <bool JOINT, bool REF, bool CONVEX = false>
static const void* fused_kernel(int var) {
  if (REF && CONVEX) return var == 5 ? reinterpret_cast<const void*>(qmpc_loop_fused_kernel<5, JOINT, true, true>)
                                     : reinterpret_cast<const void*>(qmpc_loop_fused_kernel<3, JOINT, true, true>);
  if (REF && var == 3) return reinterpret_cast<const void*>(qmpc_loop_fused_kernel<3, JOINT, true, false>);
  if (REF && var == 5) return reinterpret_cast<const void*>(qmpc_loop_fused_kernel<5, JOINT, true, false>);
  if (REF) return var >= 1 ? reinterpret_cast<const void*>(qmpc_loop_fused_kernel<1, JOINT, true, false>)
                           : reinterpret_cast<const void*>(qmpc_loop_fused_kernel<0, JOINT, true, false>);
  if (var == 3) return reinterpret_cast<const void*>(qmpc_loop_fused_kernel<3, JOINT, false, CONVEX>);
  if (var == 5) return reinterpret_cast<const void*>(qmpc_loop_fused_kernel<5, JOINT, false, CONVEX>);
  if (var == 6) return reinterpret_cast<const void*>(qmpc_loop_fused_kernel<6, JOINT, false, CONVEX>);
  return var == 2 ? reinterpret_cast<const void*>(qmpc_loop_fused_kernel<2, JOINT, false, CONVEX>)
                  : (var == 1 ? reinterpret_cast<const void*>(qmpc_loop_fused_kernel<1, JOINT, false, CONVEX>)
                              : reinterpret_cast<const void*>(qmpc_loop_fused_kernel<0, JOINT, false, CONVEX>));
}

__attribute__((visibility("hidden"))) hipError_t qmpc_fused_set_lds(int var, int bytes) {
  const void* k[8] = {fused_kernel<false, false>(var), fused_kernel<true, false>(var), fused_kernel<false, true>(var),
                      fused_kernel<true, true>(var),   fused_kernel<false, false, true>(var), fused_kernel<true, false, true>(var),
                      fused_kernel<false, true, true>(var), fused_kernel<true, true, true>(var)};
  for (int i = 0; i < 8; ++i) {
    const hipError_t e = hipFuncSetAttribute(k[i], hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e != hipSuccess) return e;
  }
  return hipSuccess;
}

// var: 0 everything in LDS, 1 gains in the workspace, 2 gains and slack arrays there (converged mode only),
// 3 / 5 the wrench form with everything in LDS / with its gains in the workspace (QuatMpc's problem, converged mode)
__attribute__((visibility("hidden"))) hipError_t qmpc_fused_launch(int var, int reference_mode, int convex, int batch, size_t lds,
                                                                   hipStream_t s,
                                                                   const void* dev_params, size_t dev_params_size,
                                                                   const qmpc_loop_params* lp, qmpc_loop_state* st,
                                                                   qmpc_input* rec, double* forces, qmpc_info* info,
                                                                   double* trace_f, double* trace_c, int ticks, double* gws,
                                                                   const qmpc_leg_geometry* geom, double* joint_pos,
                                                                   qmpc_joint_command* cmd, qmpc_joint_command* trace_cmd) {
  if (dev_params_size != sizeof(DevParams)) return hipErrorInvalidValue;
  DevParams P;
  std::memcpy(&P, dev_params, sizeof P);
  const qmpc_loop_params LP = *lp;
  FusedJoint JL;
  std::memset(&JL, 0, sizeof JL);
  if (geom) {
    static_assert(sizeof(LegGeom) == sizeof(qmpc_leg_geometry), "kernel argument mirrors the ABI struct");
    std::memcpy(&JL.G, geom, sizeof JL.G);
    JL.joint_pos = joint_pos;
    JL.cmd = cmd;
    JL.trace = trace_cmd;
  }
#define QMPC_LAUNCH_FUSED(V, J, R) \
  hipLaunchKernelGGL((qmpc_loop_fused_kernel<V, J, R>), dim3((unsigned)batch), dim3(kWave), lds, s, P, LP, st, rec, forces, \
                     info, trace_f, trace_c, ticks, batch, gws, JL)
#define QMPC_LAUNCH_FUSED_J(V, R) \
  do { if (geom) QMPC_LAUNCH_FUSED(V, true, R); else QMPC_LAUNCH_FUSED(V, false, R); } while (0)
#define QMPC_LAUNCH_FUSED_CJ(V) \
  do { if (geom) QMPC_LAUNCH_FUSED4(V, true); else QMPC_LAUNCH_FUSED4(V, false); } while (0)
#define QMPC_LAUNCH_FUSED4(V, J) \
  hipLaunchKernelGGL((qmpc_loop_fused_kernel<V, J, false, true>), dim3((unsigned)batch), dim3(kWave), lds, s, P, LP, st, rec, \
                     forces, info, trace_f, trace_c, ticks, batch, gws, JL)
  if (convex && reference_mode) {       // ConvexMpc's own solver mode: the wrench-form reference bodies only
    if (var != 3 && var != 5) return hipErrorInvalidValue;
#define QMPC_LAUNCH_FUSED5(V, J) \
  hipLaunchKernelGGL((qmpc_loop_fused_kernel<V, J, true, true>), dim3((unsigned)batch), dim3(kWave), lds, s, P, LP, st, rec, \
                     forces, info, trace_f, trace_c, ticks, batch, gws, JL)
    if (var == 3) { if (geom) QMPC_LAUNCH_FUSED5(3, true); else QMPC_LAUNCH_FUSED5(3, false); }
    else { if (geom) QMPC_LAUNCH_FUSED5(5, true); else QMPC_LAUNCH_FUSED5(5, false); }
#undef QMPC_LAUNCH_FUSED5
  } else if (convex) {
    if (var == 3) QMPC_LAUNCH_FUSED_CJ(3);
    else if (var == 5) QMPC_LAUNCH_FUSED_CJ(5);
    else if (var == 6) QMPC_LAUNCH_FUSED_CJ(6);
    else if (var == 2) QMPC_LAUNCH_FUSED_CJ(2);
    else if (var == 1) QMPC_LAUNCH_FUSED_CJ(1);
    else QMPC_LAUNCH_FUSED_CJ(0);
  } else if (reference_mode) {
    if (var == 3) QMPC_LAUNCH_FUSED_J(3, true);
    else if (var == 5) QMPC_LAUNCH_FUSED_J(5, true);
    else if (var >= 1) QMPC_LAUNCH_FUSED_J(1, true);
    else QMPC_LAUNCH_FUSED_J(0, true);
  } else {
    if (var == 3) QMPC_LAUNCH_FUSED_J(3, false);
    else if (var == 5) QMPC_LAUNCH_FUSED_J(5, false);
    else if (var == 6) QMPC_LAUNCH_FUSED_J(6, false);
    else if (var == 2) QMPC_LAUNCH_FUSED_J(2, false);
    else if (var == 1) QMPC_LAUNCH_FUSED_J(1, false);
    else QMPC_LAUNCH_FUSED_J(0, false);
  }
#undef QMPC_LAUNCH_FUSED_J
#undef QMPC_LAUNCH_FUSED_CJ
#undef QMPC_LAUNCH_FUSED4
#undef QMPC_LAUNCH_FUSED
  return hipGetLastError();
}

__attribute__((visibility("hidden"))) hipError_t qmpc_warm_set_lds(int bytes) {
  const void* k[12] = {reinterpret_cast<const void*>(qmpc_solve_warm_kernel<3, true>), reinterpret_cast<const void*>(qmpc_solve_warm_kernel<5, true>),
                      reinterpret_cast<const void*>(qmpc_solve_warm_kernel<6, true>),
                      reinterpret_cast<const void*>(qmpc_solve_warm_kernel<6, false>), reinterpret_cast<const void*>(qmpc_solve_warm_kernel<0, false>), reinterpret_cast<const void*>(qmpc_solve_warm_kernel<1, false>),
                      reinterpret_cast<const void*>(qmpc_solve_warm_kernel<2, false>), reinterpret_cast<const void*>(qmpc_solve_warm_kernel<0, true>),
                      reinterpret_cast<const void*>(qmpc_solve_warm_kernel<1, true>),  reinterpret_cast<const void*>(qmpc_solve_warm_kernel<2, true>),
                      reinterpret_cast<const void*>(qmpc_solve_warm_kernel<3, false>), reinterpret_cast<const void*>(qmpc_solve_warm_kernel<5, false>)};
  for (int i = 0; i < 12; ++i) {
    const hipError_t e = hipFuncSetAttribute(k[i], hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e != hipSuccess) return e;
  }
  return hipSuccess;
}

__attribute__((visibility("hidden"))) hipError_t qmpc_warm_launch(int var, int convex, int batch, size_t lds, hipStream_t s,
                                                                  const void* dev_params, size_t dev_params_size,
                                                                  const qmpc_input* in, const double* u_init, double* forces,
                                                                  qmpc_info* info, double* traj_u, double* gws, int check_prev) {
  if (dev_params_size != sizeof(DevParams)) return hipErrorInvalidValue;
  DevParams P;
  std::memcpy(&P, dev_params, sizeof P);
#define QMPC_LAUNCH_WARM(V, C) \
  hipLaunchKernelGGL((qmpc_solve_warm_kernel<V, C>), dim3((unsigned)batch), dim3(kWave), lds, s, P, in, u_init, forces, info, \
                     traj_u, batch, gws, check_prev)
  if (convex) {
    if (var == 3) QMPC_LAUNCH_WARM(3, true);
    else if (var == 5) QMPC_LAUNCH_WARM(5, true);
    else if (var == 6) QMPC_LAUNCH_WARM(6, true);
    else if (var == 2) QMPC_LAUNCH_WARM(2, true);
    else if (var == 1) QMPC_LAUNCH_WARM(1, true);
    else QMPC_LAUNCH_WARM(0, true);
  } else {
    if (var == 3) QMPC_LAUNCH_WARM(3, false);
    else if (var == 5) QMPC_LAUNCH_WARM(5, false);
    else if (var == 6) QMPC_LAUNCH_WARM(6, false);
    else if (var == 2) QMPC_LAUNCH_WARM(2, false);
    else if (var == 1) QMPC_LAUNCH_WARM(1, false);
    else QMPC_LAUNCH_WARM(0, false);
  }
#undef QMPC_LAUNCH_WARM
  return hipGetLastError();
}
