// qmpc_wform.hip -- fourth translation unit of libqmpc_hip.so: the wave-per-instance solve kernel with the Newton
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
// The algebra lives in qmpc_wform.h, the body of a solve in qmpc_wform_body.inc (shared with the closed loop's kernels);
// this file is the plain solve kernel and its launcher.  The shared sources are included under another namespace name
// (like qmpc_loop_fused.hip): a code object of its own, so the round-1 kernels keep their code generation to the byte.
#define QMPC_FUSED_TU 1
#define qmpc qmpc_wform_tu
#include "qmpc_kernels.hip"
#include "qmpc_ref.hip"
#include "qmpc_wform.h"

namespace qmpc {

// ---- the solve kernel (QuatMpc's problem, four contact points, everything in LDS, one wave per SIMD) -----------------
// WVAR 3: everything in LDS, one wave per SIMD (small batches); 5: gains, per-point records and per-knot blocks in the
// global workspace (one slice per instance), two waves per SIMD (mid-size batches)
// 6 (round 5): the slack / multiplier / residual arrays in the workspace slice too -- long horizons (N=20: 18 KB of LDS, two
// waves per SIMD where 5 keeps 37 KB and one)
template <bool PROF, int WVAR>
__global__ __launch_bounds__(64, (WVAR == 5 || WVAR == 6) ? 2 : 1) void qmpc_solve_w_kernel(
    DevParams P, const qmpc_input* __restrict__ in_, double* __restrict__ forces, qmpc_info* __restrict__ info,
    double* __restrict__ traj_u, double* __restrict__ traj_x, int batch, long long* __restrict__ prof_out,
    double* __restrict__ gws) {
  extern __shared__ __attribute__((aligned(16))) double sm[];
  const int b = blockIdx.x;
  if (b >= batch) return;
  const int wslot = b;                 // the instance's slice of the workspace
  const int lane = threadIdx.x;
  constexpr int warm_t = 0;            // a plain solve always starts from u_ref (QuatMpc.cpp:253)
  constexpr const double* resume = nullptr;
#include "qmpc_wform_body.inc"
}

// ---- the same solve for EIGHT contact points (QuatModelT<8>: the synthetic biped of BASELINE config 5; round 5) ----------
// The wrench space is 6-dimensional whatever the number of contact points: the backward pass, the closed-loop rollout and
// the gains are those of the four-point kernel; the per-point phases (pre-pass, input recovery + directions, apply) walk
// 8 N (knot, point) pairs, a knot's eight shares of G / r6 are summed over two lane quads.  One wave per SIMD in either
// variant: an instance holds 94 KB (3) / 49 KB (5) of LDS at N=16, so the workspace form too leaves a SIMD at most one wave.
// WVAR 6 (slack arrays in the workspace too: 18 KB of LDS at N=16) is compiled for two waves per SIMD
template <int WVAR>
__global__ __launch_bounds__(64, WVAR == 6 ? 2 : 1) void qmpc_solve8_w_kernel(
    DevParams P, const qmpc_input* __restrict__ in_, double* __restrict__ forces, qmpc_info* __restrict__ info,
    double* __restrict__ traj_u, double* __restrict__ traj_x, int batch, double* __restrict__ gws) {
  extern __shared__ __attribute__((aligned(16))) double sm[];
  const int b = blockIdx.x;
  if (b >= batch) return;
  const int wslot = b;
  const int lane = threadIdx.x;
  constexpr bool PROF = false;
  long long* prof_out = nullptr;
  constexpr int warm_t = 0;
  constexpr const double* resume = nullptr;
#define QMPC_WNL 8
#include "qmpc_wform_body.inc"
#undef QMPC_WNL
}

// ---- ConvexMpc's problem (the sibling controller: Euler-angle model, world-frame forces; ConvexMpc.cpp:81-198) on the same
// body (round 5): WVAR 3 everything in LDS, 5 gains / per-point records / per-knot blocks in the workspace (two waves per SIMD)
template <int WVAR>
__global__ __launch_bounds__(64, (WVAR == 5 || WVAR == 6) ? 2 : 1) void qmpc_solve_cw_kernel(
    DevParams P, const qmpc_input* __restrict__ in_, double* __restrict__ forces, qmpc_info* __restrict__ info,
    double* __restrict__ traj_u, double* __restrict__ traj_x, int batch, double* __restrict__ gws) {
  extern __shared__ __attribute__((aligned(16))) double sm[];
  const int b = blockIdx.x;
  if (b >= batch) return;
  const int wslot = b;
  const int lane = threadIdx.x;
  constexpr bool PROF = false;
  long long* prof_out = nullptr;
  constexpr int warm_t = 0;
  constexpr const double* resume = nullptr;
#define QMPC_WMODEL WM_CONVEX
#include "qmpc_wform_body.inc"
#undef QMPC_WMODEL
}

// The same solve over a LIST of instances (sel[0 .. *sel_count), built on the device): the workgroups walk the list with
// the grid as stride.  The straggler hand-off of large batches (qmpc_hip.hip: launch_solve): the lane-per-instance kernel
// stops after a fixed number of iterations and the few instances it leaves unconverged are solved here, where one
// interior-point iteration takes a tenth of the time.
// One workgroup per SIMD walks the list (grid <= 1024), so the workspace form too is compiled for ONE wave per SIMD here: with
// the 256-register budget of the plain <5> kernel this instantiation spilled 87 VGPRs (352 B of scratch) for nothing.
template <int WVAR>
__global__ __launch_bounds__(64, 1) void qmpc_solve_w_list_kernel(
    DevParams P, const qmpc_input* __restrict__ in_, double* __restrict__ forces, qmpc_info* __restrict__ info,
    double* __restrict__ traj_u, double* __restrict__ traj_x, const int* __restrict__ sel, const int* __restrict__ sel_count,
    double* __restrict__ gws, const double* __restrict__ hstate, int hcap) {
  extern __shared__ __attribute__((aligned(16))) double sm[];
  const int lane = threadIdx.x;
  constexpr bool PROF = false;
  constexpr int warm_t = 0;
  long long* prof_out = nullptr;
  const int wslot = blockIdx.x;
  const int count = sel_count[0];
  // the workgroups DRAW their instances from a cursor (sel_count[1], zeroed with the count before the lane launch) instead of
  // striding the list: what is left of a handed-over solve varies from one iteration to ten, and with a fixed assignment
  // the launch lasted as long as the unluckiest workgroup's share.  Which workgroup continues an instance does not touch its
  // result (wslot only names a scratch slice).
  int* cursor = const_cast<int*>(sel_count) + 1;
  for (;;) {
    int i = 0;
    if (lane == 0) i = atomicAdd(cursor, 1);
    i = __builtin_amdgcn_readfirstlane(i);
    if (i >= count) break;
    const int b = sel[i];
    // the instance's state at the lane kernel's iteration cap (null: none was kept, start from scratch)
    const double* resume = (hstate && i < hcap) ? hstate + (size_t)i * (8 + 84 * (size_t)P.N) : nullptr;
    [&]() {                            // `return` in the body (rejected input) ends this instance only
#include "qmpc_wform_body.inc"
    }();
    __syncthreads();
  }
}

// The reference's own solver mode (AL-iLQR, <= 10 iterations) on the wrench-form algebra (qmpc_wform_ref_body.inc)
// OCC: waves per SIMD the workspace form is compiled for -- 2 for batches beyond one instance per SIMD, 1 (the whole register
// file: no spill) for the batches up to 1024 that only take the workspace form because a long horizon does not fit four
// instances' LDS into a CU (N=20, the reference's own configuration)
template <int WVAR, int OCC = (WVAR == 5 ? 2 : 1)>
__global__ __launch_bounds__(64, OCC) void qmpc_ref_w_kernel(
    DevParams P, const qmpc_input* __restrict__ in_, double* __restrict__ forces, qmpc_info* __restrict__ info,
    double* __restrict__ traj_u, double* __restrict__ traj_x, int batch, double* __restrict__ gws) {
  extern __shared__ __attribute__((aligned(16))) double sm[];
  const int b = blockIdx.x;
  if (b >= batch) return;
  const int wslot = b;
  const int lane = threadIdx.x;
#include "qmpc_wform_ref_body.inc"
}

// ... and of ConvexMpc's problem (its own mode: five AL-iLQR iterations, ConvexMpc.cpp:36-38; refined stage solves at every horizon)
template <int WVAR, int OCC = 1>
__global__ __launch_bounds__(64, OCC) void qmpc_ref_cw_kernel(
    DevParams P, const qmpc_input* __restrict__ in_, double* __restrict__ forces, qmpc_info* __restrict__ info,
    double* __restrict__ traj_u, double* __restrict__ traj_x, int batch, double* __restrict__ gws) {
  extern __shared__ __attribute__((aligned(16))) double sm[];
  const int b = blockIdx.x;
  if (b >= batch) return;
  const int wslot = b;
  const int lane = threadIdx.x;
#define QMPC_WMODEL WM_CONVEX
#include "qmpc_wform_ref_body.inc"
#undef QMPC_WMODEL
}

// ... and of the eight-contact-point model (BASELINE config 5): the per-point phases walk 8 N (knot, point) pairs, the layout keeps
// the direction slots (they hold y0, the trials' costates and trial states in this mode)
template <int WVAR, int OCC = 1>
__global__ __launch_bounds__(64, OCC) void qmpc_ref8_w_kernel(
    DevParams P, const qmpc_input* __restrict__ in_, double* __restrict__ forces, qmpc_info* __restrict__ info,
    double* __restrict__ traj_u, double* __restrict__ traj_x, int batch, double* __restrict__ gws) {
  extern __shared__ __attribute__((aligned(16))) double sm[];
  const int b = blockIdx.x;
  if (b >= batch) return;
  const int wslot = b;
  const int lane = threadIdx.x;
#define QMPC_WNL 8
#include "qmpc_wform_ref_body.inc"
#undef QMPC_WNL
}

}  // namespace qmpc
#undef qmpc

#include <cstring>

using namespace qmpc_wform_tu;

// called from qmpc_hip.hip (declared there); hidden: not part of the C ABI
// kd_global: 0 everything in LDS (WVAR 3), 1 gains / records / blocks in the workspace (5), 2 the slack arrays as well (6)
__attribute__((visibility("hidden"))) size_t qmpc_wform_lds_bytes(int N, int kd_global, int nl, int convex) {
  LayoutW LW;
  return (size_t)(nl == 8 ? make_layout_w<8>(N, &LW, kd_global != 0, kd_global == 2)
                          : make_layout_w<4>(N, &LW, kd_global != 0, kd_global == 2, convex != 0)).total * sizeof(double);
}
// the reference-mode body's layout (full: the direction slots exist for eight points too); kd_global: 0 / 1
__attribute__((visibility("hidden"))) size_t qmpc_wform_ref_lds_bytes(int N, int kd_global, int nl, int convex) {
  LayoutW LW;
  return (size_t)(nl == 8 ? make_layout_w<8>(N, &LW, kd_global != 0, false, false, true)
                          : make_layout_w<4>(N, &LW, kd_global != 0, false, convex != 0, true)).total * sizeof(double);
}
__attribute__((visibility("hidden"))) size_t qmpc_wform_slice_doubles(int N, int nl) { return nl == 8 ? wform_slice<8>(N, true) : wform_slice<4>(N, true); }
__attribute__((visibility("hidden"))) hipError_t qmpc_wform_set_lds(int bytes) {
  const void* k[21] = {reinterpret_cast<const void*>(qmpc_ref8_w_kernel<3, 1>), reinterpret_cast<const void*>(qmpc_ref8_w_kernel<5, 1>),
                      reinterpret_cast<const void*>(qmpc_solve_w_kernel<false, 3>), reinterpret_cast<const void*>(qmpc_solve_w_kernel<true, 3>),
                      reinterpret_cast<const void*>(qmpc_solve_w_kernel<false, 5>), reinterpret_cast<const void*>(qmpc_solve_w_kernel<true, 5>),
                      reinterpret_cast<const void*>(qmpc_solve_w_list_kernel<3>), reinterpret_cast<const void*>(qmpc_solve_w_list_kernel<5>),
                      reinterpret_cast<const void*>(qmpc_ref_w_kernel<3>), reinterpret_cast<const void*>(qmpc_ref_w_kernel<5>),
                      reinterpret_cast<const void*>(qmpc_ref_w_kernel<5, 1>),
                      reinterpret_cast<const void*>(qmpc_solve8_w_kernel<3>), reinterpret_cast<const void*>(qmpc_solve8_w_kernel<5>),
                      reinterpret_cast<const void*>(qmpc_solve_cw_kernel<3>), reinterpret_cast<const void*>(qmpc_solve_cw_kernel<5>),
                      reinterpret_cast<const void*>(qmpc_solve_w_kernel<false, 6>), reinterpret_cast<const void*>(qmpc_solve_cw_kernel<6>),
                      reinterpret_cast<const void*>(qmpc_solve8_w_kernel<6>),
                      reinterpret_cast<const void*>(qmpc_ref_cw_kernel<3, 1>), reinterpret_cast<const void*>(qmpc_ref_cw_kernel<5, 1>),
                      reinterpret_cast<const void*>(qmpc_ref_cw_kernel<5, 2>)};
  for (int i = 0; i < 21; ++i) {
    const hipError_t e = hipFuncSetAttribute(k[i], hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e != hipSuccess) return e;
  }
  return hipSuccess;
}
// var: 3 everything in LDS, 5 gains in the workspace gws
__attribute__((visibility("hidden"))) hipError_t qmpc_wform_launch(int var, int prof, int batch, size_t lds, hipStream_t s,
                                                                   const void* dev_params, size_t dev_params_size,
                                                                   const qmpc_input* in, double* forces, qmpc_info* info,
                                                                   double* traj_u, double* traj_x, long long* prof_out, double* gws) {
  if (dev_params_size != sizeof(DevParams)) return hipErrorInvalidValue;
  DevParams P;
  std::memcpy(&P, dev_params, sizeof P);
#define QMPC_LAUNCH_W(PR, V) \
  hipLaunchKernelGGL((qmpc_solve_w_kernel<PR, V>), dim3((unsigned)batch), dim3(kWave), lds, s, P, in, forces, info, traj_u, traj_x, \
                     batch, prof_out, gws)
  if (var == 6) {
    QMPC_LAUNCH_W(false, 6);
  } else if (var == 5) {
    if (prof) QMPC_LAUNCH_W(true, 5); else QMPC_LAUNCH_W(false, 5);
  } else {
    if (prof) QMPC_LAUNCH_W(true, 3); else QMPC_LAUNCH_W(false, 3);
  }
#undef QMPC_LAUNCH_W
  return hipGetLastError();
}
// eight contact points (records of 64 doubles, 24 forces per instance); var as above
__attribute__((visibility("hidden"))) hipError_t qmpc_wform_launch8(int var, int batch, size_t lds, hipStream_t s, const void* dev_params,
                                                                    size_t dev_params_size, const void* in, double* forces, qmpc_info* info,
                                                                    double* traj_u, double* traj_x, double* gws) {
  if (dev_params_size != sizeof(DevParams)) return hipErrorInvalidValue;
  DevParams P;
  std::memcpy(&P, dev_params, sizeof P);
  const qmpc_input* in_ = static_cast<const qmpc_input*>(in);
  if (var == 6)
    hipLaunchKernelGGL(qmpc_solve8_w_kernel<6>, dim3((unsigned)batch), dim3(kWave), lds, s, P, in_, forces, info, traj_u, traj_x, batch, gws);
  else if (var == 5)
    hipLaunchKernelGGL(qmpc_solve8_w_kernel<5>, dim3((unsigned)batch), dim3(kWave), lds, s, P, in_, forces, info, traj_u, traj_x, batch, gws);
  else
    hipLaunchKernelGGL(qmpc_solve8_w_kernel<3>, dim3((unsigned)batch), dim3(kWave), lds, s, P, in_, forces, info, traj_u, traj_x, batch, gws);
  return hipGetLastError();
}
// ConvexMpc's problem (records of 48 doubles, 12 world-frame forces per instance); var as above
__attribute__((visibility("hidden"))) hipError_t qmpc_wform_launch_convex(int var, int batch, size_t lds, hipStream_t s, const void* dev_params,
                                                                          size_t dev_params_size, const void* in, double* forces,
                                                                          qmpc_info* info, double* traj_u, double* traj_x, double* gws) {
  if (dev_params_size != sizeof(DevParams)) return hipErrorInvalidValue;
  DevParams P;
  std::memcpy(&P, dev_params, sizeof P);
  const qmpc_input* in_ = static_cast<const qmpc_input*>(in);
  if (var == 6)
    hipLaunchKernelGGL(qmpc_solve_cw_kernel<6>, dim3((unsigned)batch), dim3(kWave), lds, s, P, in_, forces, info, traj_u, traj_x, batch, gws);
  else if (var == 5)
    hipLaunchKernelGGL(qmpc_solve_cw_kernel<5>, dim3((unsigned)batch), dim3(kWave), lds, s, P, in_, forces, info, traj_u, traj_x, batch, gws);
  else
    hipLaunchKernelGGL(qmpc_solve_cw_kernel<3>, dim3((unsigned)batch), dim3(kWave), lds, s, P, in_, forces, info, traj_u, traj_x, batch, gws);
  return hipGetLastError();
}
// the instances sel[0 .. *sel_count) (device memory), `grid` workgroups walking the list
__attribute__((visibility("hidden"))) hipError_t qmpc_wform_launch_list(int var, int grid, size_t lds, hipStream_t s, const void* dev_params,
                                                                        size_t dev_params_size, const qmpc_input* in, double* forces,
                                                                        qmpc_info* info, double* traj_u, double* traj_x, const int* sel,
                                                                        const int* sel_count, double* gws, const double* hstate,
                                                                        int hcap) {
  if (dev_params_size != sizeof(DevParams)) return hipErrorInvalidValue;
  DevParams P;
  std::memcpy(&P, dev_params, sizeof P);
  if (var == 5)
    hipLaunchKernelGGL(qmpc_solve_w_list_kernel<5>, dim3((unsigned)grid), dim3(kWave), lds, s, P, in, forces, info, traj_u, traj_x, sel,
                       sel_count, gws, hstate, hcap);
  else
    hipLaunchKernelGGL(qmpc_solve_w_list_kernel<3>, dim3((unsigned)grid), dim3(kWave), lds, s, P, in, forces, info, traj_u, traj_x, sel,
                       sel_count, gws, hstate, hcap);
  return hipGetLastError();
}
// reference mode; var: 3 everything in LDS, 5 gains in the workspace gws
__attribute__((visibility("hidden"))) hipError_t qmpc_wform_ref_launch(int var, int batch, size_t lds, hipStream_t s, const void* dev_params,
                                                                       size_t dev_params_size, const qmpc_input* in, double* forces,
                                                                       qmpc_info* info, double* traj_u, double* traj_x, double* gws) {
  if (dev_params_size != sizeof(DevParams)) return hipErrorInvalidValue;
  DevParams P;
  std::memcpy(&P, dev_params, sizeof P);
  // one instance per SIMD at most -- a small batch, or a horizon whose LDS (> 20 KB: N >= 11) leaves a CU four instances
  // anyway: the whole register file (the 256-register instantiation spills 157 VGPRs and would gain no occupancy for it)
  if (var == 5 && (batch <= 1024 || lds > 20 * 1024))
    hipLaunchKernelGGL((qmpc_ref_w_kernel<5, 1>), dim3((unsigned)batch), dim3(kWave), lds, s, P, in, forces, info, traj_u, traj_x, batch, gws);
  else if (var == 5)
    hipLaunchKernelGGL(qmpc_ref_w_kernel<5>, dim3((unsigned)batch), dim3(kWave), lds, s, P, in, forces, info, traj_u, traj_x, batch, gws);
  else
    hipLaunchKernelGGL(qmpc_ref_w_kernel<3>, dim3((unsigned)batch), dim3(kWave), lds, s, P, in, forces, info, traj_u, traj_x, batch, gws);
  return hipGetLastError();
}
// reference mode of the eight-point model: everything in LDS (3; one instance per CU) or gains / records / blocks in the workspace (5)
__attribute__((visibility("hidden"))) hipError_t qmpc_wform_ref_launch8(int var, int batch, size_t lds, hipStream_t s, const void* dev_params,
                                                                        size_t dev_params_size, const void* in, double* forces,
                                                                        qmpc_info* info, double* traj_u, double* traj_x, double* gws) {
  if (dev_params_size != sizeof(DevParams)) return hipErrorInvalidValue;
  DevParams P;
  std::memcpy(&P, dev_params, sizeof P);
  const qmpc_input* in_ = static_cast<const qmpc_input*>(in);
  if (var == 5)
    hipLaunchKernelGGL((qmpc_ref8_w_kernel<5, 1>), dim3((unsigned)batch), dim3(kWave), lds, s, P, in_, forces, info, traj_u, traj_x, batch, gws);
  else
    hipLaunchKernelGGL((qmpc_ref8_w_kernel<3, 1>), dim3((unsigned)batch), dim3(kWave), lds, s, P, in_, forces, info, traj_u, traj_x, batch, gws);
  return hipGetLastError();
}
// reference mode of ConvexMpc's problem; same variants
__attribute__((visibility("hidden"))) hipError_t qmpc_wform_ref_launch_convex(int var, int batch, size_t lds, hipStream_t s, const void* dev_params,
                                                                              size_t dev_params_size, const void* in, double* forces,
                                                                              qmpc_info* info, double* traj_u, double* traj_x, double* gws) {
  if (dev_params_size != sizeof(DevParams)) return hipErrorInvalidValue;
  DevParams P;
  std::memcpy(&P, dev_params, sizeof P);
  const qmpc_input* in_ = static_cast<const qmpc_input*>(in);
  if (var == 5 && (batch <= 1024 || lds > 20 * 1024))
    hipLaunchKernelGGL((qmpc_ref_cw_kernel<5, 1>), dim3((unsigned)batch), dim3(kWave), lds, s, P, in_, forces, info, traj_u, traj_x, batch, gws);
  else if (var == 5)
    hipLaunchKernelGGL((qmpc_ref_cw_kernel<5, 2>), dim3((unsigned)batch), dim3(kWave), lds, s, P, in_, forces, info, traj_u, traj_x, batch, gws);
  else
    hipLaunchKernelGGL((qmpc_ref_cw_kernel<3, 1>), dim3((unsigned)batch), dim3(kWave), lds, s, P, in_, forces, info, traj_u, traj_x, batch, gws);
  return hipGetLastError();
}
