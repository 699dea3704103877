// leg_block_bench.hip -- the per-contact-point block of the lane kernel (qmpc_lane_core.h: leg_block = barrier weights,
// rotated frame, 3 x 3 L D L') in isolation, registers only: cycles per call with ONE wavefront per SIMD, one block at a
// time and two independent blocks in one basic block (how much of the cost is dependent-chain latency).
// hipcc --offload-arch=gfx950 -O2 -std=c++17 -o leg_block_bench leg_block_bench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include "../../quaternion-mpc_amd/csrc/qmpc_lane_core.h"
using namespace qmpc;
using namespace qmpc::lane;

template <int PAIR>
__global__ __launch_bounds__(64) void bench(DevParams P, int reps, long long* out, double* sink) {
  const int lane = threadIdx.x;
  double rot[9] = {0.9, -0.1, 0.2, 0.1, 0.95, -0.05, -0.2, 0.07, 0.93};
  double cr[18], rc0[6], s0[6];
  cone_rows(P, rot, cr);
  initial_rows(P, cr, 31.0, s0, rc0);
  double sv[2][6], lv[2][6], u[2][3];
  for (int j = 0; j < 2; ++j) {
    for (int i = 0; i < 6; ++i) { sv[j][i] = 1.0 + 0.01 * (lane + i + 7 * j); lv[j][i] = 1e-3 * (1 + i + lane % 5 + j); }
    u[j][0] = 0.1 * lane; u[j][1] = -0.2 + j; u[j][2] = 30.0 + lane;
  }
  const long long t0 = clock64();
  double acc = 0.0;
  for (int r = 0; r < reps; ++r) {
    LegBlk a, b;
    leg_block(P, cr, rc0, 0, sv[0], lv[0], 0u, 0.5, 1e-3, u[0], 31.0, a);
    if (PAIR) leg_block(P, cr, rc0, 3, sv[1], lv[1], 0u, 0.5, 1e-3, u[1], 31.0, b);
    // feed the results back (next call depends on this one, like consecutive knots)
    for (int i = 0; i < 6; ++i) { sv[0][i] += 1e-9 * fabs(a.id0 + a.gq[i % 3]); lv[0][i] += 1e-12 * fabs(a.l21 + a.T[i]); }
    if (PAIR) for (int i = 0; i < 6; ++i) { sv[1][i] += 1e-9 * fabs(b.id1 + b.gq[i % 3]); lv[1][i] += 1e-12 * fabs(b.l10 + b.T[i]); }
    acc += a.id2 + (PAIR ? b.id2 : 0.0);
  }
  const long long t1 = clock64();
  if (lane == 0) out[blockIdx.x] = t1 - t0;
  sink[blockIdx.x * 64 + lane] = acc;
}

int main() {
  qmpc_params p; memset(&p, 0, sizeof p);
  p.horizon = 10; p.h = 0.01f; p.h_ref = 0.01; p.mass = 12.84; p.inertia[0] = 0.02; p.inertia[4] = 0.07; p.inertia[8] = 0.08;
  for (int i = 0; i < 13; ++i) p.q_weights[i] = 1.0;
  for (int i = 0; i < 12; ++i) p.r_weights[i] = 1e-6;
  p.w = 50; p.mu = 0.7; p.fz_max = 100; p.ipm_mu0 = 0.01;
  DevParams P; fill_dev_params(&p, &P);
  long long* d_out; double* d_sink;
  hipMalloc(&d_out, 8 * 1024); hipMalloc(&d_sink, 8 * 64 * 1024);
  const int reps = 2000;
  for (int waves : {1, 256, 1024}) {
    long long h[1024];
    hipLaunchKernelGGL(bench<0>, dim3(waves), dim3(64), 0, 0, P, reps, d_out, d_sink); hipDeviceSynchronize();
    hipMemcpy(h, d_out, 8 * waves, hipMemcpyDeviceToHost);
    double c1 = 0; for (int w = 0; w < waves; ++w) c1 += h[w]; c1 /= waves * (double)reps;
    hipLaunchKernelGGL(bench<1>, dim3(waves), dim3(64), 0, 0, P, reps, d_out, d_sink); hipDeviceSynchronize();
    hipMemcpy(h, d_out, 8 * waves, hipMemcpyDeviceToHost);
    double c2 = 0; for (int w = 0; w < waves; ++w) c2 += h[w]; c2 /= waves * (double)reps;
    printf("waves %4d: one leg_block per iteration %.0f cycles; two independent ones %.0f cycles (%.0f per block)\n", waves, c1, c2, c2 / 2);
  }
  return 0;
}
