// Microbenchmark: where do the cycles of one Gauss-Jordan pivot go?  One wave, fragment layout of the
// solver (lane 16g+c holds rows {g,4+g,8+g} of column c), 12 pivots x REP repetitions, clock64 around it.
// Variants knock out one ingredient at a time (results are then wrong; only the timing matters).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o gj_bench gj_bench.hip && ./gj_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include "../../quaternion-mpc_amd/csrc/qmpc_device.h"
using namespace qmpc;

enum { V_FULL = 0, V_NO_BPERM, V_NO_RCP, V_NO_DPP, V_NO_READLANE, V_ONLY_FMA, V_COUNT };

template <int VAR, int J>
__device__ __forceinline__ void step(double M[3], double Rr[3], int c, int g) {
  constexpr int ej = J >> 2, gj = J & 3;
  const int src = (gj << 4) | c;
  double mrow, rrow;
  if (VAR == V_NO_BPERM || VAR == V_ONLY_FMA) { mrow = M[ej] * 0.5; rrow = Rr[ej] * 0.5; }
  else { mrow = __shfl(M[ej], src); rrow = __shfl(Rr[ej], src); }
  double piv;
  if (VAR == V_NO_READLANE || VAR == V_ONLY_FMA) piv = M[ej] + 3.0;
  else piv = read_lane(M[ej], (gj << 4) | J);
  double ninv;
  if (VAR == V_NO_RCP || VAR == V_ONLY_FMA) ninv = -piv * 1e-3;
  else ninv = -fast_rcp(piv);
#pragma unroll
  for (int e = 0; e < 3; ++e) {
    double col;
    if (VAR == V_NO_DPP || VAR == V_ONLY_FMA) col = M[e] * 0.25;
    else col = row_bcast<J>(M[e]);
    const double f = ((e == ej) && (g == gj)) ? 0.0 : col * ninv;
    M[e] = fma(f, mrow, M[e]);
    Rr[e] = fma(f, rrow, Rr[e]);
  }
}

template <int VAR>
__global__ __launch_bounds__(64) void bench(const double* in, double* out, long long* cyc, int rep) {
  const int lane = threadIdx.x, c = lane & 15, g = lane >> 4;
  double M[3], Rr[3];
  for (int e = 0; e < 3; ++e) { M[e] = in[64 * e + lane]; Rr[e] = in[192 + 64 * e + lane]; }
  const long long t0 = clock64();
  for (int r = 0; r < rep; ++r) {
    step<VAR, 0>(M, Rr, c, g); step<VAR, 1>(M, Rr, c, g); step<VAR, 2>(M, Rr, c, g);
    step<VAR, 3>(M, Rr, c, g); step<VAR, 4>(M, Rr, c, g); step<VAR, 5>(M, Rr, c, g);
    step<VAR, 6>(M, Rr, c, g); step<VAR, 7>(M, Rr, c, g); step<VAR, 8>(M, Rr, c, g);
    step<VAR, 9>(M, Rr, c, g); step<VAR, 10>(M, Rr, c, g); step<VAR, 11>(M, Rr, c, g);
    // keep the data bounded and SPD-ish between repetitions
    for (int e = 0; e < 3; ++e) { M[e] = M[e] * 1e-3 + in[64 * e + lane]; Rr[e] = Rr[e] * 1e-3 + in[192 + 64 * e + lane]; }
  }
  const long long t1 = clock64();
  for (int e = 0; e < 3; ++e) { out[64 * e + lane] = M[e]; out[192 + 64 * e + lane] = Rr[e]; }
  if (lane == 0) cyc[blockIdx.x] = t1 - t0;
}

int main() {
  std::vector<double> h(384);
  for (int e = 0; e < 3; ++e)
    for (int l = 0; l < 64; ++l) {
      const int r = 4 * e + (l >> 4), cc = l & 15;
      h[64 * e + l] = (r == cc) ? 20.0 + r : 0.3 / (1 + abs(r - cc));
      h[192 + 64 * e + l] = 0.1 * (r + 1) - 0.05 * cc;
    }
  double *d_in, *d_out; long long* d_c;
  hipMalloc(&d_in, 384 * 8); hipMalloc(&d_out, 384 * 8); hipMalloc(&d_c, 8 * 4096);
  hipMemcpy(d_in, h.data(), 384 * 8, hipMemcpyHostToDevice);
  const int rep = 2000;
  for (int grid : {1, 256, 1024, 2048}) {
  printf("---- %d workgroups of one wave (%.1f per CU)\n", grid, grid / 256.0);
  const char* names[V_COUNT] = {"full", "no bpermute", "no rcp", "no dpp", "no readlane", "only fma"};
  for (int v = 0; v < V_COUNT; ++v) {
    for (int warm = 0; warm < 2; ++warm) {
      switch (v) {
        case 0: hipLaunchKernelGGL(bench<0>, dim3(grid), dim3(64), 0, 0, d_in, d_out, d_c, rep); break;
        case 1: hipLaunchKernelGGL(bench<1>, dim3(grid), dim3(64), 0, 0, d_in, d_out, d_c, rep); break;
        case 2: hipLaunchKernelGGL(bench<2>, dim3(grid), dim3(64), 0, 0, d_in, d_out, d_c, rep); break;
        case 3: hipLaunchKernelGGL(bench<3>, dim3(grid), dim3(64), 0, 0, d_in, d_out, d_c, rep); break;
        case 4: hipLaunchKernelGGL(bench<4>, dim3(grid), dim3(64), 0, 0, d_in, d_out, d_c, rep); break;
        case 5: hipLaunchKernelGGL(bench<5>, dim3(grid), dim3(64), 0, 0, d_in, d_out, d_c, rep); break;
      }
      hipDeviceSynchronize();
    }
    std::vector<long long> cy(grid); hipMemcpy(cy.data(), d_c, 8 * grid, hipMemcpyDeviceToHost);
    double mean = 0, mx = 0; for (long long x : cy) { mean += x; if (x > mx) mx = x; } mean /= grid;
    printf("%-12s %8.1f cycles per pivot (mean over waves), max %.1f\n", names[v], mean / (12.0 * rep), mx / (12.0 * rep));
  }
  }
  return 0;
}
