// lane_stream.hip -- the memory pattern of the lane-per-instance kernel in isolation: one wavefront per workgroup owns a
// contiguous block of ROWS x 512 bytes and, per "knot", reads R rows (8 bytes per lane each), does F dependent FP64
// FMAs per row value and writes W rows.  Reports cycles per knot for waves = 16 ... 1024.
// hipcc --offload-arch=gfx950 -O2 -o lane_stream lane_stream.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

__constant__ int g_active_lanes = 64;
template <int R, int W, int FMA>
__global__ __launch_bounds__(64) void stream_kernel(double* __restrict__ ws, int rows_per_wave, int knots, int reps, long long* out) {
  double* blk = ws + (size_t)blockIdx.x * rows_per_wave * 64;
  const int lane = threadIdx.x;
  double acc = 0.0;
  if (lane >= g_active_lanes) return;
  const long long t0 = clock64();
  for (int r = 0; r < reps; ++r)
    for (int k = 0; k < knots; ++k) {
      const double* p = blk + (size_t)k * (R + W) * 64 + lane;
      double v[R];
#pragma unroll
      for (int i = 0; i < R; ++i) v[i] = p[i * 64];
      double s = acc;
#pragma unroll
      for (int i = 0; i < R; ++i) s = fma(v[i], 1.0000001, s);
#pragma unroll
      for (int j = 0; j < FMA; ++j) s = fma(s, 0.999999, 1e-9);     // dependent chain: FMA x 4 cycles of compute
      acc = s;
      double* q = blk + ((size_t)k * (R + W) + R) * 64 + lane;
#pragma unroll
      for (int i = 0; i < W; ++i) q[i * 64] = s + i;
    }
  const long long t1 = clock64();
  if (lane == 0) { out[2 * blockIdx.x] = t1 - t0; out[2 * blockIdx.x + 1] = (long long)acc; }
}

template <int R, int W, int FMA>
static void run(const char* name, double* ws, long long* d_out, int knots, int reps) {
  const int rows = knots * (R + W);
  for (int waves : {16, 256, 512, 1024}) {
    hipLaunchKernelGGL((stream_kernel<R, W, FMA>), dim3(waves), dim3(64), 0, 0, ws, rows, knots, reps, d_out);
    hipDeviceSynchronize();
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL((stream_kernel<R, W, FMA>), dim3(waves), dim3(64), 0, 0, ws, rows, knots, reps, d_out);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<long long> h(2 * waves);
    hipMemcpy(h.data(), d_out, sizeof(long long) * 2 * waves, hipMemcpyDeviceToHost);
    double cyc = 0; for (int w = 0; w < waves; ++w) cyc += h[2 * w];
    cyc /= waves;
    const double per_knot = cyc / ((double)knots * reps);
    const double gb = (double)waves * knots * reps * (R + W) * 512.0 / 1e9;
    printf("%-28s waves %4d: %8.0f cycles per knot (ideal compute %d), %.2f ms, %.2f TB/s\n", name, waves, per_knot, 4 * (R + FMA), ms,
           gb / ms / 1e3 * 1e3 / 1e3);
  }
}

int main(int argc, char** argv) {
  const int al = argc > 1 ? atoi(argv[1]) : 64;
  hipMemcpyToSymbol(HIP_SYMBOL(g_active_lanes), &al, sizeof al);
  printf("active lanes %d\n", al);
  const int knots = 20, reps = 20;
  const size_t rows_max = (size_t)knots * 128;
  double* ws; long long* d_out;
  hipMalloc(&ws, sizeof(double) * rows_max * 64 * 1024);
  hipMemset(ws, 0, sizeof(double) * rows_max * 64 * 1024);
  hipMalloc(&d_out, sizeof(long long) * 2 * 1024);
  run<55, 6, 100>("R55 W6 fma100 (pass C head)", ws, d_out, knots, reps);
  run<55, 6, 1500>("R55 W6 fma1500", ws, d_out, knots, reps);
  run<15, 0, 700>("R15 W0 fma700 (one leg)", ws, d_out, knots, reps);
  run<30, 42, 4000>("R30 W42 fma4000 (pass B)", ws, d_out, knots, reps);
  run<8, 0, 0>("R8 latency probe", ws, d_out, knots, reps);
  run<1, 0, 0>("R1 latency probe", ws, d_out, knots, reps);
  return 0;
}
