// Does the FP64 VALU of gfx950 skip the 16-lane passes of a wave64 instruction whose lanes are all masked off?
// One wave per SIMD (256 workgroups x 256 threads), eight independent v_fma_f64 chains per lane, live lanes 64 / 32 / 16.
// build: hipcc --offload-arch=gfx950 -O2 -o exec_mask_bench exec_mask_bench.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void __launch_bounds__(256) chains(double* out, int live, int iters, double a, double b) {
  const int lane = threadIdx.x & 63;
  double v[8];
  for (int i = 0; i < 8; ++i) v[i] = 1.0 + 1e-3 * (threadIdx.x + i);
  if (lane < live) {
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = __builtin_fma(v[i], a, b);
      }
    }
  }
  double s = 0.0;
  for (int i = 0; i < 8; ++i) s += v[i];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
int main() {
  double* out;
  hipMalloc(&out, 256 * 256 * sizeof(double));
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 20000;
  for (int live : {64, 32, 16, 64, 32, 16, 48}) {
    chains<<<256, 256>>>(out, live, 100, 0.999999, 1e-6);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    chains<<<256, 256>>>(out, live, iters, 0.999999, 1e-6);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double inst = (double)iters * 128;      // v_fma_f64 per wave
    std::printf("live lanes %2d: %.3f ms, %.2f ns per wave-instruction (one wave per SIMD)\n", live, ms, 1e6 * ms / inst);
  }
  return 0;
}
