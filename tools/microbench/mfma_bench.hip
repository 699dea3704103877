// Microbenchmark: issue cost and dependent latency of v_mfma_f64_16x16x4_f64 on gfx950, in clock64 ticks
// and in nanoseconds (hipEvent), one wave per block.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o mfma_bench mfma_bench.hip && ./mfma_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef double d4 __attribute__((ext_vector_type(4)));

// The products are issued through inline asm with the accumulators pinned in AGPRs (the compiler otherwise
// shuttles them through VGPRs every iteration).
// MODE 0: 4 independent accumulators (pipe throughput); 1: one accumulator (dependent chain);
// 2: one accumulator, each product followed by a VALU read of the result (full latency);
// 3: independent + 12 independent FP64 FMAs after each product; 4: independent + 12 FP32 FMAs;
// 5: independent + 12 integer adds; 6: the 12 FP64 FMAs alone; 7: independent + 4 LDS reads;
// 8: two accumulators alternating; 9: chains of three products, alternating between two accumulators
#define MFMA(acc) asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+a"(acc) : "v"(a), "v"(b))
template <int MODE>
__global__ __launch_bounds__(64) void bench(const double* in, double* out, long long* cyc, int rep) {
  __shared__ double lds[256];
  const int lane = threadIdx.x;
  double a = in[lane], b = in[64 + lane];
  for (int i = lane; i < 256; i += 64) lds[i] = in[i & 127];
  __syncthreads();
  d4 acc[4];
  for (int i = 0; i < 4; ++i) acc[i] = d4{0.0, 0.0, 0.0, 0.0};
  double v[12];
  float f[12];
  int n[12];
  for (int i = 0; i < 12; ++i) { v[i] = in[lane] + i; f[i] = (float)v[i]; n[i] = lane + i; }
  double ls = 0.0;
  const long long t0 = clock64();
  for (int r = 0; r < rep; ++r) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      if (MODE == 1 || MODE == 2) MFMA(acc[0]);
      else if (MODE == 8) MFMA(acc[i & 1]);
      else if (MODE == 9) MFMA(acc[(i / 3) & 1]);
      else if (MODE != 6) MFMA(acc[i & 3]);
      if (MODE == 2) asm volatile("v_accvgpr_read_b32 %0, a0" : "=v"(n[0]) : "a"(acc[0]));
      if (MODE == 3 || MODE == 6) {
#pragma unroll
        for (int j = 0; j < 12; ++j) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(v[j]) : "v"(a), "v"(b));
      }
      if (MODE == 4) {
#pragma unroll
        for (int j = 0; j < 12; ++j) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(f[j]));
      }
      if (MODE == 5) {
#pragma unroll
        for (int j = 0; j < 12; ++j) asm volatile("v_add_u32 %0, %0, %0" : "+v"(n[j]));
      }
      if (MODE == 7) {
#pragma unroll
        for (int j = 0; j < 4; ++j) { double x; asm volatile("ds_read_b64 %0, %1" : "=v"(x) : "v"(8 * ((lane + 64 * j) & 255))); ls += x; }
      }
    }
  }
  const long long t1 = clock64();
  double s = ls;
  for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  for (int i = 0; i < 12; ++i) s += v[i] + f[i] + n[i];
  out[blockIdx.x * 64 + lane] = s + a;
  if (lane == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int MODE>
void run(const char* name, int blocks, const double* din, double* dout, long long* dcyc) {
  const int rep = 2000;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  bench<MODE><<<blocks, 64>>>(din, dout, dcyc, 10);
  hipEventRecord(e0);
  bench<MODE><<<blocks, 64>>>(din, dout, dcyc, rep);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  std::vector<long long> c(blocks);
  hipMemcpy(c.data(), dcyc, blocks * sizeof(long long), hipMemcpyDeviceToHost);
  double mean = 0; for (auto x : c) mean += x; mean /= blocks;
  printf("%-44s blocks %5d: %7.1f ticks / MFMA, %7.2f ns / MFMA (kernel %.3f ms)\n", name, blocks,
         mean / (8.0 * rep), ms * 1e6 / (8.0 * rep), ms);
}

int main() {
  std::vector<double> h(128, 1.0);
  double *din, *dout; long long* dcyc;
  hipMalloc(&din, 256 * 8); hipMalloc(&dout, 4096 * 64 * 8); hipMalloc(&dcyc, 4096 * 8);
  hipMemcpy(din, h.data(), 128 * 8, hipMemcpyHostToDevice);
  for (int blocks : {1024, 2048}) {
    run<0>("4 independent accumulators", blocks, din, dout, dcyc);
    run<1>("one accumulator (dependent chain)", blocks, din, dout, dcyc);
    run<2>("dependent + accvgpr read of each result", blocks, din, dout, dcyc);
    run<3>("independent + 12 FP64 FMAs per product", blocks, din, dout, dcyc);
    run<4>("independent + 12 FP32 FMAs per product", blocks, din, dout, dcyc);
    run<5>("independent + 12 integer adds per product", blocks, din, dout, dcyc);
    run<6>("12 FP64 FMAs alone (per group)", blocks, din, dout, dcyc);
    run<7>("independent + 4 LDS reads per product", blocks, din, dout, dcyc);
    run<8>("two alternating accumulators", blocks, din, dout, dcyc);
    run<9>("chains of three on two accumulators", blocks, din, dout, dcyc);
  }
  return 0;
}
