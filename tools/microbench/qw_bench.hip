// Microbenchmark for the "quarter-wave" layout (16 lanes per MPC instance, lane c holds column c of a 12x12
// matrix in 12 registers): issue cost of v_fmac_f64 with a DPP row_newbcast source, of a dense 12x12x12 product
// and of one Gauss-Jordan pivot in that layout, against plain v_fma_f64.  clock64 ticks, one wave per block.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o qw_bench qw_bench.hip && ./qw_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define FMAC_DPP(dst, src, mul, J) "v_fmac_f64_dpp " dst ", " src ", " mul " row_newbcast:" #J " row_mask:0xf bank_mask:0xf\n"

// C[r] += bcast_K(A[r]) * B[K]  for r = 0..11 : one k-step of C = A * B (column-per-lane)
#define KSTEP(K)                                                                                                     \
  asm volatile("s_nop 1\n" FMAC_DPP("%0", "%12", "%24", K) FMAC_DPP("%1", "%13", "%24", K) FMAC_DPP("%2", "%14", "%24", K) \
               FMAC_DPP("%3", "%15", "%24", K) FMAC_DPP("%4", "%16", "%24", K) FMAC_DPP("%5", "%17", "%24", K)             \
               FMAC_DPP("%6", "%18", "%24", K) FMAC_DPP("%7", "%19", "%24", K) FMAC_DPP("%8", "%20", "%24", K)             \
               FMAC_DPP("%9", "%21", "%24", K) FMAC_DPP("%10", "%22", "%24", K) FMAC_DPP("%11", "%23", "%24", K)           \
               : "+v"(C[0]), "+v"(C[1]), "+v"(C[2]), "+v"(C[3]), "+v"(C[4]), "+v"(C[5]), "+v"(C[6]), "+v"(C[7]),          \
                 "+v"(C[8]), "+v"(C[9]), "+v"(C[10]), "+v"(C[11])                                                          \
               : "v"(A[0]), "v"(A[1]), "v"(A[2]), "v"(A[3]), "v"(A[4]), "v"(A[5]), "v"(A[6]), "v"(A[7]), "v"(A[8]),        \
                 "v"(A[9]), "v"(A[10]), "v"(A[11]), "v"(B[K]))

__device__ __forceinline__ double fast_rcp(double x) {
  double r = __builtin_amdgcn_rcp(x);
  r = fma(fma(-x, r, 1.0), r, r);
  r = fma(fma(-x, r, 1.0), r, r);
  return r;
}
template <int J>
__device__ __forceinline__ double row_bcast(double x) {
  int lo = __double2loint(x), hi = __double2hiint(x);
  lo = __builtin_amdgcn_update_dpp(0, lo, 0x150 + J, 0xF, 0xF, true);
  hi = __builtin_amdgcn_update_dpp(0, hi, 0x150 + J, 0xF, 0xF, true);
  return __hiloint2double(hi, lo);
}

// One Gauss-Jordan pivot J on (M | R), both column-per-lane with 12 row registers: every row but J gets
//   M[r] += bcast_J(M[r]) * (-M[J]/piv),  R[r] += bcast_J(M[r]) * (-R[J]/piv)
template <int J>
__device__ __forceinline__ void pivot(double (&M)[12], double (&R)[12]) {
  const double piv = row_bcast<J>(M[J]);
  const double ninv = -fast_rcp(piv);
  const double wm = ninv * M[J], wr = ninv * R[J];
#pragma unroll
  for (int r = 0; r < 12; ++r) {
    if (r == J) continue;
    // R first (reads M[r] of lane J before it is overwritten below)
    asm volatile("s_nop 1\n" "v_fmac_f64_dpp %0, %1, %2 row_newbcast:%4 row_mask:0xf bank_mask:0xf\n"
                 "v_fmac_f64_dpp %1, %1, %3 row_newbcast:%4 row_mask:0xf bank_mask:0xf\n"
                 : "+v"(R[r]), "+v"(M[r]) : "v"(wr), "v"(wm), "n"(J));
  }
}

// MODE 0: 12 independent plain v_fma_f64; 1: 12 independent v_fmac_f64_dpp row_newbcast; 2: dense 12x12x12 product
// (144 fmac_dpp); 3: 12 Gauss-Jordan pivots on a 12 x (12 + 12) system
template <int MODE>
__global__ __launch_bounds__(64) void bench(const double* in, double* out, long long* cyc, int rep) {
  const int lane = threadIdx.x, c = lane & 15;
  double A[12], B[12], C[12];
  for (int r = 0; r < 12; ++r) {
    A[r] = (r == c ? 4.0 : 0.0) + 0.01 * in[(r * 16 + c) & 127] * (1 + ((r * 7 + c * 3) % 5));
    B[r] = 0.001 * (r + c + 1) * in[lane];
    C[r] = 0.0;
  }
  const long long t0 = clock64();
  for (int it = 0; it < rep; ++it) {
    if (MODE == 0) {
#pragma unroll
      for (int u = 0; u < 12; ++u)
#pragma unroll
        for (int r = 0; r < 12; ++r) asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(C[r]) : "v"(A[r]), "v"(B[u]));
    } else if (MODE == 1 || MODE == 2) {
      KSTEP(0); KSTEP(1); KSTEP(2); KSTEP(3); KSTEP(4); KSTEP(5); KSTEP(6); KSTEP(7); KSTEP(8); KSTEP(9); KSTEP(10); KSTEP(11);
    } else {
      double M[12], R[12];
#pragma unroll
      for (int r = 0; r < 12; ++r) { M[r] = A[r] + 1e-9 * C[r]; R[r] = B[r]; }
      pivot<0>(M, R); pivot<1>(M, R); pivot<2>(M, R); pivot<3>(M, R); pivot<4>(M, R); pivot<5>(M, R);
      pivot<6>(M, R); pivot<7>(M, R); pivot<8>(M, R); pivot<9>(M, R); pivot<10>(M, R); pivot<11>(M, R);
#pragma unroll
      for (int r = 0; r < 12; ++r) C[r] += R[r] + M[r];
    }
  }
  const long long t1 = clock64();
  double s = 0.0;
  for (int r = 0; r < 12; ++r) s += C[r];
  out[blockIdx.x * 64 + lane] = s;
  if (lane == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int MODE>
void run(const char* name, int blocks, int per_iter, const double* din, double* dout, long long* dcyc) {
  const int rep = 500;
  bench<MODE><<<blocks, 64>>>(din, dout, dcyc, 5);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0);
  bench<MODE><<<blocks, 64>>>(din, dout, dcyc, rep);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  std::vector<long long> cc(blocks);
  hipMemcpy(cc.data(), dcyc, blocks * sizeof(long long), hipMemcpyDeviceToHost);
  double mean = 0; for (auto x : cc) mean += x; mean /= blocks;
  printf("%-52s blocks %5d: %8.1f ticks / iteration, %6.2f ticks / instruction (%d per iteration), kernel %.3f ms\n", name,
         blocks, mean / rep, mean / rep / per_iter, per_iter, ms);
}

int main() {
  std::vector<double> h(128);
  for (int i = 0; i < 128; ++i) h[i] = 1.0 + 0.01 * (i % 17);
  double *din, *dout; long long* dcyc;
  hipMalloc(&din, 128 * 8); hipMalloc(&dout, 4096 * 64 * 8); hipMalloc(&dcyc, 4096 * 8);
  hipMemcpy(din, h.data(), 128 * 8, hipMemcpyHostToDevice);
  for (int blocks : {256, 1024, 2048}) {
    run<0>("144 plain v_fma_f64 (12 accumulators)", blocks, 144, din, dout, dcyc);
    run<2>("144 v_fmac_f64_dpp row_newbcast = one 12x12x12 product", blocks, 144, din, dout, dcyc);
    run<3>("12 Gauss-Jordan pivots on 12 x (12+12)", blocks, 12, din, dout, dcyc);
  }
  return 0;
}
