// Calibration of rocprofv3's FETCH_SIZE / WRITE_SIZE for the access pattern of the solve kernel (MI355X_MICROARCH.md:
// "other access widths and WRITE_SIZE are uncalibrated: calibrate on a known byte count in your own access
// pattern").  One wave per record, exactly like qmpc_solve_kernel's I/O: a 384-byte record read with ONE
// 8-byte-per-lane load (48 lanes), 96 bytes of forces written by 12 lanes, one 40-byte info record by lane 0.
// Known bytes per launch: read 384 B, write 136 B per record.
//   hipcc --offload-arch=gfx950 -O3 -o fetch_calib fetch_calib.hip
//   rocprofv3 --kernel-trace --pmc FETCH_SIZE -d out_f -- ./fetch_calib ; rocprofv3 --kernel-trace --pmc WRITE_SIZE -d out_w -- ./fetch_calib
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

struct Info { int status, iterations; double a, b, c, d; };

__global__ __launch_bounds__(64) void calib_kernel(const double* __restrict__ in, double* __restrict__ forces,
                                                   Info* __restrict__ info, int batch) {
  const int b = blockIdx.x, lane = threadIdx.x;
  if (b >= batch) return;
  const double v = (lane < 48) ? in[(size_t)b * 48 + lane] : 0.0;
  // a little arithmetic so that the load is not dead; wave sum through LDS
  __shared__ double s[64];
  s[lane] = v;
  __syncthreads();
  double acc = 0.0;
  for (int i = 0; i < 48; ++i) acc += s[i];
  if (lane < 12) forces[(size_t)b * 12 + lane] = acc + lane;
  if (lane == 0) { Info r = {0, 1, acc, 0.0, 0.0, 0.0}; info[b] = r; }
}

int main(int argc, char** argv) {
  const int sizes[2] = {1024, 1 << 20};
  for (int si = 0; si < 2; ++si) {
    const int B = sizes[si];
    double *din, *dout; Info* dinfo;
    hipMalloc(&din, (size_t)B * 384); hipMalloc(&dout, (size_t)B * 96); hipMalloc(&dinfo, (size_t)B * sizeof(Info));
    hipMemset(din, 0, (size_t)B * 384);
    for (int rep = 0; rep < 5; ++rep) {
      calib_kernel<<<B, 64>>>(din, dout, dinfo, B);
      hipDeviceSynchronize();
    }
    printf("calib_kernel B=%d: read %zu bytes, write %zu bytes per launch\n", B, (size_t)B * 384, (size_t)B * 136);
    hipFree(din); hipFree(dout); hipFree(dinfo);
  }
  return 0;
}
