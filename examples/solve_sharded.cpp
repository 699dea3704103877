// solve_sharded.cpp -- the multi-GPU path from plain C++, ONE process: one host thread and one handle per visible device,
// instances sharded in contiguous blocks (the instances of legged::QuatMpc::grf_update are independent,
// legged_ctrl/src/mpc/QuatMpc.cpp:194-215), every device solves its block with qmpc_solve_device, and ONE all-gather
// (qmpc_gather = ncclAllGather over xGMI, stream-ordered behind the solve) returns every block to every device.
//
//   g++ -O2 -std=c++17 -D__HIP_PLATFORM_AMD__ -I include -I /opt/rocm/include examples/solve_sharded.cpp -o examples/solve_sharded \
//       quaternion-mpc_amd/csrc/libqmpc_hip.so -Wl,-rpath,'$ORIGIN/../quaternion-mpc_amd/csrc' \
//       -L /opt/rocm/lib -lamdhip64 -lrccl -lpthread
//   examples/solve_sharded [total instances] [devices]
//
// The communicators come from ncclCommInitAll (one per device of this process); a multi-process host passes the ncclComm_t of
// its own rank instead -- bench.py does that through torch.distributed.  Exit code 0: every device holds every instance's
// forces, identical on all devices, every instance converged.  There is no CPU fallback.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

#include "qmpc.h"

#define HIPCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); std::exit(10); } } while (0)

static void stand_record(qmpc_input* r, int b) {
  static const double feet[4][3] = {{0.20, 0.14, -0.30}, {0.20, -0.14, -0.30}, {-0.20, 0.14, -0.30}, {-0.20, -0.14, -0.30}};
  std::memset(r, 0, sizeof *r);
  const double yaw = 0.002 * (b % 500);
  r->quat[0] = std::cos(yaw / 2); r->quat[3] = std::sin(yaw / 2);
  r->rot[0] = std::cos(yaw); r->rot[1] = -std::sin(yaw); r->rot[3] = std::sin(yaw); r->rot[4] = std::cos(yaw); r->rot[8] = 1.0;
  for (int l = 0; l < 4; ++l) {
    for (int a = 0; a < 3; ++a) r->foot_pos_body[3 * l + a] = feet[l][a];
    r->contacts[l] = (b % 3 == 0 || l == 0 || l == 3) ? 1.0 : 0.0;      // stand and trot instances
  }
  r->lin_vel_body[0] = 0.001 * (b % 200);
  std::memcpy(r->quat_d, r->quat, sizeof r->quat);
}

int main(int argc, char** argv) {
  const int total = (argc > 1) ? std::atoi(argv[1]) : 4096;
  int ndev = 0;
  HIPCHK(hipGetDeviceCount(&ndev));
  if (argc > 2 && std::atoi(argv[2]) < ndev) ndev = std::atoi(argv[2]);
  if (ndev < 1) { std::fprintf(stderr, "no HIP device\n"); return 2; }
  const int per = (total + ndev - 1) / ndev;      // equal blocks (the last one padded with copies): one fused all-gather
  const int NV = 12 + 5;                          // forces | qmpc_info (40 bytes = 5 doubles) per instance: one buffer, one collective
  std::vector<int> devs(ndev);
  for (int d = 0; d < ndev; ++d) devs[d] = d;
  std::vector<ncclComm_t> comms(ndev);
  if (ncclCommInitAll(comms.data(), ndev, devs.data()) != ncclSuccess) { std::fprintf(stderr, "ncclCommInitAll failed\n"); return 3; }

  qmpc_params p;
  qmpc_default_params(&p, /*horizon=*/10, QMPC_MODE_CONVERGED);
  std::vector<std::vector<double>> all(ndev, std::vector<double>((size_t)ndev * per * NV));
  std::vector<int> rc(ndev, 0);
  std::vector<float> kms(ndev, 0.f);
  std::vector<std::thread> th;
  for (int d = 0; d < ndev; ++d)
    th.emplace_back([&, d] {
      HIPCHK(hipSetDevice(d));
      qmpc_handle* h = nullptr;
      if (qmpc_create(&p, per, d, &h) != QMPC_OK) { rc[d] = 4; return; }
      std::vector<qmpc_input> in(per);
      for (int i = 0; i < per; ++i) stand_record(&in[i], std::min(d * per + i, total - 1));
      qmpc_input* d_in = nullptr; double *d_loc = nullptr, *d_all = nullptr;
      hipStream_t s;
      HIPCHK(hipStreamCreate(&s));
      HIPCHK(hipMalloc(&d_in, sizeof(qmpc_input) * per));
      HIPCHK(hipMalloc(&d_loc, sizeof(double) * per * NV));
      HIPCHK(hipMalloc(&d_all, sizeof(double) * (size_t)ndev * per * NV));
      HIPCHK(hipMemcpyAsync(d_in, in.data(), sizeof(qmpc_input) * per, hipMemcpyHostToDevice, s));
      // forces [per][12] and status records [per] side by side in the local block
      qmpc_status st = qmpc_solve_device(h, per, d_in, d_loc, reinterpret_cast<qmpc_info*>(d_loc + (size_t)per * 12), s);
      if (st == QMPC_OK) st = qmpc_gather(h, comms[d], d_loc, (int64_t)per * NV, d_all, s);      // stream-ordered behind the solve
      if (st != QMPC_OK) { std::fprintf(stderr, "device %d: %s\n", d, qmpc_status_string(st)); rc[d] = 5; }
      HIPCHK(hipMemcpyAsync(all[d].data(), d_all, sizeof(double) * (size_t)ndev * per * NV, hipMemcpyDeviceToHost, s));
      HIPCHK(hipStreamSynchronize(s));
      qmpc_last_kernel_ms(h, &kms[d]);
      HIPCHK(hipFree(d_in)); HIPCHK(hipFree(d_loc)); HIPCHK(hipFree(d_all));
      HIPCHK(hipStreamDestroy(s));
      qmpc_destroy(h);
    });
  for (auto& t : th) t.join();
  for (int d = 0; d < ndev; ++d) ncclCommDestroy(comms[d]);
  int bad = 0, notok = 0;
  for (int d = 0; d < ndev; ++d) {
    bad += rc[d] != 0;
    bad += std::memcmp(all[d].data(), all[0].data(), sizeof(double) * all[0].size()) != 0;      // every device holds the same gather
  }
  double fz0 = 0.0;
  for (int r = 0; r < ndev; ++r) {
    const double* blk = all[0].data() + (size_t)r * per * NV;
    const qmpc_info* info = reinterpret_cast<const qmpc_info*>(blk + (size_t)per * 12);
    for (int i = 0; i < per; ++i) notok += info[i].status != QMPC_OK;
    if (r == 0) for (int l = 0; l < 4; ++l) fz0 += blk[3 * l + 2];
  }
  std::printf("devices %d, instances %d (%d per device), kernel ms per device:", ndev, ndev * per, per);
  for (int d = 0; d < ndev; ++d) std::printf(" %.3f", kms[d]);
  std::printf("\ninstance 0: sum fz = %.6f N (m g = %.6f); not converged: %d; gathers identical on all devices: %s\n", fz0, p.mass * 9.81, notok,
              bad ? "NO" : "yes");
  return (bad || notok || std::fabs(fz0 - p.mass * 9.81) > 1e-3) ? 1 : 0;
}
