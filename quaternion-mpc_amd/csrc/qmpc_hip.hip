// qmpc_hip.hip -- C ABI (include/qmpc.h) over the gfx950 kernels.
//
// Drop-in boundary: replaces the ALTRO set-up / Solve() / GetInput(0) block of
// legged::QuatMpc::grf_update (legged_ctrl/src/mpc/QuatMpc.cpp:217-265) -- and, for
// handles created with params.model = QMPC_MODEL_CONVEX, of legged::ConvexMpc::grf_update
// (legged_ctrl/src/mpc/ConvexMpc.cpp:84-186) -- for a batch of independent LeggedState
// records.  No CPU fallback exists here: with no HIP device every entry point returns
// QMPC_NO_DEVICE.
#include "qmpc_kernels.hip"
#include "qmpc_loop.hip"
#include "qmpc_joint.hip"
#include "qmpc_ref.hip"

#include <dlfcn.h>

#include <atomic>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>

using namespace qmpc;

// qmpc_loop_fused.hip (second translation unit): the closed loop's persistent kernel
hipError_t qmpc_fused_set_lds(int var, int bytes);
hipError_t qmpc_fused_launch(int var, int reference_mode, int convex, int batch, size_t lds, hipStream_t s, const void* dev_params, size_t dev_params_size,
                             const qmpc_loop_params* lp, qmpc_loop_state* st, qmpc_input* rec, double* forces,
                             qmpc_info* info, double* trace_f, double* trace_c, int ticks, double* gws,
                             const qmpc_leg_geometry* geom, double* joint_pos, qmpc_joint_command* cmd,
                             qmpc_joint_command* trace_cmd);
hipError_t qmpc_warm_set_lds(int bytes);
hipError_t qmpc_warm_launch(int var, int convex, int batch, size_t lds, hipStream_t s, const void* dev_params, size_t dev_params_size,
                            const qmpc_input* in, const double* u_init, double* forces, qmpc_info* info, double* traj_u,
                            double* gws, int check_prev);

// qmpc_wform.hip (fourth translation unit): the wave-per-instance kernel with the wrench-form elimination (small batches)
size_t qmpc_wform_lds_bytes(int N, int kd_global, int nl, int convex);
size_t qmpc_wform_slice_doubles(int N, int nl);
hipError_t qmpc_wform_ref_launch_convex(int var, int batch, size_t lds, hipStream_t s, const void* dev_params, size_t dev_params_size,
                                        const void* in, double* forces, qmpc_info* info, double* traj_u, double* traj_x, double* gws);
hipError_t qmpc_wform_launch_convex(int var, int batch, size_t lds, hipStream_t s, const void* dev_params, size_t dev_params_size,
                                    const void* in, double* forces, qmpc_info* info, double* traj_u, double* traj_x, double* gws);
hipError_t qmpc_wform_launch8(int var, int batch, size_t lds, hipStream_t s, const void* dev_params, size_t dev_params_size, const void* in,
                              double* forces, qmpc_info* info, double* traj_u, double* traj_x, double* gws);
hipError_t qmpc_wform_set_lds(int bytes);
size_t qmpc_wform_ref_lds_bytes(int N, int kd_global, int nl, int convex);
hipError_t qmpc_wform_ref_launch8(int var, int batch, size_t lds, hipStream_t s, const void* dev_params, size_t dev_params_size, const void* in,
                                  double* forces, qmpc_info* info, double* traj_u, double* traj_x, double* gws);
hipError_t qmpc_wform_launch(int var, int prof, int batch, size_t lds, hipStream_t s, const void* dev_params, size_t dev_params_size,
                             const qmpc_input* in, double* forces, qmpc_info* info, double* traj_u, double* traj_x,
                             long long* prof_out, double* gws);
hipError_t qmpc_wform_ref_launch(int var, int batch, size_t lds, hipStream_t s, const void* dev_params, size_t dev_params_size,
                                 const qmpc_input* in, double* forces, qmpc_info* info, double* traj_u, double* traj_x, double* gws);
hipError_t qmpc_wform_launch_list(int var, int grid, size_t lds, hipStream_t s, const void* dev_params, size_t dev_params_size,
                                  const qmpc_input* in, double* forces, qmpc_info* info, double* traj_u, double* traj_x,
                                  const int* sel, const int* sel_count, double* gws, const double* hstate, int hcap);

// qmpc_lane.hip (third translation unit): the lane-per-instance kernel of large batches
size_t qmpc_lane_ws_bytes(int N, int nl, unsigned slots, int wide);
size_t qmpc_lane_scratch_bytes(int batch);
int qmpc_lane_param_slots();
hipError_t qmpc_lane_upload_params(int pslot, hipStream_t s, const void* dev_params, size_t dev_params_size);
hipError_t qmpc_lane_launch(int nl, int pslot, int batch, hipStream_t s, const void* dev_params, size_t dev_params_size, const void* in,
                            double* forces, qmpc_info* info, double* ws, unsigned slots, int* scratch, int upload_params,
                            const double* u_init, double* traj_u, int check_prev, int order_prev, double* traj_x, int iter_cap,
                            int* hcount, int* hsel, double* hstate, int hcap);
size_t qmpc_lane_handoff_list_bytes(int batch);
size_t qmpc_lane_handoff_record_doubles(int N);

struct qmpc_handle {
  qmpc_params params;
  DevParams dev;
  int device;
  int max_batch;
  hipStream_t stream;
  hipEvent_t ev0, ev1;
  bool timed;
  qmpc_input* d_in;
  double* d_forces;
  qmpc_info* d_info;
  double* d_traj_u;
  double* d_traj_x;
  double* d_A;
  double* d_B;
  size_t lds_bytes;       // LDS-resident gains
  size_t lds_bytes_g;     // gains in the global workspace
  size_t lds_bytes_s;     // gains and slack arrays in the global workspace
  size_t lds_bytes_w;     // the wrench-form kernel (qmpc_wform.hip), everything in LDS
  size_t lds_bytes_ws;    // ... and with its slack arrays there too (WVAR 6; long horizons)
  size_t lds_bytes_wg;    // ... with its gains / per-point records / per-knot blocks in the global workspace
  size_t lds_bytes_wr, lds_bytes_wgr;   // the reference-mode body's layouts (they differ from the two above for eight contact points only)
  int wform;              // 1: batches that keep everything in LDS take the wrench-form kernel (env QMPC_WFORM, default 1)
  int* d_loop_row;        // trace row counter of the closed loop (qmpc_loop_run*)
  double* d_leg;          // staging of the host-buffer leg calls (grown on demand, freed with the handle)
  size_t leg_cap;         // its capacity in doubles
  double* d_loop;         // staging of qmpc_loop_run (states and traces; grown on demand, freed with the handle)
  size_t loop_cap;        // its capacity in doubles
  double* d_gws;          // [max_batch][N*(156+84)] workspace of the global-gains variant
  int variant;            // 0: auto, 1: LDS gains, 2: global gains, 3: + slack arrays, 4: lane per instance (env QMPC_VARIANT)
  double* d_lane_ws;      // structure-of-arrays workspace of the lane-per-instance kernel: [elements][lane_slots], on first use
  unsigned lane_slots;    // resident lanes it is sized for
  int* d_lane_scratch;    // counting sort of the batch on the stance mask: hist | cursor | perm[max_batch]
  int lane_min_batch;     // batches from this size on take the lane-per-instance kernel (env QMPC_LANE_MIN)
  int lane_sort;          // 1: order the batch by stance mask first (env QMPC_LANE_SORT)
  int lane_pslot;         // this handle's slot in the lane kernel's constant-memory parameter table
  bool lane_params_resident;   // set while a stream capture repeats launches with unchanged parameters (closed loop)
  bool lane_loop_cold;         // set during a cold-started qmpc_loop_run*: the loop's own switch-over applies
  int lane_min_loop_cold;
  int lane_min_warm;           // warm-started solves and loop ticks (their lane passes are not pair-split)
  bool lane_order_prev;        // closed loop: d_info holds every robot's previous record -- order the batch by its iteration count too
  int* d_handoff;              // straggler hand-off: count | list of instances the capped lane launch left (on first use)
  double* d_hstate;            // ... and their state records (hstate_cap of them)
  int hstate_cap;
  int lane_ref_min;            // reference-mode batches from this size on take the lane kernel (env QMPC_LANE_REF_MIN)
  int lane_cap;                // straggler hand-off: iteration cap of the lane kernel in cold plain solves (0: off; env QMPC_LANE_CAP)
  int lane_cap_warm;           // ... and in warm-started solves of the closed loop (env QMPC_LANE_CAP_WARM; 0: off)
  int lane_cap_loop;           // ... and in the solves of a cold-started closed loop (in-gait states need fewer iterations; env QMPC_LANE_CAP_LOOP)
  int handoff_failed;          // 1: the hand-off records could not be allocated -- this handle runs the pure lane kernel (qmpc_query)
  int last_kernel;             // QMPC_KERNEL_* of the most recent solve launch (qmpc_query)
  // host-buffer calls (qmpc_solve*, qmpc_solve_async): pinned staging owned by the handle.  Batches below the lane kernel's
  // threshold are solved ZERO-COPY: the wavefront of an instance reads its 384-byte record from pinned host memory with its
  // one coalesced load and writes forces / status straight back, so H2D, kernel and D2H are one launch and one
  // synchronisation (records cross the link while other wavefronts compute)
  unsigned char* h_stage_in;   // [max_batch] records
  unsigned char* h_stage_out;  // [max_batch] (forces | info)
  int zero_copy;               // env QMPC_ZERO_COPY (default 1)
  struct { double* forces; qmpc_info* info; size_t fbytes, ibytes; } pending;   // copy-out owed to a pageable caller (qmpc_wait)
  int stage_in_busy;           // a non-blocking zero-copy launch may still be READING its records from h_stage_in: the staging is
                               // not refilled before the stream has drained (qmpc_solve_async with a pageable `in`, pinned outputs)
};

constexpr unsigned kLaneMaxSlots = 1024 * 64;   // one wavefront per SIMD of the chip
constexpr int kLaneMinLoopCold = 18432;       // ... of the cold-started closed loop (its states need fewer iterations and spread less; measured:
                                              // 16384 robots 3.95 vs 3.91 M robot-ticks/s, 20480: 4.81 vs 3.97 M; warm-started the general threshold holds)
// measured switch-over against the wave-per-instance kernels (QMPC_LANE_MIN overrides).  QuatMpc, round 4 (the wave side is the
// wrench-form kernel with its gains in the workspace): N=10 24576: lane 3.20 vs wave 3.38 M solves/s, 28672: 3.61 vs 3.44;
// N=20 16384: 1.03 vs 0.99, 24576: 1.49 vs 1.00 (long horizons run one wave per SIMD on either side).  ConvexMpc and the
// 8-point model keep the round-1 wave kernels and cross earlier (ConvexMpc N=10 / 20: equal at 16384 / 20480; 8-point
// 16384: 0.78 vs 0.82 M, 20480: 0.97 vs 0.84 M)
// End of round 5: half-filled wavefronts run as lane PAIRS (qmpc_lane.hip: the per-point blocks of the backward and the trial
// pass split across the partner lanes) and a round of the lane kernel costs 15 % less at every size below 32768 -- cold plain
// solves and cold loops cross over earlier (tools/lane_switch_scan.py): N=10 12288 instances wave 3.26 vs lane 2.66 M solves/s,
// 16384: 3.37 vs 3.50, 20480: 3.49 vs 4.32, 24576: 3.54 vs 5.16; N=16 16384: 2.15 vs 2.13, 20480: 2.22 vs 2.59; N=20 16384: 1.62
// vs 1.59, 20480: 1.66 vs 1.95; N=24 12288: 1.16 vs 0.98, 16384: 1.22 vs 1.25.  Warm-started launches have their own switch-over: kLaneMinWarm*.
// (with the warm instantiations of the split passes, warm-started loops, lane vs wave kernels: N=10 16384 robots 7.13 vs 7.70 M
// robot-ticks/s, 20480: 8.59 vs 7.84, 24576: 9.95 vs 7.98, 32768: 12.1 vs 8.2; N=20 16384: 3.68 vs 4.19, 24576: 5.19 vs 4.31)
constexpr int kLaneMinWarm = 18432, kLaneMinWarmLong = 20480, kLaneMinWarmVeryLong = 18432;
// Round 6 (apply pass split across the lane pair, stores outside the per-lane conditions: a round of the lane kernel another
// 7-10 % cheaper): N=10 13312 instances wave 4.14 vs lane 4.39 ms, 14336: 4.35 vs 4.42, 16384: 4.88 vs 4.45; N=16 14336: 6.93 vs 7.22,
// 16384: 7.80 vs 7.37; N=20 14336: 9.15 vs 9.65, 16384: 10.16 vs 9.73; N=24 14336: 11.8 vs 12.2 (tools/lane_switch_scan.py)
// ... and once more after the backward pass of the pair form was split by blocks and took its constants / the knot's state through LDS
// (profiles/r06_lane_pair_lds.txt): N=10 13824: 4.18 vs 4.22 ms, 14336: 4.36 vs 4.22; N=16 14848: 6.94 vs 6.93; N=20 14848: 9.31 vs 9.23;
// N=24 14336: 11.9 vs 11.7
constexpr int kLaneMinBatch = 14336;          // QuatMpc, horizons up to 12
// QuatMpc, longer horizons; round 5 (the wave side is the wrench-form kernel with its slack arrays in the workspace, WVAR 6):
// N=16 20480: wave 2.20 vs lane 2.15 M solves/s, 24576: 2.23 vs 2.52; N=20 20480: 1.64 vs 1.60, 24576: 1.67 vs 1.90;
// N=24 16384: 1.20 vs 1.03, 20480: 1.20 vs 1.25
constexpr int kLaneMinBatchLong = 14848;
constexpr int kLaneMinBatchVeryLong = 14848;  // horizons beyond 22
constexpr int kLaneMinBatchOther = 18432;      // ConvexMpc, short horizons (round-1 wave kernels below it)
// ConvexMpc at its own horizon (N=20; WVAR 6 below the threshold): 20480 instances wave 1.21 vs lane 1.11 M, 24576: 1.22 vs 1.29
constexpr int kLaneMinBatchConvexLong = 22528;
// 8-point model, round 5 (the wave side is the wrench-form kernel, with its slack arrays in the workspace beyond one resident
// round: two waves per SIMD at N=16): 32768 instances wave 1.95 vs lane 1.35 M solves/s, 49152: 1.99 vs 1.85 M; 65536: lane 2.3 M
constexpr int kLaneMinBatch8 = 57344;
// reference mode (AL-iLQR, <= 10 iterations; qmpc_lane_ref_kernel): measured against the wave-per-instance reference kernels
// (tools/refmode_lane_bench.py, N=10): 16384: 1.49 vs 1.74 M solves/s, 32768: 2.70 vs 1.78 M, 65536: 4.59 vs 1.83 M (N=20: 2.53 vs 0.79 M)
// iteration cap of the lane kernel in the solves of a cold-started closed loop, 11 + N/10 (in-gait states: 10.3 iterations on
// average, 17 at most, against 13.6 / 23 of the random states of the plain-solve benchmark): 32768 robots 7.47 -> 7.96 M
// robot-ticks/s, 65536: 11.98 -> 12.76 M (caps 10 .. 13 scanned, tools/loop_bench.py; QMPC_LANE_CAP_LOOP=0 switches it off)
constexpr int kLaneCapLoopBase = 11;
// ... and in its warm-started ticks (5.7 iterations on average, 13-17 at most; the records then carry the rows' initial slack
// residuals): 32768 robots 8.45 -> 9.97 M robot-ticks/s, 65536: 14.5 -> 15.9 M; N=20: 3.44 -> 4.45 M, 5.87 -> 6.86 M (caps 5 .. 10
// scanned; QMPC_LANE_CAP_WARM=0 switches it off)
constexpr int kLaneCapWarm = 8;
// (round 5, against the wrench-form reference kernels: N=10 24576 instances wave 2.98 vs lane 2.77 M solves/s, 32768: 3.04 vs 3.39 M,
// 40960: 3.07 vs 4.11 M; N=16 20480: 1.68 vs 1.52 M, 28672: 1.69 vs 1.99 M; N=20 20480: 1.28 vs 1.24 M, 24576: 1.29 vs 1.45 M)
// (end of round 5: the AL passes keep their feedback gains in double precision -- 78 instead of 42 elements per knot, every
// truncated iterate within 7e-9 N of the oracle's on 0.6 M instances where the packed form left 0.07-1 % beyond 1e-6 N and a few
// line searches per 100 000 decided the other way -- and pay for it in traffic: N=10 32768 instances wave 2.88 vs lane 2.74 M,
// 36864: 2.89 vs 3.08 M, 65536: 2.97 vs 4.61 M; N=16 24576: 1.70 vs 1.45 M, 32768: 1.71 vs 1.83 M; N=20 24576: 1.30 vs 1.21 M,
// 28672: 1.30 vs 1.36 M, 65536: 1.32 vs 2.53 M)
// Round 6: the trial sweeps and the AL backward pass run as lane PAIRS below 32769 instances (a trial of the sweep per partner lane,
// a point of the pair per lane in the per-point blocks): N=10 18432 instances wave 6.42 vs lane 6.65 ms, 20480: 7.12 vs 6.85, 32768: 11.2 vs
// 8.2 (4.0 M solves/s); N=16 14336: 8.63 vs 9.45, 18432: 10.9 vs 10.2; N=20 14336: 11.4 vs 11.8, 16384: 12.9 vs 12.0, 32768: 25.2 vs 15.1
constexpr int kLaneRefMinBatch = 19456;       // N <= 12
constexpr int kLaneRefMinBatchLong = 14848;   // horizons beyond 12 (N=20 14336: 11.4 vs 11.4 ms, 16384: 12.9 vs 11.9 after the pair forms' LDS staging)
// ConvexMpc's own mode (five iterations; tools/refmode_lane_bench.py --model convex): N=20 16384 instances wave 1.70 vs lane 1.68 M solves/s,
// 24576: 1.72 vs 2.38 M, 65536: 1.74 vs 5.61 M; N=10 16384: 3.85 vs 3.37 M, 32768: 4.00 vs 6.00 M, 65536: 4.05 vs 10.5 M
// 8-point model (N=16; tools/refmode_lane_bench.py --model biped8), against its wrench-form reference kernels (qmpc_ref8_w_kernel):
// 16384 instances wave 1.12 vs lane 0.50 M solves/s, 32768: 1.14 vs 0.85 M, 49152: 1.16 vs 1.18 M, 65536: 1.16 vs 1.45 M
// (the round-1 dense reference kernels it ran on before: 0.43 M at 8192, 0.46 M at 65536)
constexpr int kLaneRefMinBatch8 = 49152;
// (ConvexMpc with double-precision gains: N=10 20480 instances wave 4.06 vs lane 3.70 M, 24576: 4.08 vs 4.26 M, 65536: 4.18 vs
// 8.89 M; N=20 16384: 1.75 vs 1.54 M, 20480: 1.75 vs 1.81 M, 65536: 1.79 vs 4.63 M; the 8-point model's lane rate did not move)
constexpr int kLaneRefMinBatchConvex = 22528;
constexpr int kLaneRefMinBatchConvexLong = 19456;

#define HIP_TRY(expr)                                                                      \
  do {                                                                                     \
    hipError_t e_ = (expr);                                                                \
    if (e_ != hipSuccess) {                                                                \
      std::fprintf(stderr, "qmpc: %s failed: %s\n", #expr, hipGetErrorString(e_));         \
      return QMPC_HIP_ERROR;                                                               \
    }                                                                                      \
  } while (0)

extern "C" {

const char* qmpc_version(void) { return "qmpc-hip 0.3 (gfx950, wave-per-instance IPM/Riccati with fp64 MFMA; lane-per-instance kernel for large batches; device-resident closed loop)"; }
int32_t qmpc_sizeof_input(void) { return (int32_t)sizeof(qmpc_input); }
int32_t qmpc_sizeof_params(void) { return (int32_t)sizeof(qmpc_params); }
int32_t qmpc_sizeof_info(void) { return (int32_t)sizeof(qmpc_info); }
int32_t qmpc_sizeof_convex_input(void) { return (int32_t)sizeof(qmpc_convex_input); }
int32_t qmpc_sizeof_input8(void) { return (int32_t)sizeof(qmpc_input8); }
static_assert(sizeof(qmpc_input8) == 8 * Dim<8>::REC && sizeof(qmpc_input) == 8 * Dim<4>::REC, "record sizes");
static int model_nl(int model) { return model == QMPC_MODEL_QUAT8 ? 8 : 4; }
// free list of the lane kernel's parameter slots (qmpc_lane.hip: ql_params[])
static std::mutex g_lane_slot_mutex;
static unsigned long long g_lane_slot_used = 0;
static int lane_slot_acquire() {
  std::lock_guard<std::mutex> lock(g_lane_slot_mutex);
  const int n = qmpc_lane_param_slots();
  for (int i = 0; i < n && i < 64; ++i)
    if (!((g_lane_slot_used >> i) & 1ull)) { g_lane_slot_used |= 1ull << i; return i; }
  return -1;
}
static void lane_slot_release(int slot) {
  if (slot < 0) return;
  std::lock_guard<std::mutex> lock(g_lane_slot_mutex);
  g_lane_slot_used &= ~(1ull << slot);
}
static_assert(sizeof(qmpc_convex_input) == sizeof(qmpc_input), "both records are 48 doubles");

const char* qmpc_status_string(int32_t s) {
  switch (s) {
    case QMPC_OK: return "ok";
    case QMPC_MAX_ITER: return "iteration cap reached";
    case QMPC_NO_CONTACT: return "no stance leg";
    case QMPC_NAN_INPUT: return "non-finite input";
    case QMPC_LINESEARCH_FAIL: return "line search failed";
    case QMPC_NOT_PD: return "Quu not positive definite";
    case QMPC_BAD_ARGUMENT: return "bad argument";
    case QMPC_NO_DEVICE: return "no HIP device (there is no CPU fallback)";
    case QMPC_HIP_ERROR: return "HIP runtime error";
    case QMPC_BATCH_TOO_LARGE: return "batch exceeds the handle's capacity";
    case QMPC_UNSUPPORTED: return "optional dependency not available";
    default: return "unknown status";
  }
}

// legged_ctrl/config/gazebo_go1_quat_mpc.yaml:36-75,115-122; QuatMpc.cpp:21-26,182
void qmpc_default_params(qmpc_params* p, int32_t horizon, int32_t mode) {
  std::memset(p, 0, sizeof *p);
  p->horizon = horizon;
  p->h = (float)(10.0 / 1000.0);
  p->h_ref = 10.0 / 1000.0;
  p->mass = 12.84;
  const double trunk[3] = {0.0168128557, 0.063009565, 0.0716547275};
  for (int a = 0; a < 3; ++a) p->inertia[4 * a] = 1.2 * trunk[a];
  const double q[13] = {2.5, 2.5, 10.0, 0, 0, 0, 0, 0.1, 0.1, 0.1, 0.15, 0.15, 0.15};
  std::memcpy(p->q_weights, q, sizeof q);
  for (int j = 0; j < 12; ++j) p->r_weights[j] = 0.000001;
  p->w = 50.0;
  p->mu = 0.7;
  p->fz_max = 100.0;
  p->mode = mode;
  p->penalty_initial = 1.0;
  p->penalty_max = 1e8;
  p->tol_stationarity = 1e-4;
  p->tol_cost_intermediate = 1e-4;
  p->linesearch_max = 10;
  p->drop_ang_vel = 1;
  if (mode == QMPC_MODE_REFERENCE) {
    p->iterations_max = 10;     // QuatMpc.cpp:22
    p->penalty_scaling = 20.0;  // QuatMpc.cpp:26
    p->tol_feasibility = 1e-4;
  } else {
    p->iterations_max = 120;  // horizon 32 needs up to 86 interior-point iterations on the synthetic states
    p->penalty_scaling = 10.0;
    p->tol_feasibility = 1e-8;
    p->tol_step = 1e-8;
    p->ipm_mu0 = 0.01;
    p->ipm_mu_final = 1e-12;
    p->ipm_sigma = 0.2;
    p->ipm_sigma_fast = 0.01;
    p->ipm_tau = 0.995;
  }
}

// legged_ctrl/config/gazebo_go1_convex_mpc.yaml:35-73; AltroUtils.cpp:239,270-272 (the model's
// hard-coded mass and un-scaled trunk inertia); ConvexMpc.cpp:36-38
void qmpc_default_convex_params(qmpc_params* p, int32_t horizon, int32_t mode) {
  qmpc_default_params(p, horizon, mode);
  p->model = QMPC_MODEL_CONVEX;
  p->h = (float)(5.0 / 1000.0);
  p->h_ref = 5.0 / 1000.0;
  const double trunk[3] = {0.0168128557, 0.063009565, 0.0716547275};
  for (int a = 0; a < 3; ++a) p->inertia[4 * a] = trunk[a];
  const double q[13] = {3.0, 3.0, 3.0, 1.0, 1.0, 20.0, 0.0, 0.0, 3.0, 2.0, 3.0, 2.0, 0.0};
  std::memcpy(p->q_weights, q, sizeof q);
  p->w = 0.0;
  p->mu = 0.6;
  p->fz_max = 200.0;
  p->drop_ang_vel = 0;
  if (mode == QMPC_MODE_REFERENCE) p->iterations_max = 5;   // ConvexMpc.cpp:37
}

// BASELINE.json config 5: SYNTHETIC 30 kg biped, two 0.2 x 0.1 m feet with 4 corner contact points
// each.  The humanoid branch is not in the reference checkout; nothing upstream pins these values.
void qmpc_default_biped8_params(qmpc_params* p, int32_t horizon, int32_t mode) {
  qmpc_default_params(p, horizon, mode);
  p->model = QMPC_MODEL_QUAT8;
  p->mass = 30.0;
  std::memset(p->inertia, 0, sizeof p->inertia);
  p->inertia[0] = 1.2; p->inertia[4] = 1.0; p->inertia[8] = 0.3;
  p->fz_max = 250.0;
}

qmpc_status qmpc_set_params(qmpc_handle* h, const qmpc_params* params) {
  if (!h || !params) return QMPC_BAD_ARGUMENT;
  DevParams d;
  const int st = fill_dev_params(params, &d);
  if (st != QMPC_OK) return (qmpc_status)st;
  if (params->horizon != h->params.horizon) return QMPC_BAD_ARGUMENT;  // buffers are sized by N
  if (params->model != h->params.model) return QMPC_BAD_ARGUMENT;
  h->params = *params;
  h->dev = d;
  return QMPC_OK;
}

// device resources of a handle; on failure the caller destroys the (partially filled) handle
static qmpc_status create_resources(qmpc_handle* h, int N, int nl, int nu) {
  const qmpc_params* params = &h->params;
  const int32_t max_batch = h->max_batch;
  HIP_TRY(hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking));
  HIP_TRY(hipEventCreate(&h->ev0));
  HIP_TRY(hipEventCreate(&h->ev1));
  HIP_TRY(hipMalloc(&h->d_in, sizeof(double) * (32 + 4 * nl) * (size_t)max_batch));
  HIP_TRY(hipMalloc(&h->d_forces, sizeof(double) * nu * (size_t)max_batch));
  HIP_TRY(hipMalloc(&h->d_info, sizeof(qmpc_info) * (size_t)max_batch));
  // the attribute belongs to the kernel, not to the handle: always raise it to the CU's 160 KB so that handles
  // with different horizons can coexist (a smaller value set by a later handle would fail the earlier one's launches)
#define QMPC_SET_LDS(kern, bytes) \
  HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, \
                              (bytes) <= 160 * 1024 ? 160 * 1024 : (int)(bytes)))
  if (params->model == QMPC_MODEL_QUAT8) {
    QMPC_SET_LDS((qmpc_solve_kernel<Quat8Model, false, 1>), h->lds_bytes_g);   // never everything in LDS
    QMPC_SET_LDS((qmpc_solve_kernel<Quat8Model, false, 2>), h->lds_bytes_s);
  } else if (params->model == QMPC_MODEL_CONVEX) {
    if (h->lds_bytes <= 160 * 1024) QMPC_SET_LDS((qmpc_solve_kernel<ConvexModel, false, 0>), h->lds_bytes);
    QMPC_SET_LDS((qmpc_solve_kernel<ConvexModel, false, 1>), h->lds_bytes_g);
    QMPC_SET_LDS((qmpc_solve_kernel<ConvexModel, false, 2>), h->lds_bytes_s);
    QMPC_SET_LDS(qmpc_linearize_kernel<ConvexModel>, h->lds_bytes_g);
  } else {
    if (h->lds_bytes <= 160 * 1024) {
      QMPC_SET_LDS((qmpc_solve_kernel<QuatModel, false, 0>), h->lds_bytes);
      QMPC_SET_LDS((qmpc_solve_kernel<QuatModel, true, 0>), h->lds_bytes);
    }
    QMPC_SET_LDS((qmpc_solve_kernel<QuatModel, false, 1>), h->lds_bytes_g);
    QMPC_SET_LDS((qmpc_solve_kernel<QuatModel, true, 1>), h->lds_bytes_g);
    QMPC_SET_LDS((qmpc_solve_kernel<QuatModel, false, 2>), h->lds_bytes_s);
    QMPC_SET_LDS(qmpc_linearize_kernel<QuatModel>, h->lds_bytes_g);
  }
  if (params->model != QMPC_MODEL_QUAT8)
    for (int v = 0; v < 7; ++v) if (v != 4) HIP_TRY(qmpc_fused_set_lds(v, 160 * 1024));     // the closed loop's persistent kernels
  if (params->model != QMPC_MODEL_QUAT8) HIP_TRY(qmpc_warm_set_lds(160 * 1024));
  HIP_TRY(qmpc_wform_set_lds(160 * 1024));
  if (params->mode == QMPC_MODE_REFERENCE) {
    if (params->model == QMPC_MODEL_QUAT8) {
      QMPC_SET_LDS((qmpc_ref_kernel<Quat8Model, 1>), h->lds_bytes_g);     // never everything in LDS
    } else if (params->model == QMPC_MODEL_CONVEX) {
      if (h->lds_bytes <= 160 * 1024) QMPC_SET_LDS((qmpc_ref_kernel<ConvexModel, 0>), h->lds_bytes);
      QMPC_SET_LDS((qmpc_ref_kernel<ConvexModel, 1>), h->lds_bytes_g);
    } else {
      if (h->lds_bytes <= 160 * 1024) QMPC_SET_LDS((qmpc_ref_kernel<QuatModel, 0>), h->lds_bytes);
      QMPC_SET_LDS((qmpc_ref_kernel<QuatModel, 1>), h->lds_bytes_g);
    }
  }
#undef QMPC_SET_LDS
  HIP_TRY(hipMalloc(&h->d_gws, sizeof(double) * (size_t)N * (13 * nu + 21 * nl + 30 * nl) * (size_t)max_batch));
  return QMPC_OK;
}

qmpc_status qmpc_create(const qmpc_params* params, int32_t max_batch, int32_t device, qmpc_handle** out) {
  if (!out) return QMPC_BAD_ARGUMENT;
  *out = nullptr;
  if (!params || max_batch < 1) return QMPC_BAD_ARGUMENT;
  DevParams d;
  const int st = fill_dev_params(params, &d);
  if (st != QMPC_OK) return (qmpc_status)st;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1 || device < 0 || device >= ndev) {
    std::fprintf(stderr, "qmpc_create: no HIP device %d (found %d); there is no CPU fallback\n", device, ndev);
    return QMPC_NO_DEVICE;
  }
  HIP_TRY(hipSetDevice(device));
  qmpc_handle* h = new (std::nothrow) qmpc_handle();
  if (!h) return QMPC_HIP_ERROR;
  std::memset(h, 0, sizeof *h);
  h->lane_pslot = -1;
  h->params = *params;
  h->dev = d;
  h->device = device;
  h->max_batch = max_batch;
  const int N = params->horizon;
  const int nl = model_nl(params->model), nu = 3 * nl;
  const Layout L = make_layout(N, false, nl), Lg = make_layout(N, true, nl), Ls = make_layout(N, true, nl, true);
  h->lds_bytes = (size_t)L.total * sizeof(double);
  h->lds_bytes_g = (size_t)Lg.total * sizeof(double);
  h->lds_bytes_s = (size_t)Ls.total * sizeof(double);
  if (h->lds_bytes_g > 160 * 1024) { delete h; return QMPC_BAD_ARGUMENT; }

  {
    const char* v = std::getenv("QMPC_VARIANT");
    h->variant = v ? std::atoi(v) : 0;
    const char* wf = std::getenv("QMPC_WFORM");
    h->wform = wf ? std::atoi(wf) : 1;
    h->lds_bytes_w = qmpc_wform_lds_bytes(N, 0, nl, params->model == QMPC_MODEL_CONVEX);
    h->lds_bytes_wg = qmpc_wform_lds_bytes(N, 1, nl, params->model == QMPC_MODEL_CONVEX);
    h->lds_bytes_wr = qmpc_wform_ref_lds_bytes(N, 0, nl, params->model == QMPC_MODEL_CONVEX);
    h->lds_bytes_wgr = qmpc_wform_ref_lds_bytes(N, 1, nl, params->model == QMPC_MODEL_CONVEX);
    h->lds_bytes_ws = qmpc_wform_lds_bytes(N, 2, nl, params->model == QMPC_MODEL_CONVEX);
    const char* lm = std::getenv("QMPC_LANE_MIN");
    h->lane_min_batch = lm ? std::atoi(lm) : (params->model == QMPC_MODEL_QUAT ? (N <= 12 ? kLaneMinBatch : (N <= 22 ? kLaneMinBatchLong : kLaneMinBatchVeryLong))
                                                          : (params->model == QMPC_MODEL_QUAT8 ? kLaneMinBatch8
                                                                                               : (N > 12 ? kLaneMinBatchConvexLong : kLaneMinBatchOther)));
    // (warm-started solves share the plain solve's variants and switch-over; the cold-started loop's in-gait states switch earlier)
    h->lane_min_loop_cold = lm ? h->lane_min_batch : (kLaneMinLoopCold < h->lane_min_batch ? kLaneMinLoopCold : h->lane_min_batch);
    h->lane_min_warm = (lm || params->model != QMPC_MODEL_QUAT) ? h->lane_min_batch
                                                                : (N <= 12 ? kLaneMinWarm : (N <= 22 ? kLaneMinWarmLong : kLaneMinWarmVeryLong));
    // Straggler hand-off (cold plain solves of QuatMpc's problem on the lane kernel): a launch of the lane kernel lasts as
    // long as its slowest instance -- 23 interior-point iterations at N=10 (mean 13.6), 31 at N=20 (mean 14.6) -- while
    // only 8 % / 10 % of the instances are still running after 16 / 17.  The lane kernel stops there, leaves the state of
    // those instances in a record each, and the wave-per-instance kernel, whose iteration takes a tenth of the time,
    // CONTINUES them (launch_solve; qmpc_wform_body.inc `resume`).  The cap is a fixed function of the horizon, so the
    // result of an instance depends neither on timing nor on the batch it is part of.  Measured (caps 14 .. 20 scanned):
    // B=32768 N=10 4.14 -> 5.2 M solves/s, B=65536 N=10 6.8 -> 8.3 M, B=65536 N=20 3.25 -> 3.83 M, B=262144 N=10 9.1 -> 9.8 M.
    const char* lrm = std::getenv("QMPC_LANE_REF_MIN");
    h->lane_ref_min = lrm ? std::atoi(lrm)
                          : (params->model == QMPC_MODEL_CONVEX ? (N <= 12 ? kLaneRefMinBatchConvex : kLaneRefMinBatchConvexLong)
                             : params->model == QMPC_MODEL_QUAT8 ? kLaneRefMinBatch8
                                                                 : (N <= 12 ? kLaneRefMinBatch : kLaneRefMinBatchLong));
    const char* lc = std::getenv("QMPC_LANE_CAP");
    h->lane_cap = lc ? std::atoi(lc) : 15 + N / 10;
    const char* lcl = std::getenv("QMPC_LANE_CAP_LOOP");
    h->lane_cap_loop = lcl ? std::atoi(lcl) : kLaneCapLoopBase + N / 10;
    const char* lcw = std::getenv("QMPC_LANE_CAP_WARM");
    h->lane_cap_warm = lcw ? std::atoi(lcw) : kLaneCapWarm;
    const char* ls = std::getenv("QMPC_LANE_SORT");
    h->lane_sort = ls ? std::atoi(ls) : 1;
    const char* zc = std::getenv("QMPC_ZERO_COPY");
    h->zero_copy = zc ? std::atoi(zc) : 1;
    // the lane kernel reads its parameters from a constant-memory table with one slot per LIVE handle (a slot is rewritten
    // before every launch of its handle, on that launch's stream): slots come from a free list and go back in
    // qmpc_destroy; a handle created while all of them are taken keeps the wave-per-instance kernels
    h->lane_pslot = lane_slot_acquire();
  }
  const qmpc_status rs = create_resources(h, N, nl, nu);
  if (rs != QMPC_OK) { qmpc_destroy(h); return rs; }   // release whatever was created
  *out = h;
  return QMPC_OK;
}

void qmpc_destroy(qmpc_handle* h) {
  if (!h) return;
  (void)hipSetDevice(h->device);
  if (h->d_in) (void)hipFree(h->d_in);
  if (h->d_forces) (void)hipFree(h->d_forces);
  if (h->d_gws) (void)hipFree(h->d_gws);
  if (h->d_lane_ws) (void)hipFree(h->d_lane_ws);
  if (h->d_lane_scratch) (void)hipFree(h->d_lane_scratch);
  if (h->d_handoff) (void)hipFree(h->d_handoff);
  if (h->d_hstate) (void)hipFree(h->d_hstate);
  if (h->h_stage_in) (void)hipHostFree(h->h_stage_in);
  if (h->h_stage_out) (void)hipHostFree(h->h_stage_out);
  if (h->d_leg) (void)hipFree(h->d_leg);
  if (h->d_loop_row) (void)hipFree(h->d_loop_row);
  if (h->d_loop) (void)hipFree(h->d_loop);
  if (h->d_info) (void)hipFree(h->d_info);
  if (h->d_traj_u) (void)hipFree(h->d_traj_u);
  if (h->d_traj_x) (void)hipFree(h->d_traj_x);
  if (h->d_A) (void)hipFree(h->d_A);
  if (h->d_B) (void)hipFree(h->d_B);
  if (h->ev0) (void)hipEventDestroy(h->ev0);
  if (h->ev1) (void)hipEventDestroy(h->ev1);
  if (h->stream) (void)hipStreamDestroy(h->stream);
  lane_slot_release(h->lane_pslot);      // after the frees above (they wait for the device): no launch of this handle reads the slot any more
  delete h;
}

// Variant choice (0: all LDS, 1: gains in the workspace, 2: gains and slack arrays in the workspace).
// With everything in LDS an instance needs 39.6 KB (N=10) / 75 KB (N=20): 4 / 2 instances per CU.  Small
// batches (<= one instance per SIMD) keep everything in LDS (lowest latency); long horizons and large batches
// move the gains (N=10: 19 KB, two waves per SIMD) and, when that is still more than 20 KB, the slack arrays
// (N=20: 36 KB -> 17 KB) to the workspace to raise the number of resident instances.
static int pick_variant(const qmpc_handle* h, int32_t batch) {
  // one instance per SIMD (1024 on the chip) is the break-even: beyond it a second resident wave per SIMD
  // (x1.6 throughput) beats a second round of one-wave instances (measured at B = 2048 / 4096)
  const bool big = batch > 1024;
  // QMPC_VARIANT override (experiments).  Only the instantiations that exist may be named: the 8-point model has no
  // all-LDS kernel, and an all-LDS request that does not fit the CU falls back to the workspace
  const bool no_lds_variant = h->params.model == QMPC_MODEL_QUAT8 || h->lds_bytes > 160 * 1024;
  if (h->variant == 1) return no_lds_variant ? 1 : 0;
  if (h->variant == 2) return 1;
  // variant 2's set-up scratch (one record) aliases X..U..Xc: it needs (N + 1) * 13 >= the record length, or a warm start
  // loaded into U before the set-up would be overwritten
  if (h->variant == 3) return ((h->params.horizon + 1) * 13 >= 32 + 4 * model_nl(h->params.model)) ? 2 : 1;
  if (h->params.model == QMPC_MODEL_QUAT8) return (batch > 768 && h->lds_bytes_g > 40 * 1024) ? 2 : 1;  // 3 per CU in LDS
  if (h->lds_bytes > 40 * 1024) return (big && h->lds_bytes_g > 20 * 1024) ? 2 : 1;   // < 4 instances per CU otherwise
  return big ? 1 : 0;
}
static bool use_global_gains(const qmpc_handle* h, int32_t batch) { return pick_variant(h, batch) >= 1; }

// Large batches of the converged mode go to the lane-per-instance kernel (qmpc_lane.hip): one lane per instance, the
// working set streamed through a structure-of-arrays HBM workspace sized by the RESIDENT lanes (<= 1024 wavefronts).
// It returns forces, info and (on request) the input and state trajectories.
static bool use_lane(const qmpc_handle* h, int32_t batch, const double* d_tu, const double* d_tx, bool warm = false) {
  (void)d_tu; (void)d_tx;
  if (h->params.mode != QMPC_MODE_CONVERGED || h->lane_pslot < 0) return false;
  if (h->variant == 4) return true;
  return h->variant == 0 && batch >= (warm ? h->lane_min_warm : (h->lane_loop_cold ? h->lane_min_loop_cold : h->lane_min_batch));
}
// workspace of the lane kernel, allocated at first use (never inside a stream capture: qmpc_loop_run calls this first)
static qmpc_status ensure_lane_buffers(qmpc_handle* h) {
  const int nl = model_nl(h->params.model);
  if (!h->d_lane_ws) {
    // one workspace block per wavefront; a wavefront may run with 32 of its lanes (qmpc_lane.hip), hence max_batch / 32
    const unsigned want = (unsigned)(((size_t)h->max_batch + 31) / 32) * 64;
    h->lane_slots = want < kLaneMaxSlots ? want : kLaneMaxSlots;
    // both buffers or none: a handle with a workspace but no sort scratch would run unsorted for the rest of its life
    double* ws = nullptr;
    int* sc = nullptr;
    if (hipMalloc(&ws, qmpc_lane_ws_bytes(h->params.horizon, nl, h->lane_slots, h->params.mode == QMPC_MODE_REFERENCE)) != hipSuccess ||
        hipMalloc(&sc, qmpc_lane_scratch_bytes(h->max_batch)) != hipSuccess) {
      std::fprintf(stderr, "qmpc: lane-kernel workspace allocation failed: %s\n", hipGetErrorString(hipGetLastError()));
      if (ws) (void)hipFree(ws);
      return QMPC_HIP_ERROR;
    }
    h->d_lane_ws = ws;
    h->d_lane_scratch = sc;
  }
  return QMPC_OK;
}
// Hand-off buffers, on first use.  One state record (8 + 84 N doubles) per instance of the handle's capacity: 8-10 % of a batch
// is handed over in the measured workloads, but WHICH record an instance gets is decided by an atomic counter, so only room
// for all of them keeps the results independent of timing (445 MB at 65536 x N=10, 1.8 GB at 262144 x N=10; held until
// qmpc_destroy, like the lane kernel's workspace).  If the memory is not there the hand-off is switched off for this handle.
static bool ensure_handoff_buffers(qmpc_handle* h) {
  if (h->d_handoff) return true;
  if (h->handoff_failed) return false;
  h->hstate_cap = h->max_batch;
  const size_t rec_bytes = sizeof(double) * qmpc_lane_handoff_record_doubles(h->params.horizon);
  if (hipMalloc(&h->d_handoff, qmpc_lane_handoff_list_bytes(h->max_batch)) != hipSuccess ||
      hipMalloc(&h->d_hstate, rec_bytes * (size_t)h->hstate_cap) != hipSuccess) {
    // NOT silent: a handle without records runs the pure lane kernel, whose results agree with the hand-off's to ~1e-10 N but
    // not bit for bit -- the caller can see it (stderr, qmpc_query(QMPC_QUERY_HANDOFF_ACTIVE)) and avoid it (qmpc_prepare
    // right after qmpc_create, before other allocations take the memory)
    std::fprintf(stderr, "qmpc: straggler hand-off records (%zu MB) could not be allocated: %s -- this handle keeps the pure lane kernel\n",
                 (rec_bytes * (size_t)h->hstate_cap) >> 20, hipGetErrorString(hipGetLastError()));
    if (h->d_handoff) (void)hipFree(h->d_handoff);
    h->d_handoff = nullptr; h->d_hstate = nullptr;
    h->handoff_failed = 1;
    h->lane_cap = 0;
    h->lane_cap_loop = 0;
    h->lane_cap_warm = 0;
    return false;
  }
  return true;
}
// d_u_init / d_traj_u: previous solutions [batch][N][3 NL] to start from (null: cold) / where to leave this one (null:
// not wanted); they may be the same buffer.  check_prev: d_info still holds the records of the previous solves
static qmpc_status launch_lane(qmpc_handle* h, int32_t batch, const qmpc_input* d_in, double* d_forces, qmpc_info* d_info,
                               hipStream_t s, const double* d_u_init = nullptr, double* d_traj_u = nullptr, int check_prev = 0,
                               double* d_traj_x = nullptr, int iter_cap = 0) {
  const int nl = h->params.model == QMPC_MODEL_CONVEX ? -4 : model_nl(h->params.model);     // -4: ConvexMpc's model (qmpc_lane.hip)
  const qmpc_status es = ensure_lane_buffers(h);
  if (es != QMPC_OK) return es;
  if (iter_cap > 0 && !ensure_handoff_buffers(h)) iter_cap = 0;      // (never inside a stream capture: qmpc_loop_run calls it first)
  HIP_TRY(qmpc_lane_launch(nl, h->lane_pslot, (int)batch, s, &h->dev, sizeof h->dev, d_in, d_forces, d_info, h->d_lane_ws, h->lane_slots,
                           h->lane_sort ? h->d_lane_scratch : nullptr, h->lane_params_resident ? 0 : 1, d_u_init, d_traj_u,
                           check_prev, h->lane_order_prev ? 1 : 0, d_traj_x, iter_cap, iter_cap > 0 ? h->d_handoff : nullptr,
                           iter_cap > 0 ? h->d_handoff + 64 : nullptr, iter_cap > 0 ? h->d_hstate : nullptr, h->hstate_cap));
  return QMPC_OK;
}

// QuatMpc's problem in converged mode takes the wrench-form kernels (qmpc_wform.hip): with everything in LDS (3) where
// the round-1 family would keep everything in LDS (one instance per SIMD at most) and four instances fit a CU with its
// layout, with the gains in the workspace (5) for the mid-size batches below the lane kernel's threshold.
// QMPC_WFORM=0 keeps the round-1 kernels (A/B runs); QMPC_WFORM=3 restricts it to the all-LDS form.
static bool wform6_ok(const qmpc_handle* h) {      // env QMPC_WFORM6=0 switches the variant off (A/B runs)
  static const int on = std::getenv("QMPC_WFORM6") ? std::atoi(std::getenv("QMPC_WFORM6")) : 1;
  return on && h->params.horizon >= 4 && h->lds_bytes_ws <= 80 * 1024;
}
// Every launch form of a model (plain solve, warm-started solve, per-tick and persistent closed loop) includes the same body, so
// ONE rule names the variant for all of them -- they are bit-identical only then.
static int wform_variant(const qmpc_handle* h, int32_t batch) {
  if (!h->wform || h->params.mode != QMPC_MODE_CONVERGED) return 0;
  if (h->params.model == QMPC_MODEL_CONVEX) {
    if (h->variant >= 2) return 0;
    // the same rule as QuatMpc's problem at its horizon (N=20: 75 KB per instance): everything in LDS while every instance
    // finds a CU with room, the workspace form (two waves per SIMD) beyond
    if (h->lds_bytes_w <= 160 * 1024 && batch <= 256 * (int)((160 * 1024) / h->lds_bytes_w)) return 3;
    // ... as long as the batch is ONE round of resident instances (N=20: 37 KB, four per CU = 1024): beyond that the round-1
    // kernel with its slack arrays in the workspace too (17 KB: two waves per SIMD) wins -- measured at N=20, 8192 instances:
    // 0.87 M (round-1) against 0.68 M solves/s
    if (h->lds_bytes_wg <= 80 * 1024 && batch <= 256 * (int)((160 * 1024) / h->lds_bytes_wg)) return 5;
    return (wform6_ok(h) && h->lds_bytes_wg > 20 * 1024) ? 6 : 0;      // (short horizons: the round-1 kernel, 1.60 against 1.61 M at N=10)
  }
  if (h->params.model == QMPC_MODEL_QUAT8) {
    // eight contact points (round 5): 94 KB (everything in LDS) / 49 KB (workspace form) per instance at N=16, one wave per
    // SIMD either way -- everything in LDS while every instance finds a CU with room, the workspace form (three per CU) beyond
    if (h->variant >= 2 && h->variant != 3) return h->lds_bytes_wg <= 160 * 1024 ? 5 : 0;
    if (h->lds_bytes_w <= 160 * 1024 && batch <= 256 * (int)((160 * 1024) / h->lds_bytes_w)) return 3;
    // beyond one resident round of the workspace form: the slack arrays out as well (WVAR 6: 18 KB at N=16, two waves per SIMD)
    if (wform6_ok(h) && h->lds_bytes_wg <= 160 * 1024 && batch > 256 * (int)((160 * 1024) / h->lds_bytes_wg)) return 6;
    return h->lds_bytes_wg <= 160 * 1024 ? 5 : (h->lds_bytes_w <= 160 * 1024 ? 3 : 0);
  }
  if (h->params.model != QMPC_MODEL_QUAT) return 0;
  const int pv = pick_variant(h, batch);
  if (pv == 0) return h->lds_bytes_w <= 40 * 1024 ? 3 : 0;
  // Longer horizons (N=20, the reference's own configuration: 75 KB per instance): everything in LDS as long as every
  // instance of the batch finds a CU with room -- two per CU up to N=21 (512 instances), one per CU beyond (256) -- i.e. for
  // the single robot and small fleets; the workspace form (two waves per SIMD) from there on.  Round 5, tools/latency_b1.py.
  static const int small_lds = std::getenv("QMPC_WFORM_SMALL_LDS") ? std::atoi(std::getenv("QMPC_WFORM_SMALL_LDS")) : 1;
  if (small_lds && h->variant == 0 && h->lds_bytes_w <= 160 * 1024 && batch <= 256 * (int)((160 * 1024) / h->lds_bytes_w)) return 3;
  // Long horizons, mid-size batches (round 5): with 37 KB of LDS (N=20) the workspace form leaves a SIMD ONE wave, and the
  // round-1 kernel with its slack arrays in the workspace (two waves per SIMD) was faster -- N=20: 8192 instances 1.12 M against
  // 0.97 M solves/s.  WVAR 6 moves the wrench form's slack arrays out as well (18 KB); every launch form of QuatMpc's problem
  // (plain, warm-started, the closed loop's two forms) is instantiated on it, so they stay bit-identical.
  if (h->wform != 3 && wform6_ok(h) && h->lds_bytes_wg > 20 * 1024 && batch > 256 * (int)((160 * 1024) / h->lds_bytes_wg)) return 6;
  return (h->wform != 3 && h->lds_bytes_wg <= 80 * 1024) ? 5 : 0;
}
static bool use_wform(const qmpc_handle* h, int32_t batch) { return wform_variant(h, batch) == 3; }

// variant of the converged-mode kernels that share a body (plain solve, warm-started solve, persistent loop kernel):
// pick_variant's 0 / 1 / 2, or 3 = the wrench form where it applies -- the three launch forms must agree, they are
// bit-identical only on the same body
static int body_variant(const qmpc_handle* h, int32_t batch) { const int wv = wform_variant(h, batch); return wv ? wv : pick_variant(h, batch); }
static size_t variant_lds(const qmpc_handle* h, int var) {
  return var == 6 ? h->lds_bytes_ws : var == 5 ? h->lds_bytes_wg : (var == 3 ? h->lds_bytes_w : (var == 2 ? h->lds_bytes_s : (var == 1 ? h->lds_bytes_g : h->lds_bytes)));
}
static double* variant_gws(const qmpc_handle* h, int var) { return (var == 1 || var == 2 || var == 5 || var == 6) ? h->d_gws : nullptr; }

// Straggler hand-off: the wave kernel that CONTINUES what a capped lane launch leaves -- 3 (everything in LDS), 5 (gains in
// the workspace; up to 80 KB of LDS, i.e. every horizon the handle accepts: two workgroups per CU instead of four), 0: no
// hand-off for this handle.  ONE predicate for launch_solve, the closed loop's ticks and the pre-allocation before a capture.
static int handoff_variant(const qmpc_handle* h) {
  if (h->variant != 0 || !h->wform || h->params.model != QMPC_MODEL_QUAT || h->params.mode != QMPC_MODE_CONVERGED ||
      h->handoff_failed)
    return 0;
  return h->lds_bytes_w <= 40 * 1024 ? 3 : (h->lds_bytes_wg <= 80 * 1024 ? 5 : 0);
}
static int handoff_grid(const qmpc_handle* h, int wv) { return variant_lds(h, wv) <= 40 * 1024 ? 1024 : 512; }   // one resident round
static int handoff_cap(const qmpc_handle* h, int kind) {      // kind 1: plain cold solve, 2: cold closed loop, 3: warm closed loop
  const int cap = kind == 3 ? h->lane_cap_warm : (kind == 2 ? h->lane_cap_loop : h->lane_cap);
  return (cap > 0 && cap < h->params.iterations_max && handoff_variant(h)) ? cap : 0;
}

// reference mode of QuatMpc's problem: the wrench-form kernels (3: everything in LDS, one instance per SIMD; 5: gains in the
// workspace), with the rule of the round-1 reference kernels for which of the two; 0: keep the round-1 kernels.
// Horizons up to 12 only: a TRUNCATED iterate does not damp the rounding of its Newton systems, and the 6 x 6 wrench-space
// system (condition ~1e7, growing with the horizon) is solved to ~1e-9 of the step where the rotated 12 x 12 elimination
// keeps every direction to its own scale.  Measured against the round-1 kernels (tools/refmode_bench.py): N=10 all status
// words and iteration counts equal, forces within 4e-8 N; with four trial step lengths per rollout and the costate sweep in
// row-parallel form 0.91 -> 1.95 M solves/s at 1024 instances, 1.24 -> 2.9 M at 8192.  N=20 (first version): status words and
// iteration counts equal but only 65 % of the forces within 1e-6 N (median 6e-7); against the ORACLE the N=20 workload of
// tests/test_gpu_parity.py agrees on 354 of 512 instances (median 4.9e-7 N) where the round-1 kernels agree on 499 (median
// 1.1e-8 N): W' = S6 (I + G S6) carries cond(S6) twice.  N=16: all within 1e-6 N of the round-1 kernels (median 2.5e-8,
// worst 8.7e-7).  QMPC_REF_WFORM_MAXN overrides the limit (experiments).
static int ref_wform_variant(const qmpc_handle* h, int32_t batch) {
  if (!h->wform || h->params.mode != QMPC_MODE_REFERENCE) return 0;
  static const int maxn = std::getenv("QMPC_REF_WFORM_MAXN") ? std::atoi(std::getenv("QMPC_REF_WFORM_MAXN")) : QMPC_MAX_HORIZON;
  if (h->params.horizon > maxn || h->params.horizon < 2) return 0;      // (one knot: the input weights would not fit behind the trial states)
  if (h->params.model == QMPC_MODEL_QUAT8) {      // eight points (round 5): one wave per SIMD in either form; everything in LDS while
    if (h->variant == 0 && h->lds_bytes_wr <= 160 * 1024 && batch <= 256 * (int)((160 * 1024) / h->lds_bytes_wr)) return 3;   // every instance finds a CU
    return h->lds_bytes_wgr <= 160 * 1024 ? 5 : 0;
  }
  if (h->variant < 2) {
    if (batch <= 1024 && h->lds_bytes_w <= 40 * 1024) return 3;
    // longer horizons: everything in LDS while every instance finds a CU with room (wform_variant's rule)
    if (h->variant == 0 && h->lds_bytes_w <= 160 * 1024 && batch <= 256 * (int)((160 * 1024) / h->lds_bytes_w)) return 3;
  }
  return h->lds_bytes_wg <= 80 * 1024 ? 5 : 0;
}

// Reference-mode batches of Monte-Carlo scale take the AL variant of the lane passes (qmpc_lane_core.h: lane_solve_ref;
// qmpc_lane.hip: qmpc_lane_ref_kernel): QuatMpc's problem (four or eight contact points) and ConvexMpc's (its own mode: five iterations)
// In a closed loop the in-gait states need 1.8 iterations on average (plain solves of the benchmark states: 8.6) and the
// lane kernel's fixed costs weigh more: N=10 32768 robots wave kernels 12.7 vs lane 9.2 M robot-ticks/s, 65536: 13.2 vs 16.4 M
// (49152: 12.7 vs 10.1 M; ConvexMpc: 12.9 vs 8.9 M at 32768, 13.6 vs 16.6 M at 65536; N=20 65536 robots: 4.28 vs 4.99 M,
// ConvexMpc 4.73 vs 8.37 M; tools/loop_bench.py --mode 1)
constexpr int kLaneRefMinLoop = 61440;
static bool ref_lane_batch(const qmpc_handle* h, int32_t batch, bool loop = false) {
  if (h->params.mode != QMPC_MODE_REFERENCE || h->lane_pslot < 0) return false;
  if (h->variant == 4) return true;
  const int min_batch = (loop && h->lane_ref_min < kLaneRefMinLoop && !std::getenv("QMPC_LANE_REF_MIN")) ? kLaneRefMinLoop : h->lane_ref_min;
  return h->variant == 0 && batch >= min_batch;
}

static qmpc_status launch_solve(qmpc_handle* h, int32_t batch, const qmpc_input* d_in, double* d_forces,
                                qmpc_info* d_info, double* d_tu, double* d_tx, hipStream_t s, bool timed = true,
                                int handoff = 1) {      // 0: no straggler hand-off, 1: plain solve (lane_cap), 2: closed loop (lane_cap_loop)
  if (batch > h->max_batch) return QMPC_BATCH_TOO_LARGE;   // the gains workspace is sized by max_batch
  if (timed) HIP_TRY(hipEventRecord(h->ev0, s));
  if (h->params.mode == QMPC_MODE_REFERENCE) {     // the reference's own AL-iLQR mode (qmpc_ref.hip)
    const bool ws = batch > 1024 || h->lds_bytes > 40 * 1024 || h->variant >= 2 || h->params.model == QMPC_MODEL_QUAT8;
    // Monte-Carlo scale (plain solves of QuatMpc's problem): one lane per instance, the AL variant of the lane passes
    // (qmpc_lane_core.h: lane_solve_ref; qmpc_lane.hip: qmpc_lane_ref_kernel)
    if (handoff != 0 && ref_lane_batch(h, batch, handoff == 2)) {      // plain solves and the ticks of a closed loop (its workspace and parameters are set up before the capture)
      const qmpc_status ls = launch_lane(h, batch, d_in, d_forces, d_info, s, nullptr, d_tu, 0, d_tx, 0);
      if (ls != QMPC_OK) return ls;
      h->last_kernel = QMPC_KERNEL_LANE;
      if (timed) {
        HIP_TRY(hipEventRecord(h->ev1, s));
        h->timed = true;
      }
      return QMPC_OK;
    }
    if (const int wv = ref_wform_variant(h, batch)) {      // QuatMpc's problem: on the wrench-form algebra (qmpc_wform_ref_body.inc)
      h->last_kernel = wv >= 5 ? QMPC_KERNEL_WFORM_WS : QMPC_KERNEL_WFORM_LDS;
      if (h->params.model == QMPC_MODEL_QUAT8)
        HIP_TRY(qmpc_wform_ref_launch8(wv, (int)batch, wv == 5 ? h->lds_bytes_wgr : h->lds_bytes_wr, s, &h->dev, sizeof h->dev, d_in, d_forces,
                                       d_info, d_tu, d_tx, variant_gws(h, wv)));
      else if (h->params.model == QMPC_MODEL_CONVEX)
        HIP_TRY(qmpc_wform_ref_launch_convex(wv, (int)batch, variant_lds(h, wv), s, &h->dev, sizeof h->dev, d_in, d_forces, d_info, d_tu,
                                             d_tx, variant_gws(h, wv)));
      else
      HIP_TRY(qmpc_wform_ref_launch(wv, (int)batch, variant_lds(h, wv), s, &h->dev, sizeof h->dev, d_in, d_forces, d_info, d_tu, d_tx,
                                    variant_gws(h, wv)));
      if (timed) {
        HIP_TRY(hipEventRecord(h->ev1, s));
        h->timed = true;
      }
      return QMPC_OK;
    }
    const size_t lds_r = ws ? h->lds_bytes_g : h->lds_bytes;
    double* gws_r = ws ? h->d_gws : nullptr;
    h->last_kernel = ws ? QMPC_KERNEL_DENSE_WS : QMPC_KERNEL_DENSE_LDS;
#define QMPC_LAUNCH_REF(kern) \
  hipLaunchKernelGGL(kern, dim3((unsigned)batch), dim3(kWave), lds_r, s, h->dev, d_in, d_forces, d_info, d_tu, d_tx, \
                     (int)batch, gws_r)
    if (h->params.model == QMPC_MODEL_QUAT8) {
      QMPC_LAUNCH_REF((qmpc_ref_kernel<Quat8Model, 1>));
    } else if (h->params.model == QMPC_MODEL_CONVEX) {
      if (ws) QMPC_LAUNCH_REF((qmpc_ref_kernel<ConvexModel, 1>));
      else QMPC_LAUNCH_REF((qmpc_ref_kernel<ConvexModel, 0>));
    } else {
      if (ws) QMPC_LAUNCH_REF((qmpc_ref_kernel<QuatModel, 1>));
      else QMPC_LAUNCH_REF((qmpc_ref_kernel<QuatModel, 0>));
    }
#undef QMPC_LAUNCH_REF
    HIP_TRY(hipGetLastError());
    if (timed) {
      HIP_TRY(hipEventRecord(h->ev1, s));
      h->timed = true;
    }
    return QMPC_OK;
  }
  if (use_lane(h, batch, d_tu, d_tx)) {
    // straggler hand-off (see qmpc_create): only where the library chose the lane kernel by itself (QMPC_VARIANT=4 forces
    // the pure lane kernel) and there at every batch size (a shard of a batch gives the bits of the whole batch), with status
    // records to select from
    const int cap = (handoff && d_info) ? handoff_cap(h, handoff) : 0;
    const int wv = cap ? handoff_variant(h) : 0;
    const qmpc_status ls = launch_lane(h, batch, d_in, d_forces, d_info, s, nullptr, d_tu, 0, d_tx, wv ? cap : 0);
    if (ls != QMPC_OK) return ls;
    h->last_kernel = (wv && h->d_handoff) ? QMPC_KERNEL_LANE_HANDOFF : QMPC_KERNEL_LANE;
    if (wv && h->d_handoff) {     // one workgroup per SIMD walks the list the lane kernel left (8-10 % of the batch in the measured workloads)
      // QMPC_HANDOFF_RESTART=1 (experiments, tests): the wave kernel ignores the state records and solves the list from scratch
      const char* hr = std::getenv("QMPC_HANDOFF_RESTART");
      const bool restart = hr && std::atoi(hr) != 0;
      HIP_TRY(qmpc_wform_launch_list(wv, handoff_grid(h, wv), variant_lds(h, wv), s, &h->dev, sizeof h->dev, d_in, d_forces, d_info, d_tu, d_tx,
                                     h->d_handoff + 64, h->d_handoff, variant_gws(h, wv), restart ? nullptr : h->d_hstate, h->hstate_cap));
    }
    if (timed) {
      HIP_TRY(hipEventRecord(h->ev1, s));
      h->timed = true;
    }
    return QMPC_OK;
  }
  if (const int wv = wform_variant(h, batch)) {
    h->last_kernel = wv >= 5 ? QMPC_KERNEL_WFORM_WS : QMPC_KERNEL_WFORM_LDS;
    if (h->params.model == QMPC_MODEL_CONVEX) {
      HIP_TRY(qmpc_wform_launch_convex(wv, (int)batch, variant_lds(h, wv), s, &h->dev, sizeof h->dev, d_in, d_forces,
                                       d_info, d_tu, d_tx, variant_gws(h, wv)));
    } else if (h->params.model == QMPC_MODEL_QUAT8) {
      HIP_TRY(qmpc_wform_launch8(wv, (int)batch, variant_lds(h, wv), s, &h->dev, sizeof h->dev, d_in, d_forces, d_info,
                                 d_tu, d_tx, variant_gws(h, wv)));
    } else
    HIP_TRY(qmpc_wform_launch(wv, 0, (int)batch, variant_lds(h, wv), s, &h->dev, sizeof h->dev, d_in, d_forces, d_info,
                              d_tu, d_tx, nullptr, variant_gws(h, wv)));
    if (timed) {
      HIP_TRY(hipEventRecord(h->ev1, s));
      h->timed = true;
    }
    return QMPC_OK;
  }
  const int var = pick_variant(h, batch);
  const size_t lds = var == 2 ? h->lds_bytes_s : (var == 1 ? h->lds_bytes_g : h->lds_bytes);
  double* gws = var >= 1 ? h->d_gws : nullptr;
  h->last_kernel = var >= 1 ? QMPC_KERNEL_DENSE_WS : QMPC_KERNEL_DENSE_LDS;
#define QMPC_LAUNCH(kern) \
  hipLaunchKernelGGL(kern, dim3((unsigned)batch), dim3(kWave), lds, s, h->dev, d_in, d_forces, d_info, d_tu, d_tx, \
                     (int)batch, (long long*)nullptr, gws)
  if (h->params.model == QMPC_MODEL_QUAT8) {
    if (var == 2) QMPC_LAUNCH((qmpc_solve_kernel<Quat8Model, false, 2>));
    else QMPC_LAUNCH((qmpc_solve_kernel<Quat8Model, false, 1>));
  } else if (h->params.model == QMPC_MODEL_CONVEX) {
    if (var == 2) QMPC_LAUNCH((qmpc_solve_kernel<ConvexModel, false, 2>));
    else if (var == 1) QMPC_LAUNCH((qmpc_solve_kernel<ConvexModel, false, 1>));
    else QMPC_LAUNCH((qmpc_solve_kernel<ConvexModel, false, 0>));
  } else {
    if (var == 2) QMPC_LAUNCH((qmpc_solve_kernel<QuatModel, false, 2>));
    else if (var == 1) QMPC_LAUNCH((qmpc_solve_kernel<QuatModel, false, 1>));
    else QMPC_LAUNCH((qmpc_solve_kernel<QuatModel, false, 0>));
  }
#undef QMPC_LAUNCH
  HIP_TRY(hipGetLastError());
  if (timed) {
    HIP_TRY(hipEventRecord(h->ev1, s));
    h->timed = true;
  }
  return QMPC_OK;
}

qmpc_status qmpc_solve_device(qmpc_handle* h, int32_t batch, const qmpc_input* d_in, double* d_forces_body,
                              qmpc_info* d_info, void* stream) {
  if (!h || batch < 0 || (batch > 0 && (!d_in || !d_forces_body))) return QMPC_BAD_ARGUMENT;
  if (h->params.model != QMPC_MODEL_QUAT) return QMPC_BAD_ARGUMENT;
  if (batch == 0) return QMPC_OK;
  HIP_TRY(hipSetDevice(h->device));
  hipStream_t s = stream ? (hipStream_t)stream : h->stream;
  return launch_solve(h, batch, d_in, d_forces_body, d_info, nullptr, nullptr, s);
}

qmpc_status qmpc_solve8_device(qmpc_handle* h, int32_t batch, const qmpc_input8* d_in, double* d_forces_body,
                               qmpc_info* d_info, void* stream) {
  if (!h || batch < 0 || (batch > 0 && (!d_in || !d_forces_body))) return QMPC_BAD_ARGUMENT;
  if (h->params.model != QMPC_MODEL_QUAT8) return QMPC_BAD_ARGUMENT;
  if (batch == 0) return QMPC_OK;
  HIP_TRY(hipSetDevice(h->device));
  hipStream_t s = stream ? (hipStream_t)stream : h->stream;
  return launch_solve(h, batch, reinterpret_cast<const qmpc_input*>(d_in), d_forces_body, d_info, nullptr, nullptr, s);
}

qmpc_status qmpc_convex_solve_device(qmpc_handle* h, int32_t batch, const qmpc_convex_input* d_in,
                                     double* d_forces_world, qmpc_info* d_info, void* stream) {
  if (!h || batch < 0 || (batch > 0 && (!d_in || !d_forces_world))) return QMPC_BAD_ARGUMENT;
  if (h->params.model != QMPC_MODEL_CONVEX) return QMPC_BAD_ARGUMENT;
  if (batch == 0) return QMPC_OK;
  HIP_TRY(hipSetDevice(h->device));
  hipStream_t s = stream ? (hipStream_t)stream : h->stream;
  return launch_solve(h, batch, reinterpret_cast<const qmpc_input*>(d_in), d_forces_world, d_info, nullptr, nullptr, s);
}

static void finish_pending(qmpc_handle* h);
qmpc_status qmpc_wait(qmpc_handle* h) {
  if (!h) return QMPC_BAD_ARGUMENT;
  HIP_TRY(hipSetDevice(h->device));
  if (h->timed) HIP_TRY(hipEventSynchronize(h->ev1));
  HIP_TRY(hipStreamSynchronize(h->stream));
  finish_pending(h);          // zero-copy host call into pageable buffers: the copy-out it still owes
  h->stage_in_busy = 0;
  return QMPC_OK;
}

qmpc_status qmpc_last_kernel_ms(qmpc_handle* h, float* ms) {
  if (!h || !ms) return QMPC_BAD_ARGUMENT;
  if (!h->timed) { *ms = 0.0f; return QMPC_OK; }
  HIP_TRY(hipEventSynchronize(h->ev1));
  HIP_TRY(hipEventElapsedTime(ms, h->ev0, h->ev1));
  return QMPC_OK;
}

// What kind of memory is p?  1: host memory the device can address (hipHostMalloc / hipHostRegister / qmpc_host_alloc; *dev =
// its device-side alias), 2: device (or managed) memory -- a caller that hands a host-buffer entry point such a pointer gets
// the explicit-copy path, which takes any kind --, 0: pageable host memory (or unknown to the runtime)
static int pointer_kind(const void* p, void** dev) {
  hipPointerAttribute_t a;
  std::memset(&a, 0, sizeof a);
  if (hipPointerGetAttributes(&a, p) != hipSuccess) { (void)hipGetLastError(); return 0; }
  if (a.type == hipMemoryTypeHost && a.devicePointer) { *dev = a.devicePointer; return 1; }
  if (a.type == hipMemoryTypeDevice || a.type == hipMemoryTypeManaged || a.type == hipMemoryTypeArray) return 2;
  return 0;
}
static qmpc_status ensure_stage(qmpc_handle* h, int nl) {
  if (h->h_stage_in) return QMPC_OK;
  const size_t rec = sizeof(double) * (32 + 4 * nl), out = sizeof(double) * 3 * nl + sizeof(qmpc_info);
  void *a = nullptr, *b = nullptr;
  if (hipHostMalloc(&a, rec * (size_t)h->max_batch, hipHostMallocDefault) != hipSuccess ||
      hipHostMalloc(&b, out * (size_t)h->max_batch, hipHostMallocDefault) != hipSuccess) {
    (void)hipGetLastError();
    if (a) (void)hipHostFree(a);
    h->zero_copy = 0;            // no pinned memory to be had: the copies below take the runtime's own staging
    return QMPC_OK;
  }
  h->h_stage_in = static_cast<unsigned char*>(a);
  h->h_stage_out = static_cast<unsigned char*>(b);
  return QMPC_OK;
}
// copy-out owed to a pageable caller of the zero-copy path (after the stream has drained)
static void finish_pending(qmpc_handle* h) {
  if (h->pending.forces) std::memcpy(h->pending.forces, h->h_stage_out, h->pending.fbytes);
  if (h->pending.info) std::memcpy(h->pending.info, h->h_stage_out + h->pending.fbytes, h->pending.ibytes);
  h->pending.forces = nullptr;
  h->pending.info = nullptr;
}

// host-buffer solve shared by the models (nx = doubles per state in traj_x).
// Batches that take a wave-per-instance kernel run ZERO-COPY (see qmpc_handle): records are read from, forces and status
// written to, host memory the device can address -- the caller's own buffers when they are pinned (qmpc_host_alloc,
// hipHostMalloc, hipHostRegister), the handle's pinned staging otherwise (one memcpy in, one out).  Lane-kernel batches
// (which sort and re-read their records) and calls that want trajectories keep explicit copies on the handle's stream.
static qmpc_status solve_host(qmpc_handle* h, int32_t batch, const qmpc_input* in, double* forces_body,
                              qmpc_info* info, double* traj_u, double* traj_x, int model, int nx,
                              bool blocking = true) {
  if (!h || batch < 0 || (batch > 0 && (!in || !forces_body))) return QMPC_BAD_ARGUMENT;
  if (h->params.model != model) return QMPC_BAD_ARGUMENT;
  if (batch == 0) return QMPC_OK;
  if (batch > h->max_batch) return QMPC_BATCH_TOO_LARGE;
  HIP_TRY(hipSetDevice(h->device));
  // an earlier qmpc_solve_async was never waited for: complete it first -- it owes a pageable caller its copy-out, or its
  // kernel may still be reading records out of the staging this call is about to refill
  if (h->pending.forces || h->pending.info || h->stage_in_busy) {
    HIP_TRY(hipStreamSynchronize(h->stream));
    finish_pending(h);
    h->stage_in_busy = 0;
  }
  const int N = h->params.horizon;
  const int nl = model_nl(model), nu = 3 * nl;
  const size_t rec = sizeof(double) * (32 + 4 * nl);
  const size_t fbytes = sizeof(double) * nu * (size_t)batch, ibytes = sizeof(qmpc_info) * (size_t)batch;
  if (traj_u && !h->d_traj_u) HIP_TRY(hipMalloc(&h->d_traj_u, sizeof(double) * nu * N * (size_t)h->max_batch));
  if (traj_x && !h->d_traj_x) HIP_TRY(hipMalloc(&h->d_traj_x, sizeof(double) * 13 * (N + 1) * (size_t)h->max_batch));
  const bool lane = use_lane(h, batch, nullptr, nullptr) || ref_lane_batch(h, batch);
  if (h->zero_copy && !lane) {
    void *din = nullptr, *df = nullptr, *di = nullptr;
    const int k_in = pointer_kind(in, &din), k_f = pointer_kind(forces_body, &df), k_i = info ? pointer_kind(info, &di) : 1;
    const bool in_pinned = k_in == 1, out_pinned = k_f == 1 && k_i == 1;
    const bool any_device = k_in == 2 || k_f == 2 || k_i == 2;      // not host buffers at all: the copy path below takes them
    if (!any_device && (!in_pinned || !out_pinned)) {
      const qmpc_status es = ensure_stage(h, nl);
      if (es != QMPC_OK) return es;
    }
    if (h->zero_copy && !any_device) {
      if (!in_pinned) { std::memcpy(h->h_stage_in, in, rec * (size_t)batch); din = h->h_stage_in; }
      if (!out_pinned) { df = h->h_stage_out; di = h->h_stage_out + fbytes; }
      const qmpc_status st = launch_solve(h, batch, static_cast<const qmpc_input*>(din), static_cast<double*>(df),
                                          info ? static_cast<qmpc_info*>(di) : h->d_info, traj_u ? h->d_traj_u : nullptr,
                                          traj_x ? h->d_traj_x : nullptr, h->stream);
      if (st != QMPC_OK) return st;
      if (traj_u) HIP_TRY(hipMemcpyAsync(traj_u, h->d_traj_u, sizeof(double) * nu * N * (size_t)batch, hipMemcpyDeviceToHost, h->stream));
      if (traj_x) HIP_TRY(hipMemcpyAsync(traj_x, h->d_traj_x, sizeof(double) * nx * (N + 1) * (size_t)batch, hipMemcpyDeviceToHost, h->stream));
      if (!out_pinned) {
        h->pending.forces = forces_body; h->pending.fbytes = fbytes;
        h->pending.info = info; h->pending.ibytes = ibytes;
      }
      if (blocking) {
        HIP_TRY(hipStreamSynchronize(h->stream));
        finish_pending(h);
      } else if (!in_pinned) {
        h->stage_in_busy = 1;
      }
      return QMPC_OK;
    }
  }
  HIP_TRY(hipMemcpyAsync(h->d_in, in, rec * (size_t)batch, hipMemcpyDefault, h->stream));
  const qmpc_status st = launch_solve(h, batch, h->d_in, h->d_forces, h->d_info, traj_u ? h->d_traj_u : nullptr,
                                      traj_x ? h->d_traj_x : nullptr, h->stream);
  if (st != QMPC_OK) return st;
  HIP_TRY(hipMemcpyAsync(forces_body, h->d_forces, fbytes, hipMemcpyDefault, h->stream));
  if (info) HIP_TRY(hipMemcpyAsync(info, h->d_info, ibytes, hipMemcpyDefault, h->stream));
  if (traj_u) HIP_TRY(hipMemcpyAsync(traj_u, h->d_traj_u, sizeof(double) * nu * N * (size_t)batch, hipMemcpyDeviceToHost, h->stream));
  if (traj_x) HIP_TRY(hipMemcpyAsync(traj_x, h->d_traj_x, sizeof(double) * nx * (N + 1) * (size_t)batch, hipMemcpyDeviceToHost, h->stream));
  if (blocking) HIP_TRY(hipStreamSynchronize(h->stream));
  return QMPC_OK;
}

// ---- warm-started solve (converged mode, QuatMpc): every instance starts from u_init shifted by one knot ----------------
qmpc_status qmpc_solve_warm_device(qmpc_handle* h, int32_t batch, const qmpc_input* d_in, const double* d_u_init,
                                   double* d_forces_body, qmpc_info* d_info, double* d_traj_u, void* stream) {
  if (!h || batch < 0 || (batch > 0 && (!d_in || !d_forces_body))) return QMPC_BAD_ARGUMENT;
  if (h->params.model != QMPC_MODEL_QUAT || h->params.mode != QMPC_MODE_CONVERGED) return QMPC_BAD_ARGUMENT;
  if (batch == 0) return QMPC_OK;
  if (batch > h->max_batch) return QMPC_BATCH_TOO_LARGE;
  HIP_TRY(hipSetDevice(h->device));
  if (use_lane(h, batch, nullptr, nullptr, true))      // large batches: the lane-per-instance kernel, same start rule
    return launch_lane(h, batch, d_in, d_forces_body, d_info, stream ? (hipStream_t)stream : h->stream, d_u_init, d_traj_u, 0);
  const int var = body_variant(h, batch);
  HIP_TRY(qmpc_warm_launch(var, 0, (int)batch, variant_lds(h, var), stream ? (hipStream_t)stream : h->stream, &h->dev, sizeof h->dev, d_in, d_u_init,
                           d_forces_body, d_info, d_traj_u, variant_gws(h, var), 0));
  return QMPC_OK;
}

qmpc_status qmpc_solve_warm(qmpc_handle* h, int32_t batch, const qmpc_input* in, const double* u_init, double* forces_body,
                            qmpc_info* info, double* traj_u) {
  if (!h || batch < 0 || (batch > 0 && (!in || !forces_body))) return QMPC_BAD_ARGUMENT;
  if (h->params.model != QMPC_MODEL_QUAT || h->params.mode != QMPC_MODE_CONVERGED) return QMPC_BAD_ARGUMENT;
  if (batch == 0) return QMPC_OK;
  if (batch > h->max_batch) return QMPC_BATCH_TOO_LARGE;
  HIP_TRY(hipSetDevice(h->device));
  const int N = h->params.horizon;
  const size_t nu = sizeof(double) * 12 * N * (size_t)batch;
  // the device trajectory buffer doubles as the staging of u_init (read before it is overwritten: one wave per instance
  // loads its slice into LDS first)
  if (!h->d_traj_u) HIP_TRY(hipMalloc(&h->d_traj_u, sizeof(double) * 12 * N * (size_t)h->max_batch));
  HIP_TRY(hipMemcpyAsync(h->d_in, in, sizeof(qmpc_input) * (size_t)batch, hipMemcpyHostToDevice, h->stream));
  if (u_init) HIP_TRY(hipMemcpyAsync(h->d_traj_u, u_init, nu, hipMemcpyHostToDevice, h->stream));
  const qmpc_status st = qmpc_solve_warm_device(h, batch, h->d_in, u_init ? h->d_traj_u : nullptr, h->d_forces, h->d_info,
                                                h->d_traj_u, h->stream);
  if (st != QMPC_OK) return st;
  HIP_TRY(hipMemcpyAsync(forces_body, h->d_forces, sizeof(double) * 12 * (size_t)batch, hipMemcpyDeviceToHost, h->stream));
  if (info) HIP_TRY(hipMemcpyAsync(info, h->d_info, sizeof(qmpc_info) * (size_t)batch, hipMemcpyDeviceToHost, h->stream));
  if (traj_u) HIP_TRY(hipMemcpyAsync(traj_u, h->d_traj_u, nu, hipMemcpyDeviceToHost, h->stream));
  HIP_TRY(hipStreamSynchronize(h->stream));
  return QMPC_OK;
}

// non-blocking host-buffer call: copies and kernel are queued on the handle's stream; qmpc_wait completes them
qmpc_status qmpc_solve_async(qmpc_handle* h, int32_t batch, const qmpc_input* in, double* forces_body, qmpc_info* info) {
  return solve_host(h, batch, in, forces_body, info, nullptr, nullptr, QMPC_MODEL_QUAT, 13, false);
}

qmpc_status qmpc_solve_traj(qmpc_handle* h, int32_t batch, const qmpc_input* in, double* forces_body,
                            qmpc_info* info, double* traj_u, double* traj_x) {
  return solve_host(h, batch, in, forces_body, info, traj_u, traj_x, QMPC_MODEL_QUAT, 13);
}

qmpc_status qmpc_solve8_traj(qmpc_handle* h, int32_t batch, const qmpc_input8* in, double* forces_body,
                             qmpc_info* info, double* traj_u, double* traj_x) {
  return solve_host(h, batch, reinterpret_cast<const qmpc_input*>(in), forces_body, info, traj_u, traj_x,
                    QMPC_MODEL_QUAT8, 13);
}

qmpc_status qmpc_solve8(qmpc_handle* h, int32_t batch, const qmpc_input8* in, double* forces_body, qmpc_info* info) {
  return qmpc_solve8_traj(h, batch, in, forces_body, info, nullptr, nullptr);
}

qmpc_status qmpc_convex_solve_traj(qmpc_handle* h, int32_t batch, const qmpc_convex_input* in, double* forces_world,
                                   qmpc_info* info, double* traj_u, double* traj_x) {
  return solve_host(h, batch, reinterpret_cast<const qmpc_input*>(in), forces_world, info, traj_u, traj_x,
                    QMPC_MODEL_CONVEX, 12);
}

qmpc_status qmpc_convex_solve(qmpc_handle* h, int32_t batch, const qmpc_convex_input* in, double* forces_world,
                              qmpc_info* info) {
  return qmpc_convex_solve_traj(h, batch, in, forces_world, info, nullptr, nullptr);
}

qmpc_status qmpc_solve(qmpc_handle* h, int32_t batch, const qmpc_input* in, double* forces_body, qmpc_info* info) {
  return qmpc_solve_traj(h, batch, in, forces_body, info, nullptr, nullptr);
}

static qmpc_status linearize_host(qmpc_handle* h, int32_t batch, const qmpc_input* in, double* Abar, double* Bbar,
                                  double* X, int model, int nx) {
  if (!h || batch < 0 || (batch > 0 && (!in || !Abar || !Bbar || !X))) return QMPC_BAD_ARGUMENT;
  if (h->params.model != model) return QMPC_BAD_ARGUMENT;
  if (batch == 0) return QMPC_OK;
  if (batch > h->max_batch) return QMPC_BATCH_TOO_LARGE;
  HIP_TRY(hipSetDevice(h->device));
  const int N = h->params.horizon;
  const size_t nA = sizeof(double) * 144 * N * (size_t)h->max_batch;
  if (!h->d_A) HIP_TRY(hipMalloc(&h->d_A, nA));
  if (!h->d_B) HIP_TRY(hipMalloc(&h->d_B, nA));
  if (!h->d_traj_x) HIP_TRY(hipMalloc(&h->d_traj_x, sizeof(double) * 13 * (N + 1) * (size_t)h->max_batch));
  HIP_TRY(hipMemcpyAsync(h->d_in, in, sizeof(qmpc_input) * (size_t)batch, hipMemcpyHostToDevice, h->stream));
  if (model == QMPC_MODEL_CONVEX)
    hipLaunchKernelGGL(qmpc_linearize_kernel<ConvexModel>, dim3((unsigned)batch), dim3(kWave), h->lds_bytes_g, h->stream,
                       h->dev, h->d_in, h->d_A, h->d_B, h->d_traj_x, (int)batch);
  else
    hipLaunchKernelGGL(qmpc_linearize_kernel<QuatModel>, dim3((unsigned)batch), dim3(kWave), h->lds_bytes_g, h->stream,
                       h->dev, h->d_in, h->d_A, h->d_B, h->d_traj_x, (int)batch);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipMemcpyAsync(Abar, h->d_A, sizeof(double) * 144 * N * (size_t)batch, hipMemcpyDeviceToHost, h->stream));
  HIP_TRY(hipMemcpyAsync(Bbar, h->d_B, sizeof(double) * 144 * N * (size_t)batch, hipMemcpyDeviceToHost, h->stream));
  HIP_TRY(hipMemcpyAsync(X, h->d_traj_x, sizeof(double) * nx * (N + 1) * (size_t)batch, hipMemcpyDeviceToHost, h->stream));
  HIP_TRY(hipStreamSynchronize(h->stream));
  return QMPC_OK;
}

qmpc_status qmpc_linearize(qmpc_handle* h, int32_t batch, const qmpc_input* in, double* Abar, double* Bbar, double* X) {
  return linearize_host(h, batch, in, Abar, Bbar, X, QMPC_MODEL_QUAT, 13);
}

qmpc_status qmpc_convex_linearize(qmpc_handle* h, int32_t batch, const qmpc_convex_input* in, double* A, double* B,
                                  double* X) {
  return linearize_host(h, batch, reinterpret_cast<const qmpc_input*>(in), A, B, X, QMPC_MODEL_CONVEX, 12);
}

// ---- pinned host buffers, eager allocation, handle queries ------------------------------------------------------------------
void* qmpc_host_alloc(size_t bytes) {
  void* p = nullptr;
  if (bytes == 0 || hipHostMalloc(&p, bytes, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
  return p;
}
void qmpc_host_free(void* p) {
  if (p) (void)hipHostFree(p);
}

// Everything a solve of `batch` instances will need, allocated NOW: the lane kernel's workspace and sort scratch, the
// hand-off records, the pinned staging of the host-buffer calls.  Afterwards no solve of up to `batch` instances allocates
// (safe inside the caller's own stream capture) and qmpc_query(QMPC_QUERY_HANDOFF_ACTIVE) says which family of roundings
// the handle's large-batch results belong to.
qmpc_status qmpc_prepare(qmpc_handle* h, int32_t batch) {
  if (!h || batch < 1) return QMPC_BAD_ARGUMENT;
  if (batch > h->max_batch) return QMPC_BATCH_TOO_LARGE;
  HIP_TRY(hipSetDevice(h->device));
  const bool ref_lane = ref_lane_batch(h, batch);
  bool lane_loop = false;
  if (h->params.mode == QMPC_MODE_CONVERGED && h->lane_pslot >= 0 && h->variant == 0)
    lane_loop = batch >= (h->lane_min_loop_cold < h->lane_min_batch ? h->lane_min_loop_cold : h->lane_min_batch);
  if (use_lane(h, batch, nullptr, nullptr) || ref_lane || lane_loop) {
    const qmpc_status es = ensure_lane_buffers(h);
    if (es != QMPC_OK) return es;
    if (!ref_lane && (handoff_cap(h, 1) || handoff_cap(h, 2) || handoff_cap(h, 3))) (void)ensure_handoff_buffers(h);
  }
  // the pinned staging of the host-buffer calls: ALWAYS (a handle prepared for a lane-kernel batch may still be handed a smaller
  // batch on host buffers, which runs zero-copy), and the buffers the closed loops and the trajectory / warm-started calls
  // otherwise allocate on first use -- nothing of that may happen inside a caller's stream capture
  if (h->zero_copy) {
    const qmpc_status es = ensure_stage(h, model_nl(h->params.model));
    if (es != QMPC_OK) return es;
  }
  const int nu = 3 * model_nl(h->params.model), N = h->params.horizon;
  if (!h->d_traj_u) HIP_TRY(hipMalloc(&h->d_traj_u, sizeof(double) * nu * N * (size_t)h->max_batch));
  if (!h->d_traj_x) HIP_TRY(hipMalloc(&h->d_traj_x, sizeof(double) * 13 * (N + 1) * (size_t)h->max_batch));
  if (!h->d_loop_row) HIP_TRY(hipMalloc(&h->d_loop_row, sizeof(int) * 4));
  return QMPC_OK;
}

// kernel family launch_solve gives a plain solve of `batch` instances on this handle (QMPC_KERNEL_*)
static int kernel_for_batch(const qmpc_handle* h, int32_t batch) {
  if (h->params.mode == QMPC_MODE_REFERENCE) {
    if (ref_lane_batch(h, batch)) return QMPC_KERNEL_LANE;
    if (const int wv = ref_wform_variant(h, batch)) return wv >= 5 ? QMPC_KERNEL_WFORM_WS : QMPC_KERNEL_WFORM_LDS;
    const bool ws = batch > 1024 || h->lds_bytes > 40 * 1024 || h->variant >= 2 || h->params.model == QMPC_MODEL_QUAT8;
    return ws ? QMPC_KERNEL_DENSE_WS : QMPC_KERNEL_DENSE_LDS;
  }
  if (use_lane(h, batch, nullptr, nullptr)) return handoff_cap(h, 1) ? QMPC_KERNEL_LANE_HANDOFF : QMPC_KERNEL_LANE;
  if (const int wv = wform_variant(h, batch)) return wv >= 5 ? QMPC_KERNEL_WFORM_WS : QMPC_KERNEL_WFORM_LDS;
  return pick_variant(h, batch) >= 1 ? QMPC_KERNEL_DENSE_WS : QMPC_KERNEL_DENSE_LDS;
}

qmpc_status qmpc_query(qmpc_handle* h, int32_t what, int64_t arg, int64_t* value) {
  if (!h || !value) return QMPC_BAD_ARGUMENT;
  switch (what) {
    case QMPC_QUERY_HANDOFF_ACTIVE:      // 1: capped lane launches hand their stragglers over; 0: pure lane kernel (off, or allocation failed)
      *value = (handoff_cap(h, 1) || handoff_cap(h, 2) || handoff_cap(h, 3)) ? 1 : 0;
      return QMPC_OK;
    case QMPC_QUERY_HANDOFF_ALLOC_FAILED: *value = h->handoff_failed; return QMPC_OK;
    case QMPC_QUERY_KERNEL_FOR_BATCH:
      if (arg < 1 || arg > h->max_batch) return QMPC_BAD_ARGUMENT;
      *value = kernel_for_batch(h, (int32_t)arg);
      return QMPC_OK;
    case QMPC_QUERY_LAST_KERNEL: *value = h->last_kernel; return QMPC_OK;
    case QMPC_QUERY_LANE_CAP: *value = handoff_cap(h, arg == 3 ? 3 : (arg == 2 ? 2 : 1)); return QMPC_OK;
    case QMPC_QUERY_DEVICE_BYTES: {      // device memory the handle holds right now
      const int N = h->params.horizon, nl = model_nl(h->params.model), nu = 3 * nl;
      size_t b = (sizeof(double) * (32 + 4 * nl) + sizeof(double) * nu + sizeof(qmpc_info)) * (size_t)h->max_batch;
      b += sizeof(double) * (size_t)N * (13 * nu + 21 * nl + 30 * nl) * (size_t)h->max_batch;
      if (h->d_lane_ws) b += qmpc_lane_ws_bytes(N, nl, h->lane_slots, h->params.mode == QMPC_MODE_REFERENCE) + qmpc_lane_scratch_bytes(h->max_batch);
      if (h->d_handoff) b += qmpc_lane_handoff_list_bytes(h->max_batch) + sizeof(double) * qmpc_lane_handoff_record_doubles(N) * (size_t)h->hstate_cap;
      if (h->d_traj_u) b += sizeof(double) * nu * N * (size_t)h->max_batch;
      if (h->d_traj_x) b += sizeof(double) * 13 * (N + 1) * (size_t)h->max_batch;
      if (h->d_A) b += 2 * sizeof(double) * 144 * N * (size_t)h->max_batch;
      b += sizeof(double) * (h->leg_cap + h->loop_cap);
      *value = (int64_t)b;
      return QMPC_OK;
    }
    case QMPC_QUERY_ZERO_COPY: *value = h->zero_copy; return QMPC_OK;
    default: return QMPC_BAD_ARGUMENT;
  }
}

// ---- multi-GPU: the one collective of the path --------------------------------
// RCCL is resolved at run time (dlopen), so the library carries no link-time dependency on it and a
// single-GPU user never loads it.  ncclAllGather(sendbuff, recvbuff, sendcount, datatype, comm, stream);
// ncclFloat64 = 8 in nccl.h.
typedef int (*nccl_all_gather_fn)(const void*, void*, size_t, int, void*, hipStream_t);
static nccl_all_gather_fn resolve_all_gather() {
  static nccl_all_gather_fn fn = nullptr;
  static std::once_flag once;      // two handles may gather first from two threads
  std::call_once(once, [] {
    void* sym = dlsym(RTLD_DEFAULT, "ncclAllGather");          // already in the process (e.g. under torch)?
    if (!sym) {
      const char* names[] = {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so"};
      for (const char* n : names) {
        void* lib = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
        if (lib && (sym = dlsym(lib, "ncclAllGather"))) break;
      }
    }
    fn = reinterpret_cast<nccl_all_gather_fn>(sym);
  });
  return fn;
}

qmpc_status qmpc_gather(qmpc_handle* h, void* nccl_comm, const double* d_local, int64_t count, double* d_all,
                        void* stream) {
  if (!h || !nccl_comm || count < 0 || (count > 0 && (!d_local || !d_all))) return QMPC_BAD_ARGUMENT;
  if (count == 0) return QMPC_OK;
  nccl_all_gather_fn fn = resolve_all_gather();
  if (!fn) {
    std::fprintf(stderr, "qmpc_gather: RCCL (ncclAllGather) not found\n");
    return QMPC_UNSUPPORTED;
  }
  HIP_TRY(hipSetDevice(h->device));
  const int rc = fn(d_local, d_all, (size_t)count, /*ncclFloat64*/ 8, nccl_comm, stream ? (hipStream_t)stream : h->stream);
  if (rc != 0) {
    std::fprintf(stderr, "qmpc_gather: ncclAllGather failed (%d)\n", rc);
    return QMPC_HIP_ERROR;
  }
  return QMPC_OK;
}

// ---- leg kinematics / torque map (BaseInterface.cpp:10-34,209-212,343-408) ----------------
void qmpc_default_go1_geometry(qmpc_leg_geometry* g) {
  std::memset(g, 0, sizeof *g);
  const double sx[4] = {1, 1, -1, -1}, sy[4] = {1, -1, 1, -1};
  for (int l = 0; l < 4; ++l) {
    g->rho_fix[l][0] = sx[l] * 0.1881;    // leg_offset_x
    g->rho_fix[l][1] = sy[l] * 0.04675;   // leg_offset_y
    g->rho_fix[l][2] = sy[l] * 0.0812;    // motor_offset
    g->rho_fix[l][3] = 0.213;             // UPPER_LEG_LENGTH
    g->rho_fix[l][4] = 0.213;             // LOWER_LEG_LENGTH
  }
}

static_assert(sizeof(LegGeom) == sizeof(qmpc_leg_geometry), "kernel argument mirrors the ABI struct");

static qmpc_status launch_leg(const qmpc_leg_geometry* g, int32_t batch, const double* d_q,
                              const double* d_f, const double* d_c, int walking, double* d_p, double* d_J,
                              double* d_tau, hipStream_t s) {
  LegGeom G;
  std::memcpy(&G, g, sizeof G);
  const bool aligned = ((reinterpret_cast<uintptr_t>(d_q) | reinterpret_cast<uintptr_t>(d_f) |
                         reinterpret_cast<uintptr_t>(d_tau)) & 15u) == 0;
  if (d_tau && !d_p && !d_J && aligned) {      // the torque map proper: LDS-staged streaming pass, 64 instances per block
    hipLaunchKernelGGL(qmpc_tau_kernel, dim3((unsigned)((batch + 63) / 64)), dim3(256), 0, s, G, d_q, d_f, d_c, walking,
                       d_tau, (int)batch);
    HIP_TRY(hipGetLastError());
    return QMPC_OK;
  }
  const unsigned threads = 256, blocks = (unsigned)(((size_t)batch * 4 + threads - 1) / threads);
  hipLaunchKernelGGL(qmpc_leg_kernel, dim3(blocks), dim3(threads), 0, s, G, d_q, d_f, d_c, walking, d_p, d_J, d_tau,
                     (int)batch);
  HIP_TRY(hipGetLastError());
  return QMPC_OK;
}

qmpc_status qmpc_torque_map_device(qmpc_handle* h, const qmpc_leg_geometry* g, int32_t batch, const double* d_joint_pos,
                                   const double* d_forces_body, const double* d_contacts, int32_t walking,
                                   double* d_tau, void* stream) {
  if (!h || !g || batch < 0 || (batch > 0 && (!d_joint_pos || !d_forces_body || !d_tau))) return QMPC_BAD_ARGUMENT;
  if (batch == 0) return QMPC_OK;
  HIP_TRY(hipSetDevice(h->device));
  return launch_leg(g, batch, d_joint_pos, d_forces_body, d_contacts, walking, nullptr, nullptr, d_tau,
                    stream ? (hipStream_t)stream : h->stream);
}

// host-buffer variants (not the hot path): staged through a buffer the handle keeps
static qmpc_status leg_host(qmpc_handle* h, const qmpc_leg_geometry* g, int32_t batch, const double* q, const double* f,
                            const double* c, int walking, double* p, double* J, double* tau) {
  HIP_TRY(hipSetDevice(h->device));
  const size_t B = (size_t)batch;
  // layout: q[12B] f[12B] c[4B] p[12B] J[36B] tau[12B]; the staging buffer belongs to the handle and only grows
  if (h->leg_cap < B * 88) {
    if (h->d_leg) (void)hipFree(h->d_leg);
    h->d_leg = nullptr;
    h->leg_cap = 0;
    HIP_TRY(hipMalloc(&h->d_leg, sizeof(double) * B * 88));
    h->leg_cap = B * 88;
  }
  double* d = h->d_leg;
  double *dq = d, *df = d + 12 * B, *dc = d + 24 * B, *dp = d + 28 * B, *dJ = d + 40 * B, *dt = d + 76 * B;
  qmpc_status st = QMPC_OK;
  do {
    if (hipMemcpyAsync(dq, q, sizeof(double) * 12 * B, hipMemcpyHostToDevice, h->stream) != hipSuccess) { st = QMPC_HIP_ERROR; break; }
    if (f && hipMemcpyAsync(df, f, sizeof(double) * 12 * B, hipMemcpyHostToDevice, h->stream) != hipSuccess) { st = QMPC_HIP_ERROR; break; }
    if (c && hipMemcpyAsync(dc, c, sizeof(double) * 4 * B, hipMemcpyHostToDevice, h->stream) != hipSuccess) { st = QMPC_HIP_ERROR; break; }
    st = launch_leg(g, batch, dq, f ? df : nullptr, c ? dc : nullptr, walking, p ? dp : nullptr, J ? dJ : nullptr,
                    tau ? dt : nullptr, h->stream);
    if (st != QMPC_OK) break;
    if (p && hipMemcpyAsync(p, dp, sizeof(double) * 12 * B, hipMemcpyDeviceToHost, h->stream) != hipSuccess) { st = QMPC_HIP_ERROR; break; }
    if (J && hipMemcpyAsync(J, dJ, sizeof(double) * 36 * B, hipMemcpyDeviceToHost, h->stream) != hipSuccess) { st = QMPC_HIP_ERROR; break; }
    if (tau && hipMemcpyAsync(tau, dt, sizeof(double) * 12 * B, hipMemcpyDeviceToHost, h->stream) != hipSuccess) { st = QMPC_HIP_ERROR; break; }
    if (hipStreamSynchronize(h->stream) != hipSuccess) st = QMPC_HIP_ERROR;
  } while (0);
  return st;
}

qmpc_status qmpc_leg_kinematics(qmpc_handle* h, const qmpc_leg_geometry* g, int32_t batch, const double* joint_pos,
                                double* foot_pos_body, double* jac) {
  if (!h || !g || batch < 0 || (batch > 0 && (!joint_pos || (!foot_pos_body && !jac)))) return QMPC_BAD_ARGUMENT;
  if (batch == 0) return QMPC_OK;
  return leg_host(h, g, batch, joint_pos, nullptr, nullptr, 0, foot_pos_body, jac, nullptr);
}

qmpc_status qmpc_torque_map(qmpc_handle* h, const qmpc_leg_geometry* g, int32_t batch, const double* joint_pos,
                            const double* forces_body, const double* contacts, int32_t walking, double* tau) {
  if (!h || !g || batch < 0 || (batch > 0 && (!joint_pos || !forces_body || !tau))) return QMPC_BAD_ARGUMENT;
  if (batch == 0) return QMPC_OK;
  return leg_host(h, g, batch, joint_pos, forces_body, contacts, walking, nullptr, nullptr, tau);
}

// ---- joint-level commands (BaseInterface.cpp:343-408; kernels in qmpc_joint.hip) ------------------
static qmpc_status grow_leg_staging(qmpc_handle* h, size_t doubles) {
  if (h->leg_cap >= doubles) return QMPC_OK;
  if (h->d_leg) (void)hipFree(h->d_leg);
  h->d_leg = nullptr;
  h->leg_cap = 0;
  HIP_TRY(hipMalloc(&h->d_leg, sizeof(double) * doubles));
  h->leg_cap = doubles;
  return QMPC_OK;
}

qmpc_status qmpc_joint_commands_device(qmpc_handle* h, const qmpc_leg_geometry* g, int32_t batch,
                                       const qmpc_joint_feedback* d_fb, qmpc_joint_command* d_cmd, void* stream) {
  if (!h || !g || batch < 0 || (batch > 0 && (!d_fb || !d_cmd))) return QMPC_BAD_ARGUMENT;
  if (batch == 0) return QMPC_OK;
  HIP_TRY(hipSetDevice(h->device));
  LegGeom G;
  std::memcpy(&G, g, sizeof G);
  hipLaunchKernelGGL(qmpc_joint_cmd_kernel, dim3((unsigned)((batch + kJointTile - 1) / kJointTile)), dim3(256), 0,
                     stream ? (hipStream_t)stream : h->stream, G, d_fb, d_cmd, (int)batch);
  HIP_TRY(hipGetLastError());
  return QMPC_OK;
}

qmpc_status qmpc_joint_commands(qmpc_handle* h, const qmpc_leg_geometry* g, int32_t batch, const qmpc_joint_feedback* fb,
                                qmpc_joint_command* cmd) {
  if (!h || !g || batch < 0 || (batch > 0 && (!fb || !cmd))) return QMPC_BAD_ARGUMENT;
  if (batch == 0) return QMPC_OK;
  HIP_TRY(hipSetDevice(h->device));
  const size_t B = (size_t)batch;
  const qmpc_status gs = grow_leg_staging(h, B * (kJointFb + kJointCmd));
  if (gs != QMPC_OK) return gs;
  qmpc_joint_feedback* dfb = reinterpret_cast<qmpc_joint_feedback*>(h->d_leg);
  qmpc_joint_command* dcmd = reinterpret_cast<qmpc_joint_command*>(h->d_leg + B * kJointFb);
  HIP_TRY(hipMemcpyAsync(dfb, fb, sizeof(qmpc_joint_feedback) * B, hipMemcpyHostToDevice, h->stream));
  const qmpc_status st = qmpc_joint_commands_device(h, g, batch, dfb, dcmd, h->stream);
  if (st != QMPC_OK) return st;
  HIP_TRY(hipMemcpyAsync(cmd, dcmd, sizeof(qmpc_joint_command) * B, hipMemcpyDeviceToHost, h->stream));
  HIP_TRY(hipStreamSynchronize(h->stream));
  return QMPC_OK;
}

qmpc_status qmpc_leg_inverse_kinematics(qmpc_handle* h, const qmpc_leg_geometry* g, int32_t batch,
                                        const double* foot_pos_body, const double* cur_joint_pos, double* joint_pos) {
  if (!h || !g || batch < 0 || (batch > 0 && (!foot_pos_body || !cur_joint_pos || !joint_pos))) return QMPC_BAD_ARGUMENT;
  if (batch == 0) return QMPC_OK;
  HIP_TRY(hipSetDevice(h->device));
  const size_t B = (size_t)batch;
  const qmpc_status gs = grow_leg_staging(h, B * 36);
  if (gs != QMPC_OK) return gs;
  double *dp = h->d_leg, *dc = h->d_leg + 12 * B, *dq = h->d_leg + 24 * B;
  HIP_TRY(hipMemcpyAsync(dp, foot_pos_body, sizeof(double) * 12 * B, hipMemcpyHostToDevice, h->stream));
  HIP_TRY(hipMemcpyAsync(dc, cur_joint_pos, sizeof(double) * 12 * B, hipMemcpyHostToDevice, h->stream));
  LegGeom G;
  std::memcpy(&G, g, sizeof G);
  hipLaunchKernelGGL(qmpc_leg_inverse_kernel, dim3((unsigned)((B * 4 + 255) / 256)), dim3(256), 0, h->stream, G,
                     (const double*)dp, (const double*)dc, dq, (int)batch);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipMemcpyAsync(joint_pos, dq, sizeof(double) * 12 * B, hipMemcpyDeviceToHost, h->stream));
  HIP_TRY(hipStreamSynchronize(h->stream));
  return QMPC_OK;
}

qmpc_status qmpc_loop_joint_commands_device(qmpc_handle* h, const qmpc_leg_geometry* g, int32_t batch,
                                            const qmpc_loop_state* d_states, double* d_joint_pos,
                                            qmpc_joint_feedback* d_fb, qmpc_joint_command* d_cmd, void* stream) {
  if (!h || !g || batch < 0 || (batch > 0 && (!d_states || !d_joint_pos || !d_cmd))) return QMPC_BAD_ARGUMENT;
  if (batch == 0) return QMPC_OK;
  HIP_TRY(hipSetDevice(h->device));
  LegGeom G;
  std::memcpy(&G, g, sizeof G);
  hipLaunchKernelGGL(qmpc_loop_joint_kernel, dim3((unsigned)(((size_t)batch * 4 + 255) / 256)), dim3(256), 0,
                     stream ? (hipStream_t)stream : h->stream, G, d_states, d_joint_pos, d_fb, d_cmd,
                     (qmpc_joint_command*)nullptr, (const int*)nullptr, (int)batch);
  HIP_TRY(hipGetLastError());
  return QMPC_OK;
}

// host-buffer form of the above (not the hot path): staged through temporaries
qmpc_status qmpc_loop_joint_commands(qmpc_handle* h, const qmpc_leg_geometry* g, int32_t batch, const qmpc_loop_state* states,
                                     double* joint_pos, qmpc_joint_feedback* fb, qmpc_joint_command* cmd) {
  if (!h || !g || batch < 0 || (batch > 0 && (!states || !joint_pos || !cmd))) return QMPC_BAD_ARGUMENT;
  if (batch == 0) return QMPC_OK;
  HIP_TRY(hipSetDevice(h->device));
  const size_t B = (size_t)batch;
  qmpc_loop_state* d_st = nullptr;
  qmpc_status rs = QMPC_OK;
  do {
    if (hipMalloc(&d_st, sizeof(qmpc_loop_state) * B) != hipSuccess) { rs = QMPC_HIP_ERROR; break; }
    rs = grow_leg_staging(h, B * (12 + kJointFb + kJointCmd));
    if (rs != QMPC_OK) break;
    double* d_jp = h->d_leg;
    qmpc_joint_feedback* d_fb = reinterpret_cast<qmpc_joint_feedback*>(h->d_leg + 12 * B);
    qmpc_joint_command* d_cmd = reinterpret_cast<qmpc_joint_command*>(h->d_leg + (12 + kJointFb) * B);
    if (hipMemcpyAsync(d_st, states, sizeof(qmpc_loop_state) * B, hipMemcpyHostToDevice, h->stream) != hipSuccess ||
        hipMemcpyAsync(d_jp, joint_pos, sizeof(double) * 12 * B, hipMemcpyHostToDevice, h->stream) != hipSuccess) { rs = QMPC_HIP_ERROR; break; }
    rs = qmpc_loop_joint_commands_device(h, g, batch, d_st, d_jp, d_fb, d_cmd, h->stream);
    if (rs != QMPC_OK) break;
    if (hipMemcpyAsync(joint_pos, d_jp, sizeof(double) * 12 * B, hipMemcpyDeviceToHost, h->stream) != hipSuccess ||
        hipMemcpyAsync(cmd, d_cmd, sizeof(qmpc_joint_command) * B, hipMemcpyDeviceToHost, h->stream) != hipSuccess ||
        (fb && hipMemcpyAsync(fb, d_fb, sizeof(qmpc_joint_feedback) * B, hipMemcpyDeviceToHost, h->stream) != hipSuccess)) { rs = QMPC_HIP_ERROR; break; }
    if (hipStreamSynchronize(h->stream) != hipSuccess) rs = QMPC_HIP_ERROR;
  } while (0);
  if (d_st) (void)hipFree(d_st);
  return rs;
}

void qmpc_loop_joint_init(double* joint_pos, int32_t batch) {
  const double stand[3] = {0.0, 0.67, -1.3};
  for (size_t t = 0; t < (size_t)(batch > 0 ? batch : 0) * 4; ++t)
    for (int j = 0; j < 3; ++j) joint_pos[3 * t + j] = stand[j];
}

// ---- device-resident closed loop (kernels in qmpc_loop.hip) ----------------
int32_t qmpc_sizeof_loop_state(void) { return (int32_t)sizeof(qmpc_loop_state); }
static_assert(sizeof(qmpc_loop_state) == 8 * 820, "qmpc_loop_state is 820 doubles");

void qmpc_default_loop_params(qmpc_loop_params* p) {
  std::memset(p, 0, sizeof *p);
  p->gait_freq = 2.2;                                               // LeggedState.h / yaml gait_freq
  const double f[4][3] = {{0.20, 0.14, -0.3}, {0.20, -0.14, -0.3}, {-0.20, 0.14, -0.3}, {-0.20, -0.14, -0.3}};
  for (int l = 0; l < 4; ++l)
    for (int a = 0; a < 3; ++a) p->default_foot_pos_rel[3 * l + a] = f[l][a];   // yaml default_foot_pos_*
  p->dt = 5.0 / 1000.0;
  p->contact_height = 1e-3;
}

void qmpc_loop_state_init(qmpc_loop_state* s, const qmpc_loop_params* lp, const double joy[6], double movement_mode,
                          double height, double yaw) {
  std::memset(s, 0, sizeof *s);
  s->pos_world[2] = height;
  s->quat[0] = std::cos(0.5 * yaw);
  s->quat[3] = std::sin(0.5 * yaw);
  double R[9], Rz[9];
  qmpc_loop::quat_to_rot(s->quat, R);
  qmpc_loop::rot_to_rot_z(R, Rz);
  for (int l = 0; l < 4; ++l)
    for (int r = 0; r < 3; ++r)
      s->foot_pos_world[3 * l + r] = Rz[3 * r] * lp->default_foot_pos_rel[3 * l] + Rz[3 * r + 1] * lp->default_foot_pos_rel[3 * l + 1] +
                                     Rz[3 * r + 2] * lp->default_foot_pos_rel[3 * l + 2] + s->pos_world[r];
  for (int a = 0; a < 6; ++a) s->joy[a] = joy[a];
  s->movement_mode = movement_mode;
  for (int a = 0; a < 3; ++a) s->pos_d_world[a] = s->pos_world[a];
  s->pos_d_init = 1.0;
  for (int a = 0; a < 4; ++a) s->quat_d[a] = s->quat[a];
  for (int l = 0; l < 4; ++l) {          // LeggedContactFSM::reset_params (:4-9) + set_default_gait_pattern
    qmpc_loop_leg& L = s->leg[l];
    L.state = 1.0;
    L.pattern_index = 0.0;
    L.prev_pattern_index = 1.0;
    L.start_time = 0.0;
    L.end_time = 0.5;
    s->contacts[l] = 1.0;
  }
}

// g != NULL: the joint-level kernel closes every tick (d_joint_pos in/out, d_cmd = the last tick's commands, d_trace_cmd
// one row per tick; either of the two may be NULL)
static qmpc_status loop_run_impl(qmpc_handle* h, const qmpc_loop_params* lp, int32_t batch, qmpc_loop_state* d_states,
                                 int32_t ticks, double* d_trace_forces, double* d_trace_contacts,
                                 const qmpc_leg_geometry* g, double* d_joint_pos, qmpc_joint_command* d_cmd,
                                 qmpc_joint_command* d_trace_cmd, void* stream) {
  if (!h || !lp || batch < 0 || ticks < 0 || (batch > 0 && !d_states)) return QMPC_BAD_ARGUMENT;
  if (g && batch > 0 && (!d_joint_pos || (!d_cmd && !d_trace_cmd))) return QMPC_BAD_ARGUMENT;
  if (h->params.model != QMPC_MODEL_QUAT && h->params.model != QMPC_MODEL_CONVEX) return QMPC_BAD_ARGUMENT;
  const bool convex = h->params.model == QMPC_MODEL_CONVEX;
  // the device tick of ConvexMpc carries the controller period as the literal 5 ms (velocity ramp, gait clock): a handle
  // with another knot spacing would silently part from the host class and the reference (ConvexMpc.cpp:9,62,208)
  if (convex && h->params.h != (float)(5.0 / 1000.0)) return QMPC_UNSUPPORTED;
  if (batch == 0 || ticks == 0) return QMPC_OK;
  if (batch > h->max_batch) return QMPC_BATCH_TOO_LARGE;
  HIP_TRY(hipSetDevice(h->device));
  hipStream_t s = stream ? (hipStream_t)stream : h->stream;
  if (!h->d_loop_row) HIP_TRY(hipMalloc(&h->d_loop_row, sizeof(int)));
  HIP_TRY(hipMemsetAsync(h->d_loop_row, 0xFF, sizeof(int), s));     // row counter = -1, stream-ordered (no host staging)
  const unsigned blocks = (unsigned)((batch + 63) / 64);
  const qmpc_loop_params LP = *lp;
  LegGeom G;
  std::memset(&G, 0, sizeof G);
  if (g) std::memcpy(&G, g, sizeof G);
  // warm start in the per-tick form: the solution travels from tick to tick through the handle's trajectory buffer (the
  // persistent kernel keeps it in LDS); the first tick of a call starts cold
  const bool warm = lp->warm_start != 0.0 && h->params.mode == QMPC_MODE_CONVERGED;
  if (warm && !h->d_traj_u)
    HIP_TRY(hipMalloc(&h->d_traj_u, sizeof(double) * 12 * (size_t)h->params.horizon * (size_t)h->max_batch));
  auto one_tick = [&](bool first) -> qmpc_status {
    if (convex)
      hipLaunchKernelGGL(qmpc_loop_front_convex_kernel, dim3(blocks), dim3(64), 0, s, LP, d_states,
                         reinterpret_cast<qmpc_convex_input*>(h->d_in), h->d_loop_row, (int)batch);
    else
      hipLaunchKernelGGL(qmpc_loop_front_kernel, dim3(blocks), dim3(64), 0, s, LP, d_states, h->d_in, h->d_loop_row, (int)batch);
    HIP_TRY(hipGetLastError());
    if (warm && use_lane(h, batch, nullptr, nullptr, true)) {
      // straggler hand-off of the warm-started ticks (not the cold first one): the records carry the rows' initial residuals
      const int wcap = (!first && h->d_handoff) ? handoff_cap(h, 3) : 0;
      const int wv = wcap ? handoff_variant(h) : 0;
      const qmpc_status st = launch_lane(h, batch, h->d_in, h->d_forces, h->d_info, s, first ? nullptr : h->d_traj_u, h->d_traj_u,
                                         /*check_prev=*/1, nullptr, wv ? wcap : 0);
      if (st != QMPC_OK) return st;
      if (wv)
        HIP_TRY(qmpc_wform_launch_list(wv, handoff_grid(h, wv), variant_lds(h, wv), s, &h->dev, sizeof h->dev, h->d_in, h->d_forces, h->d_info, h->d_traj_u,
                                       nullptr, h->d_handoff + 64, h->d_handoff, variant_gws(h, wv), h->d_hstate, h->hstate_cap));
    } else if (warm) {
      const int var = body_variant(h, batch);
      HIP_TRY(qmpc_warm_launch(var, convex ? 1 : 0, (int)batch, variant_lds(h, var), s, &h->dev, sizeof h->dev, h->d_in, first ? nullptr : h->d_traj_u,
                               h->d_forces, h->d_info, h->d_traj_u, variant_gws(h, var), /*check_prev=*/1));
    } else {
      const qmpc_status st = launch_solve(h, batch, h->d_in, h->d_forces, h->d_info, nullptr, nullptr, s, /*timed=*/false, /*handoff=*/2);
      if (st != QMPC_OK) return st;
    }
    if (convex)
      hipLaunchKernelGGL(qmpc_loop_post_kernel<true>, dim3(blocks), dim3(64), 0, s, h->dev, LP, d_states, (const double*)h->d_forces,
                         (const qmpc_info*)h->d_info, d_trace_forces, d_trace_contacts, (const int*)h->d_loop_row, (int)batch);
    else
      hipLaunchKernelGGL(qmpc_loop_post_kernel<false>, dim3(blocks), dim3(64), 0, s, h->dev, LP, d_states, (const double*)h->d_forces,
                         (const qmpc_info*)h->d_info, d_trace_forces, d_trace_contacts, (const int*)h->d_loop_row, (int)batch);
    HIP_TRY(hipGetLastError());
    if (g) {
      hipLaunchKernelGGL(qmpc_loop_joint_kernel, dim3((unsigned)(((size_t)batch * 4 + 255) / 256)), dim3(256), 0, s, G,
                         (const qmpc_loop_state*)d_states, d_joint_pos, (qmpc_joint_feedback*)nullptr, d_cmd, d_trace_cmd,
                         (const int*)h->d_loop_row, (int)batch);
      HIP_TRY(hipGetLastError());
    }
    return QMPC_OK;
  };
  // Converged mode without the joint level: ONE launch, a persistent wave per robot for all ticks (qmpc_loop_fused_kernel;
  // the per-tick tails of different robots average out instead of adding up).  QMPC_LOOP_FUSED=0 keeps the per-tick
  // launch sequence below, which is also the path of the reference mode and of the joint-level loop.
  // At most two robots per SIMD: ONE launch, a persistent wave per robot for all ticks
  // (qmpc_loop_fused_kernel: the per-tick tails of different robots average out instead of adding up; +29 % at 1024
  // robots with different commands).  Larger batches keep the per-tick sequence below (several robots per SIMD hide the
  // tails, and the fused kernel pays for its register pressure).
  // QMPC_LOOP_FUSED=0 / 1 forces one or the other (experiments, tests).
  static const int fused_env = [] { const char* e = std::getenv("QMPC_LOOP_FUSED"); return e ? (e[0] == '0' ? 0 : 1) : -1; }();
  // measured, persistent vs per-tick: +25 % (256), +28 % (1024), +8 % (2048), -3 % (4096); with the warm start, whose
  // iteration counts spread more: +61 % (1024), +33 % (2048), +8 % (4096), -14 % (16384)
  // (ConvexMpc's own solver mode: the persistent kernel exists on the wrench-form reference bodies only)
  // Round 6 (tools/r06_loop_decide.sh, profiles/r06_loop_decide.txt): the workspace-form instantiations (two waves per SIMD,
  // 256 registers, 41 ... 165 spilled VGPRs outside their inner loops) were measured against the per-tick form on every
  // configuration that selects them -- persistent +8 ... +45 % everywhere except ConvexMpc's own solver mode in the workspace
  // form (N=20, 2048 robots: 0.849 vs 0.832 ms per tick), which therefore takes the per-tick form unless forced.
  const bool conv_ref_ws = convex && h->params.mode == QMPC_MODE_REFERENCE && ref_wform_variant(h, batch) == 5;
  const bool fused = (convex && h->params.mode == QMPC_MODE_REFERENCE && !ref_wform_variant(h, batch))
                         ? false
                         : (fused_env >= 0 ? fused_env == 1 : (batch <= (warm ? 4096 : 2048) && !conv_ref_ws));
  if (fused) {
    const bool ref = h->params.mode == QMPC_MODE_REFERENCE;
    // the reference-mode kernels exist with everything in LDS (0) and with the gains in the workspace (1): launch_solve's rule
    const int rwv = ref ? ref_wform_variant(h, batch) : 0;
    const int var = ref ? (rwv ? rwv : ((batch > 1024 || h->lds_bytes > 40 * 1024 || h->variant >= 2) ? 1 : 0)) : body_variant(h, batch);
    HIP_TRY(qmpc_fused_launch(var, ref ? 1 : 0, convex ? 1 : 0, (int)batch, variant_lds(h, var), s, &h->dev, sizeof h->dev, &LP, d_states, h->d_in, h->d_forces,
                              h->d_info, d_trace_forces, d_trace_contacts, (int)ticks, variant_gws(h, var), g,
                              d_joint_pos, d_cmd, d_trace_cmd));
    return QMPC_OK;
  }
  // one tick = three kernels (four with the joint level): captured once into a graph and replayed (the sequence is launch-bound for small
  // batches); plain launches when capture is not available on this stream
  hipGraph_t graph = nullptr;
  hipGraphExec_t exec = nullptr;
  bool captured = false;
  int t_start = 0;
  // Large cold-start batches solve with the lane-per-instance kernel: its workspace is allocated and its parameter block
  // uploaded HERE, once, on the stream -- neither belongs inside the capture below (an allocation is not capturable,
  // and the parameters do not change between the ticks of a call).
  struct ResidentGuard {
    qmpc_handle* h;
    ~ResidentGuard() { h->lane_params_resident = false; h->lane_order_prev = false; h->lane_loop_cold = false; }
  } resident_guard{h};
  h->lane_loop_cold = !warm;
  if (use_lane(h, batch, nullptr, nullptr, warm)) {
    const qmpc_status es = ensure_lane_buffers(h);
    if (es != QMPC_OK) return es;
    HIP_TRY(qmpc_lane_upload_params(h->lane_pslot, s, &h->dev, sizeof h->dev));
    h->lane_params_resident = true;
    static const bool order_env = [] { const char* e = std::getenv("QMPC_LANE_ORDER_PREV"); return !e || e[0] != '0'; }();
    h->lane_order_prev = order_env;
    if (handoff_cap(h, warm ? 3 : 2)) (void)ensure_handoff_buffers(h);      // not capturable either; only where the ticks will hand over
  } else if (ref_lane_batch(h, batch, true)) {      // the reference's solver mode at Monte-Carlo scale: qmpc_lane_ref_kernel in every tick
    const qmpc_status es = ensure_lane_buffers(h);
    if (es != QMPC_OK) return es;
    HIP_TRY(qmpc_lane_upload_params(h->lane_pslot, s, &h->dev, sizeof h->dev));
    h->lane_params_resident = true;
  }
  if (warm) {                            // the cold first tick is not the tick the graph repeats
    const qmpc_status st = one_tick(true);
    if (st != QMPC_OK) return st;
    t_start = 1;
  }
  if (ticks - t_start > 1 && hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal) == hipSuccess) {
    const qmpc_status st = one_tick(false);
    const hipError_t ee = hipStreamEndCapture(s, &graph);
    if (st == QMPC_OK && ee == hipSuccess && graph && hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0) == hipSuccess)
      captured = true;
    else
      (void)hipGetLastError();
  }
  qmpc_status rs = QMPC_OK;
  for (int t = t_start; t < ticks && rs == QMPC_OK; ++t) {
    if (captured) {
      if (hipGraphLaunch(exec, s) != hipSuccess) rs = QMPC_HIP_ERROR;
    } else {
      rs = one_tick(!warm);
    }
  }
  if (exec) {
    (void)hipStreamSynchronize(s);        // the executable graph must outlive its launches
    (void)hipGraphExecDestroy(exec);
  }
  if (graph) (void)hipGraphDestroy(graph);
  return rs;
}

qmpc_status qmpc_loop_run_device(qmpc_handle* h, const qmpc_loop_params* lp, int32_t batch, qmpc_loop_state* d_states,
                                 int32_t ticks, double* d_trace_forces, double* d_trace_contacts, void* stream) {
  return loop_run_impl(h, lp, batch, d_states, ticks, d_trace_forces, d_trace_contacts, nullptr, nullptr, nullptr, nullptr,
                       stream);
}

qmpc_status qmpc_loop_run_joint_device(qmpc_handle* h, const qmpc_loop_params* lp, const qmpc_leg_geometry* g, int32_t batch,
                                       qmpc_loop_state* d_states, double* d_joint_pos, int32_t ticks,
                                       qmpc_joint_command* d_cmd, qmpc_joint_command* d_trace_cmd, void* stream) {
  if (!g) return QMPC_BAD_ARGUMENT;
  return loop_run_impl(h, lp, batch, d_states, ticks, nullptr, nullptr, g, d_joint_pos, d_cmd, d_trace_cmd, stream);
}

qmpc_status qmpc_loop_run(qmpc_handle* h, const qmpc_loop_params* lp, int32_t batch, qmpc_loop_state* states,
                          int32_t ticks, double* trace_forces, double* trace_contacts) {
  if (!h || !lp || batch < 0 || ticks < 0 || (batch > 0 && !states)) return QMPC_BAD_ARGUMENT;
  if (h->params.model != QMPC_MODEL_QUAT && h->params.model != QMPC_MODEL_CONVEX) return QMPC_BAD_ARGUMENT;
  if (batch == 0 || ticks == 0) return QMPC_OK;
  if (batch > h->max_batch) return QMPC_BATCH_TOO_LARGE;
  HIP_TRY(hipSetDevice(h->device));
  // staging that belongs to the handle and only grows: [states | force trace | contact trace]
  const size_t B = (size_t)batch, T = (size_t)ticks;
  const size_t n_st = (sizeof(qmpc_loop_state) / sizeof(double)) * B, n_tf = trace_forces ? 12 * B * T : 0,
               n_tc = trace_contacts ? 4 * B * T : 0;
  if (h->loop_cap < n_st + n_tf + n_tc) {
    if (h->d_loop) (void)hipFree(h->d_loop);
    h->d_loop = nullptr;
    h->loop_cap = 0;
    HIP_TRY(hipMalloc(&h->d_loop, sizeof(double) * (n_st + n_tf + n_tc)));
    h->loop_cap = n_st + n_tf + n_tc;
  }
  qmpc_loop_state* d_st = reinterpret_cast<qmpc_loop_state*>(h->d_loop);
  double* d_tf = trace_forces ? h->d_loop + n_st : nullptr;
  double* d_tc = trace_contacts ? h->d_loop + n_st + n_tf : nullptr;
  qmpc_status rs = QMPC_OK;
  do {
    if (hipMemcpyAsync(d_st, states, sizeof(qmpc_loop_state) * B, hipMemcpyHostToDevice, h->stream) != hipSuccess) { rs = QMPC_HIP_ERROR; break; }
    rs = qmpc_loop_run_device(h, lp, batch, d_st, ticks, d_tf, d_tc, nullptr);
    if (rs != QMPC_OK) break;
    if (hipMemcpyAsync(states, d_st, sizeof(qmpc_loop_state) * B, hipMemcpyDeviceToHost, h->stream) != hipSuccess) { rs = QMPC_HIP_ERROR; break; }
    if (d_tf && hipMemcpyAsync(trace_forces, d_tf, sizeof(double) * 12 * B * T, hipMemcpyDeviceToHost, h->stream) != hipSuccess) { rs = QMPC_HIP_ERROR; break; }
    if (d_tc && hipMemcpyAsync(trace_contacts, d_tc, sizeof(double) * 4 * B * T, hipMemcpyDeviceToHost, h->stream) != hipSuccess) { rs = QMPC_HIP_ERROR; break; }
    if (hipStreamSynchronize(h->stream) != hipSuccess) rs = QMPC_HIP_ERROR;
  } while (0);
  return rs;
}

// Diagnostic: per-instance phase cycle counts (s_memtime) of one solve launch.
// cycles_out: [batch][16] int64 on the host; slots 0..14 follow the PH_* enum of qmpc_kernels.hip (setup, expansions,
// operand build, MFMA + stage terms, stage solve, cost-to-go update, directions, rollout, misc, rotation pre-pass,
// rollout gain / broadcast / step, apply, MFMA drain); slot 15 = iterations.
qmpc_status qmpc_debug_profile(qmpc_handle* h, int32_t batch, const qmpc_input* in, int64_t* cycles_out) {
  if (!h || batch < 1 || !in || !cycles_out) return QMPC_BAD_ARGUMENT;
  if (h->params.model != QMPC_MODEL_QUAT || h->params.mode != QMPC_MODE_CONVERGED) return QMPC_BAD_ARGUMENT;
  if (batch > h->max_batch) return QMPC_BATCH_TOO_LARGE;
  HIP_TRY(hipSetDevice(h->device));
  long long* d_prof = nullptr;
  HIP_TRY(hipMalloc(&d_prof, sizeof(long long) * 16 * (size_t)batch));
  HIP_TRY(hipMemsetAsync(d_prof, 0, sizeof(long long) * 16 * (size_t)batch, h->stream));
  HIP_TRY(hipMemcpyAsync(h->d_in, in, sizeof(qmpc_input) * (size_t)batch, hipMemcpyHostToDevice, h->stream));
  if (const int wv = wform_variant(h, batch))
    HIP_TRY(qmpc_wform_launch(wv, 1, (int)batch, variant_lds(h, wv), h->stream, &h->dev, sizeof h->dev, h->d_in, h->d_forces, h->d_info,
                              nullptr, nullptr, d_prof, variant_gws(h, wv)));
  else if (use_global_gains(h, batch))
    hipLaunchKernelGGL((qmpc_solve_kernel<QuatModel, true, 1>), dim3((unsigned)batch), dim3(kWave), h->lds_bytes_g, h->stream,
                       h->dev, h->d_in, h->d_forces, h->d_info, (double*)nullptr, (double*)nullptr, (int)batch, d_prof,
                       h->d_gws);
  else
    hipLaunchKernelGGL((qmpc_solve_kernel<QuatModel, true, 0>), dim3((unsigned)batch), dim3(kWave), h->lds_bytes, h->stream,
                       h->dev, h->d_in, h->d_forces, h->d_info, (double*)nullptr, (double*)nullptr, (int)batch, d_prof,
                       (double*)nullptr);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipMemcpyAsync(cycles_out, d_prof, sizeof(long long) * 16 * (size_t)batch, hipMemcpyDeviceToHost, h->stream));
  HIP_TRY(hipStreamSynchronize(h->stream));
  HIP_TRY(hipFree(d_prof));
  return QMPC_OK;
}

// Diagnostic (not part of the drop-in surface): C = X' * Y on [12][16] tiles via
// the FP64 MFMA path; host buffers of 192 doubles each.
qmpc_status qmpc_selftest_mtm(int32_t device, const double* X, const double* Y, double* Cout) {
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1 || device >= ndev) return QMPC_NO_DEVICE;
  HIP_TRY(hipSetDevice(device));
  double* d = nullptr;
  HIP_TRY(hipMalloc(&d, sizeof(double) * 3 * MAT));
  HIP_TRY(hipMemcpy(d, X, sizeof(double) * MAT, hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(d + MAT, Y, sizeof(double) * MAT, hipMemcpyHostToDevice));
  hipLaunchKernelGGL(qmpc_selftest_kernel, dim3(1), dim3(kWave), 0, 0, d, d + MAT, d + 2 * MAT);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipMemcpy(Cout, d + 2 * MAT, sizeof(double) * MAT, hipMemcpyDeviceToHost));
  HIP_TRY(hipFree(d));
  return QMPC_OK;
}

// Diagnostic: cross-lane primitives on 64 doubles; out holds 9 x 64 doubles
// (row-group broadcasts 0..3, wave sum/max/min, row_newbcast:5, quad_perm[1,1,1,1]).
qmpc_status qmpc_selftest_lanes(int32_t device, const double* in, double* out) {
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1 || device >= ndev) return QMPC_NO_DEVICE;
  HIP_TRY(hipSetDevice(device));
  double* d = nullptr;
  HIP_TRY(hipMalloc(&d, sizeof(double) * 64 * 10));
  HIP_TRY(hipMemcpy(d, in, sizeof(double) * 64, hipMemcpyHostToDevice));
  hipLaunchKernelGGL(qmpc_selftest_lanes_kernel, dim3(1), dim3(kWave), 0, 0, d, d + 64);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipMemcpy(out, d + 64, sizeof(double) * 64 * 9, hipMemcpyDeviceToHost));
  HIP_TRY(hipFree(d));
  return QMPC_OK;
}

}  // extern "C"
