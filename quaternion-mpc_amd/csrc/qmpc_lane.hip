// qmpc_lane.hip -- third translation unit of libqmpc_hip.so: the lane-per-instance solve kernel for large batches
// (qmpc_lane_core.h has the algorithm and the reference citations) and its launcher.
//
// Mapping.  A 64-lane workgroup (one wavefront) is persistent: it owns one block of the HBM workspace for the whole
// launch and walks the batch in strides of the resident lane count, so the workspace is sized by the number of RESIDENT
// lanes (at most 1024 wavefronts x 64), not by the batch.  A block is laid out [element][lane]: every wave-level load /
// store is one contiguous 512-byte row, consecutive elements are consecutive rows (one base register, immediate
// offsets), and a wave streams through one contiguous region (DRAM pages, TLB).  The cost-to-go matrix of the backward
// pass lives in LDS (78 rows of 512 bytes per wave: 4 waves per CU use 156 of the 160 KB), the per-instance constants in
// registers.  There is no cross-lane operation anywhere in the solve; lanes whose instance has converged idle until the
// slowest instance of the wave is done.
//
// Contact patterns.  The per-contact-point code is guarded by a per-lane stance test; the hardware skips a guarded
// region when no lane of the wave needs it.  A counting sort on the stance mask (qmpc_lane_sort_*) orders the batch so
// that a wave's 64 instances share their pattern (trot pairs do half the per-point work of a four-stance instance); in
// the closed loop the order within a pattern follows every robot's iteration count at its last tick.
//
// Calls served: qmpc_solve* / qmpc_solve8* / qmpc_convex_solve* (with or without trajectory outputs), qmpc_solve_warm* and
// the per-tick form of the device-resident closed loop (cold or warm-started), converged mode, from the switch-over batch
// size of qmpc_hip.hip on.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "qmpc_lane_core.h"

// The file is compiled as TWO units of the library (__graft_entry__.py): QL_UNIT 1 -- the converged mode's kernel, the sort
// kernels and the launcher (qmpc_lane.hip itself, scheduling strategy max-ilp: +3 % for the pair forms, nothing for the plain
// ones) -- and QL_UNIT 2 -- the reference mode's kernel with its own launcher (qmpc_lane_ref.hip includes this file; the default
// strategy: max-ilp costs that kernel 5-8 %).  QL_UNIT 0 (tools/lane_variants.py): everything in one unit.
#ifndef QL_UNIT
#define QL_UNIT 0
#endif
namespace qmpc {
namespace lane {

constexpr int kLaneWave = 64;
constexpr int kParamSlots = 64;

// Parameter blocks live in constant memory, one slot per handle: every pass reads what it needs with scalar loads and
// nothing of it occupies registers across the passes.  (static: the file is compiled as two units, each with its own table)
static __constant__ DevParams ql_params[kParamSlots];


// The three passes (and set-up / outputs) are compiled as SEPARATE functions: each gets the register file to itself --
// inlined into one kernel body the allocator spilled several hundred registers around the backward pass.  The
// instance's constants and scalars are handed over through the lane's private memory (read once per pass); wave-uniform
// arguments arrive in vector registers by the calling convention and are made scalar again with v_readfirstlane.
// struct <-> private memory, word by word (the implicit copy operations of a class do not take address-space pointers)
template <class T>
__device__ __forceinline__ void priv_load(T& dst, QL_PRIV_AS const T* src) {
  static_assert(sizeof(T) % 4 == 0, "dword copy");
  QL_PRIV_AS const unsigned* s = (QL_PRIV_AS const unsigned*)src;
  unsigned* d = reinterpret_cast<unsigned*>(&dst);
#pragma unroll
  for (unsigned i = 0; i < sizeof(T) / 4; ++i) d[i] = s[i];
}
template <class T>
__device__ __forceinline__ void priv_store(QL_PRIV_AS T* dst, const T& src) {
  static_assert(sizeof(T) % 4 == 0, "dword copy");
  QL_PRIV_AS unsigned* d = (QL_PRIV_AS unsigned*)dst;
  const unsigned* s = reinterpret_cast<const unsigned*>(&src);
#pragma unroll
  for (unsigned i = 0; i < sizeof(T) / 4; ++i) d[i] = s[i];
}

struct PassArgs {
  int pslot;
  unsigned ws_lo, ws_hi;     // this wave's workspace block
  unsigned lane8;            // 8 x lane (pair mode: 8 x (lane & 31) -- the partner lanes share a column)
  unsigned warm;             // wave-uniform: the launch carries previous solutions (qmpc_solve_warm*, the warm-started loop)
  unsigned half;             // pair mode: 1 on lanes 32..63 (Ctx::half); 0 otherwise
  unsigned lmask;            // wave-uniform: Ctx::lmask (0x1F8, pair mode 0xF8)
};
template <int NL>
__device__ __forceinline__ Ctx pass_ctx(const PassArgs& a) {
  extern __shared__ __attribute__((aligned(16))) double ql_lds[];
  const unsigned lo = __builtin_amdgcn_readfirstlane(a.ws_lo), hi = __builtin_amdgcn_readfirstlane(a.ws_hi);
  QL_GLOBAL_AS double* ws = reinterpret_cast<QL_GLOBAL_AS double*>(((unsigned long long)hi << 32) | lo);
  Ctx c = {ws, 8u * kLaneWave, a.lane8, (QL_LDS_AS double*)ql_lds, 8u * kLaneWave, a.lane8, a.half,
           (unsigned)__builtin_amdgcn_readfirstlane(a.lmask)};
  return c;
}
template <int NL, int MD>
__device__ __noinline__ void call_setup(PassArgs a, unsigned long long rec, unsigned long long u_prev, QL_PRIV_AS LaneK<NL>* Kp,
                                        QL_PRIV_AS LaneState* sp) {
  const DevParams& P = ql_params[__builtin_amdgcn_readfirstlane(a.pslot)];
  const Ctx c = pass_ctx<NL>(a);
  const WsOff O = make_wsoff<NL>(P.N);
  LaneK<NL> K;
  LaneState st;
  lane_setup<NL, MD>(P, c, O, reinterpret_cast<const double*>(rec), K, st, __builtin_amdgcn_readfirstlane(a.warm) != 0,
                 reinterpret_cast<const double*>(u_prev));
  priv_store(Kp, K);
  priv_store(sp, st);
}
template <int NL, bool WARM, int MD, bool PAIR = false>
__device__ __noinline__ void call_A(PassArgs a, QL_PRIV_AS const LaneK<NL>* Kp, QL_PRIV_AS LaneState* sp) {
  const DevParams& P = ql_params[__builtin_amdgcn_readfirstlane(a.pslot)];
  const Ctx c = pass_ctx<NL>(a);
  const WsOff O = make_wsoff<NL>(P.N);
  LaneK<NL> K;
  priv_load(K, Kp);
  LaneState st;
  priv_load(st, (QL_PRIV_AS const LaneState*)sp);
  st.it += 1;
  pass_A<NL, WARM, MD, PAIR>(P, c, O, K, st, st.it == 1, (FootPtr)Kp->foot);
  priv_store(sp, st);
}
template <int NL, bool WARM, int MD, bool PAIR = false>
__device__ __noinline__ bool call_B(PassArgs a, QL_PRIV_AS const LaneK<NL>* Kp, QL_PRIV_AS LaneState* sp) {
  const DevParams& P = ql_params[__builtin_amdgcn_readfirstlane(a.pslot)];
  const Ctx c = pass_ctx<NL>(a);
  const WsOff O = make_wsoff<NL>(P.N);
  LaneK<NL> K;
  priv_load(K, Kp);
  LaneState st;
  priv_load(st, (QL_PRIV_AS const LaneState*)sp);
  const bool ok = pass_B<NL, WARM, MD, false, PAIR>(P, c, O, K, st, (FootPtr)Kp->foot);
#if defined(QL_PROFILE)
  priv_store(sp, st);
#endif
  return ok;
}
template <int NL, bool WARM, int MD, bool PAIR = false>
__device__ __noinline__ void call_C(PassArgs a, QL_PRIV_AS const LaneK<NL>* Kp, QL_PRIV_AS LaneState* sp) {
  const DevParams& P = ql_params[__builtin_amdgcn_readfirstlane(a.pslot)];
  const Ctx c = pass_ctx<NL>(a);
  const WsOff O = make_wsoff<NL>(P.N);
  LaneK<NL> K;
  priv_load(K, Kp);
  LaneState st;
  priv_load(st, (QL_PRIV_AS const LaneState*)sp);
  pass_C<NL, WARM, MD, PAIR>(P, c, O, K, st, (FootPtr)Kp->foot);
  if (!st.bad_step) st.iters = st.it;
  priv_store(sp, st);
}
template <int NL, int MD>
__device__ __noinline__ void call_finish(PassArgs a, QL_PRIV_AS const LaneK<NL>* Kp, QL_PRIV_AS const LaneState* sp,
                                         unsigned long long forces, unsigned long long info, unsigned long long traj_u,
                                         unsigned long long traj_x) {
  const DevParams& P = ql_params[__builtin_amdgcn_readfirstlane(a.pslot)];
  const Ctx c = pass_ctx<NL>(a);
  const WsOff O = make_wsoff<NL>(P.N);
  LaneK<NL> K;
  priv_load(K, Kp);
  LaneState st;
  priv_load(st, sp);
  lane_finish<NL, MD>(P, c, O, K, st, reinterpret_cast<double*>(forces), reinterpret_cast<qmpc_info*>(info),
                  reinterpret_cast<double*>(traj_u), reinterpret_cast<double*>(traj_x));
}

// ---- reference mode (qmpc_lane_core.h: "Reference mode on the lane passes"): the passes as separate functions, the
// augmented-Lagrangian scalars of the lane handed over through its private memory like the other per-instance state ---------
template <int NL>
__device__ __noinline__ void call_setup_ref(PassArgs a, QL_PRIV_AS const LaneState* sp, QL_PRIV_AS LaneAL* alp) {
  const DevParams& P = ql_params[__builtin_amdgcn_readfirstlane(a.pslot)];
  const Ctx c = pass_ctx<NL>(a);
  const WsOff O = make_wsoff<NL>(P.N);
  LaneState st;
  priv_load(st, sp);
  LaneAL al;
  lane_setup_ref<NL>(P, c, O, st, al);
  priv_store(alp, al);
}
template <int NL, bool UPDATE, int MD>
__device__ __noinline__ void call_M(PassArgs a, QL_PRIV_AS const LaneK<NL>* Kp, QL_PRIV_AS const LaneState* sp, QL_PRIV_AS LaneAL* alp) {
  const DevParams& P = ql_params[__builtin_amdgcn_readfirstlane(a.pslot)];
  const Ctx c = pass_ctx<NL>(a);
  const WsOff O = make_wsoff<NL>(P.N);
  LaneK<NL> K;
  priv_load(K, Kp);
  LaneState st;
  priv_load(st, sp);
  LaneAL al;
  priv_load(al, (QL_PRIV_AS const LaneAL*)alp);
  pass_M<NL, UPDATE, MD>(P, c, O, K, st, al);
  priv_store(alp, al);
}
template <int NL, int MD, bool PAIR = false>
__device__ __noinline__ bool call_B_AL(PassArgs a, QL_PRIV_AS const LaneK<NL>* Kp, QL_PRIV_AS const LaneState* sp, QL_PRIV_AS LaneAL* alp) {
  const DevParams& P = ql_params[__builtin_amdgcn_readfirstlane(a.pslot)];
  const Ctx c = pass_ctx<NL>(a);
  const WsOff O = make_wsoff<NL>(P.N);
  LaneK<NL> K;
  priv_load(K, Kp);
  LaneState st;
  priv_load(st, sp);
  LaneAL al;
  priv_load(al, (QL_PRIV_AS const LaneAL*)alp);
  const bool ok = pass_B<NL, false, MD, true, PAIR>(P, c, O, K, st, (FootPtr)Kp->foot, &al);
  priv_store(alp, al);
  return ok;
}
template <int NL, int MD, bool PAIR = false>
__device__ __noinline__ void call_C_AL(PassArgs a, QL_PRIV_AS const LaneK<NL>* Kp, QL_PRIV_AS const LaneState* sp, QL_PRIV_AS LaneAL* alp) {
  const DevParams& P = ql_params[__builtin_amdgcn_readfirstlane(a.pslot)];
  const Ctx c = pass_ctx<NL>(a);
  const WsOff O = make_wsoff<NL>(P.N);
  LaneK<NL> K;
  priv_load(K, Kp);
  LaneState st;
  priv_load(st, sp);
  LaneAL al;
  priv_load(al, (QL_PRIV_AS const LaneAL*)alp);
  pass_C_AL<NL, 2, MD, PAIR>(P, c, O, K, st, al, true);
  priv_store(alp, al);
}
template <int NL, int MD>
__device__ __noinline__ void call_A_AL(PassArgs a, QL_PRIV_AS const LaneK<NL>* Kp, QL_PRIV_AS const LaneState* sp, int sel) {
  const DevParams& P = ql_params[__builtin_amdgcn_readfirstlane(a.pslot)];
  const Ctx c = pass_ctx<NL>(a);
  const WsOff O = make_wsoff<NL>(P.N);
  LaneK<NL> K;
  priv_load(K, Kp);
  LaneState st;
  priv_load(st, sp);
  pass_A_AL<NL, MD>(P, c, O, K, st, sel);
}
template <int NL, int MD>
__device__ __noinline__ void call_S(PassArgs a, QL_PRIV_AS const LaneK<NL>* Kp, QL_PRIV_AS const LaneState* sp, QL_PRIV_AS LaneAL* alp) {
  const DevParams& P = ql_params[__builtin_amdgcn_readfirstlane(a.pslot)];
  const Ctx c = pass_ctx<NL>(a);
  const WsOff O = make_wsoff<NL>(P.N);
  LaneK<NL> K;
  priv_load(K, Kp);
  LaneState st;
  priv_load(st, sp);
  LaneAL al;
  priv_load(al, (QL_PRIV_AS const LaneAL*)alp);
  pass_S<NL, MD>(P, c, O, K, st, al);
  priv_store(alp, al);
}

// The reference's own solver mode, one lane per instance: the steps of lane_solve_ref (qmpc_lane_core.h) in lock step.  A
// lane whose line search has ended waits (masked off) while others of its wavefront try shorter steps.
template <int NL, int MD = MD_QUAT>
__global__ __launch_bounds__(kLaneWave) void qmpc_lane_ref_kernel(int pslot, const double* __restrict__ in, double* __restrict__ forces,
                                                                  qmpc_info* __restrict__ info, int batch, double* __restrict__ ws,
                                                                  unsigned slots, int lanes, const int* __restrict__ perm,
                                                                  double* traj_u, double* traj_x, long long* __restrict__ prof) {
  typedef LDim<NL> D;
  const int lane = threadIdx.x;
  const DevParams& P = ql_params[pslot];
#if defined(QL_PROFILE)
  long long tp[6] = {0, 0, 0, 0, 0, 0}, tl = clock64();      // diagnostic build: cycles in B / C / A / S / M / rest (lane 0's clock)
#define QL_RTICK(i) do { const long long n_ = clock64(); tp[i] += n_ - tl; tl = n_; } while (0)
#else
#define QL_RTICK(i) do { } while (0)
#endif
  const size_t block_elems = (size_t)make_wsoff<NL>(P.N, true).total * kLaneWave;      // wide layout: the second gain block
  const unsigned long long wsb = reinterpret_cast<unsigned long long>(ws + (size_t)blockIdx.x * block_elems);
  // Lane pairs (four-point quaternion model, batches that fill half of every wavefront; round 6): lanes i and i + 32 take the SAME
  // instance, every pass duplicated on the partner except the per-point blocks of the trial sweeps, which the pair splits
  // (pass_C_AL<..., PAIR>).  The launcher passes lanes = -34; QMPC_LANE_PAIR=0 restores the masked half.
  constexpr bool kPairable = NL == 4 && MD == MD_QUAT;
  const bool pairm = kPairable && lanes < 0;
  if (lanes < 0) lanes = 32;
  const int lane_i = pairm ? (lane & 31) : lane;
  const PassArgs a = {pslot, (unsigned)wsb, (unsigned)(wsb >> 32), 8u * (unsigned)lane_i, 0u, pairm ? (unsigned)(lane >> 5) : 0u,
                      pairm ? 0xF8u : 0x1F8u};
  const size_t tstride = (size_t)P.N * D::NU;
  LaneK<NL> K;
  LaneState st;
  LaneAL al;
  QL_PRIV_AS LaneK<NL>* Kp = (QL_PRIV_AS LaneK<NL>*)&K;
  QL_PRIV_AS LaneState* sp = (QL_PRIV_AS LaneState*)&st;
  QL_PRIV_AS LaneAL* alp = (QL_PRIV_AS LaneAL*)&al;
  for (long long base = (long long)blockIdx.x * lanes; base < batch; base += slots) {
    const long long pos = base + lane_i;
    const bool valid = (pairm || lane < lanes) && pos < batch;
    const int b = valid ? (perm ? perm[pos] : (int)pos) : 0;
    bool active = false;
    if (valid) {
      call_setup<NL, MD>(a, reinterpret_cast<unsigned long long>(in + (size_t)b * D::REC), 0ull, Kp, sp);
      active = st.active;
    }
    int iter = 0;
    if (active) {
      call_setup_ref<NL>(a, sp, alp);
      call_A<NL, false, MD>(a, Kp, sp);        // X <- rollout of U = u_ref (first iteration of the apply pass)
      call_M<NL, false, MD>(a, Kp, sp, alp);
      st.status = QMPC_MAX_ITER;
      st.last_step = 0.0;
    }
    while (__any(active)) {
      bool searching = false;
      QL_RTICK(5);
      if (active) {
        ++iter;
        if (!((kPairable && pairm) ? call_B_AL<NL, MD, kPairable>(a, Kp, sp, alp) : call_B_AL<NL, MD>(a, Kp, sp, alp))) { st.status = QMPC_NOT_PD; --iter; active = false; }
        else { al.alpha = 1.0; searching = true; }
      }
      QL_RTICK(0);
      bool accepted = false;
      int ls = 0;
      while (__any(searching)) {
        if (searching) {
          // trials ls and ls + 1 (step lengths alpha, alpha / 2) in one sweep
          if (kPairable && pairm) call_C_AL<NL, MD, kPairable>(a, Kp, sp, alp); else call_C_AL<NL, MD>(a, Kp, sp, alp);
          if (al_accept_pair(P, al, ls)) { accepted = true; searching = false; }
          else {
            ls += 2;
            if (ls > P.linesearch_max) searching = false;
          }
        }
      }
      QL_RTICK(1);
      if (active && !accepted) { st.status = QMPC_LINESEARCH_FAIL; --iter; active = false; }
      if (active) {
        call_A_AL<NL, MD>(a, Kp, sp, al.sel);
        QL_RTICK(2);
        st.last_step = al.stp;
        const double dJ = al.J - al.Jn;
        al.J = al.Jn; al.Jp = al.Jnp; al.viol = al.vn;
        call_S<NL, MD>(a, Kp, sp, alp);
        QL_RTICK(3);
        if (al.stat < P.tol_stat && al.viol < P.tol_feas) { st.status = QMPC_OK; active = false; }
        else {
          if (al.stat < P.tol_stat || fabs(dJ) < P.tol_cost_int) call_M<NL, true, MD>(a, Kp, sp, alp);
          if (iter >= P.iterations_max) active = false;
        }
        QL_RTICK(4);
      }
    }
#if defined(QL_PROFILE)
    if (prof && base < (long long)slots && lane == 0)
      for (int i = 0; i < 6; ++i) prof[16 * blockIdx.x + i] = tp[i];
#endif
    if (valid) {
      st.iters = iter;
      if (st.active) st.mu = al.rho;       // the info record's last field is the penalty in this mode
      call_finish<NL, MD>(a, Kp, sp, reinterpret_cast<unsigned long long>(forces + (size_t)b * D::NU),
                               info ? reinterpret_cast<unsigned long long>(info + b) : 0ull,
                               traj_u ? reinterpret_cast<unsigned long long>(traj_u + (size_t)b * tstride) : 0ull,
                               traj_x ? reinterpret_cast<unsigned long long>(traj_x + (size_t)b * (size_t)(P.N + 1) * (MD == MD_CONVEX ? 12 : 13)) : 0ull);
    }
  }
}

// Straggler hand-off (four-point QuatMpc, cold launches): the state of an instance that reached the iteration cap, for
// the wave-per-instance kernel to CONTINUE from (qmpc_wform_body.inc) -- one record of 8 + 84 N doubles:
// rho, last alpha_p, last alpha_d, last full step, iterations done, 3 spare; U [N][12]; slacks [N][24] (the Tapia flag in
// the sign); multipliers [N][24]; initial slack residuals [N][24] (warm-started launches; flag in slot 5).  The state is that of the top of the next iteration (the step applied, the barrier
// parameter not yet evaluated), which is where the wave kernel's loop begins.
template <int NL>
__device__ __noinline__ void call_dump(PassArgs a, QL_PRIV_AS const LaneState* sp, unsigned long long out, int warm) {
  const DevParams& P = ql_params[__builtin_amdgcn_readfirstlane(a.pslot)];
  const Ctx c = pass_ctx<NL>(a);
  const WsOff O = make_wsoff<NL>(P.N);
  LaneState st;
  priv_load(st, sp);
  double* o = reinterpret_cast<double*>(out);
  const int N = P.N;
  o[0] = st.rho; o[1] = st.last_ap; o[2] = st.last_ad; o[3] = st.last_step; o[4] = (double)st.iters; o[5] = warm ? 1.0 : 0.0; o[6] = 0.0; o[7] = 0.0;
  for (int i = 0; i < 3 * NL * N; ++i) o[8 + i] = c.W(O.U + i);
  for (int i = 0; i < 6 * NL * N; ++i) {
    const bool on = (st.con >> ((i % (6 * NL)) / 6)) & 1u;
    o[8 + 3 * NL * N + i] = on ? c.W(O.S + i) : 1.0;
    o[8 + 9 * NL * N + i] = on ? c.W(O.LAM + i) : 0.0;
  }
  // a warm-started launch carries the rows' initial slack residuals per knot (rc_i = rho rc0_i): they travel too
  if (warm)
    for (int i = 0; i < 6 * NL * N; ++i) o[8 + 15 * NL * N + i] = c.W(O.RC + i);
}

template <int NL, int MD = MD_QUAT>
__global__ __launch_bounds__(kLaneWave) void qmpc_lane_kernel(int pslot, const double* __restrict__ in,
                                                              double* __restrict__ forces, qmpc_info* __restrict__ info,
                                                              int batch, double* __restrict__ ws, unsigned slots,
                                                              int lanes, const int* __restrict__ perm,
                                                              long long* __restrict__ prof, const double* u_init, double* traj_u,
                                                              int check_prev, double* traj_x, int iter_cap,
                                                              int* __restrict__ hcount, int* __restrict__ hsel,
                                                              double* __restrict__ hstate, int hcap) {
  typedef LDim<NL> D;
  const int lane = threadIdx.x;
  const DevParams& P = ql_params[pslot];
  // iter_cap > 0: instances that have not converged after iter_cap iterations stop as QMPC_MAX_ITER with that count --
  // the launcher hands them to the wave-per-instance kernel (straggler hand-off, qmpc_hip.hip)
  const int itmax = (iter_cap > 0 && iter_cap < P.iterations_max) ? iter_cap : P.iterations_max;
  const size_t block_elems = (size_t)make_wsoff<NL>(P.N).total * kLaneWave;
#if defined(QL_DIAG_ALIAS)
  // diagnostic builds only (tools/lane_variants.py, profiles/r06_lane_traffic_bound.txt): every wavefront of an XCD works in
  // ONE workspace block, so every access hits that XCD's L2 -- the results are garbage, the instruction stream is not
  // (QL_DIAG_ROUNDS fixes the control flow)
  const unsigned long long wsb = reinterpret_cast<unsigned long long>(ws + (size_t)(blockIdx.x % QL_DIAG_ALIAS) * block_elems);
#else
  const unsigned long long wsb = reinterpret_cast<unsigned long long>(ws + (size_t)blockIdx.x * block_elems);
#endif
  // Lane pairs (four-point quaternion model, batches that fill half of every wavefront): lanes i and i + 32 take the SAME instance;
  // every pass runs duplicated on the partner -- a wave64 FP64 instruction issues its four passes whatever the mask -- except the
  // per-point blocks of the trial pass, which the pair splits (pass_C<..., PAIR>).  QMPC_LANE_PAIR=0 restores the masked half.
  constexpr bool kPairable = NL == 4 && MD == MD_QUAT;
  const bool pairm = kPairable && lanes <= -32;      // (the launcher passes -34 for pair mode; -33: the trial pass of cold rounds only; -32: cold rounds only)
  const bool pair_b = lanes == -32 || lanes == -34;
  const bool pair_w = lanes == -34;
  if (lanes < 0) lanes = 32;
  const int lane_i = pairm ? (lane & 31) : lane;
  const PassArgs a = {pslot, (unsigned)wsb, (unsigned)(wsb >> 32), 8u * (unsigned)lane_i, u_init ? 1u : 0u,
                      pairm ? (unsigned)(lane >> 5) : 0u, pairm ? 0xF8u : 0x1F8u};
  const bool warm = u_init != nullptr;      // kernel argument: scalar
  const size_t tstride = (size_t)P.N * D::NU;      // doubles per instance in u_init / traj_u
  LaneK<NL> K;
  LaneState st;
  QL_PRIV_AS LaneK<NL>* Kp = (QL_PRIV_AS LaneK<NL>*)&K;
  QL_PRIV_AS LaneState* sp = (QL_PRIV_AS LaneState*)&st;
  // `lanes` (64, or 32 when the batch would otherwise leave SIMDs without a wavefront) lanes of a wave take instances
  for (long long base = (long long)blockIdx.x * lanes; base < batch; base += slots) {
    const long long pos = base + lane_i;
    const bool valid = (pairm || lane < lanes) && pos < batch;
    const int b = valid ? (perm ? perm[pos] : (int)pos) : 0;
    bool active = false;
    if (valid) {
      // the previous solution of this instance is usable unless its last solve failed (check_prev: info[b] still holds
      // that solve's record -- the rule of qmpc_solve_warm_kernel)
      const bool usable = u_init && (!check_prev || info[b].status == QMPC_OK || info[b].status == QMPC_MAX_ITER);
      call_setup<NL, MD>(a, reinterpret_cast<unsigned long long>(in + (size_t)b * D::REC),
                     usable ? reinterpret_cast<unsigned long long>(u_init + (size_t)b * tstride) : 0ull, Kp, sp);
      active = st.active;
    }
#if defined(QL_PROFILE)
    for (int i = 0; i < LP_COUNT; ++i) st.t[i] = 0;
    st.last = clock64();
    int rounds = 0;
#endif
#if defined(QL_DIAG_ROUNDS)
    // diagnostic builds only: a FIXED number of rounds whatever the iterates do (no convergence test, no failure exit), so
    // that variants whose data are garbage (aliased workspace, elided stores) run the instruction stream of a real launch
    for (int round = 0; round < QL_DIAG_ROUNDS; ++round) {
      if (active) {
        if (warm && __any(st.rho != 0.0)) {
          if (kPairable && pairm && pair_b && pair_w) call_A<NL, true, MD, kPairable>(a, Kp, sp); else call_A<NL, true, MD>(a, Kp, sp);
        } else if (kPairable && pairm && pair_b) call_A<NL, false, MD, kPairable>(a, Kp, sp);
        else call_A<NL, false, MD>(a, Kp, sp);
        double sg = P.sigma;
        const double amin = fmin(st.last_ap, st.last_ad);
        if (st.it > 1 && amin >= 0.99) sg = P.sigma_fast;
        else if (st.it > 1 && amin < 0.2) sg = fmax(sg, 0.8);
        else if (st.it > 1 && amin < 0.5) sg = fmax(sg, 0.5);
        st.target = sg * st.mu;
        const bool wrows = warm && __any(st.rho != 0.0);
        const bool pb = kPairable && pairm && pair_b;
        if (wrows) { if (pb && pair_w) call_B<NL, true, MD, kPairable>(a, Kp, sp); else call_B<NL, true, MD>(a, Kp, sp); }
        else if (pb) call_B<NL, false, MD, kPairable>(a, Kp, sp);
        else call_B<NL, false, MD>(a, Kp, sp);
        if (wrows) { if (kPairable && pairm && pair_w) call_C<NL, true, MD, kPairable>(a, Kp, sp); else call_C<NL, true, MD>(a, Kp, sp); }
        else if (kPairable && pairm) call_C<NL, false, MD, kPairable>(a, Kp, sp);
        else call_C<NL, false, MD>(a, Kp, sp);
      }
    }
    if (active) { st.status = QMPC_OK; active = false; }
#endif
    while (__any(active)) {
      if (active) {
        // one interior-point iteration (the control flow of lane_iteration in qmpc_lane_core.h)
        // The warm instantiations of the passes (per-row initial residuals travelling with the rows) are needed only while
        // some lane still carries a slack residual: rho is exactly 0 after a lane's first full step, and from then on the
        // cold passes compute the same thing with fewer registers (their rc0 is multiplied by rho = 0).
        // (the apply pass is split across the lane pair wherever the backward pass is: QMPC_LANE_PAIR)
        if (warm && __any(st.rho != 0.0)) {
          if (kPairable && pairm && pair_b && pair_w) call_A<NL, true, MD, kPairable>(a, Kp, sp); else call_A<NL, true, MD>(a, Kp, sp);
        } else if (kPairable && pairm && pair_b) call_A<NL, false, MD, kPairable>(a, Kp, sp);
        else call_A<NL, false, MD>(a, Kp, sp);
        const double resid = st.rho * st.rcmax;
        if (st.mu <= P.mu_final && resid <= P.tol_feas && st.last_step <= P.tol_step) { st.status = QMPC_OK; active = false; }
        else if (st.it > itmax) { st.status = QMPC_MAX_ITER; active = false; }
        else {
          double sg = P.sigma;
          const double amin = fmin(st.last_ap, st.last_ad);
          if (st.it > 1 && amin >= 0.99) sg = P.sigma_fast;
          else if (st.it > 1 && amin < 0.2) sg = fmax(sg, 0.8);
          else if (st.it > 1 && amin < 0.5) sg = fmax(sg, 0.5);
          st.target = sg * st.mu;
          const bool wrows = warm && __any(st.rho != 0.0);
          // (the warm instantiations are split too since rho * rc reaches its sums rounded in every instantiation: ql_rounded)
          const bool pb = kPairable && pairm && pair_b;
          const bool okB = wrows ? ((pb && pair_w) ? call_B<NL, true, MD, kPairable>(a, Kp, sp) : call_B<NL, true, MD>(a, Kp, sp))
                                 : (pb ? call_B<NL, false, MD, kPairable>(a, Kp, sp) : call_B<NL, false, MD>(a, Kp, sp));
          if (!okB) { st.status = QMPC_NOT_PD; active = false; }
          else {
            if (wrows) { if (kPairable && pairm && pair_w) call_C<NL, true, MD, kPairable>(a, Kp, sp); else call_C<NL, true, MD>(a, Kp, sp); }
            else if (kPairable && pairm) call_C<NL, false, MD, kPairable>(a, Kp, sp);
            else call_C<NL, false, MD>(a, Kp, sp);
            if (st.bad_step) { st.status = QMPC_NOT_PD; active = false; }     // a non-finite trial step is not applied
          }
        }
      }
#if defined(QL_PROFILE)
      rounds++;
#endif
    }
    if (valid)
      call_finish<NL, MD>(a, Kp, sp, reinterpret_cast<unsigned long long>(forces + (size_t)b * D::NU),
                      info ? reinterpret_cast<unsigned long long>(info + b) : 0ull,
                      traj_u ? reinterpret_cast<unsigned long long>(traj_u + (size_t)b * tstride) : 0ull,
                      traj_x ? reinterpret_cast<unsigned long long>(traj_x + (size_t)b * (size_t)(P.N + 1) * (MD == MD_CONVEX ? 12 : 13)) : 0ull);
    // hand-off: an instance stopped by the cap joins the list; its state travels with it while the buffer has room (the
    // wave kernel starts the others from scratch)
    if (NL == 4 && hcount && valid && (!pairm || lane < 32) && itmax < P.iterations_max && st.status == QMPC_MAX_ITER) {
      const int ord = atomicAdd(hcount, 1);
      hsel[ord] = b;
      if (ord < hcap) call_dump<NL>(a, sp, reinterpret_cast<unsigned long long>(hstate + (size_t)ord * (8 + 84 * (size_t)P.N)), warm ? 1 : 0);
    }
#if defined(QL_PROFILE)
    if (prof && base < (long long)slots) {      // first round of every wave; lane 0's clock, every lane's own iteration count
      if (lane == 0) {
        for (int i = 0; i < LP_COUNT; ++i) prof[16 * blockIdx.x + i] = st.t[i];
        prof[16 * blockIdx.x + 15] = st.it;      // lane 0's own iterations: its clock stops when its instance is done
        prof[16 * blockIdx.x + 14] = rounds;
      }
    }
#endif
  }
}

// ---- counting sort of the batch on the stance mask (keys 0 .. 2^NL - 1) ---------------------------------------------------
// scratch layout (ints): hist[256] | cursor[256] | perm[batch].  A block histograms its 256 instances in LDS and touches
// the global counters once per key it holds (a batch has a handful of distinct masks: per-thread global atomics on three
// addresses took 0.65 ms per launch).
template <int NL>
// prev (four-point models only; null: not used): the records of the instances' PREVIOUS solves -- in the closed loop a robot
// needs about as many iterations as at its last tick, and a wavefront runs as long as the slowest of its lanes, so within a
// stance pattern the batch is ordered by that count (16 classes: 3 ... 18 iterations)
__device__ __forceinline__ unsigned stance_key(const double* __restrict__ in, int b, int con_off, const qmpc_info* __restrict__ prev) {
  const double* rec = in + (size_t)b * LDim<NL>::REC + con_off;
  unsigned key = 0;
#pragma unroll
  for (int l = 0; l < NL; ++l) key |= (rec[l] != 0.0) ? (1u << l) : 0u;
  if (NL == 4 && prev) {
    int it = prev[b].iterations - 3;
    it = it < 0 ? 0 : (it > 15 ? 15 : it);
    key = (key << 4) | (unsigned)it;
  }
  return key;
}
template <int NL>
__global__ __launch_bounds__(256) void qmpc_lane_sort_count(const double* __restrict__ in, int batch, int* __restrict__ scratch, int con_off,
                                                                const qmpc_info* __restrict__ prev) {
  __shared__ int hist[256];
  hist[threadIdx.x] = 0;
  __syncthreads();
  const int b = blockIdx.x * 256 + threadIdx.x;
  if (b < batch) atomicAdd(&hist[stance_key<NL>(in, b, con_off, prev)], 1);
  __syncthreads();
  if (hist[threadIdx.x]) atomicAdd(&scratch[threadIdx.x], hist[threadIdx.x]);
}
#if QL_UNIT != 2      // (not a template: defined once)
__global__ __launch_bounds__(64) void qmpc_lane_sort_scan(int* __restrict__ scratch) {
  if (threadIdx.x != 0) return;
  int run = 0;
  for (int k = 0; k < 256; ++k) {
    const int n = scratch[k];
    scratch[256 + k] = run;
    run += n;
  }
}
#endif
template <int NL>
__global__ __launch_bounds__(256) void qmpc_lane_sort_scatter(const double* __restrict__ in, int batch, int* __restrict__ scratch, int con_off,
                                                                  const qmpc_info* __restrict__ prev) {
  __shared__ int hist[256], base[256];
  hist[threadIdx.x] = 0;
  __syncthreads();
  const int b = blockIdx.x * 256 + threadIdx.x;
  unsigned key = 0;
  int rank = 0;
  if (b < batch) {
    key = stance_key<NL>(in, b, con_off, prev);
    rank = atomicAdd(&hist[key], 1);
  }
  __syncthreads();
  if (hist[threadIdx.x]) base[threadIdx.x] = atomicAdd(&scratch[256 + threadIdx.x], hist[threadIdx.x]);
  __syncthreads();
  if (b < batch) scratch[512 + base[key] + rank] = b;
}

}  // namespace lane
}  // namespace qmpc

using namespace qmpc;
using namespace qmpc::lane;

// the reference mode's launch (its own unit of the library when QL_UNIT is 1 / 2: the kernel is scheduled differently, see the top
// of the file); dev_params: the block to upload into THIS unit's table first, or null (uploaded already)
#if QL_UNIT != 1
__attribute__((visibility("hidden"))) hipError_t qmpc_lane_ref_upload_params(int pslot, hipStream_t s, const void* dev_params) {
#if QL_UNIT == 2
  return hipMemcpyToSymbolAsync(HIP_SYMBOL(ql_params), dev_params, sizeof(DevParams), sizeof(DevParams) * (size_t)pslot, hipMemcpyHostToDevice, s);
#else
  (void)pslot; (void)s; (void)dev_params;
  return hipSuccess;      // one unit: one table
#endif
}
__attribute__((visibility("hidden"))) hipError_t qmpc_lane_ref_launch(int nl, int pslot, int batch, hipStream_t s, const void* dev_params,
                                                                       const double* rec, double* forces, qmpc_info* info, double* ws,
                                                                       unsigned waves, unsigned used, int lanes, const int* perm, double* traj_u,
                                                                       double* traj_x, size_t lds, long long* prof) {
  const bool convex = nl == -4;
  if (convex) nl = 4;
#if QL_UNIT == 2
  if (dev_params) {
    const hipError_t e = qmpc_lane_ref_upload_params(pslot, s, dev_params);
    if (e != hipSuccess) return e;
  }
#endif
#if defined(QL_PROFILE)
  long long* d_prof = prof;
  DevParams P;
  if (dev_params) memcpy(&P, dev_params, sizeof P); else memset(&P, 0, sizeof P);
#endif
  {
    if (nl == 8)
      hipLaunchKernelGGL(qmpc_lane_ref_kernel<8>, dim3(waves), dim3(kLaneWave), lds, s, pslot, rec, forces, info, batch, ws, used, lanes, perm,
                         traj_u, traj_x, prof);
    else if (convex)
      hipLaunchKernelGGL((qmpc_lane_ref_kernel<4, MD_CONVEX>), dim3(waves), dim3(kLaneWave), lds, s, pslot, rec, forces, info, batch, ws, used,
                         lanes, perm, traj_u, traj_x, prof);
    else {
      static const int pair_ref_env = std::getenv("QMPC_LANE_PAIR") ? std::atoi(std::getenv("QMPC_LANE_PAIR")) : 1;
      hipLaunchKernelGGL(qmpc_lane_ref_kernel<4>, dim3(waves), dim3(kLaneWave), lds, s, pslot, rec, forces, info, batch, ws, used,
                         (lanes == 32 && pair_ref_env) ? -34 : lanes, perm, traj_u, traj_x, prof);
    }
#if defined(QL_PROFILE)
    {
      static long long hp[16 * 1024];
      if (hipStreamSynchronize(s) == hipSuccess && hipMemcpy(hp, d_prof, sizeof hp, hipMemcpyDeviceToHost) == hipSuccess) {
        static const char* names[6] = {"B_AL", "C_AL (line search)", "A_AL", "S", "M (+tests)", "rest"};
        const unsigned nw = waves < 1024 ? waves : 1024;
        double sum[6] = {0}, tot = 0.0;
        for (unsigned w = 0; w < nw; ++w) for (int i = 0; i < 6; ++i) sum[i] += (double)hp[16 * w + i];
        for (int i = 0; i < 6; ++i) tot += sum[i];
        std::fprintf(stderr, "lane ref profile: batch %d N %d waves %u, %.0f cycles per wavefront\n", batch, P.N, waves, tot / nw);
        for (int i = 0; i < 6; ++i) std::fprintf(stderr, "  %-20s %10.0f cycles  %5.1f %%\n", names[i], sum[i] / nw, 100.0 * sum[i] / tot);
      }
    }
#endif
    return hipGetLastError();
  }
}
#else
__attribute__((visibility("hidden"))) hipError_t qmpc_lane_ref_upload_params(int pslot, hipStream_t s, const void* dev_params);
__attribute__((visibility("hidden"))) hipError_t qmpc_lane_ref_launch(int nl, int pslot, int batch, hipStream_t s, const void* dev_params, const double* rec, double* forces,
                                qmpc_info* info, double* ws, unsigned waves, unsigned used, int lanes, const int* perm, double* traj_u, double* traj_x,
                                size_t lds, long long* prof);
#endif
#if QL_UNIT != 2
// called from qmpc_hip.hip (declared there); hidden: not part of the C ABI
__attribute__((visibility("hidden"))) size_t qmpc_lane_ws_bytes(int N, int nl, unsigned slots, int wide) {
  return sizeof(double) * lane_ws_elements(N, nl, wide != 0) * (size_t)slots;      // wide: handles in the reference's solver mode
}
__attribute__((visibility("hidden"))) size_t qmpc_lane_scratch_bytes(int batch) { return sizeof(int) * (512 + (size_t)batch); }
// hand-off list of a capped launch: count | instance indices [batch]; the state records live in their own buffer
__attribute__((visibility("hidden"))) size_t qmpc_lane_handoff_list_bytes(int batch) { return sizeof(int) * (64 + (size_t)batch); }
__attribute__((visibility("hidden"))) size_t qmpc_lane_handoff_record_doubles(int N) { return 8 + 84 * (size_t)N; }

// slots: resident lanes (multiple of 64); scratch: qmpc_lane_scratch_bytes(batch) bytes, or null for no sort
// pslot: the handle's slot in the constant-memory parameter table (qmpc_lane_param_slots() of them); the block is copied
// there stream-ordered before every launch, so qmpc_set_params takes effect like for the other kernels
__attribute__((visibility("hidden"))) int qmpc_lane_param_slots() { return kParamSlots; }
__attribute__((visibility("hidden"))) hipError_t qmpc_lane_upload_params(int pslot, hipStream_t s, const void* dev_params,
                                                                          size_t dev_params_size) {
  if (dev_params_size != sizeof(DevParams) || pslot < 0 || pslot >= kParamSlots) return hipErrorInvalidValue;
  const hipError_t e = qmpc_lane_ref_upload_params(pslot, s, dev_params);      // (the reference mode's unit has its own table)
  if (e != hipSuccess) return e;
  return hipMemcpyToSymbolAsync(HIP_SYMBOL(ql_params), dev_params, sizeof(DevParams), sizeof(DevParams) * (size_t)pslot,
                                hipMemcpyHostToDevice, s);
}
__attribute__((visibility("hidden"))) hipError_t qmpc_lane_launch(int nl, int pslot, int batch, hipStream_t s, const void* dev_params,
                                                                   size_t dev_params_size, const void* in, double* forces,
                                                                   qmpc_info* info, double* ws, unsigned slots, int* scratch,
                                                                   int upload_params, const double* u_init, double* traj_u,
                                                                   int check_prev, int order_prev, double* traj_x, int iter_cap,
                                                                   int* hcount, int* hsel, double* hstate, int hcap) {
  // nl: 4 (QuatMpc), 8 (the 8-contact-point model) or -4 (ConvexMpc's model: four points, world-frame forces)
  const bool convex = nl == -4;
  if (convex) nl = 4;
  const int con_off = convex ? 24 : (nl == 8 ? LDim<8>::R_CON : LDim<4>::R_CON);      // contacts[] inside a record
  if (dev_params_size != sizeof(DevParams) || (nl != 4 && nl != 8) || slots % kLaneWave || pslot < 0 || pslot >= kParamSlots)
    return hipErrorInvalidValue;
  DevParams P;
  memcpy(&P, dev_params, sizeof P);
  if (upload_params) {     // 0: the caller uploaded them on this stream already and repeats the launch (captured closed loop)
    const hipError_t e = hipMemcpyToSymbolAsync(HIP_SYMBOL(ql_params), dev_params, sizeof(DevParams), sizeof(DevParams) * (size_t)pslot,
                                                hipMemcpyHostToDevice, s);
    if (e != hipSuccess) return e;
  }
  const double* rec = static_cast<const double*>(in);
  if (hcount) {
    const hipError_t e = hipMemsetAsync(hcount, 0, 2 * sizeof(int), s);      // the list's length and the list kernel's cursor
    if (e != hipSuccess) return e;
  }
  const int* perm = nullptr;
  if (scratch) {
    const qmpc_info* prev = (order_prev && info) ? info : nullptr;      // the previous solves' records, still in the output buffer
    hipError_t e = hipMemsetAsync(scratch, 0, sizeof(int) * 512, s);
    if (e != hipSuccess) return e;
    const unsigned blocks = (unsigned)((batch + 255) / 256);
    if (nl == 8) {
      hipLaunchKernelGGL(qmpc_lane_sort_count<8>, dim3(blocks), dim3(256), 0, s, rec, batch, scratch, con_off, prev);
      hipLaunchKernelGGL(qmpc_lane_sort_scan, dim3(1), dim3(64), 0, s, scratch);
      hipLaunchKernelGGL(qmpc_lane_sort_scatter<8>, dim3(blocks), dim3(256), 0, s, rec, batch, scratch, con_off, prev);
    } else {
      hipLaunchKernelGGL(qmpc_lane_sort_count<4>, dim3(blocks), dim3(256), 0, s, rec, batch, scratch, con_off, prev);
      hipLaunchKernelGGL(qmpc_lane_sort_scan, dim3(1), dim3(64), 0, s, scratch);
      hipLaunchKernelGGL(qmpc_lane_sort_scatter<4>, dim3(blocks), dim3(256), 0, s, rec, batch, scratch, con_off, prev);
    }
    perm = scratch + 512;
  }
  // batches that would occupy at most half of the chip's SIMDs with full wavefronts run with 32 lanes per wavefront
  static const int lanes_env = std::getenv("QMPC_LANE_WIDTH") ? std::atoi(std::getenv("QMPC_LANE_WIDTH")) : 0;
  const int lanes = lanes_env == 32 || lanes_env == 64 ? lanes_env : ((size_t)batch * 2 <= slots ? 32 : 64);
  const unsigned need = (unsigned)(((size_t)batch + lanes - 1) / lanes);
  const unsigned waves = need < slots / kLaneWave ? need : slots / kLaneWave;
  const unsigned used = waves * (unsigned)lanes;     // instances in flight: the batch stride
  const size_t lds = sizeof(double) * kLaneWave * LDim<4>::PLDS;
  long long* prof = nullptr;
#if defined(QL_PROFILE)
  static long long* d_prof = nullptr;      // diagnostic build only: per-wave phase cycles of the first round, printed after the launch
  if (!d_prof && hipMalloc(&d_prof, sizeof(long long) * 16 * 1024) != hipSuccess) return hipErrorOutOfMemory;
  prof = d_prof;
  (void)hipMemsetAsync(d_prof, 0, sizeof(long long) * 16 * 1024, s);
#endif
  if (P.mode == QMPC_MODE_REFERENCE) {      // the reference's own solver mode (QuatMpc with four or eight contact points, ConvexMpc): qmpc_lane_ref_kernel
    if (u_init) return hipErrorInvalidValue;
    return qmpc_lane_ref_launch(convex ? -4 : nl, pslot, batch, s, upload_params ? dev_params : nullptr, rec, forces, info, ws, waves, used, lanes, perm,
                                traj_u, traj_x, lds, prof);
  }
  if (nl == 8)
    hipLaunchKernelGGL(qmpc_lane_kernel<8>, dim3(waves), dim3(kLaneWave), lds, s, pslot, rec, forces, info, batch, ws, used, lanes, perm, prof,
                       u_init, traj_u, check_prev, traj_x, iter_cap, hcount, hsel, hstate, hcap);
  else if (convex)
    hipLaunchKernelGGL((qmpc_lane_kernel<4, MD_CONVEX>), dim3(waves), dim3(kLaneWave), lds, s, pslot, rec, forces, info, batch, ws, used, lanes,
                       perm, prof, u_init, traj_u, check_prev, traj_x, iter_cap, hcount, hsel, hstate, hcap);
  else {
    // half-filled wavefronts of the four-point quaternion model: lane pairs (the kernel reads -32 as "32 instances, pairs")
    static const int pair_env = std::getenv("QMPC_LANE_PAIR") ? std::atoi(std::getenv("QMPC_LANE_PAIR")) : 1;
    const int lanes_arg = (lanes == 32 && pair_env) ? (pair_env == 2 ? -33 : (pair_env == 4 ? -32 : -34)) : lanes;      // 4: cold rounds only      // QMPC_LANE_PAIR=2: split the trial pass only
    hipLaunchKernelGGL(qmpc_lane_kernel<4>, dim3(waves), dim3(kLaneWave), lds, s, pslot, rec, forces, info, batch, ws, used, lanes_arg, perm, prof,
                       u_init, traj_u, check_prev, traj_x, iter_cap, hcount, hsel, hstate, hcap);
  }
#if defined(QL_PROFILE)
  {
    static long long hp[16 * 1024];
    if (hipStreamSynchronize(s) == hipSuccess && hipMemcpy(hp, d_prof, sizeof hp, hipMemcpyDeviceToHost) == hipSuccess) {
      static const char* names[LP_COUNT] = {"A", "B.head", "B.legs", "B.expand", "B.MP", "B.congr", "B.fact", "B.upd", "B.gain",
                                            "C.head", "C.legs", "C.step"};
      const unsigned nw = waves < 1024 ? waves : 1024;
      double tot = 0.0, sum[LP_COUNT] = {0}, rounds = 0.0, wrounds = 0.0;
      for (unsigned w = 0; w < nw; ++w) {
        for (int i = 0; i < LP_COUNT; ++i) sum[i] += (double)hp[16 * w + i];
        rounds += (double)hp[16 * w + 15];
        wrounds += (double)hp[16 * w + 14];
      }
      for (int i = 0; i < LP_COUNT; ++i) tot += sum[i];
      std::fprintf(stderr, "lane profile: batch %d N %d waves %u, rounds per wave %.1f (lane 0: %.1f iterations), cycles per knot-iteration %.0f\n",
                   batch, P.N, waves, wrounds / nw, rounds / nw, tot / rounds / P.N);
      for (int i = 0; i < LP_COUNT; ++i)
        std::fprintf(stderr, "  %-9s %7.0f cycles per knot-iteration  %5.1f %%\n", names[i], sum[i] / rounds / P.N, 100.0 * sum[i] / tot);
    }
  }
#endif
  return hipGetLastError();
}
#endif      // QL_UNIT != 2
