// qmpc_lane_core.h -- the lane-per-instance solver core: ONE GPU lane owns ONE MPC instance, so a wavefront
// advances 64 independent solves in lock step with no cross-lane traffic at all, and the per-instance working set
// (trajectory, slacks / multipliers, feedback gains) streams through an HBM workspace laid out structure-of-arrays
// over the lanes ("stacked horizon matrices", coalesced 512-byte accesses per wave instruction).  This is the
// large-batch form of the converged mode of include/qmpc.h; the wave-per-instance kernels (qmpc_kernels.hip) remain
// the low-latency form for batches that do not fill the chip with lanes.
//
// Same problem, same interior-point iteration as qmpc_solve_body.inc (reference: QuatMpc.cpp:109-276 poses it,
// AltroUtils.cpp:363-439 / :9-22,78-110 are the dynamics and their midpoint linearisation): centring rule, Tapia
// flags on weakly active rows, analytic slack residual, rotated per-contact-point blocks.  What differs is how the
// Newton system of a knot is eliminated.  The 3 NL inputs act on the state only through the 6-dimensional wrench
//     Bbar_k = M_k Wr,   Wr = [c_l I ; Bw0_l] (6 x 3NL, the same for every knot),   M_k (12 x 6),
// and the input Hessian is block diagonal apart from that coupling, Quu = D + Wr' (M' P M) Wr with D = blkdiag(D_l),
// D_l = R_l + sum_i w_i a_i a_i' (3 x 3 per contact point).  With S6 = M'PM = L L', G = Wr D^-1 Wr' = sum_l V_l D_l^-1 V_l'
// and H = I + L' G L = C C' (eigenvalues >= 1) the push-through identity gives
//     du_l = -D_l^-1 (Wr_l' zeta + g_l),   zeta = L H^-1 (L^-1 (Y dx + y) - L' r6),   Y = M'P Abar, y = M'p,
//     P  <- Abar'P Abar + lxx - Yt'Yt + Yh'Yh,        Yt = L^-1 Y,  Yh = C^-1 Yt,
// i.e. NL independent 3 x 3 solves (in the rotated frame that keeps 1e14 : 1e-6 weight ratios accurate, DESIGN.md
// section 2) and two 6 x 6 Cholesky factorisations instead of a 3NL x 3NL Gauss-Jordan; Abar's block structure
// ([[I,0,hI,0],[0,A1,0,A3],[0,0,I,0],[0,0,0,I]]) is exploited in every product.  About 5.5 kflop per knot and
// iteration instead of the 35 kflop of the dense recursion -- and what is stored between the backward and the
// forward pass is the 6 x 13 wrench-space gain, not 3NL x 13.
//
// Where things live (device): the cost-to-go matrix P of the backward pass (78 doubles) in LDS, one 512-byte row per
// entry; the instance's constants (rotation, references: 25 doubles) in registers for a pass, the contact points'
// positions in the lane's private memory (they are fetched with a point's rows); everything indexed by knot in the HBM
// workspace, laid out [wave][element][lane]: a wavefront's working set is one contiguous block and consecutive elements
// are consecutive 512-byte rows.  One wavefront per SIMD has nobody to switch to: rows are prefetched one contact point
// ahead into the registers just consumed (LegAhead), and nothing is allowed to spill inside the loops -- the memory
// counter is in order, so a spill re-load behind a prefetch waits for the prefetch.
//
// Template parameters of the passes: NL (4 or 8 contact points), WARM (a warm-started launch: the rows' initial slack
// residuals come from the workspace while any lane carries one), MD (MD_QUAT, or MD_CONVEX: ConvexMpc's Euler-angle model,
// whose transition has the same block shape once the yaw-dependent inertia goes into the per-point map).
//
// The file is plain C++ on purpose: hipcc compiles it into qmpc_lane_kernel (qmpc_lane.hip), g++ compiles the very
// same text into the CPU numerics test of the core (tests/lane_core_host.cpp; test infrastructure, never a product
// path -- the product has no CPU fallback).
#pragma once

#include <math.h>
#include <stddef.h>
#include <stdint.h>
#include <type_traits>

#include "qmpc_params_dev.h"

#if defined(__HIPCC__)
#define QL_FN __device__ __forceinline__
#define QL_HD __host__ __device__ __forceinline__
#define QL_DEVICE 1
#else
#define QL_FN inline
#define QL_HD inline
#define QL_DEVICE 0
#endif

namespace qmpc {
namespace lane {

// a product that must reach its consumer ROUNDED (never contracted into the add that follows): whether the compiler fuses
// rho * rc into a neighbouring sum depends on how many uses the product has in the surrounding code, i.e. on the
// instantiation -- the plain and the pair-split warm passes differed by one rounding there
QL_FN double ql_rounded(double x) {
#if QL_DEVICE
  asm volatile("" : "+v"(x));
#endif
  return x;
}
// a value the caller knows to be wave-uniform, pinned to scalar registers: a per-lane choice between two such values then
// stays a choice between VALUES (the compiler otherwise turns `cond ? p[i] : p[j]` into one vector load through a chosen address)
QL_FN double ql_uniform(double x) {
#if QL_DEVICE
  const int lo = __builtin_amdgcn_readfirstlane(__double2loint(x)), hi = __builtin_amdgcn_readfirstlane(__double2hiint(x));
  return __hiloint2double(hi, lo);
#else
  return x;
#endif
}
QL_FN double ql_rcp(double x) {
#if QL_DEVICE
  double r = __builtin_amdgcn_rcp(x);
  r = fma(fma(-x, r, 1.0), r, r);
  r = fma(fma(-x, r, 1.0), r, r);
  return r;
#else
  return 1.0 / x;
#endif
}
QL_FN double ql_rsqrt(double x) {
#if QL_DEVICE
  double r = __builtin_amdgcn_rsq(x);
  r = r * fma(-0.5 * x * r, r, 1.5);
  r = r * fma(-0.5 * x * r, r, 1.5);
  return r;
#else
  return 1.0 / sqrt(x);
#endif
}

// ---- sizes, record offsets (qmpc_input / qmpc_input8) and per-lane constant slots -------------------------------
template <int NL>
struct LDim {
  static constexpr int NU = 3 * NL, NC = 6 * NL;
  static constexpr int REC = 32 + 4 * NL;
  static constexpr int R_FOOT = 19, R_CON = 19 + 3 * NL, R_POS = 19 + 4 * NL, R_QD = R_POS + 9;
  static constexpr int PLDS = 78;     // entries of the symmetric cost-to-go matrix kept in lane-private LDS rows
  // wrench-space gain of a knot: the feedback part Xg (6 x 12) in SINGLE precision, two entries per 8-byte element
  // (element 3 j + i/2 holds rows i, i+1 of column j) -- it multiplies the state increment of the rollout, so its rounding
  // changes the Newton direction by 6e-8 of the step and not the fixed point -- followed by the feed-forward zeta0 (6) in
  // double precision
  static constexpr int GAIN = 36 + 6;
  // The reference's solver mode (AL passes) keeps the feedback part in DOUBLE precision: a truncated iterate does not damp
  // the rounding of its directions (5.9e-8 of the step in the packed form: 99.9 / 99.1 % of the N = 10 / 20 forces within
  // 1e-6 N of the oracle's, and a handful of line searches per 100 000 instances decided the other way).  Columns 0..5
  // (36 doubles) and zeta0 fill the SAME 42-element slot; columns 6..11 live in a second block of GAIN2 per knot at the end
  // of the (wide) layout, so that no offset of the converged mode moves.
  static constexpr int GAIN2 = 36;
};

// constants of one instance (registers)
template <int NL>
struct LaneK {
  double rot[9];         // body -> world rotation, row-major (cone rows are C_mat * R, QuatMpc.cpp:47-52,203)
  double foot[3 * NL];   // contact points in the body frame
  double wd0[3];         // Iinv (c x 5.204 g_body)   (AltroUtils.cpp:373-374,391)
  double refp[13];       // reference parameters: pos vel acc quat_d
};

#ifndef QL_AL_G2_AHEAD
#define QL_AL_G2_AHEAD 2      // second gain block of the trial rollouts: 0 at the top of its knot, 1 a whole knot ahead (it spills: 14.2 -> 17.0 ms at 65536 instances), 2 at the end of the previous knot
#endif
// workspace offsets in ELEMENTS of one lane's column (element e of lane s of a wave lives at wave_base[64 e + s])
struct WsOff {
  int X, U, dU, S, LAM, G, RC, total, G2;
};
// wide: the layout of the reference mode's passes (a second gain block G2 after everything else; every other offset equal)
template <int NL>
QL_HD WsOff make_wsoff(int N, bool wide = false) {
  WsOff o;
  int p = 0;
  o.X = p; p += 13 * (N + 1);
  o.U = p; p += 3 * NL * N;
  o.dU = p; p += 3 * NL * N;
  o.S = p; p += 6 * NL * N;
  o.LAM = p; p += 6 * NL * N;
  o.G = p; p += LDim<NL>::GAIN * N;
  o.RC = p; p += 6 * NL * N;      // initial slack residuals per row: only a warm-started launch has them per knot
  o.G2 = p;
  if (wide) p += LDim<NL>::GAIN2 * N;
  o.total = p;
  return o;
}
inline size_t lane_ws_elements(int N, int nl, bool wide = false) {
  return nl == 8 ? (size_t)make_wsoff<8>(N, wide).total : (size_t)make_wsoff<4>(N, wide).total;
}

// Addresses: a wave-uniform base (scalar registers, constant for the whole solve) plus one 32-bit per-lane byte offset
// whose element part is a small multiple of the row size -- consecutive elements differ by an immediate.  On the device
// the pointers carry their address spaces (global / LDS), so that the accesses stay global_load / ds_read when the
// passes are compiled as separate functions.
#if QL_DEVICE
#define QL_GLOBAL_AS __attribute__((address_space(1)))
#define QL_LDS_AS __attribute__((address_space(3)))
#define QL_PRIV_AS __attribute__((address_space(5)))
#else
#define QL_GLOBAL_AS
#define QL_LDS_AS
#define QL_PRIV_AS
#endif
// The contact points' positions are read from the instance's constants in the lane's PRIVATE memory together with the
// rows of the point (LegAhead), three doubles at a time, instead of living in 6 NL registers through every pass: where
// the compiler parked them in scratch by itself it re-loaded them at the top of each point's block, behind the
// prefetches just issued (the memory counter is in order: waiting for the re-load waited for the prefetch).
constexpr bool kFootAhead = true;
typedef QL_PRIV_AS const double* FootPtr;
#ifndef QL_NT_ST         // the plain forms' sweep stores are non-temporal (nothing re-reads a row before the workspace has streamed
#define QL_NT_ST 1         // through the caches once): config 3 -2 %, B=65536 N=10 -1 %; the pair forms (half the bytes per wavefront:
#endif                    // their rows do come back from the L2) keep ordinary stores (StOwn): +1 % with the hint
#ifndef QL_NT_LD
#define QL_NT_LD 0
#endif
struct Ctx {
  QL_GLOBAL_AS double* ws;   // this wave's block of the workspace: [element][lane]
  unsigned wrow;             // bytes per workspace row (8 x lanes per wave)
  unsigned woff;             // this lane's byte offset inside a row
  QL_LDS_AS double* pl;      // lane-private rows for the cost-to-go matrix (LDS): [entry][lane]
  unsigned prow, poff;
  // Lane PAIRS (round 5; batches that fill only half of every wavefront): lanes i and i + 32 work on the SAME instance --
  // same column of the workspace and of the LDS rows (woff = poff = 8 (lane & 31)), every pass duplicated except the
  // per-point blocks of the passes that split them (pass_C<..., PAIR>: lane i takes the first point of a pair, lane i + 32
  // the second).  A wave64 FP64 instruction issues its four 16-lane passes whatever the mask, so the partner lanes are free.
  unsigned half = 0;         // 1 on the upper partner lane of a pair
  unsigned lmask = 0x1F8;    // relane(): 8 (lane & 63), or 8 (lane & 31) in pair mode (0xF8)
  int foot_row = -1;         // pair forms: first of the LDS staging rows that hold the contact points' body-frame positions (-1: private memory)
  QL_FN QL_GLOBAL_AS double& W(int e) const {
    return *reinterpret_cast<QL_GLOBAL_AS double*>(reinterpret_cast<QL_GLOBAL_AS char*>(ws) + ((unsigned)e * wrow + woff));
  }
  // store of a pass (the hot sweeps A / B / C): the same instruction as `W(e) = v`.  Diagnostic builds
  // (-DQL_DIAG_NOSTORE, tools/lane_variants.py: the traffic-bound experiment of profiles/r06_lane_traffic_bound.txt)
  // keep the value alive and drop the store.
  QL_FN void St(int e, double v) const {
#if defined(QL_DIAG_NOSTORE) && QL_DEVICE
    asm volatile("" ::"v"(v));
#elif defined(QL_PAIR_ST_LOWER) && QL_DEVICE
    if (!half) W(e) = v;      // pair mode: the partner lanes' copies of a duplicated pass carry the same values to the same addresses
#elif QL_NT_ST && QL_DEVICE
    __builtin_nontemporal_store(v, &W(e));
#else
    W(e) = v;
#endif
  }
  // a row read that nothing re-reads before the workspace has streamed through the caches once (experiment: QL_NT_LD)
  QL_FN double Ld(int e) const {
#if QL_NT_LD && QL_DEVICE
    return __builtin_nontemporal_load(&W(e));
#else
    return W(e);
#endif
  }
  // a store that is this lane's OWN in pair mode (the split trial pass: each partner writes its point's increments)
  QL_FN void StOwn(int e, double v) const {
#if defined(QL_DIAG_NOSTORE) && QL_DEVICE
    asm volatile("" ::"v"(v));
#elif QL_NT_ST && QL_DEVICE
    __builtin_nontemporal_store(v, &W(e));
#else
    W(e) = v;
#endif
  }
  QL_FN QL_LDS_AS double& PL(int i) const {
    return *reinterpret_cast<QL_LDS_AS double*>(reinterpret_cast<QL_LDS_AS char*>(pl) + ((unsigned)i * prow + poff));
  }
  // Pair mode: the partner lanes share the lower 256 B of every 512-B row of the LDS block, the upper halves are unused -- 78
  // staging rows that `global_load_lds_dwordx4` fills WITHOUT passing through registers (the pair forms have none to spare for
  // prefetches: every register prefetch they were given came back as scratch traffic).  stage<n>(e0, h0): rows e0 .. e0 + n - 1
  // of the workspace (256 B each in pair mode) into the staging rows h0 .. h0 + n - 1, n <= 8: sixteen lanes carry 16 B each
  // (LDS address = M0 + instruction offset + 16 x lane id; the instruction offset also advances the global address, and
  // both kinds of rows are 512 B apart).  The transfers count in vmcnt like loads: staged() waits for everything in flight.
  template <int n>
  QL_FN void stage(int e0, int h0) const {
#if QL_DEVICE
    static_assert(n >= 1 && n <= 8, "instruction offsets reach 4095");
    // (the lanes 0..15 are forced on for the transfers whatever the exec mask is -- a lane carries 16 B = the values of TWO
    // instances, neither of which need be its own -- so their addresses are formed inside, under the forced mask)
    const unsigned long long sa = (unsigned long long)ws + (unsigned long long)(unsigned)e0 * wrow;
    const unsigned sa_lo = __builtin_amdgcn_readfirstlane((unsigned)sa), sa_hi = __builtin_amdgcn_readfirstlane((unsigned)(sa >> 32));
    const unsigned long long sbase = ((unsigned long long)sa_hi << 32) | sa_lo;
    const unsigned m0v = __builtin_amdgcn_readfirstlane((unsigned)(unsigned long long)pl + (unsigned)h0 * prow + 256u);
    unsigned long long sv;
    unsigned vo;
    asm volatile(
        "s_mov_b64 %[sv], exec\n\t"
        "s_mov_b64 exec, 0xffff\n\t"
        "v_mbcnt_lo_u32_b32 %[vo], -1, 0\n\t"
        "v_lshlrev_b32 %[vo], 4, %[vo]\n\t"
        "s_mov_b32 m0, %[m0v]\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %[vo], %[sb]\n\t"
        ".if %[n] > 1\n\tglobal_load_lds_dwordx4 %[vo], %[sb] offset:512\n\t.endif\n\t"
        ".if %[n] > 2\n\tglobal_load_lds_dwordx4 %[vo], %[sb] offset:1024\n\t.endif\n\t"
        ".if %[n] > 3\n\tglobal_load_lds_dwordx4 %[vo], %[sb] offset:1536\n\t.endif\n\t"
        ".if %[n] > 4\n\tglobal_load_lds_dwordx4 %[vo], %[sb] offset:2048\n\t.endif\n\t"
        ".if %[n] > 5\n\tglobal_load_lds_dwordx4 %[vo], %[sb] offset:2560\n\t.endif\n\t"
        ".if %[n] > 6\n\tglobal_load_lds_dwordx4 %[vo], %[sb] offset:3072\n\t.endif\n\t"
        ".if %[n] > 7\n\tglobal_load_lds_dwordx4 %[vo], %[sb] offset:3584\n\t.endif\n\t"
        "s_mov_b64 exec, %[sv]"
        : [sv] "=&s"(sv), [vo] "=&v"(vo)
        : [sb] "s"(sbase), [m0v] "s"(m0v), [n] "n"(n)
        : "memory", "m0");
#else
    (void)e0; (void)h0;
#endif
  }
  QL_FN void staged() const {
#if QL_DEVICE
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
  }
  QL_FN double SR(int h) const {      // this instance's value of staging row h
    return *reinterpret_cast<QL_LDS_AS const double*>(reinterpret_cast<QL_LDS_AS const char*>(pl) + ((unsigned)h * prow + 256u + poff));
  }
  QL_FN void SRst(int h, double v) const {      // ... and a value parked there (both partner lanes write the same value)
    *reinterpret_cast<QL_LDS_AS double*>(reinterpret_cast<QL_LDS_AS char*>(pl) + ((unsigned)h * prow + 256u + poff)) = v;
  }
  // Recompute the lane's row offset (three VALU instructions) instead of keeping it live: where the register file is full
  // the compiler parks this value in scratch and re-loads it before every group of workspace accesses -- a round trip to
  // L2 that also waits for every load issued before it (the memory counters are in order), i.e. for the prefetches.
  QL_FN void relane() {
#if QL_DEVICE
    unsigned x;
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0\n\tv_lshlrev_b32 %0, 3, %0" : "=v"(x));
    woff = x & lmask;
#endif
  }
};

// optional phase-level cycle accounting (diagnostic builds: -DQL_PROFILE; tools/lane_prof.sh)
enum { LP_A = 0, LP_B_HEAD, LP_B_LEGS, LP_B_EXPAND, LP_B_MP, LP_B_CONGR, LP_B_FACT, LP_B_UPD, LP_B_GAIN, LP_C_HEAD, LP_C_LEGS,
       LP_C_STEP, LP_COUNT };
// compiler-level memory fence: values read from the lane-private rows before it are re-read after it instead of being
// kept in registers across a phase boundary (the point of keeping P in LDS is to get it OUT of the register file)
// true when ANY active lane of the wave has the predicate: a scalar branch, so that loads placed under it are issued
// for the whole wave ahead of the per-lane work (lanes that do not need the data ignore what they read)
#if QL_DEVICE
#define QL_ANY(x) (__any((int)(x)) != 0)
#else
#define QL_ANY(x) (x)
#endif
// values of the two partner lanes of a pair, seen by BOTH of them in the same order (v_permlane32_swap: no LDS): lo = the lower
// half-wave's lane, hi = the upper's -- whatever is formed from (lo, hi) is bit-identical in the two lanes
#if QL_DEVICE
QL_FN void ql_pair(double x, double& lo, double& hi) {
  typedef unsigned u2v_ __attribute__((ext_vector_type(2)));
  const unsigned xl = (unsigned)__double2loint(x), xh = (unsigned)__double2hiint(x);
  const u2v_ l = __builtin_amdgcn_permlane32_swap(xl, xl, false, false);
  const u2v_ h = __builtin_amdgcn_permlane32_swap(xh, xh, false, false);
  lo = __hiloint2double((int)h[0], (int)l[0]);
  hi = __hiloint2double((int)h[1], (int)l[1]);
}
#else
QL_FN void ql_pair(double x, double& lo, double& hi) { lo = x; hi = x; }
#endif
// the PARTNER lane's value of x (lane ^ 32), through the LDS crossbar (ds_bpermute_b32: no memory, one instruction per half) --
// where only the partner's value is wanted this replaces two copies, two swaps and two selects per value
#if QL_DEVICE
QL_FN double ql_partner(double x, unsigned addr) {      // addr = 4 (lane ^ 32)
  const int lo = __builtin_amdgcn_ds_bpermute((int)addr, __double2loint(x));
  const int hi = __builtin_amdgcn_ds_bpermute((int)addr, __double2hiint(x));
  return __hiloint2double(hi, lo);
}
QL_FN unsigned ql_partner_addr() {
  unsigned l;
  asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));
  return (l ^ 32u) << 2;
}
#else
QL_FN double ql_partner(double x, unsigned) { return x; }
QL_FN unsigned ql_partner_addr() { return 0; }
#endif
// scheduling barrier: the instruction scheduler moves nothing across it (two unrolled per-point blocks whose temporaries
// would otherwise be live together)
#if QL_DEVICE && !defined(QL_NO_SCHED_BARRIER)
#define QL_SCHED_BARRIER() __builtin_amdgcn_sched_barrier(0)
#else
#define QL_SCHED_BARRIER() do { } while (0)
#endif
// wait until no vector-memory operation of the wavefront is in flight (s_waitcnt vmcnt(0); gfx9 encoding: vmcnt in bits
// 3:0 and 15:14, expcnt 6:4 and lgkmcnt 11:8 left at their maxima = no wait)
#if QL_DEVICE
#define QL_WAIT_VMEM() __builtin_amdgcn_s_waitcnt(0x0F70)
#else
#define QL_WAIT_VMEM() do { } while (0)
#endif
#if QL_DEVICE && !defined(QL_NO_FENCE)
#define QL_FENCE() asm volatile("" ::: "memory")
#else
#define QL_FENCE() do { } while (0)
#endif
#if defined(QL_PROFILE) && QL_DEVICE
#define QL_TICK(st, ph) do { const long long now_ = clock64(); (st).t[ph] += now_ - (st).last; (st).last = now_; } while (0)
#else
#define QL_TICK(st, ph) do { } while (0)
#endif

// per-instance scalars (registers)
struct LaneState {
#if defined(QL_PROFILE) && QL_DEVICE
  long long t[LP_COUNT], last;
#endif
  unsigned con;        // stance mask
  int nc;              // stance count
  int status, iters, it;
  bool active;
  double rho;          // slack residual scale: rc_i = rho * rc0_i (rc shrinks by (1 - alpha_p) per step, 0 after a full one)
  double target;       // sigma * mu of the iteration whose step is pending
  double mu, last_ap, last_ad, last_step;
  double ap, ad;       // step lengths of the pending step
  int bad_step;        // the trial increment of the last forward pass was not finite (it is NOT applied: QMPC_NOT_PD)
  double uz;           // u_ref z-component of a stance contact point
  double rcmax;        // largest |rc0_i| over the enabled rows (the feasibility residual is rho * rcmax)
};

// ---- small dense helpers (everything is unrolled into registers) --------------------------------------------------
// symmetric 12 x 12 in 78 entries, upper triangle row-major
constexpr int SI_(int i, int j) { return i * 12 - i * (i - 1) / 2 + (j - i); }
constexpr int SI(int i, int j) { return i <= j ? SI_(i, j) : SI_(j, i); }
// pass B, step 5: which lane of a pair / which column of the plain form computes the entry (i, 6 + j) of the cost-to-go's
// off-diagonal block: kCross(i, j) -> as Y_i . z_(6+j) (column 6 + j), otherwise as z_i . Y_(6+j) (column i).  A tournament on
// 0..5 plus the diagonal: for i != j exactly one of kCross(i, j), kCross(j, i) holds.
constexpr bool kCross(int i, int j) { return i == j || (i - j + 6) % 6 == 1 || (i - j + 6) % 6 == 2 || ((i - j + 6) % 6 == 3 && i < j); }
static_assert(kCross(1, 0) && !kCross(0, 1) && kCross(0, 3) && !kCross(3, 0) && kCross(5, 5), "tournament");
// symmetric 6 x 6 in 21 entries
constexpr int S6_(int i, int j) { return i * 6 - i * (i - 1) / 2 + (j - i); }
constexpr int S6I(int i, int j) { return i <= j ? S6_(i, j) : S6_(j, i); }
// lower triangle of a 6 x 6, row-major: (i,j), j <= i
constexpr int LI(int i, int j) { return i * (i + 1) / 2 + j; }

QL_FN void rdblk(const Ctx& cx, int a, int b, double M[9]) {
#pragma unroll
  for (int r = 0; r < 3; ++r)
#pragma unroll
    for (int c = 0; c < 3; ++c) M[3 * r + c] = cx.PL(SI(3 * a + r, 3 * b + c));
}
// a < b: all nine entries; a == b: the upper triangle
QL_FN void wrblk(const Ctx& cx, int a, int b, const double M[9]) {
#pragma unroll
  for (int r = 0; r < 3; ++r)
#pragma unroll
    for (int c = 0; c < 3; ++c)
      if (a != b || r <= c) cx.PL(SI(3 * a + r, 3 * b + c)) = M[3 * r + c];
}
// C = A B, C = A' B (3 x 3 row-major)
QL_FN void mm(const double A[9], const double B[9], double C[9]) {
#pragma unroll
  for (int r = 0; r < 3; ++r)
#pragma unroll
    for (int c = 0; c < 3; ++c) C[3 * r + c] = A[3 * r] * B[c] + A[3 * r + 1] * B[3 + c] + A[3 * r + 2] * B[6 + c];
}
QL_FN void mtm(const double A[9], const double B[9], double C[9]) {
#pragma unroll
  for (int r = 0; r < 3; ++r)
#pragma unroll
    for (int c = 0; c < 3; ++c) C[3 * r + c] = A[r] * B[c] + A[3 + r] * B[3 + c] + A[6 + r] * B[6 + c];
}

// two floats in one 8-byte workspace element
QL_FN double pack2f(float a, float b) {
  unsigned long long u;
  unsigned ua, ub;
  __builtin_memcpy(&ua, &a, 4);
  __builtin_memcpy(&ub, &b, 4);
  u = ((unsigned long long)ub << 32) | ua;
  double d;
  __builtin_memcpy(&d, &u, 8);
  return d;
}
QL_FN void unpack2f(double d, float& a, float& b) {
  unsigned long long u;
  __builtin_memcpy(&u, &d, 8);
  const unsigned ua = (unsigned)u, ub = (unsigned)(u >> 32);
  __builtin_memcpy(&a, &ua, 4);
  __builtin_memcpy(&b, &ub, 4);
}

// G(q) (4 x 3), QuaternionUtils.cpp:48-52
QL_FN void quatG(const double* q, double G[12]) {
  const double s = q[0], x = q[1], y = q[2], z = q[3];
  G[0] = -x; G[1] = -y; G[2] = -z;
  G[3] = s;  G[4] = -z; G[5] = y;
  G[6] = z;  G[7] = s;  G[8] = -x;
  G[9] = -y; G[10] = x; G[11] = s;
}
// Omega(w) v (AltroUtils.cpp:408-410 without the 1/2)
QL_FN void omega_mul(const double* w, const double* v, double* o) {
  o[0] = -w[0] * v[1] - w[1] * v[2] - w[2] * v[3];
  o[1] = w[0] * v[0] + w[2] * v[2] - w[1] * v[3];
  o[2] = w[1] * v[0] - w[2] * v[1] + w[0] * v[3];
  o[3] = w[2] * v[0] + w[1] * v[1] - w[0] * v[2];
}

// Bw0_l = Iinv skew(r_l) (AltroUtils.cpp:431-434) for a stance contact point
QL_FN void leg_bw0(const DevParams& P, const double r[3], double B[9]) {
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    const double i0 = P.Iinv[3 * a], i1 = P.Iinv[3 * a + 1], i2 = P.Iinv[3 * a + 2];
    B[3 * a] = i1 * r[2] - i2 * r[1];
    B[3 * a + 1] = i2 * r[0] - i0 * r[2];
    B[3 * a + 2] = i0 * r[1] - i1 * r[0];
  }
}

// explicit midpoint step (AltroUtils.cpp:9-22 on :363-392) given the force sum F and the angular acceleration wd
QL_FN void srbd_step_fw(const DevParams& P, const double gb[3], const double* x, const double F[3], const double wd[3],
                        double* xn) {
  double vd[3];
#pragma unroll
  for (int a = 0; a < 3; ++a) vd[a] = F[a] * P.inv_mass + gb[a];
  double G[12];
  quatG(&x[3], G);
  double qm[4], wm[3];
#pragma unroll
  for (int r = 0; r < 4; ++r)
    qm[r] = x[3 + r] + P.hh * (0.5 * (G[3 * r] * x[10] + G[3 * r + 1] * x[11] + G[3 * r + 2] * x[12]));
#pragma unroll
  for (int a = 0; a < 3; ++a) wm[a] = x[10 + a] + P.hh * wd[a];
  quatG(qm, G);
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    xn[a] = x[a] + P.h * (x[7 + a] + P.hh * vd[a]);
    xn[7 + a] = x[7 + a] + P.h * vd[a];
    xn[10 + a] = x[10 + a] + P.h * wd[a];
  }
#pragma unroll
  for (int r = 0; r < 4; ++r)
    xn[3 + r] = x[3 + r] + P.h * (0.5 * (G[3 * r] * wm[0] + G[3 * r + 1] * wm[1] + G[3 * r + 2] * wm[2]));
}

// ---- ConvexMpc's model on the same core (MD = 1; ConvexMpc.cpp:81-198, AltroUtils.cpp:224-359) ----------------------------
// State in the workspace as in the record: [rpy, pos, ang_vel_world, lin_vel_world] (12 of the 13 rows of a knot), forces in
// the WORLD frame.  Inside the recursion the blocks are ordered like the quaternion model's error state, [p, phi, v, w],
// and the transition has the same shape: Abar = [[I,0,hI,0],[0,A1,0,A3],[0,0,I,0],[0,0,0,I]] with
//   A1 = I + h [0 0 j0; 0 0 j1; 0 0 0],   A3 = h [c s hh j0; -s c hh j1; 0 0 1]      (c, s, j at the midpoint yaw),
// and Bbar = M Wr_k with the wrench (F, t') where t' = Iw(yaw_m)^-1 sum r x u (the inertia at the MIDPOINT yaw goes into
// the per-point map Bw0 = Iw_m^-1 [r]x, knot by knot), M's attitude block Wt = h hh Rz_m' Iw(yaw)^-1 Iw(yaw_m).
enum { MD_QUAT = 0, MD_CONVEX = 1 };
enum { CVR_YAW = 0, CVR_RATE, CVR_POS, CVR_VX = 5, CVR_VY = 6 };      // refp of the convex model (qmpc_device.h CR_*)
// Iw(yaw)^-1 = Rz diag(1/I) Rz': W = {w00, w01, w11, wzz}
QL_FN void cv_winv(const DevParams& P, double c, double s, double W[4]) {
  const double a = P.Iinv[0], b = P.Iinv[4];
  W[0] = c * c * a + s * s * b;
  W[1] = c * s * (a - b);
  W[2] = s * s * a + c * c * b;
  W[3] = P.Iinv[8];
}
// B = Iw^-1 [r]x
QL_FN void cv_leg_bw0(const double W[4], const double r[3], double B[9]) {
  const double S[9] = {0.0, -r[2], r[1], r[2], 0.0, -r[0], -r[1], r[0], 0.0};
#pragma unroll
  for (int b = 0; b < 3; ++b) {
    B[b] = W[0] * S[b] + W[1] * S[3 + b];
    B[3 + b] = W[1] * S[b] + W[2] * S[3 + b];
    B[6 + b] = W[3] * S[6 + b];
  }
}
QL_FN void cv_cross_acc(const double r[3], const double u[3], double t[3]) {      // t += r x u
  t[0] += r[1] * u[2] - r[2] * u[1];
  t[1] += r[2] * u[0] - r[0] * u[2];
  t[2] += r[0] * u[1] - r[1] * u[0];
}
// explicit midpoint of ct_srb_dynamics (AltroUtils.cpp:9-22 on :224-293) given the force sum F and the torque sum tau
QL_FN void cv_step_fw(const DevParams& P, const double* x, const double F[3], const double tau[3], double* xn) {
  double s0, c0, W[4];
  sincos(x[2], &s0, &c0);
  cv_winv(P, c0, s0, W);
  const double vd[3] = {F[0] * P.inv_mass, F[1] * P.inv_mass, F[2] * P.inv_mass - 9.81};
  const double wd0[3] = {W[0] * tau[0] + W[1] * tau[1], W[1] * tau[0] + W[2] * tau[1], W[3] * tau[2]};
  const double yawm = x[2] + P.hh * x[8];
  const double wm[3] = {x[6] + P.hh * wd0[0], x[7] + P.hh * wd0[1], x[8] + P.hh * wd0[2]};
  double sm_, cm_;
  sincos(yawm, &sm_, &cm_);
  cv_winv(P, cm_, sm_, W);
  const double wdm[3] = {W[0] * tau[0] + W[1] * tau[1], W[1] * tau[0] + W[2] * tau[1], W[3] * tau[2]};
  xn[0] = x[0] + P.h * (cm_ * wm[0] + sm_ * wm[1]);
  xn[1] = x[1] + P.h * (-sm_ * wm[0] + cm_ * wm[1]);
  xn[2] = x[2] + P.h * wm[2];
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    xn[3 + a] = x[3 + a] + P.h * (x[9 + a] + P.hh * vd[a]);
    xn[6 + a] = x[6 + a] + P.h * wdm[a];
    xn[9 + a] = x[9 + a] + P.h * vd[a];
  }
  xn[12] = 0.0;
}
// reference state of knot k (ConvexMpc.cpp:95-106)
QL_FN void cv_xref_at(const DevParams& P, const double rp[13], int k, double* xr) {
  const double h_ms = P.h_ref * 1000.0;
#pragma unroll
  for (int i = 0; i < 13; ++i) xr[i] = 0.0;
  xr[2] = rp[CVR_YAW] + rp[CVR_RATE] * h_ms / 1000.0 * k;
  xr[3] = rp[CVR_POS]; xr[4] = rp[CVR_POS + 1]; xr[5] = rp[CVR_POS + 2];
  xr[8] = rp[CVR_RATE];
  xr[9] = rp[CVR_VX]; xr[10] = rp[CVR_VY];
}
// Iw(yaw_m)^-1 of knot state x (the per-point map of the linearisation at that knot)
QL_FN void cv_winv_mid(const DevParams& P, double yaw, double wz, double W[4]) {
  double sm_, cm_;
  sincos(yaw + P.hh * wz, &sm_, &cm_);
  cv_winv(P, cm_, sm_, W);
}
// A1, A3, Wt of a knot from its state and torque sum (AltroUtils.cpp:295-359 through the midpoint rule :78-110)
QL_FN void cv_expansion(const DevParams& P, const double* x, const double tau[3], double A1[9], double A3[9], double Wt[9]) {
  double s0, c0, W0[4], Wm[4];
  sincos(x[2], &s0, &c0);
  cv_winv(P, c0, s0, W0);
  const double wm0 = x[6] + P.hh * (W0[0] * tau[0] + W0[1] * tau[1]);
  const double wm1 = x[7] + P.hh * (W0[1] * tau[0] + W0[2] * tau[1]);
  double sm_, cm_;
  sincos(x[2] + P.hh * x[8], &sm_, &cm_);
  cv_winv(P, cm_, sm_, Wm);
  const double j0 = wm1 * cm_ - wm0 * sm_, j1 = -wm0 * cm_ - wm1 * sm_;
#pragma unroll
  for (int i = 0; i < 9; ++i) { A1[i] = (i % 4 == 0) ? 1.0 : 0.0; }
  A1[2] = P.h * j0; A1[5] = P.h * j1;
  A3[0] = P.h * cm_;  A3[1] = P.h * sm_; A3[2] = P.h * P.hh * j0;
  A3[3] = -P.h * sm_; A3[4] = P.h * cm_; A3[5] = P.h * P.hh * j1;
  A3[6] = 0.0; A3[7] = 0.0; A3[8] = P.h;
  // M1 = Rz_m' Iw(yaw)^-1 (upper-left 2 x 2; the rest is 1 / Izz),  Iw(yaw_m) = Rz_m diag(I) Rz_m'
  const double m00 = cm_ * W0[0] + sm_ * W0[1], m01 = cm_ * W0[1] + sm_ * W0[2];
  const double m10 = -sm_ * W0[0] + cm_ * W0[1], m11 = -sm_ * W0[1] + cm_ * W0[2];
  const double ixx = 1.0 / P.Iinv[0], iyy = 1.0 / P.Iinv[4];
  const double i00 = cm_ * cm_ * ixx + sm_ * sm_ * iyy, i01 = cm_ * sm_ * (ixx - iyy), i11 = sm_ * sm_ * ixx + cm_ * cm_ * iyy;
  const double hhh = P.h * P.hh;
  Wt[0] = hhh * (m00 * i00 + m01 * i01); Wt[1] = hhh * (m00 * i01 + m01 * i11); Wt[2] = 0.0;
  Wt[3] = hhh * (m10 * i00 + m11 * i01); Wt[4] = hhh * (m10 * i01 + m11 * i11); Wt[5] = 0.0;
  Wt[6] = 0.0; Wt[7] = 0.0; Wt[8] = hhh;
}

// reference state of knot k (QuatMpc.cpp:148-176) from refp = pos(3) vel(3) acc(3) quat_d(4)
QL_FN void xref_at(const DevParams& P, const double rp[13], int k, double* xr) {
  const double t = (double)k * P.h_ref;
  const double h_ms = P.h_ref * 1000.0;
  xr[0] = rp[0] + rp[3] * k * h_ms / 1000.0 + 0.5 * rp[6] * t * t;
  xr[1] = rp[1] + rp[4] * k * h_ms / 1000.0 + 0.5 * rp[7] * t * t;
  xr[2] = rp[2] + 0.5 * rp[8] * t * t;
  xr[3] = rp[9]; xr[4] = rp[10]; xr[5] = rp[11]; xr[6] = rp[12];
  xr[7] = rp[3] + rp[6] * t; xr[8] = rp[4] + rp[7] * t; xr[9] = rp[5] + rp[8] * t;
  xr[10] = 0.0; xr[11] = 0.0; xr[12] = 0.0;
}

// cone rows a_i' = (C_mat R)_i  (QuatMpc.cpp:47-52,203): (1,0,-mu),(-1,0,-mu),(0,1,-mu),(0,-1,-mu),(0,0,1),(0,0,-1) times R
QL_FN void cone_rows(const DevParams& P, const double rot[9], double cr[18]) {
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    const double m = P.mu * rot[6 + a];
    cr[a] = rot[a] - m;
    cr[3 + a] = -rot[a] - m;
    cr[6 + a] = rot[3 + a] - m;
    cr[9 + a] = -rot[3 + a] - m;
    cr[12 + a] = rot[6 + a];
    cr[15 + a] = -rot[6 + a];
  }
}
// slack residual of row i at the initial guess u = u_ref (the same at every knot and stance point): c0 + max(-c0, 1)
QL_FN void initial_rows(const DevParams& P, const double cr[18], double uz, double s0[6], double rc0[6]) {
#pragma unroll
  for (int i = 0; i < 6; ++i) {
    double c0 = cr[3 * i + 2] * uz;
    if (i == 4) c0 += -P.fz_max;
    s0[i] = fmax(-c0, 1.0);
    rc0[i] = c0 + s0[i];
  }
}

// ---- one contact point at one knot: barrier weights, rotated frame, factorised 3 x 3 block ---------------------------
// (the arithmetic of rotation_prepass in qmpc_kernels.hip, followed by the L D L' factorisation that the Gauss-Jordan
// pivots 3l, 3l+1, 3l+2 perform there)
struct LegBlk {
  double T[9];                  // frame, T[3a+b] = (q_b)_a
  double l10, l20, l21;         // unit lower factor of Db = T' D_l T
  double id0, id1, id2;         // inverse pivots
  double gq[3];                 // T' g_l
  double is[6];                 // 1 / s_i (the directions of pass C divide by the slacks again)
};
// Rw: the point's three input weights when the CALLER has them (the pair forms, where the point index differs between the
// partner lanes: indexing P.R per lane is a vector load -- in flight together with the prefetches, and waiting for it is waiting
// for them); null: P.R[3 (l mod 4) ..] with a wave-uniform l (scalar loads)
QL_FN void leg_block(const DevParams& P, const double cr[18], const double rc0[6], int l, const double sv[6],
                     const double lv[6], unsigned kap, double rho, double target, const double u[3], double uz, LegBlk& o,
                     const double* Rw = nullptr) {
  double w[6], gi[6];
#pragma unroll
  for (int i = 0; i < 6; ++i) {
    const double is = ql_rcp(sv[i]);
    o.is[i] = is;
    const double rc = ql_rounded(rho * rc0[i]);
    w[i] = lv[i] * is;
    gi[i] = (target + lv[i] * rc) * is - (((kap >> i) & 1u) ? lv[i] : 0.0);
  }
  // heaviest row i1, second heaviest non-(anti)parallel row i2 (rows 4,5 are antiparallel)
  int i1 = 0;
  double w1 = w[0];
#pragma unroll
  for (int i = 1; i < 6; ++i) if (w[i] > w1) { w1 = w[i]; i1 = i; }
  int i2 = -1;
  double w2 = -1.0;
#pragma unroll
  for (int i = 0; i < 6; ++i) {
    const bool skip = (i == i1) || ((i1 >= 4) && (i >= 4));
    if (!skip && w[i] > w2) { w2 = w[i]; i2 = i; }
  }
  double a1[3] = {0, 0, 0}, a2[3] = {0, 0, 0};
#pragma unroll
  for (int i = 0; i < 6; ++i)
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      a1[a] = (i == i1) ? cr[3 * i + a] : a1[a];
      a2[a] = (i == i2) ? cr[3 * i + a] : a2[a];
    }
  const double in1 = ql_rsqrt(a1[0] * a1[0] + a1[1] * a1[1] + a1[2] * a1[2]);
  double q1[3], q2[3], q3[3];
#pragma unroll
  for (int a = 0; a < 3; ++a) q1[a] = a1[a] * in1;
#pragma unroll
  for (int pass = 0; pass < 2; ++pass) {
    const double dp = a2[0] * q1[0] + a2[1] * q1[1] + a2[2] * q1[2];
#pragma unroll
    for (int a = 0; a < 3; ++a) a2[a] -= dp * q1[a];
  }
  const double in2 = ql_rsqrt(a2[0] * a2[0] + a2[1] * a2[1] + a2[2] * a2[2]);
#pragma unroll
  for (int a = 0; a < 3; ++a) q2[a] = a2[a] * in2;
  q3[0] = q1[1] * q2[2] - q1[2] * q2[1];
  q3[1] = q1[2] * q2[0] - q1[0] * q2[2];
  q3[2] = q1[0] * q2[1] - q1[1] * q2[0];
#pragma unroll
  for (int a = 0; a < 3; ++a) { o.T[3 * a] = q1[a]; o.T[3 * a + 1] = q2[a]; o.T[3 * a + 2] = q3[a]; }
  const double* T = o.T;
  const int l4 = l & 3;     // R holds 12 weights: input j uses R[j % 12]
  const double Rl[3] = {Rw ? Rw[0] : P.R[3 * l4], Rw ? Rw[1] : P.R[3 * l4 + 1], Rw ? Rw[2] : P.R[3 * l4 + 2]};
  const double ru[3] = {Rl[0] * u[0], Rl[1] * u[1], Rl[2] * (u[2] - uz)};
  // Db (upper triangle: 00 01 02 11 12 22) and the rotated gradient
  double d00, d01, d02, d11, d12, d22;
  {
    double tr[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) tr[i] = T[i] * Rl[i / 3];
    d00 = tr[0] * T[0] + tr[3] * T[3] + tr[6] * T[6];
    d01 = tr[0] * T[1] + tr[3] * T[4] + tr[6] * T[7];
    d02 = tr[0] * T[2] + tr[3] * T[5] + tr[6] * T[8];
    d11 = tr[1] * T[1] + tr[4] * T[4] + tr[7] * T[7];
    d12 = tr[1] * T[2] + tr[4] * T[5] + tr[7] * T[8];
    d22 = tr[2] * T[2] + tr[5] * T[5] + tr[8] * T[8];
  }
#pragma unroll
  for (int a = 0; a < 3; ++a) o.gq[a] = T[a] * ru[0] + T[3 + a] * ru[1] + T[6 + a] * ru[2];
#pragma unroll
  for (int i = 0; i < 6; ++i) {
    double at[3];  // rotated row
#pragma unroll
    for (int b = 0; b < 3; ++b) at[b] = T[b] * cr[3 * i] + T[3 + b] * cr[3 * i + 1] + T[6 + b] * cr[3 * i + 2];
#pragma unroll
    for (int a = 0; a < 3; ++a) o.gq[a] += gi[i] * at[a];
    const double w0 = w[i] * at[0], w1_ = w[i] * at[1];
    d00 += w0 * at[0]; d01 += w0 * at[1]; d02 += w0 * at[2];
    d11 += w1_ * at[1]; d12 += w1_ * at[2];
    d22 += w[i] * at[2] * at[2];
  }
  // L D L' in the pivot order of the frame (heaviest direction first)
  o.id0 = ql_rcp(d00);
  o.l10 = d01 * o.id0;
  o.l20 = d02 * o.id0;
  const double e11 = d11 - o.l10 * d01;
  o.id1 = ql_rcp(e11);
  const double e21 = d12 - o.l20 * d01;
  o.l21 = e21 * o.id1;
  const double e22 = d22 - o.l20 * d02 - o.l21 * e21;
  o.id2 = ql_rcp(e22);
}

// load the six slack / multiplier pairs of contact point l at knot k; the Tapia flag rides in the sign of the slack
template <int NL>
QL_FN unsigned load_rows(const Ctx& c, const WsOff& O, int k, int l, double sv[6], double lv[6]) {
  unsigned kap = 0;
#pragma unroll
  for (int i = 0; i < 6; ++i) {
    const double s = c.W(O.S + 6 * NL * k + 6 * l + i);
    lv[i] = c.W(O.LAM + 6 * NL * k + 6 * l + i);
    kap |= (s < 0.0) ? (1u << i) : 0u;
    sv[i] = fabs(s);
  }
  return kap;
}

// The backward pass has no registers for the rows of all contact points: it keeps ONE point ahead.  While contact
// point l of knot k is worked on, the rows of the NEXT stance point in processing order (the next point of the same
// knot, or the first point of knot k-1) are already on their way.  `order` is the wave-uniform list of points that any
// lane has in stance; the address of the next point is a scalar computation.
// WARM: the launch is warm-started -- the rows' initial slack residuals travel with the rows while any lane's rho is not 0.
// The passes exist in both forms: the six extra registers per point in flight cost the cold path 10-13 % otherwise.
template <bool WARM>
struct LegAheadT {
  static constexpr bool kHasRc = true;
  double u[3], du[3], s[6], lam[6], foot[3];
  double rc[6];
};
template <>
struct LegAheadT<false> {
  static constexpr bool kHasRc = false;
  double u[3], du[3], s[6], lam[6], foot[3];
  static constexpr double rc[6] = {0, 0, 0, 0, 0, 0};
};
template <class RT>
QL_FN void fetch_foot(const Ctx& c, FootPtr fp, int l, RT& R) {
  if (kFootAhead)
#pragma unroll
    for (int a = 0; a < 3; ++a) R.foot[a] = (c.foot_row >= 0) ? c.SR(c.foot_row + 3 * l + a) : fp[3 * l + a];
}
template <int NL, bool WITH_DU = false, class RT>
QL_FN void fetch_ahead(const Ctx& c, const WsOff& O, int k, int l, RT& R, FootPtr fp, bool rcrows = false) {     // k, l wave-uniform run-time values
  fetch_foot(c, fp, l, R);
  if constexpr (RT::kHasRc) {
    if (rcrows)
#pragma unroll
      for (int i = 0; i < 6; ++i) R.rc[i] = c.Ld(O.RC + 6 * NL * k + 6 * l + i);
  }
#pragma unroll
  for (int a = 0; a < 3; ++a) R.u[a] = c.Ld(O.U + 3 * NL * k + 3 * l + a);
  if (WITH_DU)
#pragma unroll
    for (int a = 0; a < 3; ++a) R.du[a] = c.Ld(O.dU + 3 * NL * k + 3 * l + a);
#pragma unroll
  for (int i = 0; i < 6; ++i) {
    R.s[i] = c.Ld(O.S + 6 * NL * k + 6 * l + i);
    R.lam[i] = c.Ld(O.LAM + 6 * NL * k + 6 * l + i);
  }
}
// wave-uniform stance mask: bit l set when any lane of the wavefront has contact point l in stance
template <int NL>
QL_FN unsigned any_stance(unsigned con) {
  unsigned m = 0;
#pragma unroll
  for (int l = 0; l < NL; ++l) m |= QL_ANY((con >> l) & 1u) ? (1u << l) : 0u;
  return m;
}
QL_FN int first_bit(unsigned m) {
  int l = 0;
  while (!((m >> l) & 1u)) ++l;
  return l;
}
// first stance point after l, or -1
QL_FN int next_bit(unsigned m, int l) {
  for (int j = l + 1; j < 32; ++j)
    if ((m >> j) & 1u) return j;
  return -1;
}

#ifndef QL_PF_CH      // old state and gains of pass C one knot ahead
#define QL_PF_CH 1
#endif
#ifndef QL_C_SPLIT       // pass C, plain form: the two points of a pair jointly in one basic block (0), one after the other, each
                         // re-fetching its rows after its own block (1), or both re-fetching after the second block (2: the first
                         // point's next rows are not in flight under the second point's temporaries -- no spill inside the loop)
#define QL_C_SPLIT 2
#endif
#ifndef QL_C_CR_PER_LEG  // ... rebuilding the cone rows per point instead of keeping 24 registers through the pass
#define QL_C_CR_PER_LEG 0
#endif
#ifndef QL_C_SPEC        // pass C, pair form: the sweep exists twice (both diagonal pairs in stance / one) instead of once with the
#define QL_C_SPEC 0      // second pair's block under a wave-uniform condition
#endif
#ifndef QL_A_KNOT_AHEAD  // pass A, plain form, four points: a buffer per point, fetched a knot ahead
#define QL_A_KNOT_AHEAD 1
#endif
#ifndef QL_A_PAIR_AHEAD  // pass A, pair form: per-round row buffers fetched a knot ahead
#define QL_A_PAIR_AHEAD 0
#endif
#ifndef QL_B_BPERM       // pass B, pair forms, step 5: the partner's z through ds_bpermute instead of swap + select
#define QL_B_BPERM 1
#endif
#ifndef QL_CAL_KLDS_PLAIN   // ... in the plain forms as well (the lanes' own slots of the LDS block)
#define QL_CAL_KLDS_PLAIN 1
#endif
#ifndef QL_CAL_KLDS      // reference mode's trial sweep, pair form: per-instance constants in LDS staging rows
#define QL_CAL_KLDS 30      // bits: 1 cone rows (off: with them read per knot the per-point block contracts its sums differently from the plain form), 2 gravity, 4 wd0, 8 contact points, 16 reference parameters
#endif
#ifndef QL_B_KLDS        // pass B, pair forms: the per-instance constants in LDS staging rows instead of private memory
#define QL_B_KLDS 1
#endif
#ifndef QL_B_PARK        // pass B, pair forms: p waits in LDS staging rows across the contact points
#define QL_B_PARK 0
#endif
#ifndef QL_B_XSTAGE      // pass B, pair forms: the knot's state through the LDS staging rows (global_load_lds)
#define QL_B_XSTAGE 1
#endif
#ifndef QL_B_COLSPLIT    // pass B, pair form: the twelve gain columns split between the partner lanes
#define QL_B_COLSPLIT 1
#endif
#ifndef QL_B_RW          // pair forms: the point's input weights from wave-uniform reads and a per-lane choice (1) or indexed per lane (0)
#define QL_B_RW 1
#endif
#ifndef QL_C_RW
#define QL_C_RW 1
#endif
#ifndef QL_C_FIRST_RT    // pass C, pair form: the first pair in stance runs unconditionally with a run-time pair index
#define QL_C_FIRST_RT 1
#endif
#ifndef QL_B_XAHEAD      // pass B: the state of the stage cost is requested before the column sweep
#define QL_B_XAHEAD 1
#endif
#ifndef QL_CR_PER_KNOT    // pass B rebuilds the cone rows per knot instead of keeping 24 registers through the factorisations
#define QL_CR_PER_KNOT 1
#endif

// ---- set-up: record -> constants, initial guess U = u_ref (QuatMpc.cpp:253), slacks and multipliers ------------------
// warm_launch (wave-uniform): the launch carries previous solutions; the rows' initial slack residuals are then kept per
// knot (O.RC) for EVERY lane of the launch.  u_prev: this instance's previous inputs [N][3 NL] or null (cold start of
// this instance).  The warm guess is the rule of qmpc_solve_body.inc: the previous solution shifted by one knot (the last
// knot repeats), swing points pinned to 0, a component that was 0 (the point has just landed) starts from u_ref.
template <int NL, int MD = MD_QUAT>
QL_FN void lane_setup(const DevParams& P, const Ctx& c, const WsOff& O, const double* rec, LaneK<NL>& K, LaneState& st,
                      bool warm_launch = false, const double* u_prev = nullptr) {
  typedef LDim<NL> D;
  const int N = P.N;
  bool bad = false;
  double raw[D::REC];
#pragma unroll
  for (int i = 0; i < D::REC; ++i) { raw[i] = rec[i]; bad = bad || !isfinite(raw[i]); }
  st.con = 0; st.nc = 0;
#pragma unroll
  for (int l = 0; l < NL; ++l) if (raw[(MD == MD_CONVEX ? 24 : D::R_CON) + l] != 0.0) { st.con |= 1u << l; st.nc++; }
  st.status = bad ? QMPC_NAN_INPUT : (st.nc == 0 ? QMPC_NO_CONTACT : QMPC_OK);
  st.iters = 0; st.it = 0;
  st.active = st.status == QMPC_OK;
  st.rho = 1.0; st.mu = 0.0; st.target = 0.0; st.last_ap = 0.0; st.last_ad = 0.0; st.last_step = 1e300;
  st.ap = 1.0; st.ad = 1.0; st.uz = 0.0; st.bad_step = 0; st.rcmax = 0.0;
  if (!st.active) return;
  if constexpr (MD == MD_CONVEX) {
    // qmpc_convex_input: euler 0..2, pos 3..5, ang_vel 6..8, lin_vel 9..11, foot 12..23, contacts 24..27, pos_d 28..30,
    // lin_vel_d 31..33, yaw_rate_d 34.  World-frame forces: the pyramid acts on them directly (rot = I), gravity is
    // (0,0,-9.81), no centre-of-mass torque
#pragma unroll
    for (int i = 0; i < 9; ++i) K.rot[i] = (i % 4 == 0) ? 1.0 : 0.0;
#pragma unroll
    for (int i = 0; i < 3 * NL; ++i) K.foot[i] = raw[12 + i];
#pragma unroll
    for (int a = 0; a < 3; ++a) K.wd0[a] = 0.0;
#pragma unroll
    for (int i = 0; i < 13; ++i) K.refp[i] = 0.0;
    K.refp[CVR_YAW] = raw[2]; K.refp[CVR_RATE] = raw[34];
    K.refp[CVR_POS] = raw[28]; K.refp[CVR_POS + 1] = raw[29]; K.refp[CVR_POS + 2] = raw[30];
    K.refp[CVR_VX] = raw[31]; K.refp[CVR_VY] = raw[32];
#pragma unroll
    for (int i = 0; i < 13; ++i) c.W(O.X + i) = (i < 12) ? raw[i] : 0.0;      // x_init, ConvexMpc.cpp:156-167
  } else {
#pragma unroll
  for (int i = 0; i < 9; ++i) K.rot[i] = raw[4 + i];
#pragma unroll
  for (int i = 0; i < 3 * NL; ++i) K.foot[i] = raw[D::R_FOOT + i];
  {
    // wd0 = Iinv (c x 5.204 g_body), g_body = R'(0,0,-9.81)  (AltroUtils.cpp:368-374,391)
    const double gb[3] = {raw[4 + 6] * (-9.81), raw[4 + 7] * (-9.81), raw[4 + 8] * (-9.81)};
    const double com[3] = {0.0223, 0.002, -0.0005};
    const double fg[3] = {5.204 * gb[0], 5.204 * gb[1], 5.204 * gb[2]};
    const double mg[3] = {com[1] * fg[2] - com[2] * fg[1], com[2] * fg[0] - com[0] * fg[2], com[0] * fg[1] - com[1] * fg[0]};
#pragma unroll
    for (int a = 0; a < 3; ++a) K.wd0[a] = P.Iinv[3 * a] * mg[0] + P.Iinv[3 * a + 1] * mg[1] + P.Iinv[3 * a + 2] * mg[2];
  }
#pragma unroll
  for (int i = 0; i < 13; ++i) K.refp[i] = (i < 9) ? raw[D::R_POS + i] : raw[D::R_QD + i - 9];
  // x_init (QuatMpc.cpp:231-246; the angular velocity is dropped by the ';' at :242 unless drop_ang_vel = 0)
  {
    double x0[13];
#pragma unroll
    for (int i = 0; i < 13; ++i) {
      double v = 0.0;
      if (i >= 3 && i < 7) v = raw[i - 3];
      else if (i >= 7 && i < 10) v = raw[13 + i - 7];
      else if (i >= 10) v = P.drop_ang_vel ? 0.0 : raw[16 + i - 10];
      x0[i] = v;
    }
#pragma unroll
    for (int i = 0; i < 13; ++i) c.W(O.X + i) = x0[i];
  }
  }
  // u_ref (QuatMpc.cpp:118-125): weight shared by the stance points; cone values there are the same at every knot
  st.uz = 1.0 * P.mass * 9.81 / (double)st.nc;
  double s0[6], l0[6];
  {
    double cr[18], rc0[6];
    cone_rows(P, K.rot, cr);
    initial_rows(P, cr, st.uz, s0, rc0);
    st.rcmax = 0.0;
#pragma unroll
    for (int i = 0; i < 6; ++i) { l0[i] = P.mu0 / s0[i]; st.rcmax = fmax(st.rcmax, fabs(rc0[i])); }
  }
  if (warm_launch) {
    double cr[18];
    cone_rows(P, K.rot, cr);
    double slsum = 0.0, rcmax = 0.0;
    for (int k = 0; k < N; ++k) {
      const int ks = (k + 1 < N) ? k + 1 : k;
#pragma unroll
      for (int l = 0; l < NL; ++l) {
        const bool on = (st.con >> l) & 1u;
        double u[3];
#pragma unroll
        for (int a = 0; a < 3; ++a) {
          const double ur = (a == 2) ? st.uz : 0.0;
          double v = ur;
          if (u_prev) {
            const double prev = u_prev[3 * NL * ks + 3 * l + a];
            v = (prev != 0.0) ? prev : ur;
          }
          u[a] = on ? v : 0.0;
          c.W(O.U + 3 * NL * k + 3 * l + a) = u[a];
        }
        if (on) {
#pragma unroll
          for (int i = 0; i < 6; ++i) {
            double c0 = cr[3 * i] * u[0] + cr[3 * i + 1] * u[1] + cr[3 * i + 2] * u[2];
            if (i == 4) c0 += -P.fz_max;
            const double sv = fmax(-c0, 1.0), lv = P.mu0 / sv, rc = c0 + sv;
            c.W(O.S + 6 * NL * k + 6 * l + i) = sv;
            c.W(O.LAM + 6 * NL * k + 6 * l + i) = lv;
            c.W(O.RC + 6 * NL * k + 6 * l + i) = rc;
            slsum += sv * lv;
            rcmax = fmax(rcmax, fabs(rc));
          }
        }
      }
    }
    st.mu = slsum / (double)(6 * N * st.nc);
    st.rcmax = rcmax;
    return;
  }
  double slsum = 0.0;
#pragma unroll
  for (int i = 0; i < 6; ++i) slsum += s0[i] * l0[i];
  for (int k = 0; k < N; ++k)
#pragma unroll
    for (int l = 0; l < NL; ++l) {
      const bool on = (st.con >> l) & 1u;
#pragma unroll
      for (int a = 0; a < 3; ++a) c.W(O.U + 3 * NL * k + 3 * l + a) = (on && a == 2) ? st.uz : 0.0;
      if (on) {
#pragma unroll
        for (int i = 0; i < 6; ++i) {
          c.W(O.S + 6 * NL * k + 6 * l + i) = s0[i];
          c.W(O.LAM + 6 * NL * k + 6 * l + i) = l0[i];
        }
      }
    }
  st.mu = slsum / 6.0;      // mean of s*lambda over the enabled rows (every stance point and knot carries the same six)
}

// ---- pass A: apply the pending step (none at the first iteration) and roll the states out open loop -----------------
// Slack / multiplier directions are recomputed from the stored trial increment dU exactly as pass C formed them (the
// cone rows are linear in u), so nothing but dU has to be kept between the passes.  A shortened primal step scales
// the increment; rc <- (1 - alpha_p) rc, exactly 0 after a full step.
// WARM: a warm-started launch -- the inputs of the first iteration are read (they are not u_ref), and the rows' initial
// slack residuals come from the workspace while any lane still carries a residual (rho != 0)
// PAIR (lane pairs, see Ctx and pass_B): the stance points of the wavefront, in ascending order, are taken two at a time -- the
// first of a round by the lower partner lane, the second by the upper -- so a lane applies the step to ONE point's rows per
// round (and has its next rows in flight for a whole round of the pair).  What the plain form accumulates in order is
// accumulated in the same order here: the sum of s * lambda is a chain of fused multiply-adds that runs over the lower lane's
// six rows, crosses to the upper lane (v_permlane32_swap), runs over its six rows and crosses back; force and torque sums
// take the two shares as (acc + first) + second.  A pair-mode launch returns the bits of a plain one.
template <int NL, bool WARM = false, int MD = MD_QUAT, bool PAIR = false>
QL_FN void pass_A(const DevParams& P, const Ctx& c, const WsOff& O, const LaneK<NL>& K, LaneState& st, bool first_in, FootPtr fp) {
  static_assert(!PAIR || (MD != MD_CONVEX && NL == 4), "pair split: the four-point quaternion model");
  constexpr bool warm = WARM;
#if QL_DEVICE
  // the lanes of a wavefront that are in this call are all at the same iteration: make the flag a scalar, so that the two
  // forms of the sweep are a wave-uniform branch and not two masked regions with memory operations in them
  const bool first = __builtin_amdgcn_readfirstlane((int)first_in) != 0;
#else
  const bool first = first_in;
#endif
  const int N = P.N;
  const double gb[3] = {K.rot[6] * (-9.81), K.rot[7] * (-9.81), K.rot[8] * (-9.81)};
  const double* wd0 = K.wd0;
  double cr[18], rc0[6];
  {
    double s0[6];
    cone_rows(P, K.rot, cr);
    initial_rows(P, cr, st.uz, s0, rc0);
  }
  double x[13], xn[13];
#pragma unroll
  for (int i = 0; i < 13; ++i) x[i] = c.W(O.X + i);
  const double ap = st.ap, ad = st.ad;
  const bool full = ap >= 1.0;
  const bool tapia = (ap >= 0.99) && (ad >= 0.99);
  double slsum = 0.0;
  const unsigned order = any_stance<NL>(st.con);
  const bool rcrows = WARM && QL_ANY(st.rho != 0.0);
  LegAheadT<WARM> R;         // rows, inputs and trial increments of the NEXT contact point in processing order (not at the first
                      // iteration: nothing is pending then and every input is at its reference)
  // Plain form of the four-point models: one buffer PER POINT, fetched a whole knot ahead (point l of knot k + 1 is requested
  // when point l of knot k has been copied out): the sweep has the registers (84 doubles) that the other two do not.  Worth
  // 1 % only: with 1024 full wavefronts streaming this sweep sits at the HBM roof (40 KB per wavefront and knot in ~8 k cycles
  // = 10 TB/s asked for), not on a latency (profiles/HISTORY_r06.md).
  constexpr bool kKnotAhead = QL_A_KNOT_AHEAD && NL == 4 && !PAIR;
  LegAheadT<WARM> Rk[kKnotAhead ? NL : 1];
  // Pair form: one buffer per ROUND (this lane's point of the round), fetched a whole knot ahead as well -- the pair form of this
  // sweep uses a fifth of the accumulation registers, and one round of it is shorter than a trip to HBM
  constexpr bool kPairAhead = QL_A_PAIR_AHEAD && NL == 4 && PAIR;
  LegAheadT<WARM> Rp[kPairAhead ? NL / 2 : 1];
  // pair form: the stance points of the wavefront in ascending order, four bits each (0xF: none)
  unsigned plist = 0xFFFFu;
  int pcount = 0;
  if (PAIR) {
    plist = 0;
    for (int l = NL - 1; l >= 0; --l)
      if ((order >> l) & 1u) { plist = (plist << 4) | (unsigned)l; ++pcount; }
    plist |= 0xFFFFu << (4 * pcount);
  }
  // this lane's point of round r (the upper partner's may not exist: it then shadows the lower one's point and adds nothing)
  auto pair_point = [&](int r, bool& exists) {
    const unsigned pa = (plist >> (8 * r)) & 0xFu, pb = (plist >> (8 * r + 4)) & 0xFu;
    exists = !(c.half && pb == 0xFu);
    return (int)((c.half && pb != 0xFu) ? pb : pa);
  };
  if constexpr (kKnotAhead) {
#pragma unroll
    for (int l = 0; l < NL; ++l) {
      if (!((order >> l) & 1u)) continue;       // wave-uniform
      if (!first) fetch_ahead<NL, true>(c, O, 0, l, Rk[l], fp, rcrows);
      else fetch_foot(c, fp, l, Rk[l]);
    }
  } else if constexpr (kPairAhead) {
#pragma unroll
    for (int rd = 0; rd < NL / 2; ++rd) {
      if (2 * rd >= pcount) continue;       // wave-uniform
      bool ex;
      if (!first) fetch_ahead<NL, true>(c, O, 0, pair_point(rd, ex), Rp[rd], fp, rcrows);
      else fetch_foot(c, fp, pair_point(rd, ex), Rp[rd]);
    }
  } else {
    bool ex;
    const int l0 = PAIR ? pair_point(0, ex) : first_bit(order);
    if (!first) fetch_ahead<NL, true>(c, O, 0, l0, R, fp, rcrows);
    else fetch_foot(c, fp, l0, R);
  }
  for (int k = 0; k < N; ++k) {
    const int kn = (k + 1 < N) ? k + 1 : k;
    double F[3] = {0, 0, 0}, wd[3] = {wd0[0], wd0[1], wd0[2]};
    if constexpr (PAIR) {
#pragma unroll
      for (int rd = 0; rd < NL / 2; ++rd) {
        if (2 * rd >= pcount) continue;       // wave-uniform
        bool exists;
        const int lm = pair_point(rd, exists);
        const bool on_m = exists && ((st.con >> lm) & 1u);
        const unsigned pa_ = (plist >> (8 * rd)) & 0xFu, pb_ = (plist >> (8 * rd + 4)) & 0xFu;
        const bool on_lo = (st.con >> pa_) & 1u, on_hi = pb_ != 0xFu && ((st.con >> (pb_ & 3u)) & 1u);
        const bool more = 2 * (rd + 1) < pcount;      // another round of this knot follows
        LegAheadT<WARM>& Rr = kPairAhead ? Rp[kPairAhead ? rd : 0] : R;      // this lane's rows of the round
        double u[3] = {0.0, 0.0, st.uz}, r[3];
#pragma unroll
        for (int a = 0; a < 3; ++a) r[a] = Rr.foot[a];
        double rcl[6];
#pragma unroll
        for (int i = 0; i < 6; ++i) rcl[i] = rcrows ? Rr.rc[i] : rc0[i];
        if (first) {
          bool ex;
          if (!kPairAhead) fetch_foot(c, fp, pair_point(more ? rd + 1 : 0, ex), Rr);      // (a round's own buffer keeps its position)
          if (warm)      // the warm guess (once per solve: read in place)
#pragma unroll
            for (int a = 0; a < 3; ++a) u[a] = c.W(O.U + 3 * NL * k + 3 * lm + a);
        } else {
          double du[3], sv[6], lv[6], so[6], lo[6], s1v[6], l1v[6];
          unsigned kap = 0;
#pragma unroll
          for (int a = 0; a < 3; ++a) { u[a] = Rr.u[a]; du[a] = Rr.du[a]; }
#pragma unroll
          for (int i = 0; i < 6; ++i) {
            lv[i] = Rr.lam[i];
            kap |= (Rr.s[i] < 0.0) ? (1u << i) : 0u;
            sv[i] = fabs(Rr.s[i]);
            so[i] = Rr.s[i];
            lo[i] = Rr.lam[i];
            s1v[i] = 0.0;
            l1v[i] = 0.0;
          }
          {     // this lane's next rows: its point of the next round of this knot, or of the first round of the next knot
            bool ex;
            if (kPairAhead) fetch_ahead<NL, true>(c, O, kn, lm, Rr, fp, rcrows);      // the same round, one knot on (the last knot re-reads itself)
            else fetch_ahead<NL, true>(c, O, more ? k : kn, pair_point(more ? rd + 1 : 0, ex), Rr, fp, rcrows);
          }
          if (on_m) {
#pragma unroll
            for (int i = 0; i < 6; ++i) {
              const double jd = cr[3 * i] * du[0] + cr[3 * i + 1] * du[1] + cr[3 * i + 2] * du[2];
              const double kp = ((kap >> i) & 1u) ? 1.0 : 0.0;
              const double dsv = -(jd + ql_rounded(st.rho * rcl[i]));
              const double dlv = (st.target - (1.0 + kp) * sv[i] * lv[i] - lv[i] * dsv) * ql_rcp(sv[i]);
              const double s1 = sv[i] + ap * dsv;
              const double l1 = lv[i] + ad * dlv;
              const bool sig = tapia && (s1 < 0.6 * sv[i]) && (l1 < 0.6 * lv[i]) &&
                               (kp != 0.0 || ((s1 > 0.4 * sv[i]) && (l1 > 0.4 * lv[i])));
              so[i] = sig ? -s1 : s1;
              lo[i] = l1;
              s1v[i] = s1;
              l1v[i] = l1;
            }
#pragma unroll
            for (int a = 0; a < 3; ++a) u[a] += full ? du[a] : ap * du[a];
          }
          // the plain form's chain of fused multiply-adds: over the first point's rows on the lower lane, then over the second
          // point's rows on the upper lane
          {
            double t = slsum, a_, b_;
#pragma unroll
            for (int i = 0; i < 6; ++i) t = fma(s1v[i], l1v[i], t);
            t = on_m ? t : slsum;
            ql_pair(t, a_, b_);
            slsum = a_;
            t = slsum;
#pragma unroll
            for (int i = 0; i < 6; ++i) t = fma(s1v[i], l1v[i], t);
            t = on_m ? t : slsum;
            ql_pair(t, a_, b_);
            slsum = b_;
          }
#pragma unroll
          for (int i = 0; i < 6; ++i) {
            c.StOwn(O.S + 6 * NL * k + 6 * lm + i, so[i]);
            c.StOwn(O.LAM + 6 * NL * k + 6 * lm + i, lo[i]);
          }
#pragma unroll
          for (int a = 0; a < 3; ++a) c.StOwn(O.U + 3 * NL * k + 3 * lm + a, u[a]);
        }
        {
          double B[9];
          leg_bw0(P, r, B);
#pragma unroll
          for (int a = 0; a < 3; ++a) {
            const double tq = B[3 * a] * u[0] + B[3 * a + 1] * u[1] + B[3 * a + 2] * u[2];
            double flo, fhi, tlo, thi;
            ql_pair(u[a], flo, fhi);
            ql_pair(tq, tlo, thi);
            if (on_lo) { F[a] += flo; wd[a] += tlo; }
            if (on_hi) { F[a] += fhi; wd[a] += thi; }
          }
        }
      }
    } else
#pragma unroll
    for (int l = 0; l < NL; ++l) {
      if (!((order >> l) & 1u)) continue;       // wave-uniform
      // No memory operation of this block sits under a per-lane condition: the wavefront's memory counter is in order, and a
      // wait for the prefetched rows is priced at the path with the FEWEST younger operations -- stores under `if (stance)`
      // (a branch the compiler lets a wavefront skip) made every such wait a wait for those stores as well.  A lane whose
      // point is not in stance stores back what it read.
      const bool on = (st.con >> l) & 1u;
      LegAheadT<WARM>& Rc = kKnotAhead ? Rk[kKnotAhead ? l : 0] : R;      // this point's rows
      double u[3] = {0.0, 0.0, st.uz}, r[3];
#pragma unroll
      for (int a = 0; a < 3; ++a) r[a] = kFootAhead ? Rc.foot[a] : K.foot[3 * l + a];
      double rcl[6];
#pragma unroll
      for (int i = 0; i < 6; ++i) rcl[i] = rcrows ? Rc.rc[i] : rc0[i];
      if (first) {
        const int ln = next_bit(order, l);
        if (!kKnotAhead) fetch_foot(c, fp, ln >= 0 ? ln : first_bit(order), R);      // (a point's own buffer keeps its position)
        if (warm)      // the warm guess (once per solve: read in place)
#pragma unroll
          for (int a = 0; a < 3; ++a) u[a] = c.W(O.U + 3 * NL * k + 3 * l + a);
      } else {
        double du[3], sv[6], lv[6], so[6], lo[6];      // so, lo: what goes back into the rows
        unsigned kap = 0;
#pragma unroll
        for (int a = 0; a < 3; ++a) { u[a] = Rc.u[a]; du[a] = Rc.du[a]; }
#pragma unroll
        for (int i = 0; i < 6; ++i) {
          lv[i] = Rc.lam[i];
          kap |= (Rc.s[i] < 0.0) ? (1u << i) : 0u;
          sv[i] = fabs(Rc.s[i]);
          so[i] = Rc.s[i];
          lo[i] = Rc.lam[i];
        }
        const int ln = next_bit(order, l);
        if (kKnotAhead) fetch_ahead<NL, true>(c, O, kn, l, Rc, fp, rcrows);      // the same point, one knot on (the last knot re-reads itself)
        else fetch_ahead<NL, true>(c, O, ln >= 0 ? k : kn, ln >= 0 ? ln : first_bit(order), R, fp, rcrows);
        if (on) {
#pragma unroll
          for (int i = 0; i < 6; ++i) {
            const double jd = cr[3 * i] * du[0] + cr[3 * i + 1] * du[1] + cr[3 * i + 2] * du[2];
            const double kp = ((kap >> i) & 1u) ? 1.0 : 0.0;
            const double dsv = -(jd + ql_rounded(st.rho * rcl[i]));
            const double dlv = (st.target - (1.0 + kp) * sv[i] * lv[i] - lv[i] * dsv) * ql_rcp(sv[i]);
            const double s1 = sv[i] + ap * dsv;
            const double l1 = lv[i] + ad * dlv;
            // Tapia indicators (see ipm_apply in qmpc_kernels.hip): both ratios near 1/2 on a full Newton step
            const bool sig = tapia && (s1 < 0.6 * sv[i]) && (l1 < 0.6 * lv[i]) &&
                             (kp != 0.0 || ((s1 > 0.4 * sv[i]) && (l1 > 0.4 * lv[i])));
            so[i] = sig ? -s1 : s1;
            lo[i] = l1;
            slsum = fma(s1, l1, slsum);      // (explicitly fused: the pair form runs the same chain across two lanes)
          }
#pragma unroll
          for (int a = 0; a < 3; ++a) u[a] += full ? du[a] : ap * du[a];
        }
#pragma unroll
        for (int i = 0; i < 6; ++i) {
          c.St(O.S + 6 * NL * k + 6 * l + i, so[i]);
          c.St(O.LAM + 6 * NL * k + 6 * l + i, lo[i]);
        }
#pragma unroll
        for (int a = 0; a < 3; ++a) c.St(O.U + 3 * NL * k + 3 * l + a, u[a]);
      }
      if (on) {
      if constexpr (MD == MD_CONVEX) {      // wd collects the torque sum
#pragma unroll
        for (int a = 0; a < 3; ++a) F[a] += u[a];
        cv_cross_acc(r, u, wd);
      } else {
      double B[9];
      leg_bw0(P, r, B);
#pragma unroll
      for (int a = 0; a < 3; ++a) {
        F[a] += u[a];
        wd[a] += B[3 * a] * u[0] + B[3 * a + 1] * u[1] + B[3 * a + 2] * u[2];
      }
      }
      }
    }
    if constexpr (MD == MD_CONVEX) cv_step_fw(P, x, F, wd, xn);
    else srbd_step_fw(P, gb, x, F, wd, xn);
#pragma unroll
    for (int i = 0; i < 13; ++i) { x[i] = xn[i]; if (PAIR) c.StOwn(O.X + 13 * (k + 1) + i, xn[i]); else c.St(O.X + 13 * (k + 1) + i, xn[i]); }
  }
  if (!first) {
    st.mu = slsum / (double)(6 * N * st.nc);
    st.rho = full ? 0.0 : (1.0 - ap) * st.rho;
  }
  QL_TICK(st, LP_A);
}

// ---- pass B: expansions + Riccati backward pass in the wrench form; writes the 6 x 13 gains -------------------------
// Per knot (k = N-1 .. 0), with P, p the cost-to-go of knot k+1 in registers:
//   1. contact points        G = sum_l V_l D_l^-1 V_l' (6 x 6),  r6 = sum_l V_l D_l^-1 g_l
//   2. dynamics expansion    A1 = Abar_phiphi, A3 = Abar_phiw, Wt = (h h / 4) Gn'Gm
//   3. S6 = M'PM straight from P;  S6 = L L',  H = I + L'GL = C C',  Z = L^-T (I - H^-1) L^-1 = Wr Quu^-1 Wr'
//   4. P <- Abar'P Abar, p <- Abar'p in place;  Y = M'P_old Abar = Mt'P with Mt = Abar^-1 M (only its attitude block
//      What = A1^-1 (Wt - h A3) differs from M), y = Mt'p, y' = y - S6 r6
//   5. column by column:  z_j = Z y_j,  gain column  xg_j = y_j - S6 z_j  (= (I + S6 G)^-1 y_j) stored at once,
//      P[i][j] -= y_i . z_j;  gradient column with y' and z + r6
//   6. stage cost of knot k (the state is re-read: keeping its expansion live through 3-5 would cost 18 registers)
// The order keeps at most P (90) + Y (78) + Z, S6 (42) + a dozen temporaries live -- the 512-register budget of a
// wave that owns its SIMD.  Returns false when S6 loses positive definiteness (QMPC_NOT_PD).
template <int NL, int MD = MD_QUAT>
QL_FN void cost_expansion(const DevParams& P, const Ctx& c, const WsOff& O, const LaneK<NL>& K, int k, double lx[12],
                          double lxx[6], const double* xpre = nullptr) {
  double x[13];
#pragma unroll
  for (int i = 0; i < 13; ++i) x[i] = xpre ? xpre[i] : c.W(O.X + 13 * k + i);
  if constexpr (MD == MD_CONVEX) {       // quadratic in the state; blocks reordered [p, phi, v, w]
    double xr[13];
    cv_xref_at(P, K.refp, k, xr);
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      lx[a] = P.Q[3 + a] * (x[3 + a] - xr[3 + a]);
      lx[3 + a] = P.Q[a] * (x[a] - xr[a]);
      lx[6 + a] = P.Q[9 + a] * (x[9 + a] - xr[9 + a]);
      lx[9 + a] = P.Q[6 + a] * (x[6 + a] - xr[6 + a]);
    }
    lxx[0] = P.Q[0]; lxx[1] = 0.0; lxx[2] = 0.0; lxx[3] = P.Q[1]; lxx[4] = 0.0; lxx[5] = P.Q[2];
    return;
  }
  double xr[13];
  xref_at(P, K.refp, k, xr);
  double lxf[13];
#pragma unroll
  for (int i = 0; i < 13; ++i) lxf[i] = P.Q[i] * (x[i] - xr[i]);
  const double dq = xr[3] * x[3] + xr[4] * x[4] + xr[5] * x[5] + xr[6] * x[6];
  const double sg = (dq >= 0.0) ? 1.0 : -1.0;
#pragma unroll
  for (int r = 0; r < 4; ++r) lxf[3 + r] += -sg * P.w * xr[3 + r];
  const double qh = -(x[3] * lxf[3] + x[4] * lxf[4] + x[5] * lxf[5] + x[6] * lxf[6]);
  double G[12];
  quatG(&x[3], G);
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    lx[a] = lxf[a];
    lx[6 + a] = lxf[7 + a];
    lx[9 + a] = lxf[10 + a];
    lx[3 + a] = G[a] * lxf[3] + G[3 + a] * lxf[4] + G[6 + a] * lxf[5] + G[9 + a] * lxf[6];
  }
  // attitude block (upper triangle 00 01 02 11 12 22)
  int q = 0;
#pragma unroll
  for (int a = 0; a < 3; ++a)
#pragma unroll
    for (int b = a; b < 3; ++b) {
      double s = (a == b) ? qh : 0.0;
#pragma unroll
      for (int t = 0; t < 4; ++t) s += G[3 * t + a] * P.Q[3 + t] * G[3 * t + b];
      lxx[q++] = s;
    }
}

// AL (reference mode, lane_iteration_ref below): the rows carry augmented-Lagrangian weights instead of barrier weights --
//   z_i = lam_i + rho c_i,  w_i = rho [z_i > 0],  g_i = max(z_i, 0)      (c_i = a_i . u + b_i from the input)
// fed to leg_block() as a slack of 1, a multiplier w_i and a residual z_i / rho (target 0) -- and the pass also returns the
// expected decrease of a full step, dV1 = sum_k d_k' Qu_k = -sum_k Qu_k' Quu_k^-1 Qu_k, through the wrench form:
//   Qu' Quu^-1 Qu = y0'G y0 + 2 y0'r6 + sum_l g_l' D_l^-1 g_l - q6' S6 (I + G S6)^-1 q6,   y0 = M'p, q6 = G y0 + r6.
struct LaneAL {
  double rho, irho;         // penalty and its reciprocal
  double J, Jp, viol;       // AL merit, plain objective and largest violation of the current trajectory
  double dV1;               // expected decrease of a full step (pass B)
  double alpha;             // step length of the line-search trial
  double Jn, Jnp, vn, stp;  // the trial's merit, plain objective, violation, largest input increment
  double Jn2, Jnp2, vn2, stp2;  // ... and those of the second trial of the same pass (step length alpha / 2)
  double stat;              // stationarity |grad_U L_A|_inf (pass S)
  int searching;            // this lane's line search is still running (its trial increments may be overwritten)
  int sel;                  // which of the pass's two trials was accepted: its increments are in the dU slot (0) or the RC slot (1)
};
// PAIR (lane pairs, see Ctx and pass_C): the stance points of the wavefront, in ascending order, are taken two at a time -- the
// first of a round by the lower partner lane, the second by the upper -- and each lane's contribution to wd, r6 and G (30
// doubles) reaches both partners through v_permlane32_swap, added in the plain form's order (first point, then second).
template <int NL, bool WARM = false, int MD = MD_QUAT, bool AL = false, bool PAIR = false>
QL_FN bool pass_B(const DevParams& P, const Ctx& c_in, const WsOff& O, const LaneK<NL>& K, LaneState& st, FootPtr fp,
                  LaneAL* al = nullptr) {
  static_assert(!PAIR || (MD != MD_CONVEX && NL == 4), "pair split: the four-point quaternion model");
  Ctx c = c_in;
  typedef LDim<NL> D;
  const int N = P.N;
  // where the position / velocity / angular-velocity weights sit in P.Q (the two models order their states differently)
  constexpr int QPo = (MD == MD_CONVEX) ? 3 : 0, QVo = (MD == MD_CONVEX) ? 9 : 7, QWo = (MD == MD_CONVEX) ? 6 : 10;
  double pv[12];      // cost-to-go  1/2 dx'P dx + p'dx: P in the lane-private rows c.PL(), p in registers
  double cr[18], rc0[6];
  // Pair forms: the per-instance constants of the pass (rotation, wd0, contact points: 24 values) wait in LDS staging rows
  // 25.. instead of private memory -- the register allocator keeps none of them across a knot, and what it re-reads from
  // scratch at the top of every knot comes back through the vector-memory counter, behind the sweep's stores and the prefetches
  constexpr bool kKLds = PAIR && QL_B_KLDS && QL_DEVICE;
  constexpr int kRotRow = 25, kWdRow = 34, kFootRow = 37;
  if constexpr (kKLds) {
#pragma unroll
    for (int i = 0; i < 9; ++i) c.SRst(kRotRow + i, K.rot[i]);
#pragma unroll
    for (int a = 0; a < 3; ++a) c.SRst(kWdRow + a, K.wd0[a]);
#pragma unroll
    for (int i = 0; i < 3 * NL; ++i) c.SRst(kFootRow + i, K.foot[i]);
    c.foot_row = kFootRow;
  }
  if (!QL_CR_PER_KNOT) {
    double s0[6];
    cone_rows(P, K.rot, cr);
    initial_rows(P, cr, st.uz, s0, rc0);
  }
  const unsigned order = any_stance<NL>(st.con);      // at least one bit: the lanes of this call have a stance point
  const bool rcrows = WARM && QL_ANY(st.rho != 0.0);
  LegAheadT<WARM> R;         // rows of the NEXT contact point in processing order
  // pair form: the stance points of the wavefront in ascending order, four bits each (0xF: none)
  unsigned plist = 0xFFFFu;
  int pcount = 0;
  if (PAIR) {
    plist = 0;
    for (int l = NL - 1; l >= 0; --l)
      if ((order >> l) & 1u) { plist = (plist << 4) | (unsigned)l; ++pcount; }
    plist |= 0xFFFFu << (4 * pcount);
  }
  // this lane's point of round r (the upper partner's may not exist: it then idles on the lower one's rows)
  auto pair_point = [&](int r, bool& exists) {
    const unsigned pa = (plist >> (8 * r)) & 0xFu, pb = (plist >> (8 * r + 4)) & 0xFu;
    exists = !(c.half && pb == 0xFu);
    return (int)((c.half && pb != 0xFu) ? pb : pa);
  };
  if (PAIR) { bool ex; fetch_ahead<NL>(c, O, N - 1, pair_point(0, ex), R, fp, rcrows); }
  else fetch_ahead<NL>(c, O, N - 1, first_bit(order), R, fp, rcrows);
  bool ok = true;
  double dV1 = 0.0;           // AL only
  const double m1 = P.h * (P.hh * (1.0 / P.mass)), m2 = P.h * (1.0 / P.mass);
  {
    double lx[12], lxx[6];
    cost_expansion<NL, MD>(P, c, O, K, N, lx, lxx);
#pragma unroll
    for (int i = 0; i < 78; ++i) c.PL(i) = 0.0;
    int q = 0;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      c.PL(SI(a, a)) = P.Q[QPo + a];
      c.PL(SI(6 + a, 6 + a)) = P.Q[QVo + a];
      c.PL(SI(9 + a, 9 + a)) = P.Q[QWo + a];
#pragma unroll
      for (int b = a; b < 3; ++b) c.PL(SI(3 + a, 3 + b)) = lxx[q++];
    }
#pragma unroll
    for (int i = 0; i < 12; ++i) pv[i] = lx[i];
  }
  // Pair forms: the knot's state for the stage cost (step 6) goes through the LDS staging rows 0..12, requested a whole knot
  // before its use (the plain cold form requests it into registers before the column sweep, kXAhead below)
  constexpr bool kXStage = PAIR && QL_B_XSTAGE && QL_DEVICE;
  // Pair forms: the gradient p of the cost-to-go (24 registers, idle from the end of a knot to step 4 of the next -- across the
  // contact points, where the pressure peaks) waits in the staging rows 13..24: what the register allocator would otherwise
  // park in scratch comes back through the vector-memory counter, i.e. behind every store and prefetch in flight
  constexpr bool kPark = PAIR && QL_B_PARK && QL_DEVICE;
  if constexpr (kPark)
#pragma unroll
    for (int i = 0; i < 12; ++i) c.SRst(13 + i, pv[i]);
  for (int k = N - 1; k >= 0; --k) {
    QL_FENCE();
    c.relane();
    QL_TICK(st, LP_B_HEAD);
    if constexpr (kXStage) {
      c.template stage<8>(O.X + 13 * k, 0);
      c.template stage<5>(O.X + 13 * k + 8, 8);
    }
    // ---- 1. contact points; wd for the expansion ----
    double G6[21], r6[6];
#pragma unroll
    for (int i = 0; i < 21; ++i) G6[i] = 0.0;
#pragma unroll
    for (int i = 0; i < 6; ++i) r6[i] = 0.0;
    double wd[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) wd[a] = kKLds ? c.SR(kWdRow + a) : K.wd0[a];
    double gam = 0.0;         // AL only: sum_l g_l' D_l^-1 g_l
    const int kn = (k > 0) ? k - 1 : 0;
    double Wk[4] = {0, 0, 0, 0};       // ConvexMpc's model: Iw^-1 at this knot's midpoint yaw
    if constexpr (MD == MD_CONVEX) cv_winv_mid(P, c.W(O.X + 13 * k + 2), c.W(O.X + 13 * k + 8), Wk);
    if (QL_CR_PER_KNOT) {      // rebuilt per knot: 24 registers that need not live through the factorisations
      double s0[6];
      if constexpr (kKLds) {
        double rotk[9];
#pragma unroll
        for (int i = 0; i < 9; ++i) rotk[i] = c.SR(kRotRow + i);
        cone_rows(P, rotk, cr);
      } else {
        cone_rows(P, K.rot, cr);
      }
      initial_rows(P, cr, st.uz, s0, rc0);
    }
    if (PAIR) {
#pragma unroll
      for (int rd = 0; rd < NL / 2; ++rd) {
        if (2 * rd >= pcount) continue;       // wave-uniform
        bool exists;
        const int lm = pair_point(rd, exists);
        const bool on_m = exists && ((st.con >> lm) & 1u);
        double u[3], sv[6], lv[6], r[3];
        unsigned kap = 0;
#pragma unroll
        for (int a = 0; a < 3; ++a) { u[a] = R.u[a]; r[a] = R.foot[a]; }
        double rcl[6];
#pragma unroll
        for (int i = 0; i < 6; ++i) rcl[i] = rcrows ? R.rc[i] : rc0[i];
#pragma unroll
        for (int i = 0; i < 6; ++i) {
          lv[i] = R.lam[i];
          kap |= (R.s[i] < 0.0) ? (1u << i) : 0u;
          sv[i] = fabs(R.s[i]);
        }
        if (2 * (rd + 1) < pcount) {     // the next round's rows into the registers just copied out
          bool ex;
          c.relane();
          fetch_ahead<NL>(c, O, k, pair_point(rd + 1, ex), R, fp, rcrows);
        }
        double wp[3] = {0, 0, 0}, r6p[6] = {0, 0, 0, 0, 0, 0}, G6p[21], gamp = 0.0;
#pragma unroll
        for (int i = 0; i < 21; ++i) G6p[i] = 0.0;
        {
          double B[9];
          leg_bw0(P, r, B);
#pragma unroll
          for (int a = 0; a < 3; ++a) wp[a] = B[3 * a] * u[0] + B[3 * a + 1] * u[1] + B[3 * a + 2] * u[2];
          LegBlk lb;
          // the weights of this lane's point from two wave-uniform reads (scalar loads) and a per-lane choice
          double Rw[3];
          {
            const unsigned qa = (plist >> (8 * rd)) & 3u, qb = (plist >> (8 * rd + 4)) & 3u;
            const bool second = c.half && ((plist >> (8 * rd + 4)) & 0xFu) != 0xFu;
#pragma unroll
            for (int a = 0; a < 3; ++a) Rw[a] = second ? ql_uniform(P.R[3 * qb + a]) : ql_uniform(P.R[3 * qa + a]);
          }
          if constexpr (AL) {      // augmented-Lagrangian weights instead of barrier weights (see the plain form below)
            kap = 0;
#pragma unroll
            for (int i = 0; i < 6; ++i) {
              double cv = cr[3 * i] * u[0] + cr[3 * i + 1] * u[1] + cr[3 * i + 2] * u[2];
              if (i == 4) cv += -P.fz_max;
              const double z = lv[i] + al->rho * cv;
              const bool act = z > 0.0;
              sv[i] = 1.0;
              rcl[i] = act ? z * al->irho : 0.0;
              lv[i] = act ? al->rho : 0.0;
            }
          }
          leg_block(P, cr, rcl, lm, sv, lv, kap, AL ? 1.0 : st.rho, AL ? 0.0 : st.target, u, st.uz, lb, QL_B_RW ? Rw : nullptr);
          double V[18];
#pragma unroll
          for (int i = 0; i < 9; ++i) V[i] = lb.T[i];
          mm(B, lb.T, &V[9]);
          const double y0 = lb.gq[0], y1 = lb.gq[1] - lb.l10 * y0, y2 = lb.gq[2] - lb.l20 * y0 - lb.l21 * y1;
          const double z0 = lb.id0 * y0, z1 = lb.id1 * y1, z2 = lb.id2 * y2;
          if constexpr (AL) gamp = y0 * z0 + y1 * z1 + y2 * z2;
          double v0[6], v1[6], v2[6];
#pragma unroll
          for (int i = 0; i < 6; ++i) {
            v0[i] = V[3 * i];
            v1[i] = V[3 * i + 1] - lb.l10 * v0[i];
            v2[i] = V[3 * i + 2] - lb.l20 * v0[i] - lb.l21 * v1[i];
            r6p[i] = v0[i] * z0 + v1[i] * z1 + v2[i] * z2;
          }
#pragma unroll
          for (int i = 0; i < 6; ++i) {
            const double a0 = lb.id0 * v0[i], a1 = lb.id1 * v1[i], a2 = lb.id2 * v2[i];
#pragma unroll
            for (int j = i; j < 6; ++j) G6p[S6I(i, j)] = a0 * v0[j] + a1 * v1[j] + a2 * v2[j];
          }
        }
        // both partners add the two points' shares in the plain form's order; a point that is not in stance contributes nothing
        const unsigned pb_ = (plist >> (8 * rd + 4)) & 0xFu;
        const bool on_lo = (st.con >> ((plist >> (8 * rd)) & 0xFu)) & 1u, on_hi = pb_ != 0xFu && ((st.con >> (pb_ & 3u)) & 1u);
        (void)on_m;
#pragma unroll
        for (int a = 0; a < 3; ++a) {
          double lo, hi;
          ql_pair(wp[a], lo, hi);
          if (on_lo) wd[a] += lo;
          if (on_hi) wd[a] += hi;
        }
        if constexpr (AL) {
          double lo, hi;
          ql_pair(gamp, lo, hi);
          if (on_lo) gam += lo;
          if (on_hi) gam += hi;
        }
#pragma unroll
        for (int i = 0; i < 6; ++i) {
          double lo, hi;
          ql_pair(r6p[i], lo, hi);
          if (on_lo) r6[i] += lo;
          if (on_hi) r6[i] += hi;
        }
#pragma unroll
        for (int i = 0; i < 21; ++i) {
          double lo, hi;
          ql_pair(G6p[i], lo, hi);
          if (on_lo) G6[i] += lo;
          if (on_hi) G6[i] += hi;
        }
      }
    } else
#pragma unroll
    for (int l = 0; l < NL; ++l) {
      if (!((order >> l) & 1u)) continue;       // wave-uniform
      double u[3], sv[6], lv[6], r[3];
      unsigned kap = 0;
#pragma unroll
      for (int a = 0; a < 3; ++a) { u[a] = R.u[a]; r[a] = kFootAhead ? R.foot[a] : K.foot[3 * l + a]; }
      double rcl[6];
#pragma unroll
      for (int i = 0; i < 6; ++i) rcl[i] = rcrows ? R.rc[i] : rc0[i];
#pragma unroll
      for (int i = 0; i < 6; ++i) {
        lv[i] = R.lam[i];
        kap |= (R.s[i] < 0.0) ? (1u << i) : 0u;
        sv[i] = fabs(R.s[i]);
      }
      {     // the next point's rows into the registers just copied out
        const int ln = next_bit(order, l);
        c.relane();
        // (the first point of the NEXT knot is fetched after the main phase: 18 doubles fewer live through it, and the
        // cost expansion and the head of the next knot are time enough for the rows to arrive)
        if (ln >= 0) fetch_ahead<NL>(c, O, k, ln, R, fp, rcrows);
      }
      if ((st.con >> l) & 1u) {
      double B[9];
      if constexpr (MD == MD_CONVEX) {
        cv_leg_bw0(Wk, r, B);          // the linearisation's per-point map; wd collects the raw torque sum for the expansion
        cv_cross_acc(r, u, wd);
      } else {
      leg_bw0(P, r, B);
#pragma unroll
      for (int a = 0; a < 3; ++a) wd[a] += B[3 * a] * u[0] + B[3 * a + 1] * u[1] + B[3 * a + 2] * u[2];
      }
      if constexpr (AL) {
        kap = 0;
#pragma unroll
        for (int i = 0; i < 6; ++i) {
          double cv = cr[3 * i] * u[0] + cr[3 * i + 1] * u[1] + cr[3 * i + 2] * u[2];
          if (i == 4) cv += -P.fz_max;
          const double z = lv[i] + al->rho * cv;
          const bool act = z > 0.0;
          sv[i] = 1.0;
          rcl[i] = act ? z * al->irho : 0.0;
          lv[i] = act ? al->rho : 0.0;
        }
      }
      LegBlk lb;
      leg_block(P, cr, rcl, l, sv, lv, kap, AL ? 1.0 : st.rho, AL ? 0.0 : st.target, u, st.uz, lb);
      // V = [T ; Bw0 T] (6 x 3), Vt = V L^-T (columns), G += sum_j id_j vt_j vt_j', r6 += sum_j vt_j id_j y_j, y = L^-1 gq
      double V[18];
#pragma unroll
      for (int i = 0; i < 9; ++i) V[i] = lb.T[i];
      mm(B, lb.T, &V[9]);
      const double y0 = lb.gq[0], y1 = lb.gq[1] - lb.l10 * y0, y2 = lb.gq[2] - lb.l20 * y0 - lb.l21 * y1;
      const double z0 = lb.id0 * y0, z1 = lb.id1 * y1, z2 = lb.id2 * y2;
      if constexpr (AL) gam += y0 * z0 + y1 * z1 + y2 * z2;
      double v0[6], v1[6], v2[6];
#pragma unroll
      for (int i = 0; i < 6; ++i) {
        v0[i] = V[3 * i];
        v1[i] = V[3 * i + 1] - lb.l10 * v0[i];
        v2[i] = V[3 * i + 2] - lb.l20 * v0[i] - lb.l21 * v1[i];
        r6[i] += v0[i] * z0 + v1[i] * z1 + v2[i] * z2;
      }
#pragma unroll
      for (int i = 0; i < 6; ++i) {
        const double a0 = lb.id0 * v0[i], a1 = lb.id1 * v1[i], a2 = lb.id2 * v2[i];
#pragma unroll
        for (int j = i; j < 6; ++j) G6[S6I(i, j)] += a0 * v0[j] + a1 * v1[j] + a2 * v2[j];
      }
      }
    }
    QL_FENCE();
    c.relane();
    QL_TICK(st, LP_B_LEGS);
    // ---- 2. dynamics expansion (AltroUtils.cpp:78-110,153-168 in compact form) ----
    double A1[9], A3[9], Wt[9];
    if constexpr (MD == MD_CONVEX) {
      double x[12];
#pragma unroll
      for (int i = 0; i < 12; ++i) x[i] = c.W(O.X + 13 * k + i);
      cv_expansion(P, x, wd, A1, A3, Wt);
    } else {
      double x[13], xn[4];
#pragma unroll
      for (int i = 3; i < 13; ++i) x[i] = c.W(O.X + 13 * k + i);
#pragma unroll
      for (int i = 0; i < 4; ++i) xn[i] = c.W(O.X + 13 * (k + 1) + 3 + i);
      double G0[12], Gm[12], Gn[12];
      quatG(&x[3], G0);
      double qm[4], wm[3];
#pragma unroll
      for (int r = 0; r < 4; ++r)
        qm[r] = x[3 + r] + P.hh * (0.5 * (G0[3 * r] * x[10] + G0[3 * r + 1] * x[11] + G0[3 * r + 2] * x[12]));
#pragma unroll
      for (int a = 0; a < 3; ++a) wm[a] = x[10 + a] + P.hh * wd[a];
      quatG(qm, Gm);
      quatG(xn, Gn);
#pragma unroll
      for (int cc = 0; cc < 3; ++cc) {
        double g[4], gm[4], t0[4], t1[4], t2[4], ag[4], aw[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) { g[r] = G0[3 * r + cc]; gm[r] = Gm[3 * r + cc]; }
        omega_mul(&x[10], g, t0);
#pragma unroll
        for (int r = 0; r < 4; ++r) t1[r] = g[r] + (0.5 * P.hh) * t0[r];
        omega_mul(wm, t1, t2);
        omega_mul(wm, g, t0);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          ag[r] = g[r] + P.hh * t2[r];
          aw[r] = P.hh * ((0.5 * P.hh) * t0[r] + gm[r]);
        }
#pragma unroll
        for (int r = 0; r < 3; ++r) {
          A1[3 * r + cc] = Gn[r] * ag[0] + Gn[3 + r] * ag[1] + Gn[6 + r] * ag[2] + Gn[9 + r] * ag[3];
          A3[3 * r + cc] = Gn[r] * aw[0] + Gn[3 + r] * aw[1] + Gn[6 + r] * aw[2] + Gn[9 + r] * aw[3];
          Wt[3 * r + cc] = ((0.5 * P.hh) * P.h) * (Gn[r] * gm[0] + Gn[3 + r] * gm[1] + Gn[6 + r] * gm[2] + Gn[9 + r] * gm[3]);
        }
      }
    }
    QL_FENCE();
    c.relane();
    QL_TICK(st, LP_B_EXPAND);
    double q6[6] = {0, 0, 0, 0, 0, 0}, ak = 0.0;      // AL only: q6 = G y0 + r6,  ak = y0'G y0 + 2 y0'r6 + gam
    if constexpr (kPark && AL)
#pragma unroll
      for (int i = 0; i < 12; ++i) pv[i] = c.SR(13 + i);
    if constexpr (AL) {
      double y0v[6];
#pragma unroll
      for (int a = 0; a < 3; ++a) {
        y0v[a] = m1 * pv[a] + m2 * pv[6 + a];
        y0v[3 + a] = Wt[a] * pv[3] + Wt[3 + a] * pv[4] + Wt[6 + a] * pv[5] + P.h * pv[9 + a];
      }
      ak = gam;
#pragma unroll
      for (int i = 0; i < 6; ++i) {
        double gy = 0.0;
#pragma unroll
        for (int t = 0; t < 6; ++t) gy += G6[S6I(i, t)] * y0v[t];
        q6[i] = gy + r6[i];
        ak += y0v[i] * (gy + 2.0 * r6[i]);
      }
    }
    // ---- 3. S6 = M'PM from the symmetric storage; factorisations; Z ----
    double S6[21], Z[21];
    {
      // ff = m1^2 Ppp + m1 m2 (Ppv + Pvp) + m2^2 Pvv
#pragma unroll
      for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int b = a; b < 3; ++b)
          S6[S6I(a, b)] = (m1 * m1) * c.PL(SI(a, b)) + (m1 * m2) * (c.PL(SI(a, 6 + b)) + c.PL(SI(6 + a, b))) + (m2 * m2) * c.PL(SI(6 + a, 6 + b));
      // ft = (m1 Ppf + m2 Pvf) Wt + h (m1 Ppw + m2 Pvw);  tt = Wt'Pff Wt + h (Wt'Pfw + Pwf Wt) + h^2 Pww
      double Fw[9];   // Pff Wt
#pragma unroll
      for (int a = 0; a < 3; ++a) {
        double q0 = m1 * c.PL(SI(a, 3)) + m2 * c.PL(SI(6 + a, 3)), q1 = m1 * c.PL(SI(a, 4)) + m2 * c.PL(SI(6 + a, 4)),
               q2 = m1 * c.PL(SI(a, 5)) + m2 * c.PL(SI(6 + a, 5));
#pragma unroll
        for (int b = 0; b < 3; ++b)
          S6[S6I(a, 3 + b)] = q0 * Wt[b] + q1 * Wt[3 + b] + q2 * Wt[6 + b] + P.h * (m1 * c.PL(SI(a, 9 + b)) + m2 * c.PL(SI(6 + a, 9 + b)));
#pragma unroll
        for (int b = 0; b < 3; ++b) Fw[3 * a + b] = c.PL(SI(3 + a, 3)) * Wt[b] + c.PL(SI(3 + a, 4)) * Wt[3 + b] + c.PL(SI(3 + a, 5)) * Wt[6 + b];
      }
#pragma unroll
      for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int b = a; b < 3; ++b) {
          double s = (P.h * P.h) * c.PL(SI(9 + a, 9 + b));
#pragma unroll
          for (int t = 0; t < 3; ++t)
            s += Wt[3 * t + a] * Fw[3 * t + b] + P.h * (Wt[3 * t + a] * c.PL(SI(3 + t, 9 + b)) + c.PL(SI(9 + a, 3 + t)) * Wt[3 * t + b]);
          S6[S6I(3 + a, 3 + b)] = s;
        }
      double L[21], iL[6];
#pragma unroll
      for (int j = 0; j < 6; ++j) {
        double d = S6[S6I(j, j)];
#pragma unroll
        for (int t = 0; t < j; ++t) d -= L[LI(j, t)] * L[LI(j, t)];
        ok = ok && (d > 0.0);
        const double inv = ql_rsqrt(d);
        iL[j] = inv;
        L[LI(j, j)] = d * inv;
#pragma unroll
        for (int i = j + 1; i < 6; ++i) {
          double s = S6[S6I(j, i)];
#pragma unroll
          for (int t = 0; t < j; ++t) s -= L[LI(i, t)] * L[LI(j, t)];
          L[LI(i, j)] = s * inv;
        }
      }
      // H = I + L' G L = C C'
      double Cc[21], iC[6];
      {
        double GL[6][6];
#pragma unroll
        for (int i = 0; i < 6; ++i)
#pragma unroll
          for (int j = 0; j < 6; ++j) {
            double s = 0.0;
#pragma unroll
            for (int t = j; t < 6; ++t) s += G6[S6I(i, t)] * L[LI(t, j)];
            GL[i][j] = s;
          }
        double H[21];
#pragma unroll
        for (int i = 0; i < 6; ++i)
#pragma unroll
          for (int j = i; j < 6; ++j) {
            double s = (i == j) ? 1.0 : 0.0;
#pragma unroll
            for (int t = i; t < 6; ++t) s += L[LI(t, i)] * GL[t][j];
            H[S6I(i, j)] = s;
          }
#pragma unroll
        for (int j = 0; j < 6; ++j) {
          double d = H[S6I(j, j)];
#pragma unroll
          for (int t = 0; t < j; ++t) d -= Cc[LI(j, t)] * Cc[LI(j, t)];
          const double inv = ql_rsqrt(d);
          iC[j] = inv;
          Cc[LI(j, j)] = d * inv;
#pragma unroll
          for (int i = j + 1; i < 6; ++i) {
            double s = H[S6I(j, i)];
#pragma unroll
            for (int t = 0; t < j; ++t) s -= Cc[LI(i, t)] * Cc[LI(j, t)];
            Cc[LI(i, j)] = s * inv;
          }
        }
      }
      // Ci = C^-1 (lower), E = I - Ci'Ci;  Li = L^-1 (lower), Z = Li' E Li
      double Ci[21], Li[21];
#pragma unroll
      for (int j = 0; j < 6; ++j) {
        Ci[LI(j, j)] = iC[j];
        Li[LI(j, j)] = iL[j];
#pragma unroll
        for (int i = j + 1; i < 6; ++i) {
          double s = 0.0, s2 = 0.0;
#pragma unroll
          for (int t = j; t < i; ++t) { s -= Cc[LI(i, t)] * Ci[LI(t, j)]; s2 -= L[LI(i, t)] * Li[LI(t, j)]; }
          Ci[LI(i, j)] = s * iC[i];
          Li[LI(i, j)] = s2 * iL[i];
        }
      }
      double E[21];
#pragma unroll
      for (int i = 0; i < 6; ++i)
#pragma unroll
        for (int j = i; j < 6; ++j) {
          double s = (i == j) ? 1.0 : 0.0;
#pragma unroll
          for (int t = j; t < 6; ++t) s -= Ci[LI(t, i)] * Ci[LI(t, j)];
          E[S6I(i, j)] = s;
        }
      double EL[6][6];   // E Li (Li lower: column j has rows >= j)
#pragma unroll
      for (int i = 0; i < 6; ++i)
#pragma unroll
        for (int j = 0; j < 6; ++j) {
          double s = 0.0;
#pragma unroll
          for (int t = j; t < 6; ++t) s += E[S6I(i, t)] * Li[LI(t, j)];
          EL[i][j] = s;
        }
#pragma unroll
      for (int i = 0; i < 6; ++i)
#pragma unroll
        for (int j = i; j < 6; ++j) {
          double s = 0.0;
#pragma unroll
          for (int t = i; t < 6; ++t) s += Li[LI(t, i)] * EL[t][j];
          Z[S6I(i, j)] = s;
        }
    }
    QL_FENCE();
    c.relane();
    QL_TICK(st, LP_B_FACT);
    // ---- 4. P <- Abar' P Abar, p <- Abar' p  in place on the symmetric storage; What = A1^-1 (Wt - h A3) ----
    if constexpr (kPark && !AL)
#pragma unroll
      for (int i = 0; i < 12; ++i) pv[i] = c.SR(13 + i);
    {
      double F[9], W[9], t1[9], t2[9], t3[9], Rb[9];
      rdblk(c, 1, 1, F);
      rdblk(c, 1, 3, W);
      mm(F, A3, t1);            // P_ff A3
      mtm(A3, W, t2);           // A3' P_fw
      rdblk(c, 3, 3, Rb);
      mtm(A3, t1, t3);
#pragma unroll
      for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int cc = 0; cc < 3; ++cc) Rb[3 * r + cc] += t3[3 * r + cc] + t2[3 * r + cc] + t2[3 * cc + r];
      wrblk(c, 3, 3, Rb);
#pragma unroll
      for (int i = 0; i < 9; ++i) W[i] += t1[i];
      mtm(A1, W, Rb);
      wrblk(c, 1, 3, Rb);      // A1'(P_ff A3 + P_fw)
      mm(F, A1, t3);
      mtm(A1, t3, Rb);
      wrblk(c, 1, 1, Rb);      // A1' P_ff A1
    }
    {
      double Bq[9], Vq[9], t4[9], Cq[9], Dq[9], Rb[9];
      rdblk(c, 0, 1, Bq);      // P_pf
      rdblk(c, 1, 2, Vq);      // P_fv
#pragma unroll
      for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int cc = 0; cc < 3; ++cc) t4[3 * r + cc] = P.h * Bq[3 * cc + r] + Vq[3 * r + cc];
      rdblk(c, 0, 3, Cq);      // P_pw
      rdblk(c, 2, 3, Dq);      // P_vw
      mtm(t4, A3, Rb);
#pragma unroll
      for (int i = 0; i < 9; ++i) Dq[i] += P.h * Cq[i] + Rb[i];
      wrblk(c, 2, 3, Dq);
      mtm(A1, t4, Rb);
      wrblk(c, 1, 2, Rb);
      mm(Bq, A3, Rb);
#pragma unroll
      for (int i = 0; i < 9; ++i) Cq[i] += Rb[i];
      wrblk(c, 0, 3, Cq);
      mm(Bq, A1, Rb);
      wrblk(c, 0, 1, Rb);
    }
    {
      double E[9], Vv[9], Rb[9];
      rdblk(c, 0, 0, E);
      rdblk(c, 0, 2, Vv);
      rdblk(c, 2, 2, Rb);
#pragma unroll
      for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int cc = 0; cc < 3; ++cc) Rb[3 * r + cc] += P.h * (Vv[3 * r + cc] + Vv[3 * cc + r]) + (P.h * P.h) * E[3 * r + cc];
      wrblk(c, 2, 2, Rb);
#pragma unroll
      for (int i = 0; i < 9; ++i) Vv[i] += P.h * E[i];
      wrblk(c, 0, 2, Vv);
    }
    {
      const double f0 = pv[3], f1 = pv[4], f2 = pv[5];
#pragma unroll
      for (int a = 0; a < 3; ++a) {
        pv[9 + a] += A3[a] * f0 + A3[3 + a] * f1 + A3[6 + a] * f2;
        pv[6 + a] += P.h * pv[a];
      }
#pragma unroll
      for (int a = 0; a < 3; ++a) pv[3 + a] = A1[a] * f0 + A1[3 + a] * f1 + A1[6 + a] * f2;
    }
    double Wh[9];       // A1^-1 (Wt - h A3) by cofactors (A1 = I + O(h |w|))
    {
      const double c00 = A1[4] * A1[8] - A1[5] * A1[7], c01 = A1[5] * A1[6] - A1[3] * A1[8], c02 = A1[3] * A1[7] - A1[4] * A1[6];
      const double idet = ql_rcp(A1[0] * c00 + A1[1] * c01 + A1[2] * c02);
      const double Ai[9] = {c00 * idet, (A1[2] * A1[7] - A1[1] * A1[8]) * idet, (A1[1] * A1[5] - A1[2] * A1[4]) * idet,
                            c01 * idet, (A1[0] * A1[8] - A1[2] * A1[6]) * idet, (A1[2] * A1[3] - A1[0] * A1[5]) * idet,
                            c02 * idet, (A1[1] * A1[6] - A1[0] * A1[7]) * idet, (A1[0] * A1[4] - A1[1] * A1[3]) * idet};
      double Dm[9];
#pragma unroll
      for (int i = 0; i < 9; ++i) Dm[i] = Wt[i] - P.h * A3[i];
      mm(Ai, Dm, Wh);
    }
    QL_FENCE();
    c.relane();
    QL_TICK(st, LP_B_CONGR);
    // ---- 5. Y = Mt' P (rows f: mf P_p. + m2 P_v. ; rows t: What' P_f. + h P_w.), column by column ----
    // Every sum of products of this step is written as an explicit chain: which product of `a b + c d` the compiler contracts
    // depends on the instantiation, and the forms of this pass have to return each other's bits.
    //
    // Pair form (QL_B_COLSPLIT): the step is SPLIT between the partner lanes.  The lower lane owns the columns 0..5 of Y / of the
    // gains / of the cost-to-go's upper-left block, the upper lane the columns 6..11 and the lower-right block: each forms only
    // ITS six columns Y_h (36 registers instead of 72), solves them, stores their gains and updates its diagonal block.  The
    // off-diagonal block P(i, 6 + j) -= Y_i' Z Y_(6+j) needs one column of each lane: after every round the lanes swap the z
    // they just formed (v_permlane32_swap), and the entry is computed where the swap brought its operands together -- by the lower
    // lane as Y_i . z_(6+j) when i is in S(j), by the upper lane as z_i . Y_(6+j) otherwise (S: a tournament on 0..5, kCross
    // below, so that both lanes run the same row indices in the same instruction).  The plain form computes every entry by the
    // same expression -- the (i, 6 + j) with i outside S(j) while it holds z_i, i.e. in column i.
    const double mf = m1 - P.h * m2;
    constexpr bool kColSplit = PAIR && QL_B_COLSPLIT;
    constexpr int NY = kColSplit ? 6 : 12;
    double Y[NY][6];
    auto y_col = [&](const double (&pc)[12], double (&y)[6]) {      // column of Y from a column of P
#pragma unroll
      for (int a = 0; a < 3; ++a) {
        y[a] = fma(m2, pc[6 + a], mf * pc[a]);
        y[3 + a] = fma(P.h, pc[9 + a], fma(Wh[6 + a], pc[5], fma(Wh[3 + a], pc[4], Wh[a] * pc[3])));
      }
    };
#pragma unroll
    for (int j = 0; j < NY; ++j) {
      double pc[12];
#pragma unroll
      for (int r = 0; r < 12; ++r) pc[r] = kColSplit ? c.PL(c.half ? SI(r, (j + 6) % 12) : SI(r, j)) : c.PL(SI(r, j));
      y_col(pc, Y[j]);
    }
    double yg[6];     // y' = Mt'p - S6 r6
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      yg[a] = fma(m2, pv[6 + a], mf * pv[a]);
      yg[3 + a] = fma(P.h, pv[9 + a], fma(Wh[6 + a], pv[5], fma(Wh[3 + a], pv[4], Wh[a] * pv[3])));
    }
#pragma unroll
    for (int i = 0; i < 6; ++i) {
      double s = yg[i];
#pragma unroll
      for (int t = 0; t < 6; ++t) s = fma(-S6[S6I(i, t)], r6[t], s);
      yg[i] = s;
    }
    QL_FENCE();
    c.relane();
    QL_TICK(st, LP_B_MP);
    // the knot's state for the stage cost (step 6), requested before the column sweep: the registers of the point rows are
    // free here (the next knot's first point is fetched after the sweep), and read right behind that fetch the state cost a
    // full memory latency per knot.  Plain cold form only: the pair-split and the warm instantiations have no 26 registers
    // to spare (they spill 30 .. 120 B with it)
    constexpr bool kXAhead = QL_B_XAHEAD && !PAIR && !WARM && !AL;
    double xk[13];
    if (kXAhead)
#pragma unroll
      for (int i = 0; i < 13; ++i) xk[i] = c.W(O.X + 13 * k + i);
    if constexpr (kXStage) c.staged();      // (issued a knot's work ago: nothing to wait for; before the sweep's stores, which would be)
    auto dot6 = [](const double (&u)[6], const double (&v)[6]) {
      double s = u[0] * v[0];
#pragma unroll
      for (int t = 1; t < 6; ++t) s = fma(u[t], v[t], s);
      return s;
    };
    auto col_solve = [&](const double (&yj)[6], double (&z)[6], double (&xg)[6]) {      // z = Z y, xg = y - S6 z
#pragma unroll
      for (int i = 0; i < 6; ++i) {
        double s = Z[S6I(i, 0)] * yj[0];
#pragma unroll
        for (int t = 1; t < 6; ++t) s = fma(Z[S6I(i, t)], yj[t], s);
        z[i] = s;
      }
#pragma unroll
      for (int i = 0; i < 6; ++i) {
        double s = yj[i];
#pragma unroll
        for (int t = 0; t < 6; ++t) s = fma(-S6[S6I(i, t)], z[t], s);
        xg[i] = s;
      }
    };
    if constexpr (kColSplit) {
#pragma unroll
      for (int p = 0; p < 6; ++p) {
        double z[6], xg[6];
        c.relane();
        col_solve(Y[p], z, xg);
        if (AL) {      // double precision: columns 0..5 in the slot (lower lane), 6..11 in the second block (upper lane)
          const int base = c.half ? O.G2 + D::GAIN2 * k + 6 * p : O.G + D::GAIN * k + 6 * p;
#pragma unroll
          for (int i = 0; i < 6; ++i) c.StOwn(base + i, xg[i]);
        } else {
          const int base = O.G + D::GAIN * k + 3 * (c.half ? p + 6 : p);
#pragma unroll
          for (int i = 0; i < 3; ++i) c.StOwn(base + i, pack2f((float)xg[2 * i], (float)xg[2 * i + 1]));
        }
        // this lane's diagonal block: P(i, p) on the lower lane, P(6 + i, 6 + p) on the upper
#pragma unroll
        for (int i = 0; i <= p; ++i) {
          const int e = c.half ? SI(6 + i, 6 + p) : SI(i, p);
          c.PL(e) -= dot6(Y[i], z);
        }
        // the partner's z: z_(6+p) on the lower lane, z_p on the upper
        double zp[6];
        if (QL_B_BPERM) {
          const unsigned pa = ql_partner_addr();
#pragma unroll
          for (int t = 0; t < 6; ++t) zp[t] = ql_partner(z[t], pa);
        } else {
#pragma unroll
          for (int t = 0; t < 6; ++t) {
            double lo, hi;
            ql_pair(z[t], lo, hi);
            zp[t] = c.half ? lo : hi;
          }
        }
        // off-diagonal block: P(q, 6 + p) -= Y_q . z_(6+p) on the lower lane, P(p, 6 + q) -= Y_(6+q) . z_p on the upper, q in S(p);
        // the entry (p, 6 + p) is the lower lane's
#pragma unroll
        for (int q = 0; q < 6; ++q) {
          if (!kCross(q, p)) continue;
          const double dq = dot6(Y[q], zp);
          if (q == p) {
            if (!c.half) c.PL(SI(p, 6 + p)) -= dq;
          } else {
            const int e = c.half ? SI(p, 6 + q) : SI(q, 6 + p);
            c.PL(e) -= dq;
          }
        }
      }
    }
#pragma unroll
    for (int j = kColSplit ? 12 : 0; j < 13; ++j) {
      double yj[6], z[6], xg[6];
      c.relane();
      constexpr int kY0 = kColSplit ? 0 : 1;      // (the plain form's column index, 0 where the split form never reads Y[j])
#pragma unroll
      for (int i = 0; i < 6; ++i) yj[i] = (j < 12) ? Y[j < 12 ? j * kY0 : 0][i] : yg[i];
      col_solve(yj, z, xg);
      if (AL && j < 12) {      // double precision: columns 0..5 in the slot, 6..11 in the second block
#pragma unroll
        for (int i = 0; i < 6; ++i) c.W((j < 6 ? O.G + D::GAIN * k + 6 * (j < 6 ? j : 0) : O.G2 + D::GAIN2 * k + 6 * (j >= 6 && j < 12 ? j - 6 : 0)) + i) = xg[i];
      } else if (j < 12) {
#pragma unroll
        for (int i = 0; i < 3; ++i) { const double pk = pack2f((float)xg[2 * i], (float)xg[2 * i + 1]); if (PAIR) c.StOwn(O.G + D::GAIN * k + 3 * (j < 12 ? j : 0) + i, pk); else c.St(O.G + D::GAIN * k + 3 * (j < 12 ? j : 0) + i, pk); }
      } else {
#pragma unroll
        for (int i = 0; i < 6; ++i) { if (PAIR) c.StOwn(O.G + D::GAIN * k + 36 + i, xg[i]); else c.St(O.G + D::GAIN * k + 36 + i, xg[i]); }
      }
      if (j < 12) {
        if constexpr (!kColSplit) {
          // rows of the column: all of its own block; of the off-diagonal block (j >= 6) those with i in S(j - 6) ...
#pragma unroll
          for (int i = 0; i <= j; ++i) {
            if (j >= 6 && i < 6 && !kCross(i, j - 6)) continue;
            c.PL(SI(i, j < 12 ? j : 0)) -= dot6(Y[i], z);
          }
          // ... and, while z_j of a column j < 6 is at hand, the entries (j, 6 + q) of the others: z_j . Y_(6+q), q in S(j), q != j
          if (j < 6) {
#pragma unroll
            for (int q = 0; q < 6; ++q) {
              if (!kCross(q, j < 6 ? j : 0) || q == j) continue;
              c.PL(SI(j < 6 ? j : 0, 6 + q)) -= dot6(Y[(6 + q) * kY0], z);
            }
          }
        }
      } else {
#pragma unroll
        for (int t = 0; t < 6; ++t) z[t] += r6[t];
        if constexpr (AL) {      // z + r6 = (I + G S6)^-1 q6
          double t2 = 0.0;
#pragma unroll
          for (int i = 0; i < 6; ++i) {
            double sw = S6[S6I(i, 0)] * z[0];
#pragma unroll
            for (int t = 1; t < 6; ++t) sw = fma(S6[S6I(i, t)], z[t], sw);
            t2 = fma(q6[i], sw, t2);
          }
          dV1 -= ak - t2;
        }
        if constexpr (kColSplit) {      // each lane its six rows of p, then both hold all twelve again
#pragma unroll
          for (int i = 0; i < 6; ++i) {
            double own = c.half ? pv[6 + i] : pv[i];
            own -= dot6(Y[i], z);
            ql_pair(own, pv[i], pv[6 + i]);
          }
        } else {
#pragma unroll
          for (int i = 0; i < 12; ++i) pv[i] -= dot6(Y[i * kY0], z);
        }
      }
    }
    QL_FENCE();
    c.relane();
    QL_TICK(st, LP_B_UPD);
    if (PAIR) { bool ex; fetch_ahead<NL>(c, O, kn, pair_point(0, ex), R, fp, rcrows); }
    else fetch_ahead<NL>(c, O, kn, first_bit(order), R, fp, rcrows);
    // ---- 6. stage cost of knot k ----
    {
      double lx[12], lxx[6];
      if constexpr (kXStage)
#pragma unroll
        for (int i = 0; i < 13; ++i) xk[i] = c.SR(i);
      cost_expansion<NL, MD>(P, c, O, K, k, lx, lxx, (kXAhead || kXStage) ? xk : nullptr);
      int q = 0;
#pragma unroll
      for (int a = 0; a < 3; ++a) {
        c.PL(SI(a, a)) += P.Q[QPo + a];
        c.PL(SI(6 + a, 6 + a)) += P.Q[QVo + a];
        c.PL(SI(9 + a, 9 + a)) += P.Q[QWo + a];
#pragma unroll
        for (int b = a; b < 3; ++b) c.PL(SI(3 + a, 3 + b)) += lxx[q++];
      }
#pragma unroll
      for (int i = 0; i < 12; ++i) pv[i] += lx[i];
    }
    if constexpr (kPark)
#pragma unroll
      for (int i = 0; i < 12; ++i) c.SRst(13 + i, pv[i]);
    QL_FENCE();
    c.relane();
    QL_TICK(st, LP_B_GAIN);
  }
  if constexpr (AL) al->dV1 = dV1;
  return ok;
}

// contact points are worked on in PAIRS inside one basic block, so that the two independent dependency chains (weights
// -> frame -> 3 x 3 factorisation: three reciprocals and two reciprocal square roots in series) interleave in the single
// wavefront of a SIMD.  Diagonal pairs for four legs -- a trotting robot has exactly one pair in stance -- and the two
// halves of a foot edge for the 8-point model.
template <int NL>
constexpr int pair_leg(int pr, int j) { return NL == 4 ? (j == 0 ? pr : 3 - pr) : 2 * pr + j; }

// pass C's work on one contact point, free of memory operations and branches (the caller discards the result of a point
// that is not in stance)
struct LegOutC {
  double du[3], u[3], B[9];
  double r[3];                // the point's position (ConvexMpc's model: the rollout wants the raw torque r x u)
  double rp, dn, dd, stp;     // largest -ds_i / s_i; the row with the largest -dlam_i / lam_i as numerator / denominator
  bool bad;                   // a component of du is not finite (fmax / fmin drop NaNs silently)
};
template <int NL, int MD = MD_QUAT, class RT>
QL_FN void leg_compute_C(const DevParams& P, const LaneK<NL>& K, const double cr[18], const double rc0_[6], const RT& R,
                         int l, const double zeta[6], const LaneState& st, LegOutC& o, bool rcrows, const double* Wk = nullptr,
                         const double* Rw = nullptr) {
  double rc0[6];
#pragma unroll
  for (int i = 0; i < 6; ++i) rc0[i] = rcrows ? R.rc[i] : rc0_[i];
  double u[3], r[3];
#pragma unroll
  for (int a = 0; a < 3; ++a) { u[a] = R.u[a]; r[a] = kFootAhead ? R.foot[a] : K.foot[3 * l + a]; }
  if constexpr (MD == MD_CONVEX) {
    cv_leg_bw0(Wk, r, o.B);
#pragma unroll
    for (int a = 0; a < 3; ++a) o.r[a] = r[a];
  } else
  leg_bw0(P, r, o.B);
  const double* B = o.B;
  double sv[6], lv[6];
  unsigned kap = 0;
#pragma unroll
  for (int i = 0; i < 6; ++i) {
    lv[i] = R.lam[i];
    kap |= (R.s[i] < 0.0) ? (1u << i) : 0u;
    sv[i] = fabs(R.s[i]);
  }
  LegBlk lb;
  leg_block(P, cr, rc0, l, sv, lv, kap, st.rho, st.target, u, st.uz, lb, Rw);
  // rhs = T'(zeta_f + Bw0' zeta_t) + gq;  du = -T Db^-1 rhs
  double t[3], rh[3];
#pragma unroll
  for (int a = 0; a < 3; ++a) t[a] = zeta[a] + B[a] * zeta[3] + B[3 + a] * zeta[4] + B[6 + a] * zeta[5];
#pragma unroll
  for (int a = 0; a < 3; ++a) rh[a] = lb.T[a] * t[0] + lb.T[3 + a] * t[1] + lb.T[6 + a] * t[2] + lb.gq[a];
  const double y0 = rh[0], y1 = rh[1] - lb.l10 * y0, y2 = rh[2] - lb.l20 * y0 - lb.l21 * y1;
  const double z2 = y2 * lb.id2;
  const double z1 = y1 * lb.id1 - lb.l21 * z2;
  const double z0 = y0 * lb.id0 - lb.l10 * z1 - lb.l20 * z2;
  double stp = 0.0;
  bool bad = false;
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    o.du[a] = -(lb.T[3 * a] * z0 + lb.T[3 * a + 1] * z1 + lb.T[3 * a + 2] * z2);
    stp = fmax(stp, fabs(o.du[a]));
    bad = bad || !(fabs(o.du[a]) <= 1e300);
    o.u[a] = u[a] + o.du[a];
  }
  o.bad = bad;
  // directions and fraction-to-the-boundary ratios (ipm_directions in qmpc_kernels.hip)
  // The step lengths are min(1, tau / max_i(-ds_i / s_i)) and min(1, tau / max_i(-dlam_i / lam_i)): the maxima are
  // tracked without a division per row (1 / s_i is at hand; the multiplier ratios are compared by cross-multiplication)
  // and the two divisions happen once, at the end of the pass.
  double rp = 0.0, dn = 0.0, dd = 1.0;
#pragma unroll
  for (int i = 0; i < 6; ++i) {
    const double jd = cr[3 * i] * o.du[0] + cr[3 * i + 1] * o.du[1] + cr[3 * i + 2] * o.du[2];
    const double kp = ((kap >> i) & 1u) ? 1.0 : 0.0;
    const double dsv = -(jd + ql_rounded(st.rho * rc0[i]));
    const double dlv = (st.target - (1.0 + kp) * sv[i] * lv[i] - lv[i] * dsv) * lb.is[i];
    rp = fmax(rp, -dsv * lb.is[i]);
    const bool better = (-dlv) * dd > dn * lv[i];       // -dlam_i / lam_i > dn / dd   (lam_i, dd > 0)
    dn = better ? -dlv : dn;
    dd = better ? lv[i] : dd;
  }
  o.rp = rp; o.dn = dn; o.dd = dd; o.stp = stp;
}

// ---- pass C: closed-loop trial rollout (alpha = 1) + slack / multiplier directions + step lengths ----------------------
// PAIR (lane pairs, see Ctx): the two points of a pair are worked on by the two partner lanes -- one leg_compute_C per lane
// instead of two -- and what the state step needs of both (force and torque sums) is exchanged per knot; the step-length
// candidates are combined once, at the end of the pass.  Both partner lanes add the two shares in the plain form's order,
// (F + a) + b, and scan the two points' multiplier-ratio candidates in its order too (a running maximum compared by
// cross-multiplication picks among near-equal rows by position): a pair-mode launch returns the bits of a plain one.
template <int NL, bool WARM = false, int MD = MD_QUAT, bool PAIR = false>
QL_FN void pass_C(const DevParams& P, const Ctx& c, const WsOff& O, const LaneK<NL>& K, LaneState& st, FootPtr fp) {
  static_assert(!PAIR || MD != MD_CONVEX, "pair split: the quaternion model");
  typedef LDim<NL> D;
  const int N = P.N;
  const double gb[3] = {K.rot[6] * (-9.81), K.rot[7] * (-9.81), K.rot[8] * (-9.81)};
  const double* wd0 = K.wd0;
  double cr[18], rc0[6];
  {
    double s0[6];
    cone_rows(P, K.rot, cr);
    initial_rows(P, cr, st.uz, s0, rc0);
  }
  double xc[13], xn[13];
#pragma unroll
  for (int i = 0; i < 13; ++i) xc[i] = c.W(O.X + i);
  // one knot ahead: old state, gains, rows of the contact points
  double xo[13], gn[D::GAIN];
  // the pairs of contact points that any lane has in stance, and the rows of the NEXT such pair in processing order
  unsigned porder = 0;
  {
    const unsigned order = any_stance<NL>(st.con);
#pragma unroll
    for (int pr = 0; pr < NL / 2; ++pr)
      porder |= (((order >> pair_leg<NL>(pr, 0)) | (order >> pair_leg<NL>(pr, 1))) & 1u) << pr;
  }
  const bool rcrows = WARM && QL_ANY(st.rho != 0.0);
  LegAheadT<WARM> Ra, Rb;
  if (QL_PF_CH) {
#pragma unroll
    for (int i = 0; i < 13; ++i) xo[i] = xc[i];
#pragma unroll
    for (int i = 0; i < D::GAIN; ++i) gn[i] = c.W(O.G + i);
  }
  // The loads above are COMPLETE before the loop is entered: the compiler prices the wait for the gains at the head of a knot
  // on the entry path too, where three of them were the youngest loads in flight -- which made it a wait for everything in
  // flight (vmcnt(0)) at the head of EVERY knot, the rows' prefetch and the increments' stores included.  One exposed latency
  // per sweep instead of one per knot.
  QL_WAIT_VMEM();
  {
    const int p0 = first_bit(porder);
    if (PAIR) {
      fetch_ahead<NL>(c, O, 0, c.half ? pair_leg<NL>(p0, 1) : pair_leg<NL>(p0, 0), Ra, fp, rcrows);
    } else {
      fetch_ahead<NL>(c, O, 0, NL == 4 ? p0 : 2 * p0, Ra, fp, rcrows);
      fetch_ahead<NL>(c, O, 0, NL == 4 ? 3 - p0 : 2 * p0 + 1, Rb, fp, rcrows);
    }
  }
  double rp = 0.0, dnA = 0.0, ddA = 1.0, dnB = 0.0, ddB = 1.0, stp = 0.0;      // (A: first points of the pairs, B: second points)
  bool bad = false;
  // The sweep over the knots exists TWICE in the pair form -- for wavefronts with both diagonal pairs in stance and for those with
  // one (run-time pair index) -- so that a knot is straight-line code.  With the second pair's block under a wave-uniform `if`
  // inside ONE loop (i) the rows prefetched for the next knot had to be in the same registers at the join whichever path was
  // taken -- the compiler put copies on the skipping path, and a copy is a use: a trot wavefront waited for its prefetch right
  // after issuing it -- and (ii) the wait for the gains at the head of the next knot was priced on the path that skips every
  // block, where the gains' own loads are the youngest in flight: a wait for everything (vmcnt(0)) at every head.
  const int pf = first_bit(porder);
  auto sweep = [&](auto both_c) {
  constexpr bool BOTH = decltype(both_c)::value;
  for (int k = 0; k < N; ++k) {
    const int kn = (k + 1 < N) ? k + 1 : k;      // the last knot re-reads itself
    // dx = xc (-) X_k in error coordinates (inverse Cayley map, QuaternionUtils.cpp:16-18)
    double dx[12];
    if (!QL_PF_CH) {
#pragma unroll
      for (int i = 0; i < 13; ++i) xo[i] = c.W(O.X + 13 * k + i);
#pragma unroll
      for (int i = 0; i < D::GAIN; ++i) gn[i] = c.W(O.G + D::GAIN * k + i);
    }
    double Wk[4] = {0, 0, 0, 0};       // ConvexMpc's model: Iw^-1 at the OLD knot state's midpoint yaw (the linearisation point)
    if constexpr (MD == MD_CONVEX) {
      cv_winv_mid(P, xo[2], xo[8], Wk);
#pragma unroll
      for (int a = 0; a < 3; ++a) {
        dx[a] = xc[3 + a] - xo[3 + a];
        dx[3 + a] = xc[a] - xo[a];
        dx[6 + a] = xc[9 + a] - xo[9 + a];
        dx[9 + a] = xc[6 + a] - xo[6 + a];
      }
    } else {
#pragma unroll
      for (int a = 0; a < 3; ++a) {
        dx[a] = xc[a] - xo[a];
        dx[6 + a] = xc[7 + a] - xo[7 + a];
        dx[9 + a] = xc[10 + a] - xo[10 + a];
      }
      double G[12];
      quatG(&xo[3], G);
      const double isc = ql_rcp(xo[3] * xc[3] + xo[4] * xc[4] + xo[5] * xc[5] + xo[6] * xc[6]);
#pragma unroll
      for (int a = 0; a < 3; ++a) dx[3 + a] = (G[a] * xc[3] + G[3 + a] * xc[4] + G[6 + a] * xc[5] + G[9 + a] * xc[6]) * isc;
    }
    double zeta[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) zeta[i] = gn[36 + i];
#pragma unroll
    for (int j = 0; j < 12; ++j)
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        float g0, g1;
        unpack2f(gn[3 * j + i], g0, g1);
        zeta[2 * i] += (double)g0 * dx[j];
        zeta[2 * i + 1] += (double)g1 * dx[j];
      }
    // next knot's old state and gains into the registers just consumed
    if (QL_PF_CH) {
#pragma unroll
      for (int i = 0; i < 13; ++i) xo[i] = c.W(O.X + 13 * kn + i);
#pragma unroll
      for (int i = 0; i < D::GAIN; ++i) gn[i] = c.W(O.G + D::GAIN * kn + i);
      // ... all of them HERE: left to itself the scheduler sinks the six feed-forward entries below the per-point blocks (their
      // registers double as zeta), where they are the youngest loads in flight at the next head -- a wait for them is a wait
      // for everything (vmcnt(0)): the rows' prefetch and the increments' stores included
      QL_SCHED_BARRIER();
    }
    QL_TICK(st, LP_C_HEAD);
    double F[3] = {0, 0, 0}, wd[3] = {wd0[0], wd0[1], wd0[2]};
    auto pair_blk = [&](const int pr) {
      const int la = pair_leg<NL>(pr, 0), lb = pair_leg<NL>(pr, 1);
      const bool on_a = (st.con >> la) & 1u, on_b = (st.con >> lb) & 1u;
      const int lm = c.half ? lb : la;      // this lane's point of the pair
      const bool on_m = c.half ? on_b : on_a;
      LegOutC om;
      double Rw[3];      // this lane's point's weights: two wave-uniform reads and a per-lane choice (see leg_block)
#pragma unroll
      for (int a = 0; a < 3; ++a) Rw[a] = c.half ? ql_uniform(P.R[3 * (lb & 3) + a]) : ql_uniform(P.R[3 * (la & 3) + a]);
      leg_compute_C<NL, MD>(P, K, cr, rc0, Ra, lm, zeta, st, om, rcrows, Wk, QL_C_RW ? Rw : nullptr);
      {     // the next pair's row into the registers just consumed
        const int pn = next_bit(porder, pr);
        const int kq = pn >= 0 ? k : kn, pq = pn >= 0 ? pn : first_bit(porder);
        fetch_ahead<NL>(c, O, kq, c.half ? pair_leg<NL>(pq, 1) : pair_leg<NL>(pq, 0), Ra, fp, rcrows);
      }
      double fm[3], tm[3];
#pragma unroll
      for (int a = 0; a < 3; ++a) {
        c.StOwn(O.dU + 3 * NL * k + 3 * lm + a, om.du[a]);      // every lane: no store under a per-lane condition (see pass A)
        fm[a] = on_m ? om.u[a] : 0.0;
        const double t = om.B[3 * a] * om.u[0] + om.B[3 * a + 1] * om.u[1] + om.B[3 * a + 2] * om.u[2];
        tm[a] = on_m ? t : 0.0;
      }
#pragma unroll
      for (int a = 0; a < 3; ++a) {
        double lo, hi;
        ql_pair(fm[a], lo, hi);      // lo: the first point's share, hi: the second's -- added in the plain form's order
        F[a] = (F[a] + lo) + hi;
        ql_pair(tm[a], lo, hi);
        wd[a] = (wd[a] + lo) + hi;
      }
      if (on_m) { rp = fmax(rp, om.rp); stp = fmax(stp, om.stp); bad = bad || om.bad; }
      {     // the multiplier ratio is a running maximum compared by cross-multiplication: its winner among near-equal rows
            // depends on the order of the scan, so both partners scan the two points' candidates in the plain form's order
        double alo, ahi, blo, bhi;
        ql_pair(on_m ? om.dn : 0.0, alo, ahi);
        ql_pair(on_m ? om.dd : 1.0, blo, bhi);
        { const bool better = alo * ddA > dnA * blo; dnA = better ? alo : dnA; ddA = better ? blo : ddA; }
        { const bool better = ahi * ddA > dnA * bhi; dnA = better ? ahi : dnA; ddA = better ? bhi : ddA; }
      }
    };
    if constexpr (PAIR) {
      if (BOTH) { pair_blk(0); pair_blk(1); }
      else if (QL_C_FIRST_RT) {
        pair_blk(pf);
        if (!QL_C_SPEC)
#pragma unroll
          for (int pr = 1; pr < NL / 2; ++pr)
            if (pr > pf && ((porder >> pr) & 1u)) pair_blk(pr);       // wave-uniform
      } else {
#pragma unroll
        for (int pr = 0; pr < NL / 2; ++pr)
          if ((porder >> pr) & 1u) pair_blk(pr);       // wave-uniform
      }
    } else
#pragma unroll
    for (int pr = 0; pr < NL / 2; ++pr) {
      const int la = pair_leg<NL>(pr, 0), lb = pair_leg<NL>(pr, 1);
      const bool on_a = (st.con >> la) & 1u, on_b = (st.con >> lb) & 1u;
      if (PAIR) {
      } else
      if ((porder >> pr) & 1u) {       // wave-uniform
#if QL_C_SPLIT
        // one point at a time: compute, fetch the point's NEXT rows into the registers just consumed, use the result -- so that
        // neither the two points' temporaries nor their results are live together (the joint form needs 140 ... 360 B of scratch
        // per lane, and a scratch re-load behind the prefetches waits for them: the memory counter is in order).  The sums are
        // formed in the joint form's order: first point, then second.
        const int pn = next_bit(porder, pr);
        const int kq = pn >= 0 ? k : kn, pq = pn >= 0 ? pn : first_bit(porder);
#pragma unroll
        for (int hb = 0; hb < 2; ++hb) {
          const int lh = hb ? lb : la;
          const bool on_h = hb ? on_b : on_a;
          LegOutC oh;
          if (QL_C_CR_PER_LEG) {
            double s0[6];
            cone_rows(P, K.rot, cr);
            initial_rows(P, cr, st.uz, s0, rc0);
          }
          if (hb) {
            leg_compute_C<NL, MD>(P, K, cr, rc0, Rb, lh, zeta, st, oh, rcrows, Wk);
#if QL_C_SPLIT == 2
            fetch_ahead<NL>(c, O, kq, NL == 4 ? pq : 2 * pq, Ra, fp, rcrows);
#endif
            fetch_ahead<NL>(c, O, kq, NL == 4 ? 3 - pq : 2 * pq + 1, Rb, fp, rcrows);
          } else {
            leg_compute_C<NL, MD>(P, K, cr, rc0, Ra, lh, zeta, st, oh, rcrows, Wk);
#if QL_C_SPLIT != 2
            fetch_ahead<NL>(c, O, kq, NL == 4 ? pq : 2 * pq, Ra, fp, rcrows);
#endif
          }
#pragma unroll
          for (int a = 0; a < 3; ++a) c.St(O.dU + 3 * NL * k + 3 * lh + a, oh.du[a]);      // every lane: no store under a per-lane condition (see pass A)
          if (on_h) {
#pragma unroll
            for (int a = 0; a < 3; ++a) {
              F[a] += oh.u[a];
              if constexpr (MD != MD_CONVEX) wd[a] += oh.B[3 * a] * oh.u[0] + oh.B[3 * a + 1] * oh.u[1] + oh.B[3 * a + 2] * oh.u[2];
            }
            if constexpr (MD == MD_CONVEX) cv_cross_acc(oh.r, oh.u, wd);
            rp = fmax(rp, oh.rp); stp = fmax(stp, oh.stp); bad = bad || oh.bad;
            { const bool better = oh.dn * ddA > dnA * oh.dd; dnA = better ? oh.dn : dnA; ddA = better ? oh.dd : ddA; }      // (one running ratio in the plain form)
          }
          QL_SCHED_BARRIER();
        }
#else
        LegOutC oa, ob;
        leg_compute_C<NL, MD>(P, K, cr, rc0, Ra, la, zeta, st, oa, rcrows, Wk);
        leg_compute_C<NL, MD>(P, K, cr, rc0, Rb, lb, zeta, st, ob, rcrows, Wk);
        {     // the next pair's rows into the registers just consumed
          const int pn = next_bit(porder, pr);
          const int kq = pn >= 0 ? k : kn, pq = pn >= 0 ? pn : first_bit(porder);
          fetch_ahead<NL>(c, O, kq, NL == 4 ? pq : 2 * pq, Ra, fp, rcrows);
          fetch_ahead<NL>(c, O, kq, NL == 4 ? 3 - pq : 2 * pq + 1, Rb, fp, rcrows);
        }
#pragma unroll
        for (int a = 0; a < 3; ++a) {      // every lane: no store under a per-lane condition (see pass A)
          c.St(O.dU + 3 * NL * k + 3 * la + a, oa.du[a]);
          c.St(O.dU + 3 * NL * k + 3 * lb + a, ob.du[a]);
        }
        if (on_a) {
#pragma unroll
          for (int a = 0; a < 3; ++a) {
            F[a] += oa.u[a];
            if constexpr (MD != MD_CONVEX) wd[a] += oa.B[3 * a] * oa.u[0] + oa.B[3 * a + 1] * oa.u[1] + oa.B[3 * a + 2] * oa.u[2];
          }
          if constexpr (MD == MD_CONVEX) cv_cross_acc(oa.r, oa.u, wd);
          rp = fmax(rp, oa.rp); stp = fmax(stp, oa.stp); bad = bad || oa.bad;
          { const bool better = oa.dn * ddA > dnA * oa.dd; dnA = better ? oa.dn : dnA; ddA = better ? oa.dd : ddA; }
        }
        if (on_b) {
#pragma unroll
          for (int a = 0; a < 3; ++a) {
            F[a] += ob.u[a];
            if constexpr (MD != MD_CONVEX) wd[a] += ob.B[3 * a] * ob.u[0] + ob.B[3 * a + 1] * ob.u[1] + ob.B[3 * a + 2] * ob.u[2];
          }
          if constexpr (MD == MD_CONVEX) cv_cross_acc(ob.r, ob.u, wd);
          rp = fmax(rp, ob.rp); stp = fmax(stp, ob.stp); bad = bad || ob.bad;
          { const bool better = ob.dn * ddA > dnA * ob.dd; dnA = better ? ob.dn : dnA; ddA = better ? ob.dd : ddA; }      // (one running ratio in the plain form)
        }
#endif
      }
    }
    QL_TICK(st, LP_C_LEGS);
    if constexpr (MD == MD_CONVEX) cv_step_fw(P, xc, F, wd, xn);
    else srbd_step_fw(P, gb, xc, F, wd, xn);
#pragma unroll
    for (int i = 0; i < 13; ++i) xc[i] = xn[i];
    QL_TICK(st, LP_C_STEP);
  }
  };
  if (QL_C_SPEC && PAIR && porder == 3u) sweep(std::true_type{});
  else sweep(std::false_type{});
  if (PAIR) {      // the partner's candidates: the lower lane tracked the first points (A), the upper lane the second (B)
    double lo, hi;
    ql_pair(rp, lo, hi); rp = fmax(lo, hi);
    ql_pair(stp, lo, hi); stp = fmax(lo, hi);
    ql_pair(bad ? 1.0 : 0.0, lo, hi); bad = (lo != 0.0) || (hi != 0.0);
  }
  double dn = dnA, dd = ddA;
  { const bool better = dnB * dd > dn * ddB; dn = better ? dnB : dn; dd = better ? ddB : dd; }
  const double ap = (rp > P.tau) ? P.tau / rp : 1.0;
  const double ad = (dn * 1.0 > P.tau * dd) ? P.tau * dd / dn : 1.0;
  st.bad_step = (bad || !(ap > 0.0) || !(ad > 0.0)) ? 1 : 0;      // also a NaN step length (0 * inf in the ratios)
  st.ap = ap; st.ad = ad;
  st.last_ap = ap; st.last_ad = ad;
  st.last_step = stp;
}

// ---- one interior-point iteration of one lane: the control flow of qmpc_solve_body.inc ------------------------------
// returns true while the instance needs more iterations
template <int NL, int MD = MD_QUAT>
QL_FN bool lane_iteration(const DevParams& P, const Ctx& c, const WsOff& O, const LaneK<NL>& K, LaneState& st, bool warm = false) {
  st.it += 1;
  // (the kernel takes the warm instantiations only while some lane of the wavefront still carries a slack residual)
  if (warm && st.rho != 0.0) pass_A<NL, true, MD>(P, c, O, K, st, st.it == 1, (FootPtr)K.foot);
  else pass_A<NL, false, MD>(P, c, O, K, st, st.it == 1, (FootPtr)K.foot);
  const double resid = st.rho * st.rcmax;     // largest |rc|: the slack residual of every enabled row is rho * rc0_i
  if (st.mu <= P.mu_final && resid <= P.tol_feas && st.last_step <= P.tol_step) { st.status = QMPC_OK; return false; }
  if (st.it > P.iterations_max) { st.status = QMPC_MAX_ITER; return false; }
  double sg = P.sigma;
  const double amin = fmin(st.last_ap, st.last_ad);
  if (st.it > 1 && amin >= 0.99) sg = P.sigma_fast;
  else if (st.it > 1 && amin < 0.2) sg = fmax(sg, 0.8);
  else if (st.it > 1 && amin < 0.5) sg = fmax(sg, 0.5);
  st.target = sg * st.mu;
  warm = warm && st.rho != 0.0;
  if (!(warm ? pass_B<NL, true, MD>(P, c, O, K, st, (FootPtr)K.foot) : pass_B<NL, false, MD>(P, c, O, K, st, (FootPtr)K.foot))) {
    st.status = QMPC_NOT_PD;
    return false;
  }
  if (warm) pass_C<NL, true, MD>(P, c, O, K, st, (FootPtr)K.foot);
  else pass_C<NL, false, MD>(P, c, O, K, st, (FootPtr)K.foot);
  if (st.bad_step) { st.status = QMPC_NOT_PD; return false; }     // the last finite iterate stays in the workspace
  st.iters = st.it;
  return true;
}

// =====================================================================================================================
// Reference mode on the lane passes (QMPC_MODE_REFERENCE; four- and eight-point quaternion model): the reference's OWN
// operating mode -- the AL-iLQR scheme of its external solver with QuatMpc's settings, iterations_max = 10,
// penalty_scaling = 20, backtracking line search, status ignored (QuatMpc.cpp:21-26,256; the
// wave-per-instance form is qmpc_ref.hip; the CPU checker restates the same scheme):
//
//   lambda <- 0, rho <- penalty_initial; U <- u_ref; X <- rollout; J <- AL merit
//   repeat iter = 1 .. iterations_max:
//     pass B<AL>     Riccati step with AL weights, expected decrease dV1
//     pass C_AL      trial rollouts u = u + alpha d + K dx at alpha = 1, 1/2, ... until the merit decreases (Armijo, 1e-4)
//     pass A_AL      apply the accepted increment, roll the states out
//     pass S         stationarity |grad_U L_A|_inf at the new trajectory (costate recursion)
//     converged: stationarity < tol and feasibility < tol
//     if stationarity < tol or |dJ| < tol_cost_intermediate:  pass M<true>: lambda <- max(lambda + rho c, 0), rho <- rho * scaling, J
//
// Workspace use: LAM = lambda; S unused (kept at 1); dU = the increment of the line search's current trial; the cone values
// c_i = a_i . u + b_i are recomputed from the inputs wherever they are needed.  A lane whose line search has ended keeps
// its increments while the other lanes of its wavefront try shorter steps (`live`).
// =====================================================================================================================

// stage cost of knot k at state x (full coordinates) -- the terms of lane_finish / knot_cost
template <int MD = MD_QUAT>
QL_FN double al_state_cost(const DevParams& P, const double refp[13], int k, const double* x) {
  double xr[13];
  if constexpr (MD == MD_CONVEX) {      // quadratic in the 12 states (ConvexMpc.cpp:107-109)
    cv_xref_at(P, refp, k, xr);
    double Jc = 0.0;
#pragma unroll
    for (int i = 0; i < 12; ++i) {
      const double e = x[i] - xr[i];
      Jc += 0.5 * P.Q[i] * e * e;
    }
    return Jc;
  }
  xref_at(P, refp, k, xr);
  double J = 0.0, dq = 0.0;
#pragma unroll
  for (int i = 0; i < 13; ++i) {
    const double e = x[i] - xr[i];
    J += 0.5 * P.Q[i] * e * e;
    if (i >= 3 && i < 7) dq += xr[i] * x[i];
  }
  return J + P.w * (1.0 - fabs(dq));
}
// input cost and AL terms of one stance point: returns the input cost, adds (max(lam + rho c, 0)^2 - lam^2) to alsum
QL_FN double al_point_terms(const DevParams& P, const double cr[18], int l, const double u[3], double uz, const double lam[6],
                            double rho, double& alsum, double& viol) {
  const double e2 = u[2] - uz;
  const double Ju = 0.5 * P.R[(3 * l) % 12] * u[0] * u[0] + 0.5 * P.R[(3 * l + 1) % 12] * u[1] * u[1] +
                    0.5 * P.R[(3 * l + 2) % 12] * e2 * e2;
#pragma unroll
  for (int i = 0; i < 6; ++i) {
    double cv = cr[3 * i] * u[0] + cr[3 * i + 1] * u[1] + cr[3 * i + 2] * u[2];
    if (i == 4) cv += -P.fz_max;
    double z = lam[i] + rho * cv;
    if (z < 0.0) z = 0.0;
    alsum += z * z - lam[i] * lam[i];
    viol = fmax(viol, fmax(cv, 0.0));
  }
  return Ju;
}

// ---- pass M: AL merit, plain objective and violation of the CURRENT trajectory; UPDATE: the dual update first
// (lambda <- max(lambda + rho c, 0), then rho <- min(rho * scaling, max)) -------------------------------------------------
template <int NL, bool UPDATE, int MD = MD_QUAT>
QL_FN void pass_M(const DevParams& P, const Ctx& c, const WsOff& O, const LaneK<NL>& K, const LaneState& st, LaneAL& al) {
  const int N = P.N;
  double cr[18];
  cone_rows(P, K.rot, cr);
  const double rho_old = al.rho;
  if (UPDATE) {
    al.rho = fmin(al.rho * P.penalty_scaling, P.penalty_max);
    al.irho = 1.0 / al.rho;
  }
  const unsigned order = any_stance<NL>(st.con);
  // the knot's state, inputs and multipliers are fetched one knot ahead, into the registers just consumed
  double xk[13], uk[3 * NL], lk[6 * NL];
  auto load_x = [&](int k) {
#pragma unroll
    for (int i = 0; i < 13; ++i) xk[i] = c.W(O.X + 13 * k + i);
  };
  auto load_leg = [&](int k, int l) {
#pragma unroll
    for (int a = 0; a < 3; ++a) uk[3 * l + a] = c.W(O.U + 3 * NL * k + 3 * l + a);
#pragma unroll
    for (int i = 0; i < 6; ++i) lk[6 * l + i] = c.W(O.LAM + 6 * NL * k + 6 * l + i);
  };
  load_x(0);
#pragma unroll
  for (int l = 0; l < NL; ++l) if ((order >> l) & 1u) load_leg(0, l);
  double Jp = 0.0, alsum = 0.0, viol = 0.0;
  for (int k = 0; k <= N; ++k) {
    double x[13];
#pragma unroll
    for (int i = 0; i < 13; ++i) x[i] = xk[i];
    if (k < N) load_x(k + 1);
    Jp += al_state_cost<MD>(P, K.refp, k, x);
    if (k == N) break;
    const int kn = (k + 1 < N) ? k + 1 : k;
#pragma unroll
    for (int l = 0; l < NL; ++l) {
      if (!((order >> l) & 1u)) continue;       // wave-uniform
      double u[3], lam[6];
#pragma unroll
      for (int a = 0; a < 3; ++a) u[a] = uk[3 * l + a];
#pragma unroll
      for (int i = 0; i < 6; ++i) lam[i] = lk[6 * l + i];
      load_leg(kn, l);
      const bool on = (st.con >> l) & 1u;
      if (UPDATE) {
        if (on)
#pragma unroll
          for (int i = 0; i < 6; ++i) {
            double cv = cr[3 * i] * u[0] + cr[3 * i + 1] * u[1] + cr[3 * i + 2] * u[2];
            if (i == 4) cv += -P.fz_max;
            const double z = lam[i] + rho_old * cv;
            lam[i] = (z > 0.0) ? z : 0.0;
          }
        // every lane: no store under a per-lane condition (pass A: a wait behind a skippable block of stores waits for them);
        // a lane whose point is not in stance stores back what it read
#pragma unroll
        for (int i = 0; i < 6; ++i) c.St(O.LAM + 6 * NL * k + 6 * l + i, lam[i]);
      }
      if (on) Jp += al_point_terms(P, cr, l, u, st.uz, lam, al.rho, alsum, viol);
    }
  }
  al.Jp = Jp;
  al.viol = viol;
  al.J = Jp + alsum / (2.0 * al.rho);
}

// ---- pass A_AL: apply the accepted increment of the line search and roll the states out open loop ----------------------
template <int NL, int MD = MD_QUAT>
QL_FN void pass_A_AL(const DevParams& P, const Ctx& c, const WsOff& O, const LaneK<NL>& K, const LaneState& st, int sel = 0) {
  const int N = P.N;
  const int dsl = sel ? O.RC : O.dU;      // the accepted trial's increments (per lane: see pass_C_AL)
  const double gb[3] = {K.rot[6] * (-9.81), K.rot[7] * (-9.81), K.rot[8] * (-9.81)};
  const unsigned order = any_stance<NL>(st.con);
  double x[13], xn[13];
#pragma unroll
  for (int i = 0; i < 13; ++i) x[i] = c.W(O.X + i);
  double uk[3 * NL], dk[3 * NL];      // one knot ahead
  auto load_leg = [&](int k, int l) {
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      uk[3 * l + a] = c.W(O.U + 3 * NL * k + 3 * l + a);
      dk[3 * l + a] = c.W(dsl + 3 * NL * k + 3 * l + a);
    }
  };
#pragma unroll
  for (int l = 0; l < NL; ++l) if ((order >> l) & 1u) load_leg(0, l);
  for (int k = 0; k < N; ++k) {
    const int kn = (k + 1 < N) ? k + 1 : k;
    double F[3] = {0, 0, 0}, wd[3] = {K.wd0[0], K.wd0[1], K.wd0[2]};
#pragma unroll
    for (int l = 0; l < NL; ++l) {
      if (!((order >> l) & 1u)) continue;       // wave-uniform
      double u[3], B[9];
#pragma unroll
      for (int a = 0; a < 3; ++a) u[a] = uk[3 * l + a] + dk[3 * l + a];
      const double uo[3] = {uk[3 * l], uk[3 * l + 1], uk[3 * l + 2]};
      load_leg(kn, l);
      const bool on = (st.con >> l) & 1u;
#pragma unroll
      for (int a = 0; a < 3; ++a) c.St(O.U + 3 * NL * k + 3 * l + a, on ? u[a] : uo[a]);      // every lane (see pass_M)
      if (!on) continue;
      if constexpr (MD == MD_CONVEX) {      // wd collects the raw torque sum
#pragma unroll
        for (int a = 0; a < 3; ++a) F[a] += u[a];
        cv_cross_acc(&K.foot[3 * l], u, wd);
      } else {
      leg_bw0(P, &K.foot[3 * l], B);
#pragma unroll
      for (int a = 0; a < 3; ++a) {
        F[a] += u[a];
        wd[a] += B[3 * a] * u[0] + B[3 * a + 1] * u[1] + B[3 * a + 2] * u[2];
      }
      }
    }
    if constexpr (MD == MD_CONVEX) cv_step_fw(P, x, F, wd, xn);
    else srbd_step_fw(P, gb, x, F, wd, xn);
#pragma unroll
    for (int i = 0; i < 13; ++i) { x[i] = xn[i]; c.St(O.X + 13 * (k + 1) + i, xn[i]); }
  }
}

// ---- pass C_AL: one trial of the line search -- closed-loop rollout at step length al.alpha, the trial's increments
// (stored while `live`), its merit, violation and largest increment ---------------------------------------------------------
// NA = 2: TWO trials of the backtracking line search in one sweep, step lengths alpha and alpha / 2.  They share the loads (old
// state, gains, inputs, multipliers) and the per-point blocks (weights, frame, L D L': functions of the current inputs only);
// the rollout, the input recovery and the merit terms run once per step length.  The increments of the second trial go to the
// RC slot (unused in this mode), 3 NL per knot like the dU slot.
// PAIR (lane pairs, four-point quaternion model): the two points of a diagonal pair go to the two partner lanes -- one per-point
// block per lane and pair instead of two; force / torque / objective shares cross the pair (v_permlane32_swap) and are added as
// (acc + first) + second, the penalty sum runs as a chain over the lower lane's six rows, then over the upper lane's.  The plain
// form of the four-point models visits the points in the same order (0, 3, 1, 2), so a pair-mode sweep returns its bits.
template <int NL, int NA = 2, int MD = MD_QUAT, bool PAIR = false>
QL_FN void pass_C_AL(const DevParams& P, const Ctx& c, const WsOff& O, const LaneK<NL>& K, const LaneState& st, LaneAL& al, bool live) {
  static_assert(!PAIR || (NL == 4 && MD == MD_QUAT), "pair split: the four-point quaternion model");
  typedef LDim<NL> D;
  const int N = P.N;
  const double gb_[3] = {K.rot[6] * (-9.81), K.rot[7] * (-9.81), K.rot[8] * (-9.81)};
  double cr_[18];
  cone_rows(P, K.rot, cr_);
  // The sweep's per-instance constants (gravity in the body frame, wd0, contact points, reference parameters: 31 values) wait in
  // LDS and are read where they are used -- in registers across the knots they end up in scratch, and a scratch re-load comes
  // back through the vector-memory counter, behind the rows in flight.  The LDS block is free in this pass (the cost-to-go matrix
  // of the backward pass lives there, and that pass rebuilds it from the terminal cost at every call): the pair form uses the
  // staging rows (upper halves), the plain form the lanes' own slots of the same rows.
  constexpr bool kKLds = QL_CAL_KLDS && QL_DEVICE && (PAIR || QL_CAL_KLDS_PLAIN);
  auto kld = [&](int h) -> double { return PAIR ? c.SR(h) : (double)c.PL(h); };
  auto kst = [&](int h, double v) { if (PAIR) c.SRst(h, v); else c.PL(h) = v; };
  constexpr bool kLCr = kKLds && (QL_CAL_KLDS & 1), kLGb = kKLds && (QL_CAL_KLDS & 2), kLWd = kKLds && (QL_CAL_KLDS & 4), kLFoot = kKLds && (QL_CAL_KLDS & 8), kLRef = kKLds && (QL_CAL_KLDS & 16);
  constexpr int kCrRow = 0, kGbRow = 18, kWdRow = 21, kFootRow = 24, kRefRow = 24 + 3 * NL;
  if constexpr (kKLds) {
#pragma unroll
    for (int i = 0; i < 18; ++i) kst(kCrRow + i, cr_[i]);
#pragma unroll
    for (int a = 0; a < 3; ++a) { kst(kGbRow + a, gb_[a]); kst(kWdRow + a, K.wd0[a]); }
#pragma unroll
    for (int i = 0; i < 3 * NL; ++i) kst(kFootRow + i, K.foot[i]);
#pragma unroll
    for (int i = 0; i < 13; ++i) kst(kRefRow + i, K.refp[i]);
  }
  static_assert(!PAIR || NA == 2, "pair form: the two trials of a sweep go to the two partner lanes");
  double alpha[NA];
  alpha[0] = al.alpha;
  if (NA > 1) alpha[NA - 1] = 0.5 * al.alpha;
  // Pair form: the rollout of a trial -- state, gains times state difference, objective, state step -- is ONE lane's work (lower
  // lane: step length alpha, upper lane: alpha / 2; NH = 1 "head trial" per lane); the trial's costate zeta crosses to the partner
  // before the per-point blocks (each lane: ITS point, both trials), the points' shares of a trial's sums cross back to its lane
  constexpr int NH = PAIR ? 1 : NA;
  double alh[NH];
#pragma unroll
  for (int q = 0; q < NH; ++q) alh[q] = PAIR ? (c.half ? alpha[NA - 1] : alpha[0]) : alpha[q];
  double xc[NH][13], xn[13];
#pragma unroll
  for (int q = 0; q < NH; ++q)
#pragma unroll
    for (int i = 0; i < 13; ++i) xc[q][i] = c.W(O.X + i);
  const unsigned order = any_stance<NL>(st.con);
  // old state, gains, inputs and multipliers of a knot are fetched one knot ahead, into the registers just consumed
  // (gains in double precision: columns 0..5 and zeta0 travel one knot ahead like before, columns 6..11 are fetched at the top
  // of their knot and used after the first half)
  double xo[13], gn[D::GAIN], g2[D::GAIN2], uk[3 * NL], lk[6 * NL];
  auto load_head = [&](int k) {
#pragma unroll
    for (int i = 0; i < 13; ++i) xo[i] = c.W(O.X + 13 * k + i);
#pragma unroll
    for (int i = 0; i < D::GAIN; ++i) gn[i] = c.W(O.G + D::GAIN * k + i);
#if QL_AL_G2_AHEAD == 1
#pragma unroll
    for (int i = 0; i < D::GAIN2; ++i) g2[i] = c.W(O.G2 + D::GAIN2 * k + i);
#endif
  };
  auto load_leg = [&](int k, int l) {
#pragma unroll
    for (int a = 0; a < 3; ++a) uk[3 * l + a] = c.W(O.U + 3 * NL * k + 3 * l + a);
#pragma unroll
    for (int i = 0; i < 6; ++i) lk[6 * l + i] = c.W(O.LAM + 6 * NL * k + 6 * l + i);
  };
  // pair form: one buffer per diagonal pair, holding THIS lane's point of it (per-lane row address)
  auto load_pair = [&](int k, int pr) {
    const int lm = c.half ? pair_leg<NL>(pr, 1) : pair_leg<NL>(pr, 0);
#pragma unroll
    for (int a = 0; a < 3; ++a) uk[3 * pr + a] = c.W(O.U + 3 * NL * k + 3 * lm + a);
#pragma unroll
    for (int i = 0; i < 6; ++i) lk[6 * pr + i] = c.W(O.LAM + 6 * NL * k + 6 * lm + i);
  };
  // which pairs any lane of the wavefront has in stance
  unsigned porder = 0;
#pragma unroll
  for (int pr = 0; pr < NL / 2; ++pr)
    porder |= (((order >> pair_leg<NL>(pr, 0)) | (order >> pair_leg<NL>(pr, 1))) & 1u) << pr;
  load_head(0);
#if QL_AL_G2_AHEAD == 2
#pragma unroll
  for (int i = 0; i < D::GAIN2; ++i) g2[i] = c.W(O.G2 + i);
#endif
  if constexpr (PAIR) {
#pragma unroll
    for (int pr = 0; pr < NL / 2; ++pr) if ((porder >> pr) & 1u) load_pair(0, pr);
  } else {
#pragma unroll
    for (int l = 0; l < NL; ++l) if ((order >> l) & 1u) load_leg(0, l);
  }
  double Jp[NH], alsum[NH], viol[NA], stp[NA];
  bool bad[NA];
#pragma unroll
  for (int q = 0; q < NA; ++q) { viol[q] = 0.0; stp[q] = 0.0; bad[q] = false; }
#pragma unroll
  for (int q = 0; q < NH; ++q) { Jp[q] = 0.0; alsum[q] = 0.0; }
  for (int k = 0; k < N; ++k) {
    const int kn = (k + 1 < N) ? k + 1 : k;
    double zeta[NA][6], zh[NH][6];
    double cr[18], gb[3], refk[13];
    if constexpr (kKLds) QL_FENCE();      // (the reads below stay inside their knot)
#pragma unroll
    for (int i = 0; i < 13; ++i) refk[i] = kLRef ? kld(kRefRow + i) : K.refp[i];
    if constexpr (!kLCr)
#pragma unroll
      for (int i = 0; i < 18; ++i) cr[i] = cr_[i];
#pragma unroll
    for (int a = 0; a < 3; ++a) gb[a] = gb_[a];
#if QL_AL_G2_AHEAD == 0
#pragma unroll
    for (int i = 0; i < D::GAIN2; ++i) g2[i] = c.W(O.G2 + D::GAIN2 * k + i);
#endif
    double Wk[4] = {0, 0, 0, 0};       // ConvexMpc's model: Iw^-1 at the OLD knot state's midpoint yaw (the linearisation point)
    if constexpr (MD == MD_CONVEX) cv_winv_mid(P, xo[2], xo[8], Wk);
    {
      double G[12];
      if constexpr (MD != MD_CONVEX) quatG(&xo[3], G);
#pragma unroll
      for (int q = 0; q < NH; ++q) {
        Jp[q] += al_state_cost<MD>(P, refk, k, xc[q]);
        double dx[12];
        if constexpr (MD == MD_CONVEX) {      // blocks in the recursion's order [p, phi, v, w]
#pragma unroll
          for (int a = 0; a < 3; ++a) {
            dx[a] = xc[q][3 + a] - xo[3 + a];
            dx[3 + a] = xc[q][a] - xo[a];
            dx[6 + a] = xc[q][9 + a] - xo[9 + a];
            dx[9 + a] = xc[q][6 + a] - xo[6 + a];
          }
        } else {
#pragma unroll
        for (int a = 0; a < 3; ++a) {
          dx[a] = xc[q][a] - xo[a];
          dx[6 + a] = xc[q][7 + a] - xo[7 + a];
          dx[9 + a] = xc[q][10 + a] - xo[10 + a];
        }
        // (explicitly fused, term by term: which of two products a sum fuses is otherwise the compiler's choice, and it chose
        // differently in the instantiation that rolls ONE trial per lane -- the pair form must return the plain form's bits)
        const double isc = ql_rcp(fma(xo[6], xc[q][6], fma(xo[5], xc[q][5], fma(xo[4], xc[q][4], xo[3] * xc[q][3]))));
#pragma unroll
        for (int a = 0; a < 3; ++a)
          dx[3 + a] = fma(G[9 + a], xc[q][6], fma(G[6 + a], xc[q][5], fma(G[3 + a], xc[q][4], G[a] * xc[q][3]))) * isc;
        }
#pragma unroll
        for (int i = 0; i < 6; ++i) zh[q][i] = ql_rounded(alh[q] * gn[36 + i]);
#pragma unroll
        for (int j = 0; j < 6; ++j)
#pragma unroll
          for (int i = 0; i < 6; ++i) zh[q][i] = fma(gn[6 * j + i], dx[j], zh[q][i]);
#pragma unroll
        for (int j = 0; j < 6; ++j)
#pragma unroll
          for (int i = 0; i < 6; ++i) zh[q][i] = fma(g2[6 * j + i], dx[6 + j], zh[q][i]);
      }
    }
#pragma unroll
    for (int i = 0; i < 6; ++i) {
      if constexpr (PAIR) ql_pair(zh[0][i], zeta[0][i], zeta[NA - 1][i]);      // lower lane's trial, upper lane's trial: both to both
      else
#pragma unroll
        for (int q = 0; q < NA; ++q) zeta[q][i] = zh[q < NH ? q : 0][i];
    }
    load_head(kn);
    double F[NH][3], wd[NH][3];
#pragma unroll
    for (int q = 0; q < NH; ++q)
#pragma unroll
      for (int a = 0; a < 3; ++a) { F[q][a] = 0.0; wd[q][a] = kLWd ? kld(kWdRow + a) : K.wd0[a]; }
    if constexpr (kLCr) {
#pragma unroll
      for (int i = 0; i < 18; ++i) cr[i] = kld(kCrRow + i);
    }
    // one contact point: AL weights, factorised block, the trials' increments, new inputs, torque shares and merit terms
    auto point = [&](const double u[3], const double lam[6], const double r[3], const double Rw[3], int l, double dq[NA][3],
                     double fun[NA][3], double ftq[NA][3], double ju[NA], double at[NA][6]) {
      double sv[6], lv[6], rcl[6], B[9];
#pragma unroll
      for (int i = 0; i < 6; ++i) {
        double cv = fma(cr[3 * i + 2], u[2], fma(cr[3 * i + 1], u[1], cr[3 * i] * u[0]));
        if (i == 4) cv += -P.fz_max;
        const double z = fma(al.rho, cv, lam[i]);
        const bool act = z > 0.0;
        sv[i] = 1.0;
        rcl[i] = act ? z * al.irho : 0.0;
        lv[i] = act ? al.rho : 0.0;
      }
      if constexpr (MD == MD_CONVEX) cv_leg_bw0(Wk, r, B);
      else leg_bw0(P, r, B);
      LegBlk lb;
      leg_block(P, cr, rcl, l, sv, lv, 0u, 1.0, 0.0, u, st.uz, lb, Rw);
#pragma unroll
      for (int q = 0; q < NA; ++q) {
        // rhs = T'(zeta_f + Bw0' zeta_t) + alpha gq;  du = -T Db^-1 rhs
        double t[3], rh[3];
#pragma unroll
        // (every sum of products explicitly fused, term by term: the compiler's choice of WHICH product a sum fuses differed between
        // the plain and the pair instantiation of this block by an ulp of the increment)
        for (int a = 0; a < 3; ++a) t[a] = fma(B[6 + a], zeta[q][5], fma(B[3 + a], zeta[q][4], fma(B[a], zeta[q][3], zeta[q][a])));
#pragma unroll
        for (int a = 0; a < 3; ++a) rh[a] = fma(alpha[q], lb.gq[a], fma(lb.T[6 + a], t[2], fma(lb.T[3 + a], t[1], lb.T[a] * t[0])));
        const double y0 = rh[0], y1 = fma(-lb.l10, y0, rh[1]), y2 = fma(-lb.l21, y1, fma(-lb.l20, y0, rh[2]));
        const double z2 = y2 * lb.id2;
        const double z1 = fma(-lb.l21, z2, y1 * lb.id1);
        const double z0 = fma(-lb.l20, z2, fma(-lb.l10, z1, y0 * lb.id0));
        double un[3];
#pragma unroll
        for (int a = 0; a < 3; ++a) {
          const double du = -fma(lb.T[3 * a + 2], z2, fma(lb.T[3 * a + 1], z1, lb.T[3 * a] * z0));
          stp[q] = fmax(stp[q], fabs(du));
          bad[q] = bad[q] || !(fabs(du) <= 1e300);
          un[a] = u[a] + du;
          dq[q][a] = du;
          fun[q][a] = un[a];
        }
        // the terms of al_point_terms: input cost, and per row the penalty term z^2 - lam^2 (added to the running sum by the caller)
        const double e2 = un[2] - st.uz;
        ju[q] = fma(0.5 * Rw[2] * e2, e2, fma(0.5 * Rw[1] * un[1], un[1], 0.5 * Rw[0] * un[0] * un[0]));
#pragma unroll
        for (int i = 0; i < 6; ++i) {
          double cv = fma(cr[3 * i + 2], un[2], fma(cr[3 * i + 1], un[1], cr[3 * i] * un[0]));
          if (i == 4) cv += -P.fz_max;
          double z = fma(al.rho, cv, lam[i]);
          if (z < 0.0) z = 0.0;
          at[q][i] = fma(z, z, -(lam[i] * lam[i]));
          viol[q] = fmax(viol[q], fmax(cv, 0.0));
        }
        if constexpr (MD == MD_CONVEX) {      // the rollout wants the raw torque r x u: the caller forms it
#pragma unroll
          for (int a = 0; a < 3; ++a) ftq[q][a] = 0.0;
        } else {
#pragma unroll
          for (int a = 0; a < 3; ++a) ftq[q][a] = fma(B[3 * a + 2], un[2], fma(B[3 * a + 1], un[1], B[3 * a] * un[0]));
        }
      }
    };
    if constexpr (PAIR) {
#pragma unroll
      for (int pr = 0; pr < NL / 2; ++pr) {
        if (!((porder >> pr) & 1u)) continue;       // wave-uniform
        constexpr int dummy_ = 0; (void)dummy_;
        const int la = pair_leg<NL>(pr, 0), lb_ = pair_leg<NL>(pr, 1);
        const int lm = c.half ? lb_ : la;      // this lane's point of the pair
        const bool on_m = (st.con >> lm) & 1u;
        double u[3], lam[6], r[3], Rw[3];
#pragma unroll
        for (int a = 0; a < 3; ++a) {
          u[a] = uk[3 * pr + a];
          r[a] = kLFoot ? kld(kFootRow + 3 * lm + a) : (c.half ? K.foot[3 * lb_ + a] : K.foot[3 * la + a]);
          Rw[a] = c.half ? ql_uniform(P.R[3 * (lb_ & 3) + a]) : ql_uniform(P.R[3 * (la & 3) + a]);
        }
#pragma unroll
        for (int i = 0; i < 6; ++i) lam[i] = lk[6 * pr + i];
        load_pair(kn, pr);
        double dq[NA][3], fun[NA][3], ftq[NA][3], ju[NA], at[NA][6];
#pragma unroll
        for (int q = 0; q < NA; ++q) {
          ju[q] = 0.0;
#pragma unroll
          for (int a = 0; a < 3; ++a) { dq[q][a] = 0.0; fun[q][a] = 0.0; ftq[q][a] = 0.0; }
#pragma unroll
          for (int i = 0; i < 6; ++i) at[q][i] = 0.0;
        }
        if (on_m) point(u, lam, r, Rw, lm, dq, fun, ftq, ju, at);
        if (live)
#pragma unroll
          for (int q = 0; q < NA; ++q)
#pragma unroll
            for (int a = 0; a < 3; ++a) c.StOwn((q == 0 ? O.dU : O.RC) + 3 * NL * k + 3 * lm + a, dq[q][a]);
        {
          // this lane's TRIAL (lower lane: 0, upper lane: NA - 1) gets both points' shares: its own point's of that trial, and the
          // partner's point's, which the partner sends (one exchange per quantity serves both lanes); added in the plain form's
          // order -- the pair's first point (the lower lane's) before the second
          constexpr int q1 = NA - 1;
          auto other = [&](double mine_for_partner) { double lo, hi; ql_pair(mine_for_partner, lo, hi); return c.half ? lo : hi; };
#pragma unroll
          for (int a = 0; a < 3; ++a) {
            const double own = c.half ? fun[q1][a] : fun[0][a], oth = other(c.half ? fun[0][a] : fun[q1][a]);
            F[0][a] = (F[0][a] + (c.half ? oth : own)) + (c.half ? own : oth);
            const double ownt = c.half ? ftq[q1][a] : ftq[0][a], otht = other(c.half ? ftq[0][a] : ftq[q1][a]);
            wd[0][a] = (wd[0][a] + (c.half ? otht : ownt)) + (c.half ? ownt : otht);
          }
          {
            const double own = c.half ? ju[q1] : ju[0], oth = other(c.half ? ju[0] : ju[q1]);
            Jp[0] = (Jp[0] + (c.half ? oth : own)) + (c.half ? own : oth);
          }
          double own6[6], oth6[6];
#pragma unroll
          for (int i = 0; i < 6; ++i) {
            own6[i] = c.half ? at[q1][i] : at[0][i];
            oth6[i] = other(c.half ? at[0][i] : at[q1][i]);
          }
          // the penalty sum of the trial: a chain over the first point's rows, then over the second's
#pragma unroll
          for (int i = 0; i < 6; ++i) alsum[0] += c.half ? oth6[i] : own6[i];
#pragma unroll
          for (int i = 0; i < 6; ++i) alsum[0] += c.half ? own6[i] : oth6[i];
        }
      }
    } else {
#pragma unroll
    for (int pr = 0; pr < NL / 2; ++pr)
#pragma unroll
    for (int jj = 0; jj < 2; ++jj) {
      const int l = pair_leg<NL>(pr, jj);      // (the four-point models: 0, 3, 1, 2 -- the order of the pair form)
      if (!((order >> l) & 1u)) continue;       // wave-uniform
      double u[3], lam[6];
#pragma unroll
      for (int a = 0; a < 3; ++a) u[a] = uk[3 * l + a];
#pragma unroll
      for (int i = 0; i < 6; ++i) lam[i] = lk[6 * l + i];
      load_leg(kn, l);
      double dq[NA][3], fun[NA][3], ftq[NA][3], ju[NA], at[NA][6];
#pragma unroll
      for (int q = 0; q < NA; ++q)
#pragma unroll
        for (int a = 0; a < 3; ++a) dq[q][a] = 0.0;
      if ((st.con >> l) & 1u) {
        const double Rw[3] = {P.R[(3 * l) % 12], P.R[(3 * l + 1) % 12], P.R[(3 * l + 2) % 12]};
        double rl[3];
#pragma unroll
        for (int a = 0; a < 3; ++a) rl[a] = kLFoot ? kld(kFootRow + 3 * l + a) : K.foot[3 * l + a];
        point(u, lam, rl, Rw, l, dq, fun, ftq, ju, at);
#pragma unroll
        for (int q = 0; q < NA; ++q) {
          Jp[q] += ju[q];
#pragma unroll
          for (int i = 0; i < 6; ++i) alsum[q] += at[q][i];
          if constexpr (MD == MD_CONVEX) {
#pragma unroll
            for (int a = 0; a < 3; ++a) F[q][a] += fun[q][a];
            cv_cross_acc(rl, fun[q], wd[q]);
          } else {
#pragma unroll
            for (int a = 0; a < 3; ++a) {
              F[q][a] += fun[q][a];
              wd[q][a] += ftq[q][a];
            }
          }
        }
      }
      // every lane: no store under a per-lane stance condition (a wait behind a skippable block of stores waits for them; the
      // increments of a point that is not in stance are never read)
      if (live)
#pragma unroll
        for (int q = 0; q < NA; ++q)
#pragma unroll
          for (int a = 0; a < 3; ++a) c.St((q == 0 ? O.dU : O.RC) + 3 * NL * k + 3 * l + a, dq[q][a]);
    }
    }
#if QL_AL_G2_AHEAD == 2
    // the next knot's second gain block: issued once the per-point phase has released its registers, covered by the state steps
#pragma unroll
    for (int i = 0; i < D::GAIN2; ++i) g2[i] = c.W(O.G2 + D::GAIN2 * kn + i);
#endif
#pragma unroll
    for (int q = 0; q < NH; ++q) {
      if constexpr (MD == MD_CONVEX) cv_step_fw(P, xc[q], F[q], wd[q], xn);
      else {
        if constexpr (kLGb)
#pragma unroll
          for (int a = 0; a < 3; ++a) gb[a] = kld(kGbRow + a);
        srbd_step_fw(P, gb, xc[q], F[q], wd[q], xn);
      }
#pragma unroll
      for (int i = 0; i < 13; ++i) xc[q][i] = xn[i];
    }
  }
#pragma unroll
  for (int q = 0; q < NH; ++q) {
    double refN[13];
#pragma unroll
    for (int i = 0; i < 13; ++i) refN[i] = kLRef ? kld(kRefRow + i) : K.refp[i];
    Jp[q] += al_state_cost<MD>(P, refN, N, xc[q]);
  }
  double JpF[NA], alF[NA];
  if constexpr (PAIR) {
    // the trial of a lane: largest increment / violation over BOTH points (the partner sends its point's figure of that trial), then
    // every figure of both trials to both lanes
    constexpr int q1 = NA - 1;
    auto other = [&](double mine_for_partner) { double lo, hi; ql_pair(mine_for_partner, lo, hi); return c.half ? lo : hi; };
    const double sm = fmax(c.half ? stp[q1] : stp[0], other(c.half ? stp[0] : stp[q1]));
    const double vm = fmax(c.half ? viol[q1] : viol[0], other(c.half ? viol[0] : viol[q1]));
    const double bm = ((c.half ? bad[q1] : bad[0]) ? 1.0 : 0.0) + other((c.half ? bad[0] : bad[q1]) ? 1.0 : 0.0);
    double lo, hi;
    ql_pair(Jp[0], lo, hi); JpF[0] = lo; JpF[q1] = hi;
    ql_pair(alsum[0], lo, hi); alF[0] = lo; alF[q1] = hi;
    ql_pair(sm, lo, hi); stp[0] = lo; stp[q1] = hi;
    ql_pair(vm, lo, hi); viol[0] = lo; viol[q1] = hi;
    ql_pair(bm, lo, hi); bad[0] = lo != 0.0; bad[q1] = hi != 0.0;
  } else {
#pragma unroll
    for (int q = 0; q < NA; ++q) { JpF[q] = Jp[q < NH ? q : 0]; alF[q] = alsum[q < NH ? q : 0]; }
  }
  al.Jnp = JpF[0];
  al.vn = viol[0];
  al.stp = stp[0];
  al.Jn = bad[0] ? (double)NAN : JpF[0] + alF[0] / (2.0 * al.rho);
  if (NA > 1) {
    al.Jnp2 = JpF[NA - 1];
    al.vn2 = viol[NA - 1];
    al.stp2 = stp[NA - 1];
    al.Jn2 = bad[NA - 1] ? (double)NAN : JpF[NA - 1] + alF[NA - 1] / (2.0 * al.rho);
  }
}

// ---- pass S: |grad_U L_A|_inf at (X, U) through the costate recursion  y_k = lx_k + Abar_k' y_{k+1},
//      gu_l = R (u_l - uref_l) + Wr_l' (M_k' y_{k+1}) + sum_i max(lam_i + rho c_i, 0) a_i   (ref_stationarity of qmpc_ref.hip)
template <int NL, int MD = MD_QUAT>
QL_FN void pass_S(const DevParams& P, const Ctx& c, const WsOff& O, const LaneK<NL>& K, const LaneState& st, LaneAL& al) {
  const int N = P.N;
  const double m1 = P.h * (P.hh * (1.0 / P.mass)), m2 = P.h * (1.0 / P.mass);
  double cr[18];
  cone_rows(P, K.rot, cr);
  const unsigned order = any_stance<NL>(st.con);
  double y[12];
  {
    double lxx[6];
    cost_expansion<NL, MD>(P, c, O, K, N, y, lxx);
  }
  // state of the knot, quaternion of the next one, inputs and multipliers: one knot ahead, into the registers just consumed
  double xk[13], qn[4], uk[3 * NL], lk[6 * NL];
  auto load_head = [&](int k) {
#pragma unroll
    for (int i = 0; i < 13; ++i) xk[i] = c.W(O.X + 13 * k + i);
#pragma unroll
    for (int i = 0; i < 4; ++i) qn[i] = c.W(O.X + 13 * (k + 1) + 3 + i);
  };
  auto load_leg = [&](int k, int l) {
#pragma unroll
    for (int a = 0; a < 3; ++a) uk[3 * l + a] = c.W(O.U + 3 * NL * k + 3 * l + a);
#pragma unroll
    for (int i = 0; i < 6; ++i) lk[6 * l + i] = c.W(O.LAM + 6 * NL * k + 6 * l + i);
  };
  load_head(N - 1);
#pragma unroll
  for (int l = 0; l < NL; ++l) if ((order >> l) & 1u) load_leg(N - 1, l);
  double g = 0.0;
  for (int k = N - 1; k >= 0; --k) {
    const int kn = (k > 0) ? k - 1 : 0;
    double x[13], xn[4], u[3 * NL], lam[6 * NL];
#pragma unroll
    for (int i = 0; i < 13; ++i) x[i] = xk[i];
#pragma unroll
    for (int i = 0; i < 4; ++i) xn[i] = qn[i];
#pragma unroll
    for (int i = 0; i < 3 * NL; ++i) u[i] = uk[i];
#pragma unroll
    for (int i = 0; i < 6 * NL; ++i) lam[i] = lk[i];
    load_head(kn);
#pragma unroll
    for (int l = 0; l < NL; ++l) if ((order >> l) & 1u) load_leg(kn, l);
    // the knot's angular acceleration (for the expansion; ConvexMpc's model: the raw torque sum)
    double wd[3] = {K.wd0[0], K.wd0[1], K.wd0[2]};
    double Wk[4] = {0, 0, 0, 0};
    if constexpr (MD == MD_CONVEX) cv_winv_mid(P, x[2], x[8], Wk);
#pragma unroll
    for (int l = 0; l < NL; ++l) {
      if (!((st.con >> l) & 1u)) continue;
      if constexpr (MD == MD_CONVEX) cv_cross_acc(&K.foot[3 * l], &u[3 * l], wd);
      else {
      double B[9];
      leg_bw0(P, &K.foot[3 * l], B);
#pragma unroll
      for (int a = 0; a < 3; ++a) wd[a] += B[3 * a] * u[3 * l] + B[3 * a + 1] * u[3 * l + 1] + B[3 * a + 2] * u[3 * l + 2];
      }
    }
    // dynamics expansion (pass B step 2)
    double A1[9], A3[9], Wt[9];
    if constexpr (MD == MD_CONVEX) cv_expansion(P, x, wd, A1, A3, Wt);
    else {
      double G0[12], Gm[12], Gn[12];
      quatG(&x[3], G0);
      double qm[4], wm[3];
#pragma unroll
      for (int r = 0; r < 4; ++r)
        qm[r] = x[3 + r] + P.hh * (0.5 * (G0[3 * r] * x[10] + G0[3 * r + 1] * x[11] + G0[3 * r + 2] * x[12]));
#pragma unroll
      for (int a = 0; a < 3; ++a) wm[a] = x[10 + a] + P.hh * wd[a];
      quatG(qm, Gm);
      quatG(xn, Gn);
#pragma unroll
      for (int cc = 0; cc < 3; ++cc) {
        double gg[4], gm[4], t0[4], t1[4], t2[4], ag[4], aw[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) { gg[r] = G0[3 * r + cc]; gm[r] = Gm[3 * r + cc]; }
        omega_mul(&x[10], gg, t0);
#pragma unroll
        for (int r = 0; r < 4; ++r) t1[r] = gg[r] + (0.5 * P.hh) * t0[r];
        omega_mul(wm, t1, t2);
        omega_mul(wm, gg, t0);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          ag[r] = gg[r] + P.hh * t2[r];
          aw[r] = P.hh * ((0.5 * P.hh) * t0[r] + gm[r]);
        }
#pragma unroll
        for (int r = 0; r < 3; ++r) {
          A1[3 * r + cc] = Gn[r] * ag[0] + Gn[3 + r] * ag[1] + Gn[6 + r] * ag[2] + Gn[9 + r] * ag[3];
          A3[3 * r + cc] = Gn[r] * aw[0] + Gn[3 + r] * aw[1] + Gn[6 + r] * aw[2] + Gn[9 + r] * aw[3];
          Wt[3 * r + cc] = ((0.5 * P.hh) * P.h) * (Gn[r] * gm[0] + Gn[3 + r] * gm[1] + Gn[6 + r] * gm[2] + Gn[9 + r] * gm[3]);
        }
      }
    }
    // m6 = M' y_{k+1}
    double m6[6];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      m6[a] = m1 * y[a] + m2 * y[6 + a];
      m6[3 + a] = Wt[a] * y[3] + Wt[3 + a] * y[4] + Wt[6 + a] * y[5] + P.h * y[9 + a];
    }
#pragma unroll
    for (int l = 0; l < NL; ++l) {
      if (!((st.con >> l) & 1u)) continue;
      double B[9];
      if constexpr (MD == MD_CONVEX) cv_leg_bw0(Wk, &K.foot[3 * l], B);
      else leg_bw0(P, &K.foot[3 * l], B);
      const double* ul = &u[3 * l];
      double gu[3];
#pragma unroll
      for (int a = 0; a < 3; ++a)
        gu[a] = P.R[(3 * l + a) % 12] * (ul[a] - ((a == 2) ? st.uz : 0.0)) + m6[a] + (B[a] * m6[3] + B[3 + a] * m6[4] + B[6 + a] * m6[5]);
#pragma unroll
      for (int i = 0; i < 6; ++i) {
        double cv = cr[3 * i] * ul[0] + cr[3 * i + 1] * ul[1] + cr[3 * i + 2] * ul[2];
        if (i == 4) cv += -P.fz_max;
        const double z = lam[6 * l + i] + al.rho * cv;
        const double zp = (z > 0.0) ? z : 0.0;
#pragma unroll
        for (int a = 0; a < 3; ++a) gu[a] += zp * cr[3 * i + a];
      }
#pragma unroll
      for (int a = 0; a < 3; ++a) g = fmax(g, fabs(gu[a]));
    }
    // y_k = lx_k + Abar' y_{k+1}; the cost expansion of knot k from the state already in registers
    {
      double xr[13], lxf[13], lx[12];
      if constexpr (MD == MD_CONVEX) {      // cost_expansion's convex branch on the state in registers
        cv_xref_at(P, K.refp, k, xr);
#pragma unroll
        for (int a = 0; a < 3; ++a) {
          lx[a] = P.Q[3 + a] * (x[3 + a] - xr[3 + a]);
          lx[3 + a] = P.Q[a] * (x[a] - xr[a]);
          lx[6 + a] = P.Q[9 + a] * (x[9 + a] - xr[9 + a]);
          lx[9 + a] = P.Q[6 + a] * (x[6 + a] - xr[6 + a]);
        }
      } else {
      xref_at(P, K.refp, k, xr);
#pragma unroll
      for (int i = 0; i < 13; ++i) lxf[i] = P.Q[i] * (x[i] - xr[i]);
      const double dq = xr[3] * x[3] + xr[4] * x[4] + xr[5] * x[5] + xr[6] * x[6];
      const double sg = (dq >= 0.0) ? 1.0 : -1.0;
#pragma unroll
      for (int r = 0; r < 4; ++r) lxf[3 + r] += -sg * P.w * xr[3 + r];
      double G[12];
      quatG(&x[3], G);
#pragma unroll
      for (int a = 0; a < 3; ++a) {
        lx[a] = lxf[a];
        lx[6 + a] = lxf[7 + a];
        lx[9 + a] = lxf[10 + a];
        lx[3 + a] = G[a] * lxf[3] + G[3 + a] * lxf[4] + G[6 + a] * lxf[5] + G[9 + a] * lxf[6];
      }
      }
      const double f0 = y[3], f1 = y[4], f2 = y[5];
#pragma unroll
      for (int a = 0; a < 3; ++a) {
        y[9 + a] += A3[a] * f0 + A3[3 + a] * f1 + A3[6 + a] * f2;
        y[6 + a] += P.h * y[a];
      }
#pragma unroll
      for (int a = 0; a < 3; ++a) y[3 + a] = A1[a] * f0 + A1[3 + a] * f1 + A1[6 + a] * f2;
#pragma unroll
      for (int i = 0; i < 12; ++i) y[i] += lx[i];
    }
  }
  al.stat = g;
}

// set-up of the multipliers for the reference mode (after lane_setup): lambda = 0, slacks at 1
template <int NL>
QL_FN void lane_setup_ref(const DevParams& P, const Ctx& c, const WsOff& O, const LaneState& st, LaneAL& al) {
  const int N = P.N;
  for (int k = 0; k < N; ++k)
#pragma unroll
    for (int l = 0; l < NL; ++l)
      if ((st.con >> l) & 1u)
#pragma unroll
        for (int i = 0; i < 6; ++i) {
          c.W(O.S + 6 * NL * k + 6 * l + i) = 1.0;
          c.W(O.LAM + 6 * NL * k + 6 * l + i) = 0.0;
        }
  al.rho = P.penalty_initial;
  al.irho = 1.0 / al.rho;
  al.J = 0.0; al.Jp = 0.0; al.viol = 0.0; al.dV1 = 0.0; al.alpha = 1.0; al.Jn = 0.0; al.Jnp = 0.0; al.vn = 0.0; al.stp = 0.0;
  al.Jn2 = 0.0; al.Jnp2 = 0.0; al.vn2 = 0.0; al.stp2 = 0.0;
  al.stat = 0.0; al.searching = 0; al.sel = 0;
}

// ---- the whole reference-mode solve of one lane (host build: tests; the kernel runs the same steps in lock step,
// qmpc_lane.hip) -------------------------------------------------------------------------------------------------------------
// Armijo test of the two trials of one pass_C_AL sweep, in the order of the backtracking sequence (trial ls at al.alpha, trial
// ls + 1 at half of it, the second only while ls + 1 <= linesearch_max).  Accepted: al.alpha / Jn / Jnp / vn / stp are those of
// the accepted trial and al.sel names its increments; rejected: al.alpha is the step length of trial ls + 2.
QL_FN bool al_accept_pair(const DevParams& P, LaneAL& al, int ls) {
  const double slack = 1e-12 * fmax(1.0, fabs(al.J));
  if (isfinite(al.Jn) && al.Jn - al.J <= 1e-4 * (al.alpha * al.dV1) + slack) { al.sel = 0; return true; }
  if (ls + 1 <= P.linesearch_max && isfinite(al.Jn2) && al.Jn2 - al.J <= 1e-4 * (0.5 * al.alpha * al.dV1) + slack) {
    al.alpha *= 0.5; al.Jn = al.Jn2; al.Jnp = al.Jnp2; al.vn = al.vn2; al.stp = al.stp2; al.sel = 1;
    return true;
  }
  al.alpha *= 0.25;
  return false;
}
template <int NL, int MD = MD_QUAT>
QL_FN void lane_solve_ref(const DevParams& P, const Ctx& c, const WsOff& O, const LaneK<NL>& K, LaneState& st) {
  LaneAL al;
  lane_setup_ref<NL>(P, c, O, st, al);
  st.it = 1;
  pass_A<NL, false, MD>(P, c, O, K, st, true, (FootPtr)K.foot);      // X <- rollout of U = u_ref
  pass_M<NL, false, MD>(P, c, O, K, st, al);
  int iter = 0;
  st.status = QMPC_MAX_ITER;
  st.last_step = 0.0;
  for (iter = 1; iter <= P.iterations_max; ++iter) {
    if (!pass_B<NL, false, MD, true>(P, c, O, K, st, (FootPtr)K.foot, &al)) { st.status = QMPC_NOT_PD; --iter; break; }
    al.alpha = 1.0;
    bool accepted = false;
    for (int ls = 0; ls <= P.linesearch_max && !accepted; ls += 2) {
      pass_C_AL<NL, 2, MD>(P, c, O, K, st, al, true);          // trials ls and ls + 1
      accepted = al_accept_pair(P, al, ls);
    }
    if (!accepted) { st.status = QMPC_LINESEARCH_FAIL; --iter; break; }
    pass_A_AL<NL, MD>(P, c, O, K, st, al.sel);
    st.last_step = al.stp;
    const double dJ = al.J - al.Jn;
    al.J = al.Jn; al.Jp = al.Jnp; al.viol = al.vn;
    pass_S<NL, MD>(P, c, O, K, st, al);
    if (al.stat < P.tol_stat && al.viol < P.tol_feas) { st.status = QMPC_OK; break; }
    if (al.stat < P.tol_stat || fabs(dJ) < P.tol_cost_int) pass_M<NL, true, MD>(P, c, O, K, st, al);
  }
  if (iter > P.iterations_max) iter = P.iterations_max;
  st.iters = iter;
  st.mu = al.rho;           // the info record's last field is the penalty in this mode
}

// ---- outputs: GetInput(u, 0) (QuatMpc.cpp:264-265) and the info record -----------------------------------------------
// traj_u: this instance's [N][3 NL] input trajectory (the next tick's warm start), or null
template <int NL, int MD = MD_QUAT>
QL_FN void lane_finish(const DevParams& P, const Ctx& c, const WsOff& O, const LaneK<NL>& K, const LaneState& st,
                       double* forces, qmpc_info* info, double* traj_u = nullptr, double* traj_x = nullptr) {
  const int N = P.N;
  const bool solved = st.status != QMPC_NAN_INPUT && st.status != QMPC_NO_CONTACT;
#pragma unroll
  for (int j = 0; j < 3 * NL; ++j) forces[j] = (solved && ((st.con >> (j / 3)) & 1u)) ? c.W(O.U + j) : 0.0;
  if (traj_u)
    for (int k = 0; k < N; ++k)
#pragma unroll
      for (int j = 0; j < 3 * NL; ++j)
        traj_u[3 * NL * k + j] = (solved && ((st.con >> (j / 3)) & 1u)) ? c.W(O.U + 3 * NL * k + j) : 0.0;
  if (traj_x) {      // [N + 1][13] (ConvexMpc's model: [N + 1][12])
    constexpr int NX = (MD == MD_CONVEX) ? 12 : 13;
    for (int k = 0; k <= N; ++k)
#pragma unroll
      for (int i = 0; i < NX; ++i) traj_x[NX * k + i] = solved ? c.W(O.X + 13 * k + i) : 0.0;
  }
  if (!info) return;
  double J = 0.0, viol = 0.0;
  if (solved) {
    double cr[18];
    cone_rows(P, K.rot, cr);
    for (int k = 0; k <= N; ++k) {
      double xr[13];
      if constexpr (MD == MD_CONVEX) cv_xref_at(P, K.refp, k, xr);
      else xref_at(P, K.refp, k, xr);
      double dq = 0.0;
#pragma unroll
      for (int i = 0; i < (MD == MD_CONVEX ? 12 : 13); ++i) {
        const double xv = c.W(O.X + 13 * k + i);
        const double e = xv - xr[i];
        J += 0.5 * P.Q[i] * e * e;
        if (MD != MD_CONVEX && i >= 3 && i < 7) dq += xr[i] * xv;
      }
      if constexpr (MD != MD_CONVEX) J += P.w * (1.0 - fabs(dq));
      if (k == N) break;
#pragma unroll
      for (int l = 0; l < NL; ++l) {
        if (!((st.con >> l) & 1u)) continue;
        double u[3];
#pragma unroll
        for (int a = 0; a < 3; ++a) u[a] = c.W(O.U + 3 * NL * k + 3 * l + a);
        const double e2 = u[2] - st.uz;
        J += 0.5 * P.R[(3 * l) % 12] * u[0] * u[0] + 0.5 * P.R[(3 * l + 1) % 12] * u[1] * u[1] +
             0.5 * P.R[(3 * l + 2) % 12] * e2 * e2;
#pragma unroll
        for (int i = 0; i < 6; ++i) {
          double cv = cr[3 * i] * u[0] + cr[3 * i + 1] * u[1] + cr[3 * i + 2] * u[2];
          if (i == 4) cv += -P.fz_max;
          viol = fmax(viol, fmax(cv, 0.0));
        }
      }
    }
  }
  qmpc_info r = {st.status, st.iters, J, viol, solved ? st.last_step : 0.0, solved ? st.mu : 0.0};
  *info = r;
}

}  // namespace lane
}  // namespace qmpc
