/*
 * qmpc.h -- C ABI of the MI355X-native batched quaternion-MPC inner loop.
 *
 * This is the drop-in boundary for ONE hot path of zixinz990/quaternion-mpc:
 * the per-tick solve inside `legged::QuatMpc::grf_update`
 * (reference legged_ctrl/src/mpc/QuatMpc.cpp:217-265: ALTRO set-up, Solve(),
 * GetInput(0)).  The reference has no FFI layer; its seam is the C++ virtual
 * class `legged::LeggedMpc` (legged_ctrl/include/mpc/LeggedMpc.h:21-28).  The
 * host-side C++ class `legged::QuatMpcHip` (quaternion-mpc_amd/host/QuatMpcHip.h)
 * keeps that virtual surface and calls the functions below instead of
 * constructing an `altro::ALTROSolver`.
 *
 * Conventions
 *   - plain C, no torch / HIP types in any signature (streams are void*)
 *   - all reals are IEEE double (the reference's a_float, QuatMpc.cpp:196)
 *     except the knot spacing, which the reference's dynamics callbacks receive
 *     as `float h` (legged_ctrl/src/utils/AltroUtils.cpp:10,79) -- kept here
 *   - quaternions are (w, x, y, z)       (QuaternionUtils.cpp:8)
 *   - 3x3 matrices are row-major; foot_pos_body is 3x4 COLUMN-major, i.e.
 *     foot_pos_body[3*leg + axis], the memory layout of the reference's
 *     Eigen::Matrix<double,3,4> (LeggedState.h:60)
 *   - leg order FL, FR, RL, RR                     (BaseInterface.cpp:11)
 *   - forces are BODY-frame, 3 per leg, as `optimized_input[0:12]`
 *     (QuatMpc.cpp:269)
 *   - nothing throws or aborts across this boundary; every entry point returns
 *     a qmpc_status and every instance gets its own status word.
 */
#ifndef QMPC_H_
#define QMPC_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define QMPC_NX 13      /* full state  [p(3) q(4) v(3) w(3)]      QuatMpc.cpp:28  */
#define QMPC_NE 12      /* error state [dp(3) phi(3) dv(3) dw(3)] QuatMpc.cpp:212 */
#define QMPC_NU 12      /* 4 legs x (fx,fy,fz) body frame         QuatMpc.cpp:29  */
#define QMPC_NLEG 4     /* LeggedParams.h:9 */
#define QMPC_NC 24      /* friction-cone rows per knot            QuatMpc.cpp:229 */
#define QMPC_MAX_HORIZON 32

/* ---- status codes (call-level and per-instance) -------------------------- */
typedef enum qmpc_status {
  QMPC_OK = 0,              /* converged to the stated tolerances              */
  QMPC_MAX_ITER = 1,        /* iteration cap hit; forces are the last iterate
                               (this is what the reference silently returns:
                               SolveStatus ignored at QuatMpc.cpp:256)          */
  QMPC_NO_CONTACT = 2,      /* no stance leg: reference computes 0/0 at
                               QuatMpc.cpp:122; we return zero forces          */
  QMPC_NAN_INPUT = 3,       /* non-finite input record; zero forces            */
  QMPC_LINESEARCH_FAIL = 4, /* no step length reduced the merit function       */
  QMPC_NOT_PD = 5,          /* Quu lost positive definiteness                  */
  /* call-level only */
  QMPC_BAD_ARGUMENT = 16,
  QMPC_NO_DEVICE = 17,      /* HIP runtime / device missing: fail loudly       */
  QMPC_HIP_ERROR = 18,
  QMPC_BATCH_TOO_LARGE = 19,
  QMPC_UNSUPPORTED = 20     /* optional dependency missing (RCCL for qmpc_gather)  */
} qmpc_status;

/* ---- solver mode --------------------------------------------------------- */
typedef enum qmpc_mode {
  /* Converged KKT point of the NLP the reference poses (default): primal-dual
     interior point on the iLQR/Riccati core, run to tol_step / ipm_mu_final.   */
  QMPC_MODE_CONVERGED = 0,
  /* Reference-style truncated AL-iLQR: iterations_max=10, penalty_scaling=20
     (QuatMpc.cpp:22,26) and the upstream ALTRO tolerances (1e-4): the iterate the
     reference's own solver returns (status ignored, QuatMpc.cpp:256).  On the device
     for all three models.                                                       */
  QMPC_MODE_REFERENCE = 1
} qmpc_mode;

/* ---- which reference controller's inner loop ----------------------------- */
typedef enum qmpc_model {
  QMPC_MODEL_QUAT = 0,      /* legged::QuatMpc::grf_update   (QuatMpc.cpp:109-276)   */
  QMPC_MODEL_CONVEX = 1,    /* legged::ConvexMpc::grf_update (ConvexMpc.cpp:81-198):
                               12-state Euler-angle SRBD, world-frame forces    */
  QMPC_MODEL_QUAT8 = 2      /* the QuatMpc problem with 8 contact points (24 forces,
                               48 cone rows per knot): BASELINE.json config 5, the
                               SYNTHETIC stand-in for the humanoid branch that is not
                               in the reference checkout           */
} qmpc_model;

/* ---- shared, read-only problem parameters -------------------------------- */
/* Sources: legged_ctrl/config/gazebo_go1_quat_mpc.yaml:36-75,115-122 and
 * LeggedState.h:160-244 (LeggedParam). */
typedef struct qmpc_params {
  int32_t horizon;          /* N knots; param.mpc_horizon (yaml: 20)           */
  float   h;                /* knot spacing [s] AS FLOAT = (float)(mpc_update_period/1000) */
  double  h_ref;            /* knot spacing used by the reference trajectory,
                               double: i*h/1000.0 at QuatMpc.cpp:156-157       */
  double  mass;             /* param.robot_mass                                */
  double  inertia[9];       /* row-major; QuatMpc passes 1.2*trunk_inertia (QuatMpc.cpp:182) */
  double  q_weights[13];    /* diag Q on the FULL state                        */
  double  r_weights[12];    /* diag R                                          */
  double  w;                /* quaternion cost weight  w*(1-|qref'q|)          */
  double  mu;               /* friction coefficient                            */
  double  fz_max;
  int32_t mode;             /* qmpc_mode                                       */
  int32_t iterations_max;   /* both modes (reference: 10, QuatMpc.cpp:22)      */
  /* reference mode: AL-iLQR options (AltroOptions at QuatMpc.cpp:21-26 +
     upstream ALTRO defaults)                                                  */
  double  penalty_initial;
  double  penalty_scaling;
  double  penalty_max;
  double  tol_stationarity;
  double  tol_feasibility;        /* both modes (cone violation / |c+s|)       */
  double  tol_cost_intermediate;  /* dual update trigger                       */
  /* converged mode: interior-point options                                    */
  double  tol_step;               /* stop when the full Newton step |dU|_inf <=
                                     tol_step [N] ...                           */
  double  ipm_mu0;                /* initial barrier: s0=max(-c,1), lam0=mu0/s0 */
  double  ipm_mu_final;           /* ... and the barrier is <= ipm_mu_final    */
  double  ipm_sigma;              /* centering parameter                       */
  double  ipm_sigma_fast;         /* centering once full steps are taken       */
  double  ipm_tau;                /* fraction to the boundary                  */
  int32_t linesearch_max;         /* reference mode: max halvings              */
  /* parity switches for the reference's quirks */
  int32_t drop_ang_vel;     /* 1: x_init[10:13]=0 (comma-initialiser bug at
                               QuatMpc.cpp:242-245); 0: use ang_vel_body.  The default (1) is the
                               reference's behaviour; a closed loop around an ideal plant needs 0
                               beyond a few seconds                */
  int32_t model;            /* qmpc_model: which controller's problem the handle
                               solves (0 = QuatMpc, 1 = ConvexMpc)              */
} qmpc_params;

/* Fill *p with the Go1 values of gazebo_go1_quat_mpc.yaml and the solver
 * defaults for `mode`.  horizon: 10 or 20 in BASELINE.json. */
void qmpc_default_params(qmpc_params* p, int32_t horizon, int32_t mode);

/* ---- one MPC instance ----------------------------------------------------
 * Exactly the LeggedState fields grf_update reads, 48
 * doubles = 384 B, so one wavefront fetches one record with one coalesced
 * 8-byte-per-lane load. */
typedef struct qmpc_input {
  double quat[4];           /* fbk.torso_quat (w,x,y,z)       QuatMpc.cpp:236-239 */
  double rot[9];            /* fbk.torso_rot_mat, body->world :184-186,203,213   */
  double lin_vel_body[3];   /* R' * fbk.torso_lin_vel_world   :231               */
  double ang_vel_body[3];   /* fbk.torso_ang_vel_body (dropped when drop_ang_vel) */
  double foot_pos_body[12]; /* fbk.foot_pos_body, 3x4 col-major                  */
  double contacts[4];       /* ctrl.plan_contacts as 0.0 / 1.0 :118-125          */
  double pos_ref_body[3];   /* torso_pos_d_body_filtered      :156-158           */
  double vel_ref_body[3];   /* torso_lin_vel_d_body_filtered  :156-157,169       */
  double acc_ref_body[3];   /* 0 in QuatMpc; lets the accelerating reference of
                               TestAltroTrotQuatMpc.cpp:67-70 be expressed       */
  double quat_d[4];         /* ctrl.torso_quat_d AFTER the :128-137 update       */
} qmpc_input;

/* ---- one 8-contact-point instance (BASELINE config 5, synthetic) -----------
 * Same fields as qmpc_input with 8 contact points (two feet x 4 corner points):
 * 64 doubles = 512 B, one 8-byte-per-lane load of a whole wavefront. */
#define QMPC_NLEG8 8
#define QMPC_NU8 24
typedef struct qmpc_input8 {
  double quat[4];
  double rot[9];
  double lin_vel_body[3];
  double ang_vel_body[3];
  double foot_pos_body[24]; /* [3*point + axis]                                   */
  double contacts[8];
  double pos_ref_body[3];
  double vel_ref_body[3];
  double acc_ref_body[3];
  double quat_d[4];
} qmpc_input8;

/* ---- one ConvexMpc instance -------------------------
 * The LeggedState fields legged::ConvexMpc::grf_update reads
 * (legged_ctrl/src/mpc/ConvexMpc.cpp:81-198), again 48 doubles = 384 B.
 * State order of that controller: [roll pitch yaw, pos(3), ang_vel_world(3),
 * lin_vel_world(3)] (ConvexMpc.cpp:156-167); forces are WORLD-frame
 * (the caller rotates them: optimized_input = R' u, ConvexMpc.cpp:188-190). */
typedef struct qmpc_convex_input {
  double euler[3];            /* fbk.torso_euler                 ConvexMpc.cpp:156-158 */
  double pos_world[3];        /* fbk.torso_pos_world             :159-161             */
  double ang_vel_world[3];    /* fbk.torso_ang_vel_world         :162-164             */
  double lin_vel_world[3];    /* fbk.torso_lin_vel_world         :165-167             */
  double foot_pos_abs_com[12];/* fbk.foot_pos_abs_com, 3x4 col-major :115,118          */
  double contacts[4];         /* ctrl.plan_contacts as 0.0 / 1.0 :92,107-110          */
  double pos_d_world[3];      /* ctrl.torso_pos_d_world          :99-101              */
  double lin_vel_d_world[3];  /* ctrl.torso_lin_vel_d_world (x, y used; vz ref = 0) :105-107 */
  double yaw_rate_d;          /* ctrl.torso_ang_vel_d_body[2]    :98,104              */
  double reserved[13];        /* must be finite (0)                                    */
} qmpc_convex_input;

/* ---- per-instance result ------------------------------------------------- */
typedef struct qmpc_info {
  int32_t status;           /* qmpc_status                                     */
  int32_t iterations;       /* AL-iLQR iterations taken                        */
  double  cost;             /* final (un-augmented) objective                  */
  double  max_violation;    /* max(c,0) over all cone rows                     */
  double  last_step;        /* |dU|_inf of the final iteration [N]             */
  double  penalty;          /* final AL penalty                                */
} qmpc_info;

typedef struct qmpc_handle qmpc_handle;

/* ---- lifetime ------------------------------------------------------------ */
/* Creates a solver bound to HIP device `device` with capacity for `max_batch`
 * instances.  Fails with QMPC_NO_DEVICE when there is no GPU: there is NO CPU
 * fallback behind this library. */
qmpc_status qmpc_create(const qmpc_params* params, int32_t max_batch, int32_t device,
                        qmpc_handle** out);
qmpc_status qmpc_set_params(qmpc_handle* h, const qmpc_params* params);
void        qmpc_destroy(qmpc_handle* h);

/* ---- solve --------------------------------------------------------------- */
/* Synchronous, host buffers (replaces QuatMpc.cpp:217-265 for `batch`
 * independent LeggedStates).  forces_body: [batch][12]; info may be NULL;
 * traj_u ([batch][N][12]) and traj_x ([batch][N+1][13]) may be NULL (the first call that asks for a trajectory
 * allocates its device buffer, sized by max_batch; later calls reuse it).
 * batch == 1 keeps the blocking semantics of the reference's mpc_thread
 * (Main.cpp:106-108). */
qmpc_status qmpc_solve(qmpc_handle* h, int32_t batch, const qmpc_input* in,
                       double* forces_body, qmpc_info* info);
qmpc_status qmpc_solve_traj(qmpc_handle* h, int32_t batch, const qmpc_input* in,
                            double* forces_body, qmpc_info* info,
                            double* traj_u, double* traj_x);

/* Warm-started solve (QuatMpc handle, converged mode): every instance starts from u_init [batch][N][12] -- the caller's
 * previous solution of the same robot, e.g. the traj_u of last tick's call; it is shifted by one knot inside, swing legs
 * are pinned to 0 and a leg that has just landed starts from u_ref -- instead of u_ref (QuatMpc.cpp:253).  u_init NULL =
 * a cold solve.  traj_u [batch][N][12] (may be NULL; may be the same buffer as u_init) receives the new solution.  The
 * KKT point is the one qmpc_solve finds; with params.ipm_mu0 lowered to 1e-6 it takes about half the iterations for a
 * robot in its gait.  Host buffers / device buffers (stream-ordered). */
qmpc_status qmpc_solve_warm(qmpc_handle* h, int32_t batch, const qmpc_input* in, const double* u_init,
                            double* forces_body, qmpc_info* info, double* traj_u);
qmpc_status qmpc_solve_warm_device(qmpc_handle* h, int32_t batch, const qmpc_input* d_in, const double* d_u_init,
                                   double* d_forces_body, qmpc_info* d_info, double* d_traj_u, void* stream);

/* Stream-ordered, DEVICE buffers (inputs already resident in HBM).  `stream`
 * is a hipStream_t passed as void* (NULL = the handle's own stream).  Nothing
 * is synchronised; pair with qmpc_wait or your own stream sync.
 *
 * ONE launch in flight per HANDLE (not per stream): the handle owns mutable device state -- the gains workspace of the
 * mid-size kernels, the lane kernel's workspace / sort scratch, the hand-off list -- so a second qmpc_solve*_device on
 * the same handle must be stream-ordered after the first (same stream, or an event).  Independent batches in flight
 * take one handle each (bench.py: two_in_flight).
 *
 * Which kernel runs is chosen from the batch size (converged mode; QuatMpc N <= 12: everything in LDS up to 1024
 * instances, gains in a workspace up to 14335, from 14336 on one LANE per instance -- a lane PAIR while the batch fills only
 * half of every wavefront -- with the stragglers handed back to the wave kernel; N = 13 .. 22: the lane kernel from 14848;
 * warm-started launches: 18432 / 20480; the thresholds of the other models are in qmpc_hip.hip, and
 * qmpc_query(QMPC_QUERY_KERNEL_FOR_BATCH) answers for a given handle).  The kernel families solve the same
 * problem to the same KKT point but round differently: forces agree to ~1e-10 N across a threshold (tested to 1e-7 N),
 * bit for bit within a family and for a shard of a batch against the whole batch.  Their device buffers are allocated on
 * the first call that needs them, sized by max_batch, and held until qmpc_destroy: the lane kernel's workspace (<= 1024
 * wavefronts x 0.84 MB at N=10: 0.86 GB) and the state records of the straggler hand-off (8 + 84 N doubles per instance
 * of max_batch: 445 MB at 65536 x N=10) -- create the handle with the max_batch you mean to use. */
qmpc_status qmpc_solve_device(qmpc_handle* h, int32_t batch, const qmpc_input* d_in,
                              double* d_forces_body, qmpc_info* d_info, void* stream);
qmpc_status qmpc_wait(qmpc_handle* h);

/* Host buffers, NOT blocking: H2D copy, kernel and D2H copies are queued on the handle's stream and the call
 * returns; qmpc_wait(h) completes them.  `in`, `forces_body` and `info` must stay valid until then, and `in` must not
 * be modified before qmpc_wait when it is pinned (see below).  Lets the caller of the controller thread
 * (Main.cpp:103-118) do other work during the solve. */
qmpc_status qmpc_solve_async(qmpc_handle* h, int32_t batch, const qmpc_input* in,
                             double* forces_body, qmpc_info* info);

/* How the host-buffer calls (qmpc_solve, qmpc_solve_traj, qmpc_solve_async, qmpc_convex_solve*, qmpc_solve8*) move their
 * data -- SURVEY.md 8d's metric is exactly this call: records in host memory -> forces in host memory.
 *   - Batches that take a wave-per-instance kernel (every batch below the lane kernel's threshold: 14335 instances for QuatMpc at N <= 12; qmpc_query(QMPC_QUERY_KERNEL_FOR_BATCH) answers for a handle) run ZERO-COPY: every wavefront reads
 *     its 384-byte record from, and writes its forces / status record to, host memory the device can address.  Buffers
 *     from qmpc_host_alloc (or hipHostMalloc / hipHostRegister) are used in place; pageable buffers go through pinned
 *     staging the handle owns (one memcpy in, one out).  One launch, one synchronisation; results are bit-identical to
 *     qmpc_solve_device.  QMPC_ZERO_COPY=0 restores explicit copies.
 *   - Lane-kernel batches (which sort and re-read their records) use hipMemcpyAsync on the handle's stream; with pinned
 *     buffers those copies are true DMA. */
void*       qmpc_host_alloc(size_t bytes);      /* pinned, device-addressable host memory (NULL on failure) */
void        qmpc_host_free(void* p);

/* Allocate NOW whatever a solve (or closed loop) of up to `batch` instances on this handle needs later: the lane kernel's
 * workspace (<= 0.86 GB at N=10), the state records of the straggler hand-off (8 + 84 N doubles per instance of max_batch),
 * the pinned staging of the host-buffer calls.  After it returns no solve of up to `batch` instances allocates -- e.g.
 * inside the caller's own stream capture -- and qmpc_query tells whether the hand-off is available.  Optional: without it
 * the same buffers are allocated by the first call that needs them. */
qmpc_status qmpc_prepare(qmpc_handle* h, int32_t batch);

/* Handle state the results can depend on, and which kernel family a batch size selects. */
enum qmpc_query_what {
  QMPC_QUERY_HANDOFF_ACTIVE       = 1,  /* 1: lane-kernel batches hand their stragglers to the wave kernel (results: the
                                           hand-off's rounding family); 0: pure lane kernel (switched off, another model / mode,
                                           or the records could not be allocated -- ~1e-10 N apart, not bit-identical) */
  QMPC_QUERY_HANDOFF_ALLOC_FAILED = 2,  /* 1: the records' allocation failed on this handle (also reported on stderr) */
  QMPC_QUERY_KERNEL_FOR_BATCH     = 3,  /* arg = batch: the qmpc_kernel_family a plain solve of that size launches */
  QMPC_QUERY_LAST_KERNEL          = 4,  /* family of the most recent solve launch */
  QMPC_QUERY_LANE_CAP             = 5,  /* arg = 1 plain solve / 2 cold closed loop / 3 warm closed loop: iteration cap of the
                                           capped lane launch (0: no hand-off) */
  QMPC_QUERY_DEVICE_BYTES         = 6,  /* device memory the handle holds right now */
  QMPC_QUERY_ZERO_COPY            = 7   /* 1: host-buffer calls of wave-kernel batches run zero-copy */
};
enum qmpc_kernel_family {
  QMPC_KERNEL_NONE         = 0,
  QMPC_KERNEL_WFORM_LDS    = 1,   /* wave per instance, wrench form, everything in LDS */
  QMPC_KERNEL_WFORM_WS     = 2,   /* ... gains in the workspace */
  QMPC_KERNEL_DENSE_LDS    = 3,   /* wave per instance, dense 12x12 stage algebra (round-1 family), everything in LDS */
  QMPC_KERNEL_DENSE_WS     = 4,   /* ... gains (and slack arrays) in the workspace */
  QMPC_KERNEL_LANE         = 5,   /* lane per instance */
  QMPC_KERNEL_LANE_HANDOFF = 6    /* lane per instance to an iteration cap, stragglers continued by the wave kernel */
};
qmpc_status qmpc_query(qmpc_handle* h, int32_t what, int64_t arg, int64_t* value);

/* Multi-GPU: the single collective of the path.  All-gathers `count` doubles per rank (e.g. the
 * [B/G][12] force block, or forces + qmpc_info records laid out in one buffer) from every rank's `d_local` into
 * `d_all` ([ranks * count], rank order) with RCCL's ncclAllGather on `stream` (NULL = the handle's stream),
 * stream-ordered after the solve that produced `d_local`.  `nccl_comm` is the caller's ncclComm_t (one per GPU /
 * process, created by the host program).  RCCL is looked up at run time; QMPC_UNSUPPORTED when it is absent. */
qmpc_status qmpc_gather(qmpc_handle* h, void* nccl_comm, const double* d_local, int64_t count,
                        double* d_all, void* stream);

/* Time (ms, HIP events on the launch stream) of the most recent
 * qmpc_solve_device / qmpc_solve kernel, after it completed. */
qmpc_status qmpc_last_kernel_ms(qmpc_handle* h, float* ms);

/* Batched linearisation only: for every instance and knot
 * k<N, roll out x from the reference's initial guess U=u_ref and return the
 * error-state Jacobians  Abar [batch][N][12][12], Bbar [batch][N][12][12]
 * (row-major) and the rollout X [batch][N+1][13].  Host buffers. */
qmpc_status qmpc_linearize(qmpc_handle* h, int32_t batch, const qmpc_input* in,
                           double* Abar, double* Bbar, double* X);

/* ---- ConvexMpc entry points (handle created with params.model = QMPC_MODEL_CONVEX)
 * Replace ConvexMpc.cpp:84-186 (ALTRO set-up, Solve(), GetInput(0)).
 * forces_world: [batch][12]; traj_u [batch][N][12], traj_x [batch][N+1][12] may be NULL.
 * Calling a quaternion entry point on a convex handle (or vice versa) returns
 * QMPC_BAD_ARGUMENT. */
void        qmpc_default_convex_params(qmpc_params* p, int32_t horizon, int32_t mode);
qmpc_status qmpc_convex_solve(qmpc_handle* h, int32_t batch, const qmpc_convex_input* in,
                              double* forces_world, qmpc_info* info);
qmpc_status qmpc_convex_solve_traj(qmpc_handle* h, int32_t batch, const qmpc_convex_input* in,
                                   double* forces_world, qmpc_info* info,
                                   double* traj_u, double* traj_x);
qmpc_status qmpc_convex_solve_device(qmpc_handle* h, int32_t batch, const qmpc_convex_input* d_in,
                                     double* d_forces_world, qmpc_info* d_info, void* stream);
/* Discrete Jacobians A [batch][N][12][12], B [batch][N][12][12] (row-major, as the
 * reference's midpoint_jacobian of ct_srb_jacobian produces them, AltroUtils.cpp:78-110,
 * 297-359) along the rollout X [batch][N+1][12] of U = u_ref.  Host buffers. */
qmpc_status qmpc_convex_linearize(qmpc_handle* h, int32_t batch, const qmpc_convex_input* in,
                                  double* A, double* B, double* X);

/* ---- 8-contact-point entry points (handle created with params.model = QMPC_MODEL_QUAT8)
 * forces_body: [batch][24]; traj_u [batch][N][24], traj_x [batch][N+1][13] may be NULL.
 * r_weights[j % 12] is used for input j.  The defaults are a synthetic 30 kg biped
 * (stated in DESIGN.md); nothing in the reference pins them. */
void        qmpc_default_biped8_params(qmpc_params* p, int32_t horizon, int32_t mode);
qmpc_status qmpc_solve8(qmpc_handle* h, int32_t batch, const qmpc_input8* in,
                        double* forces_body, qmpc_info* info);
qmpc_status qmpc_solve8_traj(qmpc_handle* h, int32_t batch, const qmpc_input8* in,
                             double* forces_body, qmpc_info* info, double* traj_u, double* traj_x);
qmpc_status qmpc_solve8_device(qmpc_handle* h, int32_t batch, const qmpc_input8* d_in,
                               double* d_forces_body, qmpc_info* d_info, void* stream);

/* ---- force -> joint-torque consumer --------------------
 * The step right after the path: BaseInterface::tau_ctrl_update
 * (legged_ctrl/src/interfaces/BaseInterface.cpp:343-408) maps the body-frame foot
 * forces to joint torques, tau_leg = -J_leg(q)' f_leg (:368,401), with the leg
 * Jacobian of A1Kinematics::jac (legged_ctrl/src/utils/A1Kinematics.cpp:15-19,
 * evaluated in fbk.jac_foot at BaseInterface.cpp:209-212).  Batched here so that
 * Monte-Carlo sweeps get joint torques without leaving the GPU.
 * Leg geometry = the reference's rho_fix / rho_opt vectors (BaseInterface.cpp:10-34). */
typedef struct qmpc_leg_geometry {
  double rho_fix[4][5];     /* per leg: offset_x, offset_y, motor_offset, upper, lower length */
  double rho_opt[4][3];     /* per leg: contact-point offsets (0 in the reference)           */
} qmpc_leg_geometry;
void qmpc_default_go1_geometry(qmpc_leg_geometry* g);   /* BaseInterface.cpp:12-26, LeggedParams.h:14-15 */

/* Foot position in the body frame and leg Jacobian for every instance and leg
 * (A1Kinematics::fk / ::jac).  joint_pos [batch][12] (leg-major: hip, thigh, calf);
 * foot_pos_body [batch][12] (3x4 col-major, [3*leg+axis]); jac [batch][4][9], each
 * 3x3 COLUMN-major like Eigen's Matrix3d::data().  Either output may be NULL.  Host buffers. */
qmpc_status qmpc_leg_kinematics(qmpc_handle* h, const qmpc_leg_geometry* g, int32_t batch,
                                const double* joint_pos, double* foot_pos_body, double* jac);
/* tau [batch][12] = -J' f per leg; legs with contacts == 0 get zero torque when
 * `walking` != 0 (movement_mode > 0, BaseInterface.cpp:366-370); with walking == 0
 * every leg is mapped (:401).  contacts [batch][4] may be NULL (= all stance).  Host buffers. */
qmpc_status qmpc_torque_map(qmpc_handle* h, const qmpc_leg_geometry* g, int32_t batch,
                            const double* joint_pos, const double* forces_body,
                            const double* contacts, int32_t walking, double* tau);
/* Same with DEVICE buffers, stream-ordered (NULL = the handle's stream), e.g. straight
 * after qmpc_solve_device on the forces it produced. */
qmpc_status qmpc_torque_map_device(qmpc_handle* h, const qmpc_leg_geometry* g, int32_t batch,
                                   const double* d_joint_pos, const double* d_forces_body,
                                   const double* d_contacts, int32_t walking, double* d_tau,
                                   void* stream);

/* ---- the whole low-level command of a tick (completed) -----
 * BaseInterface::tau_ctrl_update (BaseInterface.cpp:343-408) turns the outputs of grf_update into what the
 * 1 kHz joint loop sends to the motors: per leg
 *   walking (movement_mode > 0):
 *     joint_ang_tgt = inv_kin(R'(optimized_state[6+3i] - torso_pos_world), joint_pos)     (:349-355; NaN -> joint_pos)
 *     joint_vel_tgt = J^-1 R'(optimized_input[12+3i] - torso_lin_vel_world)               (:358-364; NaN -> joint_vel)
 *     joint_tau_tgt = plan_contacts ? -J' optimized_input[3i] : 0                         (:367-371)
 *   standing: joint_tau_tgt = -J' f, targets = the measured joint_pos / joint_vel          (:400-403)
 * with J = A1Kinematics::jac(joint_pos) (BaseInterface.cpp:209-212) and A1Kinematics::inv_kin
 * (A1Kinematics.cpp:335-459: closed-form, single-precision atan2 approximation :291-312, the hip-angle branch
 * nearest to the current angle). */
typedef struct qmpc_joint_feedback {
  double joint_pos[12], joint_vel[12];                     /* fbk.joint_pos / joint_vel, leg-major (hip, thigh, calf)  */
  double torso_pos_world[3], torso_quat[4] /* w,x,y,z */, torso_lin_vel_world[3];
  double foot_pos_target_world[12];                        /* ctrl.optimized_state.segment<12>(6),  QuatMpc.cpp:270    */
  double foot_vel_target_world[12];                        /* ctrl.optimized_input.segment<12>(12), QuatMpc.cpp:271    */
  double forces_body[12];                                  /* ctrl.optimized_input.segment<12>(0),  QuatMpc.cpp:267    */
  double plan_contacts[4];                                 /* ctrl.plan_contacts (0 / 1)                               */
  double movement_mode;
} qmpc_joint_feedback;                                     /* 75 doubles */
typedef struct qmpc_joint_command {
  double joint_ang_tgt[12], joint_vel_tgt[12], joint_tau_tgt[12];
} qmpc_joint_command;                                      /* 36 doubles */
/* A1Kinematics::inv_kin for every instance and leg: foot_pos_body [batch][12], cur_joint_pos [batch][12] (branch
 * selection), joint_pos [batch][12] out (NaN where the foot is out of reach, as in the reference).  Host buffers. */
qmpc_status qmpc_leg_inverse_kinematics(qmpc_handle* h, const qmpc_leg_geometry* g, int32_t batch,
                                        const double* foot_pos_body, const double* cur_joint_pos, double* joint_pos);
/* tau_ctrl_update for `batch` robots.  Host buffers / device buffers (stream-ordered, NULL = the handle's stream). */
qmpc_status qmpc_joint_commands(qmpc_handle* h, const qmpc_leg_geometry* g, int32_t batch,
                                const qmpc_joint_feedback* fb, qmpc_joint_command* cmd);
qmpc_status qmpc_joint_commands_device(qmpc_handle* h, const qmpc_leg_geometry* g, int32_t batch,
                                       const qmpc_joint_feedback* d_fb, qmpc_joint_command* d_cmd, void* stream);

/* ---- device-resident closed loop -----------------------
 * The step BEFORE the path, the path and a plant chained on the GPU, state kept in HBM, one tick =
 *   feedback  (R, R_z, foot_pos_body, contact flags from the plant state)
 *   Raibert foothold targets              BaseInterface.cpp:266-288
 *   goal_update                            QuatMpc.cpp:68-107  (six 100-sample moving averages, MovingWindowFilter.hpp)
 *   foot_update: gait FSM + swing quintic  QuatMpc.cpp:278-305, LeggedContactFSM.cpp:33-86,208-246, Utils.cpp:236-293
 *   record packing (torso_quat_d update)   QuatMpc.cpp:112-176,231-246
 *   qmpc_solve_device                      QuatMpc.cpp:217-265
 *   outputs                                QuatMpc.cpp:263-273
 *   plant: single rigid body under the forces, explicit midpoint, dt = 5 ms (this repository's; the reference
 *          closes its loop through Gazebo / the robot); swing feet track the FSM target, stance feet stay.
 * The host classes (host/QuatMpcHip.h + host/ClosedLoopHost.h) run the same tick on the CPU and are the parity
 * reference, tick for tick.  Integers are stored as doubles so that the record is 820 doubles for every binding. */
#define QMPC_LOOP_WINDOW 100
typedef struct qmpc_loop_filter {      /* MovingWindowFilter.hpp:14-63 */
  double ring[QMPC_LOOP_WINDOW];
  double head, count, sum, correction;
} qmpc_loop_filter;
typedef struct qmpc_loop_leg {         /* LeggedContactFSM.h:60-108 */
  double gait_phase, state /* 0 swing, 1 stance */, pattern_index, prev_pattern_index, start_time, end_time,
         not_first_call;
  double swing_start[3], swing_end[3], swing_extend[3];
  double fsm_pos[3], fsm_vel[3], fsm_acc[3];   /* FSM_foot_{pos,vel,acc}_target_world */
  double terrain_height;
} qmpc_loop_leg;
typedef struct qmpc_loop_state {
  /* plant */
  double pos_world[3], quat[4], lin_vel_world[3], ang_vel_body[3], foot_pos_world[12];
  /* command: joy.{velx, vely, body_height, roll_rate, pitch_rate, yaw_rate}, ctrl.movement_mode */
  double joy[6], movement_mode;
  double sin_ang_vel, attitude_traj_count;   /* joy.sin_ang_vel (LeggedState.h:157): the reference's attitude-sweep test mode,
                                                torso_quat_d = euler_to_quat(pi/8 sin(2 pi k / 900) (1,1,1)), QuatMpc.cpp:138-146 */
  /* controller memory */
  double pos_d_world[3], pos_d_init, quat_d[4], lin_vel_d_rel[3];
  qmpc_loop_filter vel_filter[3], pos_filter[3];
  qmpc_loop_leg leg[4];
  /* outputs of the last tick */
  double contacts[4], gait_counter[4], forces_body[12], grf_world[12], foot_target_world[12];
  double status, iterations, tick;
} qmpc_loop_state;
typedef struct qmpc_loop_params {
  double gait_freq;                 /* param.gait_freq (yaml: 2.2)                      */
  double default_foot_pos_rel[12];  /* param.default_foot_pos_rel, [3*leg+axis]         */
  double dt;                        /* tick: the reference's hard-wired 5 ms (QuatMpc.cpp:97-98,132,294) */
  double contact_height;            /* plant: foot_contact_flag = foot z <= this [m]    */
  double warm_start;                /* 0 (default): every solve starts from u_ref like the reference (QuatMpc.cpp:253).
                                       != 0: from the second tick of a call on, the solve starts from the previous
                                       tick's solution shifted by one knot (swing legs 0, a leg that has just landed
                                       from u_ref); converged mode.  The persistent kernel keeps that solution in LDS,
                                       the per-tick form in the handle's trajectory buffer.  Same KKT points, about half the iterations when
                                       combined with a low params.ipm_mu0 (1e-6): DESIGN.md 3e */
} qmpc_loop_params;
void qmpc_default_loop_params(qmpc_loop_params* p);
/* Host-side initialiser (no GPU involved): robot standing at `height` over its default footholds, at rest,
 * yaw `yaw`; joy[6] and movement_mode as above. */
void qmpc_loop_state_init(qmpc_loop_state* s, const qmpc_loop_params* lp, const double joy[6], double movement_mode,
                          double height, double yaw);
/* `ticks` ticks for `batch` instances.  Host buffers in/out; trace_forces [ticks][batch][12] (body frame) and
 * trace_contacts [ticks][batch][4] may be NULL.  The handle is a QuatMpc handle (QMPC_MODEL_QUAT, either solver mode) or a
 * ConvexMpc handle (QMPC_MODEL_CONVEX, either solver mode): the latter runs ConvexMpc's tick -- its goal_update
 * (ConvexMpc.cpp:51-79; pos_d_world[0:2] hold joy.body_x / body_y, roll / pitch rate commands are ignored), its feedback
 * and record, R' u into the plant -- with host/ClosedLoopHost.h over ConvexMpcHipT as the parity reference.  The device
 * tick of ConvexMpc carries the controller period as the literal 5 ms (as the QuatMpc tick does upstream): a ConvexMpc
 * handle whose knot spacing params.h is not 5 ms is refused with QMPC_UNSUPPORTED. */
qmpc_status qmpc_loop_run(qmpc_handle* h, const qmpc_loop_params* lp, int32_t batch, qmpc_loop_state* states,
                          int32_t ticks, double* trace_forces, double* trace_contacts);
/* The same with DEVICE buffers, stream-ordered (NULL = the handle's stream).  Up to 2048 robots (4096 with
 * lp->warm_start) the whole loop is ONE launch of a persistent kernel in which a wavefront owns a robot for all ticks
 * (both controllers, both solver modes); beyond, the per-tick kernel sequence is captured once into a hipGraph and
 * replayed `ticks` times.  The two forms give the same bits (QMPC_LOOP_FUSED=0 / 1 forces one).
 * For roll-outs longer than a few seconds set params.drop_ang_vel = 0 (see qmpc_params). */
qmpc_status qmpc_loop_run_device(qmpc_handle* h, const qmpc_loop_params* lp, int32_t batch, qmpc_loop_state* d_states,
                                 int32_t ticks, double* d_trace_forces, double* d_trace_contacts, void* stream);
int32_t qmpc_sizeof_loop_state(void);
/* Joint-level commands of the robots of a closed loop, from their states after a tick (device buffers): the plant has
 * massless legs, so the measured joint angles are inv_kin of the plant's foot positions; d_joint_pos [batch][12] is
 * in/out (in: the angles of the previous tick, which select the hip branch - initialise with
 * qmpc_loop_joint_init; out: this tick's), the joint velocities are J^-1 R'(foot velocity - torso velocity) with
 * swing feet moving at their FSM target velocity.  d_fb (may be NULL) receives the feedback records that were built,
 * d_cmd the commands: exactly qmpc_joint_commands_device on those records. */
qmpc_status qmpc_loop_joint_commands_device(qmpc_handle* h, const qmpc_leg_geometry* g, int32_t batch,
                                            const qmpc_loop_state* d_states, double* d_joint_pos,
                                            qmpc_joint_feedback* d_fb, qmpc_joint_command* d_cmd, void* stream);
/* The same with HOST buffers (states as qmpc_loop_run leaves them; fb may be NULL). */
qmpc_status qmpc_loop_joint_commands(qmpc_handle* h, const qmpc_leg_geometry* g, int32_t batch,
                                     const qmpc_loop_state* states, double* joint_pos, qmpc_joint_feedback* fb,
                                     qmpc_joint_command* cmd);
/* The closed loop down to the motors: qmpc_loop_run_device with the joint level closing every tick (inside the
 * persistent kernel, or as the fourth kernel of the captured per-tick graph).  d_joint_pos [batch][12] in/out as above;
 * d_cmd [batch] receives the commands of the LAST tick, d_trace_cmd [ticks][batch] those of every tick (either may be
 * NULL, not both). */
qmpc_status qmpc_loop_run_joint_device(qmpc_handle* h, const qmpc_loop_params* lp, const qmpc_leg_geometry* g,
                                       int32_t batch, qmpc_loop_state* d_states, double* d_joint_pos, int32_t ticks,
                                       qmpc_joint_command* d_cmd, qmpc_joint_command* d_trace_cmd, void* stream);
/* Stand-pose joint angles (0, 0.67, -1.3 per leg: the reference's Gazebo start pose) for `batch`
 * robots, host buffer [batch][12]. */
void qmpc_loop_joint_init(double* joint_pos, int32_t batch);

/* ---- diagnostics ----------------------------------------------------------- */
/* C = X' * Y on [12][16] row-major tiles through the FP64 MFMA path the solver
 * uses (host buffers of 192 doubles each).  Lets the GPU tests pin the
 * fragment layout independently of the solver. */
qmpc_status qmpc_selftest_mtm(int32_t device, const double* X, const double* Y, double* C);
/* Cross-lane primitives of the stage solve on 64 doubles: out = 9 x 64 doubles
 * (row-group broadcasts 0..3, wave sum / max / min, row_newbcast:5, quad_perm[1,1,1,1]). */
qmpc_status qmpc_selftest_lanes(int32_t device, const double* in, double* out);
/* Per-instance phase cycle counts (s_memtime) of one instrumented solve launch:
 * cycles_out [batch][16] int64 (host); slots 0..14 = set-up, expansions, operand
 * build, MFMA + stage terms, stage solve, cost-to-go update, IPM directions,
 * rollout (rest), misc, rotation pre-pass, rollout gain / broadcast / step,
 * apply, wait for the last products; slot 15 = iterations.  Used by
 * tools/phase_profile.py. */
qmpc_status qmpc_debug_profile(qmpc_handle* h, int32_t batch, const qmpc_input* in, int64_t* cycles_out);

/* ---- introspection -------------------------------------------------------- */
const char* qmpc_status_string(int32_t status);
const char* qmpc_version(void);
int32_t     qmpc_sizeof_input(void);   /* ABI guards for foreign-language bindings */
int32_t     qmpc_sizeof_params(void);
int32_t     qmpc_sizeof_info(void);
int32_t     qmpc_sizeof_convex_input(void);
int32_t     qmpc_sizeof_input8(void);

#ifdef __cplusplus
}
#endif
#endif /* QMPC_H_ */
