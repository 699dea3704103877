/*
 * qo_convex.h -- CPU restatement of the solve inside
 * legged::ConvexMpc::grf_update (legged_ctrl/src/mpc/ConvexMpc.cpp:81-198) and of
 * its Euler-angle single-rigid-body model (legged_ctrl/src/utils/AltroUtils.cpp:
 * 224-359), taking the records of include/qmpc.h.  TEST INFRASTRUCTURE ONLY.
 *
 * PARITY UNPINNED: the reference's only artefact for this controller,
 * src/test/test_altro/convex_mpc.json, was produced by a test that is not built
 * (CMakeLists.txt:200), with forward Euler, mass 13 and one solver iteration, and
 * is not reproducible from the current model.  The model
 * below is pinned only by its own consistency checks (tests/test_oracle_convex.py).
 */
#ifndef QO_CONVEX_H_
#define QO_CONVEX_H_

#include "../include/qmpc.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct qo_convex_model {
  double foot_pos[12];      /* foot_pos_abs_com, 3x4 col-major                    */
  double inertia_diag[3];   /* body-frame trunk inertia (diagonal, AltroUtils.cpp:270-272) */
  double mass;
  double contacts[4];       /* 1 = stance.  The reference's model takes all four forces
                               (AltroUtils.cpp:284-288) and relies on the cone rows
                               0 <= fz <= fz_max*contact, |fxy| <= mu fz to drive swing-leg
                               forces to zero; here they are pinned to zero exactly (their
                               columns of B vanish), as in the quaternion path            */
} qo_convex_model;

/* QuadrupedModel::ct_srb_dynamics (AltroUtils.cpp:224-293), x(12), u(12) */
void qo_ct_srb_dynamics(const qo_convex_model* m, double* x_dot, const double* x, const double* u);
/* QuadrupedModel::ct_srb_jacobian (AltroUtils.cpp:295-359): 12 x 24 COLUMN-major.
 * As in the reference, d(I_world^-1)/d(yaw) is NOT part of it. */
void qo_ct_srb_jacobian(const qo_convex_model* m, double* jac, const double* x, const double* u);
/* explicit midpoint of the above with float h (ConvexMpc.cpp:121-122) */
void qo_convex_discrete_dynamics(const qo_convex_model* m, double* xn, const double* x,
                                 const double* u, float h);
void qo_convex_discrete_jacobian(const qo_convex_model* m, double* jac /* 12x24 col-major */,
                                 const double* x, const double* u, float h);

/* gazebo_go1_convex_mpc.yaml:35-73 + AltroUtils.cpp:239,270-272 + solver defaults */
void qo_default_convex_params(qmpc_params* p, int32_t horizon, int32_t mode);
/* As qo_convex_solve_one, plus multipliers and slacks of the cone rows [N][24] (tests/golden/make_kkt_fixtures.py) */
int qo_convex_solve_one_dual(const qmpc_params* p, const qmpc_convex_input* in, double* forces, qmpc_info* info,
                             double* traj_u, double* traj_x, double* dual, double* slack);
/* reference trajectory (ConvexMpc.cpp:94-113): xref (N+1) x 12, uref 12 */
void qo_convex_build_reference(const qmpc_params* p, const qmpc_convex_input* in, double* xref,
                               double* uref);

int qo_convex_solve_one(const qmpc_params* p, const qmpc_convex_input* in, double* forces,
                        qmpc_info* info, double* traj_u, double* traj_x, int verbose);
int qo_convex_solve_batch(const qmpc_params* p, int32_t batch, const qmpc_convex_input* in,
                          double* forces, qmpc_info* info, double* traj_u, double* traj_x,
                          int32_t threads);
/* one discrete step x -> xn with the record's footholds (test hook) */
void qo_convex_step(const qmpc_params* p, const qmpc_convex_input* in, const double* x,
                    const double* u, double* xn);
int qo_convex_linearize(const qmpc_params* p, int32_t batch, const qmpc_convex_input* in,
                        double* A, double* B, double* X);

#ifdef __cplusplus
}
#endif
#endif
