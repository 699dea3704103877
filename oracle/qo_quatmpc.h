/*
 * qo_quatmpc.h -- CPU restatement of the solve inside
 * legged::QuatMpc::grf_update (legged_ctrl/src/mpc/QuatMpc.cpp:109-276),
 * taking the same records as the C ABI in include/qmpc.h.
 * TEST INFRASTRUCTURE ONLY (checker + bench.py cpu_baseline leg).
 */
#ifndef QO_QUATMPC_H_
#define QO_QUATMPC_H_

#include "../include/qmpc.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Go1 values of legged_ctrl/config/gazebo_go1_quat_mpc.yaml + solver defaults */
void qo_default_params(qmpc_params* p, int32_t horizon, int32_t mode);

/* Reference trajectory of QuatMpc.cpp:118-125,148-176: xref (N+1)x13, uref 12 */
void qo_build_reference(const qmpc_params* p, const qmpc_input* in, double* xref, double* uref);

/* One instance.  forces: 12; traj_u (N x 12) / traj_x ((N+1) x 13) may be NULL. */
int qo_solve_one(const qmpc_params* p, const qmpc_input* in, double* forces, qmpc_info* info,
                 double* traj_u, double* traj_x, int verbose);

/* As qo_solve_one, plus the multipliers and slacks of the cone rows at the returned point, [N][24] each
 * (swing-leg rows report 0).  Used by tests/golden/make_kkt_fixtures.py only. */
/* started from the previous solution u_init [N][12] shifted by one knot (NULL: qo_solve_one); restates the product's
 * qmpc_solve_warm */
int qo_solve_one_warm(const qmpc_params* p, const qmpc_input* in, const double* u_init, double* forces, qmpc_info* info,
                      double* traj_u);
int qo_solve_one_dual(const qmpc_params* p, const qmpc_input* in, double* forces, qmpc_info* info,
                      double* traj_u, double* traj_x, double* dual, double* slack);
int qo_solve8_one_dual(const qmpc_params* p, const qmpc_input8* in, double* forces, qmpc_info* info,
                       double* traj_u, double* traj_x, double* dual, double* slack);

/* Batch, `threads` >= 1 host threads over instances (instance-parallel). */
int qo_solve_batch(const qmpc_params* p, int32_t batch, const qmpc_input* in, double* forces,
                   qmpc_info* info, double* traj_u, double* traj_x, int32_t threads);

/* 8 contact points (BASELINE config 5, synthetic biped): forces 24 per instance */
void qo_default_biped8_params(qmpc_params* p, int32_t horizon, int32_t mode);
int qo_solve8_one(const qmpc_params* p, const qmpc_input8* in, double* forces, qmpc_info* info,
                  double* traj_u, double* traj_x, int verbose);
int qo_solve8_batch(const qmpc_params* p, int32_t batch, const qmpc_input8* in, double* forces,
                    qmpc_info* info, double* traj_u, double* traj_x, int32_t threads);

/* Linearisation only: rollout of U = u_ref from x0 and the projected
 * Jacobians (mirrors qmpc_linearize). */
int qo_linearize(const qmpc_params* p, int32_t batch, const qmpc_input* in, double* Abar,
                 double* Bbar, double* X);

#ifdef __cplusplus
}
#endif
#endif
