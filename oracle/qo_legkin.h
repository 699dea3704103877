/*
 * qo_legkin.h -- CPU restatement of the leg kinematics and of the force -> joint
 * torque map that consumes the MPC forces:
 *   A1Kinematics::fk / ::jac   legged_ctrl/src/utils/A1Kinematics.cpp:9-19 (+ the
 *                              closed forms they evaluate, :38-128)
 *   BaseInterface::tau_ctrl_update  legged_ctrl/src/interfaces/BaseInterface.cpp:343-408
 * TEST INFRASTRUCTURE ONLY.  Pinned by the stand-pose foothold the reference's own
 * Gazebo interface starts from (q = (0, 0.67, -1.3) -> (0.1813, 0.12795, -0.339),
 * SURVEY.md 8d) and by central differences of the forward kinematics.
 */
#ifndef QO_LEGKIN_H_
#define QO_LEGKIN_H_

#include "../include/qmpc.h"

#ifdef __cplusplus
extern "C" {
#endif

void qo_default_go1_geometry(qmpc_leg_geometry* g);
/* p(3) body frame; q = (hip, thigh, calf) */
void qo_leg_fk(const double q[3], const double rho_opt[3], const double rho_fix[5], double p[3]);
/* J(3x3) COLUMN-major: J[3*j + i] = d p_i / d q_j */
void qo_leg_jac(const double q[3], const double rho_opt[3], const double rho_fix[5], double J[9]);
void qo_leg_kinematics(const qmpc_leg_geometry* g, int32_t batch, const double* joint_pos,
                       double* foot_pos_body, double* jac);
void qo_torque_map(const qmpc_leg_geometry* g, int32_t batch, const double* joint_pos,
                   const double* forces_body, const double* contacts, int32_t walking, double* tau);

/* A1Kinematics::inv_kin (A1Kinematics.cpp:335-459) with its single-precision atan2 approximation (:291-312).
 * q = (hip, thigh, calf); cur_q selects the hip branch.  NaN when the foot is out of reach. */
void qo_leg_inv_kin(const double p[3], const double cur_q[3], const double rho_fix[5], double q[3]);
void qo_leg_inverse_kinematics(const qmpc_leg_geometry* g, int32_t batch, const double* foot_pos_body,
                               const double* cur_joint_pos, double* joint_pos);
/* BaseInterface::tau_ctrl_update (BaseInterface.cpp:343-408) */
void qo_joint_commands(const qmpc_leg_geometry* g, int32_t batch, const qmpc_joint_feedback* fb,
                       qmpc_joint_command* cmd);

#ifdef __cplusplus
}
#endif
#endif
