// Second unit of the lane-per-instance kernels: the reference mode's kernel (qmpc_lane_ref_kernel: the AL-iLQR scheme of
// legged_ctrl/src/mpc/QuatMpc.cpp:21-26 on the lane passes) with its launcher and its own parameter table.  The source is
// qmpc_lane.hip; the unit exists because the two kernels want different instruction-scheduling strategies (see the top of that
// file and __graft_entry__.py).
#define QL_UNIT 2
#include "qmpc_lane.hip"
