// LeggedContactFSMHip.h -- the per-leg gait state machine that produces the
// contact schedule plan_contacts[4] (legged_ctrl/src/utils/LeggedContactFSM.cpp;
// SURVEY.md 8.a3).  Host-side, double-precision compare/add + enum: it must be
// BIT-EXACT, so the phase arithmetic follows the reference operation for
// operation:
//   update()            LeggedContactFSM.cpp:33-78   (gait_phase += gait_freq*dt;
//                       STANCE->SWING at phase >= end; SWING->STANCE at
//                       percent > 0.9 && contact flag, or percent >= 1.0)
//   common_enter()      :208-223                     (phase wrap "-= 1.0" when the
//                       pattern index wraps)
//   percent_in_state()  :261-270
//   reset()/reset_params()/set_default_gait_pattern()  :4-31,87-108
// The gait is a constexpr TABLE here (slots of {state, phase at which the slot ends}; the diagonal legs start in
// stance, the others in swing), walked with a slot index: same compare / add sequence on the same doubles as the
// reference's vectors, none of its text.
// The swing-foot side (not on the force path) is the second update() overload:
//   swing_enter / stance_enter / stance_exit   :80-86,225-235
//   swing_update (quintic foot target)         :237-246
// It leaves the schedule arithmetic untouched.
#pragma once

#include "SwingTrajectoryHip.h"

namespace legged {

enum LeggedContactStateHip { SWING_HIP = 0, STANCE_HIP = 1 };   // LeggedContactFSM.h:12-15

class LeggedContactFSMHip {
 public:
  void reset_params(double gait_freq_, int leg_id_) {   // :4-9
    leg_id = leg_id_;
    gait_freq = gait_freq_;
    s = STANCE_HIP;
    set_default_gait_pattern();
  }

  void set_default_gait_pattern() {                     // :87-108 (trot)
    table = (leg_id == 0 || leg_id == 3) ? kTrotDiagonal : kTrotOther;
    slot = 0;
    prev_slot = kSlots - 1;
    slot_begin = 0.0;
    slot_end = table[slot].ends_at;
  }

  void reset() {                                        // :11-31
    gait_phase = 0;
    slot = 0;
    prev_slot = kSlots - 1;
    slot_begin = 0;
    slot_end = table[slot].ends_at;
    if (s == SWING_HIP) {                               // :20-25 a swing foot goes to its saved target
      for (int a = 0; a < 3; ++a) { FSM_foot_pos_target_world[a] = swing_end_foot_pos_world[a]; FSM_foot_vel_target_world[a] = 0.0; }
    }
    s = table[slot].state;
    not_first_call = false;                             // :30 the next update() re-seeds the targets
  }

  // foot_force_flag is the reference's (bool)foot_contact_flag[i]: ANY non-zero
  // value is true (QuatMpc.cpp:295, BaseInterface.cpp:241)
  double update(double dt, double gait_freq_now, bool foot_force_flag) {   // :33-78
    gait_phase += gait_freq_now * dt;
    if (s == STANCE_HIP) {
      if (gait_phase >= slot_end) {
        common_enter();
        s = SWING_HIP;
      }
    } else if (s == SWING_HIP) {
      if (percent_in_state() > 0.9 && foot_force_flag) {
        s = STANCE_HIP;
        common_enter();
      } else if (percent_in_state() >= 1.0) {
        s = STANCE_HIP;
        common_enter();
      }
    }
    return gait_phase;
  }

  // Full update (:33-78): the schedule step above plus the foot targets of the state entered / held.
  double update(double dt, double gait_freq_now, const double foot_pos_cur_world[3],
                const double foot_pos_target_world[3], bool foot_force_flag) {
    if (not_first_call == false) {                      // :37-43
      for (int a = 0; a < 3; ++a) {
        swing_start_foot_pos_world[a] = foot_pos_cur_world[a];
        swing_end_foot_pos_world[a] = foot_pos_target_world[a];
        FSM_foot_pos_target_world[a] = foot_pos_target_world[a];
        FSM_foot_vel_target_world[a] = 0.0;
      }
      not_first_call = true;
    }
    gait_phase += gait_freq_now * dt;
    if (s == STANCE_HIP) {
      if (gait_phase >= slot_end) {
        terrain_height = foot_pos_cur_world[2];         // stance_exit, :80-84
        common_enter();                                 // swing_enter, :225-229
        for (int a = 0; a < 3; ++a) { swing_start_foot_pos_world[a] = foot_pos_cur_world[a]; swing_extend_foot_pos_world[a] = 0.0; }
        s = SWING_HIP;
      }
    } else if (s == SWING_HIP) {
      if ((percent_in_state() > 0.9 && foot_force_flag) || percent_in_state() >= 1.0) {
        s = STANCE_HIP;
        common_enter();                                 // stance_enter, :231-235
        for (int a = 0; a < 3; ++a) { FSM_foot_pos_target_world[a] = foot_pos_cur_world[a]; FSM_foot_vel_target_world[a] = 0.0; }
      }
    }
    if (s == SWING_HIP) {                               // swing_update, :237-246 (stance_update is empty upstream)
      const double t = percent_in_state();
      double fin[3], out[9];
      for (int a = 0; a < 3; ++a) fin[a] = foot_pos_target_world[a] + swing_extend_foot_pos_world[a];
      quintic_curve.get_foot_swing_target(static_cast<float>(0.5 * t / gait_freq_now),
                                          static_cast<float>(0.5 / gait_freq_now), swing_start_foot_pos_world, fin, out);
      for (int a = 0; a < 3; ++a) {
        FSM_foot_pos_target_world[a] = out[a];
        FSM_foot_vel_target_world[a] = out[3 + a];
        FSM_foot_acc_target_world[a] = out[6 + a];
      }
    }
    return gait_phase;
  }

  LeggedContactStateHip get_contact_state() const { return s; }
  double FSM_foot_pos_target_world[3] = {0, 0, 0};      // LeggedContactFSM.h:67-69
  double FSM_foot_vel_target_world[3] = {0, 0, 0};
  double FSM_foot_acc_target_world[3] = {0, 0, 0};
  double terrain_height = 0.0;
  double phase() const { return gait_phase; }
  // read-only views of the schedule internals (closed-loop parity checks, host/ClosedLoopHost.h)
  int pattern_index() const { return slot; }
  int prev_pattern_index() const { return prev_slot; }
  double state_start_time() const { return slot_begin; }
  double state_end_time() const { return slot_end; }
  bool first_call_done() const { return not_first_call; }
  const double* swing_start() const { return swing_start_foot_pos_world; }
  const double* swing_end() const { return swing_end_foot_pos_world; }

 private:
  // next slot of the table; the phase loses one whole period when the table wraps (:208-223)
  void common_enter() {
    prev_slot = slot;
    slot = (slot + 1) % kSlots;
    if (slot < prev_slot) gait_phase -= 1.0;
    slot_begin = gait_phase;
    slot_end = table[slot].ends_at;
  }
  // fraction of the current slot that has elapsed, clamped to [0, 1] (:261-270)
  double percent_in_state() const {
    const double f = (gait_phase - slot_begin) / (slot_end - slot_begin);
    return f < 0.0 ? 0.0 : (f > 1.0 ? 1.0 : f);
  }

  struct GaitSlot { LeggedContactStateHip state; double ends_at; };
  static constexpr int kSlots = 2;
  static constexpr GaitSlot kTrotDiagonal[kSlots] = {{STANCE_HIP, 0.5}, {SWING_HIP, 1.0}};
  static constexpr GaitSlot kTrotOther[kSlots] = {{SWING_HIP, 0.5}, {STANCE_HIP, 1.0}};

  bool not_first_call = false;                          // LeggedContactFSM.h:96-99,105
  double swing_start_foot_pos_world[3] = {0, 0, 0};
  double swing_end_foot_pos_world[3] = {0, 0, 0};
  double swing_extend_foot_pos_world[3] = {0, 0, 0};    // never assigned upstream before the first swing_enter
  QuinticCurveHip quintic_curve;
  int leg_id = 0;
  LeggedContactStateHip s = STANCE_HIP;
  // The reference first assigns gait_phase in reset() (:12), i.e. it is
  // indeterminate if update() runs before any stand-mode tick; we start at 0.
  double gait_phase = 0.0;
  double gait_freq = 0.0;
  const GaitSlot* table = kTrotDiagonal;
  int slot = 0, prev_slot = kSlots - 1;
  double slot_begin = 0.0, slot_end = 0.5;
};

}  // namespace legged
