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
// The swing-foot quintic trajectory (swing_update, :237-246) is NOT on the force
// path and is out of scope; the foot targets it would fill
// are left untouched.
#pragma once

#include <vector>

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
    gait_state_pattern.clear();
    gait_switch_time.clear();
    if (leg_id == 0 || leg_id == 3) {
      gait_state_pattern.push_back(STANCE_HIP);
      gait_state_pattern.push_back(SWING_HIP);
    } else {
      gait_state_pattern.push_back(SWING_HIP);
      gait_state_pattern.push_back(STANCE_HIP);
    }
    gait_switch_time.push_back(0.5);
    gait_switch_time.push_back(1.0);
    gait_pattern_size = 2;
    gait_pattern_index = 0;
    prev_gait_pattern_index = gait_pattern_size - 1;
    cur_state_start_time = 0.0;
    cur_state_end_time = gait_switch_time[gait_pattern_index];
  }

  void reset() {                                        // :11-31
    gait_phase = 0;
    gait_pattern_index = 0;
    prev_gait_pattern_index = gait_pattern_size - 1;
    cur_state_start_time = 0;
    cur_state_end_time = gait_switch_time[gait_pattern_index];
    s = gait_state_pattern[gait_pattern_index];
  }

  // foot_force_flag is the reference's (bool)foot_contact_flag[i]: ANY non-zero
  // value is true (QuatMpc.cpp:295, BaseInterface.cpp:241)
  double update(double dt, double gait_freq_now, bool foot_force_flag) {   // :33-78
    gait_phase += gait_freq_now * dt;
    if (s == STANCE_HIP) {
      if (gait_phase >= cur_state_end_time) {
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

  LeggedContactStateHip get_contact_state() const { return s; }
  double phase() const { return gait_phase; }

 private:
  void common_enter() {                                 // :208-223
    prev_gait_pattern_index = gait_pattern_index;
    gait_pattern_index = (gait_pattern_index + 1) % gait_pattern_size;
    if (gait_pattern_index < prev_gait_pattern_index) gait_phase -= 1.0;
    cur_state_start_time = gait_phase;
    cur_state_end_time = gait_switch_time[gait_pattern_index];
  }
  double percent_in_state() const {                     // :261-270
    double percent = (gait_phase - cur_state_start_time) / (cur_state_end_time - cur_state_start_time);
    if (percent < 0.0) percent = 0.0;
    else if (percent > 1.0) percent = 1.0;
    return percent;
  }

  int leg_id = 0;
  LeggedContactStateHip s = STANCE_HIP;
  // The reference first assigns gait_phase in reset() (:12), i.e. it is
  // indeterminate if update() runs before any stand-mode tick; we start at 0.
  double gait_phase = 0.0;
  double gait_freq = 0.0;
  std::vector<LeggedContactStateHip> gait_state_pattern;
  std::vector<double> gait_switch_time;
  int gait_pattern_size = 0;
  int gait_pattern_index = 0;
  int prev_gait_pattern_index = 0;
  double cur_state_start_time = 0.0;
  double cur_state_end_time = 0.0;
};

}  // namespace legged
