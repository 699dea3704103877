// MovingWindowFilter.h -- Neumaier-compensated moving average, the smoother of
// the MPC references (legged_ctrl/include/utils/MovingWindowFilter.hpp:14-63,
// used at QuatMpc.cpp:10-11,87-89,103-105).  Same arithmetic, operation for
// operation: the compensated sum is divided by the WINDOW SIZE even while the
// window is still filling (MovingWindowFilter.hpp:38).
#pragma once

#include <cmath>
#include <cstddef>
#include <vector>

namespace legged {

class MovingWindowFilterHip {
 public:
  MovingWindowFilterHip() : MovingWindowFilterHip(1) {}
  explicit MovingWindowFilterHip(int window_size)
      : window_(window_size > 0 ? window_size : 1), ring_(static_cast<std::size_t>(window_), 0.0) {}

  double CalculateAverage(double v) {
    if (count_ == window_) {
      neumaier(-ring_[head_]);           // drop the oldest sample first
    } else {
      ++count_;
    }
    neumaier(v);
    ring_[head_] = v;
    head_ = (head_ + 1) % static_cast<std::size_t>(window_);
    return (sum_ + correction_) / static_cast<double>(window_);
  }

 private:
  void neumaier(double v) {
    const double ns = sum_ + v;
    if (std::abs(sum_) >= std::abs(v)) correction_ += (sum_ - ns) + v;
    else correction_ += (v - ns) + sum_;
    sum_ = ns;
  }
  int window_;
  std::vector<double> ring_;
  std::size_t head_ = 0;
  int count_ = 0;
  double sum_ = 0.0, correction_ = 0.0;
};

}  // namespace legged
