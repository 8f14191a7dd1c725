#pragma once
// Stand-in for colmap::Timer as global_mapper.cc uses it (Start / PrintSeconds around every stage).
#include <chrono>
namespace colmap {
class Timer {
 public:
  void Start() { t0_ = std::chrono::steady_clock::now(); }
  double ElapsedSeconds() const { return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0_).count(); }
  void PrintSeconds() const {}  // (quiet: the tests read results, not logs)
 private:
  std::chrono::steady_clock::time_point t0_;
};
}  // namespace colmap
