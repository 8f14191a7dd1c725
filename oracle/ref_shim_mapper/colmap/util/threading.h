#pragma once
// Stand-in for colmap::ThreadPool as glomap/processors/image_undistorter.cc uses it: tasks run inline, in submission order.
#include <functional>
namespace colmap {
class ThreadPool {
 public:
  static constexpr int kMaxNumThreads = -1;
  explicit ThreadPool(int = kMaxNumThreads) {}
  template <typename F> void AddTask(F&& f) { f(); }
  void Wait() {}
};
}  // namespace colmap
