#pragma once
// Stand-in: glomap/estimators/relpose_estimation.h holds two PoseLib option structs by value (the relative-pose stage itself is
// skipped in the mapper runs of oracle/_ref: out of scope).
namespace poselib {
struct RansacOptions { int max_iterations = 100000; };
struct BundleOptions {};
}  // namespace poselib
