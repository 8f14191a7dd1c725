// colmap/estimators/manifold.h (un-vendored COLMAP): SetQuaternionManifold puts Ceres' EigenQuaternionManifold on a block,
// SetSubsetManifold a SubsetManifold that holds the listed coordinates constant.  Recorded, not applied (the Ceres of
// ref_shim/ceres/ceres.h does not minimise).
#pragma once
#include <vector>

#include <ceres/ceres.h>

namespace colmap {
inline void SetQuaternionManifold(ceres::Problem* problem, double* quat_xyzw) { problem->RecordManifold(quat_xyzw, 0, {}); }
inline void SetSubsetManifold(int /*size*/, const std::vector<int>& constant_params, ceres::Problem* problem, double* params) {
  problem->RecordManifold(params, 1, constant_params);
}
}  // namespace colmap
