// colmap/geometry/pose.h + sim3.h are part of the un-vendored COLMAP dependency: the two things
// reconstruction_normalizer.cc uses, restated from their published definitions.
//   Sim3d(scale, rotation, translation):  x_new = scale * (rotation * x_old) + translation
//   TransformCameraWorld(new_from_old_world, cam_from_world) = cam_from_new_world: the same camera expressed against the
//   transformed world — rotation' = R_cam R_sim^T, translation' = scale * t_cam - rotation' * t_sim.
#pragma once
#include "ref_shim_types.h"

namespace colmap {
struct Sim3d {
  double scale = 1.0;
  Eigen::Quaterniond rotation;
  Eigen::Vector3d translation;
  Sim3d() = default;
  Sim3d(double s, const Eigen::Quaterniond& r, const Eigen::Vector3d& t) : scale(s), rotation(r), translation(t) {}
};
inline Eigen::Vector3d operator*(const Sim3d& t, const Eigen::Vector3d& x) { return t.scale * (t.rotation * x) + t.translation; }
}  // namespace colmap

namespace glomap {
inline Rigid3d TransformCameraWorld(const colmap::Sim3d& new_from_old_world, const Rigid3d& cam_from_world) {
  // the normaliser only ever passes the identity rotation (reconstruction_normalizer.cc:56-57): R' = R_cam
  Rigid3d out;
  out.rotation = cam_from_world.rotation;
  out.translation = new_from_old_world.scale * cam_from_world.translation - (out.rotation * new_from_old_world.translation);
  return out;
}
}  // namespace glomap
