// colmap/estimators/cost_functions.h (un-vendored COLMAP @ b6b7b54e): the three reprojection cost functors that
// glomap/estimators/bundle_adjustment.cc instantiates through CreateCameraCostFunction, restated from their published
// definitions for the recording Ceres (values only, no Jacobians):
//   ReprojErrorCostFunctor                (q, t, X, params):                 x_c = R(q) X + t
//   RigReprojErrorConstantRigCostFunctor  (q, t, X, params), cam_from_rig:   x_c = cam_from_rig * (R(q) X + t)
//   RigReprojErrorCostFunctor             (q_s, t_s, q, t, X, params):       x_c = R(q_s) (R(q) X + t) + t_s
//   r = CameraModel::ImgFromCam(params, x_c) - point2D, zero when the point is not in front of the camera.
// Quaternion blocks are Eigen coefficient order (x, y, z, w).  Camera models: the five perspective ones up to OPENCV, and FULL_OPENCV
// (12 parameters: the block size and the SubsetManifold of a model with more than eight parameters come from the reference's code).
#pragma once
#include <cmath>
#include <limits>

#include <ceres/ceres.h>

#include "ref_shim_types.h"

namespace colmap {
template <typename CameraModel> struct ReprojErrorCostFunctor { static constexpr int kKind = 0; };
template <typename CameraModel> struct RigReprojErrorConstantRigCostFunctor { static constexpr int kKind = 1; };
template <typename CameraModel> struct RigReprojErrorCostFunctor { static constexpr int kKind = 2; };
struct RefShimAnyModel {};

inline int NumParams(CameraModelId id) {
  switch (id) {
    case CameraModelId::kSimplePinhole: return 3;
    case CameraModelId::kPinhole: return 4;
    case CameraModelId::kSimpleRadial: return 4;
    case CameraModelId::kRadial: return 5;
    case CameraModelId::kOpenCV: return 8;
    case CameraModelId::kFullOpenCV: return 12;
    default: return -1;
  }
}

class RefShimReprojCost final : public ceres::CostFunction {
 public:
  RefShimReprojCost(int kind, CameraModelId model, const Eigen::Vector2d& point2D, const glomap::Rigid3d& cam_from_rig)
      : kind_(kind), model_(model), xy_(point2D), cam_from_rig_(cam_from_rig) {
    num_residuals_ = 2;
    if (kind == 2) sizes_ = {4, 3, 4, 3, 3, NumParams(model)};
    else sizes_ = {4, 3, 3, NumParams(model)};
  }
  int kind() const { return kind_; }
  const Eigen::Vector2d& point2D() const { return xy_; }
  const glomap::Rigid3d& cam_from_rig() const { return cam_from_rig_; }
  bool Evaluate(double const* const* p, double* r, double**) const override {
    const int o = kind_ == 2 ? 2 : 0;
    const Eigen::Quaterniond q(p[o][3], p[o][0], p[o][1], p[o][2]);
    Eigen::Vector3d x = q * Eigen::Vector3d(p[o + 2][0], p[o + 2][1], p[o + 2][2]) + Eigen::Vector3d(p[o + 1][0], p[o + 1][1], p[o + 1][2]);
    if (kind_ == 1) x = cam_from_rig_.rotation * x + cam_from_rig_.translation;
    if (kind_ == 2) x = Eigen::Quaterniond(p[0][3], p[0][0], p[0][1], p[0][2]) * x + Eigen::Vector3d(p[1][0], p[1][1], p[1][2]);
    const double* k = p[o + 3];
    r[0] = r[1] = 0.0;
    if (x(2) < std::numeric_limits<double>::epsilon()) return true;
    const double u = x(0) / x(2), v = x(1) / x(2), r2 = u * u + v * v;
    switch (model_) {
      case CameraModelId::kSimplePinhole: r[0] = k[0] * u + k[1]; r[1] = k[0] * v + k[2]; break;
      case CameraModelId::kPinhole: r[0] = k[0] * u + k[2]; r[1] = k[1] * v + k[3]; break;
      case CameraModelId::kSimpleRadial: { const double d = k[3] * r2; r[0] = k[0] * (u + u * d) + k[1]; r[1] = k[0] * (v + v * d) + k[2]; break; }
      case CameraModelId::kRadial: { const double d = k[3] * r2 + k[4] * r2 * r2; r[0] = k[0] * (u + u * d) + k[1]; r[1] = k[0] * (v + v * d) + k[2]; break; }
      case CameraModelId::kOpenCV: {
        const double rad = k[4] * r2 + k[5] * r2 * r2, uv = u * v;
        const double du = u * rad + 2.0 * k[6] * uv + k[7] * (r2 + 2.0 * u * u), dv = v * rad + 2.0 * k[7] * uv + k[6] * (r2 + 2.0 * v * v);
        r[0] = k[0] * (u + du) + k[2]; r[1] = k[1] * (v + dv) + k[3]; break;
      }
      case CameraModelId::kFullOpenCV: {  // fx, fy, cx, cy, k1, k2, p1, p2, k3, k4, k5, k6
        const double r4 = r2 * r2, r6 = r4 * r2, uv = u * v;
        const double rad = (1.0 + k[4] * r2 + k[5] * r4 + k[8] * r6) / (1.0 + k[9] * r2 + k[10] * r4 + k[11] * r6);
        const double ud = u * rad + 2.0 * k[6] * uv + k[7] * (r2 + 2.0 * u * u), vd = v * rad + 2.0 * k[7] * uv + k[6] * (r2 + 2.0 * v * v);
        r[0] = k[0] * ud + k[2]; r[1] = k[1] * vd + k[3]; break;
      }
      default: return false;
    }
    r[0] -= xy_(0);
    r[1] -= xy_(1);
    return true;
  }

 private:
  int kind_;
  CameraModelId model_;
  Eigen::Vector2d xy_;
  glomap::Rigid3d cam_from_rig_;
};

template <template <typename> class CostFunctor>
ceres::CostFunction* CreateCameraCostFunction(CameraModelId model, const Eigen::Vector2d& point2D) {
  if (NumParams(model) < 0) return nullptr;
  return new RefShimReprojCost(CostFunctor<RefShimAnyModel>::kKind, model, point2D, glomap::Rigid3d());
}
template <template <typename> class CostFunctor>
ceres::CostFunction* CreateCameraCostFunction(CameraModelId model, const Eigen::Vector2d& point2D, const glomap::Rigid3d& cam_from_rig) {
  if (NumParams(model) < 0) return nullptr;
  return new RefShimReprojCost(CostFunctor<RefShimAnyModel>::kKind, model, point2D, cam_from_rig);
}
}  // namespace colmap
