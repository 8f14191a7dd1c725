// colmap/estimators/cost_functions.h (un-vendored COLMAP @ b6b7b54e): the three reprojection cost functors that
// glomap/estimators/bundle_adjustment.cc instantiates through CreateCameraCostFunction, restated from their published
// definitions (values for the recording Ceres; values + dual-number Jacobians for the solving one):
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
  // values in doubles; with the SOLVING Ceres stand-in (ref_shim_solve/ceres/ceres.h defines CERES_SHIM_SOLVING) also the
  // Jacobians, as dual-number derivatives of the same templated evaluation
  bool Evaluate(double const* const* p, double* r, double** jac) const override {
#ifdef CERES_SHIM_SOLVING
    if (jac != nullptr) {
      constexpr int kMax = 32;  // 4 + 3 + 4 + 3 + 3 + 12 at most
      using J = ceres::Jet<kMax>;
      J x[kMax];
      const J* px[6];
      int off = 0;
      for (size_t b = 0; b < sizes_.size(); ++b) {
        px[b] = x + off;
        for (int i = 0; i < sizes_[b]; ++i) x[off + i] = J(p[b][i], off + i);
        off += sizes_[b];
      }
      J rj[2];
      if (!Eval<J>(px, rj)) return false;
      r[0] = rj[0].a;
      r[1] = rj[1].a;
      off = 0;
      for (size_t b = 0; b < sizes_.size(); ++b) {
        if (jac[b] != nullptr)
          for (int k = 0; k < 2; ++k)
            for (int i = 0; i < sizes_[b]; ++i) jac[b][k * sizes_[b] + i] = rj[k].v[off + i];
        off += sizes_[b];
      }
      return true;
    }
#endif
    return Eval<double>(p, r);
  }

  // R(q) v for a quaternion block in Eigen coefficient order (x, y, z, w), as Eigen's operator*: v + 2 w (u x v) + 2 u x (u x v)
  template <typename T>
  static void Rotate(const T* q, const T v[3], T out[3]) {
    const T uv0 = (q[1] * v[2] - q[2] * v[1]) * 2.0, uv1 = (q[2] * v[0] - q[0] * v[2]) * 2.0, uv2 = (q[0] * v[1] - q[1] * v[0]) * 2.0;
    out[0] = v[0] + q[3] * uv0 + (q[1] * uv2 - q[2] * uv1);
    out[1] = v[1] + q[3] * uv1 + (q[2] * uv0 - q[0] * uv2);
    out[2] = v[2] + q[3] * uv2 + (q[0] * uv1 - q[1] * uv0);
  }

  template <typename T>
  bool Eval(T const* const* p, T* r) const {
    const int o = kind_ == 2 ? 2 : 0;
    T X[3] = {p[o + 2][0], p[o + 2][1], p[o + 2][2]}, x[3];
    Rotate<T>(p[o], X, x);
    for (int j = 0; j < 3; ++j) x[j] = x[j] + p[o + 1][j];
    if (kind_ == 1) {
      const T qc[4] = {T(cam_from_rig_.rotation.x()), T(cam_from_rig_.rotation.y()), T(cam_from_rig_.rotation.z()), T(cam_from_rig_.rotation.w())};
      T y[3];
      Rotate<T>(qc, x, y);
      for (int j = 0; j < 3; ++j) x[j] = y[j] + cam_from_rig_.translation(j);
    }
    if (kind_ == 2) {
      T y[3];
      Rotate<T>(p[0], x, y);
      for (int j = 0; j < 3; ++j) x[j] = y[j] + p[1][j];
    }
    const T* k = p[o + 3];
    r[0] = T(0.0);
    r[1] = T(0.0);
    if (x[2] < std::numeric_limits<double>::epsilon()) return true;
    const T u = x[0] / x[2], v = x[1] / x[2], r2 = u * u + v * v;
    switch (model_) {
      case CameraModelId::kSimplePinhole: r[0] = k[0] * u + k[1]; r[1] = k[0] * v + k[2]; break;
      case CameraModelId::kPinhole: r[0] = k[0] * u + k[2]; r[1] = k[1] * v + k[3]; break;
      case CameraModelId::kSimpleRadial: { const T d = k[3] * r2; r[0] = k[0] * (u + u * d) + k[1]; r[1] = k[0] * (v + v * d) + k[2]; break; }
      case CameraModelId::kRadial: { const T d = k[3] * r2 + k[4] * r2 * r2; r[0] = k[0] * (u + u * d) + k[1]; r[1] = k[0] * (v + v * d) + k[2]; break; }
      case CameraModelId::kOpenCV: {
        const T rad = k[4] * r2 + k[5] * r2 * r2, uv = u * v;
        const T du = u * rad + k[6] * uv * 2.0 + k[7] * (r2 + u * u * 2.0), dv = v * rad + k[7] * uv * 2.0 + k[6] * (r2 + v * v * 2.0);
        r[0] = k[0] * (u + du) + k[2]; r[1] = k[1] * (v + dv) + k[3]; break;
      }
      case CameraModelId::kFullOpenCV: {  // fx, fy, cx, cy, k1, k2, p1, p2, k3, k4, k5, k6
        const T r4 = r2 * r2, r6 = r4 * r2, uv = u * v;
        const T rad = (k[4] * r2 + k[5] * r4 + k[8] * r6 + 1.0) / (k[9] * r2 + k[10] * r4 + k[11] * r6 + 1.0);
        const T ud = u * rad + k[6] * uv * 2.0 + k[7] * (r2 + u * u * 2.0), vd = v * rad + k[7] * uv * 2.0 + k[6] * (r2 + v * v * 2.0);
        r[0] = k[0] * ud + k[2]; r[1] = k[1] * vd + k[3]; break;
      }
      default: return false;
    }
    r[0] = r[0] - xy_(0);
    r[1] = r[1] - xy_(1);
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
