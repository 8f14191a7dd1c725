// Interface-shaped stand-ins of the GLOMAP / COLMAP / Eigen types that include/gsfm_glomap_adapter.hpp
// touches (member names, method names and call shapes as in /root/reference/glomap/scene/*.h and the
// colmap / Eigen types they expose).  They exist only so that the adapter can be compiled and run in
// a repository that does not vendor GLOMAP: nothing here is reference code.
#pragma once
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <map>
#include <optional>
#include <string>
#include <unordered_map>
#include <utility>
#include <vector>

namespace mock_eigen {
struct Vector2d {
  double v[2] = {0, 0};
  Vector2d() = default;
  Vector2d(double a, double b) : v{a, b} {}
  double& operator[](int i) { return v[i]; }
  const double& operator[](int i) const { return v[i]; }
};
struct Vector3d {
  double v[3] = {0, 0, 0};
  Vector3d() = default;
  Vector3d(double a, double b, double c) : v{a, b, c} {}
  double& operator[](int i) { return v[i]; }
  const double& operator[](int i) const { return v[i]; }
};
struct MatrixXi {  // Eigen::MatrixXi shape used by ImagePair::matches: (row, col) access
  std::vector<int> d;
  int cols_ = 2;
  int rows() const { return static_cast<int>(d.size()) / cols_; }
  int operator()(int r, int c) const { return d[static_cast<size_t>(r) * cols_ + c]; }
  void push_row(int a, int b) {
    d.push_back(a);
    d.push_back(b);
  }
};
struct Matrix3d {  // (row, col) access like Eigen::Matrix3d; identity by default
  double m[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
  double& operator()(int r, int c) { return m[3 * r + c]; }
  const double& operator()(int r, int c) const { return m[3 * r + c]; }
};
struct Quaterniond {  // Eigen's constructor order: (w, x, y, z)
  double w_ = 1, x_ = 0, y_ = 0, z_ = 0;
  Quaterniond() = default;
  Quaterniond(double w, double x, double y, double z) : w_(w), x_(x), y_(y), z_(z) {}
  double w() const { return w_; }
  double x() const { return x_; }
  double y() const { return y_; }
  double z() const { return z_; }
};
}  // namespace mock_eigen

namespace glomap {
using camera_t = uint32_t;
using image_t = uint32_t;
using frame_t = uint32_t;
using rig_t = uint32_t;
using image_pair_t = uint64_t;
using feature_t = uint32_t;
using track_t = uint64_t;

struct Rigid3d {
  mock_eigen::Quaterniond rotation;
  mock_eigen::Vector3d translation;
};

// colmap::sensor_t / data_t / Rig (colmap/sensor/rig.h, colmap/scene/frame.h): the members the estimators touch
enum class SensorType { INVALID = -1, CAMERA = 0, IMU = 1 };
struct sensor_t {
  SensorType type = SensorType::INVALID;
  uint32_t id = 0xffffffffu;
  sensor_t() = default;
  sensor_t(SensorType t, uint32_t i) : type(t), id(i) {}
  bool operator==(const sensor_t& o) const { return type == o.type && id == o.id; }
  bool operator<(const sensor_t& o) const { return type != o.type ? type < o.type : id < o.id; }
};
struct data_t {
  sensor_t sensor_id;
  uint32_t id = 0xffffffffu;
  data_t() = default;
  data_t(sensor_t s, uint32_t i) : sensor_id(s), id(i) {}
};
}  // namespace glomap
namespace std {
template <>
struct hash<glomap::sensor_t> {
  size_t operator()(const glomap::sensor_t& s) const { return (static_cast<size_t>(s.type) << 32) ^ s.id; }
};
}  // namespace std
namespace glomap {

class Rig {
 public:
  rig_t RigId() const { return rig_id_; }
  void SetRigId(rig_t id) { rig_id_ = id; }
  void AddRefSensor(sensor_t s) { ref_ = s; }
  void AddSensor(sensor_t s, const std::optional<Rigid3d>& sensor_from_rig = std::nullopt) { sensors_[s] = sensor_from_rig; }
  sensor_t RefSensorId() const { return ref_; }
  bool IsRefSensor(sensor_t s) const { return s == ref_; }
  std::map<sensor_t, std::optional<Rigid3d>>& NonRefSensors() { return sensors_; }
  const std::map<sensor_t, std::optional<Rigid3d>>& NonRefSensors() const { return sensors_; }
  std::optional<Rigid3d> MaybeSensorFromRig(sensor_t s) const {
    auto it = sensors_.find(s);
    return it == sensors_.end() ? std::nullopt : it->second;
  }
  Rigid3d& SensorFromRig(sensor_t s) {
    auto& o = sensors_.at(s);
    if (!o.has_value()) std::abort();  // colmap THROW_CHECKs
    return *o;
  }
  const Rigid3d& SensorFromRig(sensor_t s) const {
    const auto& o = sensors_.at(s);
    if (!o.has_value()) std::abort();
    return *o;
  }
  void SetSensorFromRig(sensor_t s, const Rigid3d& v) { sensors_[s] = v; }
  void ResetSensorFromRig(sensor_t s) { sensors_[s] = std::nullopt; }

 private:
  rig_t rig_id_ = 0xffffffffu;
  sensor_t ref_;
  std::map<sensor_t, std::optional<Rigid3d>> sensors_;
};
struct Camera {
  int model_id = 0;
  std::vector<double> params;
  bool has_prior_focal_length = true;
};
// glomap::GravityInfo (scene/frame.h:11-27): R_align's second column is the gravity direction (math/gravity.cc:10-24);
// the stand-in completes the basis with a cross product instead of Eigen's Householder QR (any right-handed completion
// gives a valid alignment)
struct GravityInfo {
  bool has_gravity = false;
  const mock_eigen::Matrix3d& GetRAlign() const { return R_align_; }
  mock_eigen::Vector3d GetGravity() const { return gravity_in_rig_; }
  void SetGravity(const mock_eigen::Vector3d& g) {
    gravity_in_rig_ = g;
    const double n = std::sqrt(g[0] * g[0] + g[1] * g[1] + g[2] * g[2]);
    const double v[3] = {g[0] / n, g[1] / n, g[2] / n};
    double a[3] = {1, 0, 0};
    if (std::fabs(v[0]) > 0.9) { a[0] = 0; a[2] = 1; }
    double x[3] = {a[1] * v[2] - a[2] * v[1], a[2] * v[0] - a[0] * v[2], a[0] * v[1] - a[1] * v[0]};  // a x v
    const double xn = std::sqrt(x[0] * x[0] + x[1] * x[1] + x[2] * x[2]);
    for (double& c : x) c /= xn;
    const double z[3] = {x[1] * v[2] - x[2] * v[1], x[2] * v[0] - x[0] * v[2], x[0] * v[1] - x[1] * v[0]};  // x x v
    for (int r = 0; r < 3; ++r) {
      R_align_(r, 0) = x[r];
      R_align_(r, 1) = v[r];
      R_align_(r, 2) = z[r];
    }
    has_gravity = true;
  }

 private:
  mock_eigen::Vector3d gravity_in_rig_;
  mock_eigen::Matrix3d R_align_;
};
struct Frame {
  GravityInfo gravity_info;
  bool HasGravity() const { return gravity_info.has_gravity; }
  bool is_registered = false;
  bool has_pose = false;
  Rigid3d pose;
  bool HasPose() const { return has_pose; }
  // colmap::Frame::RigFromWorld() THROW_CHECKs that the pose is set; the stand-in aborts so that tests notice
  Rigid3d& RigFromWorld() {
    if (!has_pose) std::abort();
    return pose;
  }
  const Rigid3d& RigFromWorld() const {
    if (!has_pose) std::abort();
    return pose;
  }
  void SetRigFromWorld(const Rigid3d& p) {
    pose = p;
    has_pose = true;
  }
  // rig membership (colmap::Frame): null rig pointer = a trivial frame of its own
  rig_t RigId() const { return rig_id_; }
  void SetRigId(rig_t id) { rig_id_ = id; }
  Rig* RigPtr() const { return rig_ptr_; }
  void SetRigPtr(Rig* r) { rig_ptr_ = r; }
  void AddDataId(const data_t& d) { data_ids_.push_back(d); }
  const std::vector<data_t>& DataIds() const { return data_ids_; }
  const std::vector<data_t>& ImageIds() const { return data_ids_; }  // the stand-in holds camera data only

 private:
  rig_t rig_id_ = 0xffffffffu;
  Rig* rig_ptr_ = nullptr;
  std::vector<data_t> data_ids_;
};
struct Image {
  image_t image_id = 0;
  camera_t camera_id = 0;
  frame_t frame_id = 0;
  Frame* frame_ptr = nullptr;
  std::vector<mock_eigen::Vector2d> features;
  std::vector<mock_eigen::Vector3d> features_undist;
  bool IsRegistered() const { return frame_ptr != nullptr && frame_ptr->is_registered; }
  bool HasTrivialFrame() const {
    return frame_ptr == nullptr || frame_ptr->RigPtr() == nullptr ||
           frame_ptr->RigPtr()->IsRefSensor(sensor_t(SensorType::CAMERA, camera_id));
  }
  bool HasGravity() const {  // scene/image.h:78-84
    return frame_ptr->HasGravity() &&
           (HasTrivialFrame() || frame_ptr->RigPtr()->MaybeSensorFromRig(sensor_t(SensorType::CAMERA, camera_id)).has_value());
  }
  // CamFromWorld() = cam_from_rig * rig_from_world (image.h:62-65 through colmap::Frame::SensorFromWorld)
  Rigid3d CamFromWorld() const {
    const Rigid3d& rw = frame_ptr->RigFromWorld();
    if (HasTrivialFrame()) return rw;
    const Rigid3d& cr = frame_ptr->RigPtr()->SensorFromRig(sensor_t(SensorType::CAMERA, camera_id));
    const auto &a = cr.rotation, &b = rw.rotation;
    Rigid3d out;
    out.rotation = mock_eigen::Quaterniond(a.w() * b.w() - a.x() * b.x() - a.y() * b.y() - a.z() * b.z(),
                                           a.w() * b.x() + a.x() * b.w() + a.y() * b.z() - a.z() * b.y(),
                                           a.w() * b.y() - a.x() * b.z() + a.y() * b.w() + a.z() * b.x(),
                                           a.w() * b.z() + a.x() * b.y() - a.y() * b.x() + a.z() * b.w());
    // R(a) t_rw + t_cr
    const double w = a.w(), x = a.x(), y = a.y(), z = a.z();
    const double* v = rw.translation.v;
    const double tx = 2.0 * (y * v[2] - z * v[1]), ty = 2.0 * (z * v[0] - x * v[2]), tz = 2.0 * (x * v[1] - y * v[0]);
    out.translation = mock_eigen::Vector3d(v[0] + w * tx + (y * tz - z * ty) + cr.translation[0],
                                           v[1] + w * ty + (z * tx - x * tz) + cr.translation[1],
                                           v[2] + w * tz + (x * ty - y * tx) + cr.translation[2]);
    return out;
  }
};
using Observation = std::pair<image_t, feature_t>;
struct Track {
  track_t track_id = 0;
  mock_eigen::Vector3d xyz;
  bool is_initialized = false;
  std::vector<Observation> observations;
};
struct ImagePair {
  image_t image_id1 = 0, image_id2 = 0;
  bool is_valid = true;
  double weight = -1;
  Rigid3d cam2_from_cam1;
  std::vector<int> inliers;
  mock_eigen::MatrixXi matches;
};
struct ViewGraph {
  std::unordered_map<image_pair_t, ImagePair> image_pairs;
};

struct TrackEstablishmentOptions {  // controllers/track_establishment.h:9-24
  double thres_inconsistency = 10.;
  int min_num_tracks_per_view = -1;
  int min_num_view_per_track = 3;
  int max_num_view_per_track = 100;
  int max_num_tracks = 10000000;
};
struct SolverOptionsShape {  // the ceres::Solver::Options fields the reference sets (optimization_base.h:18-23)
  int max_num_iterations = 100;
  double function_tolerance = 1e-5;
};
struct OptimizationBaseOptions {
  double thres_loss_function = 1e-1;
  SolverOptionsShape solver_options;
};
struct RotationEstimatorOptions {
  int max_num_l1_iterations = 5;
  double l1_step_convergence_threshold = 0.001;
  int max_num_irls_iterations = 100;
  double irls_step_convergence_threshold = 0.001;
  double irls_loss_parameter_sigma = 5.0;
  enum WeightType { GEMAN_MCCLURE, HALF_NORM } weight_type = GEMAN_MCCLURE;
  bool skip_initialization = false;
  bool use_weight = false;
  bool use_gravity = false;
};
struct GlobalPositionerOptions : public OptimizationBaseOptions {
  enum ConstraintType { ONLY_POINTS, ONLY_CAMERAS, POINTS_AND_CAMERAS_BALANCED, POINTS_AND_CAMERAS };
  bool generate_random_positions = true, generate_random_points = true, generate_scales = true;
  bool optimize_positions = true, optimize_points = true, optimize_scales = true;
  bool use_gpu = true;
  std::string gpu_index = "-1";
  int min_num_images_gpu_solver = 50;
  int min_num_view_per_track = 3;
  unsigned seed = 1;
  ConstraintType constraint_type = ONLY_POINTS;
  double constraint_reweight_scale = 1.0;
};
struct BundleAdjusterOptions : public OptimizationBaseOptions {
  bool optimize_rig_poses = false, optimize_rotations = true, optimize_translation = true, optimize_intrinsics = true,
       optimize_principal_point = false, optimize_points = true;
  bool use_gpu = true;
  std::string gpu_index = "-1";
  int min_num_images_gpu_solver = 50;
  int min_num_view_per_track = 3;
  BundleAdjusterOptions() {
    thres_loss_function = 1.0;
    solver_options.max_num_iterations = 200;
  }
};
// The reference estimators themselves (global_positioning.h:56-137, bundle_adjustment.h:38-98): the adapter hands a solve
// to them when the caller asked for the CPU (`use_gpu == false`).  Stand-ins: they only count their calls.
class GlobalPositioner {
 public:
  explicit GlobalPositioner(const GlobalPositionerOptions& options) : options_(options) {}
  bool Solve(const ViewGraph&, std::unordered_map<rig_t, Rig>&, std::unordered_map<camera_t, Camera>&,
             std::unordered_map<frame_t, Frame>&, std::unordered_map<image_t, Image>&, std::unordered_map<track_t, Track>&) {
    ++calls();
    return true;
  }
  static int& calls() {
    static int n = 0;
    return n;
  }

 private:
  GlobalPositionerOptions options_;
};
class BundleAdjuster {
 public:
  explicit BundleAdjuster(const BundleAdjusterOptions& options) : options_(options) {}
  bool Solve(std::unordered_map<rig_t, Rig>&, std::unordered_map<camera_t, Camera>&, std::unordered_map<frame_t, Frame>&,
             std::unordered_map<image_t, Image>&, std::unordered_map<track_t, Track>&) {
    ++calls();
    return true;
  }
  static int& calls() {
    static int n = 0;
    return n;
  }

 private:
  BundleAdjusterOptions options_;
};
}  // namespace glomap
