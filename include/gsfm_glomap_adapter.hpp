// gsfm_glomap_adapter.hpp — drop-in replacements of GLOMAP's three estimator classes on libgsfm.
//
// Compile this header INSIDE a GLOMAP tree (it includes GLOMAP's own scene headers) and link
// libgsfm.so.  The classes keep the reference's names, constructors, Solve()/EstimateRotations()
// signatures and bool results, so the only change in the callers is the namespace / include:
//
//   glomap/controllers/rotation_averager.cc:58,165,180,193   RotationEstimator::EstimateRotations
//   glomap/controllers/global_mapper.cc:160                  GlobalPositioner::Solve
//   glomap/controllers/global_mapper.cc:209,221,302,315      BundleAdjuster::Solve
//
// What the adapter does is exactly the pack / unpack the reference performs implicitly by handing
// Ceres pointers into its containers (bundle_adjustment.cc:143-146, global_positioning.cc:326-328):
// flatten the unordered_map containers into the SoA problems of include/gsfm.h, call the C ABI, write
// the results back in place.  Scope = what `glomap mapper` exercises: trivial rigs and rigs whose
// cam_from_rig is KNOWN (calibrated multi-camera rigs; BundleAdjuster also refines them with optimize_rig_poses),
// 3-DoF and gravity-aligned rotation averaging, unknown cam_from_rig (cam blocks / centre blocks / optimize_rig_poses),
// positioning with all four constraint types; what the reference itself refuses (gravity with uncalibrated rigs,
// gra.cc:47-58; camera-to-camera constraints with non-trivial frames, gp.cc:169-176) returns false.
//
// NOTE: this header cannot be compiled in the libgsfm repository itself (GLOMAP / COLMAP / Eigen are
// not vendored); tests/adapter/ compiles it against interface-shaped stand-ins of those headers.
#pragma once

#include <algorithm>
#include <array>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <map>
#include <stdexcept>
#include <string>
#include <unordered_map>
#include <vector>

#include "glomap/controllers/track_establishment.h"
#include "glomap/estimators/bundle_adjustment.h"
#include "glomap/estimators/global_positioning.h"
#include "glomap/estimators/global_rotation_averaging.h"
#include "glomap/scene/types_sfm.h"
#include "gsfm.h"

namespace gsfm_glomap {

using glomap::camera_t;
using glomap::frame_t;
using glomap::image_t;
using glomap::rig_t;
using glomap::track_t;

// One libgsfm context per process and device (gpu_index "-1" = current device, as
// colmap::SetBestCudaDevice does for the reference, gp.cc:536-541 / ba.cc:79-84).
inline gsfm_ctx* Context(const std::string& gpu_index = "-1") {
  static gsfm_ctx* ctx = nullptr;
  if (ctx == nullptr) {
    int dev = -1;
    try {
      dev = std::stoi(gpu_index);  // first entry of the CSV list
    } catch (...) {
      dev = -1;
    }
    if (gsfm_ctx_create(dev, &ctx) != GSFM_OK) {
      std::fprintf(stderr, "[gsfm] no MI355X context: %s\n", gsfm_status_string(GSFM_ERR_NO_DEVICE));
      ctx = nullptr;
    }
  }
  return ctx;
}

// Not part of the reference's interface: where the wall time of the adapter's calls went, per entry point, accumulated since the
// process started (or the last ResetTimings()).  pack = flattening the reference's containers into the SoA problem, call = inside
// libgsfm (upload, solve, download), unpack = writing the results back in place.  A GLOMAP build that wants the numbers prints
// AdapterTimings() after GlobalMapper::Solve; oracle/ref_glue_mapper.cc does, for the reference's own controller.
struct Timing {
  double pack = 0.0, call = 0.0, unpack = 0.0;
  long calls = 0;
  long iterations = 0, linear_iterations = 0;  // estimators: what gsfm_report said (outer iterations, PCG iterations)
};
inline std::map<std::string, Timing>& AdapterTimings() {
  static std::map<std::string, Timing> t;
  return t;
}
inline void ResetTimings() { AdapterTimings().clear(); }

namespace detail {

class CallClock {  // one per adapter call: construction .. first Call() = pack, Call() = call, last Call() .. destruction = unpack
 public:
  explicit CallClock(const char* name) : name_(name), last_(Now()) {}
  ~CallClock() {
    Timing& t = AdapterTimings()[name_];
    const double rest = Now() - last_;
    t.pack += pack_ + (called_ ? 0.0 : rest);
    t.call += call_;
    t.unpack += called_ ? rest : 0.0;
    ++t.calls;
    t.iterations += iterations_;
    t.linear_iterations += linear_iterations_;
  }
  void Note(const gsfm_report& rep) {
    iterations_ += rep.iterations;
    linear_iterations_ += static_cast<long>(rep.linear_iterations);
  }
  template <typename F>
  auto Call(F&& f) -> decltype(f()) {
    const double a = Now();
    pack_ += a - last_;  // host work since the construction, or between two library calls of one adapter call
    auto rc = f();
    last_ = Now();
    call_ += last_ - a;
    called_ = true;
    return rc;
  }

 private:
  static double Now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
  const char* name_;
  double last_, pack_ = 0.0, call_ = 0.0;
  long iterations_ = 0, linear_iterations_ = 0;
  bool called_ = false;
};

// COLMAP CameraModelId -> GSFM_CAMERA_* (colmap/sensor/models.h; the ids coincide for the supported models)
inline int ModelOf(const glomap::Camera& cam) {
  switch (static_cast<int>(cam.model_id)) {
    case 0: return GSFM_CAMERA_SIMPLE_PINHOLE;
    case 1: return GSFM_CAMERA_PINHOLE;
    case 2: return GSFM_CAMERA_SIMPLE_RADIAL;
    case 3: return GSFM_CAMERA_RADIAL;
    case 4: return GSFM_CAMERA_OPENCV;
    case 5: return GSFM_CAMERA_OPENCV_FISHEYE;
    case 7: return GSFM_CAMERA_FOV;
    case 8: return GSFM_CAMERA_SIMPLE_RADIAL_FISHEYE;
    case 9: return GSFM_CAMERA_RADIAL_FISHEYE;
    // more than 8 parameters: the problem is packed with 16-wide intrinsics rows (gsfm_ba_problem::intr_stride)
    case 6: return GSFM_CAMERA_FULL_OPENCV;
    case 10: return GSFM_CAMERA_THIN_PRISM_FISHEYE;
    case 11: return GSFM_CAMERA_RAD_TAN_THIN_PRISM_FISHEYE;
    default: return -1;
  }
}
inline bool NeedsWideRows(int model) {
  return model == GSFM_CAMERA_FULL_OPENCV || model == GSFM_CAMERA_THIN_PRISM_FISHEYE || model == GSFM_CAMERA_RAD_TAN_THIN_PRISM_FISHEYE;
}

template <typename Quat>
inline void QuatToAngleAxis(const Quat& q, double* aa) {  // RotationToAngleAxis, math/rigid3d.cc:39-43
  const double n = std::sqrt(q.x() * q.x() + q.y() * q.y() + q.z() * q.z());
  if (n > 0.0) {
    const double ang = 2.0 * std::atan2(n, std::fabs(q.w()));
    const double k = (q.w() < 0.0 ? -ang : ang) / n;
    aa[0] = k * q.x();
    aa[1] = k * q.y();
    aa[2] = k * q.z();
  } else {
    aa[0] = aa[1] = aa[2] = 0.0;
  }
}
inline void AngleAxisToQuatWxyz(const double* aa, double* q) {  // AngleAxisToRotation, math/rigid3d.cc:45-63
  const double th = std::sqrt(aa[0] * aa[0] + aa[1] * aa[1] + aa[2] * aa[2]);
  if (th > 1e-12) {
    const double k = std::sin(0.5 * th) / th;
    q[0] = std::cos(0.5 * th);
    q[1] = k * aa[0];
    q[2] = k * aa[1];
    q[3] = k * aa[2];
  } else {
    q[0] = 1.0;
    q[1] = 0.5 * aa[0];
    q[2] = 0.5 * aa[1];
    q[3] = 0.5 * aa[2];
  }
}
template <typename Quat, typename Vec3>
inline void RotateInv(const Quat& q, const Vec3& v, double* out) {  // R(q)^T v
  const double w = q.w(), x = -q.x(), y = -q.y(), z = -q.z();
  const double tx = 2.0 * (y * v[2] - z * v[1]), ty = 2.0 * (z * v[0] - x * v[2]), tz = 2.0 * (x * v[1] - y * v[0]);
  out[0] = v[0] + w * tx + (y * tz - z * ty);
  out[1] = v[1] + w * ty + (z * tx - x * tz);
  out[2] = v[2] + w * tz + (x * ty - y * tx);
}
template <typename Quat>
inline void Rotate(const Quat& q, const double* v, double* out) {  // R(q) v
  const double w = q.w(), x = q.x(), y = q.y(), z = q.z();
  const double tx = 2.0 * (y * v[2] - z * v[1]), ty = 2.0 * (z * v[0] - x * v[2]), tz = 2.0 * (x * v[1] - y * v[0]);
  out[0] = v[0] + w * tx + (y * tz - z * ty);
  out[1] = v[1] + w * ty + (z * tx - x * tz);
  out[2] = v[2] + w * tz + (x * ty - y * tx);
}

// (w,x,y,z) Hamilton product a * b
inline void QuatMul(const double* a, const double* b, double* o) {
  o[0] = a[0] * b[0] - a[1] * b[1] - a[2] * b[2] - a[3] * b[3];
  o[1] = a[0] * b[1] + a[1] * b[0] + a[2] * b[3] - a[3] * b[2];
  o[2] = a[0] * b[2] - a[1] * b[3] + a[2] * b[0] + a[3] * b[1];
  o[3] = a[0] * b[3] + a[1] * b[2] - a[2] * b[1] + a[3] * b[0];
}
template <typename Quat>
inline void QuatWxyz(const Quat& q, double* o) {
  o[0] = q.w();
  o[1] = q.x();
  o[2] = q.y();
  o[3] = q.z();
}

// cam_from_rig of an image as (qw,qx,qy,qz,tx,ty,tz): identity for the reference sensor of its rig.  Returns false
// when the sensor is not calibrated (no value, or the NaN translation RotationEstimator leaves behind for estimated
// sensors, gra.cc:803-815): the callers that can estimate it ask CamFromRigState below instead.
inline bool KnownCamFromRig(const glomap::Image& im, std::unordered_map<rig_t, glomap::Rig>& rigs, double* cfr) {
  cfr[0] = 1.0;
  for (int j = 1; j < 7; ++j) cfr[j] = 0.0;
  if (im.HasTrivialFrame()) return true;
  const glomap::sensor_t sid(glomap::SensorType::CAMERA, im.camera_id);
  const auto opt = rigs.at(im.frame_ptr->RigId()).MaybeSensorFromRig(sid);
  if (!opt.has_value()) return false;
  QuatWxyz(opt->rotation, cfr);
  for (int j = 0; j < 3; ++j) {
    cfr[4 + j] = opt->translation[j];
    if (std::isnan(cfr[4 + j])) return false;
  }
  return true;
}

// 0: cam_from_rig fully known (cfr filled; identity for reference sensors), 1: rotation known but translation NaN — what
// RotationEstimator leaves behind for estimated sensors (gra.cc:803-815) and GlobalPositioner then estimates
// (gp.cc:354-368); cfr holds the rotation, a zero translation —, 2: no value at all.
inline int CamFromRigState(const glomap::Image& im, std::unordered_map<rig_t, glomap::Rig>& rigs, double* cfr) {
  cfr[0] = 1.0;
  for (int j = 1; j < 7; ++j) cfr[j] = 0.0;
  if (im.HasTrivialFrame()) return 0;
  const glomap::sensor_t sid(glomap::SensorType::CAMERA, im.camera_id);
  const auto opt = rigs.at(im.frame_ptr->RigId()).MaybeSensorFromRig(sid);
  if (!opt.has_value()) return 2;
  QuatWxyz(opt->rotation, cfr);
  bool nan = false;
  for (int j = 0; j < 3; ++j) nan = nan || std::isnan(opt->translation[j]);
  if (nan) return 1;
  for (int j = 0; j < 3; ++j) cfr[4 + j] = opt->translation[j];
  return 0;
}

// quaternion (w,x,y,z) of a rotation matrix given through (row, col) access (Eigen::Matrix3d)
template <typename Mat>
inline void MatToQuatWxyz(const Mat& R, double* q) {
  const double tr = R(0, 0) + R(1, 1) + R(2, 2);
  if (tr > 0.0) {
    const double t = std::sqrt(tr + 1.0);
    q[0] = 0.5 * t;
    q[1] = (R(2, 1) - R(1, 2)) * 0.5 / t;
    q[2] = (R(0, 2) - R(2, 0)) * 0.5 / t;
    q[3] = (R(1, 0) - R(0, 1)) * 0.5 / t;
    return;
  }
  int i = 0;
  if (R(1, 1) > R(0, 0)) i = 1;
  if (R(2, 2) > R(i, i)) i = 2;
  const int j = (i + 1) % 3, k = (j + 1) % 3;
  const double t = std::sqrt(R(i, i) - R(j, j) - R(k, k) + 1.0);
  q[1 + i] = 0.5 * t;
  q[0] = (R(k, j) - R(j, k)) * 0.5 / t;
  q[1 + j] = (R(j, i) + R(i, j)) * 0.5 / t;
  q[1 + k] = (R(k, i) + R(i, k)) * 0.5 / t;
}

// colmap::AverageQuaternions with unit weights: principal eigenvector of sum q q^T (cyclic Jacobi on the 4 x 4).
inline void AverageQuaternions(const std::vector<std::array<double, 4>>& qs, double* out) {
  double A[4][4] = {{0}}, V[4][4] = {{1, 0, 0, 0}, {0, 1, 0, 0}, {0, 0, 1, 0}, {0, 0, 0, 1}};
  for (const auto& q : qs)
    for (int i = 0; i < 4; ++i)
      for (int j = 0; j < 4; ++j) A[i][j] += q[i] * q[j];
  for (int sweep = 0; sweep < 30; ++sweep) {
    double off = 0.0;
    for (int i = 0; i < 4; ++i)
      for (int j = i + 1; j < 4; ++j) off += A[i][j] * A[i][j];
    if (off < 1e-30) break;
    for (int p = 0; p < 4; ++p)
      for (int q = p + 1; q < 4; ++q) {
        if (std::fabs(A[p][q]) < 1e-300) continue;
        const double th = 0.5 * std::atan2(2.0 * A[p][q], A[q][q] - A[p][p]);
        const double c = std::cos(th), sn = std::sin(th);
        for (int k = 0; k < 4; ++k) {  // A <- A G
          const double akp = A[k][p], akq = A[k][q];
          A[k][p] = c * akp - sn * akq;
          A[k][q] = sn * akp + c * akq;
        }
        for (int k = 0; k < 4; ++k) {  // A <- G^T A
          const double apk = A[p][k], aqk = A[q][k];
          A[p][k] = c * apk - sn * aqk;
          A[q][k] = sn * apk + c * aqk;
        }
        for (int k = 0; k < 4; ++k) {
          const double vkp = V[k][p], vkq = V[k][q];
          V[k][p] = c * vkp - sn * vkq;
          V[k][q] = sn * vkp + c * vkq;
        }
      }
  }
  int best = 0;
  for (int i = 1; i < 4; ++i)
    if (A[i][i] > A[best][best]) best = i;
  const double sg = V[0][best] < 0.0 ? -1.0 : 1.0;
  for (int i = 0; i < 4; ++i) out[i] = sg * V[i][best];
}

// Dense frame index over the frames the estimators touch.
struct FrameIndex {
  std::unordered_map<frame_t, int> of;
  std::vector<frame_t> ids;
  int Add(frame_t f) {
    auto it = of.find(f);
    if (it != of.end()) return it->second;
    const int n = static_cast<int>(ids.size());
    of.emplace(f, n);
    ids.push_back(f);
    return n;
  }
  // Dense indices in ASCENDING id order for the given frames (COLMAP ids grow in import = capture order, so co-visible
  // frames become neighbours in memory and the second-level preconditioner's index clusters are clusters of the scene),
  // instead of the iteration order of the caller's hash map.  `map_order` = the frames as the reference would walk them;
  // returns the dense index of each of them in that order (gsfm_gp_problem::cam_draw_order; gauge choices that follow the
  // reference's walk are made from it as well).
  std::vector<int32_t> AddSorted(const std::vector<frame_t>& map_order) {
    std::vector<frame_t> sorted(map_order);
    std::sort(sorted.begin(), sorted.end());
    for (frame_t f : sorted) Add(f);
    std::vector<int32_t> walk;
    walk.reserve(map_order.size());
    for (frame_t f : map_order) walk.push_back(of.at(f));
    return walk;
  }
};

// image id -> image (and its dense frame index, once known) for the packers' inner loops, which look an image up once per
// OBSERVATION: a plain vector when the ids are small integers (COLMAP numbers images 1 .. N), the caller's hash map otherwise.
// At 1.5 M observations the three hash lookups per observation were most of the packing time (tools/exp_dropin_mapper_scale.py).
class ImageTable {
 public:
  explicit ImageTable(const std::unordered_map<image_t, glomap::Image>& images) : images_(images) {
    image_t max_id = 0;
    for (const auto& [id, im] : images) max_id = std::max(max_id, id);
    if (!images.empty() && static_cast<size_t>(max_id) <= 4 * images.size() + 1024) {
      dense_.assign(static_cast<size_t>(max_id) + 1, nullptr);
      node_.assign(static_cast<size_t>(max_id) + 1, -1);
      for (const auto& [id, im] : images) dense_[id] = &im;
    }
  }
  const glomap::Image* Find(image_t id) const {
    if (!dense_.empty()) return id < dense_.size() ? dense_[id] : nullptr;
    auto it = images_.find(id);
    return it == images_.end() ? nullptr : &it->second;
  }
  const glomap::Image& At(image_t id) const {  // std::unordered_map::at semantics
    const glomap::Image* im = Find(id);
    if (im == nullptr) throw std::out_of_range("gsfm_glomap: image id not in the images map");
    return *im;
  }
  // fidx.Add(im.frame_id), remembered per image
  int Node(image_t id, const glomap::Image& im, FrameIndex& fidx) {
    if (node_.empty()) return fidx.Add(im.frame_id);
    int& n = node_[id];
    if (n < 0) n = fidx.Add(im.frame_id);
    return n;
  }

 private:
  const std::unordered_map<image_t, glomap::Image>& images_;
  std::vector<const glomap::Image*> dense_;
  std::vector<int> node_;
};

// Track-major observation lists shared by GP and BA (gp.cc:270-375, ba.cc:115-190).
struct TrackPack {
  std::vector<track_t> track_ids;
  std::vector<int64_t> pt_offset{0};
  std::vector<int32_t> obs_cam;
  std::vector<image_t> obs_image;
  std::vector<uint32_t> obs_feature;
};

// `min_views` is tested on the RAW observation count like the reference does (track.observations.size(), gp.cc:258 /
// ba.cc:122) — BEFORE unregistered images / failed undistortions are dropped — so a track with 3 raw observations of
// which one is unusable is still optimised.  Shorter tracks are not packed at all (the estimators leave them untouched).
// keep_empty (global positioning): a track that passes the raw count but has NO usable observation is packed as a
// zero-length track — the reference still draws its random start and marks it initialised (gp.cc:258-264), so it has
// to take its turn in the random stream; the library is then called with min_num_view_per_track = 0 (every packed
// track is "used").  Bundle adjustment drops such tracks (nothing happens to them in ba.cc:121-133) and calls with 1.
// Tracks are packed in ASCENDING id order (track establishment numbers tracks by their first image / feature, so sorted ids
// keep tracks of neighbouring images together — the caller's hash map does not); draw_order, when asked for, receives the
// packed index of every packed track in the hash map's own iteration order (gsfm_gp_problem::pt_draw_order).
template <typename Keep>
inline TrackPack PackTracks(std::unordered_map<image_t, glomap::Image>& images,
                            std::unordered_map<track_t, glomap::Track>& tracks, FrameIndex& fidx, Keep keep,
                            size_t min_views = 0, bool keep_empty = false, std::vector<int32_t>* draw_order = nullptr) {
  TrackPack tp;
  std::vector<track_t> map_order;
  map_order.reserve(tracks.size());
  for (auto& [tid, track] : tracks) map_order.push_back(tid);
  std::vector<track_t> sorted(map_order);
  std::sort(sorted.begin(), sorted.end());
  ImageTable table(images);
  for (track_t tid : sorted) {
    auto& track = tracks.at(tid);
    if (track.observations.size() < min_views) continue;
    const size_t before = tp.obs_cam.size();
    for (const auto& obs : track.observations) {
      const glomap::Image* im = table.Find(obs.first);
      if (im == nullptr || !keep(*im, obs.second)) continue;
      tp.obs_cam.push_back(table.Node(obs.first, *im, fidx));
      tp.obs_image.push_back(obs.first);
      tp.obs_feature.push_back(obs.second);
    }
    if (tp.obs_cam.size() == before && !keep_empty) continue;
    tp.track_ids.push_back(tid);
    tp.pt_offset.push_back(static_cast<int64_t>(tp.obs_cam.size()));
  }
  if (draw_order != nullptr) {
    draw_order->clear();
    for (track_t tid : map_order) {
      auto it = std::lower_bound(tp.track_ids.begin(), tp.track_ids.end(), tid);  // packed ids are ascending
      if (it != tp.track_ids.end() && *it == tid) draw_order->push_back(static_cast<int32_t>(it - tp.track_ids.begin()));
    }
  }
  return tp;
}

inline bool AllTrivial(std::unordered_map<image_t, glomap::Image>& images) {
  for (auto& [id, im] : images)
    if (im.frame_ptr != nullptr && !im.HasTrivialFrame()) return false;
  return true;
}

}  // namespace detail

// ---------------------------------------------------------------------------------------------
// RotationEstimator (global_rotation_averaging.h:77-141)
// ---------------------------------------------------------------------------------------------
class RotationEstimator {
 public:
  explicit RotationEstimator(const glomap::RotationEstimatorOptions& options) : options_(options) {}

  bool EstimateRotations(const glomap::ViewGraph& view_graph, std::unordered_map<rig_t, glomap::Rig>& rigs,
                         std::unordered_map<frame_t, glomap::Frame>& frames,
                         std::unordered_map<image_t, glomap::Image>& images) {
    gsfm_ctx* ctx = Context();
    detail::CallClock clk("RotationEstimator::EstimateRotations");
    if (ctx == nullptr) return false;
    const bool rigged = !detail::AllTrivial(images);
    if (options_.use_gravity) {  // gra.cc:47-58: gravity needs every cam_from_rig
      for (auto& [rig_id, rig] : rigs)
        for (const auto& [sid, sensor] : rig.NonRefSensors())
          if (!sensor.has_value()) return false;
    }
    // Two choices of the reference follow the iteration order of two DIFFERENT hash maps: the spanning tree of the start is
    // rooted at the first registered image `images` yields (tree.cc:84-88, 141 — it keeps the identity rotation), the gauge is
    // the first registered frame `frames` yields (gra.cc:248-257).  The library roots its tree at node 0 and takes the gauge as
    // fixed_node: the root's frame is numbered first, the others in the order of `frames`.
    const bool tree_start = !options_.skip_initialization && !options_.use_gravity;  // gra.cc:60-62
    tree_root_ = kNoImage;
    for (const auto& [id, im] : images)
      if (im.frame_ptr != nullptr && im.IsRegistered()) {
        tree_root_ = id;
        break;
      }
    detail::FrameIndex fidx;
    if (tree_start && tree_root_ != kNoImage) fidx.Add(images.at(tree_root_).frame_id);
    gauge_node_ = -1;
    for (auto& [fid, fr] : frames)
      if (fr.is_registered) {
        const int n = fidx.Add(fid);  // gra.cc:193-227
        if (gauge_node_ < 0) gauge_node_ = n;  // gra.cc:248-257
      }
    const int N = static_cast<int>(fidx.ids.size());
    if (N == 0) return false;
    if (rigged && !options_.use_gravity) {  // sensors whose cam_from_rig is to be estimated (no value, or NaN translation: gra.cc:173-191)
      for (auto& [id, im] : images) {
        double cfr[7];
        if (im.frame_ptr != nullptr && im.IsRegistered() && detail::CamFromRigState(im, rigs, cfr) != 0)
          return EstimateWithCamBlocks(ctx, view_graph, rigs, frames, images, fidx, clk);
      }
    }
    std::vector<int32_t> ei, ej, en;
    std::vector<double> eq, ew;
    // image-level copy of the view graph (calibrated rigs only): the spanning-tree initialisation runs over IMAGES
    std::unordered_map<image_t, int> img_of;
    std::vector<image_t> img_ids;
    std::vector<int32_t> ii, ij, in_;
    std::vector<double> iq, iw;
    auto image_index = [&](image_t id) {
      auto it = img_of.find(id);
      if (it != img_of.end()) return it->second;
      img_of.emplace(id, static_cast<int>(img_ids.size()));
      img_ids.push_back(id);
      return static_cast<int>(img_ids.size()) - 1;
    };
    if (rigged && tree_start && tree_root_ != kNoImage) image_index(tree_root_);  // node 0 of the image-level graph = the tree's root
    for (const auto& [pid, pair] : view_graph.image_pairs) {
      if (!pair.is_valid) continue;
      const auto& i1 = images.at(pair.image_id1);
      const auto& i2 = images.at(pair.image_id2);
      if (!i1.IsRegistered() || !i2.IsRegistered()) continue;
      double q21[4], qrel[4];
      detail::QuatWxyz(pair.cam2_from_cam1.rotation, q21);
      for (int j = 0; j < 4; ++j) qrel[j] = q21[j];
      if (rigged) {
        // R_rel = R_cam2_from_rig2^T * R_cam2_from_cam1 * R_cam1_from_rig1 (gra.cc:306-309); uncalibrated sensors took the
        // EstimateWithCamBlocks path above (with use_gravity they are refused, gra.cc:47-58)
        double c1[7], c2[7], tmp[4];
        if (!detail::KnownCamFromRig(i1, rigs, c1) || !detail::KnownCamFromRig(i2, rigs, c2)) return false;
        ii.push_back(image_index(pair.image_id1));
        ij.push_back(image_index(pair.image_id2));
        iq.insert(iq.end(), q21, q21 + 4);
        iw.push_back(pair.weight);
        in_.push_back(static_cast<int32_t>(pair.inliers.size()));
        if (i1.frame_id == i2.frame_id) continue;  // both sensors of one frame: no constraint (gra.cc:300-304)
        const double c2inv[4] = {c2[0], -c2[1], -c2[2], -c2[3]};
        detail::QuatMul(c2inv, q21, tmp);
        detail::QuatMul(tmp, c1, qrel);
      }
      if (options_.use_gravity) {  // R_align2^T * R_rel * R_align1 for the images that have gravity (gra.cc:315-327)
        double qa[4], tmp[4];
        if (i1.HasGravity()) {
          detail::MatToQuatWxyz(i1.frame_ptr->gravity_info.GetRAlign(), qa);
          detail::QuatMul(qrel, qa, tmp);
          for (int j = 0; j < 4; ++j) qrel[j] = tmp[j];
        }
        if (i2.HasGravity()) {
          detail::MatToQuatWxyz(i2.frame_ptr->gravity_info.GetRAlign(), qa);
          const double qinv[4] = {qa[0], -qa[1], -qa[2], -qa[3]};
          detail::QuatMul(qinv, qrel, tmp);
          for (int j = 0; j < 4; ++j) qrel[j] = tmp[j];
        }
      }
      ei.push_back(fidx.of.at(i1.frame_id));
      ej.push_back(fidx.of.at(i2.frame_id));
      eq.insert(eq.end(), qrel, qrel + 4);
      ew.push_back(pair.weight);
      en.push_back(static_cast<int32_t>(pair.inliers.size()));
    }
    std::vector<double> rot(3 * static_cast<size_t>(N));
    std::vector<uint8_t> node_gravity(static_cast<size_t>(N), 0);
    int fixed_node = gauge_node_, first_gravity = -1;
    for (int n = 0; n < N; ++n) {
      auto& fr = frames.at(fidx.ids[n]);
      if (options_.use_gravity && fr.gravity_info.has_gravity) {
        // one unknown: the angle of R_align^T * R_rig_from_world about the vertical (gra.cc:207-211), carried as (0, angle, 0)
        double qa[4], qw[4], qp[4], aa[3];
        detail::MatToQuatWxyz(fr.gravity_info.GetRAlign(), qa);
        detail::QuatWxyz(fr.RigFromWorld().rotation, qw);
        const double qinv[4] = {qa[0], -qa[1], -qa[2], -qa[3]};
        detail::QuatMul(qinv, qw, qp);
        detail::QuatToAngleAxis(QuatView{qp[0], qp[1], qp[2], qp[3]}, aa);
        rot[3 * n] = 0.0;
        rot[3 * n + 1] = aa[1];
        rot[3 * n + 2] = 0.0;
        node_gravity[n] = 1;
        if (first_gravity < 0) first_gravity = n;  // the gauge goes to the first gravity frame (gra.cc:212-216)
        continue;
      }
      if (!fr.HasPose()) fr.SetRigFromWorld(glomap::Rigid3d());  // gra.cc:219-222 (colmap::Frame::RigFromWorld() throws without a pose)
      detail::QuatToAngleAxis(fr.RigFromWorld().rotation, &rot[3 * n]);
    }
    if (first_gravity >= 0) fixed_node = first_gravity;
    bool skip_init = options_.skip_initialization || options_.use_gravity;  // gra.cc:60-62: no spanning tree with gravity
    if (rigged && !skip_init) {
      // InitializeFromMaximumSpanningTree over the images (gra.cc:87-138) — a zero-iteration gsfm_ra_solve of the
      // image-level graph — then ConvertRotationsFromImageToRig (rotation_initializer.cc:86-121): rig_from_world of a
      // frame = average over its images of cam_from_rig^-1 * cam_from_world
      const int NI = static_cast<int>(img_ids.size());
      if (NI > 0) {
        std::vector<double> irot(3 * static_cast<size_t>(NI), 0.0);
        gsfm_ra_options o0;
        gsfm_ra_options_default(&o0);
        o0.max_num_l1_iterations = 0;
        o0.max_num_irls_iterations = 0;
        gsfm_ra_problem p0{};
        p0.mem = GSFM_MEM_HOST;
        p0.num_nodes = NI;
        p0.num_edges = static_cast<int64_t>(ii.size());
        p0.edge_i = ii.data();
        p0.edge_j = ij.data();
        p0.edge_q = iq.data();
        p0.edge_weight = iw.data();
        p0.edge_ninl = in_.data();
        p0.fixed_node = 0;
        gsfm_report r0;
        if (clk.Call([&] { return gsfm_ra_solve(ctx, &p0, &o0, irot.data(), &r0); }) != GSFM_OK) return false;
    clk.Note(r0);
        std::vector<std::vector<std::array<double, 4>>> per_frame(static_cast<size_t>(N));
        for (int i = 0; i < NI; ++i) {
          const auto& im = images.at(img_ids[i]);
          double cfr[7], qcw[4], qrw[4];
          detail::KnownCamFromRig(im, rigs, cfr);
          detail::AngleAxisToQuatWxyz(&irot[3 * i], qcw);
          const double cinv[4] = {cfr[0], -cfr[1], -cfr[2], -cfr[3]};
          detail::QuatMul(cinv, qcw, qrw);
          auto& list = per_frame[static_cast<size_t>(fidx.of.at(im.frame_id))];
          if (!list.empty() && list[0][0] * qrw[0] + list[0][1] * qrw[1] + list[0][2] * qrw[2] + list[0][3] * qrw[3] < 0.0)
            for (double& v : qrw) v = -v;
          list.push_back({qrw[0], qrw[1], qrw[2], qrw[3]});
        }
        for (int n = 0; n < N; ++n) {
          if (per_frame[n].empty()) continue;
          double qa[4];
          detail::AverageQuaternions(per_frame[n], qa);
          struct Q {
            double w_, x_, y_, z_;
            double w() const { return w_; }
            double x() const { return x_; }
            double y() const { return y_; }
            double z() const { return z_; }
          } qq{qa[0], qa[1], qa[2], qa[3]};
          detail::QuatToAngleAxis(qq, &rot[3 * n]);
        }
      }
      skip_init = true;
    }
    gsfm_ra_options o;
    gsfm_ra_options_default(&o);
    o.max_num_l1_iterations = options_.max_num_l1_iterations;
    o.l1_step_convergence_threshold = options_.l1_step_convergence_threshold;
    o.max_num_irls_iterations = options_.max_num_irls_iterations;
    o.irls_step_convergence_threshold = options_.irls_step_convergence_threshold;
    o.irls_loss_parameter_sigma = options_.irls_loss_parameter_sigma;
    o.weight_type = static_cast<int>(options_.weight_type);
    o.skip_initialization = skip_init;
    o.use_weight = options_.use_weight;
    o.use_gravity = options_.use_gravity ? 1 : 0;
    gsfm_ra_problem p{};
    p.mem = GSFM_MEM_HOST;
    p.num_nodes = N;
    p.node_gravity = first_gravity >= 0 ? node_gravity.data() : nullptr;
    p.num_edges = static_cast<int64_t>(ei.size());
    p.edge_i = ei.data();
    p.edge_j = ej.data();
    p.edge_q = eq.data();
    p.edge_weight = ew.data();
    p.edge_ninl = en.data();
    p.fixed_node = fixed_node;
    gsfm_report rep;
    if (clk.Call([&] { return gsfm_ra_solve(ctx, &p, &o, rot.data(), &rep); }) != GSFM_OK) return false;
    clk.Note(rep);
    // ConvertResults (gra.cc:774-816): rotation written, translation zeroed
    for (int n = 0; n < N; ++n) {
      double q[4];
      detail::AngleAxisToQuatWxyz(&rot[3 * n], q);
      auto& fr = frames.at(fidx.ids[n]);
      if (node_gravity[n]) {  // R_align * AngleToRotUp(angle), gra.cc:786-793
        double qa[4], qr[4];
        detail::MatToQuatWxyz(fr.gravity_info.GetRAlign(), qa);
        detail::QuatMul(qa, q, qr);
        for (int j = 0; j < 4; ++j) q[j] = qr[j];
      }
      auto pose = fr.RigFromWorld();
      pose.rotation = decltype(pose.rotation)(q[0], q[1], q[2], q[3]);
      pose.translation = decltype(pose.translation)(0.0, 0.0, 0.0);
      fr.SetRigFromWorld(pose);
    }
    return true;
  }

 private:
  struct QuatView {
    double w_, x_, y_, z_;
    double w() const { return w_; }
    double x() const { return x_; }
    double y() const { return y_; }
    double z() const { return z_; }
  };

  // cam_from_rig ROTATIONS among the unknowns (gra.cc:173-191, 396-446, 646-690): the view graph goes to the library
  // image by image, each image with its frame and — for the sensors to estimate — its cam block (one per camera id,
  // like camera_id_to_idx_); calibrated sensors are folded into the relative rotations (gra.cc:306-309).
  bool EstimateWithCamBlocks(gsfm_ctx* ctx, const glomap::ViewGraph& view_graph, std::unordered_map<rig_t, glomap::Rig>& rigs,
                             std::unordered_map<frame_t, glomap::Frame>& frames,
                             std::unordered_map<image_t, glomap::Image>& images, detail::FrameIndex& fidx, detail::CallClock& clk) {
    const int N = static_cast<int>(fidx.ids.size());
    std::unordered_map<image_t, int> img_of;
    std::vector<image_t> img_ids;
    std::vector<int32_t> image_frame, image_cam, ii, ij, in_;
    std::vector<uint8_t> image_is_ref;
    std::vector<std::array<double, 4>> image_fold;  // known cam_from_rig rotation of the image (identity otherwise)
    std::vector<double> iq, iw;
    std::unordered_map<camera_t, int> cam_of;
    std::vector<camera_t> cam_ids;
    std::vector<rig_t> cam_rig;
    std::vector<uint8_t> cam_has_start;
    std::vector<double> cam_rot;  // [C][3]
    auto image_index = [&](image_t id) {
      auto it = img_of.find(id);
      if (it != img_of.end()) return it->second;
      const auto& im = images.at(id);
      double cfr[7];
      const int state = detail::CamFromRigState(im, rigs, cfr);
      int block = -1;
      if (state != 0) {
        auto ct = cam_of.find(im.camera_id);
        if (ct == cam_of.end()) {
          ct = cam_of.emplace(im.camera_id, static_cast<int>(cam_ids.size())).first;
          cam_ids.push_back(im.camera_id);
          cam_rig.push_back(im.frame_ptr->RigId());
          cam_has_start.push_back(state == 1 ? 1 : 0);
          double aa[3] = {0.0, 0.0, 0.0};  // gra.cc:231-242: the stored rotation when there is one, else zero
          if (state == 1) detail::QuatToAngleAxis(QuatView{cfr[0], cfr[1], cfr[2], cfr[3]}, aa);
          cam_rot.insert(cam_rot.end(), aa, aa + 3);
        }
        block = ct->second;
      }
      const int idx = static_cast<int>(img_ids.size());
      img_of.emplace(id, idx);
      img_ids.push_back(id);
      image_frame.push_back(fidx.of.at(im.frame_id));
      image_cam.push_back(block);
      image_is_ref.push_back(im.HasTrivialFrame() ? 1 : 0);
      image_fold.push_back(state == 0 ? std::array<double, 4>{cfr[0], cfr[1], cfr[2], cfr[3]} : std::array<double, 4>{1, 0, 0, 0});
      return idx;
    };
    if (!options_.skip_initialization && tree_root_ != kNoImage) image_index(tree_root_);  // node 0 = the spanning tree's root
    for (const auto& [pid, pair] : view_graph.image_pairs) {
      if (!pair.is_valid) continue;
      const auto& i1 = images.at(pair.image_id1);
      const auto& i2 = images.at(pair.image_id2);
      if (!i1.IsRegistered() || !i2.IsRegistered()) continue;
      const int a = image_index(pair.image_id1), b = image_index(pair.image_id2);
      if (image_frame[a] == image_frame[b] && image_cam[a] < 0 && image_cam[b] < 0) continue;  // gra.cc:300-304
      double q21[4], tmp[4], qrel[4];
      detail::QuatWxyz(pair.cam2_from_cam1.rotation, q21);
      const double c2inv[4] = {image_fold[b][0], -image_fold[b][1], -image_fold[b][2], -image_fold[b][3]};
      detail::QuatMul(c2inv, q21, tmp);
      detail::QuatMul(tmp, image_fold[a].data(), qrel);  // gra.cc:306-309
      ii.push_back(a);
      ij.push_back(b);
      iq.insert(iq.end(), qrel, qrel + 4);
      iw.push_back(pair.weight);
      in_.push_back(static_cast<int32_t>(pair.inliers.size()));
    }
    const int NI = static_cast<int>(img_ids.size()), C = static_cast<int>(cam_ids.size());
    if (NI == 0) return false;
    std::vector<double> rot(3 * static_cast<size_t>(N));
    for (int n = 0; n < N; ++n) {
      auto& fr = frames.at(fidx.ids[n]);
      if (!fr.HasPose()) fr.SetRigFromWorld(glomap::Rigid3d());  // gra.cc:219-222
      detail::QuatToAngleAxis(fr.RigFromWorld().rotation, &rot[3 * n]);
    }
    gsfm_ra_problem p{};
    p.mem = GSFM_MEM_HOST;
    p.num_edges = static_cast<int64_t>(ii.size());
    p.edge_i = ii.data();
    p.edge_j = ij.data();
    p.edge_q = iq.data();
    p.edge_weight = iw.data();
    p.edge_ninl = in_.data();
    if (!options_.skip_initialization) {
      // InitializeFromMaximumSpanningTree over the images (gra.cc:87-138), a zero-iteration solve of the image graph ...
      std::vector<double> irot(3 * static_cast<size_t>(NI), 0.0);
      gsfm_ra_options o0;
      gsfm_ra_options_default(&o0);
      o0.max_num_l1_iterations = 0;
      o0.max_num_irls_iterations = 0;
      p.num_nodes = NI;
      p.fixed_node = 0;
      gsfm_report r0;
      if (clk.Call([&] { return gsfm_ra_solve(ctx, &p, &o0, irot.data(), &r0); }) != GSFM_OK) return false;
    clk.Note(r0);
      // ... then ConvertRotationsFromImageToRig (rotation_initializer.cc:7-125).  Images without a block carry rig-level
      // rotations here (their cam_from_rig is folded into the edges).
      std::vector<std::array<double, 4>> qimg(static_cast<size_t>(NI));
      for (int i = 0; i < NI; ++i) detail::AngleAxisToQuatWxyz(&irot[3 * i], qimg[i].data());
      std::vector<int> ref(static_cast<size_t>(N), -1);
      for (int i = NI - 1; i >= 0; --i)
        if (image_is_ref[i]) ref[image_frame[i]] = i;
      auto push_aligned = [](std::vector<std::array<double, 4>>& list, const double* q) {
        double s = 1.0;
        if (!list.empty() && list[0][0] * q[0] + list[0][1] * q[1] + list[0][2] * q[2] + list[0][3] * q[3] < 0.0) s = -1.0;
        list.push_back({s * q[0], s * q[1], s * q[2], s * q[3]});
      };
      std::vector<std::array<double, 4>> qcam(static_cast<size_t>(C), {1, 0, 0, 0});
      for (int c = 0; c < C; ++c) {
        if (cam_has_start[c]) {  // a stored rotation is kept (:52-57 skips sensors that have a value)
          detail::AngleAxisToQuatWxyz(&cam_rot[3 * c], qcam[c].data());
          continue;
        }
        std::vector<std::array<double, 4>> list;
        for (int i = 0; i < NI; ++i) {
          if (image_cam[i] != c || ref[image_frame[i]] < 0) continue;
          const auto& qr = qimg[ref[image_frame[i]]];
          const double rinv[4] = {qr[0], -qr[1], -qr[2], -qr[3]};
          double q[4];
          detail::QuatMul(qimg[i].data(), rinv, q);  // cam_from_ref_cam, :66-72
          push_aligned(list, q);
        }
        if (list.empty()) continue;
        detail::AverageQuaternions(list, qcam[c].data());
        detail::QuatToAngleAxis(QuatView{qcam[c][0], qcam[c][1], qcam[c][2], qcam[c][3]}, &cam_rot[3 * c]);
      }
      std::vector<std::vector<std::array<double, 4>>> per_frame(static_cast<size_t>(N));
      for (int i = 0; i < NI; ++i) {
        double q[4] = {qimg[i][0], qimg[i][1], qimg[i][2], qimg[i][3]};
        if (image_cam[i] >= 0) {
          const auto& qc = qcam[image_cam[i]];
          const double cinv[4] = {qc[0], -qc[1], -qc[2], -qc[3]};
          detail::QuatMul(cinv, qimg[i].data(), q);  // cam_from_rig^-1 * cam_from_world, :109-111
        }
        push_aligned(per_frame[image_frame[i]], q);
      }
      for (int n = 0; n < N; ++n) {
        if (per_frame[n].empty()) continue;
        double qa[4];
        detail::AverageQuaternions(per_frame[n], qa);
        detail::QuatToAngleAxis(QuatView{qa[0], qa[1], qa[2], qa[3]}, &rot[3 * n]);
      }
    }
    gsfm_ra_options o;
    gsfm_ra_options_default(&o);
    o.max_num_l1_iterations = options_.max_num_l1_iterations;
    o.l1_step_convergence_threshold = options_.l1_step_convergence_threshold;
    o.max_num_irls_iterations = options_.max_num_irls_iterations;
    o.irls_step_convergence_threshold = options_.irls_step_convergence_threshold;
    o.irls_loss_parameter_sigma = options_.irls_loss_parameter_sigma;
    o.weight_type = static_cast<int>(options_.weight_type);
    o.skip_initialization = 1;
    o.use_weight = options_.use_weight;
    p.num_nodes = N;
    p.fixed_node = gauge_node_;
    p.num_images = NI;
    p.image_frame = image_frame.data();
    p.image_cam = image_cam.data();
    p.num_cams = C;
    p.cam_rot_aa = cam_rot.data();
    gsfm_report rep;
    if (clk.Call([&] { return gsfm_ra_solve(ctx, &p, &o, rot.data(), &rep); }) != GSFM_OK) return false;
    clk.Note(rep);
    for (int n = 0; n < N; ++n) {  // ConvertResults (gra.cc:774-799)
      double q[4];
      detail::AngleAxisToQuatWxyz(&rot[3 * n], q);
      auto& fr = frames.at(fidx.ids[n]);
      auto pose = fr.RigFromWorld();
      pose.rotation = decltype(pose.rotation)(q[0], q[1], q[2], q[3]);
      pose.translation = decltype(pose.translation)(0.0, 0.0, 0.0);
      fr.SetRigFromWorld(pose);
    }
    const double nan = std::nan("");
    for (int c = 0; c < C; ++c) {  // gra.cc:801-815: rotation set, translation NaN ("no translation yet")
      double q[4];
      detail::AngleAxisToQuatWxyz(&cam_rot[3 * c], q);
      glomap::Rigid3d cfr;
      cfr.rotation = decltype(cfr.rotation)(q[0], q[1], q[2], q[3]);
      cfr.translation = decltype(cfr.translation)(nan, nan, nan);
      rigs.at(cam_rig[c]).SetSensorFromRig(glomap::sensor_t(glomap::SensorType::CAMERA, cam_ids[c]), cfr);
    }
    return true;
  }

  const glomap::RotationEstimatorOptions& options_;  // reference keeps a reference too (global_rotation_averaging.h:140)
  static constexpr image_t kNoImage = static_cast<image_t>(-1);
  image_t tree_root_ = kNoImage;  // first registered image of `images` (tree.cc:84-88)
  int gauge_node_ = 0;            // node of the first registered frame of `frames` (gra.cc:248-257)
};

// ---------------------------------------------------------------------------------------------
// GlobalPositioner (global_positioning.h:56-137), all four constraint types
// ---------------------------------------------------------------------------------------------
class GlobalPositioner {
 public:
  explicit GlobalPositioner(const glomap::GlobalPositionerOptions& options) : options_(options) {}
  glomap::GlobalPositionerOptions& GetOptions() { return options_; }

  bool Solve(const glomap::ViewGraph& view_graph, std::unordered_map<rig_t, glomap::Rig>& rigs,
             std::unordered_map<camera_t, glomap::Camera>& cameras, std::unordered_map<frame_t, glomap::Frame>& frames,
             std::unordered_map<image_t, glomap::Image>& images, std::unordered_map<track_t, glomap::Track>& tracks) {
    // use_gpu == false: the caller asked for the CPU solver (gp.cc:506-549 picks Ceres' CPU linear algebra then) — that IS
    // the reference class, so the solve goes there unchanged instead of silently running on the device.
    // min_num_images_gpu_solver is NOT consulted: in the reference it keeps small problems away from cuDSS / CUDA dense
    // factorisations whose set-up outweighs them; libgsfm has its own small-problem path (single-workgroup PCG), so a
    // caller that leaves use_gpu on gets the device at every size.
    if (!options_.use_gpu) return glomap::GlobalPositioner(options_).Solve(view_graph, rigs, cameras, frames, images, tracks);
    gsfm_ctx* ctx = Context(options_.gpu_index);
    detail::CallClock clk("GlobalPositioner::Solve");
    if (ctx == nullptr) return false;
    const bool with_pairs = options_.constraint_type != glomap::GlobalPositionerOptions::ONLY_POINTS;
    const bool with_points = options_.constraint_type != glomap::GlobalPositionerOptions::ONLY_CAMERAS;
    if (images.empty()) return false;                                  // gp.cc:37-40
    if (view_graph.image_pairs.empty() && with_pairs) return false;   // gp.cc:41-45
    if (tracks.empty() && with_points) return false;                   // gp.cc:46-50
    const bool rigged = !detail::AllTrivial(images);  // calibrated multi-camera rigs: observations are keyed by image
    if (with_pairs && rigged) return false;  // "only trivial frames are supported for the camera to camera constraints" (gp.cc:169-176)
    detail::FrameIndex fidx;
    std::vector<frame_t> frame_walk;
    for (auto& [fid, fr] : frames) frame_walk.push_back(fid);  // every frame: ConvertResults rewrites all of them (gp.cc:566-572)
    // dense indices by ascending id; the random start is still drawn in the hash maps' iteration order (gp.cc:128-162, 258-264)
    const std::vector<int32_t> cam_draw_order = fidx.AddSorted(frame_walk);
    std::vector<int32_t> pt_draw_order;
    auto keep = [](const glomap::Image& im, uint32_t f) {  // gp.cc:279-292
      if (!im.IsRegistered()) return false;
      const auto& v = im.features_undist[f];
      return !(std::isnan(v[0]) || std::isnan(v[1]) || std::isnan(v[2]));
    };
    detail::TrackPack tp = detail::PackTracks(images, tracks, fidx, keep, static_cast<size_t>(options_.min_num_view_per_track),
                                              /*keep_empty=*/true, &pt_draw_order);
    const int N = static_cast<int>(fidx.ids.size());
    const int64_t P = static_cast<int64_t>(tp.track_ids.size()), M = static_cast<int64_t>(tp.obs_cam.size());
    if ((P == 0 || M == 0) && with_points) return false;
    // AddCameraToCameraConstraints (gp.cc:167-210): one BATA pair per valid image pair whose images are known, in the view
    // graph's iteration order (the first one's scale is the constant one, gp.cc:484-489)
    std::vector<int32_t> pair_i, pair_j;
    std::vector<double> pair_dir;
    if (with_pairs) {
      for (const auto& [pid, pair] : view_graph.image_pairs) {
        if (!pair.is_valid) continue;
        auto i1 = images.find(pair.image_id1), i2 = images.find(pair.image_id2);
        if (i1 == images.end() || i2 == images.end()) continue;
        double t[3];
        detail::RotateInv(i2->second.CamFromWorld().rotation, pair.cam2_from_cam1.translation, t);  // gp.cc:195-197
        pair_i.push_back(fidx.of.at(i1->second.frame_id));
        pair_j.push_back(fidx.of.at(i2->second.frame_id));
        for (int j = 0; j < 3; ++j) pair_dir.push_back(-t[j]);
      }
      if (pair_i.empty()) return false;
    }
    std::vector<double> dir(3 * static_cast<size_t>(M)), cen(3 * static_cast<size_t>(N)), xyz(3 * static_cast<size_t>(P));
    std::vector<uint8_t> cal(static_cast<size_t>(M));
    // known rigs (gp.cc:318-350, RigBATAPairwiseDirectionError with the rig scale constant at 1, :470-478): images become
    // the cameras of the flat problem, each with its frame and the offset R_cam_from_world^T t_cam_from_rig
    // unknown cam_from_rig translations (gp.cc:354-368, RigUnknownBATAPairwiseDirectionError): a centre block per sensor
    std::unordered_map<image_t, int> img_of;
    std::vector<int32_t> image_frame, image_sensor;
    std::vector<double> image_offset, image_rot;
    std::vector<std::pair<rig_t, camera_t>> image_key;  // sensor of the images that need a block
    std::map<std::pair<rig_t, camera_t>, int> sensor_of;
    detail::ImageTable table(images);
    camera_t last_cam = 0;  // (shared cameras: the flag of the previous observation's camera is usually the one asked for)
    int last_cal = -1;
    for (int64_t k = 0; k < M; ++k) {
      const auto& im = table.At(tp.obs_image[k]);
      const auto cam_from_world = im.CamFromWorld();
      detail::RotateInv(cam_from_world.rotation, im.features_undist[tp.obs_feature[k]], &dir[3 * k]);  // gp.cc:294-296
      if (last_cal < 0 || im.camera_id != last_cam) {
        last_cam = im.camera_id;
        last_cal = cameras.at(im.camera_id).has_prior_focal_length ? 1 : 0;
      }
      cal[k] = static_cast<uint8_t>(last_cal);  // gp.cc:313-316
      if (!rigged) continue;
      auto it = img_of.find(tp.obs_image[k]);
      if (it == img_of.end()) {
        double cfr[7], off[3] = {0.0, 0.0, 0.0}, Rrw[9];
        const int state = detail::CamFromRigState(im, rigs, cfr);
        if (state == 2) return false;  // no cam_from_rig at all: the reference dereferences the empty optional (gp.cc:323)
        if (state == 0) detail::RotateInv(cam_from_world.rotation, cfr + 4, off);  // translation_rig, gp.cc:329-333
        const auto& rig_from_world = im.frame_ptr->RigFromWorld().rotation;
        for (int j = 0; j < 3; ++j) {  // row-major R_rig_from_world: column j = R e_j
          const double e[3] = {j == 0 ? 1.0 : 0.0, j == 1 ? 1.0 : 0.0, j == 2 ? 1.0 : 0.0};
          double col[3];
          detail::Rotate(rig_from_world, e, col);
          for (int i = 0; i < 3; ++i) Rrw[3 * i + j] = col[i];
        }
        it = img_of.emplace(tp.obs_image[k], static_cast<int>(image_frame.size())).first;
        image_frame.push_back(tp.obs_cam[k]);
        image_offset.insert(image_offset.end(), off, off + 3);
        image_rot.insert(image_rot.end(), Rrw, Rrw + 9);
        image_sensor.push_back(state == 1 ? 0 : -1);  // block index assigned below
        image_key.emplace_back(im.frame_ptr->RigId(), im.camera_id);
        if (state == 1) sensor_of.emplace(image_key.back(), -1);
      }
      tp.obs_cam[k] = it->second;
    }
    // block order = the order ParameterizeVariables draws their start values in (gp.cc:442-456): rigs in map order,
    // sensors in the rig's own (sorted) order, only those that own a parameter block
    std::vector<std::pair<rig_t, camera_t>> sensor_ids;
    for (auto& [rig_id, rig] : rigs)
      for (const auto& [sid, unused] : rig.NonRefSensors()) {
        (void)unused;
        if (sid.type != glomap::SensorType::CAMERA) continue;
        auto st = sensor_of.find({rig_id, static_cast<camera_t>(sid.id)});
        if (st == sensor_of.end()) continue;
        st->second = static_cast<int>(sensor_ids.size());
        sensor_ids.emplace_back(rig_id, static_cast<camera_t>(sid.id));
      }
    for (size_t i = 0; i < image_sensor.size(); ++i)
      if (image_sensor[i] == 0) image_sensor[i] = sensor_of.at(image_key[i]);
    std::vector<double> sensor_center(3 * sensor_ids.size(), 0.0);
    for (int n = 0; n < N; ++n) {  // c = -R^T t
      const auto& pose = frames.at(fidx.ids[n]).RigFromWorld();
      double c[3];
      detail::RotateInv(pose.rotation, pose.translation, c);
      for (int j = 0; j < 3; ++j) cen[3 * n + j] = -c[j];
    }
    for (int64_t p = 0; p < P; ++p)
      for (int j = 0; j < 3; ++j) xyz[3 * p + j] = tracks.at(tp.track_ids[p]).xyz[j];
    gsfm_gp_options o;
    gsfm_gp_options_default(&o);
    o.lm.max_num_iterations = options_.solver_options.max_num_iterations;
    o.lm.function_tolerance = options_.solver_options.function_tolerance;
    o.thres_loss_function = options_.thres_loss_function;
    o.generate_random_positions = options_.generate_random_positions;
    o.generate_random_points = options_.generate_random_points;
    o.generate_scales = options_.generate_scales;
    o.optimize_positions = options_.optimize_positions;
    o.optimize_points = options_.optimize_points;
    o.optimize_scales = options_.optimize_scales;
    o.min_num_view_per_track = 0;  // the raw-count rule was applied by PackTracks; zero-length tracks take their draw
    o.seed = options_.seed;
    o.constraint_type = static_cast<int32_t>(options_.constraint_type);
    // POINTS_AND_CAMERAS_BALANCED weighs the point losses by reweight * #pairs / tracks.size() — every track, not only the
    // packed ones (gp.cc:220-233); the library divides by the tracks it is given
    o.constraint_reweight_scale = options_.constraint_reweight_scale * (tracks.empty() ? 1.0 : static_cast<double>(P) / static_cast<double>(tracks.size()));
    // RandVector3d's three draws are constructor arguments (gp.cc:12-19): the compiler that builds GLOMAP decides their order,
    // so the drop-in follows the compiler that builds THIS header (g++: right to left, the first draw is z)
#if defined(__GNUC__) && !defined(__clang__)
    o.rand_vector_order = 1;
#else
    o.rand_vector_order = 0;
#endif
    gsfm_gp_problem pr{};
    pr.mem = GSFM_MEM_HOST;
    pr.num_cams = N;
    pr.num_pts = P;
    pr.num_obs = M;
    pr.pt_offset = tp.pt_offset.data();
    pr.obs_cam = tp.obs_cam.data();
    pr.obs_dir = dir.data();
    pr.obs_calibrated = cal.data();
    if (rigged) {
      pr.num_images = static_cast<int32_t>(image_frame.size());
      pr.image_frame = image_frame.data();
      pr.image_offset = image_offset.data();
      if (!sensor_ids.empty()) {
        pr.num_sensors = static_cast<int32_t>(sensor_ids.size());
        pr.image_sensor = image_sensor.data();
        pr.image_sensor_rot = image_rot.data();
        pr.sensor_center = sensor_center.data();
      }
    }
    // the library reads N and P entries from the two permutations: an image whose frame is missing from `frames` would
    // have grown the index (PackTracks / the pair loop add unknown frames) past what was drawn — refuse such a scene
    if (cam_draw_order.size() != static_cast<size_t>(N) || pt_draw_order.size() != static_cast<size_t>(P)) return false;
    pr.cam_draw_order = cam_draw_order.data();
    pr.pt_draw_order = pt_draw_order.data();
    if (with_pairs) {
      pr.num_pairs = static_cast<int64_t>(pair_i.size());
      pr.pair_i = pair_i.data();
      pr.pair_j = pair_j.data();
      pr.pair_dir = pair_dir.data();
    }
    gsfm_report& rep = report_;
    if (clk.Call([&] { return gsfm_gp_solve(ctx, &pr, &o, cen.data(), xyz.data(), &rep); }) != GSFM_OK) return false;
    clk.Note(rep);
    for (size_t k = 0; k < sensor_ids.size(); ++k) {  // ConvertResults: centre -> translation, t = -R c (gp.cc:576-582)
      auto& cfr = rigs.at(sensor_ids[k].first).SensorFromRig(glomap::sensor_t(glomap::SensorType::CAMERA, sensor_ids[k].second));
      double t[3];
      detail::Rotate(cfr.rotation, &sensor_center[3 * k], t);
      cfr.translation = decltype(cfr.translation)(-t[0], -t[1], -t[2]);
    }
    for (int n = 0; n < N; ++n) {  // ConvertResults: t = -R c (gp.cc:566-572)
      auto& fr = frames.at(fidx.ids[n]);
      auto pose = fr.RigFromWorld();
      double t[3];
      detail::Rotate(pose.rotation, &cen[3 * n], t);
      pose.translation = decltype(pose.translation)(-t[0], -t[1], -t[2]);
      fr.SetRigFromWorld(pose);
    }
    for (int64_t p = 0; p < P && with_points; ++p) {  // every packed track passed the raw-count rule (gp.cc:258)
      auto& tr = tracks.at(tp.track_ids[p]);
      tr.xyz = decltype(tr.xyz)(xyz[3 * p], xyz[3 * p + 1], xyz[3 * p + 2]);
      if (options_.optimize_points && options_.generate_random_points) tr.is_initialized = true;  // gp.cc:261-264
    }
    return true;
  }

  // Not part of the reference's interface: what the library reported for the last Solve (iterations, initial / final cost)
  const gsfm_report& LastReport() const { return report_; }

 private:
  glomap::GlobalPositionerOptions options_;
  gsfm_report report_{};
};

// ---------------------------------------------------------------------------------------------
// BundleAdjuster (bundle_adjustment.h:38-98)
// ---------------------------------------------------------------------------------------------
class BundleAdjuster {
 public:
  explicit BundleAdjuster(const glomap::BundleAdjusterOptions& options) : options_(options) {}
  glomap::BundleAdjusterOptions& GetOptions() { return options_; }

  bool Solve(std::unordered_map<rig_t, glomap::Rig>& rigs, std::unordered_map<camera_t, glomap::Camera>& cameras,
             std::unordered_map<frame_t, glomap::Frame>& frames, std::unordered_map<image_t, glomap::Image>& images,
             std::unordered_map<track_t, glomap::Track>& tracks) {
    // use_gpu == false: the reference class solves on the CPU (ba.cc:49-92); see GlobalPositioner::Solve above
    if (!options_.use_gpu) return glomap::BundleAdjuster(options_).Solve(rigs, cameras, frames, images, tracks);
    gsfm_ctx* ctx = Context(options_.gpu_index);
    detail::CallClock clk("BundleAdjuster::Solve");
    if (ctx == nullptr) return false;
    if (images.empty() || tracks.empty()) return false;  // ba.cc:17-24
    const bool rigged = !detail::AllTrivial(images);  // RigReprojErrorConstantRigCostFunctor, ba.cc:147-160
    // optimize_rig_poses: RigReprojErrorCostFunctor (ba.cc:161-179) — one sensor block per (rig, non-reference camera)
    const bool opt_rig = rigged && options_.optimize_rig_poses;
    std::map<std::pair<rig_t, camera_t>, int> sensor_of;
    std::vector<std::pair<rig_t, camera_t>> sensor_ids;
    std::vector<int32_t> image_sensor;
    std::vector<double> sensor_cfr;
    detail::FrameIndex fidx;
    std::vector<frame_t> frame_walk;
    for (auto& [fid, fr] : frames)
      if (fr.HasPose()) frame_walk.push_back(fid);  // the order ParameterizeVariables walks (ba.cc:253)
    const std::vector<int32_t> walk_index = fidx.AddSorted(frame_walk);  // dense indices by ascending id
    auto keep = [](const glomap::Image& im, uint32_t) { return im.frame_ptr != nullptr; };  // ba.cc:124-127: no IsRegistered test
    detail::TrackPack tp = detail::PackTracks(images, tracks, fidx, keep, static_cast<size_t>(options_.min_num_view_per_track));
    const int N = static_cast<int>(fidx.ids.size());
    const int64_t P = static_cast<int64_t>(tp.track_ids.size()), M = static_cast<int64_t>(tp.obs_cam.size());
    if (N == 0 || P == 0) return false;
    // The constant frame is the first frame IN THE PROBLEM in map order: the reference's counter only advances for
    // frames that own a parameter block (ba.cc:253-269), i.e. frames observed by a track that passed ba.cc:122.  A posed
    // but unobserved leading frame (GLOMAP gives every frame a pose, also those outside the largest component) must not
    // take the gauge away from the optimised cameras.
    int fixed = -1;
    {
      std::vector<uint8_t> in_problem(static_cast<size_t>(N), 0);
      for (int32_t n : tp.obs_cam) in_problem[n] = 1;
      for (size_t i = 0; i < walk_index.size() && fixed < 0; ++i)
        if (in_problem[walk_index[i]]) fixed = walk_index[i];
    }
    // intrinsics blocks = COLMAP cameras
    std::unordered_map<camera_t, int> intr_of;
    std::vector<camera_t> intr_ids;
    std::vector<int32_t> cam_intr(static_cast<size_t>(N), 0), intr_model;
    std::vector<double> intr;  // rows of GSFM_CAMERA_MAX_PARAMS_WIDE while packing; narrowed to 8 below unless a model needs 16
    auto intr_index = [&](camera_t cid) {
      auto it = intr_of.find(cid);
      if (it != intr_of.end()) return it->second;
      const int k = static_cast<int>(intr_ids.size());
      intr_of.emplace(cid, k);
      intr_ids.push_back(cid);
      const auto& cam = cameras.at(cid);
      intr_model.push_back(detail::ModelOf(cam));
      for (int j = 0; j < GSFM_CAMERA_MAX_PARAMS_WIDE; ++j) intr.push_back(j < static_cast<int>(cam.params.size()) ? cam.params[j] : 0.0);
      return k;
    };
    std::vector<double> xy(2 * static_cast<size_t>(M)), q(4 * static_cast<size_t>(N)), t(3 * static_cast<size_t>(N)),
        xyz(3 * static_cast<size_t>(P));
    // known rigs: images become the cameras of the flat problem — frame, constant cam_from_rig, intrinsics block each
    std::unordered_map<image_t, int> img_of;
    std::vector<int32_t> image_frame, image_intr;
    std::vector<double> image_cfr;
    detail::ImageTable table(images);
    std::vector<uint8_t> have_intr(static_cast<size_t>(N), 0);  // trivial frames: one image, one camera per frame
    for (int64_t k = 0; k < M; ++k) {
      const auto& im = table.At(tp.obs_image[k]);
      const auto& f = im.features[tp.obs_feature[k]];  // distorted pixels (ba.cc:139)
      xy[2 * k] = f[0];
      xy[2 * k + 1] = f[1];
      if (!rigged) {
        if (!have_intr[tp.obs_cam[k]]) {
          cam_intr[tp.obs_cam[k]] = intr_index(im.camera_id);
          have_intr[tp.obs_cam[k]] = 1;
        }
        continue;
      }
      auto it = img_of.find(tp.obs_image[k]);
      if (it == img_of.end()) {
        double cfr[7];
        if (!detail::KnownCamFromRig(im, rigs, cfr)) return false;
        it = img_of.emplace(tp.obs_image[k], static_cast<int>(image_frame.size())).first;
        image_frame.push_back(tp.obs_cam[k]);
        image_intr.push_back(intr_index(im.camera_id));
        image_cfr.insert(image_cfr.end(), cfr, cfr + 7);
        int sb = -1;
        if (opt_rig && !im.HasTrivialFrame()) {
          const std::pair<rig_t, camera_t> key(im.frame_ptr->RigId(), im.camera_id);
          auto st = sensor_of.find(key);
          if (st == sensor_of.end()) {
            st = sensor_of.emplace(key, static_cast<int>(sensor_ids.size())).first;
            sensor_ids.push_back(key);
            sensor_cfr.insert(sensor_cfr.end(), cfr, cfr + 7);
          }
          sb = st->second;
        }
        image_sensor.push_back(sb);
      }
      tp.obs_cam[k] = it->second;
    }
    int intr_stride = GSFM_CAMERA_MAX_PARAMS;
    for (int m : intr_model) {
      if (m < 0) return false;  // unsupported camera model
      if (detail::NeedsWideRows(m)) intr_stride = GSFM_CAMERA_MAX_PARAMS_WIDE;
    }
    if (intr_stride == GSFM_CAMERA_MAX_PARAMS) {  // the common case: 8-wide rows, the library's 8-wide unit
      for (size_t k = 0; k < intr_model.size(); ++k)
        for (int j = 0; j < GSFM_CAMERA_MAX_PARAMS; ++j) intr[GSFM_CAMERA_MAX_PARAMS * k + j] = intr[GSFM_CAMERA_MAX_PARAMS_WIDE * k + j];
      intr.resize(GSFM_CAMERA_MAX_PARAMS * intr_model.size());
    }
    for (int n = 0; n < N; ++n) {
      const auto& pose = frames.at(fidx.ids[n]).RigFromWorld();
      q[4 * n] = pose.rotation.w();
      q[4 * n + 1] = pose.rotation.x();
      q[4 * n + 2] = pose.rotation.y();
      q[4 * n + 3] = pose.rotation.z();
      for (int j = 0; j < 3; ++j) t[3 * n + j] = pose.translation[j];
    }
    for (int64_t p = 0; p < P; ++p)
      for (int j = 0; j < 3; ++j) xyz[3 * p + j] = tracks.at(tp.track_ids[p]).xyz[j];
    gsfm_ba_options o;
    gsfm_ba_options_default(&o);
    o.lm.max_num_iterations = options_.solver_options.max_num_iterations;
    o.lm.function_tolerance = options_.solver_options.function_tolerance;
    o.thres_loss_function = options_.thres_loss_function;
    o.optimize_rotations = options_.optimize_rotations;
    o.optimize_translation = options_.optimize_translation;
    o.optimize_intrinsics = options_.optimize_intrinsics;
    o.optimize_principal_point = options_.optimize_principal_point;
    o.optimize_points = options_.optimize_points;
    o.min_num_view_per_track = 1;  // the raw-count rule was applied by PackTracks
    o.optimize_rig_poses = opt_rig ? 1 : 0;
    gsfm_ba_problem pr{};
    pr.mem = GSFM_MEM_HOST;
    pr.num_cams = N;
    pr.num_intr = static_cast<int32_t>(intr_ids.size());
    pr.fixed_cam = fixed;
    pr.num_pts = P;
    pr.num_obs = M;
    pr.pt_offset = tp.pt_offset.data();
    pr.obs_cam = tp.obs_cam.data();
    pr.obs_xy = xy.data();
    pr.cam_intr = cam_intr.data();
    pr.intr_model = intr_model.data();
    pr.intr_stride = intr_stride;
    if (rigged) {
      pr.num_images = static_cast<int32_t>(image_frame.size());
      pr.image_frame = image_frame.data();
      pr.image_cam_from_rig = image_cfr.data();
      pr.image_intr = image_intr.data();
      if (opt_rig && !sensor_ids.empty()) {
        pr.num_sensors = static_cast<int32_t>(sensor_ids.size());
        pr.image_sensor = image_sensor.data();
        pr.sensor_cam_from_rig = sensor_cfr.data();
      }
    }
    gsfm_report& rep = report_;
    if (clk.Call([&] { return gsfm_ba_solve(ctx, &pr, &o, q.data(), t.data(), xyz.data(), intr.data(), &rep); }) != GSFM_OK) return false;
    clk.Note(rep);
    for (size_t k = 0; k < sensor_ids.size(); ++k) {  // the cam_from_rig blocks are the rigs' own storage (ba.cc:163-175)
      auto& cfr = rigs.at(sensor_ids[k].first).SensorFromRig(glomap::sensor_t(glomap::SensorType::CAMERA, sensor_ids[k].second));
      const double* v = &sensor_cfr[7 * k];
      cfr.rotation = decltype(cfr.rotation)(v[0], v[1], v[2], v[3]);
      cfr.translation = decltype(cfr.translation)(v[4], v[5], v[6]);
    }
    for (int n = 0; n < N; ++n) {  // parameter blocks are the containers' own storage in the reference (ba.cc:143-146)
      auto& fr = frames.at(fidx.ids[n]);
      auto pose = fr.RigFromWorld();
      pose.rotation = decltype(pose.rotation)(q[4 * n], q[4 * n + 1], q[4 * n + 2], q[4 * n + 3]);
      pose.translation = decltype(pose.translation)(t[3 * n], t[3 * n + 1], t[3 * n + 2]);
      fr.SetRigFromWorld(pose);
    }
    for (int64_t p = 0; p < P; ++p) {  // every packed track passed the raw-count rule (ba.cc:122)
      auto& tr = tracks.at(tp.track_ids[p]);
      tr.xyz = decltype(tr.xyz)(xyz[3 * p], xyz[3 * p + 1], xyz[3 * p + 2]);
    }
    for (size_t k = 0; k < intr_ids.size(); ++k) {
      auto& cam = cameras.at(intr_ids[k]);
      for (size_t j = 0; j < cam.params.size(); ++j) cam.params[j] = intr[static_cast<size_t>(intr_stride) * k + j];
    }
    return true;
  }

  // Not part of the reference's interface: what the library reported for the last Solve
  const gsfm_report& LastReport() const { return report_; }

 private:
  glomap::BundleAdjusterOptions options_;
  gsfm_report report_{};
};


// ---------------------------------------------------------------------------------------------
// TrackFilter (glomap/processors/track_filter.h:9-31), NormalizeReconstruction
// (glomap/processors/reconstruction_normalizer.h), RelPoseFilter::FilterRotations
// (glomap/processors/relpose_filter.h): same signatures and in-place semantics as the reference.
// ---------------------------------------------------------------------------------------------
namespace detail {

struct ViewPack {
  TrackPack tp;
  FrameIndex fidx;
  std::vector<double> undist, xy, q, t, xyz;
  std::vector<uint8_t> cal;
  gsfm_scene_view view{};
};

// every observation of every track, images looked up with .at() like the reference (track_filter.cc:18)
inline void PackView(const std::unordered_map<camera_t, glomap::Camera>* cameras,
                     const std::unordered_map<image_t, glomap::Image>& images,
                     std::unordered_map<track_t, glomap::Track>& tracks, bool with_undist, ViewPack& vp) {
  ImageTable table(images);
  std::vector<const glomap::Image*> node_image;  // one image of every packed frame (trivial rigs: THE image)
  for (auto& [tid, track] : tracks) {
    if (track.observations.empty()) continue;
    for (const auto& obs : track.observations) {
      const auto& im = table.At(obs.first);
      const int n = table.Node(obs.first, im, vp.fidx);
      if (static_cast<size_t>(n) >= node_image.size()) node_image.resize(static_cast<size_t>(n) + 1, nullptr);
      node_image[n] = &im;
      vp.tp.obs_cam.push_back(n);
      vp.tp.obs_image.push_back(obs.first);
      vp.tp.obs_feature.push_back(obs.second);
      if (with_undist) {
        const auto& f = im.features_undist[obs.second];
        vp.undist.insert(vp.undist.end(), {f[0], f[1], f[2]});
      }
    }
    vp.tp.track_ids.push_back(tid);
    vp.tp.pt_offset.push_back(static_cast<int64_t>(vp.tp.obs_cam.size()));
    for (int j = 0; j < 3; ++j) vp.xyz.push_back(track.xyz[j]);
  }
  const size_t N = vp.fidx.ids.size(), P = vp.tp.track_ids.size(), M = vp.tp.obs_cam.size();
  vp.q.resize(4 * N);
  vp.t.resize(3 * N);
  vp.cal.assign(N, 1);
  for (size_t n = 0; n < N; ++n) {  // per frame, not per observation
    const glomap::Image& im = *node_image[n];
    const auto& pose = im.frame_ptr->RigFromWorld();  // trivial rigs: cam_from_world
    vp.q[4 * n] = pose.rotation.w();
    vp.q[4 * n + 1] = pose.rotation.x();
    vp.q[4 * n + 2] = pose.rotation.y();
    vp.q[4 * n + 3] = pose.rotation.z();
    for (int j = 0; j < 3; ++j) vp.t[3 * n + j] = pose.translation[j];
    if (cameras) vp.cal[n] = cameras->at(im.camera_id).has_prior_focal_length ? 1 : 0;
  }
  vp.view.mem = GSFM_MEM_HOST;
  vp.view.num_cams = static_cast<int32_t>(N);
  vp.view.num_pts = static_cast<int64_t>(P);
  vp.view.num_obs = static_cast<int64_t>(M);
  vp.view.pt_offset = vp.tp.pt_offset.data();
  vp.view.obs_cam = vp.tp.obs_cam.data();
  vp.view.obs_undist = with_undist ? vp.undist.data() : nullptr;
  vp.view.cam_q = vp.q.data();
  vp.view.cam_t = vp.t.data();
  vp.view.pt_xyz = vp.xyz.data();
  vp.view.cam_calibrated = vp.cal.data();
}

// rewrite Track::observations from an observation keep mask; returns nothing (the counter comes from the C ABI)
inline void ApplyObsMask(const ViewPack& vp, const std::vector<uint8_t>& keep, std::unordered_map<track_t, glomap::Track>& tracks) {
  for (size_t p = 0; p < vp.tp.track_ids.size(); ++p) {
    auto& tr = tracks.at(vp.tp.track_ids[p]);
    std::vector<glomap::Observation> kept;
    for (int64_t k = vp.tp.pt_offset[p]; k < vp.tp.pt_offset[p + 1]; ++k)
      if (keep[k]) kept.emplace_back(vp.tp.obs_image[k], vp.tp.obs_feature[k]);
    if (kept.size() != tr.observations.size()) tr.observations = kept;
  }
}

}  // namespace detail

struct TrackFilter {
  // track_filter.cc:7-52 (normalised image coordinates; the pixel-space variant needs the camera model: use the C ABI)
  static int FilterTracksByReprojection(const glomap::ViewGraph& /*view_graph*/,
                                        const std::unordered_map<camera_t, glomap::Camera>& cameras,
                                        const std::unordered_map<image_t, glomap::Image>& images,
                                        std::unordered_map<track_t, glomap::Track>& tracks,
                                        double max_reprojection_error = 1e-2, bool in_normalized_image = true) {
    gsfm_ctx* ctx = Context();
    detail::CallClock clk("TrackFilter::FilterTracksByReprojection");
    if (ctx == nullptr || !in_normalized_image) return -1;
    detail::ViewPack vp;
    detail::PackView(&cameras, images, tracks, true, vp);
    if (vp.view.num_obs == 0) return 0;
    std::vector<uint8_t> keep(static_cast<size_t>(vp.view.num_obs));
    int64_t changed = 0;
    if (clk.Call([&] { return gsfm_filter_tracks_by_reprojection(ctx, &vp.view, max_reprojection_error, 1, keep.data(), &changed); }) != GSFM_OK) return -1;
    detail::ApplyObsMask(vp, keep, tracks);
    return static_cast<int>(changed);
  }

  // track_filter.cc:54-90
  static int FilterTracksByAngle(const glomap::ViewGraph& /*view_graph*/,
                                 const std::unordered_map<camera_t, glomap::Camera>& cameras,
                                 const std::unordered_map<image_t, glomap::Image>& images,
                                 std::unordered_map<track_t, glomap::Track>& tracks, double max_angle_error = 1.) {
    gsfm_ctx* ctx = Context();
    detail::CallClock clk("TrackFilter::FilterTracksByAngle");
    if (ctx == nullptr) return -1;
    detail::ViewPack vp;
    detail::PackView(&cameras, images, tracks, true, vp);
    if (vp.view.num_obs == 0) return 0;
    std::vector<uint8_t> keep(static_cast<size_t>(vp.view.num_obs));
    int64_t changed = 0;
    if (clk.Call([&] { return gsfm_filter_tracks_by_angle(ctx, &vp.view, max_angle_error, keep.data(), &changed); }) != GSFM_OK) return -1;
    detail::ApplyObsMask(vp, keep, tracks);
    return static_cast<int>(changed);
  }

  // track_filter.cc:92-127: tracks whose rays never span min_angle lose ALL observations
  static int FilterTrackTriangulationAngle(const glomap::ViewGraph& /*view_graph*/,
                                           const std::unordered_map<image_t, glomap::Image>& images,
                                           std::unordered_map<track_t, glomap::Track>& tracks, double min_angle = 1.) {
    gsfm_ctx* ctx = Context();
    detail::CallClock clk("TrackFilter::FilterTrackTriangulationAngle");
    if (ctx == nullptr) return -1;
    detail::ViewPack vp;
    detail::PackView(nullptr, images, tracks, false, vp);
    if (vp.view.num_pts == 0) return 0;
    std::vector<uint8_t> keep(static_cast<size_t>(vp.view.num_pts));
    int64_t removed = 0;
    if (clk.Call([&] { return gsfm_filter_tracks_triangulation_angle(ctx, &vp.view, min_angle, keep.data(), &removed); }) != GSFM_OK) return -1;
    for (size_t p = 0; p < vp.tp.track_ids.size(); ++p)
      if (!keep[p]) tracks.at(vp.tp.track_ids[p]).observations.clear();
    return static_cast<int>(removed);
  }
};

// reconstruction_normalizer.cc:5-85 (trivial rigs).  Returns {scale, tx, ty, tz} of X' = scale X + t.
inline std::array<double, 4> NormalizeReconstruction(std::unordered_map<rig_t, glomap::Rig>& /*rigs*/,
                                                     std::unordered_map<camera_t, glomap::Camera>& /*cameras*/,
                                                     std::unordered_map<frame_t, glomap::Frame>& frames,
                                                     std::unordered_map<image_t, glomap::Image>& images,
                                                     std::unordered_map<track_t, glomap::Track>& tracks,
                                                     bool fixed_scale = false, double extent = 10., double p0 = 0.1,
                                                     double p1 = 0.9) {
  std::array<double, 4> sim{1.0, 0.0, 0.0, 0.0};
  gsfm_ctx* ctx = Context();
  detail::CallClock clk("NormalizeReconstruction");
  if (ctx == nullptr) return sim;
  detail::FrameIndex fidx;
  for (auto& [fid, fr] : frames)
    if (fr.HasPose()) fidx.Add(fid);
  const size_t N = fidx.ids.size();
  if (N == 0) return sim;
  std::vector<uint8_t> reg(N, 0);
  for (auto& [iid, im] : images)  // the bounding box uses the registered IMAGES (reconstruction_normalizer.cc:24-30)
    if (im.IsRegistered() && fidx.of.count(im.frame_id)) reg[fidx.of.at(im.frame_id)] = 1;
  std::vector<double> q(4 * N), t(3 * N);
  for (size_t n = 0; n < N; ++n) {
    const auto& pose = frames.at(fidx.ids[n]).RigFromWorld();
    q[4 * n] = pose.rotation.w();
    q[4 * n + 1] = pose.rotation.x();
    q[4 * n + 2] = pose.rotation.y();
    q[4 * n + 3] = pose.rotation.z();
    for (int j = 0; j < 3; ++j) t[3 * n + j] = pose.translation[j];
  }
  std::vector<track_t> tids;
  std::vector<double> xyz;
  for (auto& [tid, tr] : tracks) {  // every track is transformed, initialised or not (:80-82)
    tids.push_back(tid);
    for (int j = 0; j < 3; ++j) xyz.push_back(tr.xyz[j]);
  }
  if (clk.Call([&] { return gsfm_normalize_reconstruction(ctx, GSFM_MEM_HOST, static_cast<int32_t>(N), reg.data(), q.data(), t.data(),
                                    static_cast<int64_t>(tids.size()), xyz.data(), fixed_scale, extent, p0, p1, sim.data()); }) != GSFM_OK)
    return sim;
  for (size_t n = 0; n < N; ++n) {
    auto& fr = frames.at(fidx.ids[n]);
    auto pose = fr.RigFromWorld();
    pose.translation = decltype(pose.translation)(t[3 * n], t[3 * n + 1], t[3 * n + 2]);
    fr.SetRigFromWorld(pose);
  }
  for (size_t p = 0; p < tids.size(); ++p) {
    auto& tr = tracks.at(tids[p]);
    tr.xyz = decltype(tr.xyz)(xyz[3 * p], xyz[3 * p + 1], xyz[3 * p + 2]);
  }
  return sim;
}

// image_undistorter.cc:7-46: features_undist[i] = camera.CamFromImg(features[i]).value_or(Zero).homogeneous().normalized() for every
// image whose bearings are missing — or for all of them with clean_points (what global_mapper.cc:237,263,307,325 asks for after
// the intrinsics moved).  One gsfm_undistort_features sweep over all features of all those images instead of a thread pool.
inline void UndistortImages(std::unordered_map<camera_t, glomap::Camera>& cameras,
                            std::unordered_map<image_t, glomap::Image>& images, bool clean_points = true) {
  gsfm_ctx* ctx = Context();
  detail::CallClock clk("UndistortImages");
  if (ctx == nullptr) return;
  std::vector<image_t> ids;
  for (auto& [iid, im] : images) {
    if (im.features_undist.size() == im.features.size() && !clean_points) continue;  // already undistorted (:13-15)
    ids.push_back(iid);
  }
  if (ids.empty()) return;
  std::unordered_map<camera_t, int32_t> row;
  std::vector<int32_t> model;
  std::vector<double> intr;
  int stride = GSFM_CAMERA_MAX_PARAMS;
  for (const image_t iid : ids) {
    const camera_t cid = images.at(iid).camera_id;
    if (row.count(cid)) continue;
    const int m = detail::ModelOf(cameras.at(cid));
    if (m < 0) return;  // a camera model the library does not know: leave the images to the caller
    if (detail::NeedsWideRows(m)) stride = GSFM_CAMERA_MAX_PARAMS_WIDE;
    row.emplace(cid, static_cast<int32_t>(model.size()));
    model.push_back(m);
  }
  intr.assign(static_cast<size_t>(stride) * model.size(), 0.0);
  for (const auto& [cid, k] : row) {
    const auto& cam = cameras.at(cid);
    for (size_t j = 0; j < cam.params.size() && j < static_cast<size_t>(stride); ++j) intr[static_cast<size_t>(stride) * k + j] = cam.params[j];
  }
  std::vector<double> xy;
  std::vector<int32_t> fi;
  for (const image_t iid : ids) {
    const auto& im = images.at(iid);
    const int32_t k = row.at(im.camera_id);
    for (const auto& f : im.features) {
      xy.push_back(f[0]);
      xy.push_back(f[1]);
      fi.push_back(k);
    }
  }
  std::vector<double> rays(3 * fi.size());
  if (clk.Call([&] { return gsfm_undistort_features(ctx, GSFM_MEM_HOST, static_cast<int64_t>(fi.size()), xy.data(), fi.data(), static_cast<int32_t>(model.size()),
                              model.data(), intr.data(), stride, rays.data()); }) != GSFM_OK)
    return;
  size_t o = 0;
  for (const image_t iid : ids) {
    auto& im = images.at(iid);
    im.features_undist.clear();
    im.features_undist.reserve(im.features.size());
    for (size_t i = 0; i < im.features.size(); ++i, ++o)
      im.features_undist.emplace_back(rays[3 * o], rays[3 * o + 1], rays[3 * o + 2]);
  }
}

struct RelPoseFilter {
  // relpose_filter.cc:7-33: invalidates pairs whose measured rotation disagrees with the estimated one
  static void FilterRotations(glomap::ViewGraph& view_graph, const std::unordered_map<image_t, glomap::Image>& images,
                              double max_angle = 5.0) {
    gsfm_ctx* ctx = Context();
    detail::CallClock clk("RelPoseFilter::FilterRotations");
    if (ctx == nullptr) return;
    detail::FrameIndex fidx;
    std::vector<int32_t> ei, ej;
    std::vector<double> eq, nq;
    std::vector<glomap::ImagePair*> pairs;
    auto node = [&](const glomap::Image& im) {
      const size_t before = fidx.ids.size();
      const int n = fidx.Add(im.frame_id);
      if (fidx.ids.size() != before) {
        const auto& r = im.frame_ptr->RigFromWorld().rotation;
        nq.insert(nq.end(), {r.w(), r.x(), r.y(), r.z()});
      }
      return n;
    };
    for (auto& [pid, pair] : view_graph.image_pairs) {
      if (!pair.is_valid) continue;
      const auto& i1 = images.at(pair.image_id1);
      const auto& i2 = images.at(pair.image_id2);
      if (!i1.IsRegistered() || !i2.IsRegistered()) continue;
      ei.push_back(node(i1));
      ej.push_back(node(i2));
      const auto& r = pair.cam2_from_cam1.rotation;
      eq.insert(eq.end(), {r.w(), r.x(), r.y(), r.z()});
      pairs.push_back(&pair);
    }
    if (pairs.empty()) return;
    std::vector<uint8_t> keep(pairs.size());
    int64_t ninv = 0;
    if (clk.Call([&] { return gsfm_filter_rotations(ctx, GSFM_MEM_HOST, static_cast<int32_t>(fidx.ids.size()), nq.data(), static_cast<int64_t>(pairs.size()),
                              ei.data(), ej.data(), eq.data(), max_angle, keep.data(), &ninv); }) != GSFM_OK)
      return;
    for (size_t e = 0; e < pairs.size(); ++e)
      if (!keep[e]) pairs[e]->is_valid = false;
  }
  // relpose_filter.cc:34-47 / :49-64: the two tests on a pair's inlier list (a count per pair: nothing for a device to do) —
  // here so that `RelPoseFilter` can be switched as a whole (global_mapper.cc:69-72 names them behind the relative-pose stage)
  static void FilterInlierNum(glomap::ViewGraph& view_graph, int min_inlier_num) {
    for (auto& [pid, pair] : view_graph.image_pairs)
      if (pair.is_valid && pair.inliers.size() < static_cast<size_t>(min_inlier_num)) pair.is_valid = false;  // (size_t comparison, as written there)
  }
  static void FilterInlierRatio(glomap::ViewGraph& view_graph, double min_inlier_ratio) {
    for (auto& [pid, pair] : view_graph.image_pairs)
      if (pair.is_valid && pair.inliers.size() / double(pair.matches.rows()) < min_inlier_ratio) pair.is_valid = false;
  }
};

// ---------------------------------------------------------------------------------------------
// TrackEngine (controllers/track_establishment.h:26-61) and ViewGraph::KeepLargestConnectedComponents
// (scene/view_graph.cc:56-97).  Track ids: the reference uses the union-find root, which depends on the
// iteration order of its hash maps; here a track is named by its smallest member (image_id << 32 | feature),
// observations are ascending (image_id, feature).  See DESIGN.md section 4.7.
// ---------------------------------------------------------------------------------------------
class TrackEngine {
 public:
  TrackEngine(const glomap::ViewGraph& view_graph, const std::unordered_map<image_t, glomap::Image>& images,
              const glomap::TrackEstablishmentOptions& options)
      : options_(options), view_graph_(view_graph), images_(images) {}

  size_t EstablishFullTracks(std::unordered_map<track_t, glomap::Track>& tracks) {
    tracks.clear();
    gsfm_ctx* ctx = Context();
    detail::CallClock clk("TrackEngine::EstablishFullTracks");
    if (ctx == nullptr) return 0;
    std::vector<image_t> ids = SortedImageIds();
    std::unordered_map<image_t, int32_t> dense;
    std::vector<int64_t> feat_offset{0};
    std::vector<double> xy;
    for (size_t n = 0; n < ids.size(); ++n) {
      dense.emplace(ids[n], static_cast<int32_t>(n));
      const auto& im = images_.at(ids[n]);
      for (const auto& f : im.features) xy.insert(xy.end(), {f[0], f[1]});
      feat_offset.push_back(static_cast<int64_t>(xy.size() / 2));
    }
    std::vector<int32_t> p1, p2;
    std::vector<int64_t> poff{0};
    std::vector<uint32_t> f1, f2;
    for (const auto& [pid, pair] : view_graph_.image_pairs) {
      if (!pair.is_valid) continue;  // track_establishment.cc:33
      p1.push_back(dense.at(pair.image_id1));
      p2.push_back(dense.at(pair.image_id2));
      for (const int idx : pair.inliers) {  // :41-47
        f1.push_back(static_cast<uint32_t>(pair.matches(idx, 0)));
        f2.push_back(static_cast<uint32_t>(pair.matches(idx, 1)));
      }
      poff.push_back(static_cast<int64_t>(f1.size()));
    }
    gsfm_match_graph g{};
    g.mem = GSFM_MEM_HOST;
    g.num_images = static_cast<int32_t>(ids.size());
    g.feat_offset = feat_offset.data();
    g.feat_xy = xy.data();
    g.num_pairs = static_cast<int64_t>(p1.size());
    g.pair_image1 = p1.data();
    g.pair_image2 = p2.data();
    g.pair_valid = nullptr;
    g.pair_offset = poff.data();
    g.match_feat1 = f1.data();
    g.match_feat2 = f2.data();
    const gsfm_track_options o = Options();
    int64_t nt = 0, no = 0, nd = 0;
    if (g.num_images == 0 || clk.Call([&] { return gsfm_tracks_establish(ctx, &g, &o, &nt, &no, &nd); }) != GSFM_OK) return 0;
    Fetch(ctx, GSFM_TRACKS_FULL, nt, no, ids, /*set_track_id=*/false, tracks);
    return tracks.size();
  }

  size_t FindTracksForProblem(const std::unordered_map<track_t, glomap::Track>& tracks_full,
                              std::unordered_map<track_t, glomap::Track>& tracks_selected) {
    tracks_selected.clear();
    gsfm_ctx* ctx = Context();
    detail::CallClock clk("TrackEngine::FindTracksForProblem");
    if (ctx == nullptr || tracks_full.empty()) return 0;
    std::vector<image_t> ids = SortedImageIds();
    std::unordered_map<image_t, int32_t> dense;
    std::vector<uint8_t> reg;
    for (size_t n = 0; n < ids.size(); ++n) {
      dense.emplace(ids[n], static_cast<int32_t>(n));
      reg.push_back(images_.at(ids[n]).IsRegistered() ? 1 : 0);  // :175-179
    }
    std::vector<int64_t> tid, off{0};
    std::vector<int32_t> oimg;
    std::vector<uint32_t> ofeat;
    for (const auto& [id, tr] : tracks_full) {
      tid.push_back(static_cast<int64_t>(id));
      for (const auto& [image_id, feature_id] : tr.observations) {
        auto it = dense.find(image_id);
        if (it == dense.end()) {  // an image the engine does not know counts as unregistered (:189)
          it = dense.emplace(image_id, static_cast<int32_t>(ids.size())).first;
          ids.push_back(image_id);
          reg.push_back(0);
        }
        oimg.push_back(it->second);
        ofeat.push_back(feature_id);
      }
      off.push_back(static_cast<int64_t>(oimg.size()));
    }
    gsfm_track_set full{};
    full.mem = GSFM_MEM_HOST;
    full.num_tracks = static_cast<int64_t>(tid.size());
    full.num_obs = static_cast<int64_t>(oimg.size());
    full.track_id = tid.data();
    full.track_offset = off.data();
    full.obs_image = oimg.data();
    full.obs_feature = ofeat.data();
    const gsfm_track_options o = Options();
    int64_t nt = 0, no = 0;
    if (clk.Call([&] { return gsfm_tracks_select(ctx, &full, static_cast<int32_t>(ids.size()), reg.data(), GSFM_MEM_HOST, &o, &nt, &no); }) != GSFM_OK) return 0;
    Fetch(ctx, GSFM_TRACKS_SELECTED, nt, no, ids, /*set_track_id=*/true, tracks_selected);
    return tracks_selected.size();
  }

 private:
  std::vector<image_t> SortedImageIds() const {
    std::vector<image_t> ids;
    ids.reserve(images_.size());
    for (const auto& [id, im] : images_) ids.push_back(id);
    std::sort(ids.begin(), ids.end());  // dense index order == image_id order (ids compare like the reference's)
    return ids;
  }
  gsfm_track_options Options() const {
    gsfm_track_options o;
    o.thres_inconsistency = options_.thres_inconsistency;
    o.min_num_tracks_per_view = options_.min_num_tracks_per_view;
    o.min_num_view_per_track = options_.min_num_view_per_track;
    o.max_num_view_per_track = options_.max_num_view_per_track;
    o.max_num_tracks = options_.max_num_tracks;
    return o;
  }
  static void Fetch(gsfm_ctx* ctx, int which, int64_t nt, int64_t no, const std::vector<image_t>& ids, bool set_track_id,
                    std::unordered_map<track_t, glomap::Track>& out) {
    std::vector<int64_t> tid(static_cast<size_t>(nt) + 1), off(static_cast<size_t>(nt) + 1);
    std::vector<int32_t> oimg(static_cast<size_t>(no) + 1);
    std::vector<uint32_t> ofeat(static_cast<size_t>(no) + 1);
    gsfm_track_set ts{};
    ts.mem = GSFM_MEM_HOST;
    ts.track_id = tid.data();
    ts.track_offset = off.data();
    ts.obs_image = oimg.data();
    ts.obs_feature = ofeat.data();
    if (gsfm_tracks_fetch(ctx, which, &ts) != GSFM_OK) return;
    out.reserve(static_cast<size_t>(nt));
    for (int64_t t = 0; t < nt; ++t) {
      track_t id = static_cast<track_t>(tid[t]);
      if (which == GSFM_TRACKS_FULL)  // dense image index -> image_id in the upper half
        id = (static_cast<track_t>(ids[static_cast<size_t>(tid[t] >> 32)]) << 32) | (static_cast<track_t>(tid[t]) & 0xFFFFFFFFull);
      glomap::Track& tr = out[id];  // discarded tracks stay as empty Track objects (:131)
      if (set_track_id) tr.track_id = id;  // :191
      tr.observations.reserve(static_cast<size_t>(off[t + 1] - off[t]));
      for (int64_t k = off[t]; k < off[t + 1]; ++k) tr.observations.emplace_back(ids[static_cast<size_t>(oimg[k])], ofeat[k]);
    }
  }

  const glomap::TrackEstablishmentOptions& options_;
  const glomap::ViewGraph& view_graph_;
  const std::unordered_map<image_t, glomap::Image>& images_;
};

// ViewGraph::KeepLargestConnectedComponents (view_graph.cc:56-97) as a free function over the reference's containers.
inline int KeepLargestConnectedComponents(glomap::ViewGraph& view_graph, std::unordered_map<frame_t, glomap::Frame>& frames,
                                          std::unordered_map<image_t, glomap::Image>& images) {
  gsfm_ctx* ctx = Context();
  if (ctx == nullptr) return 0;
  std::vector<frame_t> fids;
  for (const auto& [fid, fr] : frames) fids.push_back(fid);
  std::sort(fids.begin(), fids.end());  // ties between equally large components go to the smallest frame id
  std::unordered_map<frame_t, int32_t> dense;
  for (size_t n = 0; n < fids.size(); ++n) dense.emplace(fids[n], static_cast<int32_t>(n));
  std::vector<int32_t> nimg(fids.size(), 0);
  for (const auto& [id, im] : images) {
    auto it = dense.find(im.frame_id);
    if (it != dense.end()) nimg[static_cast<size_t>(it->second)]++;
  }
  std::vector<int32_t> ei, ej;
  std::vector<uint8_t> valid;
  std::vector<glomap::ImagePair*> pairs;
  for (auto& [pid, pair] : view_graph.image_pairs) {
    ei.push_back(dense.at(images.at(pair.image_id1).frame_id));
    ej.push_back(dense.at(images.at(pair.image_id2).frame_id));
    valid.push_back(pair.is_valid ? 1 : 0);
    pairs.push_back(&pair);
  }
  std::vector<uint8_t> reg(fids.size(), 0);
  int64_t count = 0;
  if (fids.empty() || gsfm_keep_largest_connected_component(ctx, GSFM_MEM_HOST, static_cast<int32_t>(fids.size()),
                                                            static_cast<int64_t>(pairs.size()), ei.data(), ej.data(), valid.data(),
                                                            nimg.data(), reg.data(), &count) != GSFM_OK ||
      count == 0)
    return 0;  // :70
  for (size_t n = 0; n < fids.size(); ++n) frames.at(fids[n]).is_registered = reg[n] != 0;  // :75-82
  for (size_t e = 0; e < pairs.size(); ++e)
    if (!valid[e]) pairs[e]->is_valid = false;  // :84-90
  return static_cast<int>(count);
}

}  // namespace gsfm_glomap
