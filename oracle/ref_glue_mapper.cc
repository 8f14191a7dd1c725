// ref_glue_mapper.cc — the reference's top-level controller glomap/controllers/global_mapper.cc (GlobalMapper::Solve: rotation
// averaging twice with the rotation filter and the largest component between, track establishment, global positioning, the
// three track filters, the normaliser, the bundle adjustment rounds with their filters; global_mapper.cc:85-278) run on ONE set
// of containers built from flat arrays:
//   which = 0   as the reference builds it: its controller on its own estimators and processors — all reference code, compiled
//               unmodified from /root/reference, on the CPU (RotationEstimator on the Eigen / LAD stand-ins of ref_shim_ra/,
//               GlobalPositioner and BundleAdjuster on the SOLVING Ceres stand-in of ref_shim_solve/)
//   which = 1   THE DROP-IN: the same controller source with RotationEstimator / GlobalPositioner / BundleAdjuster /
//               UndistortImages switched to include/gsfm_glomap_adapter.hpp — libgsfm on the GPU (ref_dropin_mapper_on_gsfm.cc)
//   which = 2   as 1, and TrackFilter / NormalizeReconstruction / RelPoseFilter on libgsfm as well
//   which = 3   as 2, and TrackEngine (track establishment and selection) too — the same tracks under other ids, i.e. another random
//               start of global positioning: an equally valid run, not comparable with which = 0 entry by entry
// The stages outside SURVEY section 8 (preprocessing, view-graph calibration, relative-pose estimation, retriangulation, pruning)
// are skipped by GlobalMapperOptions::skip_* — the reference's own switches — and abort if reached (ref_glue_mapper_stubs.cc).
// Part of oracle/_ref/libref_dropin_mapper.so (`make -C oracle ref_mapper`; links libgsfm.so).  tests/test_dropin_reference_mapper.py.
// Trivial frames: image i = frame i = rig i; camera image_cam[i].
#include <algorithm>
#include <cmath>
#include <cstring>
#include <sstream>

#include "glomap/controllers/global_mapper.h"

#include <colmap/geometry/pose.h>

#include <chrono>
#include <cstdio>
#include <string>

#include "gsfm_glomap_adapter.hpp"  // (gsfm_glomap::AdapterTimings: where the drop-in builds' wall time went)

namespace glomap {
#define REF_DECLARE_MAPPER(NAME)                                                                                                    \
  class NAME {                                                                                                                      \
   public:                                                                                                                          \
    NAME(const GlobalMapperOptions& options) : options_(options) {}                                                                 \
    bool Solve(const colmap::Database& database, ViewGraph& view_graph, std::unordered_map<rig_t, Rig>& rigs,                       \
               std::unordered_map<camera_t, Camera>& cameras, std::unordered_map<frame_t, Frame>& frames,                           \
               std::unordered_map<image_t, Image>& images, std::unordered_map<track_t, Track>& tracks);                             \
                                                                                                                                    \
   private:                                                                                                                         \
    const GlobalMapperOptions options_;                                                                                             \
  };
// (the declaration of glomap/controllers/global_mapper.h:43-59 under the two names ref_dropin_mapper_on_gsfm.cc compiles it as)
REF_DECLARE_MAPPER(GlobalMapperOnGsfm)
REF_DECLARE_MAPPER(GlobalMapperOnGsfmAll)
REF_DECLARE_MAPPER(GlobalMapperOnGsfmTracks)
}  // namespace glomap

using namespace glomap;

namespace {
struct QuietCout {
  std::streambuf* old;
  std::ostringstream sink;
  QuietCout() : old(std::cout.rdbuf(sink.rdbuf())) {}
  ~QuietCout() { std::cout.rdbuf(old); }
};
}  // namespace

extern "C" {

struct ref_mapper_options {
  int num_iteration_bundle_adjustment;  // GlobalMapperOptions default 3
  int skip_rotation_averaging, skip_track_establishment, skip_global_positioning, skip_bundle_adjustment;
  int min_num_view_per_track;           // opt_track / opt_gp / opt_ba (their defaults: 3)
  int optimize_intrinsics;              // opt_ba.optimize_intrinsics
  unsigned gp_seed;                     // opt_gp.seed
  double max_angle_error, max_reprojection_error, min_triangulation_angle, max_rotation_error;  // InlierThresholdOptions; <= 0: default
  double thres_inconsistency;           // opt_track.thres_inconsistency; <= 0: default
};

// Inputs: cameras [K] (model id, 12 padded parameters, prior flag); images [N] with pixel features (feat_offset [N + 1], feat_xy);
// frame poses to start from (frame_q_in wxyz, frame_t_in; used when rotation averaging is skipped, otherwise overwritten);
// pairs [E]: cam2_from_cam1 (q wxyz, t), weight, validity, matches (match_offset [E + 1], feature indices in image 1 / 2; every
// match is an inlier).  Outputs: frame_q_out [N][4], frame_t_out [N][3], frame_registered_out [N], cam_params_out [K][12],
// pair_valid_out [E], counts_out [4] = {tracks, observations, initialised tracks, wall time of Solve in microseconds}; the tracks sorted by id, up to cap_tracks:
// track_id_out, track_len_out, track_xyz_out [.][3].  Returns GlobalMapper::Solve's bool (1 / 0), -1 on a bad `which`.
int ref_mapper_solve(int which, int num_cams, const int32_t* cam_model, const double* cam_params, const uint8_t* cam_has_prior,
                     int num_images, const int32_t* image_cam, const long* feat_offset, const double* feat_xy, const double* frame_q_in,
                     const double* frame_t_in, long num_pairs, const int32_t* pair_i, const int32_t* pair_j, const double* pair_q,
                     const double* pair_t, const double* pair_weight, const uint8_t* pair_valid, const long* match_offset,
                     const int32_t* match_f1, const int32_t* match_f2, const ref_mapper_options* o, double* frame_q_out,
                     double* frame_t_out, uint8_t* frame_registered_out, double* cam_params_out, uint8_t* pair_valid_out,
                     long* counts_out, long cap_tracks, uint64_t* track_id_out, int32_t* track_len_out, double* track_xyz_out) {
  QuietCout quiet;
  std::unordered_map<rig_t, Rig> rigs;
  std::unordered_map<camera_t, Camera> cameras;
  std::unordered_map<frame_t, Frame> frames;
  std::unordered_map<image_t, Image> images;
  std::unordered_map<track_t, Track> tracks;
  ViewGraph vg;
  static const int kNumParams[12] = {3, 4, 4, 5, 8, 8, 12, 5, 4, 5, 12, 16};
  for (int k = 0; k < num_cams; ++k) {
    Camera& c = cameras[static_cast<camera_t>(k)];
    c.model_id = static_cast<colmap::CameraModelId>(cam_model[k]);
    c.has_prior_focal_length = cam_has_prior ? cam_has_prior[k] != 0 : true;
    c.params.assign(cam_params + 12 * k, cam_params + 12 * k + std::min(12, kNumParams[cam_model[k]]));
  }
  rigs.reserve(static_cast<size_t>(num_images));
  frames.reserve(static_cast<size_t>(num_images));
  images.reserve(static_cast<size_t>(num_images));
  for (int n = 0; n < num_images; ++n) {
    Rig& rig = rigs[static_cast<rig_t>(n)];
    rig.ref = sensor_t(SensorType::CAMERA, static_cast<uint32_t>(image_cam[n]));
    rig.SetRigId(static_cast<rig_t>(n));
    Frame& fr = frames[static_cast<frame_t>(n)];
    fr.SetFrameId(static_cast<frame_t>(n));
    fr.rig_id = static_cast<rig_t>(n);
    fr.is_registered = true;
    fr.SetRigFromWorld(Rigid3d(Eigen::Quaterniond(frame_q_in[4 * n], frame_q_in[4 * n + 1], frame_q_in[4 * n + 2], frame_q_in[4 * n + 3]),
                               Eigen::Vector3d(frame_t_in[3 * n], frame_t_in[3 * n + 1], frame_t_in[3 * n + 2])));
  }
  for (auto& [id, fr] : frames) fr.rig_ptr = &rigs.at(fr.rig_id);
  for (int n = 0; n < num_images; ++n) {
    Image& im = images[static_cast<image_t>(n)];
    im.image_id = static_cast<image_t>(n);
    im.camera_id = static_cast<camera_t>(image_cam[n]);
    im.frame_id = static_cast<frame_t>(n);
    im.frame_ptr = &frames.at(im.frame_id);
    im.frame_ptr->data_ids.insert(data_t(sensor_t(SensorType::CAMERA, im.camera_id), im.image_id));
    im.features.reserve(static_cast<size_t>(feat_offset[n + 1] - feat_offset[n]));
    for (long f = feat_offset[n]; f < feat_offset[n + 1]; ++f) im.features.emplace_back(feat_xy[2 * f], feat_xy[2 * f + 1]);
  }
  for (long e = 0; e < num_pairs; ++e) {
    ImagePair p;
    p.image_id1 = static_cast<image_t>(pair_i[e]);
    p.image_id2 = static_cast<image_t>(pair_j[e]);
    p.is_valid = pair_valid ? pair_valid[e] != 0 : true;
    p.weight = pair_weight ? pair_weight[e] : -1;
    p.cam2_from_cam1 = Rigid3d(Eigen::Quaterniond(pair_q[4 * e], pair_q[4 * e + 1], pair_q[4 * e + 2], pair_q[4 * e + 3]),
                               Eigen::Vector3d(pair_t[3 * e], pair_t[3 * e + 1], pair_t[3 * e + 2]));
    const long m0 = match_offset[e], m1 = match_offset[e + 1];
    p.matches.d.reserve(static_cast<size_t>(2 * (m1 - m0)));
    for (long m = m0; m < m1; ++m) {
      p.matches.d.push_back(match_f1[m]);
      p.matches.d.push_back(match_f2[m]);
      p.inliers.push_back(static_cast<int>(m - m0));
    }
    vg.image_pairs.emplace(colmap::ImagePairToPairId(p.image_id1, p.image_id2), std::move(p));
  }

  GlobalMapperOptions opt;
  opt.skip_preprocessing = true;
  opt.skip_view_graph_calibration = true;
  opt.skip_relative_pose_estimation = true;
  opt.skip_retriangulation = true;
  opt.skip_pruning = true;
  opt.skip_rotation_averaging = o->skip_rotation_averaging != 0;
  opt.skip_track_establishment = o->skip_track_establishment != 0;
  opt.skip_global_positioning = o->skip_global_positioning != 0;
  opt.skip_bundle_adjustment = o->skip_bundle_adjustment != 0;
  if (o->num_iteration_bundle_adjustment > 0) opt.num_iteration_bundle_adjustment = o->num_iteration_bundle_adjustment;
  if (o->min_num_view_per_track > 0) {
    opt.opt_track.min_num_view_per_track = o->min_num_view_per_track;
    opt.opt_gp.min_num_view_per_track = o->min_num_view_per_track;
    opt.opt_ba.min_num_view_per_track = o->min_num_view_per_track;
  }
  opt.opt_ba.optimize_intrinsics = o->optimize_intrinsics != 0;
  opt.opt_gp.seed = o->gp_seed;
  if (o->max_angle_error > 0) opt.inlier_thresholds.max_angle_error = o->max_angle_error;
  if (o->max_reprojection_error > 0) opt.inlier_thresholds.max_reprojection_error = o->max_reprojection_error;
  if (o->min_triangulation_angle > 0) opt.inlier_thresholds.min_triangulation_angle = o->min_triangulation_angle;
  if (o->max_rotation_error > 0) opt.inlier_thresholds.max_rotation_error = o->max_rotation_error;
  if (o->thres_inconsistency > 0) opt.opt_track.thres_inconsistency = o->thres_inconsistency;
  // use_gpu: what it means in the reference — which = 0 solves on the CPU (the Ceres stand-in), the drop-in on the device
  opt.opt_gp.use_gpu = which != 0;
  opt.opt_ba.use_gpu = which != 0;

  const colmap::Database database;
  bool ok = false;
  gsfm_glomap::ResetTimings();
  const auto wall0 = std::chrono::steady_clock::now();
  if (which == 0) ok = GlobalMapper(opt).Solve(database, vg, rigs, cameras, frames, images, tracks);
  else if (which == 1) ok = GlobalMapperOnGsfm(opt).Solve(database, vg, rigs, cameras, frames, images, tracks);
  else if (which == 2) ok = GlobalMapperOnGsfmAll(opt).Solve(database, vg, rigs, cameras, frames, images, tracks);
  else if (which == 3) ok = GlobalMapperOnGsfmTracks(opt).Solve(database, vg, rigs, cameras, frames, images, tracks);
  else return -1;
  const double wall = std::chrono::duration<double>(std::chrono::steady_clock::now() - wall0).count();

  for (int n = 0; n < num_images; ++n) {
    const Frame& fr = frames.at(static_cast<frame_t>(n));
    const Rigid3d& T = fr.RigFromWorld();
    frame_q_out[4 * n] = T.rotation.w();
    frame_q_out[4 * n + 1] = T.rotation.x();
    frame_q_out[4 * n + 2] = T.rotation.y();
    frame_q_out[4 * n + 3] = T.rotation.z();
    for (int j = 0; j < 3; ++j) frame_t_out[3 * n + j] = T.translation(j);
    frame_registered_out[n] = fr.is_registered ? 1 : 0;
  }
  for (int k = 0; k < num_cams; ++k) {
    const Camera& c = cameras.at(static_cast<camera_t>(k));
    for (int j = 0; j < 12; ++j) cam_params_out[12 * k + j] = j < static_cast<int>(c.params.size()) ? c.params[j] : 0.0;
  }
  for (long e = 0; e < num_pairs; ++e)
    pair_valid_out[e] = vg.image_pairs.at(colmap::ImagePairToPairId(static_cast<image_t>(pair_i[e]), static_cast<image_t>(pair_j[e]))).is_valid ? 1 : 0;
  std::vector<track_t> ids;
  ids.reserve(tracks.size());
  long nobs = 0, ninit = 0;
  for (auto& [id, tr] : tracks) {
    ids.push_back(id);
    nobs += static_cast<long>(tr.observations.size());
    ninit += tr.is_initialized ? 1 : 0;
  }
  std::sort(ids.begin(), ids.end());
  counts_out[0] = static_cast<long>(tracks.size());
  counts_out[1] = nobs;
  counts_out[2] = ninit;
  counts_out[3] = static_cast<long>(wall * 1e6);  // wall time of GlobalMapper::Solve, microseconds
  for (long t = 0; t < static_cast<long>(ids.size()) && t < cap_tracks; ++t) {
    const Track& tr = tracks.at(ids[t]);
    track_id_out[t] = ids[t];
    track_len_out[t] = static_cast<int32_t>(tr.observations.size());
    for (int j = 0; j < 3; ++j) track_xyz_out[3 * t + j] = tr.xyz(j);
  }
  return ok ? 1 : 0;
}


// Where the wall time of the last drop-in run went inside the adapter (gsfm_glomap::AdapterTimings): one line per entry point,
// "name calls pack call unpack iterations linear_iterations" (seconds; the counts are what the estimators reported).  Returns the number of bytes written (0 after a which = 0 run: no adapter involved).
long ref_mapper_timings(char* buf, long cap) {
  std::string out;
  char line[256];
  for (const auto& [name, t] : gsfm_glomap::AdapterTimings()) {
    std::snprintf(line, sizeof line, "%s %ld %.6f %.6f %.6f %ld %ld\n", name.c_str(), t.calls, t.pack, t.call, t.unpack, t.iterations, t.linear_iterations);
    out += line;
  }
  const long n = std::min<long>(cap - 1, static_cast<long>(out.size()));
  if (n < 0) return 0;
  std::memcpy(buf, out.data(), static_cast<size_t>(n));
  buf[n] = 0;
  return n;
}

}  // extern "C"
