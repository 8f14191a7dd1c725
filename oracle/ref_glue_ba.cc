// ref_glue_ba.cc — flat C entry point around the REFERENCE'S OWN bundle adjustment problem builder
// (glomap/estimators/bundle_adjustment.cc, compiled from /root/reference by `make -C oracle ref` against the stand-in types of
// oracle/ref_shim/, the RECORDING Ceres of oracle/ref_shim/ceres/ceres.h and the COLMAP cost-function / manifold stand-ins of
// oracle/ref_shim_ba/).  BundleAdjuster::Solve runs as written — Reset, AddPointToCameraConstraints,
// AddCamerasAndPointsToParameterGroups, ParameterizeVariables, ceres::Solve (records, evaluates the initial cost, does not
// minimise) — and this file reads the recorded problem back.  Test infrastructure: tests/test_oracle_ref_ba.py holds the problem
// oracle/ba.py poses to it.
#include <cstring>

// problem_ and loss_function_ are private members of BundleAdjuster; this translation unit (only) reads them
#define private public
#include "glomap/estimators/bundle_adjustment.h"
#undef private

#include "ref_glue_ba_scene.h"

using namespace glomap;

extern "C" {


// Ids are indices.  Quaternions (w, x, y, z).  sensors: the non-reference sensors (rig, camera id, cam_from_rig [7]).
// Outputs (caller-allocated): per residual block r < R: res_kind (0 trivial frame, 1 constant rig, 2 optimised rig), res_frame,
// res_track, res_camera, res_sensor (-1 unless kind 2); frame_flags [F] / camera_flags [K] / sensor_flags [S] / track_flags [P]:
// bit 0 in the problem, bit 1 rotation (or the whole block) constant, bit 2 translation constant, bit 3 quaternion manifold,
// bit 4 ordering group is 0 (points) rather than 1; camera_subset [K][16]: 1 where the subset manifold holds a coordinate (cam_params is [K][16] as well);
// frame_order [F]: the walk of the frames map; info [4] = {linear solver type, preconditioner type, 0, 0}.
// Returns R, -1 (capacity) or -2 (Solve returned false).
long ref_ba_build(int num_cameras, const int32_t* cam_model, const double* cam_params, int num_rigs, const int32_t* rig_ref_cam,
                  int num_sensors, const int32_t* sensor_rig, const int32_t* sensor_cam, const double* sensor_pose, int num_frames,
                  const int32_t* frame_rig, const uint8_t* frame_has_pose, const double* frame_q, const double* frame_trn, int num_images,
                  const int32_t* image_frame, const int32_t* image_cam, const uint8_t* image_present, const long* feat_offset,
                  const double* feat_xy, long num_tracks, const long* pt_offset, const int32_t* obs_image, const int32_t* obs_feature,
                  const double* pt_xyz, const ref_ba_options* o, long cap_res, int32_t* res_kind, int32_t* res_frame, long* res_track,
                  int32_t* res_camera, int32_t* res_sensor, uint8_t* frame_flags, uint8_t* camera_flags, uint8_t* camera_subset,
                  uint8_t* sensor_flags, uint8_t* track_flags, int32_t* frame_order, long* info, double* initial_cost_out) {
  ref_glue::BaScene sc;
  sc.Build(num_cameras, cam_model, cam_params, num_rigs, rig_ref_cam, num_sensors, sensor_rig, sensor_cam, sensor_pose, num_frames, frame_rig,
           frame_has_pose, frame_q, frame_trn, num_images, image_frame, image_cam, image_present, feat_offset, feat_xy, num_tracks, pt_offset,
           obs_image, obs_feature, pt_xyz);
  auto &rigs = sc.rigs;
  auto &cameras = sc.cameras;
  auto &frames = sc.frames;
  auto &images = sc.images;
  auto &tracks = sc.tracks;
  BundleAdjusterOptions opt;
  ref_glue::FillBaOptions(o, &opt);
  BundleAdjuster ba(opt);
  if (!ba.Solve(rigs, cameras, frames, images, tracks)) return -2;

  const ceres::Problem& problem = *ba.problem_;
  const auto& order = ba.options_.solver_options.linear_solver_ordering;
  auto flags = [&](double* rot_or_block, double* trn) -> uint8_t {
    uint8_t f = 0;
    if (problem.HasParameterBlock(rot_or_block)) f |= 1;
    if (problem.IsConstant(rot_or_block)) f |= 2;
    if (trn != nullptr && problem.IsConstant(trn)) f |= 4;
    const auto m = problem.manifolds().find(rot_or_block);
    if (m != problem.manifolds().end() && m->second.kind == 0) f |= 8;
    if (order) {
      const auto g = order->groups().find(rot_or_block);
      if (g != order->groups().end() && g->second == 0) f |= 16;
      if ((f & 1) && g == order->groups().end()) f |= 32;  // in the problem but in no group
    }
    return f;
  };
  int w = 0;
  for (auto& [id, fr] : frames) frame_order[w++] = static_cast<int32_t>(id);
  std::map<const double*, int> frame_of, camera_of, sensor_of;
  std::map<const double*, long> track_of;
  for (int f = 0; f < num_frames; ++f) {
    Frame& fr = frames.at(static_cast<frame_t>(f));
    frame_flags[f] = flags(fr.RigFromWorld().rotation.coeffs().data(), fr.RigFromWorld().translation.data());
    frame_of[fr.RigFromWorld().rotation.coeffs().data()] = f;
  }
  for (int k = 0; k < num_cameras; ++k) {
    Camera& c = cameras.at(static_cast<camera_t>(k));
    camera_flags[k] = flags(c.params.data(), nullptr);
    camera_of[c.params.data()] = k;
    for (int j = 0; j < ref_glue::kCamRow; ++j) camera_subset[ref_glue::kCamRow * k + j] = 0;
    const auto m = problem.manifolds().find(c.params.data());
    if (m != problem.manifolds().end() && m->second.kind == 1)
      for (int j : m->second.constant_idxs) camera_subset[ref_glue::kCamRow * k + j] = 1;
  }
  for (int s = 0; s < num_sensors; ++s) {
    Rigid3d& t = rigs.at(static_cast<rig_t>(sensor_rig[s])).SensorFromRig(sensor_t(SensorType::CAMERA, static_cast<uint32_t>(sensor_cam[s])));
    sensor_flags[s] = flags(t.rotation.coeffs().data(), t.translation.data());
    sensor_of[t.rotation.coeffs().data()] = s;
  }
  for (long p = 0; p < num_tracks; ++p) {
    Track& t = tracks.at(static_cast<track_t>(p));
    track_flags[p] = flags(t.xyz.data(), nullptr);
    track_of[t.xyz.data()] = p;
  }
  const auto& blocks = problem.residual_blocks();
  if (static_cast<long>(blocks.size()) > cap_res) return -1;
  long r = 0;
  for (const auto& b : blocks) {
    const auto* cost = static_cast<const colmap::RefShimReprojCost*>(b.cost.get());
    const int k = cost->kind(), o2 = k == 2 ? 2 : 0;
    res_kind[r] = k;
    res_sensor[r] = k == 2 ? sensor_of.at(b.params[0]) : -1;
    res_frame[r] = frame_of.at(b.params[o2]);
    res_track[r] = track_of.at(b.params[o2 + 2]);
    res_camera[r] = camera_of.at(b.params[o2 + 3]);
    if (b.loss != ba.loss_function_.get()) res_kind[r] |= 64;  // (every block carries the one Huber loss)
    ++r;
  }
  ceres::Solver::Summary summary;
  ceres::Solve(ba.options_.solver_options, ba.problem_.get(), &summary);
  *initial_cost_out = summary.initial_cost;
  info[0] = static_cast<long>(ba.options_.solver_options.linear_solver_type);
  info[1] = static_cast<long>(ba.options_.solver_options.preconditioner_type);
  info[2] = info[3] = 0;
  return r;
}

}  // extern "C"
