// ref_glue_ba_solve.cc — the REFERENCE'S bundle adjustment run TO ITS END POINT (test infrastructure).
// glomap/estimators/bundle_adjustment.cc, compiled unmodified from /root/reference against the SOLVING Ceres stand-in of
// oracle/ref_shim_solve/ceres/ceres.h and the COLMAP cost-function / manifold stand-ins of oracle/ref_shim_ba/ (restated from
// COLMAP's published functors: values + dual-number Jacobians): BundleAdjuster::Solve as written — residual blocks, the Huber
// loss, the constant frame its hash map yields, quaternion / subset manifolds, ordering groups, ceres::Solve.
// tests/test_reference_code_gpu.py holds the HIP path's FINAL poses to it.
#include <cstring>

#define private public
#include "glomap/estimators/bundle_adjustment.h"
#undef private

#include "ref_glue_ba_scene.h"

using namespace glomap;

extern "C" {
// In: as ref_ba_build (oracle/ref_glue_ba.cc).  Out (caller-allocated): frame_q_out [F][4] (w, x, y, z) / frame_trn_out [F][3],
// xyz_out [P][3], cam_params_out [K][16], sensor_pose_out [S][7], frame_order [F] (the walk of the frames map), frame_const [F]
// (1: the pose blocks of that frame were constant), summary_out [8] (as ref_gp_solve), trace_out [cap_trace][7].
// Returns the number of trace rows, -2 when Solve returned false.
long ref_ba_solve(int num_cameras, const int32_t* cam_model, const double* cam_params, int num_rigs, const int32_t* rig_ref_cam,
                  int num_sensors, const int32_t* sensor_rig, const int32_t* sensor_cam, const double* sensor_pose, int num_frames,
                  const int32_t* frame_rig, const uint8_t* frame_has_pose, const double* frame_q, const double* frame_trn, int num_images,
                  const int32_t* image_frame, const int32_t* image_cam, const uint8_t* image_present, const long* feat_offset,
                  const double* feat_xy, long num_tracks, const long* pt_offset, const int32_t* obs_image, const int32_t* obs_feature,
                  const double* pt_xyz, const ref_ba_options* o, int max_num_iterations, double* frame_q_out, double* frame_trn_out,
                  double* xyz_out, double* cam_params_out, double* sensor_pose_out, int32_t* frame_order, uint8_t* frame_const,
                  double* summary_out, long cap_trace, double* trace_out) {
  ref_glue::BaScene sc;
  sc.Build(num_cameras, cam_model, cam_params, num_rigs, rig_ref_cam, num_sensors, sensor_rig, sensor_cam, sensor_pose, num_frames, frame_rig,
           frame_has_pose, frame_q, frame_trn, num_images, image_frame, image_cam, image_present, feat_offset, feat_xy, num_tracks, pt_offset,
           obs_image, obs_feature, pt_xyz);
  BundleAdjusterOptions opt;
  ref_glue::FillBaOptions(o, &opt);
  if (max_num_iterations > 0) opt.solver_options.max_num_iterations = max_num_iterations;
  BundleAdjuster ba(opt);
  ceres::LastSummary() = ceres::Solver::Summary();
  const bool ok = ba.Solve(sc.rigs, sc.cameras, sc.frames, sc.images, sc.tracks);
  const ceres::Solver::Summary& s = ceres::LastSummary();
  int w = 0;
  for (auto& [id, fr] : sc.frames) frame_order[w++] = static_cast<int32_t>(id);
  for (int f = 0; f < num_frames; ++f) {
    Frame& fr = sc.frames.at(static_cast<frame_t>(f));
    frame_const[f] = 0;
    if (!fr.HasPose()) continue;
    const Rigid3d& T = fr.RigFromWorld();
    frame_q_out[4 * f] = T.rotation.w();
    frame_q_out[4 * f + 1] = T.rotation.x();
    frame_q_out[4 * f + 2] = T.rotation.y();
    frame_q_out[4 * f + 3] = T.rotation.z();
    for (int j = 0; j < 3; ++j) frame_trn_out[3 * f + j] = T.translation(j);
    frame_const[f] = ba.problem_->IsConstant(fr.RigFromWorld().rotation.coeffs().data()) && ba.problem_->IsConstant(fr.RigFromWorld().translation.data());
  }
  for (long p = 0; p < num_tracks; ++p)
    for (int j = 0; j < 3; ++j) xyz_out[3 * p + j] = sc.tracks.at(static_cast<track_t>(p)).xyz(j);
  for (int k = 0; k < num_cameras; ++k) {
    const Camera& c = sc.cameras.at(static_cast<camera_t>(k));
    for (int j = 0; j < ref_glue::kCamRow; ++j) cam_params_out[ref_glue::kCamRow * k + j] = j < (int)c.params.size() ? c.params[j] : 0.0;
  }
  for (int si = 0; si < num_sensors; ++si) {
    Rigid3d& t = sc.rigs.at(static_cast<rig_t>(sensor_rig[si])).SensorFromRig(sensor_t(SensorType::CAMERA, static_cast<uint32_t>(sensor_cam[si])));
    const double v[7] = {t.rotation.w(), t.rotation.x(), t.rotation.y(), t.rotation.z(), t.translation(0), t.translation(1), t.translation(2)};
    std::memcpy(sensor_pose_out + 7 * si, v, sizeof v);
  }
  const double sm[8] = {s.initial_cost, s.final_cost, (double)s.num_iterations, (double)s.num_successful_steps, (double)s.num_line_search_steps,
                        (double)s.num_steps_shortened, (double)s.termination_type, s.is_constrained ? 1.0 : 0.0};
  std::memcpy(summary_out, sm, sizeof sm);
  const long rows = (long)(s.trace.size() / 7);
  std::memcpy(trace_out, s.trace.data(), sizeof(double) * 7 * (size_t)std::min(rows, cap_trace));
  return ok ? rows : -2;
}
}  // extern "C"
