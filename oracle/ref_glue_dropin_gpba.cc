// ref_glue_dropin_gpba.cc — include/gsfm_glomap_adapter.hpp's GlobalPositioner and BundleAdjuster run on the SAME containers
// (same flat arrays, same builders: ref_glue_gp_scene.h / ref_glue_ba_scene.h, the reference's real estimator headers) that
// oracle/ref_glue_gp.cc / ref_glue_ba.cc hand to the reference's own GlobalPositioner::Solve / BundleAdjuster::Solve on the
// recording Ceres.  Part of oracle/_ref/libref_dropin_ra.so (links libgsfm.so).  tests/test_dropin_reference_controller.py: the
// cost libgsfm reports for the start — through the adapter's packing of the reference's containers: its walk of the hash maps,
// its draw orders, its constant frame — equals the cost the reference's builder poses, and the results land in the containers
// the way the reference's ConvertResults / in-place parameter blocks leave them.
#include <cstring>

#include "ref_glue_ba_scene.h"
#include "ref_glue_gp_scene.h"

#include "gsfm_glomap_adapter.hpp"

using namespace glomap;

extern "C" {

// out_report [4] = {initial cost, final cost, iterations, 0}; out_center [N][3] = camera centres after the solve (from the
// translations the adapter wrote), out_xyz [P][3], out_initialized [P].  Returns the adapter's bool.
int ref_gp_adapter_solve(int num_cams, const double* cam_q, const double* cam_t_in, const uint8_t* cam_calibrated, const uint8_t* cam_registered,
                         long num_pts, const long* pt_offset, const int32_t* obs_cam, const double* obs_undist, const double* pt_xyz_in,
                         const uint8_t* pt_initialized, long num_pairs, const int32_t* pair_i, const int32_t* pair_j, const uint8_t* pair_valid,
                         const double* pair_t, const ref_gp_options* o, double* out_report, double* out_center, double* out_xyz,
                         uint8_t* out_initialized) {
  ref_glue::GpScene sc;
  sc.Build(num_cams, cam_q, cam_t_in, cam_calibrated, cam_registered, num_pts, pt_offset, obs_cam, obs_undist, pt_xyz_in, pt_initialized,
           num_pairs, pair_i, pair_j, pair_valid, pair_t);
  GlobalPositionerOptions opt;
  ref_glue::FillGpOptions(o, &opt);
  opt.use_gpu = true;
  gsfm_glomap::GlobalPositioner gp(opt);
  const bool ok = gp.Solve(sc.vg, sc.rigs, sc.cameras, sc.frames, sc.images, sc.tracks);
  out_report[0] = gp.LastReport().initial_cost;
  out_report[1] = gp.LastReport().final_cost;
  out_report[2] = gp.LastReport().iterations;
  out_report[3] = 0.0;
  for (int n = 0; n < num_cams; ++n) {
    const Frame& f = sc.frames.at(static_cast<frame_t>(n));
    const Eigen::Vector3d c = f.rig_from_world.rotation.inverse() * -f.rig_from_world.translation;
    for (int j = 0; j < 3; ++j) out_center[3 * n + j] = c(j);
  }
  for (long p = 0; p < num_pts; ++p) {
    const Track& t = sc.tracks.at(static_cast<track_t>(p));
    for (int j = 0; j < 3; ++j) out_xyz[3 * p + j] = t.xyz(j);
    out_initialized[p] = t.is_initialized ? 1 : 0;
  }
  return ok ? 1 : 0;
}

// out_report as above; out_frame_q [F][4] (w, x, y, z), out_frame_t [F][3], out_cam_params [K][16], out_xyz [P][3].
int ref_ba_adapter_solve(int num_cameras, const int32_t* cam_model, const double* cam_params, int num_rigs, const int32_t* rig_ref_cam,
                         int num_sensors, const int32_t* sensor_rig, const int32_t* sensor_cam, const double* sensor_pose, int num_frames,
                         const int32_t* frame_rig, const uint8_t* frame_has_pose, const double* frame_q, const double* frame_trn, int num_images,
                         const int32_t* image_frame, const int32_t* image_cam, const uint8_t* image_present, const long* feat_offset,
                         const double* feat_xy, long num_tracks, const long* pt_offset, const int32_t* obs_image, const int32_t* obs_feature,
                         const double* pt_xyz, const ref_ba_options* o, double* out_report, double* out_frame_q, double* out_frame_t,
                         double* out_cam_params, double* out_xyz) {
  ref_glue::BaScene sc;
  sc.Build(num_cameras, cam_model, cam_params, num_rigs, rig_ref_cam, num_sensors, sensor_rig, sensor_cam, sensor_pose, num_frames, frame_rig,
           frame_has_pose, frame_q, frame_trn, num_images, image_frame, image_cam, image_present, feat_offset, feat_xy, num_tracks, pt_offset,
           obs_image, obs_feature, pt_xyz);
  for (auto& [id, fr] : sc.frames) fr.is_registered = true;
  BundleAdjusterOptions opt;
  ref_glue::FillBaOptions(o, &opt);
  opt.use_gpu = true;
  gsfm_glomap::BundleAdjuster ba(opt);
  const bool ok = ba.Solve(sc.rigs, sc.cameras, sc.frames, sc.images, sc.tracks);
  out_report[0] = ba.LastReport().initial_cost;
  out_report[1] = ba.LastReport().final_cost;
  out_report[2] = ba.LastReport().iterations;
  out_report[3] = 0.0;
  for (int f = 0; f < num_frames; ++f) {
    const Frame& fr = sc.frames.at(static_cast<frame_t>(f));
    out_frame_q[4 * f] = fr.rig_from_world.rotation.w();
    out_frame_q[4 * f + 1] = fr.rig_from_world.rotation.x();
    out_frame_q[4 * f + 2] = fr.rig_from_world.rotation.y();
    out_frame_q[4 * f + 3] = fr.rig_from_world.rotation.z();
    for (int j = 0; j < 3; ++j) out_frame_t[3 * f + j] = fr.rig_from_world.translation(j);
  }
  for (int k = 0; k < num_cameras; ++k) {
    const Camera& c = sc.cameras.at(static_cast<camera_t>(k));
    for (size_t j = 0; j < c.params.size(); ++j) out_cam_params[ref_glue::kCamRow * k + j] = c.params[j];
  }
  for (long p = 0; p < num_tracks; ++p)
    for (int j = 0; j < 3; ++j) out_xyz[3 * p + j] = sc.tracks.at(static_cast<track_t>(p)).xyz(j);
  return ok ? 1 : 0;
}

}  // extern "C"
