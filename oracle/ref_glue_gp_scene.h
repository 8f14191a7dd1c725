// ref_glue_gp_scene.h — the containers of a global positioning problem built from flat arrays (trivial frames: image i =
// frame i = camera i = rig i), shared by oracle/ref_glue_gp.cc (the reference's problem builder on the recording Ceres) and
// oracle/ref_glue_dropin_gpba.cc (the adapter class of include/gsfm_glomap_adapter.hpp on the same containers).
#pragma once
#include "glomap/estimators/global_positioning.h"

extern "C" {
struct ref_gp_options {
  int generate_random_positions, generate_random_points, generate_scales;
  int optimize_positions, optimize_points, optimize_scales;
  int min_num_view_per_track;
  unsigned seed;
  int constraint_type;
  double constraint_reweight_scale;
  double thres_loss_function;
};
}

namespace ref_glue {
using namespace glomap;

struct GpScene {
  std::unordered_map<rig_t, Rig> rigs;
  std::unordered_map<camera_t, Camera> cameras;
  std::unordered_map<frame_t, Frame> frames;
  std::unordered_map<image_t, Image> images;
  std::unordered_map<track_t, Track> tracks;
  ViewGraph vg;

  void Build(int num_cams, const double* cam_q, const double* cam_t_in, const uint8_t* cam_calibrated, const uint8_t* cam_registered,
             long num_pts, const long* pt_offset, const int32_t* obs_cam, const double* obs_undist, const double* pt_xyz_in,
             const uint8_t* pt_initialized, long num_pairs, const int32_t* pair_i, const int32_t* pair_j, const uint8_t* pair_valid,
             const double* pair_t) {
    for (int n = 0; n < num_cams; ++n) {
      Frame f;
      f.is_registered = cam_registered ? cam_registered[n] != 0 : true;
      f.has_pose = true;
      f.rig_from_world.rotation = Eigen::Quaterniond(cam_q[4 * n], cam_q[4 * n + 1], cam_q[4 * n + 2], cam_q[4 * n + 3]);
      f.rig_from_world.translation = Eigen::Vector3d(cam_t_in[3 * n], cam_t_in[3 * n + 1], cam_t_in[3 * n + 2]);
      frames.emplace(static_cast<frame_t>(n), f);
      cameras[n].has_prior_focal_length = cam_calibrated ? cam_calibrated[n] != 0 : true;
      Image im;
      im.image_id = n;
      im.camera_id = n;
      im.frame_id = n;
      images.emplace(static_cast<image_t>(n), im);
    }
    // GLOMAP gives every camera without a rig a trivial rig of its own (the camera is its reference sensor): rig n for frame n
    rigs.reserve(num_cams);
    for (int n = 0; n < num_cams; ++n) rigs[n].ref = sensor_t(SensorType::CAMERA, n);
    for (auto& [fid, fr] : frames) {
      fr.rig_id = fid;
      fr.rig_ptr = &rigs.at(fid);
    }
    for (auto& [id, im] : images) im.frame_ptr = &frames.at(im.frame_id);
    for (long p = 0; p < num_pts; ++p) {
      Track tr;
      tr.track_id = p;
      tr.xyz = Eigen::Vector3d(pt_xyz_in[3 * p], pt_xyz_in[3 * p + 1], pt_xyz_in[3 * p + 2]);
      tr.is_initialized = pt_initialized ? pt_initialized[p] != 0 : false;
      for (long k = pt_offset[p]; k < pt_offset[p + 1]; ++k) {
        Image& im = images.at(obs_cam[k]);
        const feature_t fid = static_cast<feature_t>(im.features_undist.size());
        im.features_undist.emplace_back(obs_undist[3 * k], obs_undist[3 * k + 1], obs_undist[3 * k + 2]);
        tr.observations.emplace_back(static_cast<image_t>(obs_cam[k]), fid);
      }
      tracks.emplace(static_cast<track_t>(p), std::move(tr));
    }
    for (long e = 0; e < num_pairs; ++e) {
      ImagePair pr;
      pr.image_id1 = pair_i[e];
      pr.image_id2 = pair_j[e];
      pr.is_valid = pair_valid ? pair_valid[e] != 0 : true;
      pr.cam2_from_cam1.translation = Eigen::Vector3d(pair_t[3 * e], pair_t[3 * e + 1], pair_t[3 * e + 2]);
      vg.image_pairs.emplace(static_cast<image_pair_t>(e), pr);
    }
  }
};

inline void FillGpOptions(const ref_gp_options* o, GlobalPositionerOptions* out) {
  GlobalPositionerOptions& opt = *out;
  opt.generate_random_positions = o->generate_random_positions != 0;
  opt.generate_random_points = o->generate_random_points != 0;
  opt.generate_scales = o->generate_scales != 0;
  opt.optimize_positions = o->optimize_positions != 0;
  opt.optimize_points = o->optimize_points != 0;
  opt.optimize_scales = o->optimize_scales != 0;
  opt.min_num_view_per_track = o->min_num_view_per_track;
  opt.seed = o->seed;
  opt.constraint_type = static_cast<GlobalPositionerOptions::ConstraintType>(o->constraint_type);
  opt.constraint_reweight_scale = o->constraint_reweight_scale;
  opt.thres_loss_function = o->thres_loss_function;
  opt.use_gpu = false;
}
}  // namespace ref_glue
