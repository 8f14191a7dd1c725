// ref_glue_ba_scene.h — the containers of a bundle adjustment problem built from flat arrays (ids = indices), shared by
// oracle/ref_glue_ba.cc (the reference's problem builder on the recording Ceres) and oracle/ref_glue_dropin_gpba.cc (the
// adapter class of include/gsfm_glomap_adapter.hpp on the same containers).
#pragma once
#include "glomap/estimators/bundle_adjustment.h"

#include <colmap/estimators/cost_functions.h>

extern "C" {
struct ref_ba_options {
  int optimize_rig_poses, optimize_rotations, optimize_translation, optimize_intrinsics, optimize_principal_point, optimize_points;
  int min_num_view_per_track;
  double thres_loss_function;
};
}

namespace ref_glue {
using namespace glomap;

constexpr int kCamRow = 16;  // doubles per camera row of the flat interface (FULL_OPENCV: 12 parameters)

struct BaScene {
  std::unordered_map<rig_t, Rig> rigs;
  std::unordered_map<camera_t, Camera> cameras;
  std::unordered_map<frame_t, Frame> frames;
  std::unordered_map<image_t, Image> images;
  std::unordered_map<track_t, Track> tracks;

  void Build(int num_cameras, const int32_t* cam_model, const double* cam_params, int num_rigs, const int32_t* rig_ref_cam, int num_sensors,
             const int32_t* sensor_rig, const int32_t* sensor_cam, const double* sensor_pose, int num_frames, const int32_t* frame_rig,
             const uint8_t* frame_has_pose, const double* frame_q, const double* frame_trn, int num_images, const int32_t* image_frame,
             const int32_t* image_cam, const uint8_t* image_present, const long* feat_offset, const double* feat_xy, long num_tracks,
             const long* pt_offset, const int32_t* obs_image, const int32_t* obs_feature, const double* pt_xyz) {
    for (int k = 0; k < num_cameras; ++k) {
      Camera& c = cameras[static_cast<camera_t>(k)];
      c.model_id = static_cast<colmap::CameraModelId>(cam_model[k]);
      c.params.assign(cam_params + kCamRow * k, cam_params + kCamRow * k + colmap::NumParams(c.model_id));  // rows of 16 doubles
    }
    for (int r = 0; r < num_rigs; ++r) rigs[static_cast<rig_t>(r)].ref = sensor_t(SensorType::CAMERA, static_cast<uint32_t>(rig_ref_cam[r]));
    for (int s = 0; s < num_sensors; ++s) {
      const double* p = sensor_pose + 7 * s;
      rigs.at(static_cast<rig_t>(sensor_rig[s])).sensors[sensor_t(SensorType::CAMERA, static_cast<uint32_t>(sensor_cam[s]))] =
          Rigid3d(Eigen::Quaterniond(p[0], p[1], p[2], p[3]), Eigen::Vector3d(p[4], p[5], p[6]));
    }
    frames.reserve(static_cast<size_t>(num_frames));
    for (int f = 0; f < num_frames; ++f) {
      Frame& fr = frames[static_cast<frame_t>(f)];
      fr.rig_id = static_cast<rig_t>(frame_rig[f]);
      fr.is_registered = true;
      if (frame_has_pose[f])
        fr.SetRigFromWorld(Rigid3d(Eigen::Quaterniond(frame_q[4 * f], frame_q[4 * f + 1], frame_q[4 * f + 2], frame_q[4 * f + 3]),
                                   Eigen::Vector3d(frame_trn[3 * f], frame_trn[3 * f + 1], frame_trn[3 * f + 2])));
    }
    for (auto& [id, fr] : frames) fr.rig_ptr = &rigs.at(fr.rig_id);
    for (int i = 0; i < num_images; ++i) {
      if (!image_present[i]) continue;  // (an observation of an image that is not in the map is skipped, ba.cc:125)
      Image& im = images[static_cast<image_t>(i)];
      im.image_id = static_cast<image_t>(i);
      im.camera_id = static_cast<camera_t>(image_cam[i]);
      im.frame_id = static_cast<frame_t>(image_frame[i]);
      im.frame_ptr = &frames.at(im.frame_id);
      for (long k = feat_offset[i]; k < feat_offset[i + 1]; ++k) im.features.emplace_back(feat_xy[2 * k], feat_xy[2 * k + 1]);
    }
    tracks.reserve(static_cast<size_t>(num_tracks));
    for (long p = 0; p < num_tracks; ++p) {
      Track& t = tracks[static_cast<track_t>(p)];
      t.track_id = static_cast<track_t>(p);
      t.xyz = Eigen::Vector3d(pt_xyz[3 * p], pt_xyz[3 * p + 1], pt_xyz[3 * p + 2]);
      for (long k = pt_offset[p]; k < pt_offset[p + 1]; ++k)
        t.observations.emplace_back(static_cast<image_t>(obs_image[k]), static_cast<feature_t>(obs_feature[k]));
    }
  }
};

inline void FillBaOptions(const ref_ba_options* o, BundleAdjusterOptions* out) {
  BundleAdjusterOptions& opt = *out;
  opt.optimize_rig_poses = o->optimize_rig_poses != 0;
  opt.optimize_rotations = o->optimize_rotations != 0;
  opt.optimize_translation = o->optimize_translation != 0;
  opt.optimize_intrinsics = o->optimize_intrinsics != 0;
  opt.optimize_principal_point = o->optimize_principal_point != 0;
  opt.optimize_points = o->optimize_points != 0;
  opt.min_num_view_per_track = o->min_num_view_per_track;
  opt.thres_loss_function = o->thres_loss_function;
  opt.use_gpu = false;
}
}  // namespace ref_glue
