// ref_glue_ra_scene.h — the containers of a rotation averaging problem built from flat arrays (ids = indices), shared by
// oracle/ref_glue_ra.cc (the reference's estimator) and oracle/ref_glue_dropin.cc (the reference's controller on either estimator).
#pragma once
#include <limits>

#include "glomap/estimators/global_rotation_averaging.h"

#include <colmap/geometry/pose.h>

namespace ref_glue {
using namespace glomap;

inline Eigen::Quaterniond quat(const double* q) { return Eigen::Quaterniond(q[0], q[1], q[2], q[3]); }
inline void put(double* out, const Eigen::Quaterniond& q) { out[0] = q.w(); out[1] = q.x(); out[2] = q.y(); out[3] = q.z(); }

struct ref_ra_options {
  int max_num_l1_iterations;
  double l1_step_convergence_threshold;
  int max_num_irls_iterations;
  double irls_step_convergence_threshold;
  double irls_loss_parameter_sigma;
  int weight_type, skip_initialization, use_weight, use_gravity;
};

struct RaScene {
  std::unordered_map<rig_t, Rig> rigs;
  std::unordered_map<frame_t, Frame> frames;
  std::unordered_map<image_t, Image> images;
  ViewGraph view_graph;
  bool images_reversed = false;

  // sensors: sensor_state 0 = no cam_from_rig (nullopt), 1 = calibrated, 2 = rotation given but translation NaN (to be estimated)
  void Build(int num_rigs, const int32_t* rig_ref_cam, int num_sensors, const int32_t* sensor_rig, const int32_t* sensor_cam,
             const int32_t* sensor_state, const double* sensor_q, int num_frames, const int32_t* frame_rig, const uint8_t* frame_has_pose,
             const double* frame_q, const uint8_t* frame_has_gravity, const double* frame_R_align, const uint8_t* frame_registered,
             int num_images, const int32_t* image_frame, const int32_t* image_cam, long num_pairs, const int32_t* pair_i,
             const int32_t* pair_j, const double* pair_q, const double* pair_weight, const int32_t* pair_ninl, const uint8_t* pair_valid) {
    const double nan = std::numeric_limits<double>::quiet_NaN();
    for (int r = 0; r < num_rigs; ++r) {
      rigs[static_cast<rig_t>(r)].ref = sensor_t(SensorType::CAMERA, static_cast<uint32_t>(rig_ref_cam[r]));
      rigs[static_cast<rig_t>(r)].SetRigId(static_cast<rig_t>(r));
    }
    for (int s = 0; s < num_sensors; ++s) {
      Rig& rig = rigs.at(static_cast<rig_t>(sensor_rig[s]));
      const sensor_t id(SensorType::CAMERA, static_cast<uint32_t>(sensor_cam[s]));
      if (sensor_state[s] == 0) {
        rig.sensors[id] = std::nullopt;
      } else {
        Rigid3d t(quat(sensor_q + 4 * s), Eigen::Vector3d::Zero());
        if (sensor_state[s] == 2) t.translation.setConstant(nan);
        rig.sensors[id] = t;
      }
    }
    frames.reserve(static_cast<size_t>(num_frames));
    for (int f = 0; f < num_frames; ++f) {
      Frame& fr = frames[static_cast<frame_t>(f)];
      fr.SetFrameId(static_cast<frame_t>(f));
      fr.rig_id = static_cast<rig_t>(frame_rig[f]);
      fr.is_registered = frame_registered[f] != 0;
      if (frame_has_pose[f]) fr.SetRigFromWorld(Rigid3d(quat(frame_q + 4 * f), Eigen::Vector3d::Zero()));
      if (frame_has_gravity[f]) {
        fr.gravity_info.has_gravity = true;
        for (int k = 0; k < 9; ++k) fr.gravity_info.R_align.m[k] = frame_R_align[9 * f + k];
      }
    }
    for (auto& [id, fr] : frames) fr.rig_ptr = &rigs.at(fr.rig_id);
    // (images_reversed: the images map is filled in descending id order and without reserve(), so that it iterates in another
    // order than the frames map — the spanning tree's root and the gauge frame then belong to different frames)
    if (!images_reversed) images.reserve(static_cast<size_t>(num_images));
    for (int k = 0; k < num_images; ++k) {
      const int i = images_reversed ? num_images - 1 - k : k;
      Image& im = images[static_cast<image_t>(i)];
      im.image_id = static_cast<image_t>(i);
      im.camera_id = static_cast<camera_t>(image_cam[i]);
      im.frame_id = static_cast<frame_t>(image_frame[i]);
      im.frame_ptr = &frames.at(im.frame_id);
      im.frame_ptr->data_ids.insert(data_t(sensor_t(SensorType::CAMERA, im.camera_id), im.image_id));
    }
    for (long e = 0; e < num_pairs; ++e) {
      ImagePair p;
      p.image_id1 = static_cast<image_t>(pair_i[e]);
      p.image_id2 = static_cast<image_t>(pair_j[e]);
      p.is_valid = pair_valid[e] != 0;
      p.weight = pair_weight[e];
      p.cam2_from_cam1 = Rigid3d(quat(pair_q + 4 * e), Eigen::Vector3d::Zero());
      p.inliers.assign(static_cast<size_t>(pair_ninl[e]), 0);
      view_graph.image_pairs.emplace(colmap::ImagePairToPairId(p.image_id1, p.image_id2), std::move(p));
    }
  }

  void ReadBack(int num_sensors, const int32_t* sensor_rig, const int32_t* sensor_cam, int num_frames, double* out_frame_q,
                double* out_sensor_q, uint8_t* out_sensor_has) {
    for (int f = 0; f < num_frames; ++f) {
      const Frame& fr = frames.at(static_cast<frame_t>(f));
      put(out_frame_q + 4 * f, fr.HasPose() ? fr.RigFromWorld().rotation : Eigen::Quaterniond());
    }
    for (int s = 0; s < num_sensors; ++s) {
      const auto& v = rigs.at(static_cast<rig_t>(sensor_rig[s])).sensors.at(sensor_t(SensorType::CAMERA, static_cast<uint32_t>(sensor_cam[s])));
      out_sensor_has[s] = v.has_value() ? 1 : 0;
      if (v.has_value()) put(out_sensor_q + 4 * s, v.value().rotation);
    }
  }
};

inline void FillOptions(const ref_ra_options* o, RotationEstimatorOptions* opt) {
  opt->max_num_l1_iterations = o->max_num_l1_iterations;
  opt->l1_step_convergence_threshold = o->l1_step_convergence_threshold;
  opt->max_num_irls_iterations = o->max_num_irls_iterations;
  opt->irls_step_convergence_threshold = o->irls_step_convergence_threshold;
  opt->irls_loss_parameter_sigma = o->irls_loss_parameter_sigma;
  opt->weight_type = o->weight_type == 0 ? RotationEstimatorOptions::GEMAN_MCCLURE : RotationEstimatorOptions::HALF_NORM;
  opt->skip_initialization = o->skip_initialization != 0;
  opt->use_weight = o->use_weight != 0;
  opt->use_gravity = o->use_gravity != 0;
}
}  // namespace ref_glue
