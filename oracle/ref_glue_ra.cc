// ref_glue_ra.cc — flat C entry point around the REFERENCE'S OWN rotation averaging: glomap/estimators/
// global_rotation_averaging.cc, estimators/rotation_initializer.cc, math/rigid3d.cc and math/tree.cc, compiled from
// /root/reference by `make -C oracle ref` against the stand-in types of oracle/ref_shim/ and oracle/ref_shim_ra/ (Eigen's
// vectors / sparse matrices / Cholesky / AngleAxis, COLMAP's LAD solver and quaternion average, Boost's Kruskal: none of them
// exist in this image).  RotationEstimator::EstimateRotations runs as written; this file builds its containers from flat arrays
// and reads the result back.  Test infrastructure: tests/test_oracle_ref.py holds oracle/ra.py (all three variants) to it.
#include <cstring>

#include "glomap/estimators/global_rotation_averaging.h"
#include "glomap/math/tree.h"

#include <colmap/geometry/pose.h>

using namespace glomap;

namespace {
class Probe : public RotationEstimator {
 public:
  using RotationEstimator::RotationEstimator;
  image_t fixed_camera() const { return fixed_camera_id_; }
};
Eigen::Quaterniond quat(const double* q) { return Eigen::Quaterniond(q[0], q[1], q[2], q[3]); }
void put(double* out, const Eigen::Quaterniond& q) { out[0] = q.w(); out[1] = q.x(); out[2] = q.y(); out[3] = q.z(); }
}  // namespace

extern "C" {

struct ref_ra_options {
  int max_num_l1_iterations;
  double l1_step_convergence_threshold;
  int max_num_irls_iterations;
  double irls_step_convergence_threshold;
  double irls_loss_parameter_sigma;
  int weight_type, skip_initialization, use_weight, use_gravity;
};

// Ids are indices: rig r, frame f, image i; camera ids are given (a rig's reference sensor and its other sensors).
//   sensors: s < num_sensors non-reference sensors (sensor_rig, sensor_cam); sensor_state 0 = no cam_from_rig (nullopt),
//            1 = rotation known and translation known (calibrated), 2 = rotation given but translation NaN (to be estimated)
//   frames:  frame_rig, frame_has_pose + frame_q (rig_from_world, wxyz), frame_has_gravity + frame_R_align (row-major 3x3)
//   pairs:   cam2_from_cam1 rotations, weight, number of inliers, validity
// Outputs: out_frame_q [F][4], out_sensor_q [S][4] + out_sensor_has [S], out_info [8] = {fixed image id, spanning tree root
//   (-1 when no tree was built), LAD solves (= L1 iterations), ADMM iterations in total, IRLS factorisations (= IRLS
//   iterations), first frame of the frames map, 0, 0}.   Returns 1 / 0 = EstimateRotations' return value.
int ref_ra_estimate(int num_rigs, const int32_t* rig_ref_cam, int num_sensors, const int32_t* sensor_rig, const int32_t* sensor_cam,
                    const int32_t* sensor_state, const double* sensor_q, int num_frames, const int32_t* frame_rig,
                    const uint8_t* frame_has_pose, const double* frame_q, const uint8_t* frame_has_gravity, const double* frame_R_align,
                    const uint8_t* frame_registered, int num_images, const int32_t* image_frame, const int32_t* image_cam,
                    long num_pairs, const int32_t* pair_i, const int32_t* pair_j, const double* pair_q, const double* pair_weight,
                    const int32_t* pair_ninl, const uint8_t* pair_valid, const ref_ra_options* o, double* out_frame_q,
                    double* out_sensor_q, uint8_t* out_sensor_has, long* out_info) {
  std::unordered_map<rig_t, Rig> rigs;
  std::unordered_map<frame_t, Frame> frames;
  std::unordered_map<image_t, Image> images;
  ViewGraph view_graph;
  const double nan = std::numeric_limits<double>::quiet_NaN();
  for (int r = 0; r < num_rigs; ++r) rigs[static_cast<rig_t>(r)].ref = sensor_t(SensorType::CAMERA, static_cast<uint32_t>(rig_ref_cam[r]));
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
    fr.rig_id = static_cast<rig_t>(frame_rig[f]);
    fr.is_registered = frame_registered[f] != 0;
    if (frame_has_pose[f]) fr.SetRigFromWorld(Rigid3d(quat(frame_q + 4 * f), Eigen::Vector3d::Zero()));
    if (frame_has_gravity[f]) {
      fr.gravity_info.has_gravity = true;
      for (int k = 0; k < 9; ++k) fr.gravity_info.R_align.m[k] = frame_R_align[9 * f + k];
    }
  }
  for (auto& [id, fr] : frames) fr.rig_ptr = &rigs.at(fr.rig_id);
  images.reserve(static_cast<size_t>(num_images));
  for (int i = 0; i < num_images; ++i) {
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
  RotationEstimatorOptions opt;
  opt.max_num_l1_iterations = o->max_num_l1_iterations;
  opt.l1_step_convergence_threshold = o->l1_step_convergence_threshold;
  opt.max_num_irls_iterations = o->max_num_irls_iterations;
  opt.irls_step_convergence_threshold = o->irls_step_convergence_threshold;
  opt.irls_loss_parameter_sigma = o->irls_loss_parameter_sigma;
  opt.weight_type = o->weight_type == 0 ? RotationEstimatorOptions::GEMAN_MCCLURE : RotationEstimatorOptions::HALF_NORM;
  opt.skip_initialization = o->skip_initialization != 0;
  opt.use_weight = o->use_weight != 0;
  opt.use_gravity = o->use_gravity != 0;

  for (int k = 0; k < 8; ++k) out_info[k] = 0;
  out_info[1] = -1;
  if (!opt.skip_initialization && !opt.use_gravity) {  // the root EstimateRotations is about to pick (deterministic)
    std::unordered_map<image_t, image_t> parents;
    out_info[1] = static_cast<long>(MaximumSpanningTree(view_graph, images, parents, INLIER_NUM));
  }
  out_info[5] = static_cast<long>(frames.begin()->first);
  ref_shim::ra_counters() = ref_shim::RaCounters();
  Probe estimator(opt);
  const bool ok = estimator.EstimateRotations(view_graph, rigs, frames, images);
  out_info[0] = static_cast<long>(estimator.fixed_camera());
  out_info[2] = ref_shim::ra_counters().lad_solves;
  out_info[3] = ref_shim::ra_counters().lad_admm_iterations;
  out_info[4] = ref_shim::ra_counters().llt_factorizations;
  for (int f = 0; f < num_frames; ++f) put(out_frame_q + 4 * f, frames.at(static_cast<frame_t>(f)).RigFromWorld().rotation);
  for (int s = 0; s < num_sensors; ++s) {
    const auto& v = rigs.at(static_cast<rig_t>(sensor_rig[s])).sensors.at(sensor_t(SensorType::CAMERA, static_cast<uint32_t>(sensor_cam[s])));
    out_sensor_has[s] = v.has_value() ? 1 : 0;
    if (v.has_value()) put(out_sensor_q + 4 * s, v.value().rotation);
  }
  return ok ? 1 : 0;
}

}  // extern "C"
