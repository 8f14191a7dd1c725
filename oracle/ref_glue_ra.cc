// ref_glue_ra.cc — flat C entry point around the REFERENCE'S OWN rotation averaging: glomap/estimators/
// global_rotation_averaging.cc, estimators/rotation_initializer.cc, math/rigid3d.cc and math/tree.cc, compiled from
// /root/reference by `make -C oracle ref` against the stand-in types of oracle/ref_shim/ and oracle/ref_shim_ra/ (Eigen's
// vectors / sparse matrices / Cholesky / AngleAxis, COLMAP's LAD solver and quaternion average, Boost's Kruskal: none of them
// exist in this image).  RotationEstimator::EstimateRotations runs as written; this file builds its containers from flat arrays
// (oracle/ref_glue_ra_scene.h) and reads the result back.  Test infrastructure: tests/test_oracle_ref_ra.py holds oracle/ra.py
// (all three variants) to it.
#include <cstring>

#include "ref_glue_ra_scene.h"

#include "glomap/math/tree.h"

using namespace glomap;
using ref_glue::ref_ra_options;

namespace {
class Probe : public RotationEstimator {
 public:
  using RotationEstimator::RotationEstimator;
  image_t fixed_camera() const { return fixed_camera_id_; }
};
}  // namespace

extern "C" {

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
  ref_glue::RaScene sc;
  sc.Build(num_rigs, rig_ref_cam, num_sensors, sensor_rig, sensor_cam, sensor_state, sensor_q, num_frames, frame_rig, frame_has_pose, frame_q,
           frame_has_gravity, frame_R_align, frame_registered, num_images, image_frame, image_cam, num_pairs, pair_i, pair_j, pair_q,
           pair_weight, pair_ninl, pair_valid);
  RotationEstimatorOptions opt;
  ref_glue::FillOptions(o, &opt);
  for (int k = 0; k < 8; ++k) out_info[k] = 0;
  out_info[1] = -1;
  if (!opt.skip_initialization && !opt.use_gravity) {  // the root EstimateRotations is about to pick (deterministic)
    std::unordered_map<image_t, image_t> parents;
    out_info[1] = static_cast<long>(MaximumSpanningTree(sc.view_graph, sc.images, parents, INLIER_NUM));
  }
  out_info[5] = static_cast<long>(sc.frames.begin()->first);
  ref_shim::ra_counters() = ref_shim::RaCounters();
  Probe estimator(opt);
  const bool ok = estimator.EstimateRotations(sc.view_graph, sc.rigs, sc.frames, sc.images);
  out_info[0] = static_cast<long>(estimator.fixed_camera());
  out_info[2] = ref_shim::ra_counters().lad_solves;
  out_info[3] = ref_shim::ra_counters().lad_admm_iterations;
  out_info[4] = ref_shim::ra_counters().llt_factorizations;
  sc.ReadBack(num_sensors, sensor_rig, sensor_cam, num_frames, out_frame_q, out_sensor_q, out_sensor_has);
  return ok ? 1 : 0;
}

}  // extern "C"
