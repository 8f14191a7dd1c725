// ref_glue_dropin.cc — the reference's rotation-averaging CONTROLLER (glomap/controllers/rotation_averager.cc:
// SolveRotationAveraging — largest component, the stratified 1-DoF pre-solve, the trivial-rig pre-pass for unknown
// cam_from_rig, ConvertRotationsFromImageToRig, the final solve) run twice on the same containers:
//   which = 0   as the reference builds it: its controller on its own RotationEstimator (all reference code, on the CPU)
//   which = 1   THE DROP-IN: the same controller source, compiled unmodified in oracle/ref_dropin_controller_on_gsfm.cc with
//               include/gsfm_glomap_adapter.hpp's class under the name glomap::RotationEstimator — libgsfm on the GPU
// Both objects live in oracle/_ref/libref_dropin_ra.so (built where /root/reference exists, linked to libgsfm.so; it travels
// to the GPU box).  tests/test_dropin_reference_controller.py: CPU — which = 0 against the Python mirror of the controller;
// GPU — which = 1 against which = 0.
#include <cstring>

#include "ref_glue_ra_scene.h"

#include "glomap/controllers/rotation_averager.h"

namespace glomap {
bool SolveRotationAveragingOnGsfm(ViewGraph& view_graph, std::unordered_map<rig_t, Rig>& rigs, std::unordered_map<frame_t, Frame>& frames,
                                  std::unordered_map<image_t, Image>& images, const RotationAveragerOptions& options);
}

using namespace glomap;
using ref_glue::ref_ra_options;

extern "C" {

// Arguments as ref_ra_estimate (oracle/ref_glue_ra.cc) + use_stratified + images_reversed (ref_glue_ra_scene.h); outputs additionally the registered flag of every frame
// and the validity of every pair after the controller's KeepLargestConnectedComponents calls.  Returns the controller's bool.
int ref_ra_policy(int which, int num_rigs, const int32_t* rig_ref_cam, int num_sensors, const int32_t* sensor_rig, const int32_t* sensor_cam,
                  const int32_t* sensor_state, const double* sensor_q, int num_frames, const int32_t* frame_rig,
                  const uint8_t* frame_has_pose, const double* frame_q, const uint8_t* frame_has_gravity, const double* frame_R_align,
                  const uint8_t* frame_registered, int num_images, const int32_t* image_frame, const int32_t* image_cam, long num_pairs,
                  const int32_t* pair_i, const int32_t* pair_j, const double* pair_q, const double* pair_weight, const int32_t* pair_ninl,
                  const uint8_t* pair_valid, const ref_ra_options* o, int use_stratified, int images_reversed, double* out_frame_q,
                  double* out_sensor_q, uint8_t* out_sensor_has, uint8_t* out_frame_registered, uint8_t* out_pair_valid) {
  ref_glue::RaScene sc;
  sc.images_reversed = images_reversed != 0;
  sc.Build(num_rigs, rig_ref_cam, num_sensors, sensor_rig, sensor_cam, sensor_state, sensor_q, num_frames, frame_rig, frame_has_pose, frame_q,
           frame_has_gravity, frame_R_align, frame_registered, num_images, image_frame, image_cam, num_pairs, pair_i, pair_j, pair_q,
           pair_weight, pair_ninl, pair_valid);
  RotationAveragerOptions opt;
  ref_glue::FillOptions(o, &opt);
  opt.use_stratified = use_stratified != 0;
  const bool ok = which == 0 ? SolveRotationAveraging(sc.view_graph, sc.rigs, sc.frames, sc.images, opt)
                             : SolveRotationAveragingOnGsfm(sc.view_graph, sc.rigs, sc.frames, sc.images, opt);
  sc.ReadBack(num_sensors, sensor_rig, sensor_cam, num_frames, out_frame_q, out_sensor_q, out_sensor_has);
  for (int f = 0; f < num_frames; ++f) out_frame_registered[f] = sc.frames.at(static_cast<frame_t>(f)).is_registered ? 1 : 0;
  for (long e = 0; e < num_pairs; ++e)
    out_pair_valid[e] = sc.view_graph.image_pairs.at(colmap::ImagePairToPairId(static_cast<image_t>(pair_i[e]), static_cast<image_t>(pair_j[e]))).is_valid ? 1 : 0;
  return ok ? 1 : 0;
}

}  // extern "C"
