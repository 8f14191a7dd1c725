// ref_glue_mapper_stubs.cc — the stages of GlobalMapper::Solve that are OUT OF SCOPE here (preprocessing,
// view-graph calibration, relative-pose estimation, retriangulation, pruning) are named by global_mapper.cc and therefore have
// to link; the mapper runs of oracle/_ref skip them (GlobalMapperOptions::skip_*), and a call would be a test bug: abort loudly.
#include <cstdio>
#include <cstdlib>

#include "glomap/controllers/global_mapper.h"
#include "glomap/processors/image_pair_inliers.h"
#include "glomap/processors/reconstruction_pruning.h"
#include "glomap/processors/view_graph_manipulation.h"

namespace {
[[noreturn]] void out_of_scope(const char* what) {
  std::fprintf(stderr, "[oracle/_ref mapper] %s is not part of this build (skipped stage)\n", what);
  std::abort();
}
}  // namespace

namespace glomap {
bool ViewGraphCalibrator::Solve(ViewGraph&, std::unordered_map<camera_t, Camera>&, std::unordered_map<image_t, Image>&) {
  out_of_scope("ViewGraphCalibrator::Solve");
}
void EstimateRelativePoses(ViewGraph&, std::unordered_map<camera_t, Camera>&, std::unordered_map<image_t, Image>&,
                           const RelativePoseEstimationOptions&) {
  out_of_scope("EstimateRelativePoses");
}
void ImagePairsInlierCount(ViewGraph&, const std::unordered_map<camera_t, Camera>&, const std::unordered_map<image_t, Image>&,
                           const InlierThresholdOptions&, bool) {
  out_of_scope("ImagePairsInlierCount");
}
void ViewGraphManipulater::UpdateImagePairsConfig(ViewGraph&, const std::unordered_map<camera_t, Camera>&,
                                                  const std::unordered_map<image_t, Image>&) {
  out_of_scope("ViewGraphManipulater::UpdateImagePairsConfig");
}
void ViewGraphManipulater::DecomposeRelPose(ViewGraph&, std::unordered_map<camera_t, Camera>&, std::unordered_map<image_t, Image>&) {
  out_of_scope("ViewGraphManipulater::DecomposeRelPose");
}
bool RetriangulateTracks(const TriangulatorOptions&, const colmap::Database&, std::unordered_map<rig_t, Rig>&,
                         std::unordered_map<camera_t, Camera>&, std::unordered_map<frame_t, Frame>&,
                         std::unordered_map<image_t, Image>&, std::unordered_map<track_t, Track>&) {
  out_of_scope("RetriangulateTracks");
}
image_t PruneWeaklyConnectedImages(std::unordered_map<frame_t, Frame>&, std::unordered_map<image_t, Image>&,
                                   std::unordered_map<track_t, Track>&, int, int) {
  out_of_scope("PruneWeaklyConnectedImages");
}
}  // namespace glomap
