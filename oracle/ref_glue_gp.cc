// ref_glue_gp.cc — flat C entry point around the REFERENCE'S OWN global positioning problem builder
// (glomap/estimators/global_positioning.cc + cost_function.h, compiled from /root/reference by `make -C oracle ref` against the
// stand-in types of oracle/ref_shim/ and the RECORDING Ceres of oracle/ref_shim/ceres/ceres.h).  GlobalPositioner::Solve runs
// as written — SetupProblem, InitializeRandomPositions, AddCameraToCameraConstraints, AddPointToCameraConstraints,
// AddCamerasAndPointsToParameterGroups, ParameterizeVariables, ceres::Solve (records, evaluates the initial cost, does not
// minimise), ConvertResults — and this file reads the recorded problem back: the random start, every residual block with its
// loss, bound and constant flag, the initial cost.  Test infrastructure: tests/test_oracle_ref.py holds oracle/gp.py to it.
// Trivial frames (image i = frame i = camera i).
#include <cstring>
#include <sstream>

#include "ref_glue_gp_scene.h"

using namespace glomap;

namespace {
class Probe : public GlobalPositioner {  // the recorded problem is a protected member
 public:
  using GlobalPositioner::GlobalPositioner;
  const ceres::Problem& problem() const { return *problem_; }
  const ceres::LossFunction* base_loss() const { return loss_function_.get(); }
  const std::vector<double>& scales() const { return scales_; }
};
struct QuietCout {
  std::streambuf* old;
  std::ostringstream sink;
  QuietCout() : old(std::cout.rdbuf(sink.rdbuf())) {}
  ~QuietCout() { std::cout.rdbuf(old); }
};
}  // namespace

extern "C" {


// Outputs (caller-allocated):
//   frame_order_out [N], track_order_out [P]: the order in which the reference's unordered_maps are walked (its draw order)
//   center_start_out [N][3], xyz_start_out [P][3]: the start point the reference hands to Ceres (camera CENTRES)
//   cam_t_out [N][3]: rig_from_world translations after ConvertResults (no minimisation in between)
//   per residual block r < R (capacity cap_res): res_cam [r] (frame index of the first camera block, -1 none), res_cam2 [r]
//   (second camera of a camera-to-camera block, else -1), res_pt [r] (track index, -1 for a camera pair), res_scale [r],
//   res_loss_scale [r] (the factor of the ScaledLoss around the Huber loss, 1 for the bare Huber), res_lower [r] (lower bound of
//   the scale, NaN none), res_scale_const [r], res_dir [r][3] (the observed direction held by the cost functor)
//   *initial_cost_out: 1/2 sum rho(|r|^2) at the start.   Returns R, or -1 (capacity) / -2 (Solve returned false).
long ref_gp_build(int num_cams, const double* cam_q, const double* cam_t_in, const uint8_t* cam_calibrated, const uint8_t* cam_registered,
                  long num_pts, const long* pt_offset, const int32_t* obs_cam, const double* obs_undist, const double* pt_xyz_in,
                  const uint8_t* pt_initialized, long num_pairs, const int32_t* pair_i, const int32_t* pair_j, const uint8_t* pair_valid,
                  const double* pair_t, const ref_gp_options* o, int32_t* frame_order_out, long* track_order_out,
                  double* center_start_out, double* xyz_start_out, double* cam_t_out, long cap_res, int32_t* res_cam, int32_t* res_cam2,
                  long* res_pt, double* res_scale, double* res_loss_scale, double* res_lower, uint8_t* res_scale_const, double* res_dir,
                  double* initial_cost_out) {
  QuietCout quiet;
  ref_glue::GpScene sc;
  sc.Build(num_cams, cam_q, cam_t_in, cam_calibrated, cam_registered, num_pts, pt_offset, obs_cam, obs_undist, pt_xyz_in, pt_initialized,
           num_pairs, pair_i, pair_j, pair_valid, pair_t);
  auto &rigs = sc.rigs;
  auto &cameras = sc.cameras;
  auto &frames = sc.frames;
  auto &images = sc.images;
  auto &tracks = sc.tracks;
  ViewGraph& vg = sc.vg;
  {  // the walks the reference will make (frames: global_positioning.cc:158; tracks: :258)
    long i = 0;
    for (auto& [fid, fr] : frames) frame_order_out[i++] = static_cast<int32_t>(fid);
    i = 0;
    for (auto& [tid, tr] : tracks) track_order_out[i++] = static_cast<long>(tid);
  }
  GlobalPositionerOptions opt;
  ref_glue::FillGpOptions(o, &opt);
  Probe gp(opt);
  // ConvertResults turns the centres back into translations at the end of Solve; the start point is read through a hook:
  // Solve() is one call, so the centres at "Solve time" are recovered from the result: c = -R^T t (no minimisation happened)
  if (!gp.Solve(vg, rigs, cameras, frames, images, tracks)) return -2;
  const ceres::Problem& prob = gp.problem();
  for (int n = 0; n < num_cams; ++n) {
    const Frame& f = frames.at(n);
    // the start point as the recording Solve saw it (exact); frames outside the problem: CenterFromPose of the input pose
    const auto it = prob.start_values().find(f.rig_from_world.translation.data());
    Eigen::Vector3d c = f.rig_from_world.rotation.inverse() * -f.rig_from_world.translation;
    if (it != prob.start_values().end()) c = Eigen::Vector3d(it->second[0], it->second[1], it->second[2]);
    for (int j = 0; j < 3; ++j) {
      cam_t_out[3 * n + j] = f.rig_from_world.translation(j);
      center_start_out[3 * n + j] = c(j);
    }
  }
  for (long p = 0; p < num_pts; ++p)
    for (int j = 0; j < 3; ++j) xyz_start_out[3 * p + j] = tracks.at(p).xyz(j);
  // the recorded problem
  std::unordered_map<const double*, long> pt_of, scale_idx;
  std::unordered_map<const double*, int> cam_of;
  for (auto& [fid, fr] : frames) cam_of[fr.rig_from_world.translation.data()] = static_cast<int>(fid);
  for (auto& [tid, tr] : tracks) pt_of[tr.xyz.data()] = static_cast<long>(tid);
  const long R = prob.NumResidualBlocks();
  if (R > cap_res) return -1;
  long r = 0;
  for (const auto& b : prob.residual_blocks()) {
    res_cam[r] = res_cam2[r] = -1;
    res_pt[r] = -1;
    const double* scale = nullptr;
    for (double* p : b.params) {
      if (cam_of.count(p)) {
        (res_cam[r] < 0 ? res_cam[r] : res_cam2[r]) = cam_of.at(p);
      } else if (pt_of.count(p)) {
        res_pt[r] = pt_of.at(p);
      } else {
        scale = p;  // the only other block of a trivial-frame problem
      }
    }
    res_scale[r] = scale ? *scale : 0.0;
    res_scale_const[r] = scale && prob.IsConstant(scale) ? 1 : 0;
    const auto lb = prob.lower_bounds().find({const_cast<double*>(scale), 0});
    res_lower[r] = lb == prob.lower_bounds().end() ? std::numeric_limits<double>::quiet_NaN() : lb->second;
    const auto* sl = dynamic_cast<const ceres::ScaledLoss*>(b.loss);
    res_loss_scale[r] = sl ? sl->scale() : 1.0;
    // the observed direction inside the functor: evaluate the residual at zero positions / zero scale -> r = v
    {
      const double z3[3] = {0, 0, 0}, z1[1] = {0};
      const double* pp[4] = {z3, z3, z1, z1};
      double rv[3];
      b.cost->Evaluate(pp, rv, nullptr);
      for (int j = 0; j < 3; ++j) res_dir[3 * r + j] = rv[j];
    }
    ++r;
  }
  // initial cost: recompute it the way the recording Solve did, at the recovered start (ConvertResults moved the cameras,
  // but the parameter blocks of the problem ARE the containers' storage: put the centres back for the evaluation)
  for (auto& [fid, fr] : frames)
    for (int j = 0; j < 3; ++j) fr.rig_from_world.translation(j) = center_start_out[3 * fid + j];
  ceres::Solver::Summary s;
  ceres::Solve(ceres::Solver::Options(), const_cast<ceres::Problem*>(&prob), &s);
  *initial_cost_out = s.initial_cost;
  return R;
}

}  // extern "C"
