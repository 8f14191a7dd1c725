// ref_glue_gp_solve.cc — the REFERENCE'S global positioning run TO ITS END POINT (test infrastructure).
// glomap/estimators/global_positioning.cc + cost_function.h, compiled unmodified from /root/reference against the SOLVING Ceres
// stand-in of oracle/ref_shim_solve/ceres/ceres.h (`make -C oracle ref_solve`): GlobalPositioner::Solve as written — problem
// set-up, mt19937 start, residual blocks on the reference's own BATA functors (differentiated by dual numbers), losses, bounds,
// constant blocks, ordering, ceres::Solve, ConvertResults.  What this returns is what a GLOMAP build would return on the same
// containers, up to the minimiser (restated from Ceres' sources, third writing: see that header).
// tests/test_reference_code_gpu.py holds the HIP path's FINAL camera centres to it.  Trivial frames.
#include <cstring>
#include <sstream>

#include "ref_glue_gp_scene.h"

using namespace glomap;

namespace {
struct QuietCout {
  std::streambuf* old;
  std::ostringstream sink;
  QuietCout() : old(std::cout.rdbuf(sink.rdbuf())) {}
  ~QuietCout() { std::cout.rdbuf(old); }
};
}  // namespace

extern "C" {
// In: as ref_gp_build (oracle/ref_glue_gp.cc).  Out (caller-allocated): frame_order_out [N] / track_order_out [P] = the walk
// order of the reference's unordered_maps (its random draw order), center_out [N][3] = camera centres -R^T t after
// ConvertResults, xyz_out [P][3], summary_out [8] = {initial cost, final cost, LM iterations, accepted steps, line-search trials
// beyond the first, steps shortened, termination type, bounds-constrained}, trace_out [cap_trace][7] (rows as gsfm_ctx_lm_trace).
// Returns the number of trace rows (LM iterations recorded), -2 when Solve returned false.
long ref_gp_solve(int num_cams, const double* cam_q, const double* cam_t_in, const uint8_t* cam_calibrated, const uint8_t* cam_registered,
                  long num_pts, const long* pt_offset, const int32_t* obs_cam, const double* obs_undist, const double* pt_xyz_in,
                  const uint8_t* pt_initialized, long num_pairs, const int32_t* pair_i, const int32_t* pair_j, const uint8_t* pair_valid,
                  const double* pair_t, const ref_gp_options* o, int max_num_iterations, int32_t* frame_order_out, long* track_order_out,
                  double* center_out, double* xyz_out, double* summary_out, long cap_trace, double* trace_out) {
  QuietCout quiet;
  ref_glue::GpScene sc;
  sc.Build(num_cams, cam_q, cam_t_in, cam_calibrated, cam_registered, num_pts, pt_offset, obs_cam, obs_undist, pt_xyz_in, pt_initialized,
           num_pairs, pair_i, pair_j, pair_valid, pair_t);
  {
    long i = 0;
    for (auto& [fid, fr] : sc.frames) frame_order_out[i++] = static_cast<int32_t>(fid);
    i = 0;
    for (auto& [tid, tr] : sc.tracks) track_order_out[i++] = static_cast<long>(tid);
  }
  GlobalPositionerOptions opt;
  ref_glue::FillGpOptions(o, &opt);
  if (max_num_iterations > 0) opt.solver_options.max_num_iterations = max_num_iterations;
  GlobalPositioner gp(opt);
  ceres::LastSummary() = ceres::Solver::Summary();
  const bool ok = gp.Solve(sc.vg, sc.rigs, sc.cameras, sc.frames, sc.images, sc.tracks);
  const ceres::Solver::Summary& s = ceres::LastSummary();
  for (int n = 0; n < num_cams; ++n) {
    const Frame& f = sc.frames.at(n);
    const Eigen::Vector3d c = f.rig_from_world.rotation.inverse() * -f.rig_from_world.translation;
    for (int j = 0; j < 3; ++j) center_out[3 * n + j] = c(j);
  }
  for (long p = 0; p < num_pts; ++p)
    for (int j = 0; j < 3; ++j) xyz_out[3 * p + j] = sc.tracks.at(p).xyz(j);
  const double sm[8] = {s.initial_cost, s.final_cost, (double)s.num_iterations, (double)s.num_successful_steps, (double)s.num_line_search_steps,
                        (double)s.num_steps_shortened, (double)s.termination_type, s.is_constrained ? 1.0 : 0.0};
  std::memcpy(summary_out, sm, sizeof sm);
  const long rows = (long)(s.trace.size() / 7);
  std::memcpy(trace_out, s.trace.data(), sizeof(double) * 7 * (size_t)std::min(rows, cap_trace));
  return ok ? rows : -2;
}
}  // extern "C"
