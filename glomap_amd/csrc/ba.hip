// ba.hip — the bundle-adjustment entry points of the C ABI and the 8-wide unit of ba_impl.hpp.
//
// Replaces BundleAdjuster::Solve (glomap/estimators/bundle_adjustment.cc:11-106).  The solver (kernels, LM problem class,
// ba_solve_impl) is ba_impl.hpp, compiled here with intrinsics blocks of GSFM_CAMERA_MAX_PARAMS = 8 doubles — the nine camera
// models with at most eight parameters, every measured configuration — and in ba_wide.hip with 16-wide blocks for
// FULL_OPENCV, THIN_PRISM_FISHEYE and RAD_TAN_THIN_PRISM_FISHEYE (colmap::CreateCameraCostFunction dispatches on any
// CameraModelId, bundle_adjustment.cc:136-139,149-152,167-170).  gsfm_ba_solve picks the unit by gsfm_ba_problem::intr_stride.
#define GSFM_BA_KP 8
#include "ba_impl.hpp"
#include "ba_wide.hpp"

using namespace gsfm;

extern "C" void gsfm_ba_options_default(gsfm_ba_options* o) {
  if (!o) return;
  std::memset(o, 0, sizeof(*o));
  lm_options_default(&o->lm, 200);  // bundle_adjustment.h:31
  // BA starts near its solution and its LM trajectory is stiff: reduced solves to 1e-6 leave the result within 3.3e-7 rad /
  // 3e-6 of the exact-solve trajectory at configs[3] (bar: 1e-4 rad / 1e-3), and the chained
  // RA -> GP -> filters -> BA test ends 4e-8 rad / 2e-8 from the oracle chain with it; GP needs 1e-12 (gp.hip)
  o->lm.pcg_relative_tolerance = 1e-6;
  o->thres_loss_function = 1.0;     // bundle_adjustment.h:30
  o->optimize_rotations = 1;
  o->optimize_translation = 1;
  o->optimize_intrinsics = 1;
  o->optimize_principal_point = 0;
  o->optimize_points = 1;
  o->min_num_view_per_track = 3;
  o->optimize_rig_poses = 0;  // bundle_adjustment.h:15
}

extern "C" int gsfm_ba_solve(gsfm_ctx* ctx, const gsfm_ba_problem* prob, const gsfm_ba_options* opt,
                             double* cam_q_inout, double* cam_t_inout, double* pt_xyz_inout,
                             double* intr_params_inout, gsfm_report* report) {
  if (!ctx) return GSFM_ERR_INVALID_ARGUMENT;
  gsfm_report local{};
  if (!report) report = &local;
  std::memset(report, 0, sizeof(*report));
  FlatDump dump(ctx, "ba");
  const bool dumping = dump.active() && prob && opt && cam_q_inout && cam_t_inout && pt_xyz_inout && intr_params_inout;
  if (dumping) {
    const int64_t N = prob->num_cams, K = prob->num_intr, P = prob->num_pts, M = prob->num_obs;
    dump.scalar("num_cams", (double)N);
    dump.scalar("num_intr", (double)K);
    dump.scalar("fixed_cam", prob->fixed_cam);
    dump.scalar("comm_rank", ctx->comm.rank);
    dump.scalar("comm_world", ctx->comm.world);
    dump.array("pt_offset", prob->pt_offset, {P + 1}, prob->mem);
    dump.array("obs_cam", prob->obs_cam, {M}, prob->mem);
    dump.array("obs_xy", prob->obs_xy, {M, 2}, prob->mem);
    if (prob->cam_intr) dump.array("cam_intr", prob->cam_intr, {N}, prob->mem);
    if (prob->num_images > 0 && prob->image_frame && prob->image_cam_from_rig && prob->image_intr) {
      const int64_t I = prob->num_images;
      dump.array("image_frame", prob->image_frame, {I}, prob->mem);
      dump.array("image_cam_from_rig", prob->image_cam_from_rig, {I, 7}, prob->mem);
      dump.array("image_intr", prob->image_intr, {I}, prob->mem);
      if (prob->num_sensors > 0 && prob->image_sensor && prob->sensor_cam_from_rig) {
        dump.array("image_sensor", prob->image_sensor, {I}, prob->mem);
        dump.array("sensor_cam_from_rig", prob->sensor_cam_from_rig, {(int64_t)prob->num_sensors, 7}, GSFM_MEM_HOST);
      }
    }
    dump.array("intr_model", prob->intr_model, {K}, prob->mem);
    dump.array("cam_q", cam_q_inout, {N, 4}, prob->mem);
    dump.array("cam_t", cam_t_inout, {N, 3}, prob->mem);
    dump.array("pt_xyz", pt_xyz_inout, {P, 3}, prob->mem);
    dump.scalar("intr_stride", prob->intr_stride);
    dump.array("intr_params", intr_params_inout, {K, prob->intr_stride > 0 ? prob->intr_stride : GSFM_CAMERA_MAX_PARAMS}, prob->mem);
    dump_lm_options(dump, &opt->lm);
    GSFM_DUMP_OPT(dump, opt, thres_loss_function);
    GSFM_DUMP_OPT(dump, opt, optimize_rotations);
    GSFM_DUMP_OPT(dump, opt, optimize_translation);
    GSFM_DUMP_OPT(dump, opt, optimize_intrinsics);
    GSFM_DUMP_OPT(dump, opt, optimize_principal_point);
    GSFM_DUMP_OPT(dump, opt, optimize_points);
    GSFM_DUMP_OPT(dump, opt, min_num_view_per_track);
    GSFM_DUMP_OPT(dump, opt, optimize_rig_poses);
  }
  const int rc = guarded(ctx, report, [&] {
    GSFM_REQUIRE(prob != nullptr, "BA: null argument");
    const int stride = prob->intr_stride == 0 ? GSFM_CAMERA_MAX_PARAMS : prob->intr_stride;
    GSFM_REQUIRE(stride == GSFM_CAMERA_MAX_PARAMS || stride == GSFM_CAMERA_MAX_PARAMS_WIDE, "BA: intr_stride must be 0, 8 or 16");
    if (stride == GSFM_CAMERA_MAX_PARAMS_WIDE)  // the 16-wide unit (ba_wide.hip)
      return ba_solve_wide(ctx, prob, opt, cam_q_inout, cam_t_inout, pt_xyz_inout, intr_params_inout, report);
    return ba_solve_impl(ctx, prob, opt, cam_q_inout, cam_t_inout, pt_xyz_inout, intr_params_inout, report);
  });
  if (dumping) {
    const int64_t N = prob->num_cams, K = prob->num_intr, P = prob->num_pts;
    dump.array("out_cam_q", cam_q_inout, {N, 4}, prob->mem);
    dump.array("out_cam_t", cam_t_inout, {N, 3}, prob->mem);
    dump.array("out_pt_xyz", pt_xyz_inout, {P, 3}, prob->mem);
    dump.array("out_intr_params", intr_params_inout, {K, prob->intr_stride > 0 ? prob->intr_stride : GSFM_CAMERA_MAX_PARAMS}, prob->mem);
    if (prob->num_images > 0 && prob->num_sensors > 0 && prob->sensor_cam_from_rig)
      dump.array("out_sensor_cam_from_rig", prob->sensor_cam_from_rig, {(int64_t)prob->num_sensors, 7}, GSFM_MEM_HOST);
    dump.write(report, rc);
  }
  return rc;
}
