// ba_wide.hip — bundle adjustment with 16-wide intrinsics blocks: FULL_OPENCV, THIN_PRISM_FISHEYE (12 parameters) and
// RAD_TAN_THIN_PRISM_FISHEYE (16), the camera models of colmap/sensor/models.h that do not fit GSFM_CAMERA_MAX_PARAMS = 8
// (the reference dispatches on any CameraModelId: glomap/estimators/bundle_adjustment.cc:136-139,149-152,167-170).
//
// The solver is ba_impl.hpp with GSFM_BA_KP = 16: the same kernels, LM problem class and reductions as the 8-wide unit of
// ba.hip with every intrinsics width scaled (reduced vector [6 per frame | 16 per block], 16 x 16 block-Jacobi blocks, 16
// stored intrinsics planes), one projection instance (camera.hpp: distort_project_wide16), separate pose / intrinsics
// blocks, PCG with the seven gauge modes deflated (A W by operator applications).  A unit of its own rather than template parameters on forty kernels: the measured 8-wide unit
// compiles to the same instructions as before, and everything here is in ba_impl.hpp's anonymous namespace.
#define GSFM_BA_KP 16
#include "ba_impl.hpp"
#include "ba_wide.hpp"

namespace gsfm {

int ba_solve_wide(gsfm_ctx* ctx, const gsfm_ba_problem* prob, const gsfm_ba_options* opt, double* cam_q, double* cam_t,
                  double* pt_xyz, double* intr, gsfm_report* rep) {
  return ba_solve_impl(ctx, prob, opt, cam_q, cam_t, pt_xyz, intr, rep);
}

}  // namespace gsfm
