// ba_wide.hpp — entry of the 16-wide bundle-adjustment unit (ba_wide.hip), called by gsfm_ba_solve (ba.hip) for problems
// with gsfm_ba_problem::intr_stride = GSFM_CAMERA_MAX_PARAMS_WIDE.  Same contract as the 8-wide ba_solve_impl.
#pragma once

#include "../../include/gsfm.h"

namespace gsfm {

int ba_solve_wide(gsfm_ctx* ctx, const gsfm_ba_problem* prob, const gsfm_ba_options* opt, double* cam_q, double* cam_t,
                  double* pt_xyz, double* intr, gsfm_report* rep);

}  // namespace gsfm
