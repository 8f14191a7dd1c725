#pragma once
// shadows glomap/math/rigid3d.h (its .cc needs Eigen::AngleAxis): the helpers the compiled files call are restated, with
// their source lines, in ref_shim_types.h (CalcAngle, DegToRad, CenterFromPose)
#include "ref_shim_types.h"

#include "glomap/types.h"
