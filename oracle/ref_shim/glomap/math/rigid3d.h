#pragma once
// shadows glomap/math/rigid3d.h (Eigen-based helpers; track_filter.cc needs none of them beyond the types)
#include "ref_shim_types.h"
