#pragma once
// the reference's scene/frame.h pulls in math/gravity.h; the stand-in types do not, so it is added here
#include_next "glomap/scene/types_sfm.h"

#include "glomap/math/gravity.h"
