#pragma once
// For the rotation averaging library the REFERENCE'S glomap/math/rigid3d.{h,cc} are compiled (AngleAxis stand-in in
// ref_shim_linalg.h); the Makefile passes the path of the real header, which ../ref_shim/glomap/math/rigid3d.h would shadow.
#include "ref_shim_linalg.h"
#include REF_REAL_RIGID3D_H
