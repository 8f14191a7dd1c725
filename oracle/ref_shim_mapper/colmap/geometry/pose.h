// colmap/geometry/pose.h for the mapper library (oracle/Makefile ref_mapper), where the rotation averaging sources and the
// reconstruction normaliser meet in one build: what ref_shim_ra/ declares (AverageQuaternions, ImagePairToPairId) and what
// ref_shim/ declares (Sim3d, TransformCameraWorld).
#pragma once
#include_next <colmap/geometry/pose.h>
#include "ref_shim_sim3.h"
