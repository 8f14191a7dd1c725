// colmap/geometry/pose.h for the processors library (oracle/_ref/libref_glomap.so): Sim3d and TransformCameraWorld
#pragma once
#include "ref_shim_sim3.h"
