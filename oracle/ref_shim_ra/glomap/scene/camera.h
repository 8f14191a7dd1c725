#pragma once
// glomap/math/rigid3d.cc includes glomap/scene/camera.h without using it
#include "ref_shim_types.h"
