#pragma once
// shadows glomap/scene/types_sfm.h: the stand-in types, the REFERENCE'S glomap/types.h (EPS, thresholds) and view_graph.h
// (struct ViewGraph) on top of them
#include "ref_shim_types.h"

#include "glomap/scene/view_graph.h"
#include "glomap/types.h"
