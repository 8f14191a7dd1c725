#pragma once
// THE DROP-IN SWITCH, as a maintainer would flip it for one translation unit: the reference's controller
// (glomap/controllers/rotation_averager.cc) is compiled UNMODIFIED with this directory first on the include path, so that the
// name glomap::RotationEstimator it instantiates (rotation_averager.cc:56,159,178,191) is libgsfm's adapter class.
//   1. the reference's own header, with its class under another name (the option structs keep theirs)
//   2. include/gsfm_glomap_adapter.hpp
//   3. the alias
// The Makefile passes the path of the real header (this file shadows it).
#define RotationEstimator RotationEstimatorOfTheReference
#include REF_REAL_GRA_H
#undef RotationEstimator

#include "gsfm_glomap_adapter.hpp"

namespace glomap {
using RotationEstimator = gsfm_glomap::RotationEstimator;
}  // namespace glomap
