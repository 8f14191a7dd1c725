// colmap/sensor/models.h is part of the un-vendored COLMAP dependency: the model ids (ref_shim_types.h) and their names.
#pragma once
#include <string>

#include "ref_shim_types.h"

namespace colmap {
inline std::string CameraModelIdToName(CameraModelId id) { return "model " + std::to_string(static_cast<int>(id)); }
}  // namespace colmap
