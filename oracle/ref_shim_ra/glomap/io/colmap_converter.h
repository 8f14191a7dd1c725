#pragma once
// glomap/io/colmap_converter.{h,cc} convert COLMAP databases (un-vendored COLMAP) and cannot be compiled here; the one
// function controllers/rotation_averager.cc takes from it, restated from colmap_converter.cc:440-462 (a trivial frame for
// one image: frame id, rig id, rig pointer, the image as its only datum, the given pose).
#include "ref_shim_types.h"

namespace glomap {
inline void CreateFrameForImage(const Rigid3d& cam_from_world, Image& image, std::unordered_map<rig_t, Rig>& rigs,
                                std::unordered_map<frame_t, Frame>& frames, rig_t rig_id = -1, frame_t frame_id = -1) {
  Frame frame;
  if (frame_id == colmap::kInvalidFrameId) frame_id = image.image_id;
  if (rig_id == colmap::kInvalidRigId) rig_id = image.camera_id;
  frame.SetFrameId(frame_id);
  frame.SetRigId(rig_id);
  frame.SetRigPtr(rigs.find(rig_id) != rigs.end() ? &rigs[rig_id] : nullptr);
  frame.AddDataId(image.DataId());
  frame.SetRigFromWorld(cam_from_world);
  frames[frame_id] = frame;
  image.frame_id = frame_id;
  image.frame_ptr = &frames[frame_id];
}
}  // namespace glomap
