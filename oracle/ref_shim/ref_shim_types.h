// ref_shim_types.h — stand-in scene types for compiling the REFERENCE'S OWN sources into oracle/_ref (test infrastructure).
//
// GLOMAP's containers live in headers that pull in Eigen, COLMAP and glog, none of which exist in this image, so the
// reference cannot be built as a whole.  Two of its translation units, though, are pure container /
// integer logic — glomap/scene/view_graph.cc (connected components) and glomap/controllers/track_establishment.cc (union-find
// track building and the greedy selection) — and touch their types through a handful of members only.  This header declares
// exactly those members, with the reference's names and semantics (cited per member), so that `oracle/Makefile ref` can compile
// the two .cc files FROM /root/reference, unmodified, against it.  Nothing here is reference code; the logic under test is.
// The reference's own view_graph.h and track_establishment.h ARE used (struct ViewGraph, class TrackEngine): -I order puts
// this directory first, /root/reference second, and only the headers named below are shadowed.
#pragma once

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <functional>
#include <iostream>
#include <limits>
#include <queue>
#include <unordered_map>
#include <unordered_set>
#include <utility>
#include <vector>

namespace Eigen {
struct Vector2d {  // what TrackCollection needs: (a - b).norm()   (track_establishment.cc:127-128)
  double x = 0.0, y = 0.0;
  Vector2d() = default;
  Vector2d(double a, double b) : x(a), y(b) {}
  Vector2d operator-(const Vector2d& o) const { return Vector2d(x - o.x, y - o.y); }
  double norm() const { return std::sqrt(x * x + y * y); }
};
struct MatrixXi {  // ImagePair::matches: (row, col) access and rows()   (image_pair.h:46-47, track_establishment.cc:36-47)
  std::vector<int> d;
  long rows() const { return static_cast<long>(d.size() / 2); }
  int operator()(long r, long c) const { return d[static_cast<size_t>(2 * r + c)]; }
};
}  // namespace Eigen

namespace glomap {
using camera_t = uint32_t;  // colmap/util/types.h
using image_t = uint32_t;
using frame_t = uint32_t;
using rig_t = uint32_t;
typedef uint64_t image_pair_t;  // scene/types.h:32
typedef uint32_t feature_t;     // scene/types.h:35
typedef uint64_t track_t;       // scene/types.h:40
using Observation = std::pair<image_t, feature_t>;  // scene/track.h:9

struct Frame {  // scene/frame.h:29-42: the two flags the view-graph code writes
  bool is_registered = false;
  int cluster_id = -1;
};
struct Image {  // scene/image.h:10-53
  image_t image_id = 0;
  frame_t frame_id = 0;
  Frame* frame_ptr = nullptr;
  std::vector<Eigen::Vector2d> features;
  bool IsRegistered() const { return frame_ptr != nullptr && frame_ptr->is_registered; }  // image.h:65-67
};
struct ImagePair {  // scene/image_pair.h:13-57
  image_t image_id1 = 0, image_id2 = 0;
  bool is_valid = true;
  std::vector<int> inliers;
  Eigen::MatrixXi matches;
};
struct Track {  // scene/track.h:12-27
  track_t track_id = 0;
  std::vector<Observation> observations;
};
}  // namespace glomap

// glog's LOG(INFO) << ... (reached through the COLMAP headers in the reference): swallowed
namespace ref_shim {
struct NullLog {
  template <typename T>
  NullLog& operator<<(const T&) { return *this; }
};
}  // namespace ref_shim
#ifndef LOG
#define LOG(severity) ::ref_shim::NullLog()
#endif
