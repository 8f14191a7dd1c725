// colmap/math/union_find.h is part of the un-vendored COLMAP dependency (pinned b6b7b54e, thirdparty/CMakeLists.txt:23-28).
// Its published algorithm, restated: a hash-map forest; Find() inserts unknown elements as their own root and compresses the
// path; Union(a, b) hangs the root of a under the root of b ("Link the first point to the second point",
// track_establishment.cc:53-60 relies on exactly that direction).  Which element ends up as a component's root decides the
// reference's track ids, not the partition; the tests compare partitions and canonical (smallest-member) ids.
#pragma once
#include <cstddef>
#include <unordered_map>

namespace colmap {
template <typename T>
class UnionFind {
 public:
  void Reserve(size_t n) { parent_.reserve(n); }
  T Find(const T& elem) {
    auto it = parent_.find(elem);
    if (it == parent_.end()) {
      parent_.emplace(elem, elem);
      return elem;
    }
    if (it->second == elem) return elem;
    const T root = Find(it->second);
    parent_[elem] = root;
    return root;
  }
  void Union(const T& elem1, const T& elem2) {
    const T root1 = Find(elem1);
    const T root2 = Find(elem2);
    if (root1 != root2) parent_[root1] = root2;
  }

 private:
  std::unordered_map<T, T> parent_;
};
}  // namespace colmap
