#pragma once
#include <algorithm>
#include <numeric>

#include "boost/graph/adjacency_list.hpp"

namespace boost {
template <typename G, typename Out>
void kruskal_minimum_spanning_tree(G& g, Out out) {
  std::vector<std::size_t> order(g.edges.size()), parent(g.n);
  std::iota(order.begin(), order.end(), std::size_t{0});
  std::iota(parent.begin(), parent.end(), std::size_t{0});
  std::stable_sort(order.begin(), order.end(), [&](std::size_t a, std::size_t b) { return g.w[a] < g.w[b]; });
  auto find = [&](std::size_t x) {
    while (parent[x] != x) { parent[x] = parent[parent[x]]; x = parent[x]; }
    return x;
  };
  for (std::size_t e : order) {
    const std::size_t a = find(g.edges[e].s), b = find(g.edges[e].t);
    if (a == b) continue;
    parent[a] = b;
    *out++ = g.edges[e];
  }
}
}  // namespace boost
