// The handful of Boost.Graph names glomap/math/tree.cc uses (Boost is not in this image): an edge list with one double per
// edge, and Kruskal's algorithm in kruskal_min_spanning_tree.hpp.  Among edges of EQUAL weight Boost's order is an
// implementation detail of its priority queue; this stand-in takes them in insertion order (the tests use distinct weights).
#pragma once
#include <cstddef>
#include <utility>
#include <vector>

namespace boost {
struct vecS {};
struct undirectedS {};
struct no_property {};
enum edge_weight_t { edge_weight };
template <typename Tag, typename T>
struct property {};

struct ref_shim_edge {
  std::size_t s = 0, t = 0, idx = 0;
};
template <typename A, typename B, typename C, typename D, typename E>
struct adjacency_list {
  std::size_t n;
  std::vector<ref_shim_edge> edges;
  std::vector<double> w;
  explicit adjacency_list(std::size_t num_vertices) : n(num_vertices) {}
};
template <typename G>
struct graph_traits {
  using edge_descriptor = ref_shim_edge;
  using vertex_descriptor = std::size_t;
};
template <typename G, typename Tag>
struct property_map {
  struct type {
    G* g;
    double& operator[](const ref_shim_edge& e) { return g->w[e.idx]; }
  };
};
template <typename G>
typename property_map<G, edge_weight_t>::type get(edge_weight_t, G& g) {
  return typename property_map<G, edge_weight_t>::type{&g};
}
template <typename G>
std::pair<ref_shim_edge, bool> add_edge(std::size_t u, std::size_t v, G& g) {
  ref_shim_edge e{u, v, g.edges.size()};
  g.edges.push_back(e);
  g.w.push_back(0.0);
  return {e, true};
}
template <typename G>
std::size_t source(const ref_shim_edge& e, const G&) { return e.s; }
template <typename G>
std::size_t target(const ref_shim_edge& e, const G&) { return e.t; }
}  // namespace boost
