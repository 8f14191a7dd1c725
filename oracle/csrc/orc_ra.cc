// ORACLE (test infrastructure only — never linked into or called by the product path).
//
// orc_ra.cc — C++ CPU restatement of GLOMAP's rotation averaging, 3-DoF path with trivial rigs (the
// path `glomap mapper` exercises, glomap/estimators/global_rotation_averaging.h:74).  Same algorithm
// as oracle/ra.py, function by function:
//
//   spanning-tree init        global_rotation_averaging.cc:87-138, math/tree.cc:26-153
//   linear system             :141-477  (rows -I3 @ image_id1, +I3 @ image_id2; 3 gauge rows at the fixed node)
//   residuals                 :696-756   b_e = -Log(R_j^T R_rel R_i)
//   update / mean step        :627-644, :758-772
//   L1 (ADMM) stage           :479-541 + colmap::LeastAbsoluteDeviationSolver (COLMAP @ b6b7b54e, un-vendored;
//                             restated from its published ADMM algorithm)
//   IRLS stage                :543-625
//   SO(3) helpers             math/rigid3d.cc:39-63 (Exp with the first-order branch below EPS = 1e-12, Log through
//                             Eigen's matrix -> quaternion -> angle-axis path)
//
// Linear algebra: A^T W A = L_w (x) I3 + gauge, one scalar N x N SPD matrix with three right-hand sides, factored
// DIRECTLY like the reference does (CHOLMOD there; here reverse Cuthill-McKee + skyline Cholesky, single thread).
// The per-edge sweeps run on all OMP threads.  Graphs whose RCM profile is too large for a skyline factor return -7
// (the caller then falls back to oracle/ra.py, whose SuperLU has a fill-reducing ordering).
//
// parity unpinned: the reference stores no numeric vectors for RA.
#include <numeric>
#include <queue>

#include "orc_common.hpp"

namespace orc {

constexpr double EPS = 1e-12;

struct RaOptionsC {
  int32_t max_num_l1_iterations;
  double l1_step_convergence_threshold;
  int32_t max_num_irls_iterations;
  double irls_step_convergence_threshold;
  double irls_loss_parameter_sigma;
  int32_t weight_type, skip_initialization, use_weight;
  int32_t l1_admm_max_num_iterations;
  double l1_admm_rho, l1_admm_alpha, l1_admm_absolute_tolerance, l1_admm_relative_tolerance;
};

struct RaReport {
  int32_t l1_iterations, irls_iterations, factorizations, threads;
  i64 profile_entries;
  double seconds_total, seconds_factor;
};

namespace {

inline void exp_aa(const double* a, double* R) {
  const double th = std::sqrt(a[0] * a[0] + a[1] * a[1] + a[2] * a[2]);
  if (th > EPS) {
    const double kx = a[0] / th, ky = a[1] / th, kz = a[2] / th;
    const double s = std::sin(th), c = std::cos(th), t = 1.0 - c;
    R[0] = t * kx * kx + c;
    R[1] = t * kx * ky - s * kz;
    R[2] = t * kx * kz + s * ky;
    R[3] = t * kx * ky + s * kz;
    R[4] = t * ky * ky + c;
    R[5] = t * ky * kz - s * kx;
    R[6] = t * kx * kz - s * ky;
    R[7] = t * ky * kz + s * kx;
    R[8] = t * kz * kz + c;
  } else {  // I + [a]x, rigid3d.cc:56-61
    R[0] = 1; R[1] = -a[2]; R[2] = a[1];
    R[3] = a[2]; R[4] = 1; R[5] = -a[0];
    R[6] = -a[1]; R[7] = a[0]; R[8] = 1;
  }
}

inline void log_rot(const double* m, double* aa) {
  // Eigen quaternion-from-matrix (Shepperd, Eigen's branch order), then Eigen::AngleAxis(Quaternion)
  double w, v[3];
  const double tr = m[0] + m[4] + m[8];
  if (tr > 0) {
    double t = std::sqrt(tr + 1.0);
    w = 0.5 * t;
    t = 0.5 / t;
    v[0] = (m[7] - m[5]) * t;
    v[1] = (m[2] - m[6]) * t;
    v[2] = (m[3] - m[1]) * t;
  } else {
    int i = 0;
    if (m[4] > m[0]) i = 1;
    if (m[8] > m[4 * i]) i = 2;
    const int j = (i + 1) % 3, k = (j + 1) % 3;
    double t = std::sqrt(m[4 * i] - m[4 * j] - m[4 * k] + 1.0);
    v[i] = 0.5 * t;
    t = 0.5 / t;
    w = (m[3 * k + j] - m[3 * j + k]) * t;
    v[j] = (m[3 * j + i] + m[3 * i + j]) * t;
    v[k] = (m[3 * k + i] + m[3 * i + k]) * t;
  }
  const double n = std::sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
  if (!(n > 0)) {
    aa[0] = aa[1] = aa[2] = 0.0;
    return;
  }
  const double ang = 2.0 * std::atan2(n, std::fabs(w));
  const double sgn = w < 0 ? -1.0 : 1.0;
  for (int c = 0; c < 3; ++c) aa[c] = ang * (v[c] / (n * sgn));
}

inline void quat_to_rot(const double* q, double* R) {  // (w,x,y,z), Eigen toRotationMatrix
  const double w = q[0], x = q[1], y = q[2], z = q[3];
  const double tx = 2 * x, ty = 2 * y, tz = 2 * z;
  const double twx = tx * w, twy = ty * w, twz = tz * w, txx = tx * x, txy = ty * x, txz = tz * x, tyy = ty * y, tyz = tz * y,
               tzz = tz * z;
  R[0] = 1 - (tyy + tzz); R[1] = txy - twz; R[2] = txz + twy;
  R[3] = txy + twz; R[4] = 1 - (txx + tzz); R[5] = tyz - twx;
  R[6] = txz - twy; R[7] = tyz + twx; R[8] = 1 - (txx + tyy);
}

inline void mm(const double* A, const double* B, double* C) {
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) C[3 * i + j] = A[3 * i] * B[j] + A[3 * i + 1] * B[3 + j] + A[3 * i + 2] * B[6 + j];
}
inline void mtm(const double* A, const double* B, double* C) {  // A^T B
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) C[3 * i + j] = A[i] * B[j] + A[3 + i] * B[3 + j] + A[6 + i] * B[6 + j];
}

// Skyline Cholesky of the weighted Laplacian + gauge in a reverse Cuthill-McKee order.
struct Skyline {
  i64 n = 0;
  std::vector<i64> perm, iperm;  // perm[new] = old
  std::vector<i64> first, rowptr;
  std::vector<double> L;
  i64 entries = 0;

  bool analyze(i64 N, i64 E, const int32_t* ei, const int32_t* ej, i64 max_entries) {
    n = N;
    std::vector<i64> deg(N, 0);
    for (i64 e = 0; e < E; ++e) {
      deg[ei[e]]++;
      deg[ej[e]]++;
    }
    std::vector<i64> off(N + 1, 0);
    for (i64 i = 0; i < N; ++i) off[i + 1] = off[i] + deg[i];
    std::vector<int32_t> adj(off[N]);
    std::vector<i64> cur(off.begin(), off.end() - 1);
    for (i64 e = 0; e < E; ++e) {
      adj[cur[ei[e]]++] = ej[e];
      adj[cur[ej[e]]++] = ei[e];
    }
    // Cuthill-McKee from a pseudo-peripheral node of every component, neighbours by ascending degree
    std::vector<i64> order;
    order.reserve(N);
    std::vector<uint8_t> seen(N, 0);
    std::vector<i64> level(N);
    auto bfs_far = [&](i64 s) {
      std::vector<i64> q{s};
      std::vector<i64> touched{s};
      level[s] = 0;
      seen[s] = 2;
      i64 far = s;
      for (size_t h = 0; h < q.size(); ++h) {
        const i64 u = q[h];
        if (level[u] > level[far] || (level[u] == level[far] && deg[u] < deg[far])) far = u;
        for (i64 a = off[u]; a < off[u + 1]; ++a) {
          const i64 v = adj[a];
          if (seen[v] == 0) {
            seen[v] = 2;
            level[v] = level[u] + 1;
            q.push_back(v);
            touched.push_back(v);
          }
        }
      }
      for (i64 v : touched) seen[v] = 0;
      return far;
    };
    for (i64 s0 = 0; s0 < N; ++s0) {
      if (seen[s0]) continue;
      i64 s = bfs_far(s0);
      s = bfs_far(s);
      const size_t head0 = order.size();
      order.push_back(s);
      seen[s] = 1;
      for (size_t h = head0; h < order.size(); ++h) {
        const i64 u = order[h];
        std::vector<i64> nb;
        for (i64 a = off[u]; a < off[u + 1]; ++a)
          if (!seen[adj[a]]) {
            seen[adj[a]] = 1;
            nb.push_back(adj[a]);
          }
        std::sort(nb.begin(), nb.end(), [&](i64 a, i64 b) { return deg[a] != deg[b] ? deg[a] < deg[b] : a < b; });
        for (i64 v : nb) order.push_back(v);
      }
    }
    perm.assign(order.rbegin(), order.rend());
    iperm.assign(N, 0);
    for (i64 i = 0; i < N; ++i) iperm[perm[i]] = i;
    first.assign(N, 0);
    for (i64 i = 0; i < N; ++i) first[i] = i;
    for (i64 e = 0; e < E; ++e) {
      const i64 a = iperm[ei[e]], b = iperm[ej[e]];
      const i64 hi = std::max(a, b), lo = std::min(a, b);
      first[hi] = std::min(first[hi], lo);
    }
    rowptr.assign(N + 1, 0);
    for (i64 i = 0; i < N; ++i) rowptr[i + 1] = rowptr[i] + (i - first[i] + 1);
    entries = rowptr[N];
    return entries <= max_entries;
  }
  inline double& at(i64 i, i64 j) { return L[rowptr[i] + (j - first[i])]; }

  bool factor(i64 E, const int32_t* ei, const int32_t* ej, const double* w, i64 fixed_node, double gauge) {
    L.assign(entries, 0.0);
    for (i64 e = 0; e < E; ++e) {
      const i64 a = iperm[ei[e]], b = iperm[ej[e]];
      at(a, a) += w[e];
      at(b, b) += w[e];
      if (a > b)
        at(a, b) -= w[e];
      else if (b > a)
        at(b, a) -= w[e];
    }
    at(iperm[fixed_node], iperm[fixed_node]) += gauge;
    for (i64 i = 0; i < n; ++i) {
      double* Li = &L[rowptr[i]] - first[i];
      for (i64 j = first[i]; j < i; ++j) {
        const double* Lj = &L[rowptr[j]] - first[j];
        const i64 k0 = std::max(first[i], first[j]);
        double s = Li[j];
        for (i64 k = k0; k < j; ++k) s -= Li[k] * Lj[k];
        Li[j] = s / Lj[j];
      }
      double d = Li[i];
      for (i64 k = first[i]; k < i; ++k) d -= Li[k] * Li[k];
      if (!(d > 0.0)) return false;
      Li[i] = std::sqrt(d);
    }
    return true;
  }

  // x [N][3] in original numbering: solves (L L^T) x = rhs in place
  void solve3(double* x) {
    std::vector<double> y(3 * n);
    for (i64 i = 0; i < n; ++i)
      for (int c = 0; c < 3; ++c) y[3 * i + c] = x[3 * perm[i] + c];
    for (i64 i = 0; i < n; ++i) {
      const double* Li = &L[rowptr[i]] - first[i];
      double s0 = y[3 * i], s1 = y[3 * i + 1], s2 = y[3 * i + 2];
      for (i64 k = first[i]; k < i; ++k) {
        s0 -= Li[k] * y[3 * k];
        s1 -= Li[k] * y[3 * k + 1];
        s2 -= Li[k] * y[3 * k + 2];
      }
      y[3 * i] = s0 / Li[i];
      y[3 * i + 1] = s1 / Li[i];
      y[3 * i + 2] = s2 / Li[i];
    }
    for (i64 i = n - 1; i >= 0; --i) {
      const double* Li = &L[rowptr[i]] - first[i];
      for (int c = 0; c < 3; ++c) y[3 * i + c] /= Li[i];
      for (i64 k = first[i]; k < i; ++k)
        for (int c = 0; c < 3; ++c) y[3 * k + c] -= Li[k] * y[3 * i + c];
    }
    for (i64 i = 0; i < n; ++i)
      for (int c = 0; c < 3; ++c) x[3 * perm[i] + c] = y[3 * i + c];
  }
};

struct Ra {
  i64 N, E;
  const int32_t *ei, *ej;
  std::vector<double> Rrel;  // [E][9]
  std::vector<double> wrow;  // [E] row weights (use_weight)
  i64 fixed;
  double fixed_rot[3];
  OwnerLists bynode;              // 2E incidences grouped by node
  std::vector<int32_t> inc_node;  // [2E]: incidence 2e -> node i (sign -1), 2e+1 -> node j (sign +1)

  void residuals(const std::vector<double>& rot, std::vector<double>& b) const {  // b [E+1][3]
    std::vector<double> Rn(9 * N);
#pragma omp parallel for schedule(static)
    for (i64 n = 0; n < N; ++n) exp_aa(&rot[3 * n], &Rn[9 * n]);
#pragma omp parallel for schedule(static)
    for (i64 e = 0; e < E; ++e) {
      double T[9], Mx[9], aa[3];
      mm(&Rrel[9 * e], &Rn[9 * (i64)ei[e]], T);
      mtm(&Rn[9 * (i64)ej[e]], T, Mx);
      log_rot(Mx, aa);
      for (int c = 0; c < 3; ++c) b[3 * e + c] = -aa[c];
    }
    double F[9], Mx[9];
    exp_aa(fixed_rot, F);
    mtm(F, &Rn[9 * fixed], Mx);
    log_rot(Mx, &b[3 * E]);
  }

  static void update(std::vector<double>& rot, const std::vector<double>& step, i64 N) {
#pragma omp parallel for schedule(static)
    for (i64 n = 0; n < N; ++n) {
      double A[9], B[9], C[9];
      const double ms[3] = {-step[3 * n], -step[3 * n + 1], -step[3 * n + 2]};
      exp_aa(&rot[3 * n], A);
      exp_aa(ms, B);
      mm(A, B, C);
      log_rot(C, &rot[3 * n]);
    }
  }

  double mean_step(const std::vector<double>& step) const {
    const double* s = step.data();
    return chunked_sum(N, [=](i64 n) { return std::sqrt(s[3 * n] * s[3 * n] + s[3 * n + 1] * s[3 * n + 1] + s[3 * n + 2] * s[3 * n + 2]); }) /
           (double)N;
  }

  // out [N][3] = A^T diag(rw) y, y [E+1][3]; rw [E] per-edge factor (gauge row factor 1)
  void At_times(const std::vector<double>& rw, const std::vector<double>& y, std::vector<double>& out) const {
    bynode.reduce<3>(out.data(), false, [&](i64 inc, double* a) {
      const i64 e = inc >> 1;
      const double sg = (inc & 1) ? rw[e] : -rw[e];
      a[0] += sg * y[3 * e];
      a[1] += sg * y[3 * e + 1];
      a[2] += sg * y[3 * e + 2];
    });
    for (int c = 0; c < 3; ++c) out[3 * fixed + c] += y[3 * E + c];
  }
  // out [E+1][3] = diag(rw) A x
  void A_times(const std::vector<double>& rw, const std::vector<double>& x, std::vector<double>& out) const {
#pragma omp parallel for schedule(static)
    for (i64 e = 0; e < E; ++e)
      for (int c = 0; c < 3; ++c) out[3 * e + c] = rw[e] * (x[3 * (i64)ej[e] + c] - x[3 * (i64)ei[e] + c]);
    for (int c = 0; c < 3; ++c) out[3 * E + c] = x[3 * fixed + c];
  }
};

inline double norm(const std::vector<double>& v) {
  const double* p = v.data();
  return std::sqrt(chunked_sum((i64)v.size(), [=](i64 i) { return p[i] * p[i]; }));
}

}  // namespace
}  // namespace orc

extern "C" {
using orc::i64;

// Arrays as gsfm_ra_problem (include/gsfm.h).  rot_aa_inout [N][3].
// Returns 0 ok, -4 numerical failure (reference returns false), -7 graph not suited to the skyline factor.
int orc_ra_solve(int32_t num_nodes, orc::i64 num_edges, const int32_t* edge_i, const int32_t* edge_j, const double* edge_q,
                 const double* edge_weight, const int32_t* edge_ninl, int32_t fixed_node, const orc::RaOptionsC* o,
                 double* rot_aa_inout, orc::RaReport* rep, int32_t num_threads) {
  using namespace orc;
  const double t0 = omp_get_wtime();
  if (num_threads > 0) omp_set_num_threads(num_threads);
  std::memset(rep, 0, sizeof *rep);
  rep->threads = omp_get_max_threads();
  Ra g;
  g.N = num_nodes;
  g.E = num_edges;
  g.ei = edge_i;
  g.ej = edge_j;
  g.fixed = fixed_node;
  const i64 N = g.N, E = g.E;
  g.Rrel.resize(9 * E);
#pragma omp parallel for schedule(static)
  for (i64 e = 0; e < E; ++e) quat_to_rot(edge_q + 4 * e, &g.Rrel[9 * e]);
  std::vector<double> rot(rot_aa_inout, rot_aa_inout + 3 * N);

  // ---- maximum spanning tree initialisation (gra.cc:87-138, tree.cc:78-153)
  if (!o->skip_initialization && E > 0) {
    int32_t max_w = 0;
    for (i64 e = 0; e < E; ++e) max_w = std::max(max_w, edge_ninl[e]);
    std::vector<i64> order(E);
    std::iota(order.begin(), order.end(), 0);
    std::stable_sort(order.begin(), order.end(), [&](i64 a, i64 b) { return (max_w - edge_ninl[a]) < (max_w - edge_ninl[b]); });
    std::vector<i64> parent(N);
    std::iota(parent.begin(), parent.end(), 0);
    auto find = [&](i64 x) {
      while (parent[x] != x) {
        parent[x] = parent[parent[x]];
        x = parent[x];
      }
      return x;
    };
    std::vector<std::vector<std::pair<i64, i64>>> adj(N);
    for (i64 e : order) {
      const i64 a = edge_i[e], b = edge_j[e];
      const i64 ra = find(a), rb = find(b);
      if (ra != rb) {
        parent[ra] = rb;
        adj[a].push_back({b, e});
        adj[b].push_back({a, e});
      }
    }
    std::vector<double> R(9 * N, 0.0);
    for (i64 n = 0; n < N; ++n) R[9 * n] = R[9 * n + 4] = R[9 * n + 8] = 1.0;
    std::vector<uint8_t> visited(N, 0);
    std::vector<i64> queue{0};
    visited[0] = 1;
    for (size_t h = 0; h < queue.size(); ++h) {
      const i64 cur = queue[h];
      for (auto [nb, e] : adj[cur]) {
        if (visited[nb]) continue;
        visited[nb] = 1;
        if (edge_i[e] == nb)
          mtm(&g.Rrel[9 * e], &R[9 * cur], &R[9 * nb]);  // 1_R_w = 2_R_1^T 2_R_w
        else
          mm(&g.Rrel[9 * e], &R[9 * cur], &R[9 * nb]);
        queue.push_back(nb);
      }
    }
    for (i64 n = 0; n < N; ++n)
      if (visited[n]) log_rot(&R[9 * n], &rot[3 * n]);
  }
  for (int c = 0; c < 3; ++c) g.fixed_rot[c] = rot[3 * fixed_node + c];  // gra.cc:248-257

  g.wrow.assign(E, 1.0);
  if (o->use_weight)
    for (i64 e = 0; e < E; ++e) g.wrow[e] = edge_weight[e] >= 0 ? edge_weight[e] : 1.0;  // gra.cc:417-420
  g.inc_node.resize(2 * E);
  for (i64 e = 0; e < E; ++e) {
    g.inc_node[2 * e] = edge_i[e];
    g.inc_node[2 * e + 1] = edge_j[e];
  }
  g.bynode.build(N, 2 * E, g.inc_node.data());

  Skyline sky;
  if (!sky.analyze(N, E, edge_i, edge_j, (i64)4e8)) {
    rep->profile_entries = sky.entries;
    return -7;
  }
  rep->profile_entries = sky.entries;
  std::vector<double> b(3 * (E + 1)), step(3 * N), wl(E);
  int rc = 0;

  // ---- L1 stage (gra.cc:479-541): ADMM on W A with the options of gra.cc:483-486
  if (o->max_num_l1_iterations > 0) {
    for (i64 e = 0; e < E; ++e) wl[e] = g.wrow[e] * g.wrow[e];
    double tf = omp_get_wtime();
    if (!sky.factor(E, edge_i, edge_j, wl.data(), fixed_node, 1.0)) return -4;
    rep->seconds_factor += omp_get_wtime() - tf;
    rep->factorizations++;
    const i64 m = 3 * (E + 1), ncol = 3 * N;
    const double primal_abs = std::sqrt((double)m) * o->l1_admm_absolute_tolerance;
    const double dual_abs = std::sqrt((double)ncol) * o->l1_admm_absolute_tolerance;
    double last_norm = 0.0, curr_norm = 0.0;
    g.residuals(rot, b);
    std::vector<double> bw(m), z(m), u(m), zold(m), tmp(m), Ax(m), x(ncol), t3(ncol);
    for (int it = 0; it < o->max_num_l1_iterations; ++it) {
      last_norm = curr_norm;
      for (i64 e = 0; e < E; ++e)
        for (int c = 0; c < 3; ++c) bw[3 * e + c] = g.wrow[e] * b[3 * e + c];
      for (int c = 0; c < 3; ++c) bw[3 * E + c] = b[3 * E + c];
      // LeastAbsoluteDeviationSolver::Solve
      std::fill(z.begin(), z.end(), 0.0);
      std::fill(u.begin(), u.end(), 0.0);
      std::fill(x.begin(), x.end(), 0.0);
      const double rhs_norm = norm(bw);
      const double rho = o->l1_admm_rho, alpha = o->l1_admm_alpha;
      for (int ai = 0; ai < o->l1_admm_max_num_iterations; ++ai) {
#pragma omp parallel for schedule(static)
        for (i64 i = 0; i < m; ++i) tmp[i] = bw[i] + z[i] - u[i];
        g.At_times(g.wrow, tmp, x);
        sky.solve3(x.data());
        g.A_times(g.wrow, x, Ax);
        zold = z;
        const double kappa = 1.0 / rho;
#pragma omp parallel for schedule(static)
        for (i64 i = 0; i < m; ++i) {
          const double ax_hat = alpha * Ax[i] + (1.0 - alpha) * (zold[i] + bw[i]);
          const double vv = ax_hat - bw[i] + u[i];
          z[i] = std::max(0.0, vv - kappa) - std::max(0.0, -vv - kappa);
          u[i] = u[i] + ax_hat - z[i] - bw[i];
        }
#pragma omp parallel for schedule(static)
        for (i64 i = 0; i < m; ++i) tmp[i] = Ax[i] - z[i] - bw[i];
        const double r_norm = norm(tmp);
#pragma omp parallel for schedule(static)
        for (i64 i = 0; i < m; ++i) tmp[i] = z[i] - zold[i];
        g.At_times(g.wrow, tmp, t3);
        const double s_norm = rho * norm(t3);
        const double max_norm = std::max(std::max(norm(Ax), norm(z)), rhs_norm);
        const double primal_eps = primal_abs + o->l1_admm_relative_tolerance * max_norm;
        g.At_times(g.wrow, u, t3);
        const double dual_eps = dual_abs + o->l1_admm_relative_tolerance * rho * norm(t3);
        if (r_norm < primal_eps && s_norm < dual_eps) break;
      }
      step = x;
      for (double vv : step)
        if (std::isnan(vv)) rc = -4;
      if (rc) break;
      curr_norm = norm(step);
      Ra::update(rot, step, N);
      g.residuals(rot, b);
      const double avg = g.mean_step(step);
      rep->l1_iterations = it + 1;
      if (avg < o->l1_step_convergence_threshold || std::fabs(last_norm - curr_norm) < EPS) break;
    }
  }

  // ---- IRLS stage (gra.cc:543-625)
  if (rc == 0 && o->max_num_irls_iterations > 0) {
    const double sigma = o->irls_loss_parameter_sigma * M_PI / 180.0;
    g.residuals(rot, b);
    std::vector<double> ww(E), rhs(3 * N);
    for (int it = 0; it < o->max_num_irls_iterations; ++it) {
      bool nan = false;
#pragma omp parallel for schedule(static)
      for (i64 e = 0; e < E; ++e) {
        const double e2 = b[3 * e] * b[3 * e] + b[3 * e + 1] * b[3 * e + 1] + b[3 * e + 2] * b[3 * e + 2];
        double w;
        if (o->weight_type == 0) {
          const double tmp = e2 + sigma * sigma;
          w = sigma * sigma / (tmp * tmp);
        } else {
          w = std::pow(e2, (0.5 - 2) / 2);
        }
        if (std::isnan(w)) {
#pragma omp atomic write
          nan = true;
        }
        ww[e] = w * g.wrow[e];
      }
      if (nan) {
        rc = -4;
        break;
      }
      double tf = omp_get_wtime();
      if (!sky.factor(E, edge_i, edge_j, ww.data(), fixed_node, 1.0)) {
        rc = -4;
        break;
      }
      rep->seconds_factor += omp_get_wtime() - tf;
      rep->factorizations++;
      g.At_times(ww, b, rhs);
      sky.solve3(rhs.data());
      step = rhs;
      Ra::update(rot, step, N);
      g.residuals(rot, b);
      const double avg = g.mean_step(step);
      rep->irls_iterations = it + 1;
      if (avg < o->irls_step_convergence_threshold) break;
    }
  }
  std::memcpy(rot_aa_inout, rot.data(), sizeof(double) * 3 * N);
  rep->seconds_total = omp_get_wtime() - t0;
  return rc;
}
}
