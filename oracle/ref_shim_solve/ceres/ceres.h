// ceres/ceres.h stand-in, SOLVING variant (test infrastructure; oracle/Makefile `ref_solve`).
//
// ref_shim/ceres/ceres.h records the problem the reference's builders pose and stops at the first cost evaluation.  This one
// minimises, so that glomap/estimators/global_positioning.cc and bundle_adjustment.cc — compiled unmodified from
// /root/reference — run GlobalPositioner::Solve / BundleAdjuster::Solve to their END POINTS here: residuals and Jacobians come
// from the reference's own cost functors evaluated on dual numbers (AutoDiffCostFunction on ceres::Jet), robustified by the
// loss objects the reference created, over the parameter blocks / bounds / constants / manifolds / ordering it registered.
// What is restated (Ceres is un-vendored, cmake/FindDependencies.cmake:4; ceres-solver 2.x sources as published) is the
// minimiser: trust_region_minimizer.cc + levenberg_marquardt_strategy.cc + corrector.cc + line_search.cc + polynomial.cc,
// step by step as oracle/lm.py's header lists them — written a third time, in plain C++ on small dense blocks, sharing no code
// with oracle/lm.py, oracle/csrc/orc_lm.hpp or the product's lm.hpp:
//   * Jacobi scaling 1 / (1 + |J_j|) fixed at the start; LM diagonal clamp(diag(J^T J), 1e-6, 1e32) / radius;
//   * the linear system solved EXACTLY: variable elimination in the order of the reference's linear_solver_ordering (group 0,
//     then group 1, ...; GP: scales, points; BA: points) on small dense factors — what SPARSE_SCHUR + a sparse Cholesky
//     compute — and a dense Cholesky of what remains (the cameras);
//   * step validity by model_cost_change > 0 (radius *= 0.5, five in a row fail); bounds: Plus projects, and a
//     bounds-constrained program runs the Armijo line search (CUBIC interpolation, contraction in [1e-3, 0.6], <= 20 trials,
//     min step 1e-9) before the candidate counts, with the projected gradient in the gradient test;
//   * parameter tolerance, function tolerance (before acceptance), accept rho > 1e-3: radius /= max(1/3, 1 - (2 rho - 1)^3),
//     reject: radius /= decrease_factor (doubling); gradient tolerance 1e-10, min radius 1e-32, iteration cap.
// Manifolds as colmap's helpers tag them (ref_shim_ba/colmap/estimators/manifold.h): EigenQuaternionManifold (x, y, z, w;
// Plus(x, d) = [sin|d| d / |d|, cos|d|] * x) and SubsetManifold.  Small problems only (dense camera system).
#pragma once
#define CERES_SHIM_SOLVING 1
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <limits>
#include <map>
#include <memory>
#include <random>
#include <set>
#include <string>
#include <unordered_map>
#include <utility>
#include <vector>

namespace ceres {
enum Ownership { DO_NOT_TAKE_OWNERSHIP, TAKE_OWNERSHIP };
enum LinearSolverType { DENSE_NORMAL_CHOLESKY, DENSE_QR, SPARSE_NORMAL_CHOLESKY, DENSE_SCHUR, SPARSE_SCHUR, ITERATIVE_SCHUR, CGNR };
enum PreconditionerType { IDENTITY, JACOBI, SCHUR_JACOBI, CLUSTER_JACOBI, CLUSTER_TRIDIAGONAL };
enum DenseLinearAlgebraLibraryType { EIGEN, LAPACK, CUDA };
enum SparseLinearAlgebraLibraryType { SUITE_SPARSE, EIGEN_SPARSE, ACCELERATE_SPARSE, CUDA_SPARSE, NO_SPARSE };
enum TerminationType { CONVERGENCE, NO_CONVERGENCE, FAILURE };

// ---- dual numbers (jet.h): a + v . eps, eps_i eps_j = 0 ------------------------------------------------------------------
template <int N>
struct Jet {
  double a = 0.0;
  double v[N];
  Jet() { for (double& x : v) x = 0.0; }
  Jet(double s) : a(s) { for (double& x : v) x = 0.0; }  // NOLINT: implicit, as in Ceres
  Jet(double s, int k) : a(s) { for (double& x : v) x = 0.0; v[k] = 1.0; }
  Jet& operator+=(const Jet& o) { a += o.a; for (int i = 0; i < N; ++i) v[i] += o.v[i]; return *this; }
  Jet& operator-=(const Jet& o) { a -= o.a; for (int i = 0; i < N; ++i) v[i] -= o.v[i]; return *this; }
  Jet& operator*=(const Jet& o) { *this = *this * o; return *this; }
  Jet& operator/=(const Jet& o) { *this = *this / o; return *this; }
};
template <int N> Jet<N> operator-(const Jet<N>& f) { Jet<N> r; r.a = -f.a; for (int i = 0; i < N; ++i) r.v[i] = -f.v[i]; return r; }
template <int N> Jet<N> operator+(const Jet<N>& f, const Jet<N>& g) { Jet<N> r; r.a = f.a + g.a; for (int i = 0; i < N; ++i) r.v[i] = f.v[i] + g.v[i]; return r; }
template <int N> Jet<N> operator-(const Jet<N>& f, const Jet<N>& g) { Jet<N> r; r.a = f.a - g.a; for (int i = 0; i < N; ++i) r.v[i] = f.v[i] - g.v[i]; return r; }
template <int N> Jet<N> operator*(const Jet<N>& f, const Jet<N>& g) { Jet<N> r; r.a = f.a * g.a; for (int i = 0; i < N; ++i) r.v[i] = f.a * g.v[i] + f.v[i] * g.a; return r; }
template <int N> Jet<N> operator/(const Jet<N>& f, const Jet<N>& g) {
  Jet<N> r;
  const double gi = 1.0 / g.a, q = f.a * gi;
  r.a = q;
  for (int i = 0; i < N; ++i) r.v[i] = (f.v[i] - q * g.v[i]) * gi;
  return r;
}
template <int N> Jet<N> operator+(const Jet<N>& f, double s) { Jet<N> r = f; r.a += s; return r; }
template <int N> Jet<N> operator+(double s, const Jet<N>& f) { Jet<N> r = f; r.a += s; return r; }
template <int N> Jet<N> operator-(const Jet<N>& f, double s) { Jet<N> r = f; r.a -= s; return r; }
template <int N> Jet<N> operator-(double s, const Jet<N>& f) { Jet<N> r = -f; r.a += s; return r; }
template <int N> Jet<N> operator*(const Jet<N>& f, double s) { Jet<N> r; r.a = f.a * s; for (int i = 0; i < N; ++i) r.v[i] = f.v[i] * s; return r; }
template <int N> Jet<N> operator*(double s, const Jet<N>& f) { return f * s; }
template <int N> Jet<N> operator/(const Jet<N>& f, double s) { return f * (1.0 / s); }
template <int N> Jet<N> operator/(double s, const Jet<N>& g) { return Jet<N>(s) / g; }
#define CERES_SHIM_CMP(op)                                                                    \
  template <int N> bool operator op(const Jet<N>& f, const Jet<N>& g) { return f.a op g.a; } \
  template <int N> bool operator op(const Jet<N>& f, double g) { return f.a op g; }          \
  template <int N> bool operator op(double f, const Jet<N>& g) { return f op g.a; }
CERES_SHIM_CMP(<) CERES_SHIM_CMP(<=) CERES_SHIM_CMP(>) CERES_SHIM_CMP(>=) CERES_SHIM_CMP(==) CERES_SHIM_CMP(!=)
#undef CERES_SHIM_CMP
template <int N> Jet<N> chain(double value, double deriv, const Jet<N>& f) { Jet<N> r; r.a = value; for (int i = 0; i < N; ++i) r.v[i] = deriv * f.v[i]; return r; }
template <int N> Jet<N> sqrt(const Jet<N>& f) { const double s = std::sqrt(f.a); return chain(s, 0.5 / s, f); }
template <int N> Jet<N> sin(const Jet<N>& f) { return chain(std::sin(f.a), std::cos(f.a), f); }
template <int N> Jet<N> cos(const Jet<N>& f) { return chain(std::cos(f.a), -std::sin(f.a), f); }
template <int N> Jet<N> tan(const Jet<N>& f) { const double t = std::tan(f.a); return chain(t, 1.0 + t * t, f); }
template <int N> Jet<N> atan(const Jet<N>& f) { return chain(std::atan(f.a), 1.0 / (1.0 + f.a * f.a), f); }
template <int N> Jet<N> atan2(const Jet<N>& g, const Jet<N>& f) {
  Jet<N> r;
  const double d = 1.0 / (f.a * f.a + g.a * g.a);
  r.a = std::atan2(g.a, f.a);
  for (int i = 0; i < N; ++i) r.v[i] = d * (f.a * g.v[i] - g.a * f.v[i]);
  return r;
}
template <int N> Jet<N> abs(const Jet<N>& f) { return f.a < 0.0 ? -f : f; }
using std::sqrt; using std::sin; using std::cos; using std::tan; using std::atan; using std::atan2; using std::abs;

// ---- loss functions (loss_function.h) -----------------------------------------------------------------------------------
class LossFunction {
 public:
  virtual ~LossFunction() = default;
  virtual void Evaluate(double sq_norm, double out[3]) const = 0;
};
class HuberLoss final : public LossFunction {
 public:
  explicit HuberLoss(double a) : a_(a), b_(a * a) {}
  void Evaluate(double s, double rho[3]) const override {
    if (s > b_) {
      const double r = std::sqrt(s);
      rho[0] = 2.0 * a_ * r - b_;
      rho[1] = std::max(std::numeric_limits<double>::min(), a_ / r);
      rho[2] = -rho[1] / (2.0 * s);
    } else {
      rho[0] = s;
      rho[1] = 1.0;
      rho[2] = 0.0;
    }
  }
  double a() const { return a_; }

 private:
  double a_, b_;
};
class CauchyLoss final : public LossFunction {  // rho(s) = b log(1 + s / b), b = a^2 (named by view_graph_calibration.h:26; that stage is skipped)
 public:
  explicit CauchyLoss(double a) : b_(a * a), c_(1.0 / (a * a)) {}
  void Evaluate(double s, double rho[3]) const override {
    const double sum = 1.0 + s * c_, inv = 1.0 / sum;
    rho[0] = b_ * std::log(sum);
    rho[1] = std::max(std::numeric_limits<double>::min(), inv);
    rho[2] = -c_ * (inv * inv);
  }

 private:
  double b_, c_;
};
class ScaledLoss final : public LossFunction {
 public:
  ScaledLoss(const LossFunction* rho, double a, Ownership) : rho_(rho), a_(a) {}
  void Evaluate(double s, double rho[3]) const override {
    if (rho_ == nullptr) {
      rho[0] = a_ * s;
      rho[1] = a_;
      rho[2] = 0.0;
    } else {
      rho_->Evaluate(s, rho);
      rho[0] *= a_;
      rho[1] *= a_;
      rho[2] *= a_;
    }
  }
  double scale() const { return a_; }
  const LossFunction* inner() const { return rho_; }

 private:
  const LossFunction* rho_;
  double a_;
};

// ---- cost functions ------------------------------------------------------------------------------------------------------
class CostFunction {
 public:
  virtual ~CostFunction() = default;
  // jacobians (may be null; entries may be null): row-major num_residuals x block size, as in Ceres
  virtual bool Evaluate(double const* const* parameters, double* residuals, double** jacobians) const = 0;
  int num_residuals() const { return num_residuals_; }
  const std::vector<int>& parameter_block_sizes() const { return sizes_; }

 protected:
  int num_residuals_ = 0;
  std::vector<int> sizes_;
};

template <typename Functor, int kNumResiduals, int... Ns>
class AutoDiffCostFunction final : public CostFunction {
  static constexpr int kNumBlocks = sizeof...(Ns);
  static constexpr int kNumParams = (Ns + ... + 0);
  using JetT = Jet<kNumParams>;

 public:
  explicit AutoDiffCostFunction(Functor* f) : f_(f) {
    num_residuals_ = kNumResiduals;
    sizes_ = {Ns...};
  }
  bool Evaluate(double const* const* p, double* residuals, double** jacobians) const override {
    if (jacobians == nullptr) return call(p, residuals, std::make_index_sequence<kNumBlocks>());
    JetT x[kNumParams > 0 ? kNumParams : 1];
    const JetT* px[kNumBlocks > 0 ? kNumBlocks : 1];
    int off = 0;
    for (int b = 0; b < kNumBlocks; ++b) {
      px[b] = x + off;
      for (int i = 0; i < sizes_[b]; ++i) x[off + i] = JetT(p[b][i], off + i);
      off += sizes_[b];
    }
    JetT r[kNumResiduals];
    if (!call(px, r, std::make_index_sequence<kNumBlocks>())) return false;
    for (int k = 0; k < kNumResiduals; ++k) residuals[k] = r[k].a;
    off = 0;
    for (int b = 0; b < kNumBlocks; ++b) {
      if (jacobians[b] != nullptr)
        for (int k = 0; k < kNumResiduals; ++k)
          for (int i = 0; i < sizes_[b]; ++i) jacobians[b][k * sizes_[b] + i] = r[k].v[off + i];
      off += sizes_[b];
    }
    return true;
  }
  const Functor& functor() const { return *f_; }

 private:
  template <typename T, size_t... I>
  bool call(T const* const* p, T* r, std::index_sequence<I...>) const {
    return (*f_)(p[I]..., r);
  }
  std::unique_ptr<Functor> f_;
};

class ParameterBlockOrdering {
 public:
  bool AddElementToGroup(double* e, int group) {
    group_of_[e] = group;
    return true;
  }
  const std::map<double*, int>& groups() const { return group_of_; }

 private:
  std::map<double*, int> group_of_;
};

class Problem {
 public:
  struct Options {
    Ownership cost_function_ownership = TAKE_OWNERSHIP;
    Ownership loss_function_ownership = TAKE_OWNERSHIP;
  };
  struct ResidualBlock {
    std::unique_ptr<CostFunction> cost;
    LossFunction* loss;
    std::vector<double*> params;
  };
  Problem() = default;
  explicit Problem(const Options&) {}
  template <typename... Ts>
  void* AddResidualBlock(CostFunction* cost, LossFunction* loss, double* x0, Ts*... xs) {
    ResidualBlock b;
    b.cost.reset(cost);
    b.loss = loss;
    b.params = {x0, xs...};
    for (double* p : b.params) blocks_.insert(p);
    residuals_.push_back(std::move(b));
    return &residuals_.back();
  }
  void SetParameterLowerBound(double* values, int index, double lower) { lower_[{values, index}] = lower; }
  void SetParameterUpperBound(double* values, int index, double upper) { upper_[{values, index}] = upper; }
  void SetParameterBlockConstant(const double* values) { constant_.insert(values); }
  bool HasParameterBlock(const double* values) const { return blocks_.count(const_cast<double*>(values)) != 0; }
  int NumResidualBlocks() const { return static_cast<int>(residuals_.size()); }
  // ---- read back by the glue files / the stand-in's own Solve ----
  const std::vector<ResidualBlock>& residual_blocks() const { return residuals_; }
  const std::map<std::pair<double*, int>, double>& lower_bounds() const { return lower_; }
  const std::map<std::pair<double*, int>, double>& upper_bounds() const { return upper_; }
  bool IsConstant(const double* values) const { return constant_.count(values) != 0; }
  struct ManifoldTag {  // 0 = quaternion manifold (Eigen coefficient order), 1 = subset manifold with the listed constant coordinates
    int kind;
    std::vector<int> constant_idxs;
  };
  void RecordManifold(const double* values, int kind, const std::vector<int>& idxs) { manifold_[values] = ManifoldTag{kind, idxs}; }
  const std::map<const double*, ManifoldTag>& manifolds() const { return manifold_; }
  std::map<const double*, std::vector<double>>& start_values() { return start_; }
  const std::map<const double*, std::vector<double>>& start_values() const { return start_; }

 private:
  std::vector<ResidualBlock> residuals_;
  std::set<double*> blocks_;
  std::map<std::pair<double*, int>, double> lower_, upper_;
  std::set<const double*> constant_;
  std::map<const double*, std::vector<double>> start_;
  std::map<const double*, ManifoldTag> manifold_;
};

class Solver {
 public:
  struct Options {
    int num_threads = 1;
    int max_num_iterations = 50;
    bool minimizer_progress_to_stdout = false;
    double function_tolerance = 1e-6;
    double gradient_tolerance = 1e-10;
    double parameter_tolerance = 1e-8;
    double initial_trust_region_radius = 1e4;
    double max_trust_region_radius = 1e16;
    double min_trust_region_radius = 1e-32;
    double min_relative_decrease = 1e-3;
    double min_lm_diagonal = 1e-6;
    double max_lm_diagonal = 1e32;
    int max_num_consecutive_invalid_steps = 5;
    bool jacobi_scaling = true;
    int max_num_line_search_step_size_iterations = 20;
    double line_search_sufficient_function_decrease = 1e-4;
    double max_line_search_step_contraction = 1e-3;
    double min_line_search_step_contraction = 0.6;
    double min_line_search_step_size = 1e-9;
    LinearSolverType linear_solver_type = SPARSE_NORMAL_CHOLESKY;
    PreconditionerType preconditioner_type = JACOBI;
    DenseLinearAlgebraLibraryType dense_linear_algebra_library_type = EIGEN;
    SparseLinearAlgebraLibraryType sparse_linear_algebra_library_type = SUITE_SPARSE;
    std::shared_ptr<ParameterBlockOrdering> linear_solver_ordering;
  };
  struct Summary {
    double initial_cost = 0.0, final_cost = 0.0;
    int num_residual_blocks = 0;
    int num_iterations = 0, num_successful_steps = 0, num_line_search_steps = 0, num_steps_shortened = 0;
    TerminationType termination_type = NO_CONVERGENCE;
    bool is_constrained = false;
    // one row of 7 per LM iteration, the columns of the product's gsfm_ctx_lm_trace: cost | radius | model change | candidate
    // cost | line-search step size | accepted | 0
    std::vector<double> trace;
    std::string BriefReport() const {
      char b[256];
      std::snprintf(b, sizeof b, "stand-in Ceres: iterations %d, initial cost %.6e, final cost %.6e, termination %d", num_iterations,
                    initial_cost, final_cost, static_cast<int>(termination_type));
      return b;
    }
    std::string FullReport() const { return BriefReport(); }
    bool IsSolutionUsable() const { return termination_type != FAILURE; }
  };
};

// the summary of the last ceres::Solve of this process (the reference's Solve methods keep theirs local): read by the glue files
inline Solver::Summary& LastSummary() {
  static Solver::Summary s;
  return s;
}

namespace shim {
// ---- small dense helpers -----------------------------------------------------------------------------------------------
inline bool cholesky(std::vector<double>& A, int n) {  // in place, lower triangle; false when not positive definite
  for (int j = 0; j < n; ++j) {
    double d = A[(size_t)j * n + j];
    for (int k = 0; k < j; ++k) d -= A[(size_t)j * n + k] * A[(size_t)j * n + k];
    if (!(d > 0.0)) return false;
    d = std::sqrt(d);
    A[(size_t)j * n + j] = d;
    for (int i = j + 1; i < n; ++i) {
      double v = A[(size_t)i * n + j];
      for (int k = 0; k < j; ++k) v -= A[(size_t)i * n + k] * A[(size_t)j * n + k];
      A[(size_t)i * n + j] = v / d;
    }
  }
  return true;
}
inline void cholesky_solve(const std::vector<double>& L, int n, double* x, int nrhs, int ldx) {  // x: n x nrhs, row-major with stride ldx
  for (int c = 0; c < nrhs; ++c) {
    for (int i = 0; i < n; ++i) {
      double v = x[(size_t)i * ldx + c];
      for (int k = 0; k < i; ++k) v -= L[(size_t)i * n + k] * x[(size_t)k * ldx + c];
      x[(size_t)i * ldx + c] = v / L[(size_t)i * n + i];
    }
    for (int i = n - 1; i >= 0; --i) {
      double v = x[(size_t)i * ldx + c];
      for (int k = i + 1; k < n; ++k) v -= L[(size_t)k * n + i] * x[(size_t)k * ldx + c];
      x[(size_t)i * ldx + c] = v / L[(size_t)i * n + i];
    }
  }
}

struct Block {
  double* user = nullptr;
  int size = 0, tsize = 0, kind = -1, group = 1 << 30;
  std::vector<int> free_idx;        // subset manifold: the coordinates that move
  std::vector<double> lower, upper;  // per coordinate
  int xoff = 0, toff = 0;           // offsets in the ambient / tangent vectors of the reduced program
  bool bounded = false;
};

// polynomial.cc / line_search.cc (see oracle/lm.py): highest power first
inline double peval(const std::vector<double>& p, double x) { double v = 0.0; for (double a : p) v = v * x + a; return v; }
inline std::vector<double> pderiv(const std::vector<double>& p) {
  const int deg = (int)p.size() - 1;
  std::vector<double> d;
  for (int j = 0; j < deg; ++j) d.push_back((deg - j) * p[j]);
  if (d.empty()) d.push_back(0.0);
  return d;
}
inline std::vector<double> pstrip(std::vector<double> p) {
  size_t i = 0;
  while (i + 1 < p.size() && p[i] == 0.0) ++i;
  p.erase(p.begin(), p.begin() + i);
  return p;
}
struct Sample { double x, value, slope; bool valid; };
inline std::vector<double> pfit(const std::vector<Sample>& smp) {
  const int n = 2 * (int)smp.size(), deg = n - 1;
  std::vector<std::vector<double>> A(n, std::vector<double>(n + 1, 0.0));
  int row = 0;
  for (const Sample& s : smp) {
    for (int j = 0; j <= deg; ++j) A[row][j] = std::pow(s.x, deg - j);
    A[row++][n] = s.value;
    for (int j = 0; j < deg; ++j) A[row][j] = (deg - j) * std::pow(s.x, deg - j - 1);
    A[row++][n] = s.slope;
  }
  std::vector<int> perm(n);
  for (int i = 0; i < n; ++i) perm[i] = i;
  for (int k = 0; k < n; ++k) {
    int pr = k, pc = k;
    double best = -1.0;
    for (int r = k; r < n; ++r)
      for (int c = k; c < n; ++c)
        if (std::fabs(A[r][c]) > best) best = std::fabs(A[r][c]), pr = r, pc = c;
    if (!(best > 0.0)) break;
    std::swap(A[pr], A[k]);
    if (pc != k) {
      for (int r = 0; r < n; ++r) std::swap(A[r][pc], A[r][k]);
      std::swap(perm[pc], perm[k]);
    }
    for (int r = k + 1; r < n; ++r) {
      const double f = A[r][k] / A[k][k];
      if (f != 0.0)
        for (int c = k; c <= n; ++c) A[r][c] -= f * A[k][c];
    }
  }
  std::vector<double> y(n, 0.0), coef(n, 0.0);
  for (int k = n - 1; k >= 0; --k) {
    double v = A[k][n];
    for (int c = k + 1; c < n; ++c) v -= A[k][c] * y[c];
    y[k] = A[k][k] != 0.0 ? v / A[k][k] : 0.0;
  }
  for (int k = 0; k < n; ++k) coef[perm[k]] = y[k];
  return coef;
}
inline std::vector<double> proots(const std::vector<double>& pin, double lo, double hi) {
  const std::vector<double> p = pstrip(pin);
  std::vector<double> roots;
  const int deg = (int)p.size() - 1;
  if (deg <= 0) return roots;
  if (deg == 1) {
    const double r = -p[1] / p[0];
    if (r >= lo && r <= hi) roots.push_back(r);
    return roots;
  }
  std::vector<double> brk{lo};
  for (double c : proots(pderiv(p), lo, hi)) brk.push_back(c);
  brk.push_back(hi);
  for (size_t i = 0; i + 1 < brk.size(); ++i) {
    double a = brk[i], b = brk[i + 1], fa = peval(p, a);
    const double fb = peval(p, b);
    if (fa == 0.0) { if (roots.empty() || roots.back() != a) roots.push_back(a); continue; }
    if (fb == 0.0) { if (i + 2 == brk.size()) roots.push_back(b); continue; }
    if ((fa < 0.0) == (fb < 0.0)) continue;
    for (int it = 0; it < 200; ++it) {
      const double m = 0.5 * (a + b);
      if (!(m > a && m < b)) break;
      const double fm = peval(p, m);
      if (fm == 0.0) { a = b = m; break; }
      if ((fm < 0.0) == (fa < 0.0)) { a = m; fa = fm; } else { b = m; }
    }
    roots.push_back(0.5 * (a + b));
  }
  return roots;
}
inline double pminimize(const std::vector<double>& poly, double lo, double hi) {
  double bx = 0.5 * (lo + hi), bv = peval(poly, bx);
  for (double x : {lo, hi}) { const double v = peval(poly, x); if (v < bv) bx = x, bv = v; }
  const std::vector<double> p = pstrip(poly);
  if (p.size() <= 2) return bx;
  for (double x : proots(pderiv(p), lo, hi)) { const double v = peval(poly, x); if (v < bv) bx = x, bv = v; }
  return bx;
}

class Minimizer {
 public:
  Minimizer(const Solver::Options& o, Problem* prob) : o_(o), prob_(prob) {}

  void Run(Solver::Summary* sum) {
    Setup();
    sum->num_residual_blocks = prob_->NumResidualBlocks();
    sum->is_constrained = constrained_;
    sum->trace.clear();
    x_.resize(nx_);
    for (const Block& b : blocks_) std::memcpy(&x_[b.xoff], b.user, sizeof(double) * b.size);
    double cost = Evaluate(x_, true);
    sum->initial_cost = sum->final_cost = cost;
    if (nt_ == 0) { sum->termination_type = CONVERGENCE; return; }
    // Jacobi scaling from the first Jacobian
    scale_.assign(nt_, 1.0);
    if (o_.jacobi_scaling) {
      std::vector<double> colsq(nt_, 0.0);
      for (const Res& r : res_)
        for (size_t k = 0; k < r.blk.size(); ++k) {
          const Block& b = blocks_[r.blk[k]];
          for (int row = 0; row < r.nres; ++row)
            for (int c = 0; c < b.tsize; ++c) { const double v = r.J[k][(size_t)row * b.tsize + c]; colsq[b.toff + c] += v * v; }
        }
      for (int j = 0; j < nt_; ++j) scale_[j] = 1.0 / (1.0 + std::sqrt(colsq[j]));
    }
    Gradient();
    if (GradMaxNorm() <= o_.gradient_tolerance) { sum->termination_type = CONVERGENCE; return; }
    double radius = o_.initial_trust_region_radius, decrease = 2.0;
    int invalid = 0, it = 0;
    double xnorm = Norm(x_);
    sum->termination_type = NO_CONVERGENCE;
    std::vector<double> delta(nt_), cand(nx_), xt(nx_);
    while (true) {
      if (it >= o_.max_num_iterations) { sum->termination_type = NO_CONVERGENCE; break; }
      if (radius < o_.min_trust_region_radius) { sum->termination_type = CONVERGENCE; break; }
      ++it;
      sum->num_iterations = it;
      double model_change = 0.0;
      bool valid = Step(radius, &delta, &model_change);
      valid = valid && std::isfinite(model_change) && model_change > 0.0;
      if (!valid) {
        sum->trace.insert(sum->trace.end(), {cost, radius, model_change, 0.0, 1.0, -1.0, 0.0});
        if (++invalid >= o_.max_num_consecutive_invalid_steps) { sum->termination_type = FAILURE; break; }
        radius *= 0.5;
        continue;
      }
      invalid = 0;
      double step_size = 1.0;
      if (constrained_ && o_.max_num_line_search_step_size_iterations > 0) {
        // TrustRegionMinimizer::DoLineSearch: Armijo along t -> Plus(x, t delta), first trial t = 1
        double slope0 = 0.0, dmax = 0.0;
        for (int j = 0; j < nt_; ++j) { slope0 += g_[j] * delta[j]; dmax = std::max(dmax, std::fabs(delta[j])); }
        auto eval = [&](double t, double* value, double* slope) {
          std::vector<double> td(delta);
          for (double& v : td) v *= t;
          Plus(x_, td, &xt);
          *value = Evaluate(xt, true);
          std::vector<double> gsave(g_);
          Gradient();
          *slope = 0.0;
          for (int j = 0; j < nt_; ++j) *slope += g_[j] * delta[j];
          g_.swap(gsave);
        };
        const Sample lower{0.0, cost, slope0, true};
        Sample prev{0, 0, 0, false}, cur{1.0, 0, 0, false};
        eval(1.0, &cur.value, &cur.slope);
        cur.valid = std::isfinite(cur.value);
        int iters = 0;
        bool ok = true;
        while (!cur.valid || cur.value > cost + o_.line_search_sufficient_function_decrease * slope0 * cur.x) {
          if (++iters >= o_.max_num_line_search_step_size_iterations) { ok = false; break; }
          const double lo = o_.max_line_search_step_contraction * cur.x, hi = o_.min_line_search_step_contraction * cur.x;
          double t;
          if (!cur.valid) {
            t = std::min(std::max(cur.x * 0.5, lo), hi);
          } else {
            std::vector<Sample> smp{lower, cur};
            if (prev.valid) smp.push_back(prev);
            t = pminimize(pfit(smp), lo, hi);
          }
          if (t * dmax < o_.min_line_search_step_size) { ok = false; break; }
          prev = cur;
          cur = Sample{t, 0, 0, false};
          eval(t, &cur.value, &cur.slope);
          cur.valid = std::isfinite(cur.value);
          ++sum->num_line_search_steps;
        }
        if (ok && cur.x != 1.0) {
          for (double& v : delta) v *= cur.x;
          ++sum->num_steps_shortened;
        }
        step_size = ok ? cur.x : -1.0;
        Evaluate(x_, true);  // the Jacobian / residuals of the current point again (the trials overwrote them)
      }
      Plus(x_, delta, &cand);
      const double cand_cost = Evaluate(cand, false);
      double sn = 0.0;
      for (int i = 0; i < nx_; ++i) sn += (cand[i] - x_[i]) * (cand[i] - x_[i]);
      if (std::sqrt(sn) <= o_.parameter_tolerance * (xnorm + o_.parameter_tolerance)) {
        sum->trace.insert(sum->trace.end(), {cost, radius, model_change, cand_cost, step_size, 0.0, 0.0});
        sum->termination_type = CONVERGENCE;
        break;
      }
      const double cost_change = cost - cand_cost;
      if (std::fabs(cost_change) <= o_.function_tolerance * cost) {
        sum->trace.insert(sum->trace.end(), {cost, radius, model_change, cand_cost, step_size, 0.0, 0.0});
        sum->termination_type = CONVERGENCE;
        break;
      }
      const double rho = cost_change / model_change;
      sum->trace.insert(sum->trace.end(), {cost, radius, model_change, cand_cost, step_size, rho > o_.min_relative_decrease ? 1.0 : 0.0, 0.0});
      if (o_.minimizer_progress_to_stdout)
        std::fprintf(stderr, "[stand-in ceres] it %d cost %.9e cand %.9e radius %.3e t %.4f rho %.3e\n", it, cost, cand_cost, radius, step_size, rho);
      if (rho > o_.min_relative_decrease) {
        x_ = cand;
        xnorm = Norm(x_);
        cost = Evaluate(x_, true);
        Gradient();
        ++sum->num_successful_steps;
        if (GradMaxNorm() <= o_.gradient_tolerance) { sum->termination_type = CONVERGENCE; break; }
        const double t = 2.0 * rho - 1.0;
        radius = std::min(o_.max_trust_region_radius, radius / std::max(1.0 / 3.0, 1.0 - t * t * t));
        decrease = 2.0;
      } else {
        radius /= decrease;
        decrease *= 2.0;
      }
    }
    sum->final_cost = cost;
    for (const Block& b : blocks_) std::memcpy(b.user, &x_[b.xoff], sizeof(double) * b.size);
  }

 private:
  struct Res {  // a residual block of the reduced program
    const Problem::ResidualBlock* rb;
    int nres;
    std::vector<int> blk;                 // free blocks (indices into blocks_), in the block's parameter order
    std::vector<int> slot;                // their position in rb->params
    std::vector<double> r;                // robustified residuals
    std::vector<std::vector<double>> J;   // robustified tangent Jacobians, nres x tsize each (NOT Jacobi-scaled)
  };

  void Setup() {
    std::unordered_map<double*, int> idx;
    const auto* ord = o_.linear_solver_ordering.get();
    for (const auto& rb : prob_->residual_blocks()) {
      Res r;
      r.rb = &rb;
      r.nres = rb.cost->num_residuals();
      for (size_t k = 0; k < rb.params.size(); ++k) {
        double* p = rb.params[k];
        if (prob_->IsConstant(p)) continue;
        auto it = idx.find(p);
        if (it == idx.end()) {
          Block b;
          b.user = p;
          b.size = rb.cost->parameter_block_sizes()[k];
          b.tsize = b.size;
          const auto m = prob_->manifolds().find(p);
          if (m != prob_->manifolds().end()) {
            b.kind = m->second.kind;
            if (b.kind == 0) b.tsize = 3;
            if (b.kind == 1) {
              for (int i = 0; i < b.size; ++i)
                if (std::find(m->second.constant_idxs.begin(), m->second.constant_idxs.end(), i) == m->second.constant_idxs.end()) b.free_idx.push_back(i);
              b.tsize = (int)b.free_idx.size();
            }
          }
          b.lower.assign(b.size, -std::numeric_limits<double>::max());
          b.upper.assign(b.size, std::numeric_limits<double>::max());
          for (int i = 0; i < b.size; ++i) {
            const auto lo = prob_->lower_bounds().find({p, i});
            if (lo != prob_->lower_bounds().end()) b.lower[i] = lo->second, b.bounded = true;
            const auto hi = prob_->upper_bounds().find({p, i});
            if (hi != prob_->upper_bounds().end()) b.upper[i] = hi->second, b.bounded = true;
          }
          if (ord) {
            const auto g = ord->groups().find(p);
            if (g != ord->groups().end()) b.group = g->second;
          }
          b.xoff = nx_;
          b.toff = nt_;
          nx_ += b.size;
          nt_ += b.tsize;
          constrained_ = constrained_ || b.bounded;  // Program::IsBoundsConstrained: a non-constant block with a bound
          it = idx.emplace(p, (int)blocks_.size()).first;
          blocks_.push_back(b);
        }
        r.blk.push_back(it->second);
        r.slot.push_back((int)k);
      }
      res_.push_back(std::move(r));
    }
    // elimination order: every group but the last (largest id among the free blocks), blocks in order of appearance
    int last_group = std::numeric_limits<int>::min();
    for (const Block& b : blocks_) last_group = std::max(last_group, b.group);
    std::vector<int> order;
    for (int i = 0; i < (int)blocks_.size(); ++i)
      if (blocks_[i].group != last_group) order.push_back(i);
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return blocks_[a].group < blocks_[b].group; });
    elim_ = order;
  }

  static double Norm(const std::vector<double>& v) { double s = 0.0; for (double a : v) s += a * a; return std::sqrt(s); }

  // Plus of every block, projected on the bounds (ParameterBlock::Plus)
  void Plus(const std::vector<double>& x, const std::vector<double>& d, std::vector<double>* out) const {
    out->resize(nx_);
    for (const Block& b : blocks_) {
      const double* xb = &x[b.xoff];
      const double* db = &d[b.toff];
      double* ob = &(*out)[b.xoff];
      if (b.kind == 0) {  // EigenQuaternionManifold: q_delta * x, storage (x, y, z, w)
        const double n2 = db[0] * db[0] + db[1] * db[1] + db[2] * db[2];
        if (n2 > 0.0) {
          const double n = std::sqrt(n2), k = std::sin(n) / n;
          const double qd[4] = {std::cos(n), k * db[0], k * db[1], k * db[2]};  // (w, x, y, z)
          const double qx[4] = {xb[3], xb[0], xb[1], xb[2]};
          const double w = qd[0] * qx[0] - qd[1] * qx[1] - qd[2] * qx[2] - qd[3] * qx[3];
          const double i = qd[0] * qx[1] + qd[1] * qx[0] + qd[2] * qx[3] - qd[3] * qx[2];
          const double j = qd[0] * qx[2] - qd[1] * qx[3] + qd[2] * qx[0] + qd[3] * qx[1];
          const double kk = qd[0] * qx[3] + qd[1] * qx[2] - qd[2] * qx[1] + qd[3] * qx[0];
          ob[0] = i; ob[1] = j; ob[2] = kk; ob[3] = w;
        } else {
          std::memcpy(ob, xb, sizeof(double) * 4);
        }
      } else if (b.kind == 1) {
        std::memcpy(ob, xb, sizeof(double) * b.size);
        for (int c = 0; c < b.tsize; ++c) ob[b.free_idx[c]] += db[c];
      } else {
        for (int i = 0; i < b.size; ++i) ob[i] = xb[i] + db[i];
      }
      for (int i = 0; i < b.size; ++i) ob[i] = std::min(std::max(ob[i], b.lower[i]), b.upper[i]);
    }
  }

  // cost at x; with_jacobians: also the robustified residuals and tangent Jacobians of every residual block (Corrector)
  double Evaluate(const std::vector<double>& x, bool with_jacobians) {
    double total = 0.0;
    std::vector<double> amb;
    for (Res& r : res_) {
      const auto& sizes = r.rb->cost->parameter_block_sizes();
      const size_t np = r.rb->params.size();
      std::vector<const double*> pp(np);
      for (size_t k = 0; k < np; ++k) pp[k] = r.rb->params[k];
      for (size_t k = 0; k < r.blk.size(); ++k) pp[r.slot[k]] = &x[blocks_[r.blk[k]].xoff];
      std::vector<double> rbuf;  // (a cost-only evaluation must not touch the residuals / Jacobians of the current point)
      std::vector<double>& rr = with_jacobians ? r.r : rbuf;
      rr.assign(r.nres, 0.0);
      std::vector<std::vector<double>> Ja;
      std::vector<double*> jp(np, nullptr);
      if (with_jacobians) {
        Ja.resize(r.blk.size());
        for (size_t k = 0; k < r.blk.size(); ++k) {
          Ja[k].assign((size_t)r.nres * sizes[r.slot[k]], 0.0);
          jp[r.slot[k]] = Ja[k].data();
        }
      }
      if (!r.rb->cost->Evaluate(pp.data(), rr.data(), with_jacobians ? jp.data() : nullptr)) return std::numeric_limits<double>::infinity();
      double s = 0.0;
      for (double v : rr) s += v * v;
      double rho[3] = {s, 1.0, 0.0};
      if (r.rb->loss) r.rb->loss->Evaluate(s, rho);
      total += 0.5 * rho[0];
      if (!with_jacobians) continue;
      // corrector.cc
      const double sq1 = std::sqrt(rho[1]);
      double rscale = sq1, alpha_sq = 0.0;
      if (r.rb->loss && s != 0.0 && rho[2] > 0.0) {
        const double D = 1.0 + 2.0 * s * rho[2] / rho[1];
        const double alpha = 1.0 - std::sqrt(D);
        rscale = sq1 / (1.0 - alpha);
        alpha_sq = alpha / s;
      }
      r.J.resize(r.blk.size());
      for (size_t k = 0; k < r.blk.size(); ++k) {
        const Block& b = blocks_[r.blk[k]];
        const int sz = b.size;
        std::vector<double>& A = Ja[k];
        if (r.rb->loss) {
          if (alpha_sq != 0.0) {
            for (int c = 0; c < sz; ++c) {
              double rj = 0.0;
              for (int row = 0; row < r.nres; ++row) rj += r.r[row] * A[(size_t)row * sz + c];
              for (int row = 0; row < r.nres; ++row) A[(size_t)row * sz + c] = sq1 * (A[(size_t)row * sz + c] - alpha_sq * r.r[row] * rj);
            }
          } else {
            for (double& v : A) v *= sq1;
          }
        }
        // ambient -> tangent
        std::vector<double>& T = r.J[k];
        T.assign((size_t)r.nres * b.tsize, 0.0);
        const double* xb = &x[b.xoff];
        if (b.kind == 0) {  // EigenQuaternionManifold::PlusJacobian (4 x 3)
          const double z0 = xb[0], z1 = xb[1], z2 = xb[2], w = xb[3];
          const double P[12] = {w, z2, -z1, -z2, w, z0, z1, -z0, w, -z0, -z1, -z2};
          for (int row = 0; row < r.nres; ++row)
            for (int c = 0; c < 3; ++c) {
              double v = 0.0;
              for (int i = 0; i < 4; ++i) v += A[(size_t)row * 4 + i] * P[3 * i + c];
              T[(size_t)row * 3 + c] = v;
            }
        } else if (b.kind == 1) {
          for (int row = 0; row < r.nres; ++row)
            for (int c = 0; c < b.tsize; ++c) T[(size_t)row * b.tsize + c] = A[(size_t)row * sz + b.free_idx[c]];
        } else {
          T = A;
        }
      }
      if (r.rb->loss)
        for (double& v : r.r) v *= rscale;
    }
    if (with_jacobians) xeval_ = x;
    return total;
  }

  void Gradient() {  // g = J^T r of the last Evaluate(., true), tangent space, unscaled
    g_.assign(nt_, 0.0);
    for (const Res& r : res_)
      for (size_t k = 0; k < r.blk.size(); ++k) {
        const Block& b = blocks_[r.blk[k]];
        for (int row = 0; row < r.nres; ++row)
          for (int c = 0; c < b.tsize; ++c) g_[b.toff + c] += r.J[k][(size_t)row * b.tsize + c] * r.r[row];
      }
  }
  double GradMaxNorm() const {
    if (!constrained_) { double m = 0.0; for (double v : g_) m = std::max(m, std::fabs(v)); return m; }
    std::vector<double> ng(g_), out;
    for (double& v : ng) v = -v;
    Plus(xeval_, ng, &out);
    double m = 0.0;
    for (int i = 0; i < nx_; ++i) m = std::max(m, std::fabs(xeval_[i] - out[i]));
    return m;
  }

  // (Js^T Js + D) d = -Js^T r on the Jacobi-scaled system, solved by variable elimination + dense Cholesky; delta = scale * d
  struct Factor {
    std::vector<int> blk;
    std::vector<int> off;  // offsets of the blocks inside the factor
    int n = 0;
    std::vector<double> H, g;
    bool alive = true;
  };
  struct Elim {
    int e;
    std::vector<int> others, ooff;
    int ne, no;
    std::vector<double> L, Heo, ge;  // Cholesky factor of H_ee (+ damping), H_eo, g_e
  };

  bool Step(double radius, std::vector<double>* delta, double* model_change) {
    const int nb = (int)blocks_.size();
    std::vector<Factor> fac;
    fac.reserve(res_.size() * 2);
    std::vector<std::vector<int>> touching(nb);
    std::vector<double> diag(nt_, 0.0);
    for (const Res& r : res_) {
      if (r.blk.empty()) continue;
      Factor f;
      f.blk = r.blk;
      for (int b : f.blk) { f.off.push_back(f.n); f.n += blocks_[b].tsize; }
      // scaled Jacobian of the whole residual block: nres x n
      std::vector<double> J((size_t)r.nres * f.n);
      for (size_t k = 0; k < r.blk.size(); ++k) {
        const Block& b = blocks_[r.blk[k]];
        for (int row = 0; row < r.nres; ++row)
          for (int c = 0; c < b.tsize; ++c) J[(size_t)row * f.n + f.off[k] + c] = r.J[k][(size_t)row * b.tsize + c] * scale_[b.toff + c];
      }
      f.H.assign((size_t)f.n * f.n, 0.0);
      f.g.assign(f.n, 0.0);
      for (int row = 0; row < r.nres; ++row)
        for (int a = 0; a < f.n; ++a) {
          const double ja = J[(size_t)row * f.n + a];
          if (ja == 0.0) continue;
          f.g[a] += ja * r.r[row];
          for (int c = 0; c < f.n; ++c) f.H[(size_t)a * f.n + c] += ja * J[(size_t)row * f.n + c];
        }
      for (size_t k = 0; k < f.blk.size(); ++k)
        for (int c = 0; c < blocks_[f.blk[k]].tsize; ++c) diag[blocks_[f.blk[k]].toff + c] += f.H[(size_t)(f.off[k] + c) * f.n + f.off[k] + c];
      for (int b : f.blk) touching[b].push_back((int)fac.size());
      fac.push_back(std::move(f));
    }
    std::vector<double> damp(nt_);
    for (int j = 0; j < nt_; ++j) damp[j] = std::min(std::max(diag[j], o_.min_lm_diagonal), o_.max_lm_diagonal) / radius;
    std::vector<Elim> elims;
    elims.reserve(elim_.size());
    std::vector<char> gone(nb, 0);
    for (int e : elim_) {
      // union of the blocks of the live factors touching e
      std::vector<int> U{e};
      for (int fi : touching[e])
        if (fac[fi].alive)
          for (int b : fac[fi].blk)
            if (b != e && std::find(U.begin(), U.end(), b) == U.end()) U.push_back(b);
      std::sort(U.begin() + 1, U.end());
      std::vector<int> off(U.size());
      int n = 0;
      for (size_t k = 0; k < U.size(); ++k) { off[k] = n; n += blocks_[U[k]].tsize; }
      std::vector<double> M((size_t)n * n, 0.0), v(n, 0.0);
      for (int fi : touching[e]) {
        Factor& f = fac[fi];
        if (!f.alive) continue;
        std::vector<int> map(f.n);
        for (size_t k = 0; k < f.blk.size(); ++k) {
          const int pos = (int)(std::find(U.begin(), U.end(), f.blk[k]) - U.begin());
          for (int c = 0; c < blocks_[f.blk[k]].tsize; ++c) map[f.off[k] + c] = off[pos] + c;
        }
        for (int a = 0; a < f.n; ++a) {
          v[map[a]] += f.g[a];
          for (int c = 0; c < f.n; ++c) M[(size_t)map[a] * n + map[c]] += f.H[(size_t)a * f.n + c];
        }
        f.alive = false;
        f.H.clear();
        f.H.shrink_to_fit();
      }
      const int ne = blocks_[e].tsize, no = n - ne;
      for (int c = 0; c < ne; ++c) M[(size_t)c * n + c] += damp[blocks_[e].toff + c];
      Elim el;
      el.e = e;
      el.others.assign(U.begin() + 1, U.end());
      el.ooff.assign(off.begin() + 1, off.end());
      for (int& o : el.ooff) o -= ne;
      el.ne = ne;
      el.no = no;
      el.L.assign((size_t)ne * ne, 0.0);
      for (int a = 0; a < ne; ++a)
        for (int c = 0; c < ne; ++c) el.L[(size_t)a * ne + c] = M[(size_t)a * n + c];
      if (!cholesky(el.L, ne)) return false;
      el.Heo.assign((size_t)ne * no, 0.0);
      for (int a = 0; a < ne; ++a)
        for (int c = 0; c < no; ++c) el.Heo[(size_t)a * no + c] = M[(size_t)a * n + ne + c];
      el.ge.assign(v.begin(), v.begin() + ne);
      if (no > 0) {
        // W = H_ee^-1 [H_eo | g_e]
        std::vector<double> W((size_t)ne * (no + 1));
        for (int a = 0; a < ne; ++a) {
          for (int c = 0; c < no; ++c) W[(size_t)a * (no + 1) + c] = el.Heo[(size_t)a * no + c];
          W[(size_t)a * (no + 1) + no] = el.ge[a];
        }
        cholesky_solve(el.L, ne, W.data(), no + 1, no + 1);
        Factor nf;
        nf.blk = el.others;
        nf.off = el.ooff;
        nf.n = no;
        nf.H.assign((size_t)no * no, 0.0);
        nf.g.assign(no, 0.0);
        for (int a = 0; a < no; ++a) {
          double ga = v[ne + a];
          for (int k = 0; k < ne; ++k) ga -= el.Heo[(size_t)k * no + a] * W[(size_t)k * (no + 1) + no];
          nf.g[a] = ga;
          for (int c = 0; c < no; ++c) {
            double h = M[(size_t)(ne + a) * n + ne + c];
            for (int k = 0; k < ne; ++k) h -= el.Heo[(size_t)k * no + a] * W[(size_t)k * (no + 1) + c];
            nf.H[(size_t)a * no + c] = h;
          }
        }
        for (int b : nf.blk) touching[b].push_back((int)fac.size());
        fac.push_back(std::move(nf));
      }
      gone[e] = 1;
      elims.push_back(std::move(el));
    }
    // what remains: dense
    std::vector<int> rest, roff(nb, -1);
    int nr = 0;
    for (int b = 0; b < nb; ++b)
      if (!gone[b]) { roff[b] = nr; nr += blocks_[b].tsize; rest.push_back(b); }
    std::vector<double> d(nt_, 0.0);
    if (nr > 0) {
      std::vector<double> S((size_t)nr * nr, 0.0), rhs(nr, 0.0);
      for (const Factor& f : fac) {
        if (!f.alive) continue;
        std::vector<int> map(f.n);
        for (size_t k = 0; k < f.blk.size(); ++k)
          for (int c = 0; c < blocks_[f.blk[k]].tsize; ++c) map[f.off[k] + c] = roff[f.blk[k]] + c;
        for (int a = 0; a < f.n; ++a) {
          rhs[map[a]] -= f.g[a];
          for (int c = 0; c < f.n; ++c) S[(size_t)map[a] * nr + map[c]] += f.H[(size_t)a * f.n + c];
        }
      }
      for (int b : rest)
        for (int c = 0; c < blocks_[b].tsize; ++c) S[(size_t)(roff[b] + c) * nr + roff[b] + c] += damp[blocks_[b].toff + c];
      if (!cholesky(S, nr)) return false;
      cholesky_solve(S, nr, rhs.data(), 1, 1);
      for (int b : rest)
        for (int c = 0; c < blocks_[b].tsize; ++c) d[blocks_[b].toff + c] = rhs[roff[b] + c];
    }
    for (int i = (int)elims.size() - 1; i >= 0; --i) {  // back-substitution: d_e = -H_ee^-1 (g_e + H_eo d_o)
      const Elim& el = elims[i];
      std::vector<double> t(el.ge);
      for (size_t k = 0; k < el.others.size(); ++k) {
        const Block& b = blocks_[el.others[k]];
        for (int c = 0; c < b.tsize; ++c) {
          const double dv = d[b.toff + c];
          for (int a = 0; a < el.ne; ++a) t[a] += el.Heo[(size_t)a * el.no + el.ooff[k] + c] * dv;
        }
      }
      cholesky_solve(el.L, el.ne, t.data(), 1, 1);
      for (int a = 0; a < el.ne; ++a) d[blocks_[el.e].toff + a] = -t[a];
    }
    delta->resize(nt_);
    for (int j = 0; j < nt_; ++j) (*delta)[j] = d[j] * scale_[j];
    // model_cost_change = -(J delta) . (r + J delta / 2)
    double mc = 0.0;
    for (const Res& r : res_)
      for (int row = 0; row < r.nres; ++row) {
        double jd = 0.0;
        for (size_t k = 0; k < r.blk.size(); ++k) {
          const Block& b = blocks_[r.blk[k]];
          for (int c = 0; c < b.tsize; ++c) jd += r.J[k][(size_t)row * b.tsize + c] * (*delta)[b.toff + c];
        }
        mc -= jd * (r.r[row] + 0.5 * jd);
      }
    *model_change = mc;
    for (double v : *delta)
      if (!std::isfinite(v)) return false;
    return true;
  }

  const Solver::Options& o_;
  Problem* prob_;
  std::vector<Block> blocks_;
  std::vector<Res> res_;
  std::vector<int> elim_;
  std::vector<double> x_, xeval_, g_, scale_;
  int nx_ = 0, nt_ = 0;
  bool constrained_ = false;
};
}  // namespace shim

inline void Solve(const Solver::Options& options, Problem* problem, Solver::Summary* summary) {
  for (const auto& b : problem->residual_blocks())  // the start point, per block (read by the glue files)
    for (size_t i = 0; i < b.params.size(); ++i)
      if (!problem->start_values().count(b.params[i]))
        problem->start_values()[b.params[i]].assign(b.params[i], b.params[i] + b.cost->parameter_block_sizes()[i]);
  shim::Minimizer m(options, problem);
  m.Run(summary);
  LastSummary() = *summary;
}
}  // namespace ceres
