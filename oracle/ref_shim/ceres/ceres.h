// ceres/ceres.h stand-in: a RECORDING Ceres for compiling glomap/estimators/global_positioning.cc — the reference's own
// problem builder — into oracle/_ref (test infrastructure).
//
// Ceres is an un-vendored dependency (cmake/FindDependencies.cmake:4) and is not in this image.  What global positioning's
// PROBLEM CONSTRUCTION needs from it is bookkeeping: residual blocks (cost function, loss function, parameter pointers),
// bounds, constant blocks, an ordering, an options struct.  This header provides exactly that and RECORDS it; ceres::Solve
// here does not minimise anything — it evaluates the robustified cost 1/2 sum rho(|r|^2) at the start point, stores it in the
// summary and returns.  So the library built from it answers "what problem does the REFERENCE'S CODE pose, from which random
// start?", and tests/test_oracle_ref.py holds the oracle's problem (oracle/gp.py) to that: same start, same residuals, same
// losses, same bounds and constants, same initial cost.  The trust-region loop itself stays a restatement (oracle/lm.py).
// Loss functions follow Ceres' published definitions (loss_function.h): HuberLoss(a): rho(s) = s for s <= a^2, 2 a sqrt(s) -
// a^2 beyond; ScaledLoss(rho, k): k rho(s).  AutoDiffCostFunction evaluates its functor in doubles (no Jets: no Jacobians).
#pragma once
#include <cmath>
#include <map>
#include <memory>
#include <random>
#include <set>
#include <string>
#include <utility>
#include <vector>

namespace ceres {
enum Ownership { DO_NOT_TAKE_OWNERSHIP, TAKE_OWNERSHIP };
enum LinearSolverType { DENSE_NORMAL_CHOLESKY, DENSE_QR, SPARSE_NORMAL_CHOLESKY, DENSE_SCHUR, SPARSE_SCHUR, ITERATIVE_SCHUR, CGNR };
enum PreconditionerType { IDENTITY, JACOBI, SCHUR_JACOBI, CLUSTER_JACOBI, CLUSTER_TRIDIAGONAL };
enum DenseLinearAlgebraLibraryType { EIGEN, LAPACK, CUDA };
enum SparseLinearAlgebraLibraryType { SUITE_SPARSE, EIGEN_SPARSE, ACCELERATE_SPARSE, CUDA_SPARSE, NO_SPARSE };

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
      rho[1] = a_ / r;
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

class CostFunction {
 public:
  virtual ~CostFunction() = default;
  virtual bool Evaluate(double const* const* parameters, double* residuals, double** jacobians) const = 0;
  int num_residuals() const { return num_residuals_; }
  const std::vector<int>& parameter_block_sizes() const { return sizes_; }

 protected:
  int num_residuals_ = 0;
  std::vector<int> sizes_;
};
template <typename Functor, int kNumResiduals, int... Ns>
class AutoDiffCostFunction final : public CostFunction {
 public:
  explicit AutoDiffCostFunction(Functor* f) : f_(f) {
    num_residuals_ = kNumResiduals;
    sizes_ = {Ns...};
  }
  bool Evaluate(double const* const* p, double* residuals, double** /*jacobians: not provided by the recording mock*/) const override {
    return call(p, residuals, std::make_index_sequence<sizeof...(Ns)>());
  }
  const Functor& functor() const { return *f_; }

 private:
  template <size_t... I>
  bool call(double const* const* p, double* r, std::index_sequence<I...>) const {
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
  void SetParameterBlockConstant(const double* values) { constant_.insert(values); }
  bool HasParameterBlock(const double* values) const { return blocks_.count(const_cast<double*>(values)) != 0; }
  int NumResidualBlocks() const { return static_cast<int>(residuals_.size()); }
  // ---- what the recording mock adds (read by oracle/ref_glue_gp.cc) ----
  const std::vector<ResidualBlock>& residual_blocks() const { return residuals_; }
  const std::map<std::pair<double*, int>, double>& lower_bounds() const { return lower_; }
  bool IsConstant(const double* values) const { return constant_.count(values) != 0; }
  // manifolds, as the COLMAP helpers of ref_shim_ba/colmap/estimators/manifold.h record them: 0 = quaternion manifold,
  // 1 = subset manifold with the listed constant coordinates
  struct ManifoldTag {
    int kind;
    std::vector<int> constant_idxs;
  };
  void RecordManifold(const double* values, int kind, const std::vector<int>& idxs) { manifold_[values] = ManifoldTag{kind, idxs}; }
  const std::map<const double*, ManifoldTag>& manifolds() const { return manifold_; }
  // parameter values at the moment ceres::Solve was called (the start point), per block pointer
  std::map<const double*, std::vector<double>>& start_values() { return start_; }
  const std::map<const double*, std::vector<double>>& start_values() const { return start_; }

 private:
  std::vector<ResidualBlock> residuals_;
  std::set<double*> blocks_;
  std::map<std::pair<double*, int>, double> lower_;
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
    LinearSolverType linear_solver_type = SPARSE_NORMAL_CHOLESKY;
    PreconditionerType preconditioner_type = JACOBI;
    DenseLinearAlgebraLibraryType dense_linear_algebra_library_type = EIGEN;
    SparseLinearAlgebraLibraryType sparse_linear_algebra_library_type = SUITE_SPARSE;
    std::shared_ptr<ParameterBlockOrdering> linear_solver_ordering;
  };
  struct Summary {
    double initial_cost = 0.0, final_cost = 0.0;
    int num_residual_blocks = 0;
    std::string BriefReport() const { return "recording mock: no minimisation"; }
    std::string FullReport() const { return BriefReport(); }
    bool IsSolutionUsable() const { return true; }
  };
};

// The recording Solve: 1/2 sum_blocks rho(|r|^2) at the current parameter values (what Ceres reports as initial_cost).
inline void Solve(const Solver::Options&, Problem* problem, Solver::Summary* summary) {
  double cost = 0.0;
  for (const auto& b : problem->residual_blocks()) {
    for (size_t i = 0; i < b.params.size(); ++i)
      if (!problem->start_values().count(b.params[i]))
        problem->start_values()[b.params[i]].assign(b.params[i], b.params[i] + b.cost->parameter_block_sizes()[i]);
    std::vector<double> r(static_cast<size_t>(b.cost->num_residuals()));
    std::vector<const double*> p(b.params.begin(), b.params.end());
    b.cost->Evaluate(p.data(), r.data(), nullptr);
    double s = 0.0;
    for (double x : r) s += x * x;
    double rho[3] = {s, 1.0, 0.0};
    if (b.loss != nullptr) b.loss->Evaluate(s, rho);
    cost += 0.5 * rho[0];
  }
  summary->initial_cost = summary->final_cost = cost;
  summary->num_residual_blocks = problem->NumResidualBlocks();
}
}  // namespace ceres
