// colmap::LeastAbsoluteDeviationSolver is part of the un-vendored COLMAP dependency (COLMAP @ b6b7b54e,
// src/colmap/optim/least_absolute_deviations.{h,cc}): restated from its published algorithm — ADMM for min |A x - b|_1
// (Boyd et al., section 6.1) with one Cholesky factor of A^T A, the same statement as oracle/ra.py's
// LeastAbsoluteDeviationSolver, so that the reference's rotation averaging can be compiled around it.  The options are COPIED
// at construction, as COLMAP's class does (`const Options options_`): global_rotation_averaging.cc:536-537 doubles
// l1_solver_options.max_num_iterations AFTER the solver was built, which therefore has no effect on the solver.
#pragma once
#include <algorithm>
#include <cmath>

#include "ref_shim_linalg.h"

namespace colmap {
class LeastAbsoluteDeviationSolver {
 public:
  struct Options {
    double rho = 1.0;
    double alpha = 1.0;
    int max_num_iterations = 1000;
    double absolute_tolerance = 1e-4;
    double relative_tolerance = 1e-2;
    enum class SolverType { SimplicialLLT, SupernodalCholmodLLT };
    SolverType solver_type = SolverType::SimplicialLLT;
  };
  LeastAbsoluteDeviationSolver(const Options& options, const Eigen::SparseMatrix<double>& A) : options_(options), A_(A), At_(A.transpose()) {
    llt_.compute(At_ * A_);
    --ref_shim::ra_counters().llt_factorizations;  // (the counter is for the IRLS loop's factorisations)
  }
  bool Solve(const Eigen::VectorXd& b, Eigen::VectorXd* x) const {
    ++ref_shim::ra_counters().lad_solves;
    const size_t m = static_cast<size_t>(A_.rows()), n = static_cast<size_t>(A_.cols());
    Eigen::VectorXd z(m), z_old(m), u(m);
    const double rhs_norm = b.norm();
    const double primal_abs = std::sqrt(static_cast<double>(m)) * options_.absolute_tolerance;
    const double dual_abs = std::sqrt(static_cast<double>(n)) * options_.absolute_tolerance;
    for (int it = 0; it < options_.max_num_iterations; ++it) {
      ++ref_shim::ra_counters().lad_admm_iterations;
      *x = llt_.solve(At_ * (b + z - u));
      const Eigen::VectorXd Ax = A_ * *x;
      const Eigen::VectorXd Ax_hat = options_.alpha * Ax + (1.0 - options_.alpha) * (z + b);
      z_old = z;
      const Eigen::VectorXd v = Ax_hat - b + u;
      const double kappa = 1.0 / options_.rho;
      for (size_t i = 0; i < m; ++i) z.a[i] = std::max(0.0, v.a[i] - kappa) - std::max(0.0, -v.a[i] - kappa);
      u += Ax_hat - z - b;
      const double r_norm = (Ax - z - b).norm();
      const double s_norm = (-options_.rho * (At_ * (z - z_old))).norm();
      const double max_norm = std::max(std::max(Ax.norm(), z.norm()), rhs_norm);
      const double primal_eps = primal_abs + options_.relative_tolerance * max_norm;
      const double dual_eps = dual_abs + options_.relative_tolerance * (options_.rho * (At_ * u)).norm();
      if (r_norm < primal_eps && s_norm < dual_eps) break;
    }
    return true;
  }

 private:
  const Options options_;
  Eigen::SparseMatrix<double> A_, At_;
  Eigen::CholmodSupernodalLLT<Eigen::SparseMatrix<double>> llt_;
};
}  // namespace colmap
