// lm.hpp — host-side trust-region Levenberg-Marquardt loop shared by global positioning and bundle
// adjustment.  It replaces `ceres::Solve` as called at glomap/estimators/global_positioning.cc:83
// and bundle_adjustment.cc:99 (options: optimization_base.h:18-23 + Ceres defaults)
// and mirrors, decision by decision, oracle/lm.py:
//   Jacobi scaling fixed at the initial point, LM diagonal clamp(diag)/radius, step validity by
//   model_cost_change > 0, parameter tolerance, function tolerance checked BEFORE acceptance,
//   accept: radius /= max(1/3, 1 - (2 rho - 1)^3); reject: radius /= decrease_factor (doubling),
//   gradient tolerance after accepted steps, min radius, iteration cap.
// All heavy work lives behind the LmProblem interface (device kernels); this loop only moves a few
// scalars per iteration.
#pragma once

#include <algorithm>
#include <cmath>
#include <cstdlib>

#include "common.hpp"

namespace gsfm {

struct LmProblem {
  virtual ~LmProblem() = default;
  // Evaluate cost, robust weights, gradient and squared column norms at the current point.
  // Returns the cost; *grad_max_norm receives max |J^T r|.
  virtual double linearize(double* grad_max_norm) = 0;
  // Fix the Jacobi scaling from the column norms of the last linearize() (or identity).
  virtual void set_jacobi_scaling(bool enabled) = 0;
  // Build the damped normal equations for `radius`, solve them, form the candidate point.
  // Outputs: model_cost_change, candidate cost, |candidate - x|, |x|.  Returns false when the step
  // contains non-finite values.
  virtual bool step(double radius, double* model_change, double* cand_cost, double* step_norm,
                    double* x_norm, long* linear_iterations) = 0;
  // Make the candidate the current point.
  virtual void accept() = 0;
  // A problem may finish the linearisation of a newly accepted point inside the first step() that follows (global
  // positioning folds the camera half into its build sweep: same gathers).  gradient_pending(): the max-norm linearize()
  // returned is incomplete, so the gradient-tolerance test waits; take_pending_gradient(): after step(), the complete
  // max-norm of that point when it became known in this call.
  virtual bool gradient_pending() const { return false; }
  virtual bool take_pending_gradient(double* grad_max_norm) { (void)grad_max_norm; return false; }
  // The loop is about to end (iteration cap, minimum radius) while gradient_pending(): complete the linearisation of the
  // accepted point now, without a step, so that its gradient test is not lost (Ceres tests it right after the acceptance
  // and would report CONVERGENCE).  Returns true when *grad_max_norm was produced.
  virtual bool finish_pending_gradient(double* grad_max_norm) { (void)grad_max_norm; return false; }
};

inline void lm_options_default(gsfm_lm_options* o, int max_iterations) {
  o->max_num_iterations = max_iterations;
  o->function_tolerance = 1e-5;
  o->gradient_tolerance = 1e-10;
  o->parameter_tolerance = 1e-8;
  o->initial_trust_region_radius = 1e4;
  o->max_trust_region_radius = 1e16;
  o->min_trust_region_radius = 1e-32;
  o->min_relative_decrease = 1e-3;
  o->min_lm_diagonal = 1e-6;
  o->max_lm_diagonal = 1e32;
  o->jacobi_scaling = 1;
  o->max_num_consecutive_invalid_steps = 5;
  o->pcg_relative_tolerance = 1e-8;
  o->pcg_max_iterations = 1000;
}

// Returns GSFM_OK or GSFM_ERR_NOT_USABLE; fills the iteration fields of `rep`.
inline int lm_minimize(LmProblem& prob, const gsfm_lm_options& o, gsfm_report* rep) {
  double gmax = 0.0;
  double cost = prob.linearize(&gmax);
  prob.set_jacobi_scaling(o.jacobi_scaling != 0);
  const double initial_cost = cost;
  int iterations = 0, successful = 0, invalid = 0;
  long lin_total = 0;
  int termination = GSFM_TERM_NO_CONVERGENCE;
  bool usable = true;
  double radius = o.initial_trust_region_radius;
  double decrease_factor = 2.0;
  const bool verbose = std::getenv("GSFM_VERBOSE") != nullptr;  // like minimizer_progress_to_stdout
  if (verbose) fprintf(stderr, "[gsfm lm] it 0 cost %.9e gmax %.3e\n", cost, gmax);
  if (!(gmax > o.gradient_tolerance)) {
    termination = GSFM_TERM_CONVERGENCE;
  } else {
    while (true) {
      if (iterations >= o.max_num_iterations) {
        termination = GSFM_TERM_NO_CONVERGENCE;
        if (prob.gradient_pending() && prob.finish_pending_gradient(&gmax) && !(gmax > o.gradient_tolerance))
          termination = GSFM_TERM_CONVERGENCE;  // the point accepted last satisfies the gradient tolerance
        break;
      }
      if (radius < o.min_trust_region_radius) {
        termination = GSFM_TERM_CONVERGENCE;
        break;
      }
      ++iterations;
      double model_change = 0.0, cand_cost = 0.0, step_norm = 0.0, x_norm = 0.0;
      long lin = 0;
      bool valid = prob.step(radius, &model_change, &cand_cost, &step_norm, &x_norm, &lin);
      if (prob.take_pending_gradient(&gmax) && !(gmax > o.gradient_tolerance)) {
        // the gradient test of the point accepted in the previous iteration (trust_region_minimizer.cc tests it right after
        // the acceptance): the step just computed is not taken and does not count, nor do its linear iterations
        --iterations;
        termination = GSFM_TERM_CONVERGENCE;
        break;
      }
      lin_total += lin;
      if (verbose)
        fprintf(stderr, "[gsfm lm] it %d radius %.3e pcg %ld model %.6e cand %.9e (cost %.9e) step %.3e\n", iterations,
                radius, lin, model_change, cand_cost, cost, step_norm);
      valid = valid && std::isfinite(model_change) && model_change > 0.0;
      if (!valid) {
        // Ceres fails on the max_num_consecutive_invalid_steps-th consecutive invalid step (trust_region_minimizer.cc)
        if (++invalid >= o.max_num_consecutive_invalid_steps) {
          termination = GSFM_TERM_FAILURE;
          usable = false;
          break;
        }
        radius *= 0.5;
        continue;
      }
      invalid = 0;
      if (step_norm <= o.parameter_tolerance * (x_norm + o.parameter_tolerance)) {
        termination = GSFM_TERM_CONVERGENCE;
        break;
      }
      const double cost_change = cost - cand_cost;
      if (std::fabs(cost_change) <= o.function_tolerance * cost) {
        termination = GSFM_TERM_CONVERGENCE;
        break;
      }
      const double rho = cost_change / model_change;
      if (rho > o.min_relative_decrease) {
        prob.accept();
        cost = prob.linearize(&gmax);
        ++successful;
        if (!prob.gradient_pending() && !(gmax > o.gradient_tolerance)) {
          termination = GSFM_TERM_CONVERGENCE;
          break;
        }
        const double t = 2.0 * rho - 1.0;
        radius = radius / std::max(1.0 / 3.0, 1.0 - t * t * t);
        radius = std::min(o.max_trust_region_radius, radius);
        decrease_factor = 2.0;
      } else {
        radius = radius / decrease_factor;
        decrease_factor *= 2.0;
      }
    }
  }
  if (rep) {
    rep->iterations = iterations;
    rep->successful_steps = successful;
    rep->linear_iterations = lin_total;
    rep->initial_cost = initial_cost;
    rep->final_cost = cost;
    rep->termination = termination;
  }
  return usable ? GSFM_OK : GSFM_ERR_NOT_USABLE;
}

}  // namespace gsfm
