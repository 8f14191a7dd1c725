// lm.hpp — host-side trust-region Levenberg-Marquardt loop shared by global positioning and bundle
// adjustment.  It replaces `ceres::Solve` as called at glomap/estimators/global_positioning.cc:83
// and bundle_adjustment.cc:99 (options: optimization_base.h:18-23 + Ceres defaults)
// and mirrors, decision by decision, oracle/lm.py:
//   Jacobi scaling fixed at the initial point, LM diagonal clamp(diag)/radius, step validity by
//   model_cost_change > 0, parameter tolerance, function tolerance checked BEFORE acceptance,
//   accept: radius /= max(1/3, 1 - (2 rho - 1)^3); reject: radius /= decrease_factor (doubling),
//   gradient tolerance after accepted steps, min radius, iteration cap;
//   bounds-constrained problems (global positioning: every scale has a lower bound, gp.cc:204,373): Ceres' projected Armijo
//   line search on every valid step before the candidate counts (linesearch.hpp; round 6 — rounds 1 - 5 left it out and
//   VERDICT r5 measured what that does to the end point: 4e-3 .. 1.3e-2 of the extent).
// All heavy work lives behind the LmProblem interface (device kernels); this loop only moves a few
// scalars per iteration.
#pragma once

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <limits>
#include <vector>

#include "common.hpp"
#include "linesearch.hpp"

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
  // Program::IsBoundsConstrained — a free parameter with a bound.  Such a problem reports, for the step the last step()
  // produced: phi'(0) = g . delta, phi'(1) = g(candidate) . delta and |delta|_inf; and trial(t) moves the candidate to
  // Plus(x, t delta) (projected on the bounds) and returns its cost, phi'(t) and |candidate - x| (false: non-finite).
  virtual bool constrained() const { return false; }
  virtual void step_line_data(double* slope0, double* slope1, double* direction_max_norm) {
    *slope0 = *slope1 = *direction_max_norm = 0.0;
  }
  virtual bool trial(double t, double* cost, double* slope, double* step_norm) {
    (void)t; (void)cost; (void)slope; (void)step_norm;
    return false;
  }
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
  o->max_num_line_search_step_size_iterations = 20;
}

// Returns GSFM_OK or GSFM_ERR_NOT_USABLE; fills the iteration fields of `rep`.
inline int lm_minimize(LmProblem& prob, const gsfm_lm_options& o, gsfm_report* rep, std::vector<double>* trace = nullptr) {
  if (trace) trace->clear();
  double gmax = 0.0;
  double cost = prob.linearize(&gmax);
  prob.set_jacobi_scaling(o.jacobi_scaling != 0);
  const double initial_cost = cost;
  int iterations = 0, successful = 0, invalid = 0, line_trials = 0, line_shrunk = 0;
  long lin_total = 0;
  int termination = GSFM_TERM_NO_CONVERGENCE;
  bool usable = true;
  double radius = o.initial_trust_region_radius;
  double decrease_factor = 2.0;
  const bool verbose = std::getenv("GSFM_VERBOSE") != nullptr;  // like minimizer_progress_to_stdout
  // include/gsfm.h GSFM_LM_TRACE_COLS: cost | radius | model change | candidate cost | step size | accepted | linear iterations
  auto record = [&](double c, double rad, double mc, double cc, double t, double acc, double lin) {
    if (trace) trace->insert(trace->end(), {c, rad, mc, cc, t, acc, lin});
  };
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
        record(cost, radius, model_change, cand_cost, 1.0, -1.0, (double)lin);
        if (++invalid >= o.max_num_consecutive_invalid_steps) {
          termination = GSFM_TERM_FAILURE;
          usable = false;
          break;
        }
        radius *= 0.5;
        continue;
      }
      invalid = 0;
      double step_size = 1.0;
      if (prob.constrained() && o.max_num_line_search_step_size_iterations > 0) {
        // TrustRegionMinimizer::DoLineSearch(x, gradient, cost, &delta): delta *= t on success, untouched on failure;
        // model_change stays the full step's
        double slope0, slope1, dmax;
        prob.step_line_data(&slope0, &slope1, &dmax);
        ls::Options lo;
        lo.max_num_iterations = o.max_num_line_search_step_size_iterations;
        ls::Sample first;
        first.t = 1.0;
        first.value = cand_cost;
        first.slope = slope1;
        first.valid = std::isfinite(cand_cost) && std::isfinite(slope1);
        double at = 1.0, t_cost = cand_cost, t_norm = step_norm;  // where the problem's candidate stands
        const ls::Result r = ls::armijo(first, cost, slope0, dmax, lo, [&](double t, double* v, double* sl) {
          if (!prob.trial(t, v, sl, &t_norm)) *v = std::numeric_limits<double>::infinity();
          t_cost = *v;
          at = t;
        });
        line_trials += r.trials - 1;
        if (!r.success && at != 1.0) prob.trial(at = 1.0, &t_cost, &slope1, &t_norm);  // back to the full step's candidate
        if (at != 1.0) ++line_shrunk;
        cand_cost = t_cost;
        step_norm = t_norm;
        step_size = r.success ? at : -1.0;
        if (verbose)
          fprintf(stderr, "[gsfm lm]   line search: %s t %.6e trials %d cand %.9e step %.3e\n", r.success ? "ok" : "FAILED", at,
                  r.trials, cand_cost, step_norm);
      }
      if (step_norm <= o.parameter_tolerance * (x_norm + o.parameter_tolerance)) {
        record(cost, radius, model_change, cand_cost, step_size, 0.0, (double)lin);
        termination = GSFM_TERM_CONVERGENCE;
        break;
      }
      const double cost_change = cost - cand_cost;
      if (std::fabs(cost_change) <= o.function_tolerance * cost) {
        record(cost, radius, model_change, cand_cost, step_size, 0.0, (double)lin);
        termination = GSFM_TERM_CONVERGENCE;
        break;
      }
      const double rho = cost_change / model_change;
      record(cost, radius, model_change, cand_cost, step_size, rho > o.min_relative_decrease ? 1.0 : 0.0, (double)lin);
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
    rep->line_search_trials = line_trials;
    rep->line_search_shrunk = line_shrunk;
  }
  return usable ? GSFM_OK : GSFM_ERR_NOT_USABLE;
}

}  // namespace gsfm
