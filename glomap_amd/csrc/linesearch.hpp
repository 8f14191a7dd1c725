// linesearch.hpp — the projected Armijo line search Ceres' trust-region minimizer runs on bounds-constrained programs.
//
// GlobalPositioner puts a lower bound on every scale (glomap/estimators/global_positioning.cc:204,373), so
// `ceres::Solve` (gp.cc:83) sees Program::IsBoundsConstrained() and TrustRegionMinimizer::Minimize calls
// DoLineSearch(x, gradient, cost, &delta) on every valid LM step before it evaluates the candidate
// (ceres-solver 2.x internal/ceres/trust_region_minimizer.cc, line_search.cc, polynomial.cc — un-vendored; restated):
//
//   phi(t) = cost(Plus(x, t delta))  with Plus projecting on the bounds;   phi'(t) := delta . gradient(Plus(x, t delta))
//   ArmijoLineSearch, first trial t = 1, accepted when  phi(t) <= phi(0) + 1e-4 t phi'(0);
//   otherwise t <- argmin over [1e-3 t, 0.6 t] of the polynomial interpolating value AND slope at 0, at the current trial
//   and (from the second contraction on) at the previous trial: a cubic, then quintics (CUBIC interpolation type);
//   MinimizePolynomial looks at the midpoint, the two ends and the critical points inside the interval, in that order,
//   each taking over only when strictly lower;  at most 20 trials;  failure when t |delta|_inf < 1e-9.
//   Success: delta *= t.  Failure: delta stays the full step.  model_cost_change is the full step's in both cases.
//
// Host code, a few dozen flops per LM iteration; the trial evaluations are device sweeps behind LmProblem (lm.hpp).
#pragma once

#include <algorithm>
#include <cmath>
#include <vector>

namespace gsfm {
namespace ls {

struct Sample {
  double t = 0.0, value = 0.0, slope = 0.0;
  bool valid = false;
};

// polynomials: coefficients highest power first
inline double peval(const std::vector<double>& p, double x) {
  double v = 0.0;
  for (double a : p) v = v * x + a;
  return v;
}

inline std::vector<double> pderiv(const std::vector<double>& p) {
  const int deg = (int)p.size() - 1;
  std::vector<double> d;
  d.reserve(std::max(1, deg));
  for (int j = 0; j < deg; ++j) d.push_back((double)(deg - j) * p[j]);
  if (d.empty()) d.push_back(0.0);
  return d;
}

inline std::vector<double> pstrip(std::vector<double> p) {  // RemoveLeadingZeros: exact zeros only
  size_t i = 0;
  while (i + 1 < p.size() && p[i] == 0.0) ++i;
  p.erase(p.begin(), p.begin() + (long)i);
  return p;
}

// FindInterpolatingPolynomial: value and slope rows per sample, solved with full pivoting (Eigen::FullPivLU in Ceres)
inline std::vector<double> interpolate(const Sample* smp, int ns) {
  const int n = 2 * ns, deg = n - 1, ld = n + 1;
  std::vector<double> A((size_t)n * ld, 0.0);
  int row = 0;
  for (int i = 0; i < ns; ++i) {
    for (int j = 0; j <= deg; ++j) A[(size_t)row * ld + j] = std::pow(smp[i].t, deg - j);
    A[(size_t)row * ld + n] = smp[i].value;
    ++row;
    for (int j = 0; j < deg; ++j) A[(size_t)row * ld + j] = (double)(deg - j) * std::pow(smp[i].t, deg - j - 1);
    A[(size_t)row * ld + n] = smp[i].slope;
    ++row;
  }
  std::vector<int> perm(n);
  for (int i = 0; i < n; ++i) perm[i] = i;
  for (int k = 0; k < n; ++k) {
    int pr = k, pc = k;
    double best = -1.0;
    for (int r = k; r < n; ++r)
      for (int c = k; c < n; ++c) {
        const double a = std::fabs(A[(size_t)r * ld + c]);
        if (a > best) best = a, pr = r, pc = c;
      }
    if (!(best > 0.0)) break;
    if (pr != k)
      for (int c = 0; c < ld; ++c) std::swap(A[(size_t)pr * ld + c], A[(size_t)k * ld + c]);
    if (pc != k) {
      for (int r = 0; r < n; ++r) std::swap(A[(size_t)r * ld + pc], A[(size_t)r * ld + k]);
      std::swap(perm[pc], perm[k]);
    }
    for (int r = k + 1; r < n; ++r) {
      const double f = A[(size_t)r * ld + k] / A[(size_t)k * ld + k];
      if (f == 0.0) continue;
      for (int c = k; c < ld; ++c) A[(size_t)r * ld + c] -= f * A[(size_t)k * ld + c];
    }
  }
  std::vector<double> y(n, 0.0), coef(n, 0.0);
  for (int k = n - 1; k >= 0; --k) {
    double v = A[(size_t)k * ld + n];
    for (int c = k + 1; c < n; ++c) v -= A[(size_t)k * ld + c] * y[c];
    const double piv = A[(size_t)k * ld + k];
    y[k] = piv != 0.0 ? v / piv : 0.0;
  }
  for (int k = 0; k < n; ++k) coef[perm[k]] = y[k];
  return coef;
}

// real roots of p in [lo, hi], ascending.  Between two consecutive critical points p is monotone, so a sign change
// brackets exactly one root: bisection to the last bit.  (Ceres takes companion-matrix eigenvalues and then discards
// what lies outside the interval; complex pairs contribute their real part, an interior point that can never undercut
// the candidates containing the interval's minimiser.)
inline std::vector<double> roots_in(const std::vector<double>& p_in, double lo, double hi) {
  const std::vector<double> p = pstrip(p_in);
  std::vector<double> roots;
  const int deg = (int)p.size() - 1;
  if (deg <= 0) return roots;
  if (deg == 1) {
    const double r = -p[1] / p[0];
    if (r >= lo && r <= hi) roots.push_back(r);
    return roots;
  }
  std::vector<double> brk;
  brk.push_back(lo);
  for (double c : roots_in(pderiv(p), lo, hi)) brk.push_back(c);
  brk.push_back(hi);
  for (size_t i = 0; i + 1 < brk.size(); ++i) {
    double a = brk[i], b = brk[i + 1];
    double fa = peval(p, a);
    const double fb = peval(p, b);
    if (fa == 0.0) {
      if (roots.empty() || roots.back() != a) roots.push_back(a);
      continue;
    }
    if (fb == 0.0) {
      if (i + 2 == brk.size()) roots.push_back(b);
      continue;
    }
    if ((fa < 0.0) == (fb < 0.0)) continue;
    for (int it = 0; it < 200; ++it) {
      const double m = 0.5 * (a + b);
      if (!(m > a && m < b)) break;
      const double fm = peval(p, m);
      if (fm == 0.0) {
        a = b = m;
        break;
      }
      if ((fm < 0.0) == (fa < 0.0)) {
        a = m;
        fa = fm;
      } else {
        b = m;
      }
    }
    roots.push_back(0.5 * (a + b));
  }
  return roots;
}

inline double minimize_on(const std::vector<double>& poly, double x_min, double x_max) {
  double best_x = 0.5 * (x_min + x_max);
  double best_v = peval(poly, best_x);
  const double ends[2] = {x_min, x_max};
  for (double x : ends) {
    const double v = peval(poly, x);
    if (v < best_v) best_x = x, best_v = v;
  }
  const std::vector<double> p = pstrip(poly);
  if (p.size() <= 2) return best_x;
  for (double x : roots_in(pderiv(p), x_min, x_max)) {
    const double v = peval(poly, x);
    if (v < best_v) best_x = x, best_v = v;
  }
  return best_x;
}

struct Options {
  int max_num_iterations = 20;          // Solver::Options::max_num_line_search_step_size_iterations
  double sufficient_decrease = 1e-4;    // line_search_sufficient_function_decrease
  double max_step_contraction = 1e-3;   // max_line_search_step_contraction
  double min_step_contraction = 0.6;    // min_line_search_step_contraction
  double min_step_size = 1e-9;          // min_line_search_step_size
};

struct Result {
  bool success = false;
  double t = 1.0;
  int trials = 0;  // evaluations of phi, the one at t = 1 included
};

// `first` = the trial at t = 1 (the LM step itself, already evaluated by the caller); eval(t, &value, &slope) evaluates
// another trial and leaves the problem's candidate at that trial.
template <class Eval>
Result armijo(const Sample& first, double cost0, double slope0, double direction_max_norm, const Options& o, Eval eval) {
  Result res;
  const Sample lower{0.0, cost0, slope0, true};
  Sample previous, cur = first;
  int iterations = 0;
  res.trials = 1;
  while (!cur.valid || cur.value > cost0 + o.sufficient_decrease * slope0 * cur.t) {
    if (++iterations >= o.max_num_iterations) return res;
    const double x_lo = o.max_step_contraction * cur.t, x_hi = o.min_step_contraction * cur.t;
    double t;
    if (!cur.valid) {
      t = std::min(std::max(cur.t * 0.5, x_lo), x_hi);
    } else {
      Sample smp[3] = {lower, cur, previous};
      t = minimize_on(interpolate(smp, previous.valid ? 3 : 2), x_lo, x_hi);
    }
    if (t * direction_max_norm < o.min_step_size) return res;
    previous = cur;
    cur = Sample{};
    cur.t = t;
    eval(t, &cur.value, &cur.slope);
    cur.valid = std::isfinite(cur.value) && std::isfinite(cur.slope);
    ++res.trials;
  }
  res.success = true;
  res.t = cur.t;
  return res;
}

}  // namespace ls
}  // namespace gsfm
