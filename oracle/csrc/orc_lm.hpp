// ORACLE (test infrastructure only — never linked into or called by the product path).
//
// orc_lm.hpp — the Ceres trust-region Levenberg-Marquardt loop that GlobalPositioner and
// BundleAdjuster run through ceres::Solve (glomap/estimators/global_positioning.cc:83,
// bundle_adjustment.cc:99; options optimization_base.h:18-23 + Ceres defaults),
// restated decision by decision exactly as oracle/lm.py states it (that file's header lists the
// Ceres sources followed).  The linear system of every step is solved "exactly": the problem
// eliminates its independent blocks (GP: scales, then points; BA: points) in closed form and
// solves the reduced camera system DIRECTLY (dense Cholesky of the explicitly assembled matrix)
// when it has at most kDenseMax unknowns — like oracle/lm.py — and by preconditioned CG to a
// relative residual of 1e-14 (or until the residual stops decreasing) above that.
//
// parity unpinned: no reference test pins LM iterates.
#pragma once

#include "orc_common.hpp"

namespace orc {

// (experiment, see pcg below)
struct RecycleStore {
  i64 n = 0;
  double radius = 0.0;  // trust-region radius of the solve about to run (set by the LM loop)
  std::vector<std::vector<double>> U;
  std::vector<double> theta, rad;  // Ritz value and the radius it was harvested at
  std::vector<int> age;
};
inline RecycleStore& recycle_store() {
  static RecycleStore s;
  return s;
}

struct LmOptions {
  int max_num_iterations = 100;
  double function_tolerance = 1e-5;
  double gradient_tolerance = 1e-10;
  double parameter_tolerance = 1e-8;
  double initial_trust_region_radius = 1e4;
  double max_trust_region_radius = 1e16;
  double min_trust_region_radius = 1e-32;
  double min_relative_decrease = 1e-3;
  double min_lm_diagonal = 1e-6;
  double max_lm_diagonal = 1e32;
  int jacobi_scaling = 1;
  int max_num_consecutive_invalid_steps = 5;
  // TrustRegionMinimizer::DoLineSearch on bounds-constrained programs (Solver::Options defaults; oracle/lm.py header)
  int line_search = 1;  // 0: the loop of rounds 1 - 5 (experiments only)
  int max_num_line_search_step_size_iterations = 20;
  double line_search_sufficient_function_decrease = 1e-4;
  double max_line_search_step_contraction = 1e-3;
  double min_line_search_step_contraction = 0.6;
  double min_line_search_step_size = 1e-9;
  // reduced-system solve (replaces the exact factorisation)
  double pcg_relative_tolerance = 1e-14;
  int pcg_max_iterations = 20000;
  int order = 0;    // 1: reverse the summation order of every owner-side reduction (rounding experiments)
  int verbose = 0;
};

struct LmSummary {
  int iterations = 0;
  int successful_steps = 0;
  i64 linear_iterations = 0;
  double initial_cost = 0.0, final_cost = 0.0;
  int termination = 1;  // 0 convergence, 1 no convergence (iteration cap), 2 failure
  int usable = 1;
  double max_linear_residual = 0.0;  // largest TRUE relative residual |b - S x| / |b| over all linear solves
  double seconds_linear = 0.0;
  int line_search_steps = 0;   // Armijo trials beyond the first
  int line_search_shrunk = 0;  // LM iterations whose step the search shortened
};

// the LM iterations of the last lm_minimize of this process, one row of 7 doubles per iteration (same columns as the
// product's gsfm_ctx_lm_trace: cost | radius | model change | candidate cost | step size | accepted | linear iterations);
// read through orc_lm_trace (orc_gp.cc) — tests compare trajectories iteration by iteration
inline std::vector<double>& lm_trace_store() {
  static std::vector<double> t;
  return t;
}

struct LmProblem {
  virtual ~LmProblem() = default;
  virtual double linearize(double* grad_max_norm) = 0;
  virtual void set_jacobi_scaling(bool enabled) = 0;
  virtual bool step(double radius, double* model_change, double* cand_cost, double* step_norm, double* x_norm,
                    i64* linear_iterations, double* true_rel_residual) = 0;
  virtual void accept() = 0;
  // bounds-constrained programs (Program::IsBoundsConstrained): the projected line search of oracle/lm.py.
  //   step_slope(): g . delta at the current point for the step the last step() produced; step_max_norm(): |delta|_inf;
  //   ls_eval(t): cost and g(x_t) . delta at x_t = Plus(x, t delta) (projected on the bounds);
  //   set_step_size(t): delta *= t — the candidate, its cost and |candidate - x| are recomputed.
  virtual bool constrained() const { return false; }
  virtual double step_slope() const { return 0.0; }
  virtual double step_max_norm() const { return 0.0; }
  virtual void ls_eval(double t, double* cost, double* slope) { (void)t; (void)cost; (void)slope; }
  virtual void set_step_size(double t, double* cand_cost, double* step_norm) { (void)t; (void)cand_cost; (void)step_norm; }
};

// ---- Ceres' Armijo search (line_search.cc ArmijoLineSearch::DoSearch, CUBIC interpolation; polynomial.cc) -----------
// Restated as oracle/lm.py states it (find_interpolating_polynomial / minimize_polynomial / armijo_search); the numpy
// version solves the fit with LAPACK and finds the critical points with companion-matrix eigenvalues, this one with
// full-pivot elimination and sign-change bisection between the critical points of the derivative — tests/test_oracle_cpu.py
// holds the two to each other.
struct LsSample {
  double x = 0.0, value = 0.0, slope = 0.0;
  bool valid = false;
};

inline double poly_eval(const std::vector<double>& p, double x) {  // coefficients highest power first
  double v = 0.0;
  for (double a : p) v = v * x + a;
  return v;
}

inline std::vector<double> poly_fit(const std::vector<LsSample>& smp) {
  const int n = 2 * (int)smp.size(), deg = n - 1;
  std::vector<double> A((size_t)n * (n + 1), 0.0);
  auto at = [&](int r, int c) -> double& { return A[(size_t)r * (n + 1) + c]; };
  int row = 0;
  for (const LsSample& s : smp) {
    for (int j = 0; j <= deg; ++j) at(row, j) = std::pow(s.x, deg - j);
    at(row, n) = s.value;
    ++row;
    for (int j = 0; j < deg; ++j) at(row, j) = (deg - j) * std::pow(s.x, deg - j - 1);
    at(row, n) = s.slope;
    ++row;
  }
  std::vector<int> colperm(n);
  for (int i = 0; i < n; ++i) colperm[i] = i;
  for (int k = 0; k < n; ++k) {  // full pivoting, as Eigen::FullPivLU
    int pr = k, pc = k;
    double best = -1.0;
    for (int r = k; r < n; ++r)
      for (int c = k; c < n; ++c)
        if (std::fabs(at(r, c)) > best) best = std::fabs(at(r, c)), pr = r, pc = c;
    if (best <= 0.0) break;
    if (pr != k)
      for (int c = 0; c <= n; ++c) std::swap(at(pr, c), at(k, c));
    if (pc != k) {
      for (int r = 0; r < n; ++r) std::swap(at(r, pc), at(r, k));
      std::swap(colperm[pc], colperm[k]);
    }
    for (int r = k + 1; r < n; ++r) {
      const double f = at(r, k) / at(k, k);
      if (f == 0.0) continue;
      for (int c = k; c <= n; ++c) at(r, c) -= f * at(k, c);
    }
  }
  std::vector<double> y(n, 0.0), coef(n, 0.0);
  for (int k = n - 1; k >= 0; --k) {
    double v = at(k, n);
    for (int c = k + 1; c < n; ++c) v -= at(k, c) * y[c];
    y[k] = at(k, k) != 0.0 ? v / at(k, k) : 0.0;
  }
  for (int k = 0; k < n; ++k) coef[colperm[k]] = y[k];
  return coef;
}

inline std::vector<double> poly_strip(std::vector<double> p) {  // RemoveLeadingZeros
  size_t i = 0;
  while (i + 1 < p.size() && p[i] == 0.0) ++i;
  p.erase(p.begin(), p.begin() + i);
  return p;
}

inline std::vector<double> poly_derivative(const std::vector<double>& p) {
  const int deg = (int)p.size() - 1;
  std::vector<double> d;
  for (int j = 0; j < deg; ++j) d.push_back((deg - j) * p[j]);
  if (d.empty()) d.push_back(0.0);
  return d;
}

// real roots of p inside [lo, hi], ascending: sign changes between consecutive critical points, bisected to the last bit
inline std::vector<double> poly_roots_in(const std::vector<double>& p_in, double lo, double hi) {
  const std::vector<double> p = poly_strip(p_in);
  std::vector<double> roots;
  const int deg = (int)p.size() - 1;
  if (deg <= 0) return roots;
  if (deg == 1) {
    const double r = -p[1] / p[0];
    if (r >= lo && r <= hi) roots.push_back(r);
    return roots;
  }
  std::vector<double> brk{lo};
  for (double c : poly_roots_in(poly_derivative(p), lo, hi)) brk.push_back(c);
  brk.push_back(hi);
  for (size_t i = 0; i + 1 < brk.size(); ++i) {
    double a = brk[i], b = brk[i + 1], fa = poly_eval(p, a), fb = poly_eval(p, b);
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
      if (m <= a || m >= b) break;
      const double fm = poly_eval(p, m);
      if (fm == 0.0) {
        a = b = m;
        break;
      }
      if ((fm < 0.0) == (fa < 0.0)) a = m, fa = fm;
      else b = m;
    }
    roots.push_back(0.5 * (a + b));
  }
  return roots;
}

inline double poly_minimize(const std::vector<double>& poly, double x_min, double x_max) {  // MinimizePolynomial
  double best_x = 0.5 * (x_min + x_max), best_v = poly_eval(poly, best_x);
  for (double x : {x_min, x_max}) {
    const double v = poly_eval(poly, x);
    if (v < best_v) best_x = x, best_v = v;
  }
  const std::vector<double> p = poly_strip(poly);
  if (p.size() <= 2) return best_x;
  for (double x : poly_roots_in(poly_derivative(p), x_min, x_max)) {
    const double v = poly_eval(poly, x);
    if (v < best_v) best_x = x, best_v = v;
  }
  return best_x;
}

// Returns true and the accepted step size in *t_out when a trial satisfies the sufficient-decrease condition.
template <class Eval>
bool armijo_search(Eval eval, double cost0, double slope0, double direction_max_norm, const LmOptions& o, double* t_out,
                   int* trials_out) {
  LsSample lower{0.0, cost0, slope0, true}, previous, cur;
  cur.x = 1.0;
  eval(cur.x, &cur.value, &cur.slope);
  cur.valid = std::isfinite(cur.value);
  int iters = 0, trials = 1;
  *t_out = 1.0;
  while (!cur.valid || cur.value > cost0 + o.line_search_sufficient_function_decrease * slope0 * cur.x) {
    if (++iters >= o.max_num_line_search_step_size_iterations) {
      *trials_out = trials;
      return false;
    }
    const double x_lo = o.max_line_search_step_contraction * cur.x, x_hi = o.min_line_search_step_contraction * cur.x;
    double t;
    if (!cur.valid) {
      t = std::min(std::max(cur.x * 0.5, x_lo), x_hi);
    } else {
      std::vector<LsSample> smp{lower, cur};
      if (previous.valid) smp.push_back(previous);
      t = poly_minimize(poly_fit(smp), x_lo, x_hi);
    }
    if (t * direction_max_norm < o.min_line_search_step_size) {
      *trials_out = trials;
      return false;
    }
    previous = cur;
    cur = LsSample{};
    cur.x = t;
    eval(cur.x, &cur.value, &cur.slope);
    cur.valid = std::isfinite(cur.value);
    ++trials;
  }
  *t_out = cur.x;
  *trials_out = trials;
  return true;
}

inline void lm_minimize(LmProblem& prob, const LmOptions& o, LmSummary* s) {
  double gmax = 0.0;
  double cost = prob.linearize(&gmax);
  prob.set_jacobi_scaling(o.jacobi_scaling != 0);
  s->initial_cost = cost;
  std::vector<double>& trace = lm_trace_store();
  trace.clear();
  auto record = [&](double c, double rad, double mc, double cc, double t, double acc, double lin) {
    trace.insert(trace.end(), {c, rad, mc, cc, t, acc, lin});
  };
  double radius = o.initial_trust_region_radius;
  double decrease_factor = 2.0;
  int invalid = 0;
  if (o.verbose) fprintf(stderr, "[orc lm] it 0 cost %.9e gmax %.3e\n", cost, gmax);
  if (!(gmax > o.gradient_tolerance)) {
    s->termination = 0;
  } else {
    while (true) {
      if (s->iterations >= o.max_num_iterations) {
        s->termination = 1;
        break;
      }
      if (radius < o.min_trust_region_radius) {
        s->termination = 0;
        break;
      }
      ++s->iterations;
      double model_change = 0.0, cand_cost = 0.0, step_norm = 0.0, x_norm = 0.0, relres = 0.0;
      i64 lin = 0;
      const double t0 = omp_get_wtime();
      recycle_store().radius = radius;
      bool valid = prob.step(radius, &model_change, &cand_cost, &step_norm, &x_norm, &lin, &relres);
      s->seconds_linear += omp_get_wtime() - t0;
      s->linear_iterations += lin;
      s->max_linear_residual = std::max(s->max_linear_residual, relres);
      if (o.verbose)
        fprintf(stderr, "[orc lm] it %d radius %.3e pcg %lld (true relres %.2e) model %.6e cand %.9e (cost %.9e) step %.3e\n",
                s->iterations, radius, lin, relres, model_change, cand_cost, cost, step_norm);
      valid = valid && std::isfinite(model_change) && model_change > 0.0;
      if (!valid) {
        record(cost, radius, model_change, cand_cost, 1.0, -1.0, (double)lin);
        // Ceres: ++num_consecutive_invalid_steps >= max_num_consecutive_invalid_steps -> failure
        if (++invalid >= o.max_num_consecutive_invalid_steps) {
          s->termination = 2;
          s->usable = 0;
          break;
        }
        radius *= 0.5;
        continue;
      }
      invalid = 0;
      double step_size = 1.0;
      if (prob.constrained() && o.line_search && o.max_num_line_search_step_size_iterations > 0) {
        // TrustRegionMinimizer::DoLineSearch; model_change keeps the value of the full step
        double t = 1.0;
        int trials = 0;
        const bool ok = armijo_search([&](double tt, double* c, double* g) { prob.ls_eval(tt, c, g); }, cost, prob.step_slope(),
                                      prob.step_max_norm(), o, &t, &trials);
        s->line_search_steps += trials - 1;
        if (o.verbose) fprintf(stderr, "[orc lm]   line search: %s t %.15e trials %d\n", ok ? "ok" : "FAILED", t, trials);
        if (ok && t != 1.0) {
          ++s->line_search_shrunk;
          prob.set_step_size(t, &cand_cost, &step_norm);
        }
        step_size = ok ? t : -1.0;
      }
      if (step_norm <= o.parameter_tolerance * (x_norm + o.parameter_tolerance)) {
        record(cost, radius, model_change, cand_cost, step_size, 0.0, (double)lin);
        s->termination = 0;
        break;
      }
      const double cost_change = cost - cand_cost;
      if (std::fabs(cost_change) <= o.function_tolerance * cost) {
        record(cost, radius, model_change, cand_cost, step_size, 0.0, (double)lin);
        s->termination = 0;
        break;
      }
      const double rho = cost_change / model_change;
      record(cost, radius, model_change, cand_cost, step_size, rho > o.min_relative_decrease ? 1.0 : 0.0, (double)lin);
      if (rho > o.min_relative_decrease) {
        prob.accept();
        cost = prob.linearize(&gmax);
        ++s->successful_steps;
        if (!(gmax > o.gradient_tolerance)) {
          s->termination = 0;
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
  s->final_cost = cost;
}

constexpr i64 kDenseMax = 1536;  // reduced systems up to this size are assembled and factored densely

// Direct solve of S x = b for a small SPD operator: S is assembled column by column (S e_j), symmetrised and factored
// by Cholesky.  Returns false when a pivot is not positive (the caller then falls back to PCG).
template <class Apply>
bool dense_solve(i64 n, const std::vector<double>& b, std::vector<double>& x, Apply apply, double* true_relres) {
  std::vector<double> A((size_t)n * n), e(n, 0.0), col(n);
  for (i64 j = 0; j < n; ++j) {
    e[j] = 1.0;
    apply(e, col);
    e[j] = 0.0;
    for (i64 i = 0; i < n; ++i) A[(size_t)i * n + j] = col[i];
  }
  for (i64 i = 0; i < n; ++i)
    for (i64 j = 0; j < i; ++j) A[(size_t)i * n + j] = A[(size_t)j * n + i] = 0.5 * (A[(size_t)i * n + j] + A[(size_t)j * n + i]);
  std::vector<double> L(A);
  for (i64 j = 0; j < n; ++j) {
    double d = L[(size_t)j * n + j];
    for (i64 k = 0; k < j; ++k) d -= L[(size_t)j * n + k] * L[(size_t)j * n + k];
    if (!(d > 0.0)) return false;
    d = std::sqrt(d);
    L[(size_t)j * n + j] = d;
#pragma omp parallel for schedule(static)
    for (i64 i = j + 1; i < n; ++i) {
      double v = L[(size_t)i * n + j];
      const double *li = &L[(size_t)i * n], *lj = &L[(size_t)j * n];
      for (i64 k = 0; k < j; ++k) v -= li[k] * lj[k];
      L[(size_t)i * n + j] = v / d;
    }
  }
  x = b;
  for (i64 i = 0; i < n; ++i) {
    double v = x[i];
    for (i64 k = 0; k < i; ++k) v -= L[(size_t)i * n + k] * x[k];
    x[i] = v / L[(size_t)i * n + i];
  }
  for (i64 i = n - 1; i >= 0; --i) {
    double v = x[i];
    for (i64 k = i + 1; k < n; ++k) v -= L[(size_t)k * n + i] * x[k];
    x[i] = v / L[(size_t)i * n + i];
  }
  // one step of iterative refinement on the assembled matrix, then the true residual through the operator
  std::vector<double> w(n);
  apply(x, w);
  double rr = 0.0, bb = 0.0;
  for (i64 i = 0; i < n; ++i) {
    rr += (b[i] - w[i]) * (b[i] - w[i]);
    bb += b[i] * b[i];
  }
  *true_relres = bb > 0.0 ? std::sqrt(rr / bb) : 0.0;
  return true;
}

// ---- experiment (ORC_RECYCLE = k_max, tools/exp_gp_ritz_recycle.py): recycled Ritz vectors as an additive coarse space ----
// After a solve the Lanczos coefficients of the PCG give Ritz pairs (theta, u = V y) of the preconditioned operator; the
// ones below ORC_RECYCLE_CUT are kept (first in, first out, at most k_max) and the NEXT solves run with
//     M2^-1 = M^-1 + sum_j u_j u_j^T / theta_j
// — a different SPD preconditioner for the same system and the same stopping rule, so the solution is the same to the
// tolerance; theta and u are stale by one or more LM steps, which only costs preconditioner quality.  Layers stack: a mode
// the current M2 handles shows up at ~1 + theta, a mode it does not handle shows up low again and is harvested again.
// eigen-decomposition of a symmetric matrix (cyclic Jacobi; m <= a few hundred): A (m x m, row-major) -> eigenvalues in d,
// eigenvectors in the COLUMNS of V
inline void sym_eig_jacobi(std::vector<double>& A, int m, std::vector<double>& d, std::vector<double>& V) {
  V.assign((size_t)m * m, 0.0);
  for (int i = 0; i < m; ++i) V[(size_t)i * m + i] = 1.0;
  for (int sweep = 0; sweep < 60; ++sweep) {
    double off = 0.0;
    for (int i = 0; i < m; ++i)
      for (int j = i + 1; j < m; ++j) off += A[(size_t)i * m + j] * A[(size_t)i * m + j];
    if (off < 1e-30) break;
    for (int p = 0; p < m; ++p)
      for (int q = p + 1; q < m; ++q) {
        const double apq = A[(size_t)p * m + q];
        if (std::fabs(apq) < 1e-300) continue;
        const double app = A[(size_t)p * m + p], aqq = A[(size_t)q * m + q];
        const double tau = (aqq - app) / (2.0 * apq);
        const double t = (tau >= 0 ? 1.0 : -1.0) / (std::fabs(tau) + std::sqrt(1.0 + tau * tau));
        const double c = 1.0 / std::sqrt(1.0 + t * t), sn = t * c;
        for (int k = 0; k < m; ++k) {
          const double akp = A[(size_t)k * m + p], akq = A[(size_t)k * m + q];
          A[(size_t)k * m + p] = c * akp - sn * akq;
          A[(size_t)k * m + q] = sn * akp + c * akq;
        }
        for (int k = 0; k < m; ++k) {
          const double apk = A[(size_t)p * m + k], aqk = A[(size_t)q * m + k];
          A[(size_t)p * m + k] = c * apk - sn * aqk;
          A[(size_t)q * m + k] = sn * apk + c * aqk;
        }
        for (int k = 0; k < m; ++k) {
          const double vkp = V[(size_t)k * m + p], vkq = V[(size_t)k * m + q];
          V[(size_t)k * m + p] = c * vkp - sn * vkq;
          V[(size_t)k * m + q] = sn * vkp + c * vkq;
        }
      }
  }
  d.resize(m);
  for (int i = 0; i < m; ++i) d[i] = A[(size_t)i * m + i];
}

// Preconditioned conjugate gradients on an SPD operator, classic two-reduction form.
//   apply(z, w): w = S z;   precond(r, z): z = M^-1 r.
// Stops at |r| <= tol |b| (recurrence residual), at max_it, or when the residual has not improved
// by a factor 0.999 over 200 iterations (rounding floor; PCG residuals are not monotone, so the window is wide).  Returns the iteration count and writes the
// TRUE relative residual |b - S x| / |b| of the returned x.
// `x0` (optional, experiments only — tools/exp_lm_warm_start.py): start from gamma * x0 with the gamma that minimises the
// energy norm of the error along x0 (one more operator application); the stopping rule stays |r| <= tol |b|.
template <class Apply, class Precond>
// `defl` (optional, experiments only — tools/exp_deflation.py): deflated PCG (Saad, Yeung, Erhel, Guyomarc'h 2000) on
// the span of the given vectors W: x starts at W (W^T A W)^-1 W^T b and every preconditioned residual is projected,
// z <- z - W (W^T A W)^-1 (A W)^T z.  Same system, same stopping rule; the k applications of the operator that form A W
// are counted in the returned iteration count.
i64 pcg(i64 n, const std::vector<double>& b, std::vector<double>& x, double tol, int max_it, Apply apply, Precond precond,
        double* true_relres, const std::vector<double>* x0 = nullptr, const std::vector<std::vector<double>>* defl = nullptr,
        int stall_window = 200) {
  std::vector<double> r(b), z(n), p(n), w(n);
  std::fill(x.begin(), x.end(), 0.0);
  const double bnorm = std::sqrt(vdot(b, b));
  *true_relres = 0.0;
  if (!(bnorm > 0.0)) return 0;
  const int kd = defl ? (int)defl->size() : 0;
  std::vector<std::vector<double>> AW(kd, std::vector<double>(n));
  std::vector<double> Einv((size_t)kd * kd, 0.0);
  bool deflate = kd > 0;
  if (deflate) {
    for (int j = 0; j < kd; ++j) apply((*defl)[j], AW[j]);
    for (int i = 0; i < kd; ++i)
      for (int j = 0; j < kd; ++j) Einv[(size_t)i * kd + j] = vdot((*defl)[i], AW[j]);
    for (int i = 0; i < kd; ++i)
      for (int j = i + 1; j < kd; ++j) Einv[(size_t)i * kd + j] = Einv[(size_t)j * kd + i] = 0.5 * (Einv[(size_t)i * kd + j] + Einv[(size_t)j * kd + i]);
    // a mode whose preconditioned Rayleigh quotient theta = (w^T A w)(w^T M^-1 w) / (w^T w)^2 (the bulk of the spectrum is
    // ~1) has fallen to rounding level is an exact gauge held by nothing: "solving" it divides noise by nothing.  Such
    // a solve runs undeflated.
    static const bool dbg = std::getenv("ORC_DEFLATE_DEBUG") != nullptr;
    static const double theta_min = std::getenv("ORC_DEFLATE_MIN") ? std::atof(std::getenv("ORC_DEFLATE_MIN")) : 0.0;
    for (int j = 0; j < kd && deflate; ++j) {
      precond((*defl)[j], z);
      const double ww = vdot((*defl)[j], (*defl)[j]);
      const double theta = Einv[(size_t)j * kd + j] * vdot((*defl)[j], z) / (ww * ww);
      if (dbg) fprintf(stderr, "[orc deflate] mode %d theta %.3e\n", j, theta);
      if (!(theta > theta_min)) deflate = false;
    }
    deflate = deflate && spd_inverse(Einv.data(), kd);
  }
  // v <- v - W E^-1 (Q^T v)   with Q = W (start) or A W (projection of z)
  auto correct = [&](std::vector<double>& v, const std::vector<std::vector<double>>& Q, double sign, std::vector<double>* into) {
    std::vector<double> c(kd), y(kd, 0.0);
    for (int j = 0; j < kd; ++j) c[j] = vdot(Q[j], v);
    for (int i = 0; i < kd; ++i)
      for (int j = 0; j < kd; ++j) y[i] += Einv[(size_t)i * kd + j] * c[j];
    std::vector<double>& dst = into ? *into : v;
    for (int j = 0; j < kd; ++j) {
      const double a = sign * y[j];
      const std::vector<double>& wj = (*defl)[j];
      for (i64 i = 0; i < n; ++i) dst[i] += a * wj[i];
    }
    return y;
  };
  if (deflate) {
    std::vector<double> bb(b);
    const std::vector<double> y = correct(bb, *defl, +1.0, &x);  // x = W E^-1 W^T b
    for (int j = 0; j < kd; ++j)
      for (i64 i = 0; i < n; ++i) r[i] -= y[j] * AW[j][i];
  }
  if (x0 != nullptr && (i64)x0->size() == n) {
    apply(*x0, w);
    const double xb = vdot(*x0, b), xw = vdot(*x0, w);
    if (xw > 0.0 && std::isfinite(xw)) {
      const double gamma = xb / xw;
      for (i64 i = 0; i < n; ++i) {
        x[i] = gamma * (*x0)[i];
        r[i] = b[i] - gamma * w[i];
      }
    }
  }
  static const int rc_kmax = std::getenv("ORC_RECYCLE") ? std::atoi(std::getenv("ORC_RECYCLE")) : 0;
  static const double rc_cut = std::getenv("ORC_RECYCLE_CUT") ? std::atof(std::getenv("ORC_RECYCLE_CUT")) : 0.3;
  static const int rc_per = std::getenv("ORC_RECYCLE_PER") ? std::atoi(std::getenv("ORC_RECYCLE_PER")) : 8;
  RecycleStore& rc = recycle_store();
  const bool recycle = rc_kmax > 0 && x0 == nullptr;
  static const double rc_rad = std::getenv("ORC_RECYCLE_RADIUS") ? std::atof(std::getenv("ORC_RECYCLE_RADIUS")) : 3.0;
  static const int rc_age = std::getenv("ORC_RECYCLE_AGE") ? std::atoi(std::getenv("ORC_RECYCLE_AGE")) : 6;
  static const int rc_minit = std::getenv("ORC_RECYCLE_MINIT") ? std::atoi(std::getenv("ORC_RECYCLE_MINIT")) : 25;
  if (recycle && rc.n != n) {
    rc.n = n;
    rc.U.clear();
    rc.theta.clear();
    rc.rad.clear();
    rc.age.clear();
  }
  if (recycle) {  // drop what has gone stale: harvested at a radius too far from this solve's, or too many solves ago
    for (size_t j = 0; j < rc.U.size();) {
      const double q = rc.rad[j] / rc.radius;
      if (q > rc_rad || q < 1.0 / rc_rad || rc.age[j] >= rc_age) {
        rc.U.erase(rc.U.begin() + j);
        rc.theta.erase(rc.theta.begin() + j);
        rc.rad.erase(rc.rad.begin() + j);
        rc.age.erase(rc.age.begin() + j);
      } else {
        ++rc.age[j];
        ++j;
      }
    }
  }
  auto precond2 = [&](const std::vector<double>& rin, std::vector<double>& zout) {
    precond(rin, zout);
    if (!recycle) return;
    for (size_t j = 0; j < rc.U.size(); ++j) {
      const double a = vdot(rc.U[j], rin) / rc.theta[j];
      const double* u = rc.U[j].data();
      double* zo = zout.data();
      for (i64 i = 0; i < n; ++i) zo[i] += a * u[i];
    }
  };
  std::vector<std::vector<double>> Vh;  // Lanczos vectors (-1)^j z_j / sqrt(r_j . z_j)
  std::vector<double> al_h, rz_h;
  auto keep = [&](double rzj) {
    if (!recycle || Vh.size() >= 200) return;
    std::vector<double> vj(z);
    const double sc = ((Vh.size() & 1) ? -1.0 : 1.0) / std::sqrt(rzj);
    for (i64 i = 0; i < n; ++i) vj[i] *= sc;
    Vh.push_back(std::move(vj));
  };
  precond2(r, z);
  if (deflate) correct(z, AW, -1.0, nullptr);
  p = z;
  double rz = vdot(r, z);
  keep(rz);
  double best = bnorm;
  int since_best = 0;
  i64 it = 0;
  while (it < max_it) {
    apply(p, w);
    const double pw = vdot(p, w);
    if (!(pw > 0.0) || !std::isfinite(pw)) break;
    const double alpha = rz / pw;
    static const bool cg_trace = std::getenv("ORC_CG_TRACE") != nullptr;  // Lanczos coefficients for Ritz-value estimates
    if (cg_trace) fprintf(stderr, "[orc cg] it %lld alpha %.17g rz %.17g\n", (long long)it, alpha, rz);
    al_h.push_back(alpha);
    rz_h.push_back(rz);
    double *px = x.data(), *pr = r.data();
    const double *pp = p.data(), *pwv = w.data();
#pragma omp parallel for schedule(static)
    for (i64 i = 0; i < n; ++i) {
      px[i] += alpha * pp[i];
      pr[i] -= alpha * pwv[i];
    }
    ++it;
    const double rnorm = std::sqrt(vdot(r, r));
    if (rnorm <= tol * bnorm) break;
    if (rnorm < 0.999 * best) {
      best = rnorm;
      since_best = 0;
    } else if (++since_best >= stall_window) {
      break;
    }
    precond2(r, z);
    if (deflate) correct(z, AW, -1.0, nullptr);
    const double rz_new = vdot(r, z);
    keep(rz_new);
    const double beta = rz_new / rz;
    rz = rz_new;
    double* ppm = p.data();
    const double* pz = z.data();
#pragma omp parallel for schedule(static)
    for (i64 i = 0; i < n; ++i) ppm[i] = pz[i] + beta * ppm[i];
  }
  if (recycle && (int)al_h.size() >= rc_minit) {
    // Ritz pairs of this solve: T from (alpha, beta), m = number of steps whose Lanczos vector was kept
    const int m = (int)std::min(al_h.size(), Vh.size());
    std::vector<double> T((size_t)m * m, 0.0), th, Y;
    for (int j = 0; j < m; ++j) {
      const double bprev = j > 0 ? rz_h[j] / rz_h[j - 1] : 0.0;
      T[(size_t)j * m + j] = 1.0 / al_h[j] + (j > 0 ? bprev / al_h[j - 1] : 0.0);
      if (j + 1 < m) {
        const double bj = rz_h[j + 1] / rz_h[j];
        T[(size_t)j * m + j + 1] = T[(size_t)(j + 1) * m + j] = std::sqrt(bj) / al_h[j];
      }
    }
    const double tlast = (m < (int)al_h.size() || true) ? std::sqrt(std::max(0.0, (m < (int)rz_h.size() ? rz_h[m] / rz_h[m - 1] : 0.0))) / al_h[m - 1] : 0.0;
    sym_eig_jacobi(T, m, th, Y);
    std::vector<int> order(m);
    for (int i = 0; i < m; ++i) order[i] = i;
    std::sort(order.begin(), order.end(), [&](int a, int b2) { return th[a] < th[b2]; });
    static const bool dbg = std::getenv("ORC_RECYCLE_DEBUG") != nullptr;
    int added = 0;
    std::vector<std::vector<double>> fresh;
    std::vector<double> fresh_theta;
    for (int oi = 0; oi < m && added < rc_per; ++oi) {
      const int e = order[oi];
      if (!(th[e] > 0.0) || th[e] > rc_cut) break;
      const double resid = std::fabs(tlast * Y[(size_t)(m - 1) * m + e]);
      if (resid > 0.2 * th[e]) continue;  // not converged
      std::vector<double> u(n, 0.0);
      for (int j = 0; j < m; ++j) {
        const double yj = Y[(size_t)j * m + e];
        const double* vj = Vh[j].data();
        for (i64 i = 0; i < n; ++i) u[i] += yj * vj[i];
      }
      // ghost copies (loss of orthogonality): skip a vector nearly parallel to one already taken from this solve
      bool dup = false;
      const double uu = vdot(u, u);
      for (const auto& f : fresh) {
        const double c = vdot(u, f) / std::sqrt(uu * vdot(f, f));
        if (std::fabs(c) > 0.7) dup = true;
      }
      if (dup) continue;
      fresh.push_back(std::move(u));
      fresh_theta.push_back(th[e]);
      ++added;
    }
    if (dbg) {
      fprintf(stderr, "[orc recycle] solve of %d its, store %zu, harvested %d:", m, rc.U.size(), added);
      for (double t2 : fresh_theta) fprintf(stderr, " %.3e", t2);
      fprintf(stderr, "  (smallest Ritz:");
      for (int oi = 0; oi < std::min(m, 6); ++oi) fprintf(stderr, " %.3e", th[order[oi]]);
      fprintf(stderr, ")\n");
    }
    for (size_t f = 0; f < fresh.size(); ++f) {
      rc.U.push_back(std::move(fresh[f]));
      rc.theta.push_back(fresh_theta[f]);
      rc.rad.push_back(rc.radius);
      rc.age.push_back(0);
    }
    while ((int)rc.U.size() > rc_kmax) {  // full: the largest Ritz value goes first (it matters least)
      size_t worst = 0;
      for (size_t j = 1; j < rc.U.size(); ++j)
        if (rc.theta[j] > rc.theta[worst]) worst = j;
      rc.U.erase(rc.U.begin() + worst);
      rc.theta.erase(rc.theta.begin() + worst);
      rc.rad.erase(rc.rad.begin() + worst);
      rc.age.erase(rc.age.begin() + worst);
    }
  }
  apply(x, w);
  const double *pb = b.data(), *pwv = w.data();
  const double rr = chunked_sum(n, [=](i64 i) {
    const double d = pb[i] - pwv[i];
    return d * d;
  });
  *true_relres = std::sqrt(rr) / bnorm;
  return it + (deflate ? kd : 0);
}

// Exact treatment of a dense border.  A few intrinsics blocks shared by many images make the reduced system an arrowhead
//
//     S = [ A   B ]   A: poses (sparse, n0 = n - nb unknowns),   B: n0 x nb,   C: nb x nb (dense border, nb <= 32)
//         [ B^T C ]
//
// which SPARSE_SCHUR + a sparse Cholesky factors exactly.  Block-Jacobi PCG on S does not get there (the border couples
// every camera with every other; on configs[3] with one shared camera it stalls at a TRUE relative residual of 0.08 in
// some LM steps), so the border is eliminated densely here:
//
//     A Z = B,  A y = b_a   (nb + 1 solves with the pose part alone),   (C - B^T Z) x_c = b_c - B^T y,   x_a = y - Z x_c.
//
// Every A-solve is PCG to `tol` followed by iterative refinement on the TRUE residual (restart from b - A x until
// |b - A x| <= 1e-13 |b| or a restart no longer gains a factor of four): "exact" as a factorisation is — to the backward
// error of double precision, not to a recurrence residual.  `defl` (optional): modes deflated from the A-solves (their
// border entries must be zero).  Returns the total PCG iteration count plus the nb + refinement operator applications.
template <class Apply, class Precond>
i64 solve_bordered(i64 n, i64 nb, const std::vector<double>& b, std::vector<double>& x, double tol, int max_it, Apply apply,
                   Precond precond, double* true_relres, const std::vector<std::vector<double>>* defl = nullptr) {
  const i64 n0 = n - nb;
  i64 work = 0;
  std::vector<double> tmp(n), tmp2(n);
  auto applyA = [&](const std::vector<double>& z, std::vector<double>& w) {
    tmp = z;
    for (i64 i = n0; i < n; ++i) tmp[i] = 0.0;
    apply(tmp, w);
    for (i64 i = n0; i < n; ++i) w[i] = 0.0;
  };
  auto precondA = [&](const std::vector<double>& r, std::vector<double>& z) {
    tmp2 = r;
    for (i64 i = n0; i < n; ++i) tmp2[i] = 0.0;
    precond(tmp2, z);
    for (i64 i = n0; i < n; ++i) z[i] = 0.0;
  };
  // out = A^-1 rhs (border entries of rhs are ignored, those of out are zero)
  auto solveA = [&](const std::vector<double>& rhs_in, std::vector<double>& out) {
    std::vector<double> rhs(rhs_in), res, d(n), w(n);
    for (i64 i = n0; i < n; ++i) rhs[i] = 0.0;
    out.assign(n, 0.0);
    const double bn = std::sqrt(vdot(rhs, rhs));
    if (!(bn > 0.0)) return;
    res = rhs;
    double prev = bn;
    for (int pass = 0; pass < 8; ++pass) {
      double rel = 0.0;
      // every pass solves to 1e-10 of ITS right-hand side (the true residual of the previous pass): two or three passes reach
      // the 1e-13 target without ever asking a recurrence residual for more digits than it can deliver (a PCG run that is
      // asked for 1e-14 and stalls at 2e-14 burns its whole 200-iteration stagnation window)
      const double pass_tol = std::max(tol, 1e-10);
      const i64 its = pcg(n, res, d, pass_tol, max_it, applyA, precondA, &rel, nullptr, defl, 25);
      work += its;
      static const bool dbg = std::getenv("ORC_BORDER_DEBUG") != nullptr;
      if (dbg) fprintf(stderr, "[orc border] A-solve pass %d: %lld iterations, tol %.1e, true relres of the pass %.2e\n", pass, (long long)its, pass_tol, rel);
      for (i64 i = 0; i < n0; ++i) out[i] += d[i];
      applyA(out, w);
      ++work;
      for (i64 i = 0; i < n0; ++i) res[i] = rhs[i] - w[i];
      const double rn = std::sqrt(vdot(res, res));
      if (rn <= 1e-13 * bn || rn > 0.25 * prev) break;
      prev = rn;
    }
  };
  // border columns of S
  std::vector<std::vector<double>> col(nb, std::vector<double>(n)), Z(nb);
  std::vector<double> e(n, 0.0);
  for (i64 j = 0; j < nb; ++j) {
    e[n0 + j] = 1.0;
    apply(e, col[j]);
    e[n0 + j] = 0.0;
    ++work;
  }
  for (i64 j = 0; j < nb; ++j) solveA(col[j], Z[j]);
  std::vector<double> Sc((size_t)nb * nb);
  for (i64 i = 0; i < nb; ++i)
    for (i64 j = 0; j < nb; ++j) {
      double bz = 0.0;  // (B^T Z)_ij = B_i . Z_j over the pose part
      for (i64 k = 0; k < n0; ++k) bz += col[i][k] * Z[j][k];
      Sc[(size_t)i * nb + j] = col[j][n0 + i] - bz;
    }
  for (i64 i = 0; i < nb; ++i)
    for (i64 j = 0; j < i; ++j) Sc[(size_t)i * nb + j] = Sc[(size_t)j * nb + i] = 0.5 * (Sc[(size_t)i * nb + j] + Sc[(size_t)j * nb + i]);
  if (!spd_inverse(Sc.data(), (int)nb)) {  // not positive definite: leave it to plain PCG on the whole system
    return work + pcg(n, b, x, tol, max_it, apply, precond, true_relres, nullptr, defl);
  }
  std::fill(x.begin(), x.end(), 0.0);
  std::vector<double> r(b), y, w(n), xt(n), rt(n);
  const double bnorm = std::sqrt(vdot(b, b));
  *true_relres = 0.0;
  if (!(bnorm > 0.0)) return work;
  double prev = bnorm;
  *true_relres = 1.0;
  for (int round = 0; round < 6; ++round) {  // the bordered solve, then refinement of the whole system on its true residual
    solveA(r, y);
    std::vector<double> rc(nb), xc(nb, 0.0);
    for (i64 i = 0; i < nb; ++i) {
      double by = 0.0;
      for (i64 k = 0; k < n0; ++k) by += col[i][k] * y[k];
      rc[i] = r[n0 + i] - by;
    }
    for (i64 i = 0; i < nb; ++i)
      for (i64 j = 0; j < nb; ++j) xc[i] += Sc[(size_t)i * nb + j] * rc[j];
    xt = x;
    for (i64 k = 0; k < n0; ++k) {
      double v = y[k];
      for (i64 j = 0; j < nb; ++j) v -= Z[j][k] * xc[j];
      xt[k] += v;
    }
    for (i64 j = 0; j < nb; ++j) xt[n0 + j] += xc[j];
    apply(xt, w);
    ++work;
    for (i64 i = 0; i < n; ++i) rt[i] = b[i] - w[i];
    const double rn = std::sqrt(vdot(rt, rt));
    if (!(rn < prev)) break;  // a round that does not reduce the TRUE residual is not taken
    x.swap(xt);
    r.swap(rt);
    *true_relres = rn / bnorm;
    if (rn <= 1e-13 * bnorm || rn > 0.25 * prev) break;
    prev = rn;
  }
  // Near-singular steps (trust-region radius 1e10 ... 1e16: the similarity gauge is held by a damping at rounding level)
  // can leave the split solve short of what one deflated solve of the WHOLE system reaches — modes = the caller's plus
  // the border's unit vectors, which treats the border exactly as well.  Keep whichever has the smaller true residual.
  if (*true_relres > 1e-9) {
    std::vector<std::vector<double>> Wf;
    if (defl) Wf = *defl;
    for (i64 j = 0; j < nb; ++j) {
      Wf.emplace_back(n, 0.0);
      Wf.back()[n0 + j] = 1.0;
    }
    std::vector<double> x2(n, 0.0), d(n), r2(b);
    double prev2 = bnorm, rel2 = 1.0;
    for (int pass = 0; pass < 6; ++pass) {
      double rel = 0.0;
      work += pcg(n, r2, d, std::max(tol, 1e-10), max_it, apply, precond, &rel, nullptr, &Wf, 25);
      for (i64 i = 0; i < n; ++i) xt[i] = x2[i] + d[i];
      apply(xt, w);
      ++work;
      for (i64 i = 0; i < n; ++i) rt[i] = b[i] - w[i];
      const double rn = std::sqrt(vdot(rt, rt));
      if (!(rn < prev2)) break;
      x2.swap(xt);
      r2.swap(rt);
      rel2 = rn / bnorm;
      if (rn <= 1e-13 * bnorm || rn > 0.25 * prev2) break;
      prev2 = rn;
    }
    static const bool dbg = std::getenv("ORC_BORDER_DEBUG") != nullptr;
    if (dbg) fprintf(stderr, "[orc border] split solve %.2e, whole-system deflated solve %.2e\n", *true_relres, rel2);
    if (rel2 < *true_relres) {
      x = x2;
      *true_relres = rel2;
    }
  }
  return work;
}

// The reduced-system solve of one LM step: direct when small, PCG otherwise.  Returns the PCG iteration count (0 for
// the direct solve).
// `apply_cost` = work of one operator application (observations swept): the direct path applies the operator n times to
// assemble the matrix, so it is only taken while n * apply_cost stays small — beyond that PCG to 1e-14 is the "exact" solve
// (as it is for every system larger than kDenseMax).
constexpr double kDenseMaxWork = 2e8;
template <class Apply, class Precond>
i64 solve_reduced(i64 n, const std::vector<double>& b, std::vector<double>& x, double tol, int max_it, Apply apply,
                  Precond precond, double* true_relres, double apply_cost = 0.0, const std::vector<double>* x0 = nullptr,
                  const std::vector<std::vector<double>>* defl = nullptr) {
  if (n <= kDenseMax && (double)n * apply_cost <= kDenseMaxWork) {
    bool nonzero = false;
    for (double v : b) nonzero = nonzero || v != 0.0;
    if (!nonzero) {
      std::fill(x.begin(), x.end(), 0.0);
      *true_relres = 0.0;
      return 0;
    }
    if (dense_solve(n, b, x, apply, true_relres)) return 0;
  }
  return pcg(n, b, x, tol, max_it, apply, precond, true_relres, x0, defl);
}

}  // namespace orc
