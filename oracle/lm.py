"""ORACLE (test infrastructure only — never imported by the product path).

Restatement of the Ceres trust-region Levenberg-Marquardt loop that GLOMAP's GlobalPositioner and
BundleAdjuster run through `ceres::Solve` (gp.cc:83, ba.cc:99) with the options of
glomap/estimators/optimization_base.h:18-23 and Ceres' defaults for everything else
(SURVEY.md Appendix A.4).  Ceres itself is un-vendored (`find_package(Ceres)`,
cmake/FindDependencies.cmake:4) and absent here, so this follows its published algorithm
(ceres-solver 2.x: internal/ceres/trust_region_minimizer.cc, levenberg_marquardt_strategy.cc,
corrector.cc, loss_function.cc):

  * robustification: residual and Jacobian of each block scaled by sqrt(rho'(s)) (Corrector with
    rho'' <= 0, which is always the case for Huber and ScaledLoss(Huber));
  * Jacobi scaling computed ONCE at the initial point: scale_j = 1 / (1 + |J_j|_2);
  * LM step: (Js^T Js + diag(clamp(diag(Js^T Js), 1e-6, 1e32)) / radius) d = -Js^T r, delta = scale*d;
  * model_cost_change = -(J delta).(r + J delta / 2); invalid if <= 0 (radius *= 0.5);
  * candidate = Plus(x, delta) projected onto bounds (ParameterBlock::Plus);
  * bounds-constrained programs (Program::IsBoundsConstrained — GP: every scale has a lower bound,
    gp.cc:204,373) run TrustRegionMinimizer::DoLineSearch on every valid step before the candidate
    is evaluated: ArmijoLineSearch (line_search.cc) along t -> Plus(x, t delta) (projected), first
    trial t = 1, sufficient decrease f(t) <= f(0) + 1e-4 t g.delta, CUBIC interpolation (value and
    directional derivative g(x_t).delta at every trial; two-sample cubic first, then the
    three-sample quintic through the previous trial as well), new t = the minimiser of that
    polynomial over [1e-3 t, 0.6 t] (polynomial.cc: MinimizeInterpolatingPolynomial), at most 20
    trials, failure when t |delta|_inf < 1e-9; on success delta *= t, on failure delta is kept;
    model_cost_change stays the one of the FULL step (ComputeTrustRegionStep computed it before);
    the gradient-tolerance test of such programs uses |x - Plus(x, -g)|_inf (the projected gradient);
    rounds 1 - 5 omitted this search ("returns step size 1 whenever the projected step decreases the
    cost" — wrong: it needs SUFFICIENT decrease against the unprojected slope, and on GP it shrinks the
    step in about half of the iterations; VERDICT r5, tools/exp_gp_line_search.py);
  * termination order per iteration: parameter tolerance, function tolerance
    (|cost_change| <= ftol * cost, checked BEFORE the step is accepted — the candidate is then
    NOT applied), then accept (rho > 1e-3: radius /= max(1/3, 1-(2 rho-1)^3)) or reject
    (radius /= decrease_factor; decrease_factor *= 2);
  * gradient tolerance (max-norm) after every accepted step; min trust-region radius; iteration cap.

The linear system is solved EXACTLY: the independent diagonal blocks named by
`problem.elimination` (GP: the per-observation scales, then the 3x3 point blocks; BA: the 3x3
point blocks — the reference's ordering groups, gp.cc:388-429 / ba.cc:204-241) are eliminated by
block Gaussian elimination (Schur complement) and the remaining camera system is solved densely;
that is the same step SPARSE_SCHUR + sparse Cholesky compute.

parity unpinned: no reference test pins LM iterates; only converged solutions are compared.
"""
from __future__ import annotations

from dataclasses import dataclass, field

import numpy as np
import scipy.sparse as sp
import scipy.sparse.linalg as spla


@dataclass
class LmOptions:
    max_num_iterations: int = 100
    function_tolerance: float = 1e-5
    gradient_tolerance: float = 1e-10
    parameter_tolerance: float = 1e-8
    initial_trust_region_radius: float = 1e4
    max_trust_region_radius: float = 1e16
    min_trust_region_radius: float = 1e-32
    min_relative_decrease: float = 1e-3
    min_lm_diagonal: float = 1e-6
    max_lm_diagonal: float = 1e32
    jacobi_scaling: bool = True
    max_num_consecutive_invalid_steps: int = 5
    # TrustRegionMinimizer::DoLineSearch (bounds-constrained programs only) — Solver::Options defaults
    line_search: bool = True  # False: the loop of rounds 1 - 5 (experiments only)
    max_num_line_search_step_size_iterations: int = 20
    line_search_sufficient_function_decrease: float = 1e-4
    max_line_search_step_contraction: float = 1e-3
    min_line_search_step_contraction: float = 0.6
    min_line_search_step_size: float = 1e-9


@dataclass
class LmSummary:
    iterations: int = 0
    successful_steps: int = 0
    initial_cost: float = 0.0
    final_cost: float = 0.0
    termination: str = ""
    usable: bool = True
    costs: list = field(default_factory=list)
    radii: list = field(default_factory=list)
    line_search_steps: int = 0      # Armijo trials beyond the first (Summary::num_line_search_steps)
    line_search_shrunk: int = 0     # LM iterations whose step the search shortened
    step_sizes: list = field(default_factory=list)


class HuberLoss:
    """ceres::HuberLoss(a): rho(s) = s for s <= a^2, 2 a sqrt(s) - a^2 beyond (loss_function.cc)."""

    def __init__(self, a: float, scale: float = 1.0):
        self.a = a
        self.b = a * a
        self.scale = scale  # ceres::ScaledLoss multiplier

    def evaluate(self, s):
        s = np.asarray(s, dtype=np.float64)
        out = s > self.b
        r = np.sqrt(np.where(out, s, 1.0))
        rho0 = np.where(out, 2.0 * self.a * r - self.b, s)
        rho1 = np.where(out, np.maximum(np.finfo(np.float64).tiny, self.a / r), 1.0)
        return self.scale * rho0, self.scale * rho1


def _block_diag_inverse(Ad, bsize):
    """Inverse of a block-diagonal sparse matrix with dense bsize x bsize diagonal blocks."""
    nb = Ad.shape[0] // bsize
    if bsize == 1:
        return sp.diags(1.0 / Ad.diagonal())
    idx = np.arange(nb * bsize).reshape(nb, bsize)
    rows = np.repeat(idx, bsize, axis=1).ravel()
    cols = np.tile(idx, (1, bsize)).ravel()
    blocks = np.asarray(Ad[rows, cols]).reshape(nb, bsize, bsize)
    inv = np.linalg.inv(blocks)
    return sp.csr_matrix((inv.ravel(), (rows, cols)), shape=Ad.shape)


def schur_solve(A, rhs, elimination):
    """Solves A x = rhs exactly.  `elimination` = [(start, count, bsize), ...]: index ranges whose
    diagonal blocks are mutually independent; they are eliminated in order, the rest is solved
    densely, then back-substituted."""
    n = A.shape[0]
    if not elimination:
        return np.linalg.solve(A.toarray(), rhs) if n <= 4000 else spla.splu(A.tocsc()).solve(rhs)
    start, count, bsize = elimination[0]
    e = np.arange(start, start + count * bsize)
    mask = np.ones(n, dtype=bool)
    mask[e] = False
    rest = np.nonzero(mask)[0]
    Aee = A[e][:, e]
    Aer = A[e][:, rest]
    Are = A[rest][:, e]
    Arr = A[rest][:, rest]
    Aee_inv = _block_diag_inverse(Aee.tocsr(), bsize)
    T = Are @ Aee_inv
    S = (Arr - T @ Aer).tocsr()
    rr = rhs[rest] - T @ rhs[e]
    # shift the remaining elimination ranges into the index space of `rest`
    pos = -np.ones(n, dtype=np.int64)
    pos[rest] = np.arange(rest.shape[0])
    nxt = [(int(pos[s0]), c0, b0) for (s0, c0, b0) in elimination[1:]]
    xr = schur_solve(S, rr, nxt)
    xe = Aee_inv @ (rhs[e] - Aer @ xr)
    x = np.empty(n)
    x[rest] = xr
    x[e] = xe
    return x


# ---- Ceres' Armijo line search on the projected step (line_search.cc, polynomial.cc) ----------------


def find_interpolating_polynomial(samples):
    """polynomial.cc FindInterpolatingPolynomial: samples = [(x, value, gradient or None)], every value valid;
    coefficients highest power first.  (Ceres: Eigen::FullPivLU of the same system.)"""
    nc = sum(1 + (g is not None) for (_, _, g) in samples)
    deg = nc - 1
    lhs = np.zeros((nc, nc))
    rhs = np.zeros(nc)
    row = 0
    for (x, v, g) in samples:
        for j in range(deg + 1):
            lhs[row, j] = x ** (deg - j)
        rhs[row] = v
        row += 1
        if g is not None:
            for j in range(deg):
                lhs[row, j] = (deg - j) * x ** (deg - j - 1)
            rhs[row] = g
            row += 1
    return np.linalg.solve(lhs, rhs)


def minimize_polynomial(poly, x_min, x_max):
    """polynomial.cc MinimizePolynomial: midpoint, the two ends, then the critical points inside [x_min, x_max], each taking
    over only when strictly lower.  (Ceres takes the real PARTS of all roots of the derivative — companion-matrix
    eigenvalues; the real part of a complex pair is an ordinary interior point and can never beat the candidates that
    contain the interval's true minimiser, so only real roots are looked at here.)"""
    best_x = 0.5 * (x_min + x_max)
    best_v = float(np.polyval(poly, best_x))
    for x in (x_min, x_max):
        v = float(np.polyval(poly, x))
        if v < best_v:
            best_x, best_v = x, v
    p = np.array(poly, dtype=np.float64)
    nz = np.nonzero(p)[0]
    p = p[nz[0]:] if nz.size else p[-1:]
    if p.shape[0] <= 2:
        return best_x, best_v
    d = np.polyder(p)
    nz = np.nonzero(d)[0]
    d = d[nz[0]:] if nz.size else d[-1:]
    if d.shape[0] < 2:
        return best_x, best_v
    roots = np.roots(d)
    for r in roots:
        if abs(r.imag) > 1e-9 * max(1.0, abs(r.real)):
            continue
        x = float(r.real)
        if x < x_min or x > x_max:
            continue
        v = float(np.polyval(poly, x))
        if v < best_v:
            best_x, best_v = x, v
    return best_x, best_v


def armijo_search(evaluate, cost0, gdot0, direction_max_norm, o: "LmOptions"):
    """line_search.cc ArmijoLineSearch::DoSearch with step_size_estimate 1 and CUBIC interpolation.
    evaluate(t) -> (cost at Plus(x, t delta), g(that point) . delta).  Returns (success, t, trials)."""
    lower = (0.0, cost0, gdot0)
    previous = None
    t = 1.0
    cur = (t,) + tuple(evaluate(t))
    iters = 0
    trials = 1
    while (not np.isfinite(cur[1])) or cur[1] > cost0 + o.line_search_sufficient_function_decrease * gdot0 * cur[0]:
        iters += 1
        if iters >= o.max_num_line_search_step_size_iterations:
            return False, 1.0, trials
        x_lo = o.max_line_search_step_contraction * cur[0]
        x_hi = o.min_line_search_step_contraction * cur[0]
        if not np.isfinite(cur[1]):
            t = min(max(cur[0] * 0.5, x_lo), x_hi)
        else:
            samples = [lower, (cur[0], cur[1], cur[2] if np.isfinite(cur[2]) else None)]
            if previous is not None and np.isfinite(previous[1]):
                samples.append((previous[0], previous[1], previous[2] if np.isfinite(previous[2]) else None))
            t, _ = minimize_polynomial(find_interpolating_polynomial(samples), x_lo, x_hi)
        if t * direction_max_norm < o.min_line_search_step_size:
            return False, 1.0, trials
        previous = cur
        cur = (t,) + tuple(evaluate(t))
        trials += 1
    return True, cur[0], trials


def solve(problem, x0, options: LmOptions):
    """problem API:
         evaluate(x)            -> (cost, r_tilde [m], J_tilde csr [m x n])   robustified, tangent space at x
         cost(x)                -> cost
         plus(x, delta)         -> x (+) delta, projected onto bounds
         x_norm(x)              -> |x| used by the parameter-tolerance test
         is_constrained         (optional attribute) the program has a bounded, non-constant parameter block
    Returns (x, LmSummary)."""
    o = options
    x = x0
    cost, r, J = problem.evaluate(x)
    summ = LmSummary(initial_cost=cost, final_cost=cost)
    summ.costs.append(cost)
    n = J.shape[1]
    g = J.T @ r
    constrained = bool(getattr(problem, "is_constrained", False))

    def grad_max_norm(x_, g_):
        # trust_region_minimizer.cc EvaluateGradientAndJacobian: |x - Plus(x, -g)|_inf when options.is_constrained
        if constrained:
            return float(np.abs(x_ - problem.plus(x_, -g_)).max(initial=0.0))
        return float(np.abs(g_).max(initial=0.0))

    if o.jacobi_scaling:
        colsq = np.asarray(J.multiply(J).sum(axis=0)).ravel()
        scale = 1.0 / (1.0 + np.sqrt(colsq))
    else:
        scale = np.ones(n)
    if grad_max_norm(x, g) <= o.gradient_tolerance:
        summ.termination = "CONVERGENCE (gradient)"
        return x, summ
    radius = o.initial_trust_region_radius
    decrease_factor = 2.0
    invalid = 0
    it = 0
    while True:
        if it >= o.max_num_iterations:
            summ.termination = "NO_CONVERGENCE (max iterations)"
            break
        if radius < o.min_trust_region_radius:
            summ.termination = "CONVERGENCE (min trust region radius)"
            break
        it += 1
        summ.iterations = it
        summ.radii.append(radius)
        Js = J @ sp.diags(scale)
        A = (Js.T @ Js).tocsc()
        diag = np.clip(A.diagonal(), o.min_lm_diagonal, o.max_lm_diagonal)
        A = A + sp.diags(diag / radius)
        rhs = -(Js.T @ r)
        try:
            ds = schur_solve(A.tocsr(), rhs, getattr(problem, "elimination", []))
        except (RuntimeError, np.linalg.LinAlgError):
            ds = np.full(n, np.nan)
        delta = scale * ds
        valid = np.all(np.isfinite(delta))
        model_change = 0.0
        if valid:
            mr = J @ delta
            model_change = -float(mr @ (r + 0.5 * mr))
            valid = model_change > 0.0
        if not valid:
            invalid += 1
            # Ceres: ++num_consecutive_invalid_steps >= max_num_consecutive_invalid_steps -> FAILURE
            if invalid >= o.max_num_consecutive_invalid_steps:
                summ.termination = "FAILURE (invalid steps)"
                summ.usable = False
                break
            radius *= 0.5
            continue
        invalid = 0
        if constrained and o.line_search and o.max_num_line_search_step_size_iterations > 0:
            # TrustRegionMinimizer::DoLineSearch(x, gradient, cost, &delta); model_change keeps the full step's value
            def ls_eval(t):
                ct, rt, Jt = problem.evaluate(problem.plus(x, t * delta))
                return ct, float((Jt.T @ rt) @ delta)

            ok, t, trials = armijo_search(ls_eval, cost, float(g @ delta), float(np.abs(delta).max(initial=0.0)), o)
            summ.line_search_steps += trials - 1
            summ.step_sizes.append(t if ok else -1.0)
            if ok:
                if t != 1.0:
                    summ.line_search_shrunk += 1
                delta = delta * t
        cand = problem.plus(x, delta)
        cand_cost = problem.cost(cand)
        step_norm = problem.step_norm(x, cand)
        if step_norm <= o.parameter_tolerance * (problem.x_norm(x) + o.parameter_tolerance):
            summ.termination = "CONVERGENCE (parameter tolerance)"
            break
        cost_change = cost - cand_cost
        if abs(cost_change) <= o.function_tolerance * cost:
            summ.termination = "CONVERGENCE (function tolerance)"
            break
        rho = cost_change / model_change
        if rho > o.min_relative_decrease:
            x = cand
            cost, r, J = problem.evaluate(x)
            summ.successful_steps += 1
            summ.costs.append(cost)
            g = J.T @ r
            if grad_max_norm(x, g) <= o.gradient_tolerance:
                summ.termination = "CONVERGENCE (gradient)"
                break
            radius = radius / max(1.0 / 3.0, 1.0 - (2.0 * rho - 1.0) ** 3)
            radius = min(o.max_trust_region_radius, radius)
            decrease_factor = 2.0
        else:
            radius = radius / decrease_factor
            decrease_factor *= 2.0
    summ.final_cost = cost
    return x, summ
