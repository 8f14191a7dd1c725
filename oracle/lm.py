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
  * candidate = Plus(x, delta) projected onto bounds (ParameterBlock::Plus); the Armijo line
    search Ceres adds for bounded problems is omitted (it returns step size 1 whenever the
    projected LM step already decreases the cost) — deliberate deviation, same in the product;
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


def solve(problem, x0, options: LmOptions):
    """problem API:
         evaluate(x)            -> (cost, r_tilde [m], J_tilde csr [m x n])   robustified, tangent space at x
         cost(x)                -> cost
         plus(x, delta)         -> x (+) delta, projected onto bounds
         x_norm(x)              -> |x| used by the parameter-tolerance test
    Returns (x, LmSummary)."""
    o = options
    x = x0
    cost, r, J = problem.evaluate(x)
    summ = LmSummary(initial_cost=cost, final_cost=cost)
    summ.costs.append(cost)
    n = J.shape[1]
    g = J.T @ r
    if o.jacobi_scaling:
        colsq = np.asarray(J.multiply(J).sum(axis=0)).ravel()
        scale = 1.0 / (1.0 + np.sqrt(colsq))
    else:
        scale = np.ones(n)
    if np.abs(g).max(initial=0.0) <= o.gradient_tolerance:
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
            if np.abs(g).max(initial=0.0) <= o.gradient_tolerance:
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
