"""Where do the PCG iterations of rotation averaging go, and what would remove them?  (CPU study, no GPU needed.)

Replays the L1 (ADMM) + IRLS loop of the numpy oracle (oracle/ra.py) on a ring view graph with the linear solves done
the way ra.hip does them at 2 048 < N <= 32 768 — PCG on the weighted graph Laplacian (three right-hand sides as one
vector), preconditioned by fp32 inverses of index-contiguous diagonal blocks after a BFS relabelling — and counts PCG
iterations per stage for a few variants:

  exact        sparse LU (the oracle itself): the reference result
  current      cold first x-update of every ADMM solve to 1e-10, warm-started corrections to 1e-10 OF THE WARM-START
               RESIDUAL (what ra.hip does), fresh block inverses for every IRLS weight set
  abs          as current, but the warm solves stop at tol * |rhs| (the error left in x is the same as a cold solve's)
  abs+recycle  as abs, plus a Galerkin start from the search directions of the earlier solves with the same matrix
  interface    block inverses + an exact Schur complement on the nodes that couple the blocks (substructuring) as
               preconditioner, abs stopping rule
  interface-stale   the same, factored ONCE for the unit-weight Laplacian of the L1 stage and kept for the IRLS weights
  coarse<m>    block inverses + an additive piecewise-constant coarse space over aggregates of m BFS-consecutive nodes (round 5)
  abs+deflate  as abs, with the constant vector (the global rotation, held by the gauge row alone) deflated from the PCG

and reports, against `exact`, the largest difference of the resulting rotations.
Usage: python tools/exp_ra_linear_solves.py [num_cams] [successors | geometric | hub | chords] [variant ...]"""
import sys
import time

import numpy as np
import scipy.sparse as sp
import scipy.sparse.linalg as spla
from scipy.sparse.csgraph import breadth_first_order

sys.path.insert(0, ".")
from glomap_amd import so3 as gso3  # noqa: E402
from glomap_amd import synthetic  # noqa: E402
from oracle import ra as ora  # noqa: E402
from oracle import so3  # noqa: E402

BLOCK_MAX = 2048  # kDenseMaxN of ra.hip


class Laplacian:
    """L_w + gauge on N nodes (the 3N system is L (x) I3: three right-hand sides)."""

    def __init__(self, N, ei, ej, fixed):
        self.N, self.ei, self.ej, self.fixed = N, ei, ej, fixed
        adj = sp.csr_matrix((np.ones(ei.shape[0]), (ei, ej)), shape=(N, N))
        order = breadth_first_order(adj + adj.T, fixed, directed=False, return_predecessors=False)
        rest = np.setdiff1d(np.arange(N), order)
        self.order = np.concatenate([order, rest])
        nblk = (N + BLOCK_MAX - 1) // BLOCK_MAX
        nb = (N + nblk - 1) // nblk
        self.blocks = [self.order[k * nb : min(N, (k + 1) * nb)] for k in range(nblk)]
        self.set_weights(np.ones(ei.shape[0]))

    def set_weights(self, w):
        N, ei, ej = self.N, self.ei, self.ej
        d = np.bincount(ei, w, N) + np.bincount(ej, w, N)
        d[self.fixed] += 1.0
        self.L = (sp.csr_matrix((np.concatenate([-w, -w]), (np.concatenate([ei, ej]), np.concatenate([ej, ei]))), shape=(N, N))
                  + sp.diags(d)).tocsr()

    def factor_blocks(self):
        self.inv = [np.linalg.inv(self.L[b][:, b].toarray()).astype(np.float32) for b in self.blocks]

    def apply_blocks(self, R):
        out = np.empty_like(R)
        for b, Bi in zip(self.blocks, self.inv):
            out[b] = Bi.astype(np.float64) @ R[b]
        return out

    def factor_coarse(self, agg):
        """Two-level additive: block inverses + a piecewise-constant coarse space over aggregates of `agg` consecutive nodes
        of the BFS order (aggregation AMG's tentative prolongator): M^-1 = blockdiag^-1 + W (W^T L W)^-1 W^T."""
        self.factor_blocks()
        self.agg_of = np.empty(self.N, dtype=np.int64)
        self.agg_of[self.order] = np.arange(self.N) // agg
        self.nagg = int(self.agg_of.max()) + 1
        W = sp.csr_matrix((np.ones(self.N), (np.arange(self.N), self.agg_of)), shape=(self.N, self.nagg))
        self.W = W
        self.Einv = np.linalg.inv((W.T @ self.L @ W).toarray())

    def apply_coarse(self, R):
        return self.apply_blocks(R) + self.W @ (self.Einv @ (self.W.T @ R))

    def factor_interface(self):
        """Substructuring: I = nodes all of whose neighbours are in their own block, B = the rest."""
        self.factor_blocks()
        blk = np.empty(self.N, dtype=np.int64)
        for k, b in enumerate(self.blocks):
            blk[b] = k
        cross = blk[self.ei] != blk[self.ej]
        isb = np.zeros(self.N, dtype=bool)
        isb[self.ei[cross]] = True
        isb[self.ej[cross]] = True
        self.B = np.nonzero(isb)[0]
        self.Iblocks = [b[~isb[b]] for b in self.blocks]
        self.Iinv = [np.linalg.inv(self.L[b][:, b].toarray()).astype(np.float32) for b in self.Iblocks]
        L = self.Lf = self.L  # the matrix the factors belong to (set_weights makes a new self.L)
        S = L[self.B][:, self.B].toarray()
        for b, Bi in zip(self.Iblocks, self.Iinv):
            LBI = L[self.B][:, b].toarray()
            S -= LBI @ (Bi.astype(np.float64) @ LBI.T)
        self.Sinv = np.linalg.inv(S)

    def apply_interface(self, R):
        L = self.Lf
        y = np.zeros_like(R)
        for b, Bi in zip(self.Iblocks, self.Iinv):
            y[b] = Bi.astype(np.float64) @ R[b]
        rb = R[self.B] - (L[self.B] @ y)  # y is zero on B
        xb = self.Sinv @ rb
        out = np.zeros_like(R)
        out[self.B] = xb
        tmp = np.zeros_like(R)
        tmp[self.B] = xb
        Lt = L @ tmp
        for b, Bi in zip(self.Iblocks, self.Iinv):
            out[b] = Bi.astype(np.float64) @ (R[b] - Lt[b])
        return out


def pcg(A, B, X0, precond, tol_abs2, max_it=2000, keep=None, deflate=False):
    """Joint PCG over the columns of B; stops at |r|^2 <= tol_abs2.  keep: list collecting (p, Ap, pAp).
    deflate: project the constant vector (the global rotation, anchored by the gauge row alone) out of every column."""
    X = X0.copy()
    R = B - A @ X
    if deflate:
        one = np.ones(A.shape[0])
        a1 = A @ one
        e = float(one @ a1)
        X = X + np.outer(one, one @ R) / e
        R = B - A @ X
        inner = precond
        precond = lambda r: (lambda z: z - np.outer(one, a1 @ z) / e)(inner(r))  # noqa: E731
    it = 0
    rr = float((R * R).sum())
    if rr <= tol_abs2:
        return X, 0
    Z = precond(R)
    P = Z.copy()
    rz = float((R * Z).sum())
    while it < max_it:
        AP = A @ P
        pap = float((P * AP).sum())
        if keep is not None:
            keep.append((P.copy(), AP.copy(), pap))
        a = rz / pap
        X += a * P
        R -= a * AP
        it += 1
        rr = float((R * R).sum())
        if rr <= tol_abs2:
            break
        Z = precond(R)
        rz_new = float((R * Z).sum())
        P = Z + (rz_new / rz) * P
        rz = rz_new
    return X, it


class Solver:
    def __init__(self, lap: Laplacian, variant):
        self.lap, self.variant = lap, variant
        self.iters = {"l1": 0, "irls": 0}
        self.factorizations = 0
        self.basis = []
        self.lu = None

    def new_matrix(self):
        if self.variant == "interface-stale" and self.factorizations > 0:
            return
        self.basis = []
        self.lu = None
        if self.variant == "exact":
            self.lu = spla.splu(self.lap.L.tocsc())
        elif self.variant.startswith("interface"):
            self.lap.factor_interface()
        elif self.variant.startswith("coarse"):
            self.lap.factor_coarse(int(self.variant[6:] or 32))
        else:
            self.lap.factor_blocks()
        self.factorizations += 1

    def solve(self, B, stage, X_warm=None, tol=1e-10):
        lap = self.lap
        if self.variant == "exact":
            return self.lu.solve(B)
        precond = lap.apply_interface if self.variant.startswith("interface") else (lap.apply_coarse if self.variant.startswith("coarse") else lap.apply_blocks)
        bb = float((B * B).sum())
        X0 = np.zeros_like(B) if X_warm is None else X_warm
        keep = None
        if self.variant == "abs+recycle":
            keep = self.basis
            if self.basis:
                R = B - lap.L @ X0
                for P, AP, pap in self.basis:  # Galerkin correction in the span of the stored directions
                    c = float((P * R).sum()) / pap
                    X0 = X0 + c * P
                    R = R - c * AP
                if len(self.basis) > 400:
                    del self.basis[:-400]
        if self.variant == "current" and X_warm is not None:
            R0 = B - lap.L @ X0
            ref2 = float((R0 * R0).sum())
        else:
            ref2 = bb
        X, it = pcg(lap.L, B, X0, precond, tol * tol * ref2, keep=keep, deflate=self.variant.endswith("+deflate"))
        self.iters[stage] += it
        return X


def run(p, variant, opt):
    N = p.num_nodes
    ei, ej = p.edge_i.astype(np.int64), p.edge_j.astype(np.int64)
    E = ei.shape[0]
    edge_R = so3.quat_wxyz_to_rotmat(p.edge_q)
    rot = ora.maximum_spanning_tree_init(N, ei, ej, edge_R, p.edge_ninl, np.array(p.node_aa0, dtype=np.float64))
    fixed = p.fixed_node
    fixed_rot = rot[fixed].copy()
    lap = Laplacian(N, ei, ej, fixed)
    sol = Solver(lap, variant)

    def residuals(r):
        return ora.compute_residuals(r, ei, ej, edge_R, fixed, fixed_rot)

    def At(y):  # A^T y as [N,3]
        ye = y[: 3 * E].reshape(E, 3)
        out = np.zeros((N, 3))
        np.add.at(out, ej, ye)
        np.add.at(out, ei, -ye)
        out[fixed] += y[3 * E :]
        return out

    def Ax(X):
        return np.concatenate([(X[ej] - X[ei]).reshape(-1), X[fixed]])

    # ---- L1 (oracle/ra.py LeastAbsoluteDeviationSolver, unit weights)
    sol.new_matrix()
    b = residuals(rot)
    m, n = 3 * E + 3, 3 * N
    last = cur = 0.0
    l1_its = 0
    for it in range(opt.max_num_l1_iterations):
        z = np.zeros(m)
        u = np.zeros(m)
        X = None
        rhs_norm = np.linalg.norm(b)
        for a in range(opt.l1_admm_max_num_iterations):
            X = sol.solve(At(b + z - u), "l1", X_warm=X)
            ax = Ax(X)
            z_old = z
            z = ora._shrink(ax - b + u, 1.0 / opt.l1_admm_rho)
            u = u + ax - z - b
            r_norm = np.linalg.norm(ax - z - b)
            s_norm = np.linalg.norm(At(z - z_old))
            eps_p = np.sqrt(m) * opt.l1_admm_absolute_tolerance + opt.l1_admm_relative_tolerance * max(np.linalg.norm(ax), np.linalg.norm(z), rhs_norm)
            eps_d = np.sqrt(n) * opt.l1_admm_absolute_tolerance + opt.l1_admm_relative_tolerance * np.linalg.norm(At(u))
            if r_norm < eps_p and s_norm < eps_d:
                break
        last, cur = cur, float(np.linalg.norm(X))
        rot = ora.update_global_rotations(rot, X)
        b = residuals(rot)
        l1_its = it + 1
        if ora.average_step_size(X) < opt.l1_step_convergence_threshold or abs(last - cur) < so3.EPS:
            break
    # ---- IRLS (Geman-McClure)
    sigma = np.radians(opt.irls_loss_parameter_sigma)
    irls_its = 0
    for it in range(opt.max_num_irls_iterations):
        e2 = (b[: 3 * E].reshape(-1, 3) ** 2).sum(axis=1)
        w = sigma * sigma / (e2 + sigma * sigma) ** 2
        lap.set_weights(w)
        sol.new_matrix()
        wb = b.copy()
        wb[: 3 * E] *= np.repeat(w, 3)
        X = sol.solve(At(wb), "irls")
        rot = ora.update_global_rotations(rot, X)
        b = residuals(rot)
        irls_its = it + 1
        if ora.average_step_size(X) < opt.irls_step_convergence_threshold:
            break
    return rot, dict(l1_outer=l1_its, irls=irls_its, pcg_l1=sol.iters["l1"], pcg_irls=sol.iters["irls"], factorizations=sol.factorizations,
                     interface_nodes=int(getattr(lap, "B", np.zeros(0)).shape[0]))


def main():
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000
    kind = sys.argv[2] if len(sys.argv) > 2 else "50"
    variants = sys.argv[3:] or ["exact", "current", "abs", "abs+recycle", "interface", "interface-stale"]
    if kind.isdigit():
        p = synthetic.make_ring_view_graph(N, int(kind), seed=0)
        kind = f"ring, {kind} successors"
    else:
        p = synthetic.make_view_graph(kind, N, degree=100, seed=0)
    opt = ora.RotationEstimatorOptions()
    print(f"view graph ({kind}): {N} cameras, {p.num_edges} edges", flush=True)
    ref = None
    for v in variants:
        t0 = time.time()
        rot, info = run(p, v, opt)
        if ref is None:
            ref = rot
        d = np.radians(gso3.rotation_angle_deg(gso3.aa_to_rotmat(rot), gso3.aa_to_rotmat(ref)))
        print(f"{v:12s} {info}  max |rot - first variant| = {d.max():.2e} rad   ({time.time() - t0:.0f} s)", flush=True)


if __name__ == "__main__":
    main()
