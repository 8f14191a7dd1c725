"""What a stronger preconditioner could buy the GP reduced camera system (CPU study, no GPU needed).

Builds the normal equations of a synthetic GP problem at the random start and near the solution with the numpy oracle's
Jacobian, eliminates scales and points exactly (scipy sparse algebra), and runs PCG on the 3N x 3N Schur complement with
  * block-Jacobi on the 3 x 3 camera blocks (what gp.hip does),
  * cluster-Jacobi: exact inverses of the diagonal blocks of k index-contiguous cameras (cameras are ring-ordered, so
    contiguous = spatially adjacent), k = 4, 16, 64,
  * the same clusters after a co-visibility ordering (reverse Cuthill-McKee of S's block pattern).
Prints PCG iterations to a relative residual of 1e-8.  Usage: python tools/exp_precond.py [num_cams] [num_pts]"""
import sys
import time

import numpy as np
import scipy.sparse as sp
import scipy.sparse.linalg as spla
from scipy.sparse.csgraph import reverse_cuthill_mckee

sys.path.insert(0, ".")
from glomap_amd import synthetic  # noqa: E402
from oracle import gp as ogp  # noqa: E402
from oracle import lm  # noqa: E402


def schur_system(prob, x, lam=1e-4):
    cost, r, J = prob.evaluate(x)
    N, P, M = prob.N, prob.P, prob.M
    H = (J.T @ J).tocsr()
    g = J.T @ r
    d = H.diagonal()
    H = H + sp.diags(lam * np.maximum(d, 1e-6))
    nc, npnt = 3 * N, 3 * P
    ic = np.arange(nc)
    ix = nc + np.arange(npnt)
    isc = nc + npnt + np.arange(M)
    Hss = H[isc][:, isc].diagonal()
    Hss[0] = 1.0  # the constant first scale: zero row / column in J
    rest = np.concatenate([ic, ix])
    Hrs = H[rest][:, isc]
    Hrr = H[rest][:, rest] - Hrs @ sp.diags(1.0 / Hss) @ Hrs.T
    gr = g[rest] - Hrs @ (g[isc] / Hss)
    Hxx = Hrr[nc:][:, nc:].tocsr()
    # 3 x 3 block inverse of the point part
    blocks = np.zeros((P, 3, 3))
    coo = Hxx.tocoo()
    same = (coo.row // 3) == (coo.col // 3)
    np.add.at(blocks, (coo.row[same] // 3, coo.row[same] % 3, coo.col[same] % 3), coo.data[same])
    inv = np.linalg.inv(blocks)
    rows = (3 * np.arange(P)[:, None, None] + np.arange(3)[None, :, None]).repeat(3, axis=2)
    cols = (3 * np.arange(P)[:, None, None] + np.arange(3)[None, None, :]).repeat(3, axis=1)
    Hxx_inv = sp.csr_matrix((inv.ravel(), (rows.ravel(), cols.ravel())), shape=(npnt, npnt))
    Hcx = Hrr[:nc][:, nc:]
    S = (Hrr[:nc][:, :nc] - Hcx @ Hxx_inv @ Hcx.T).tocsr()
    b = -(gr[:nc] - Hcx @ (Hxx_inv @ gr[nc:]))
    return S, b


def pcg_iters(S, b, Minv, tol=1e-8, max_it=5000):
    it = [0]

    def cb(_):
        it[0] += 1

    x, info = spla.cg(S, b, rtol=tol, atol=0.0, maxiter=max_it, M=Minv, callback=cb)
    return it[0], np.linalg.norm(S @ x - b) / np.linalg.norm(b)


def cluster_precond(S, k_cams, perm=None):
    n = S.shape[0]
    N = n // 3
    order = np.arange(N) if perm is None else perm
    idx = (3 * order[:, None] + np.arange(3)[None, :]).ravel()
    Sp = S[idx][:, idx].tocsr()
    blocks = []
    for c0 in range(0, N, k_cams):
        sl = slice(3 * c0, 3 * min(N, c0 + k_cams))
        blocks.append(np.linalg.inv(Sp[sl][:, sl].toarray()))
    inv_idx = np.empty(n, dtype=np.int64)
    inv_idx[idx] = np.arange(n)

    def apply(v):
        vp = v[idx]
        out = np.empty_like(vp)
        o = 0
        for B in blocks:
            out[o : o + B.shape[0]] = B @ vp[o : o + B.shape[0]]
            o += B.shape[0]
        return out[inv_idx]

    return spla.LinearOperator((n, n), matvec=apply)


def main():
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 400
    P = int(sys.argv[2]) if len(sys.argv) > 2 else 40_000
    p = synthetic.make_gp_problem(num_cams=N, num_pts=P, seed=0)
    opt = ogp.GlobalPositionerOptions()
    lens = np.diff(p.pt_offset)
    used = lens >= opt.min_num_view_per_track
    obs_pt = np.repeat(np.arange(P), lens)
    keep = used[obs_pt]
    remap = -np.ones(P, dtype=np.int64)
    remap[used] = np.arange(int(used.sum()))
    prob = ogp._GpProblem(N, p.obs_cam[keep].astype(np.int64), remap[obs_pt[keep]], p.obs_dir[keep], p.obs_calibrated[keep], opt,
                          int(used.sum()))
    rng = np.random.default_rng(0)
    starts = {
        "random start": np.concatenate([100 * rng.uniform(-1, 1, 3 * N), 100 * rng.uniform(-1, 1, 3 * prob.P), np.ones(prob.M)]),
    }
    # near the solution: ground truth with consistent scales
    d = p.gt_xyz[used][prob.pt] - p.gt_center[prob.cam]
    starts["near solution"] = np.concatenate([p.gt_center.ravel(), p.gt_xyz[used].ravel(), 1.0 / np.linalg.norm(d, axis=1)])
    for name, x in starts.items():
        t0 = time.time()
        S, b = schur_system(prob, x)
        print(f"== {name}: N={N} P={prob.P} M={prob.M}  S nnz blocks/row = {S.nnz / 9 / N:.0f}  (assembly {time.time() - t0:.1f} s)")
        pat = sp.csr_matrix((np.ones(S.nnz), S.indices // 3, S.indptr))[::3]
        pat = sp.csr_matrix((np.ones(pat.nnz), (np.repeat(np.arange(N), np.diff(pat.indptr)), pat.indices)), shape=(N, N))
        rcm = np.asarray(reverse_cuthill_mckee(pat.tocsr(), symmetric_mode=True))
        for k in (1, 4, 16, 64):
            it, res = pcg_iters(S, b, cluster_precond(S, k))
            it2, _ = pcg_iters(S, b, cluster_precond(S, k, rcm)) if k > 1 else (it, res)
            print(f"   cluster of {k:3d} cameras: {it:5d} PCG iterations (ring order)   {it2:5d} (RCM order)   true relres {res:.1e}")


if __name__ == "__main__":
    main()
