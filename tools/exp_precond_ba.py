"""What a stronger preconditioner could buy the BA reduced camera system (CPU study, no GPU needed).

configs[3] spends 600 of its 688 PCG iterations in the LM steps with a wide trust region (radius 1e4 ... 2.4e6; 38-67
iterations each, tools/exp_lm_warm_start.py).  This builds the Jacobi-scaled, LM-damped normal equations of a synthetic
BA problem at its start point with the numpy oracle's Jacobian (one SIMPLE_RADIAL camera per image, as configs[3]),
eliminates the points exactly and runs PCG to 1e-8 on the reduced system for a few radii with
  * 6 x 6 pose blocks and the intrinsics blocks on their own,
  * joint pose + intrinsics blocks (8 x 8 — what ba.hip uses when every image has its own camera),
  * exact inverses of the joint blocks of k index-adjacent cameras (ring order = spatially adjacent), k = 4, 16, 64.
Usage: python tools/exp_precond_ba.py [num_cams] [num_pts]"""
import sys
import time

import numpy as np
import scipy.sparse as sp
import scipy.sparse.linalg as spla

sys.path.insert(0, ".")
from glomap_amd import synthetic  # noqa: E402
from oracle import ba as oba  # noqa: E402


def reduced_system(prob, x, radius):
    cost, r, J = prob.evaluate(x)
    H = (J.T @ J).tocsr()
    g = J.T @ r
    d = H.diagonal()
    keep = d > 0  # constant parameters (fixed frame, frozen principal point ...) have empty columns
    js = np.where(keep, 1.0 / (1.0 + np.sqrt(d)), 0.0)  # Ceres' Jacobi scaling
    Hs = sp.diags(js) @ H @ sp.diags(js)
    gs = js * g
    dm = np.clip(Hs.diagonal(), 1e-6, 1e32) / radius
    Hs = (Hs + sp.diags(np.where(keep, dm, 1.0))).tocsr()
    nc = prob.pt_col0
    P = prob.P
    Hxx = Hs[nc:][:, nc:].tocoo()
    blocks = np.zeros((P, 3, 3))
    same = (Hxx.row // 3) == (Hxx.col // 3)
    np.add.at(blocks, (Hxx.row[same] // 3, Hxx.row[same] % 3, Hxx.col[same] % 3), Hxx.data[same])
    inv = np.linalg.inv(blocks)
    rows = (3 * np.arange(P)[:, None, None] + np.arange(3)[None, :, None]).repeat(3, axis=2)
    cols = (3 * np.arange(P)[:, None, None] + np.arange(3)[None, None, :]).repeat(3, axis=1)
    Hxx_inv = sp.csr_matrix((inv.ravel(), (rows.ravel(), cols.ravel())), shape=(3 * P, 3 * P))
    Hcx = Hs[:nc][:, nc:]
    S = (Hs[:nc][:, :nc] - Hcx @ Hxx_inv @ Hcx.T).tocsr()
    b = -(gs[:nc] - Hcx @ (Hxx_inv @ gs[nc:]))
    free = keep[:nc]
    idx = np.nonzero(free)[0]
    return S[idx][:, idx].tocsr(), b[idx], idx


def block_precond(S, groups):
    """groups: list of index arrays (into S) — exact inverse of each diagonal block."""
    n = S.shape[0]
    inv = [np.linalg.inv(S[gidx][:, gidx].toarray()) for gidx in groups]

    def apply(v):
        out = np.zeros_like(v)
        for gidx, B in zip(groups, inv):
            out[gidx] = B @ v[gidx]
        return out

    return spla.LinearOperator((n, n), matvec=apply)


def pcg_iters(S, b, M, tol=1e-8):
    it = [0]

    def cb(_):
        it[0] += 1

    x, info = spla.cg(S, b, rtol=tol, atol=0.0, maxiter=5000, M=M, callback=cb)
    return it[0], np.linalg.norm(S @ x - b) / np.linalg.norm(b)


def main():
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 600
    P = int(sys.argv[2]) if len(sys.argv) > 2 else 60_000
    p = synthetic.make_ba_problem(N, P, seed=0)
    opt = oba.BundleAdjusterOptions()
    lens = np.diff(p.pt_offset)
    used = lens >= opt.min_num_view_per_track
    obs_pt = np.repeat(np.arange(P), lens)
    keep = used[obs_pt]
    remap = -np.ones(P, dtype=np.int64)
    remap[used] = np.arange(int(used.sum()))
    prob = oba._BaProblem(N, p.obs_cam[keep].astype(np.int64), remap[obs_pt[keep]], p.obs_xy[keep], p.cam_intr.astype(np.int64),
                          p.intr_model.astype(np.int64), int(p.fixed_cam), int(used.sum()), opt)
    x0 = prob.pack(p.cam_q, p.cam_t, p.pt_xyz[used], p.intr_params)
    print(f"BA: cameras {N} tracks {prob.P} observations {prob.M}", flush=True)
    for radius in (1e4, 1e6):
        t0 = time.time()
        S, b, idx = reduced_system(prob, x0, radius)
        pos = -np.ones(prob.pt_col0, dtype=np.int64)
        pos[idx] = np.arange(idx.shape[0])

        def cols_of_cam(n, with_intr):
            c = [6 * n + j for j in range(6)]
            if with_intr:
                c += [int(v) for v in prob.intr_col[p.cam_intr[n]] if v >= 0]
            c = pos[np.array(c)]
            return c[c >= 0]

        print(f"== radius {radius:.0e}: reduced system {S.shape[0]} unknowns, {S.nnz / S.shape[0]:.0f} nnz per row (assembly {time.time() - t0:.1f} s)")
        sep = [cols_of_cam(n, False) for n in range(N)] + [pos[prob.intr_col[k][prob.intr_col[k] >= 0]] for k in range(prob.K)]
        sep = [gidx for gidx in sep if gidx.shape[0]]
        it, res = pcg_iters(S, b, block_precond(S, sep))
        print(f"   pose blocks + intrinsics blocks on their own: {it:5d} PCG iterations  (true relres {res:.1e})")
        for k in (1, 4, 16, 64):
            groups = []
            for c0 in range(0, N, k):
                gidx = np.concatenate([cols_of_cam(n, True) for n in range(c0, min(N, c0 + k))])
                if gidx.shape[0]:
                    groups.append(np.unique(gidx))
            it, res = pcg_iters(S, b, block_precond(S, groups))
            print(f"   joint blocks of {k:3d} cameras: {it:5d} PCG iterations  (true relres {res:.1e})")


if __name__ == "__main__":
    main()
