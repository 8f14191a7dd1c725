"""Scenes with the locality of a walk-around capture (synthetic capture="sequential": every point seen by a run of
consecutive cameras) have a CHAIN-like reduced camera system: block-Jacobi PCG with the four global gauge modes deflated
needs hundreds of iterations per solve where the random-visibility scenes need twenty (bench extra gp_c3_sequential_capture:
9 491 instead of 507 operator applications per GP solve).  CPU study on the dense Schur complement of a small GP problem:
PCG iterations to 1e-8 with block-Jacobi alone, with the global gauge modes deflated, and with a piecewise-constant coarse
space (per cluster of `m` consecutive cameras: three translation modes, optionally a local scale mode) deflated.

    python tools/exp_coarse_space.py [num_cams num_pts]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import exp_precond as G  # noqa: E402
from glomap_amd import synthetic  # noqa: E402
from oracle import gp as ogp  # noqa: E402


def pcg(S, b, Minv_blocks, W=None, tol=1e-8, max_it=5000):
    n = b.size

    def prec(r):
        return np.einsum("nij,nj->ni", Minv_blocks, r.reshape(-1, 3)).ravel()

    if W is not None:
        AW = S @ W
        E = W.T @ AW
        Einv = np.linalg.inv(0.5 * (E + E.T))
        x = W @ (Einv @ (W.T @ b))
    else:
        x = np.zeros(n)
    r = b - S @ x
    bn = np.linalg.norm(b)
    z = prec(r)
    if W is not None:
        z -= W @ (Einv @ (AW.T @ z))
    p = z.copy()
    rz = r @ z
    for it in range(1, max_it + 1):
        w = S @ p
        a = rz / (p @ w)
        x += a * p
        r -= a * w
        if np.linalg.norm(r) <= tol * bn:
            return it, x
        z = prec(r)
        if W is not None:
            z -= W @ (Einv @ (AW.T @ z))
        rz2 = r @ z
        p = z + (rz2 / rz) * p
        rz = rz2
    return max_it, x


def main():
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 600
    P = int(sys.argv[2]) if len(sys.argv) > 2 else 60_000
    for capture in ("random", "sequential"):
        p = synthetic.make_gp_problem(num_cams=N, num_pts=P, seed=0, capture=capture)
        opt = ogp.GlobalPositionerOptions()
        lens = np.diff(p.pt_offset)
        used = lens >= opt.min_num_view_per_track
        obs_pt = np.repeat(np.arange(P), lens)
        keep = used[obs_pt]
        remap = -np.ones(P, dtype=np.int64)
        remap[used] = np.arange(int(used.sum()))
        prob = ogp._GpProblem(N, p.obs_cam[keep].astype(np.int64), remap[obs_pt[keep]], p.obs_dir[keep], p.obs_calibrated[keep], opt, int(used.sum()))
        # a state on the way: ground truth + noise (the conditioning is what matters, not the point on the LM path)
        rng = np.random.default_rng(0)
        c = p.gt_center + rng.normal(0, 1.0, p.gt_center.shape)
        X = p.gt_xyz[used] + rng.normal(0, 1.0, (int(used.sum()), 3))
        d = X[prob.pt] - c[prob.cam]
        s = np.maximum(1e-5, np.einsum("mj,mj->m", prob.v, d) / np.einsum("mj,mj->m", d, d))
        x = np.concatenate([c.ravel(), X.ravel(), s])
        for lam in (1e-4, 1e-6):
            S, b = G.schur_system(prob, x, lam)
            Sd = S.toarray()
            Mi = np.stack([np.linalg.inv(Sd[3 * n : 3 * n + 3, 3 * n : 3 * n + 3]) for n in range(N)])
            rows = []
            it, _ = pcg(Sd, b, Mi)
            rows.append(("block-Jacobi", it))
            Wg = np.zeros((3 * N, 4))
            for a in range(3):
                Wg[a::3, a] = 1.0
            Wg[:, 3] = c.ravel()
            it, _ = pcg(Sd, b, Mi, Wg)
            rows.append(("+ 4 global gauge modes", it))
            for m in (100, 50, 25, 12):
                nc = (N + m - 1) // m
                for local_scale in (False, True):
                    k = nc * (4 if local_scale else 3)
                    W = np.zeros((3 * N, k))
                    for q in range(nc):
                        sl = slice(q * m, min(N, (q + 1) * m))
                        for a in range(3):
                            W[3 * sl.start + a : 3 * sl.stop : 3, q * 3 + a] = 1.0
                        if local_scale:
                            cc = c[sl] - c[sl].mean(0)
                            W[3 * sl.start : 3 * sl.stop, 3 * nc + q] = cc.ravel()
                    if not local_scale:
                        W = np.column_stack([W, c.ravel()])
                    it, _ = pcg(Sd, b, Mi, W)
                    rows.append((f"+ clusters of {m} cameras ({W.shape[1]} modes{', local scale' if local_scale else ' + global scale'})", it))
            print(f"{capture:10s} N={N} damping {lam:.0e}: " + "; ".join(f"{n}: {i}" for n, i in rows), flush=True)


if __name__ == "__main__" and "additive" not in sys.argv:
    main()


def pcg_additive(S, b, Minv_blocks, W, E, tol=1e-8, max_it=5000):
    """PCG with the two-level additive preconditioner  M^-1 = blockdiag^-1 + W E^-1 W^T  (E need not be exact: it only
    shapes the preconditioner)."""
    Einv = np.linalg.inv(0.5 * (E + E.T))

    def prec(r):
        return np.einsum("nij,nj->ni", Minv_blocks, r.reshape(-1, 3)).ravel() + W @ (Einv @ (W.T @ r))

    x = np.zeros(b.size)
    r = b.copy()
    bn = np.linalg.norm(b)
    z = prec(r)
    p = z.copy()
    rz = r @ z
    for it in range(1, max_it + 1):
        w = S @ p
        a = rz / (p @ w)
        x += a * p
        r -= a * w
        if np.linalg.norm(r) <= tol * bn:
            return it
        z = prec(r)
        rz2 = r @ z
        p = z + (rz2 / rz) * p
        rz = rz2
    return max_it


def study_additive(N=600, P=60_000):
    """Additive coarse correction instead of deflation, E = W^T A W exact or from three-colour probing (A applied to the
    sum of the modes of every third cluster, restricted to the cluster's neighbourhood)."""
    p = synthetic.make_gp_problem(num_cams=N, num_pts=P, seed=0, capture="sequential")
    opt = ogp.GlobalPositionerOptions()
    lens = np.diff(p.pt_offset)
    used = lens >= opt.min_num_view_per_track
    obs_pt = np.repeat(np.arange(P), lens)
    keep = used[obs_pt]
    remap = -np.ones(P, dtype=np.int64)
    remap[used] = np.arange(int(used.sum()))
    prob = ogp._GpProblem(N, p.obs_cam[keep].astype(np.int64), remap[obs_pt[keep]], p.obs_dir[keep], p.obs_calibrated[keep], opt, int(used.sum()))
    rng = np.random.default_rng(0)
    c = p.gt_center + rng.normal(0, 1.0, p.gt_center.shape)
    X = p.gt_xyz[used] + rng.normal(0, 1.0, (int(used.sum()), 3))
    d = X[prob.pt] - c[prob.cam]
    s = np.maximum(1e-5, np.einsum("mj,mj->m", prob.v, d) / np.einsum("mj,mj->m", d, d))
    x = np.concatenate([c.ravel(), X.ravel(), s])
    for lam in (1e-4, 1e-6):
        S, b = G.schur_system(prob, x, lam)
        Sd = S.toarray()
        Mi = np.stack([np.linalg.inv(Sd[3 * n : 3 * n + 3, 3 * n : 3 * n + 3]) for n in range(N)])
        out = []
        for m in (50, 25, 12):
            nc = (N + m - 1) // m
            k = 4 * nc
            W = np.zeros((3 * N, k))
            cl = np.minimum(np.arange(N) // m, nc - 1)
            for q in range(nc):
                sl = slice(q * m, min(N, (q + 1) * m))
                for a in range(3):
                    W[3 * sl.start + a : 3 * sl.stop : 3, 4 * q + a] = 1.0
                W[3 * sl.start : 3 * sl.stop, 4 * q + 3] = (c[sl] - c[sl].mean(0)).ravel()
            E_exact = W.T @ Sd @ W
            # three-colour probing (ring: cluster q's neighbours are q - 1 and q + 1 mod nc; needs nc % 3 == 0 or a 4th colour)
            ncol = 3 if nc % 3 == 0 else 4
            E_probe = np.zeros((k, k))
            for col in range(ncol):
                for t in range(4):
                    sel = [4 * q + t for q in range(nc) if q % ncol == col]
                    AV = Sd @ W[:, sel].sum(1)
                    for q in range(nc):
                        if q % ncol != col:
                            continue
                        for qq in ((q - 1) % nc, q, (q + 1) % nc):
                            rows = np.repeat(cl == qq, 3)
                            E_probe[4 * qq : 4 * qq + 4, 4 * q + t] = W[rows][:, 4 * qq : 4 * qq + 4].T @ AV[rows]
            e_err = np.abs(E_probe - E_exact).max() / np.abs(E_exact).max()
            it_defl, _ = pcg(Sd, b, Mi, W)
            out.append(f"m={m} ({k} modes): deflation {it_defl}, additive exact E {pcg_additive(Sd, b, Mi, W, E_exact)}, "
                       f"additive probed E {pcg_additive(Sd, b, Mi, W, E_probe)} (|E_probe - E| / |E| = {e_err:.1e})")
        print(f"sequential N={N} damping {lam:.0e}: " + "; ".join(out), flush=True)


if __name__ == "__main__" and "additive" in sys.argv:
    study_additive()
