"""Second-level preconditioner for bundle adjustment on scenes with the locality of a real capture (CPU study on the dense
reduced camera system of a small problem; the GP version is built, tools/exp_coarse_space.py and DESIGN.md 4.2).

bench extra ba_c4_sequential_capture: 2 193 operator applications per BA solve (73 per reduced solve) where the
random-visibility scene needs 195.  This builds the Jacobi-scaled, LM-damped reduced system of a synthetic BA problem (one
SIMPLE_RADIAL camera per image, random or sequential visibility) with the numpy oracle's Jacobian and counts PCG iterations
to 1e-6 with joint pose + intrinsics block-Jacobi
  * alone,
  * with the 7 global gauge modes deflated (what ba.hip does),
  * with a two-level ADDITIVE preconditioner  M^-1 = blockdiag^-1 + W E^-1 W^T,  E = W^T S W,  W = per cluster of m
    consecutive cameras the 7 similarity modes about the cluster's centroid (translation dt = -R a; rotation
    drot = -R w / 2, dt = R (w x cbar); scale dt = t + R cbar) — the piecewise coarse space of the GP solver, 7 wide.

    python tools/exp_coarse_space_ba.py [num_cams num_pts]"""
import os
import sys

import numpy as np
import scipy.sparse.linalg as spla

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
os.chdir(ROOT)
import exp_precond_ba as B  # noqa: E402
from glomap_amd import so3, synthetic  # noqa: E402
from oracle import ba as oba  # noqa: E402


def gauge_modes(R, t, members, cbar):
    """[6 N, 7] similarity modes of the cameras in `members` about `cbar` (zero rows elsewhere), unscaled variables."""
    N = R.shape[0]
    W = np.zeros((6 * N, 7))
    for n in members:
        for a in range(3):
            e = np.zeros(3)
            e[a] = 1.0
            W[6 * n + 3 : 6 * n + 6, a] = -R[n] @ e
            W[6 * n : 6 * n + 3, 3 + a] = -0.5 * (R[n] @ e)
            W[6 * n + 3 : 6 * n + 6, 3 + a] = R[n] @ np.cross(e, cbar)
        W[6 * n + 3 : 6 * n + 6, 6] = t[n] + R[n] @ cbar
    return W


def pcg(S, b, prec, tol=1e-6, max_it=3000, defl=None):
    x = np.zeros_like(b)
    if defl is not None:
        W, AW, Einv = defl
        x = W @ (Einv @ (W.T @ b))
    r = b - S @ x
    bn = np.linalg.norm(r)

    def P(v):
        z = prec(v)
        if defl is not None:
            z = z - W @ (Einv @ (AW.T @ z))
        return z

    z = P(r)
    p = z.copy()
    rz = r @ z
    for it in range(1, max_it + 1):
        w = S @ p
        a = rz / (p @ w)
        x += a * p
        r -= a * w
        if np.linalg.norm(r) <= tol * bn:
            return it
        z = P(r)
        rz2 = r @ z
        p = z + (rz2 / rz) * p
        rz = rz2
    return max_it


def main():
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 480
    P = int(sys.argv[2]) if len(sys.argv) > 2 else 40_000
    for capture in ("random", "sequential"):
        p = synthetic.make_ba_problem(N, P, seed=0, capture=capture)
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
        R = so3.quat_to_rotmat(np.asarray(p.cam_q))
        t = np.asarray(p.cam_t)
        cen = -np.einsum("nji,nj->ni", R, t)
        print(f"--- {capture} visibility: {N} cameras, {int(used.sum())} points, {int(keep.sum())} observations")
        for radius in (1e4, 1e6):
            S, b, idx = B.reduced_system(prob, x0, radius)
            Sd = S.toarray()
            nfree = idx.shape[0]
            pos = -np.ones(prob.pt_col0, dtype=np.int64)
            pos[idx] = np.arange(nfree)
            # Jacobi scaling of the reduced variables (reduced_system works in scaled variables: x = js * x_scaled)
            _, _, J = prob.evaluate(x0)
            d = np.asarray((J.multiply(J)).sum(axis=0)).ravel()[: prob.pt_col0]
            js = np.where(d > 0, 1.0 / (1.0 + np.sqrt(d)), 0.0)[idx]
            groups = []
            for n in range(N):
                c = [6 * n + j for j in range(6)] + [int(v) for v in prob.intr_col[p.cam_intr[n]] if v >= 0]
                c = pos[np.array(c)]
                c = c[c >= 0]
                if c.size:
                    groups.append(c)
            inv = [np.linalg.inv(Sd[np.ix_(g, g)]) for g in groups]

            def bj(v):
                out = np.zeros_like(v)
                for g, Bi in zip(groups, inv):
                    out[g] = Bi @ v[g]
                return out

            def coarse(members_list):
                cols = []
                for mem in members_list:
                    Wc = gauge_modes(R, t, mem, cen[mem].mean(axis=0))
                    Wf = np.zeros((nfree, 7))
                    ok = pos[: 6 * N] >= 0
                    Wf[pos[: 6 * N][ok]] = Wc[ok]
                    cols.append(Wf / js[:, None])  # scaled variables
                W = np.concatenate(cols, axis=1)
                W = W[:, np.abs(W).sum(axis=0) > 0]
                AW = Sd @ W
                E = W.T @ AW
                return W, AW, np.linalg.pinv(0.5 * (E + E.T), rcond=1e-13)

            it_plain = pcg(Sd, b, bj)
            Wg = coarse([np.arange(N)])
            it_defl = pcg(Sd, b, bj, defl=Wg)
            line = f"radius {radius:.0e}: block-Jacobi {it_plain}, + 7 global modes deflated {it_defl}"
            for m in (16, 32, 64):
                nc = max(1, N // m)
                clusters = [np.arange(N)[(np.arange(N) * nc) // N == q] for q in range(nc)]
                W, AW, Einv = coarse(clusters)

                def two_level(v, W=W, Einv=Einv):
                    return bj(v) + W @ (Einv @ (W.T @ v))

                line += f", two-level m={m} ({W.shape[1]} modes) {pcg(Sd, b, two_level)}"
            print(line)


if __name__ == "__main__":
    main()
