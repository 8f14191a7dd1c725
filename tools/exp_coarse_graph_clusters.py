"""Clusters of the second-level GP preconditioner from the CAMERA GRAPH instead of the camera numbering (CPU study).

The built version (gp.hip, GpCoarseDev) clusters cameras by index range — fine when frames are numbered in capture order,
useless when they are not (the C++ adapter numbers frames in the iteration order of an unordered_map).  This prototype
  1. grows connected clusters of ~m cameras by breadth-first region growing on the co-visibility chain graph (consecutive
     cameras of every track),
  2. calls two clusters adjacent when some track touches both (exactly the non-zero blocks of W^T S W),
  3. colours the cluster graph greedily at distance 2 (two clusters of one colour share no neighbour), so that probing
     S with the sum of one mode type over one colour gives every column of E = W^T S W exactly,
and checks on the dense reduced system of a sequential-capture scene with SHUFFLED camera numbers: probed E against
W^T S W, colours needed, PCG iterations (additive two-level) against index-range clusters on the unshuffled scene.

    python tools/exp_coarse_graph_clusters.py [num_cams num_pts]"""
import os
import sys
from collections import deque

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
os.chdir(ROOT)
sys.argv.append("additive")
import exp_coarse_space as C  # noqa: E402
import exp_precond as G  # noqa: E402
from glomap_amd import synthetic  # noqa: E402
from oracle import gp as ogp  # noqa: E402


def grow_clusters(N, off, cam, m):
    """Region growing: clusters in BFS order of their seeds; returns cluster id per camera."""
    nbr = [set() for _ in range(N)]
    for p in range(len(off) - 1):
        cs = cam[off[p] : off[p + 1]]
        for a, b in zip(cs[:-1], cs[1:]):
            if a != b:
                nbr[a].add(b)
                nbr[b].add(a)
    clu = -np.ones(N, dtype=np.int64)
    order = deque()
    seen = np.zeros(N, dtype=bool)
    nc = 0
    for root in range(N):  # every connected component
        if seen[root]:
            continue
        seen[root] = True
        order.append(root)
        while order:
            seed = order.popleft()
            if clu[seed] >= 0:
                continue
            # grow one cluster from `seed` among the unassigned cameras
            q = deque([seed])
            clu[seed] = nc
            size = 1
            while q and size < m:
                u = q.popleft()
                for v in sorted(nbr[u]):
                    if clu[v] < 0 and size < m:
                        clu[v] = nc
                        size += 1
                        q.append(v)
            # the unassigned rim of this cluster seeds the next ones (keeps the clusters in a sweep across the graph)
            for u in np.nonzero(clu == nc)[0]:
                for v in sorted(nbr[u]):
                    if clu[v] < 0 and not seen[v]:
                        seen[v] = True
                        order.append(v)
            nc += 1
    return clu, nc


def cluster_adjacency(nc, off, cam, clu):
    adj = [set() for _ in range(nc)]
    for p in range(len(off) - 1):
        qs = np.unique(clu[cam[off[p] : off[p + 1]]])
        for a in qs:
            for b in qs:
                if a != b:
                    adj[a].add(b)
    return adj


def colour_distance2(adj):
    nc = len(adj)
    col = -np.ones(nc, dtype=np.int64)
    for q in range(nc):
        taken = set()
        for a in adj[q]:
            taken.add(col[a])
            for b in adj[a]:
                taken.add(col[b])
        c = 0
        while c in taken:
            c += 1
        col[q] = c
    return col


def main():
    N = int(sys.argv[1]) if len(sys.argv) > 2 else 600
    P = int(sys.argv[2]) if len(sys.argv) > 3 else 60_000
    p = synthetic.make_gp_problem(num_cams=N, num_pts=P, seed=0, capture="sequential")
    rng = np.random.default_rng(1)
    perm = rng.permutation(N)  # new number of camera n
    opt = ogp.GlobalPositionerOptions()
    lens = np.diff(p.pt_offset)
    used = lens >= opt.min_num_view_per_track
    obs_pt = np.repeat(np.arange(P), lens)
    keep = used[obs_pt]
    remap = -np.ones(P, dtype=np.int64)
    remap[used] = np.arange(int(used.sum()))
    for label, relabel in (("capture order", np.arange(N)), ("shuffled numbers", perm)):
        cam = relabel[p.obs_cam[keep].astype(np.int64)]
        prob = ogp._GpProblem(N, cam, remap[obs_pt[keep]], p.obs_dir[keep], p.obs_calibrated[keep], opt, int(used.sum()))
        gt = np.zeros_like(p.gt_center)
        gt[relabel] = p.gt_center
        r2 = np.random.default_rng(0)
        c = gt + r2.normal(0, 1.0, gt.shape)
        X = p.gt_xyz[used] + r2.normal(0, 1.0, (int(used.sum()), 3))
        d = X[prob.pt] - c[prob.cam]
        s = np.maximum(1e-5, np.einsum("mj,mj->m", prob.v, d) / np.einsum("mj,mj->m", d, d))
        x = np.concatenate([c.ravel(), X.ravel(), s])
        # track-major CSR of the kept observations (tracks in their own order)
        order = np.argsort(prob.pt, kind="stable")
        cam_sorted = prob.cam[order]
        off = np.concatenate([[0], np.cumsum(np.bincount(prob.pt, minlength=prob.P))])
        S, b = G.schur_system(prob, x, 1e-6)
        Sd = S.toarray()
        Mi = np.stack([np.linalg.inv(Sd[3 * n : 3 * n + 3, 3 * n : 3 * n + 3]) for n in range(N)])
        it_bj, _ = C.pcg(Sd, b, Mi)
        line = f"{label}: block-Jacobi {it_bj}"
        for m in (16, 32):
            for kind in ("index ranges", "graph clusters"):
                if kind == "index ranges":
                    nc = (N + m - 1) // m
                    clu = (np.arange(N) * nc) // N
                else:
                    clu, nc = grow_clusters(N, off, cam_sorted, m)
                W = np.zeros((3 * N, 4 * nc))
                for q in range(nc):
                    mem = np.nonzero(clu == q)[0]
                    for a in range(3):
                        W[3 * mem + a, 4 * q + a] = 1.0
                    cc = c[mem] - c[mem].mean(0)
                    for a in range(3):
                        W[3 * mem + a, 4 * q + 3] = cc[:, a]
                E = W.T @ Sd @ W
                extra = ""
                if kind == "graph clusters":
                    adj = cluster_adjacency(nc, off, cam_sorted, clu)
                    col = colour_distance2(adj)
                    ncol = int(col.max()) + 1
                    Ep = np.zeros_like(E)
                    for cc_ in range(ncol):
                        for t in range(4):
                            z = W[:, [4 * q + t for q in range(nc) if col[q] == cc_]].sum(axis=1)
                            w = Sd @ z
                            for qq in range(nc):
                                owners = [q for q in list(adj[qq]) + [qq] if col[q] == cc_]
                                assert len(owners) <= 1
                                if owners:
                                    Ep[4 * qq : 4 * qq + 4, 4 * owners[0] + t] = W[:, 4 * qq : 4 * qq + 4].T @ w
                    extra = (f" [{nc} clusters, sizes {np.bincount(clu).min()}-{np.bincount(clu).max()}, {ncol} colours = {4 * ncol} probes, "
                             f"max |probed E - W^T S W| / max |E| = {np.abs(Ep - E).max() / np.abs(E).max():.1e}]")
                    E = Ep
                live = np.abs(W).sum(axis=0) > 0  # a cluster of one camera has no scale mode
                it = C.pcg_additive(Sd, b, Mi, W[:, live], E[np.ix_(live, live)])
                line += f"; m={m} {kind}: {it}{extra}"
        print(line, flush=True)


if __name__ == "__main__":
    main()
