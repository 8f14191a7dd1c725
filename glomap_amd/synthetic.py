"""Synthetic view graphs / track sets with ground truth.

The reference's own tests build their fixtures with ``colmap::SynthesizeDataset``
(glomap/controllers/global_mapper_test.cc:56-64, rotation_averager_test.cc:131-141), which is
un-vendored; this module is the stand-in generator.  Pure numpy, deterministic for a given seed.
"""
from __future__ import annotations

from dataclasses import dataclass

import numpy as np

from . import so3
from .flat import (
    CAMERA_SIMPLE_RADIAL,
    CAMERA_MAX_PARAMS,
    BaProblem,
    GpProblem,
    RaProblem,
)


# --------------------------------------------------------------------------------------------
# Rotation averaging (C2)
# --------------------------------------------------------------------------------------------
def make_ring_view_graph(
    num_cams: int = 1000,
    num_succ: int = 50,
    noise_deg: float = 1.0,
    outlier_ratio: float = 0.05,
    seed: int = 0,
    init: str = "identity",
) -> RaProblem:
    """Ring view graph: camera i is linked to its ``num_succ`` successors (mod N).

    GT rotation i = Exp(U(-0.2,0.2)^3) * RotY(2*pi*i/N);  R_ij = R_j R_i^T Exp(n),
    n ~ N(0, noise^2 I); ``outlier_ratio`` of the edges get a uniformly random rotation.
    """
    rng = np.random.default_rng(seed)
    N = num_cams
    yaw = 2 * np.pi * np.arange(N) / N
    aa_y = np.zeros((N, 3))
    aa_y[:, 1] = yaw
    R_gt = so3.aa_to_rotmat(rng.uniform(-0.2, 0.2, (N, 3))) @ so3.aa_to_rotmat(aa_y)

    ii = np.repeat(np.arange(N), num_succ)
    jj = (ii + np.tile(np.arange(1, num_succ + 1), N)) % N
    # COLMAP pair ids order the two images (image_id1 < image_id2); keep that invariant.
    lo = np.minimum(ii, jj)
    hi = np.maximum(ii, jj)
    # drop duplicates that appear when 2*num_succ >= N
    key = lo.astype(np.int64) * N + hi
    _, uniq = np.unique(key, return_index=True)
    uniq.sort()
    lo, hi = lo[uniq], hi[uniq]
    E = lo.shape[0]

    noise = so3.aa_to_rotmat(rng.normal(0.0, np.radians(noise_deg), (E, 3)))
    R_rel = R_gt[hi] @ np.transpose(R_gt[lo], (0, 2, 1)) @ noise
    outlier = rng.random(E) < outlier_ratio
    n_out = int(outlier.sum())
    if n_out:
        q_rand = rng.normal(size=(n_out, 4))
        q_rand /= np.linalg.norm(q_rand, axis=1, keepdims=True)
        R_rel[outlier] = so3.quat_to_rotmat(q_rand)
    edge_q = so3.rotmat_to_quat(R_rel)
    ninl = rng.integers(30, 501, E).astype(np.int32)

    if init == "identity":
        aa0 = np.zeros((N, 3))
    elif init == "gt_noisy":
        aa0 = so3.quat_to_aa(
            so3.rotmat_to_quat(R_gt @ so3.aa_to_rotmat(rng.normal(0, np.radians(5.0), (N, 3))))
        )
    else:
        raise ValueError(init)
    return RaProblem(
        num_nodes=N,
        edge_i=lo.astype(np.int32),
        edge_j=hi.astype(np.int32),
        edge_q=edge_q,
        edge_weight=np.ones(E),
        edge_ninl=ninl,
        node_aa0=aa0,
        fixed_node=0,
        gt_R=R_gt,
        outlier=outlier,
    )


def break_inlier_ties_by_index(edge_ninl: np.ndarray) -> np.ndarray:
    """Distinct inlier counts with the order (count descending, edge index ascending): the maximum spanning tree of the
    reference's start (tree.cc:78-153) is then unique.  With tied counts it depends on the order in which Boost's Kruskal
    pops equal-weight edges off its priority queue — an implementation detail of the Boost version the reference is built
    with; this library and the oracle break ties by edge index, which is what the returned counts encode."""
    E = edge_ninl.shape[0]
    order = np.lexsort((np.arange(E), -edge_ninl.astype(np.int64)))  # best edge first
    out = np.empty(E, dtype=np.int32)
    out[order] = np.arange(E, 0, -1, dtype=np.int32) + 10
    return out


def make_view_graph(
    kind: str = "geometric",
    num_cams: int = 1000,
    degree: int = 20,
    noise_deg: float = 1.0,
    outlier_ratio: float = 0.05,
    seed: int = 0,
    shuffle: bool = True,
    num_hubs: int = 8,
    chord_ratio: float = 0.05,
) -> RaProblem:
    """View graphs that are NOT banded rings (real view graphs have hubs, loop closures and arbitrary image ids):

      "geometric"  cameras scattered in the unit square, each linked to its `degree` nearest neighbours
      "hub"        the geometric graph plus `num_hubs` images linked to a quarter of all images each
      "chords"     a thin ring (`degree` // 2 successors) plus `chord_ratio` * E random long-range loop closures

    Same noise / outlier / inlier-count model as make_ring_view_graph.  `shuffle` permutes the node ids, so nothing can
    rely on index locality; node 0 of the result is the gauge node as everywhere else."""
    rng = np.random.default_rng(seed)
    N = num_cams
    R_gt = so3.aa_to_rotmat(rng.uniform(-np.pi / 2, np.pi / 2, (N, 3)) * np.array([0.3, 1.0, 0.3]))
    if kind in ("geometric", "hub"):
        from scipy.spatial import cKDTree

        xy = rng.random((N, 2))
        _, nb = cKDTree(xy).query(xy, k=min(N, degree // 2 + 1))
        ii = np.repeat(np.arange(N), nb.shape[1] - 1)
        jj = nb[:, 1:].reshape(-1)
        if kind == "hub":
            hubs = rng.choice(N, size=min(num_hubs, N), replace=False)
            for h in hubs:
                others = rng.choice(N, size=N // 4, replace=False)
                ii = np.concatenate([ii, np.full(others.shape[0], h)])
                jj = np.concatenate([jj, others])
    elif kind == "chords":
        succ = max(1, degree // 2)
        ii = np.repeat(np.arange(N), succ)
        jj = (ii + np.tile(np.arange(1, succ + 1), N)) % N
        nch = int(chord_ratio * ii.shape[0])
        ii = np.concatenate([ii, rng.integers(0, N, nch)])
        jj = np.concatenate([jj, rng.integers(0, N, nch)])
    else:
        raise ValueError(kind)
    keep = ii != jj
    lo, hi = np.minimum(ii[keep], jj[keep]), np.maximum(ii[keep], jj[keep])
    key = lo.astype(np.int64) * N + hi
    _, uniq = np.unique(key, return_index=True)
    uniq.sort()
    lo, hi = lo[uniq], hi[uniq]
    # keep the graph connected whatever the sampling did: a spanning chain over the (shuffled) node order
    if shuffle:
        perm = rng.permutation(N)
        R_gt = R_gt[np.argsort(perm)]
        lo, hi = perm[lo], perm[hi]
        lo, hi = np.minimum(lo, hi), np.maximum(lo, hi)
    import scipy.sparse as sp
    from scipy.sparse.csgraph import connected_components

    ncomp, lab = connected_components(sp.coo_matrix((np.ones(lo.shape[0]), (lo, hi)), shape=(N, N)), directed=False)
    if ncomp > 1:  # link component representatives in a chain
        reps = np.array([np.nonzero(lab == c)[0][0] for c in range(ncomp)])
        a, b = reps[:-1], reps[1:]
        lo = np.concatenate([lo, np.minimum(a, b)])
        hi = np.concatenate([hi, np.maximum(a, b)])
    E = lo.shape[0]
    noise = so3.aa_to_rotmat(rng.normal(0.0, np.radians(noise_deg), (E, 3)))
    R_rel = R_gt[hi] @ np.transpose(R_gt[lo], (0, 2, 1)) @ noise
    outlier = rng.random(E) < outlier_ratio
    n_out = int(outlier.sum())
    if n_out:
        q_rand = rng.normal(size=(n_out, 4))
        q_rand /= np.linalg.norm(q_rand, axis=1, keepdims=True)
        R_rel[outlier] = so3.quat_to_rotmat(q_rand)
    return RaProblem(
        num_nodes=N,
        edge_i=lo.astype(np.int32),
        edge_j=hi.astype(np.int32),
        edge_q=so3.rotmat_to_quat(R_rel),
        edge_weight=np.ones(E),
        edge_ninl=rng.integers(30, 501, E).astype(np.int32),
        node_aa0=np.zeros((N, 3)),
        fixed_node=0,
        gt_R=R_gt,
        outlier=outlier,
    )


# --------------------------------------------------------------------------------------------
# Shared camera / track geometry for C3 / C4
# --------------------------------------------------------------------------------------------
def _ring_cameras(rng, N, radius, jitter_deg=5.0):
    """Cameras on a ring of ``radius`` in the x-z plane looking at the origin (+z optical axis)."""
    phi = 2 * np.pi * np.arange(N) / N
    centers = np.stack([radius * np.cos(phi), rng.normal(0, 0.02 * radius, N), radius * np.sin(phi)], 1)
    zc = -centers / np.linalg.norm(centers, axis=1, keepdims=True)  # optical axis in world
    up = np.tile(np.array([0.0, 1.0, 0.0]), (N, 1))
    xc = np.cross(up, zc)
    xc /= np.linalg.norm(xc, axis=1, keepdims=True)
    yc = np.cross(zc, xc)
    R_cw = np.stack([xc, yc, zc], axis=1)  # rows = camera axes in world => cam_from_world
    R_cw = so3.aa_to_rotmat(rng.normal(0, np.radians(jitter_deg), (N, 3))) @ R_cw
    return centers, R_cw


def _ball_points(rng, P, radius):
    d = rng.normal(size=(P, 3))
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    r = radius * rng.random(P) ** (1.0 / 3.0)
    return d * r[:, None]


def _zipf_popularity(N, zipf, seed):
    """Per-camera popularity ~ rank^-zipf over a seeded random ranking (None when zipf == 0): real images differ by an
    order of magnitude and more in how many tracks they see; uniform synthetic visibility hides that."""
    if not zipf:
        return None
    rank = np.random.default_rng([seed, 424243]).permutation(N) + 1.0
    return rank ** (-float(zipf))


def _sample_tracks(rng, centers, R_cw, X, mean_extra, min_len=3, max_len=100, half_fov_deg=30.0, ncand=None,
                   chunk=65536, cam_popularity=None, sequential=False):
    """Pick, per point, L = min(min_len + Poisson(mean_extra), max_len) distinct cameras that see it
    inside the field of view.  Returns CSR (pt_offset, obs_cam) in track-major order; points that
    end up with fewer than ``min_len`` views keep what they have (the estimators skip them,
    reference gp.cc:258 / ba.cc:122).  Points are processed in fixed-size chunks (bounded memory;
    the chunk size is part of the definition of the sequence for a given seed)."""
    P = X.shape[0]
    N = centers.shape[0]
    cosfov = np.cos(np.radians(half_fov_deg))
    zaxis = np.ascontiguousarray(R_cw[:, 2, :])
    counts_all = np.zeros(P, dtype=np.int64)
    cams_all = []
    for p0 in range(0, P, chunk):
        Xc = X[p0 : p0 + chunk]
        Pc = Xc.shape[0]
        L = np.minimum(min_len + rng.poisson(mean_extra, Pc), min(max_len, N))
        nc = ncand if ncand is not None else int(min(N, max(16, 3 * int(L.max()))))
        if sequential:
            # sequential capture: a point is seen by a RUN of consecutive cameras (ring order) that starts at a random camera
            # near the one whose optical axis passes closest — what a walk-around capture produces, instead of L random
            # cameras among all that have the point in view
            # (the ring lies in the x-z plane, camera i at azimuth 2 pi i / N: a point is nearest to the optical axes of
            # the cameras at its own azimuth)
            az = np.arctan2(Xc[:, 2], Xc[:, 0])
            c0 = np.rint(az / (2 * np.pi) * N).astype(np.int64) + rng.integers(-nc // 2, 1, Pc)
            cand = (c0[:, None] + np.arange(nc)[None, :]) % N
        elif cam_popularity is None:
            cand = rng.integers(0, N, (Pc, nc))
        else:  # skewed visibility: candidates drawn with probability proportional to the camera's popularity
            cdf = np.cumsum(cam_popularity / cam_popularity.sum())
            cand = np.minimum(np.searchsorted(cdf, rng.random((Pc, nc))), N - 1)
        if not sequential:
            cand.sort(axis=1)
        dup = np.zeros_like(cand, dtype=bool)
        dup[:, 1:] = cand[:, 1:] == cand[:, :-1]
        d = Xc[:, None, :] - centers[cand]  # [Pc,nc,3]
        depth = np.einsum("pkj,pkj->pk", d, zaxis[cand])
        nrm = np.linalg.norm(d, axis=2)
        vis = (depth > cosfov * nrm) & ~dup
        # random order among the visible candidates (sequential: the run's own order), invisible ones last
        score = (np.arange(nc)[None, :] / (nc + 1.0) if sequential else rng.random((Pc, nc))) + (~vis) * 10.0
        order = np.argsort(score, axis=1)
        cand = np.take_along_axis(cand, order, axis=1)
        vis = np.take_along_axis(vis, order, axis=1)
        take = (np.arange(nc)[None, :] < L[:, None]) & vis
        counts_all[p0 : p0 + Pc] = take.sum(axis=1)
        cams_all.append(cand[take].astype(np.int32))  # row-major boolean indexing == track-major
    pt_offset = np.zeros(P + 1, dtype=np.int64)
    np.cumsum(counts_all, out=pt_offset[1:])
    obs_cam = np.concatenate(cams_all) if cams_all else np.zeros(0, dtype=np.int32)
    return pt_offset, obs_cam


def _sort_tracks_by_first_camera(pt_offset, obs_cam, X):
    """Track order of an incremental capture: tracks sorted by the first camera that sees them (stable).  Returns the
    permuted CSR, points and the permutation."""
    P = X.shape[0]
    lens = np.diff(pt_offset)
    first = np.full(P, np.iinfo(np.int32).max, dtype=np.int64)
    has = lens > 0
    first[has] = np.minimum.reduceat(obs_cam, pt_offset[:-1][has])
    perm = np.argsort(first, kind="stable")
    new_off = np.zeros(P + 1, dtype=np.int64)
    np.cumsum(lens[perm], out=new_off[1:])
    idx = np.repeat(pt_offset[:-1][perm] - new_off[:-1], lens[perm]) + np.arange(new_off[-1])  # the observation ranges, gathered
    return new_off, obs_cam[idx], X[perm], perm


def make_gp_problem(
    num_cams: int = 5000,
    num_pts: int = 500_000,
    mean_extra: float = 3.0,
    dir_noise: float = 1e-3,
    outlier_ratio: float = 0.02,
    uncalibrated_ratio: float = 0.0,
    seed: int = 0,
    shard=None,
    zipf: float = 0.0,
    capture: str = "random",
) -> GpProblem:
    """C3-style global positioning problem (cameras on a radius-50 ring, points in a radius-30 ball).
    capture = "sequential": every point is seen by a run of consecutive cameras and the tracks come in the order of the
    first camera that sees them (a walk-around capture: co-visible points are neighbours in memory), instead of L random
    cameras among the ~1/6 of the ring that has the point in view and tracks in random order.

    shard = (rank, world): generate only this rank's `num_pts` tracks of a `world * num_pts`-track
    problem — the cameras come from `seed` alone (identical on every rank), the points from a
    rank-specific stream — so weak-scaling runs never materialise the whole problem on one host."""
    rng = np.random.default_rng(seed)
    centers, R_cw = _ring_cameras(rng, num_cams, 50.0)
    if shard is not None:
        calibrated_all = (rng.random(num_cams) >= uncalibrated_ratio).astype(np.uint8)
        rng = np.random.default_rng([seed, 7919, int(shard[0])])
    X = _ball_points(rng, num_pts, 30.0)
    pt_offset, obs_cam = _sample_tracks(rng, centers, R_cw, X, mean_extra, cam_popularity=_zipf_popularity(num_cams, zipf, seed),
                                        sequential=capture == "sequential")
    if capture == "sequential":
        pt_offset, obs_cam, X, _ = _sort_tracks_by_first_camera(pt_offset, obs_cam, X)
    M = obs_cam.shape[0]
    obs_pt = np.repeat(np.arange(num_pts), np.diff(pt_offset))
    d = X[obs_pt] - centers[obs_cam]
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    # tangent noise, then renormalise (features_undist are unit rays, image_undistorter.cc:33-38)
    n = rng.normal(0, dir_noise, (M, 3))
    n -= np.einsum("mj,mj->m", n, d)[:, None] * d
    d = d + n
    out = rng.random(M) < outlier_ratio
    n_out = int(out.sum())
    if n_out:
        r = rng.normal(size=(n_out, 3))
        d[out] = r
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    calibrated = calibrated_all if shard is not None else (rng.random(num_cams) >= uncalibrated_ratio).astype(np.uint8)
    return GpProblem(
        num_cams=num_cams,
        num_pts=num_pts,
        pt_offset=pt_offset,
        obs_cam=obs_cam,
        obs_dir=np.ascontiguousarray(d),
        obs_calibrated=calibrated[obs_cam],
        cam_center=np.zeros((num_cams, 3)),
        pt_xyz=np.zeros((num_pts, 3)),
        cam_R=R_cw,
        gt_center=centers,
        gt_xyz=X,
    )


def project_simple_radial(params, xc):
    f, cx, cy, k = params[..., 0], params[..., 1], params[..., 2], params[..., 3]
    u = xc[..., 0] / xc[..., 2]
    v = xc[..., 1] / xc[..., 2]
    r2 = u * u + v * v
    rad = 1.0 + k * r2
    return np.stack([f * u * rad + cx, f * v * rad + cy], axis=-1)


def project_full_opencv(params, xc):
    """COLMAP FULL_OPENCV (models.h): params [M,12+] = fx, fy, cx, cy, k1, k2, p1, p2, k3, k4, k5, k6; xc [M,3] in the camera frame."""
    u, v = xc[:, 0] / xc[:, 2], xc[:, 1] / xc[:, 2]
    fx, fy, cx, cy, k1, k2, p1, p2, k3, k4, k5, k6 = (params[:, i] for i in range(12))
    u2, v2, uv = u * u, v * v, u * v
    r2 = u2 + v2
    r4, r6 = r2 * r2, r2 * r2 * r2
    radial = (1.0 + k1 * r2 + k2 * r4 + k3 * r6) / (1.0 + k4 * r2 + k5 * r4 + k6 * r6)
    du = u * radial + 2.0 * p1 * uv + p2 * (r2 + 2.0 * u2) - u
    dv = v * radial + 2.0 * p2 * uv + p1 * (r2 + 2.0 * v2) - v
    return np.stack([fx * (u + du) + cx, fy * (v + dv) + cy], 1)


def make_ba_problem_wide(num_cams: int = 10_000, num_pts: int = 1_000_000, model: str = "full_opencv", seed: int = 0,
                         shared_intrinsics: bool = False, pixel_noise: float = 0.5, num_intr_groups: int = 0, **kw) -> BaProblem:
    """make_ba_problem's scene observed through a 12-parameter FULL_OPENCV camera (16-wide intrinsics rows: the unit of
    csrc/ba_wide.hip; the reference dispatches any CameraModelId, bundle_adjustment.cc:136-139): same cameras, points, tracks,
    perturbed start and outliers; the observations of the inlier tracks re-projected through the wide model + pixel noise.
    num_intr_groups = G > 0: G cameras shared by the images round-robin (default: one camera per image)."""
    from .flat import CAMERA_FULL_OPENCV, CAMERA_MAX_PARAMS_WIDE

    assert model == "full_opencv"
    p = make_ba_problem(num_cams, num_pts, seed=seed, shared_intrinsics=shared_intrinsics, pixel_noise=pixel_noise, **kw)
    K = p.num_intr
    vals = np.array([1200.0, 1190.0, 640.0, 480.0, 0.02, -0.01, 0.001, -0.002, 0.003, 0.01, -0.004, 0.002])
    wide = np.zeros((K, CAMERA_MAX_PARAMS_WIDE))
    wide[:, :12] = vals
    obs_pt = np.repeat(np.arange(p.num_pts), np.diff(p.pt_offset))
    R = so3.quat_to_rotmat(p.gt_q)
    xc = np.einsum("mij,mj->mi", R[p.obs_cam], p.gt_xyz[obs_pt]) + p.gt_t[p.obs_cam]
    old = project_simple_radial(p.gt_intr[p.cam_intr[p.obs_cam]], xc)
    resid = p.obs_xy - old  # the noise / outlier offsets of the original observations, kept
    inl = np.abs(resid).max(1) < 10.0 * max(pixel_noise, 1e-9) + 1e-9
    xy = project_full_opencv(wide[p.cam_intr[p.obs_cam]], xc)
    p.obs_xy = np.ascontiguousarray(np.where(inl[:, None], xy + resid, p.obs_xy))
    if num_intr_groups > 0 and not shared_intrinsics:
        # G physical cameras, image n taken with camera n mod G (a capture with a handful of devices): twelve parameters per
        # camera are determined by N / G images each instead of by one image's few hundred observations
        K = min(num_intr_groups, num_cams)
        p.cam_intr = (np.arange(num_cams) % K).astype(np.int32)
        p.num_intr = K
        wide = wide[:K]
    p.intr_model = np.full(K, CAMERA_FULL_OPENCV, dtype=np.int32)
    p.intr_params = wide.copy()
    p.gt_intr = wide
    return p


def make_ba_problem(
    num_cams: int = 10_000,
    num_pts: int = 1_000_000,
    mean_extra: float = 2.0,
    pixel_noise: float = 0.5,
    outlier_ratio: float = 0.01,
    shared_intrinsics: bool = False,
    rot_noise_deg: float = 0.5,
    pos_noise: float = 0.01,
    depth_noise: float = 0.01,
    intr_noise: float = 0.0,
    seed: int = 0,
    shard=None,
    zipf: float = 0.0,
    capture: str = "random",
) -> BaProblem:
    """C4-style bundle-adjustment problem: SIMPLE_RADIAL (f=1200,cx=640,cy=480,k=0.02), state =
    ground truth perturbed by rotation / position / depth noise.  capture = "sequential": see make_gp_problem.

    shard = (rank, world): only this rank's `num_pts` tracks are generated; cameras, intrinsics and
    their perturbed start come from `seed` alone and are identical on every rank."""
    rng = np.random.default_rng(seed)
    N, P = num_cams, num_pts
    radius = 50.0
    centers, R_cw = _ring_cameras(rng, N, radius)
    rng_cam = None
    if shard is not None:
        rng_cam = np.random.default_rng([seed, 104729])
        rng = np.random.default_rng([seed, 7919, int(shard[0])])
    X = _ball_points(rng, P, 30.0)
    pt_offset, obs_cam = _sample_tracks(rng, centers, R_cw, X, mean_extra, half_fov_deg=25.0,
                                        cam_popularity=_zipf_popularity(N, zipf, seed), sequential=capture == "sequential")
    if capture == "sequential":
        pt_offset, obs_cam, X, _ = _sort_tracks_by_first_camera(pt_offset, obs_cam, X)
    M = obs_cam.shape[0]
    obs_pt = np.repeat(np.arange(P), np.diff(pt_offset))
    t_gt = -np.einsum("nij,nj->ni", R_cw, centers)
    K = 1 if shared_intrinsics else N
    intr_gt = np.zeros((K, CAMERA_MAX_PARAMS))
    intr_gt[:, :4] = np.array([1200.0, 640.0, 480.0, 0.02])
    cam_intr = np.zeros(N, dtype=np.int32) if shared_intrinsics else np.arange(N, dtype=np.int32)

    xc = np.einsum("mij,mj->mi", R_cw[obs_cam], X[obs_pt]) + t_gt[obs_cam]
    xy = project_simple_radial(intr_gt[cam_intr[obs_cam]], xc)
    xy += rng.normal(0, pixel_noise, (M, 2))
    out = rng.random(M) < outlier_ratio
    n_out = int(out.sum())
    if n_out:
        xy[out] = np.stack([rng.uniform(0, 1280, n_out), rng.uniform(0, 960, n_out)], 1)

    # perturbed start
    rc = rng_cam if rng_cam is not None else rng
    R0 = so3.aa_to_rotmat(rc.normal(0, np.radians(rot_noise_deg), (N, 3))) @ R_cw
    c0 = centers + rc.normal(0, pos_noise * radius, (N, 3))
    t0 = -np.einsum("nij,nj->ni", R0, c0)
    X0 = X * (1.0 + rng.normal(0, depth_noise, (P, 1))) + rng.normal(0, depth_noise * 1.0, (P, 3))
    intr0 = intr_gt.copy()
    if intr_noise > 0:
        intr0[:, 0] *= 1.0 + rc.normal(0, intr_noise, K)
    return BaProblem(
        num_cams=N,
        num_pts=P,
        num_intr=K,
        pt_offset=pt_offset,
        obs_cam=obs_cam,
        obs_xy=np.ascontiguousarray(xy),
        cam_intr=cam_intr,
        cam_q=so3.rotmat_to_quat(R0),
        cam_t=t0,
        pt_xyz=X0,
        intr_model=np.full(K, CAMERA_SIMPLE_RADIAL, dtype=np.int32),
        intr_params=intr0,
        fixed_cam=0,
        gt_q=so3.rotmat_to_quat(R_cw),
        gt_t=t_gt,
        gt_xyz=X,
        gt_intr=intr_gt,
    )


# --------------------------------------------------------------------------------------------
# One scene for the whole hot path: RA -> GP -> BA chained as GlobalMapper::Solve chains them
# (global_mapper.cc:92-223: rotations from RA orient the bearings GP reads, GP's centres / points start BA)
# --------------------------------------------------------------------------------------------
@dataclass
class ChainedScene:
    """View graph + pixel observations of ONE synthetic scene; the three stage problems are derived from it by
    chain_gp_problem / chain_ba_problem, each from the previous stage's RESULT (not from ground truth)."""

    ra: RaProblem
    num_cams: int
    num_pts: int
    pt_offset: np.ndarray  # [P+1]
    obs_cam: np.ndarray  # [M] int32
    obs_xy: np.ndarray  # [M,2] pixels (SIMPLE_RADIAL, noise, gross outliers)
    intr: np.ndarray  # [N,8] intrinsics BA starts from (the calibration GP undistorts with)
    gt_R: np.ndarray  # [N,3,3] cam_from_world
    gt_center: np.ndarray  # [N,3]
    gt_xyz: np.ndarray  # [P,3]


def unproject_simple_radial(params, xy, iterations=12):
    """Pixel -> unit bearing in the camera frame for SIMPLE_RADIAL (what image_undistorter.cc:33-38 stores in
    features_undist): Newton on r_d = r (1 + k r^2), a fixed number of iterations (r_d < 0.7, k = 0.02: converged to
    rounding after 5)."""
    f, cx, cy, k = params[..., 0], params[..., 1], params[..., 2], params[..., 3]
    ud = (xy[..., 0] - cx) / f
    vd = (xy[..., 1] - cy) / f
    rd = np.sqrt(ud * ud + vd * vd)
    r = rd.copy()
    for _ in range(iterations):
        r = r - (r * (1.0 + k * r * r) - rd) / (1.0 + 3.0 * k * r * r)
    sc = np.where(rd > 0, r / np.where(rd > 0, rd, 1.0), 1.0)
    b = np.stack([ud * sc, vd * sc, np.ones_like(ud)], axis=-1)
    return b / np.linalg.norm(b, axis=-1, keepdims=True)


def make_chained_scene(num_cams=10_000, num_pts=1_000_000, seed=0, num_succ=50, rot_noise_deg=1.0, rot_outlier_ratio=0.05,
                       mean_extra=2.0, pixel_noise=0.5, outlier_ratio=0.01) -> ChainedScene:
    """configs[3]-shaped scene for the chained parity test: the ring cameras / ball points of make_ba_problem, a ring view
    graph whose relative rotations come from THESE cameras' rotations (make_ring_view_graph's noise model), SIMPLE_RADIAL
    pixel observations with noise and gross outliers."""
    rng = np.random.default_rng([seed, 271828])
    N, P = int(num_cams), int(num_pts)
    centers, R_cw = _ring_cameras(rng, N, 50.0)
    num_succ = min(num_succ, (N - 1) // 2)
    ii = np.repeat(np.arange(N), num_succ)
    jj = (ii + np.tile(np.arange(1, num_succ + 1), N)) % N
    lo, hi = np.minimum(ii, jj), np.maximum(ii, jj)
    E = lo.shape[0]
    R_rel = R_cw[hi] @ np.transpose(R_cw[lo], (0, 2, 1)) @ so3.aa_to_rotmat(rng.normal(0.0, np.radians(rot_noise_deg), (E, 3)))
    bad = rng.random(E) < rot_outlier_ratio
    if bad.any():
        qr = rng.normal(size=(int(bad.sum()), 4))
        R_rel[bad] = so3.quat_to_rotmat(qr / np.linalg.norm(qr, axis=1, keepdims=True))
    ra = RaProblem(num_nodes=N, edge_i=lo.astype(np.int32), edge_j=hi.astype(np.int32), edge_q=so3.rotmat_to_quat(R_rel),
                   edge_weight=np.ones(E), edge_ninl=rng.integers(30, 501, E).astype(np.int32), node_aa0=np.zeros((N, 3)),
                   fixed_node=0, gt_R=R_cw, outlier=bad)
    X = _ball_points(rng, P, 30.0)
    pt_offset, obs_cam = _sample_tracks(rng, centers, R_cw, X, mean_extra, half_fov_deg=25.0)
    M = obs_cam.shape[0]
    obs_pt = np.repeat(np.arange(P), np.diff(pt_offset))
    intr = np.zeros((N, CAMERA_MAX_PARAMS))
    intr[:, :4] = np.array([1200.0, 640.0, 480.0, 0.02])
    xc = np.einsum("mij,mj->mi", R_cw[obs_cam], X[obs_pt] - centers[obs_cam])
    xy = project_simple_radial(intr[obs_cam], xc) + rng.normal(0, pixel_noise, (M, 2))
    out = rng.random(M) < outlier_ratio
    if out.any():
        xy[out] = np.stack([rng.uniform(0, 1280, int(out.sum())), rng.uniform(0, 960, int(out.sum()))], 1)
    return ChainedScene(ra=ra, num_cams=N, num_pts=P, pt_offset=pt_offset, obs_cam=obs_cam, obs_xy=np.ascontiguousarray(xy),
                        intr=intr, gt_R=R_cw, gt_center=centers, gt_xyz=X)


def chain_gp_problem(scene: ChainedScene, R_est: np.ndarray) -> GpProblem:
    """Global positioning as global_mapper.cc:157-163 poses it: bearings = the undistorted features rotated into the world
    frame by the rotations ROTATION AVERAGING returned (cost_function.h:15-41 reads R^T * feature_undist), centres and points
    drawn at random by the solver."""
    b = unproject_simple_radial(scene.intr[scene.obs_cam], scene.obs_xy)
    d = np.einsum("mji,mj->mi", R_est[scene.obs_cam], b)
    return GpProblem(num_cams=scene.num_cams, num_pts=scene.num_pts, pt_offset=scene.pt_offset, obs_cam=scene.obs_cam,
                     obs_dir=np.ascontiguousarray(d), obs_calibrated=np.ones(scene.obs_cam.shape[0], np.uint8),
                     cam_center=np.zeros((scene.num_cams, 3)), pt_xyz=np.zeros((scene.num_pts, 3)), cam_R=R_est,
                     gt_center=scene.gt_center, gt_xyz=scene.gt_xyz)


# --------------------------------------------------------------------------------------------
# Calibrated multi-camera rigs (the shape of global_mapper_test.cc:89-126 WithoutNoiseWithNonTrivialKnownRig)
# --------------------------------------------------------------------------------------------
def make_rig_problems(
    num_frames: int = 14,
    cams_per_rig: int = 2,
    num_pts: int = 400,
    num_rigs: int = 2,
    sensor_rot_deg: float = 5.0,
    sensor_trans: float = 0.1,
    pixel_noise: float = 0.0,
    dir_noise: float = 0.0,
    outlier_ratio: float = 0.0,
    seed: int = 0,
    mean_extra: float = 3.0,
):
    """Frames (rig poses) on the usual ring, `cams_per_rig` sensors per frame: sensor 0 is the reference sensor
    (cam_from_rig = identity), the others are rotated by ~`sensor_rot_deg` and shifted by ~`sensor_trans` * ring radius / 50.
    Frames alternate between `num_rigs` rigs with different calibrations.  Images = frames x sensors, one SIMPLE_RADIAL
    camera per (rig, sensor).  Returns (GpProblem, BaProblem, info) over the SAME images and tracks:
    GP with image_frame / image_offset, BA with image_frame / image_cam_from_rig / image_intr; info holds ground truth."""
    rng = np.random.default_rng(seed)
    N, S = num_frames, cams_per_rig
    radius = 50.0
    centers_f, R_f = _ring_cameras(rng, N, radius)  # rig_from_world rotation, rig centre
    t_f = -np.einsum("nij,nj->ni", R_f, centers_f)
    rig_of_frame = np.arange(N) % num_rigs
    # calibrations: [rig][sensor] cam_from_rig
    Rs = np.tile(np.eye(3), (num_rigs, S, 1, 1))
    ts = np.zeros((num_rigs, S, 3))
    for r in range(num_rigs):
        for s_ in range(1, S):
            Rs[r, s_] = so3.aa_to_rotmat(rng.normal(0, np.radians(sensor_rot_deg), 3))
            ts[r, s_] = rng.normal(0, sensor_trans * radius / 50.0 * 10.0, 3)
    I = N * S
    image_frame = np.repeat(np.arange(N), S).astype(np.int32)
    image_sensor = np.tile(np.arange(S), N)
    R_s = Rs[rig_of_frame[image_frame], image_sensor]
    t_s = ts[rig_of_frame[image_frame], image_sensor]
    R_cw = R_s @ R_f[image_frame]
    t_cw = np.einsum("iab,ib->ia", R_s, t_f[image_frame]) + t_s
    c_img = -np.einsum("iba,ib->ia", R_cw, t_cw)
    X = _ball_points(rng, num_pts, 30.0)
    pt_offset, obs_img = _sample_tracks(rng, c_img, R_cw, X, mean_extra, half_fov_deg=28.0)
    M = obs_img.shape[0]
    obs_pt = np.repeat(np.arange(num_pts), np.diff(pt_offset))
    K = num_rigs * S
    image_intr = (rig_of_frame[image_frame] * S + image_sensor).astype(np.int32)
    intr_gt = np.zeros((K, CAMERA_MAX_PARAMS))
    intr_gt[:, :4] = np.array([1200.0, 640.0, 480.0, 0.02])
    xc = np.einsum("mij,mj->mi", R_cw[obs_img], X[obs_pt]) + t_cw[obs_img]
    xy = project_simple_radial(intr_gt[image_intr[obs_img]], xc)
    xy += rng.normal(0, pixel_noise, (M, 2)) if pixel_noise else 0.0
    out = rng.random(M) < outlier_ratio
    if out.any():
        xy[out] = np.stack([rng.uniform(0, 1280, int(out.sum())), rng.uniform(0, 960, int(out.sum()))], 1)
    ray = xc / np.linalg.norm(xc, axis=1, keepdims=True)
    if dir_noise:
        n = rng.normal(0, dir_noise, (M, 3))
        n -= np.einsum("mj,mj->m", n, ray)[:, None] * ray
        ray = ray + n
        ray /= np.linalg.norm(ray, axis=1, keepdims=True)
    obs_dir = np.einsum("mji,mj->mi", R_cw[obs_img], ray)  # R_cw^T * features_undist (gp.cc:294-296)
    image_offset = np.einsum("iba,ib->ia", R_cw, t_s)  # R_cw^T t_cam_from_rig (gp.cc:329-333)
    q_s = so3.rotmat_to_quat(R_s)
    gp = GpProblem(num_cams=N, num_pts=num_pts, pt_offset=pt_offset, obs_cam=obs_img.astype(np.int32),
                   obs_dir=np.ascontiguousarray(obs_dir), obs_calibrated=np.ones(M, np.uint8), cam_center=np.zeros((N, 3)),
                   pt_xyz=np.zeros((num_pts, 3)), cam_R=R_f, gt_center=centers_f, gt_xyz=X,
                   image_frame=image_frame, image_offset=np.ascontiguousarray(image_offset))
    # BA start: perturbed frame poses / points, exact calibration
    R0 = so3.aa_to_rotmat(rng.normal(0, np.radians(0.5), (N, 3))) @ R_f
    c0 = centers_f + rng.normal(0, 0.01 * radius, (N, 3))
    ba = BaProblem(num_cams=N, num_pts=num_pts, num_intr=K, pt_offset=pt_offset, obs_cam=obs_img.astype(np.int32),
                   obs_xy=np.ascontiguousarray(xy), cam_intr=np.zeros(N, np.int32), cam_q=so3.rotmat_to_quat(R0),
                   cam_t=-np.einsum("nij,nj->ni", R0, c0), pt_xyz=X * (1.0 + rng.normal(0, 0.01, (num_pts, 1))),
                   intr_model=np.full(K, CAMERA_SIMPLE_RADIAL, dtype=np.int32), intr_params=intr_gt.copy(), fixed_cam=0,
                   gt_q=so3.rotmat_to_quat(R_f), gt_t=t_f, gt_xyz=X, gt_intr=intr_gt, image_frame=image_frame,
                   image_cam_from_rig=np.ascontiguousarray(np.concatenate([q_s, t_s], axis=1)), image_intr=image_intr)
    # sensor blocks (ba.cc:161-179): one per (rig, non-reference sensor); -1 for the images of reference sensors
    blk = np.where(image_sensor > 0, rig_of_frame[image_frame] * (S - 1) + image_sensor - 1, -1).astype(np.int32)
    sens_gt = np.zeros((num_rigs * (S - 1), 7))
    for r in range(num_rigs):
        for s_ in range(1, S):
            sens_gt[r * (S - 1) + s_ - 1] = np.concatenate([so3.rotmat_to_quat(Rs[r, s_][None])[0], ts[r, s_]])
    info = dict(R_cw=R_cw, t_cw=t_cw, R_s=R_s, t_s=t_s, image_sensor=image_sensor, rig_of_frame=rig_of_frame,
                sensor_block=blk, sensor_cam_from_rig=sens_gt)
    return gp, ba, info


def forget_rig_translations(gp: GpProblem, info) -> GpProblem:
    """The GP problem of make_rig_problems with the cam_from_rig TRANSLATIONS of all non-reference sensors unknown
    (global_positioning.cc:354-368): their images lose the offset and point at a centre block instead; the rotations
    (folded into obs_dir) stay known, as they are after rotation averaging."""
    import copy

    p = copy.deepcopy(gp)
    blk = info["sensor_block"]
    p.image_sensor = blk.copy()
    p.image_offset = np.where((blk >= 0)[:, None], 0.0, gp.image_offset)
    p.image_sensor_rot = np.ascontiguousarray(gp.cam_R[gp.image_frame])
    sg = info["sensor_cam_from_rig"]
    Rq = so3.quat_to_rotmat(sg[:, :4])
    p.sensor_center = np.zeros((sg.shape[0], 3))
    info["sensor_center"] = -np.einsum("sji,sj->si", Rq, sg[:, 4:])  # ground truth: -R^T t
    return p


# --------------------------------------------------------------------------------------------
# Gauge-free comparisons (reference: rotation_averager_test.cc:85-106, global_mapper_test.cc:26-38)
# --------------------------------------------------------------------------------------------
def align_rotations(R_est: np.ndarray, R_ref: np.ndarray) -> np.ndarray:
    """Best global right-rotation G (3x3) such that R_est @ G ~= R_ref (chordal mean)."""
    Mx = np.einsum("nji,njk->ik", R_est, R_ref)
    U, _, Vt = np.linalg.svd(Mx)
    D = np.diag([1.0, 1.0, np.sign(np.linalg.det(U @ Vt))])
    return U @ D @ Vt


def rotation_errors_deg(R_est: np.ndarray, R_ref: np.ndarray) -> np.ndarray:
    G = align_rotations(R_est, R_ref)
    return so3.rotation_angle_deg(R_est @ G, R_ref)


def align_sim3(src: np.ndarray, dst: np.ndarray):
    """Umeyama similarity: returns (s, R, t) minimising |s R src + t - dst|."""
    mu_s, mu_d = src.mean(0), dst.mean(0)
    a, b = src - mu_s, dst - mu_d
    H = a.T @ b / src.shape[0]
    U, S, Vt = np.linalg.svd(H)
    D = np.diag([1.0, 1.0, np.sign(np.linalg.det(Vt.T @ U.T))])
    R = Vt.T @ D @ U.T
    s = np.trace(np.diag(S) @ D) / (a * a).sum() * src.shape[0]
    t = mu_d - s * R @ mu_s
    return s, R, t


def scene_extent(c: np.ndarray) -> float:
    """Largest distance of a camera centre from the centroid: the length north_star's "1e-3 relative" is relative to."""
    return float(np.linalg.norm(c - c.mean(0), axis=1).max())


def center_errors_after_sim3(est: np.ndarray, ref: np.ndarray) -> np.ndarray:
    """Per-camera centre error after Sim(3) alignment, ALREADY divided by the extent of ``ref`` (scene_extent): callers
    compare the returned numbers with a relative bar directly and must not divide by the extent a second time
    (rounds 1-4 did at a dozen call sites, which made the GP gauge extent of ~20 hide a factor of 20)."""
    s, R, t = align_sim3(est, ref)
    al = (s * (R @ est.T)).T + t
    return np.linalg.norm(al - ref, axis=1) / scene_extent(ref)


def center_distance_stats(est: np.ndarray, ref: np.ndarray) -> dict:
    """max / 99th percentile / median of center_errors_after_sim3 (relative to the extent of ``ref``, divided once)."""
    d = center_errors_after_sim3(est, ref)
    return {"max": float(d.max()), "p99": float(np.percentile(d, 99)), "median": float(np.median(d))}


def make_match_graph(n_images=200, n_tracks=5000, seed=0, max_gap=12, ring=50, match_prob=0.7, false_match_frac=0.01,
                     twin_frac=0.005, mean_extra=3.0, max_len=24, width=1280.0, height=960.0):
    """Synthetic input of track establishment (track_establishment.cc:19-63 reads `image_pair.matches` rows selected
    by `image_pair.inliers`, and `image.features`): a ring view graph (image i paired with its `ring` successors),
    ground-truth tracks whose members sit `1..max_gap` images apart, each co-visible member pair matched with
    probability `match_prob`; `false_match_frac` of the matches join two unrelated features (merging tracks, most of
    which then fail the same-image consistency test), `twin_frac` of the members get a twin feature 1-3 px away in the
    same image that is matched too (consistent duplicates, which the reference keeps).  Every image also has as many
    unmatched features (odd feature indices) as matched ones.

    Returns a dict of flat arrays in the layout of gsfm_match_graph (include/gsfm.h)."""
    rng = np.random.default_rng(seed)
    N = int(n_images)
    ring = min(ring, (N - 1) // 2)
    L = np.minimum(3 + rng.poisson(mean_extra, n_tracks), max_len).astype(np.int64)
    off = np.zeros(n_tracks + 1, dtype=np.int64)
    off[1:] = np.cumsum(L)
    M = int(off[-1])
    trk = np.repeat(np.arange(n_tracks), L)
    pos = np.arange(M) - off[trk]
    gaps = rng.integers(1, max_gap + 1, M)
    gaps[off[:-1]] = 0
    csum = np.cumsum(gaps)
    rel = csum - csum[off[trk]]
    span_ok = rel < N  # members of one track are distinct images as long as the span stays below N
    base = rng.integers(0, N, n_tracks)
    cam = ((base[trk] + rel) % N).astype(np.int64)
    # feature index inside the image: 2 * rank among the image's members (odd indices = unmatched features)
    order = np.argsort(cam, kind="stable")
    counts = np.bincount(cam, minlength=N)
    start = np.zeros(N + 1, dtype=np.int64)
    start[1:] = np.cumsum(counts)
    rank = np.empty(M, dtype=np.int64)
    rank[order] = np.arange(M) - start[cam[order]]
    n_twin = int(twin_frac * M)
    twin_of = rng.choice(M, n_twin, replace=False) if n_twin else np.zeros(0, dtype=np.int64)
    # twins take the feature slots after the regular ones of their image
    tcam = cam[twin_of]
    torder = np.argsort(tcam, kind="stable")
    tcount = np.bincount(tcam, minlength=N)
    tstart = np.zeros(N + 1, dtype=np.int64)
    tstart[1:] = np.cumsum(tcount)
    trank = np.empty(n_twin, dtype=np.int64)
    trank[torder] = np.arange(n_twin) - tstart[tcam[torder]]
    nfeat = 2 * counts + tcount
    feat_offset = np.zeros(N + 1, dtype=np.int64)
    feat_offset[1:] = np.cumsum(nfeat)
    F = int(feat_offset[-1])
    feat = 2 * rank
    tfeat = 2 * counts[tcam] + trank
    xy = np.stack([rng.uniform(0, width, F), rng.uniform(0, height, F)], axis=1)
    d = rng.uniform(1.0, 3.0, n_twin)
    a = rng.uniform(0, 2 * np.pi, n_twin)
    xy[feat_offset[tcam] + tfeat] = xy[feat_offset[cam[twin_of]] + feat[twin_of]] + np.stack([d * np.cos(a), d * np.sin(a)], 1)

    m_i1, m_i2, m_f1, m_f2 = [], [], [], []

    def emit(ia, fa, ib, fb):
        dd = (ib - ia) % N
        fwd = (dd >= 1) & (dd <= ring)
        bwd = ((-dd) % N >= 1) & ((-dd) % N <= ring) & ~fwd
        for sel, (i1, f1, i2, f2) in ((fwd, (ia, fa, ib, fb)), (bwd, (ib, fb, ia, fa))):
            m_i1.append(i1[sel]); m_f1.append(f1[sel]); m_i2.append(i2[sel]); m_f2.append(f2[sel])

    Lmax = int(L.max()) if n_tracks else 0
    for s in range(1, Lmax):
        a_idx = np.nonzero((pos + s < L[trk]) & span_ok)[0]
        b_idx = a_idx + s
        keep = rng.random(len(a_idx)) < match_prob
        a_idx, b_idx = a_idx[keep], b_idx[keep]
        emit(cam[a_idx], feat[a_idx], cam[b_idx], feat[b_idx])
    # twins: matched to one other member of the track of their original
    if n_twin:
        o = twin_of
        partner = np.where(pos[o] + 1 < L[trk[o]], o + 1, o - 1)
        emit(tcam, tfeat, cam[partner], feat[partner])
    i1 = np.concatenate(m_i1); f1 = np.concatenate(m_f1); i2 = np.concatenate(m_i2); f2 = np.concatenate(m_f2)
    n_false = int(false_match_frac * len(i1))
    if n_false:
        a_idx = rng.integers(0, M, n_false)
        dd = rng.integers(1, ring + 1, n_false)
        target = (cam[a_idx] + dd) % N
        has = counts[target] > 0
        a_idx, target = a_idx[has], target[has]
        tf = 2 * rng.integers(0, counts[target])
        i1 = np.concatenate([i1, cam[a_idx]]); f1 = np.concatenate([f1, feat[a_idx]])
        i2 = np.concatenate([i2, target]); f2 = np.concatenate([f2, tf])
    # group by image pair (image1, ring offset)
    pid = i1 * ring + ((i2 - i1) % N - 1)
    o = np.argsort(pid, kind="stable")
    pid, f1, f2 = pid[o], f1[o], f2[o]
    upid, pstart = np.unique(pid, return_index=True)
    pair_offset = np.append(pstart, len(pid)).astype(np.int64)
    pair_image1 = (upid // ring).astype(np.int32)
    pair_image2 = ((pair_image1 + upid % ring + 1) % N).astype(np.int32)
    return dict(num_images=N, feat_offset=feat_offset, feat_xy=xy, pair_image1=pair_image1, pair_image2=pair_image2,
                pair_valid=np.ones(len(upid), dtype=np.uint8), pair_offset=pair_offset,
                match_feat1=f1.astype(np.uint32), match_feat2=f2.astype(np.uint32),
                gt_num_tracks=n_tracks, gt_num_members=M)


def make_pipeline_scene(n_images=14, n_points=50, seed=0, pixel_noise=0.0, num_succ=6, rot_outlier_pairs=0,
                        false_match_frac=0.0, isolated_pair=False, radius=10.0, ball=2.5, focal=1200.0, layout="inward",
                        shell=(1.0, 3.0)):
    """A whole-pipeline scene in flat form (what `GlobalMapper::Solve` holds after relative-pose estimation,
    global_mapper.cc:85): images on a ring looking at the origin, two shared PINHOLE cameras, 3-D points seen by
    every image that has them in view, per-image feature lists (pixels + undistorted rays), a view graph linking
    each image to its `num_succ` successors with the exact relative rotation and the inlier matches of the shared
    points.  `rot_outlier_pairs` pairs get a random relative rotation (RelPoseFilter::FilterRotations must drop
    them), `false_match_frac` adds matches between unrelated features, `isolated_pair` appends two images that are
    linked only to each other (KeepLargestConnectedComponents must drop them).  Mirrors the role of
    colmap::SynthesizeDataset in global_mapper_test.cc:56-66 (un-vendored; own generator).
    layout="outward": the ring looks OUTWARD at points on a cylindrical shell `shell[0] .. shell[1]` beyond it, so that an image
    sees only the points of its own sector and a point is seen by a run of neighbouring images (the locality of a capture:
    track length ~ n_images * 35 deg-cone / ring, instead of every image seeing every point) — the scale tests of the drop-in."""
    rng = np.random.default_rng(seed)
    N0 = int(n_images)
    centers, R_cw = _ring_cameras(rng, N0, radius, jitter_deg=3.0)
    if layout == "outward":
        R_cw = np.diag([-1.0, 1.0, -1.0]) @ R_cw  # half a turn about the camera's y axis
        ang = rng.uniform(0, 2 * np.pi, n_points)
        rad = radius + rng.uniform(shell[0], shell[1], n_points)
        X = np.stack([rad * np.cos(ang), rng.uniform(-0.5, 0.5, n_points) * shell[1], rad * np.sin(ang)], 1)
    else:
        X = _ball_points(rng, n_points, ball)
    extra = 2 if isolated_pair else 0
    if extra:  # far away, looking elsewhere: they share no point with the ring
        c2 = np.array([[100.0, 0.0, 0.0], [101.0, 0.0, 0.0]])
        centers = np.vstack([centers, c2])
        R_cw = np.concatenate([R_cw, np.tile(np.eye(3), (2, 1, 1))])
    N = N0 + extra
    K = 2
    cam_intr = (np.arange(N) % K).astype(np.int32)
    intr_params = np.zeros((K, 8))
    intr_params[:, :4] = [[focal, focal, 640.0, 480.0], [1.1 * focal, 1.1 * focal, 620.0, 500.0]]
    # visibility: inside a 35 degree half-angle cone and in front
    d = X[None, :, :] - centers[:, None, :]  # [N,P,3]
    xc = np.einsum("nij,npj->npi", R_cw, d)
    vis = (xc[:, :, 2] > 0.1) & (xc[:, :, 2] > np.cos(np.radians(35.0)) * np.linalg.norm(xc, axis=2))
    vis[N0:] = False
    img_of, pt_of = np.nonzero(vis)  # image-major
    feat_count = np.bincount(img_of, minlength=N)
    # the two isolated images get private features so that they can be matched to each other
    priv = 30 if extra else 0
    feat_count[N0:] = priv
    feat_offset = np.zeros(N + 1, dtype=np.int64)
    feat_offset[1:] = np.cumsum(feat_count)
    F = int(feat_offset[-1])
    feat_xy = np.zeros((F, 2))
    feat_undist = np.zeros((F, 3))
    feat_pt = np.full(F, -1, dtype=np.int64)
    fidx = np.arange(len(img_of))  # image-major nonzero(): the ring images' features are laid out in this order
    p = intr_params[cam_intr[img_of]]
    xcv = xc[img_of, pt_of]
    uv = np.stack([p[:, 0] * xcv[:, 0] / xcv[:, 2] + p[:, 2], p[:, 1] * xcv[:, 1] / xcv[:, 2] + p[:, 3]], 1)
    uv += rng.normal(0, pixel_noise, uv.shape) if pixel_noise > 0 else 0.0
    feat_xy[fidx] = uv
    ray = np.stack([(uv[:, 0] - p[:, 2]) / p[:, 0], (uv[:, 1] - p[:, 3]) / p[:, 1], np.ones(len(uv))], 1)
    feat_undist[fidx] = ray / np.linalg.norm(ray, axis=1, keepdims=True)
    feat_pt[fidx] = pt_of
    for n in range(N0, N):
        sl = slice(feat_offset[n], feat_offset[n + 1])
        feat_xy[sl] = rng.uniform(100, 1000, (priv, 2))
        feat_undist[sl] = [0.0, 0.0, 1.0]
    # feature index of point p in image n (or -1)
    feat_of = np.full((N, n_points), -1, dtype=np.int64)
    feat_of[img_of, pt_of] = fidx - feat_offset[img_of]
    # view graph
    pi, pj, pq, poff, f1, f2 = [], [], [], [0], [], []
    pairs = [(i, (i + dlt) % N0) for i in range(N0) for dlt in range(1, num_succ + 1)]
    pairs = sorted({(min(a, b), max(a, b)) for a, b in pairs if a != b})
    if extra:
        pairs.append((N0, N0 + 1))
    bad_pairs = set(rng.choice(len(pairs) - (1 if extra else 0), rot_outlier_pairs, replace=False).tolist()) if rot_outlier_pairs else set()
    for k, (i, j) in enumerate(pairs):
        Rij = R_cw[j] @ R_cw[i].T
        if k in bad_pairs:
            Rij = so3.aa_to_rotmat(rng.normal(0, 1.0, (1, 3)))[0]
        if i >= N0:
            a = np.arange(priv)
            m1, m2 = a, a
        else:
            shared = np.nonzero((feat_of[i] >= 0) & (feat_of[j] >= 0))[0]
            m1, m2 = feat_of[i, shared], feat_of[j, shared]
            nf = int(rng.poisson(false_match_frac * len(shared))) if false_match_frac > 0 else 0
            if nf and feat_count[i] and feat_count[j]:
                m1 = np.concatenate([m1, rng.integers(0, feat_count[i], nf)])
                m2 = np.concatenate([m2, rng.integers(0, feat_count[j], nf)])
        if len(m1) < 5:
            continue
        pi.append(i); pj.append(j); pq.append(so3.rotmat_to_quat(Rij[None])[0])
        f1.append(m1); f2.append(m2)
        poff.append(poff[-1] + len(m1))
    return dict(
        num_images=N, num_ring_images=N0, gt_R=R_cw, gt_center=centers, gt_xyz=X,
        cam_intr=cam_intr, intr_model=np.full(K, 1, dtype=np.int32), intr_params=intr_params,
        feat_offset=feat_offset, feat_xy=feat_xy, feat_undist=feat_undist, feat_point=feat_pt,
        pair_image1=np.array(pi, dtype=np.int32), pair_image2=np.array(pj, dtype=np.int32),
        pair_q=np.array(pq), pair_offset=np.array(poff, dtype=np.int64),
        match_feat1=np.concatenate(f1).astype(np.uint32), match_feat2=np.concatenate(f2).astype(np.uint32),
        pair_rot_outlier=np.array([k in bad_pairs for k in range(len(pairs))])[: len(pi)] if len(pi) == len(pairs) else None,
    )
