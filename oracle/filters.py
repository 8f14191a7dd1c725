"""CPU restatement (test infrastructure) of the processors that run between the estimator calls:

  TrackFilter::FilterTracksByReprojection     glomap/processors/track_filter.cc:7-52
  TrackFilter::FilterTracksByAngle            glomap/processors/track_filter.cc:54-90
  TrackFilter::FilterTrackTriangulationAngle  glomap/processors/track_filter.cc:92-127
  NormalizeReconstruction                     glomap/processors/reconstruction_normalizer.cc:5-85
  RelPoseFilter::FilterRotations              glomap/processors/relpose_filter.cc:7-33

Flat arrays as in include/gsfm.h (gsfm_scene_view).  Only tests/, smoke() and bench.py's cpu_baseline leg
may import this module.

PINNED TO REFERENCE CODE (round 5): the three .cc files above compile in this image against the stand-in types of
oracle/ref_shim/ (`make -C oracle ref` -> oracle/_ref/libref_glomap.so, from /root/reference, unmodified); tests/test_oracle_ref.py
holds every function of this file to them — keep masks and counters bit for bit, the normaliser's scale and translation bit
for bit."""
from __future__ import annotations

import numpy as np

from . import ba as oba
from . import so3

EPS = 1e-12  # glomap/types.h:14


def _cam_points(pt_offset, obs_cam, cam_q, cam_t, pt_xyz):
    lens = np.diff(pt_offset)
    pt = np.repeat(np.arange(len(lens)), lens)
    R = so3.quat_wxyz_to_rotmat(np.asarray(cam_q, dtype=np.float64))
    pc = np.einsum("mij,mj->mi", R[obs_cam], pt_xyz[pt]) + cam_t[obs_cam]
    return pt, pc, R


def _changed(pt, keep, P):
    dropped = np.bincount(pt, weights=(~keep).astype(np.float64), minlength=P)
    return int((dropped > 0).sum())


def filter_tracks_by_reprojection(pt_offset, obs_cam, cam_q, cam_t, pt_xyz, max_err=1e-2, in_normalized_image=True,
                                  obs_undist=None, obs_xy=None, cam_intr=None, intr_model=None, intr_params=None):
    """track_filter.cc:7-52.  Returns (keep [M] bool, tracks_changed)."""
    pt, pc, _ = _cam_points(pt_offset, obs_cam, cam_q, cam_t, pt_xyz)
    front = ~(pc[:, 2] < EPS)
    z = np.where(front, pc[:, 2], 1.0)
    if in_normalized_image:
        u = obs_undist
        e = pc[:, :2] / z[:, None] - u[:, :2] / (u[:, 2:3] + EPS)
    else:
        ik = cam_intr[obs_cam]
        uv, _, _, valid = oba.project(intr_model[ik], intr_params[ik], np.where(front[:, None], pc, [0.0, 0.0, 1.0]))
        uv = np.where(valid[:, None], uv, 0.0)  # ImgFromCam(...).value_or(Zero)
        e = uv - obs_xy
    keep = front & (np.sqrt((e * e).sum(1)) < max_err)
    return keep, _changed(pt, keep, len(pt_offset) - 1)


def filter_tracks_by_angle(pt_offset, obs_cam, cam_q, cam_t, pt_xyz, obs_undist, max_angle_deg=1.0, cam_calibrated=None):
    """track_filter.cc:54-90.  Returns (keep [M] bool, tracks_changed)."""
    pt, pc, _ = _cam_points(pt_offset, obs_cam, cam_q, cam_t, pt_xyz)
    front = ~(pc[:, 2] < EPS)
    thr = np.cos(np.radians(max_angle_deg))
    thr_u = np.cos(np.radians(2.0 * max_angle_deg))
    n = pc / np.linalg.norm(pc, axis=1, keepdims=True)
    c = (n * obs_undist).sum(1)
    cal = np.ones(len(cam_q), bool) if cam_calibrated is None else np.asarray(cam_calibrated, bool)
    keep = front & (c > np.where(cal[obs_cam], thr, thr_u))
    return keep, _changed(pt, keep, len(pt_offset) - 1)


def filter_tracks_triangulation_angle(pt_offset, obs_cam, cam_q, cam_t, pt_xyz, min_angle_deg=1.0):
    """track_filter.cc:92-127.  Returns (keep_track [P] bool, tracks_removed)."""
    R = so3.quat_wxyz_to_rotmat(np.asarray(cam_q, dtype=np.float64))
    centers = -np.einsum("nji,nj->ni", R, cam_t)
    thr = np.cos(np.radians(min_angle_deg))
    P = len(pt_offset) - 1
    keep = np.zeros(P, bool)
    for p in range(P):
        k0, k1 = int(pt_offset[p]), int(pt_offset[p + 1])
        if k1 - k0 < 2:
            continue
        d = pt_xyz[p] - centers[obs_cam[k0:k1]]
        d /= np.linalg.norm(d, axis=1, keepdims=True)
        g = d @ d.T
        iu = np.triu_indices(k1 - k0, 1)
        keep[p] = bool((g[iu] < thr).any())
    return keep, int((~keep).sum())


def filter_tracks_triangulation_angle_grouped(pt_offset, obs_cam, cam_q, cam_t, pt_xyz, min_angle_deg=1.0):
    """filter_tracks_triangulation_angle for millions of tracks: the same pairwise test, tracks grouped by length so that
    each group is one batched Gram product (tests/test_filters.py pins it to the per-track loop above)."""
    R = so3.quat_wxyz_to_rotmat(np.asarray(cam_q, dtype=np.float64))
    centers = -np.einsum("nji,nj->ni", R, cam_t)
    thr = np.cos(np.radians(min_angle_deg))
    off = np.asarray(pt_offset, dtype=np.int64)
    lens = np.diff(off)
    P = len(lens)
    keep = np.zeros(P, bool)
    pt = np.repeat(np.arange(P), lens)
    d = pt_xyz[pt] - centers[obs_cam]
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    for L in np.unique(lens):
        if L < 2:
            continue
        trk = np.nonzero(lens == L)[0]
        idx = off[trk][:, None] + np.arange(L)[None, :]
        D = d[idx]  # [n, L, 3]
        for c0 in range(0, len(trk), 1 << 16):  # bounded memory for long tracks
            Dc = D[c0 : c0 + (1 << 16)]
            G = np.einsum("nij,nkj->nik", Dc, Dc)
            iu = np.triu_indices(int(L), 1)
            keep[trk[c0 : c0 + (1 << 16)]] = (G[:, iu[0], iu[1]] < thr).any(axis=1)
    return keep, int((~keep).sum())


def normalize_reconstruction(cam_q, cam_t, pt_xyz, cam_registered=None, fixed_scale=False, extent=10.0, p0=0.1, p1=0.9):
    """reconstruction_normalizer.cc:5-85.  Returns (cam_t', pt_xyz', (scale, translation))."""
    R = so3.quat_wxyz_to_rotmat(np.asarray(cam_q, dtype=np.float64))
    centers = -np.einsum("nji,nj->ni", R, cam_t)
    reg = np.ones(len(cam_q), bool) if cam_registered is None else np.asarray(cam_registered, bool)
    c = np.sort(centers[reg].astype(np.float32), axis=0)  # the reference keeps floats and sorts each axis
    n = c.shape[0]
    P0 = int(p0 * (n - 1)) if n > 3 else 0
    P1 = int(p1 * (n - 1)) if n > 3 else n - 1
    bmin, bmax = c[P0].astype(np.float64), c[P1].astype(np.float64)
    mean = np.zeros(3)
    for i in range(P0, P1 + 1):  # accumulate in double, in order, like the reference
        mean += c[i].astype(np.float64)
    mean /= P1 - P0 + 1
    scale = 1.0
    if not fixed_scale:
        old = np.linalg.norm(bmax - bmin)
        if old >= np.finfo(np.float64).eps:
            scale = extent / old
    t_new = scale * (cam_t + np.einsum("nij,j->ni", R, mean))  # TransformCameraWorld
    X_new = scale * pt_xyz - scale * mean
    return t_new, X_new, (scale, -scale * mean)


def filter_rotations(node_q, edge_i, edge_j, edge_q, max_angle_deg):
    """relpose_filter.cc:7-33 (registered images, valid pairs).  Returns (keep [E] bool, num_invalid)."""
    Rn = so3.quat_wxyz_to_rotmat(np.asarray(node_q, dtype=np.float64))
    Re = so3.quat_wxyz_to_rotmat(np.asarray(edge_q, dtype=np.float64))
    calc = np.einsum("eij,ekj->eik", Rn[edge_j], Rn[edge_i])  # R_j R_i^T
    rel = np.einsum("eji,ejk->eik", calc, Re)
    cos = np.clip((np.trace(rel, axis1=1, axis2=2) - 1.0) / 2.0, -1.0, 1.0)
    ang = np.degrees(np.arccos(cos))
    keep = ~(ang > max_angle_deg)
    return keep, int((~keep).sum())


def undistort_features(feat_xy, feat_intr, intr_model, intr_params, max_iter=100, max_step_sq=1e-10, rel_step=1e-6):
    """UndistortImages, glomap/processors/image_undistorter.cc:7-46: rays [F,3] =
    camera.CamFromImg(xy).value_or(Zero).homogeneous().normalized().

    CamFromImg of a COLMAP camera model (colmap/sensor/models.h — un-vendored; restated as published, PARITY UNPINNED for this
    function) maps the pixel to the normalised plane through the model's (f, c) and, for a model with distortion, inverts
    x + dx(x) = x0 by BaseCameraModel::IterativeUndistortion: Newton with a CENTRAL-DIFFERENCE Jacobian (relative step 1e-6,
    floor machine epsilon), at most 100 iterations, stop when |step|^2 < 1e-10.  Written here on top of oracle.ba.project (the
    forward model ImgFromCam the bundle adjustment is pinned with): x + dx(x) = K^-1 (project((x, 1)) - c).  The product
    (filters.hip k_undistort) runs the same iteration with the analytic Jacobian of its own projection; the two agree to the
    stop's precision and both satisfy project(ray) = pixel (tests/test_filters.py)."""
    xy = np.asarray(feat_xy, dtype=np.float64)
    ik = np.asarray(feat_intr, dtype=np.int64)
    model = np.asarray(intr_model)[ik]
    par = np.asarray(intr_params, dtype=np.float64)[ik]
    F = xy.shape[0]

    def img(x):  # pixel of the ray (x, 1)
        uv, _, _, _ = oba.project(model, par, np.concatenate([x, np.ones((F, 1))], 1))
        return uv

    zero = np.zeros((F, 2))
    c = img(zero)
    h = 1e-6
    fx = (img(zero + [h, 0.0]) - img(zero - [h, 0.0]))[:, 0] / (2 * h)  # the model's focal lengths, read off its own projection
    fy = (img(zero + [0.0, h]) - img(zero - [0.0, h]))[:, 1] / (2 * h)
    K = np.stack([fx, fy], 1)
    x0 = (xy - c) / K
    x = x0.copy()
    active = np.ones(F, dtype=bool)
    for _ in range(max_iter):
        if not active.any():
            break
        step = np.maximum(np.finfo(np.float64).eps, np.abs(rel_step * x))
        d = lambda y: (img(y) - c) / K - y  # dx(y)
        dx = d(x)
        e0, e1 = np.zeros((F, 2)), np.zeros((F, 2))
        e0[:, 0], e1[:, 1] = step[:, 0], step[:, 1]
        d0f, d0b, d1f, d1b = d(x + e0), d(x - e0), d(x + e1), d(x - e1)
        J = np.empty((F, 2, 2))
        J[:, 0, 0] = 1 + (d0f[:, 0] - d0b[:, 0]) / (2 * step[:, 0])
        J[:, 0, 1] = (d1f[:, 0] - d1b[:, 0]) / (2 * step[:, 1])
        J[:, 1, 0] = (d0f[:, 1] - d0b[:, 1]) / (2 * step[:, 0])
        J[:, 1, 1] = 1 + (d1f[:, 1] - d1b[:, 1]) / (2 * step[:, 1])
        rhs = x + dx - x0
        det = J[:, 0, 0] * J[:, 1, 1] - J[:, 0, 1] * J[:, 1, 0]
        sx = np.stack([(J[:, 1, 1] * rhs[:, 0] - J[:, 0, 1] * rhs[:, 1]) / det, (J[:, 0, 0] * rhs[:, 1] - J[:, 1, 0] * rhs[:, 0]) / det], 1)
        sx = np.where(active[:, None], sx, 0.0)
        x = x - sx
        active &= ~((sx * sx).sum(1) < max_step_sq)
    ok = np.isfinite(x).all(1)
    x = np.where(ok[:, None], x, 0.0)
    ray = np.concatenate([x, np.ones((F, 1))], 1)
    return ray / np.linalg.norm(ray, axis=1, keepdims=True)
