"""ORACLE (test infrastructure only — never imported by the product path).

CPU restatement of the reference's SO(3) helpers, written to follow the reference's arithmetic
path (matrix -> quaternion -> angle-axis), not to be fast.

  exp_aa   : glomap/math/rigid3d.cc:45-63  (AngleAxisToRotation, first-order branch for |a| <= EPS,
             EPS = 1e-12 from glomap/types.h:14)
  log_rot  : glomap/math/rigid3d.cc:39-43  (RotationToAngleAxis = Eigen::AngleAxisd(Matrix3d):
             matrix -> quaternion (Shepperd) -> angle = 2 atan2(|v|, |w|), axis = v/|v| * sign(w))

parity unpinned: the reference holds no stored numeric vectors for these.
"""
from __future__ import annotations

import numpy as np

EPS = 1e-12


def exp_aa(a: np.ndarray) -> np.ndarray:
    a = np.asarray(a, dtype=np.float64)
    th = np.linalg.norm(a, axis=-1)
    big = th > EPS
    safe = np.where(big, th, 1.0)
    k = a / safe[..., None]
    s = np.sin(th)
    c = np.cos(th)
    kx, ky, kz = k[..., 0], k[..., 1], k[..., 2]
    R = np.empty(a.shape[:-1] + (3, 3))
    # Eigen AngleAxis::toRotationMatrix
    t = 1.0 - c
    R[..., 0, 0] = t * kx * kx + c
    R[..., 0, 1] = t * kx * ky - s * kz
    R[..., 0, 2] = t * kx * kz + s * ky
    R[..., 1, 0] = t * kx * ky + s * kz
    R[..., 1, 1] = t * ky * ky + c
    R[..., 1, 2] = t * ky * kz - s * kx
    R[..., 2, 0] = t * kx * kz - s * ky
    R[..., 2, 1] = t * ky * kz + s * kx
    R[..., 2, 2] = t * kz * kz + c
    if not np.all(big):
        small = ~big
        ax, ay, az = a[..., 0], a[..., 1], a[..., 2]
        F = np.empty_like(R)
        F[..., 0, 0] = 1
        F[..., 1, 0] = az
        F[..., 2, 0] = -ay
        F[..., 0, 1] = -az
        F[..., 1, 1] = 1
        F[..., 2, 1] = ax
        F[..., 0, 2] = ay
        F[..., 1, 2] = -ax
        F[..., 2, 2] = 1
        R = np.where(small[..., None, None], F, R)
    return R


def rotmat_to_quat_eigen(R: np.ndarray) -> np.ndarray:
    """Eigen's quaternion-from-matrix (w,x,y,z); vectorised Shepperd with Eigen's branch order."""
    R = np.asarray(R, dtype=np.float64)
    m = R.reshape(-1, 3, 3)
    n = m.shape[0]
    q = np.empty((n, 4))
    tr = m[:, 0, 0] + m[:, 1, 1] + m[:, 2, 2]
    pos = tr > 0
    if np.any(pos):
        mp = m[pos]
        t = np.sqrt(tr[pos] + 1.0)
        w = 0.5 * t
        t = 0.5 / t
        q[pos, 0] = w
        q[pos, 1] = (mp[:, 2, 1] - mp[:, 1, 2]) * t
        q[pos, 2] = (mp[:, 0, 2] - mp[:, 2, 0]) * t
        q[pos, 3] = (mp[:, 1, 0] - mp[:, 0, 1]) * t
    neg = np.nonzero(~pos)[0]
    for idx in neg:
        mm = m[idx]
        i = 0
        if mm[1, 1] > mm[0, 0]:
            i = 1
        if mm[2, 2] > mm[i, i]:
            i = 2
        j = (i + 1) % 3
        k = (j + 1) % 3
        t = np.sqrt(mm[i, i] - mm[j, j] - mm[k, k] + 1.0)
        v = [0.0, 0.0, 0.0]
        v[i] = 0.5 * t
        t = 0.5 / t
        q[idx, 0] = (mm[k, j] - mm[j, k]) * t
        v[j] = (mm[j, i] + mm[i, j]) * t
        v[k] = (mm[k, i] + mm[i, k]) * t
        q[idx, 1:] = v
    return q.reshape(R.shape[:-2] + (4,))


def quat_to_aa_eigen(q: np.ndarray) -> np.ndarray:
    """Eigen::AngleAxis(Quaternion): angle in [0, pi], axis sign absorbs w < 0."""
    w = q[..., 0]
    v = q[..., 1:]
    n = np.linalg.norm(v, axis=-1)
    ang = 2.0 * np.arctan2(n, np.abs(w))
    sgn = np.where(w < 0, -1.0, 1.0)
    safe = np.where(n > 0, n, 1.0)
    axis = v / (safe * sgn)[..., None]
    aa = ang[..., None] * axis
    return np.where((n > 0)[..., None], aa, 0.0)


def log_rot(R: np.ndarray) -> np.ndarray:
    return quat_to_aa_eigen(rotmat_to_quat_eigen(R))


def quat_wxyz_to_rotmat(q: np.ndarray) -> np.ndarray:
    q = np.asarray(q, dtype=np.float64)
    w, x, y, z = q[..., 0], q[..., 1], q[..., 2], q[..., 3]
    R = np.empty(q.shape[:-1] + (3, 3))
    tx, ty, tz = 2 * x, 2 * y, 2 * z
    twx, twy, twz = tx * w, ty * w, tz * w
    txx, txy, txz = tx * x, ty * x, tz * x
    tyy, tyz, tzz = ty * y, tz * y, tz * z
    R[..., 0, 0] = 1 - (tyy + tzz)
    R[..., 0, 1] = txy - twz
    R[..., 0, 2] = txz + twy
    R[..., 1, 0] = txy + twz
    R[..., 1, 1] = 1 - (txx + tzz)
    R[..., 1, 2] = tyz - twx
    R[..., 2, 0] = txz - twy
    R[..., 2, 1] = tyz + twx
    R[..., 2, 2] = 1 - (txx + tyy)
    return R
