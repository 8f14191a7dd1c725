"""Minimal SO(3) helpers (numpy, vectorised) used by the host-side packing code and the
synthetic generators.  Conventions follow the reference:

* quaternions at the C-ABI boundary are (w, x, y, z); Eigen/COLMAP storage order (x, y, z, w)
  is only used by the BundleAdjuster mirror (reference: glomap/estimators/bundle_adjustment.cc:143).
* rotations are cam_from_world:  x_cam = R x_world + t   (reference: docs/rotation_averager.md:46).
"""
from __future__ import annotations

import numpy as np


def quat_to_rotmat(q: np.ndarray) -> np.ndarray:
    """(w,x,y,z) unit quaternions [...,4] -> rotation matrices [...,3,3]."""
    q = np.asarray(q, dtype=np.float64)
    q = q / np.linalg.norm(q, axis=-1, keepdims=True)
    w, x, y, z = q[..., 0], q[..., 1], q[..., 2], q[..., 3]
    R = np.empty(q.shape[:-1] + (3, 3))
    R[..., 0, 0] = 1 - 2 * (y * y + z * z)
    R[..., 0, 1] = 2 * (x * y - w * z)
    R[..., 0, 2] = 2 * (x * z + w * y)
    R[..., 1, 0] = 2 * (x * y + w * z)
    R[..., 1, 1] = 1 - 2 * (x * x + z * z)
    R[..., 1, 2] = 2 * (y * z - w * x)
    R[..., 2, 0] = 2 * (x * z - w * y)
    R[..., 2, 1] = 2 * (y * z + w * x)
    R[..., 2, 2] = 1 - 2 * (x * x + y * y)
    return R


def rotmat_to_quat(R: np.ndarray) -> np.ndarray:
    """Rotation matrices [...,3,3] -> (w,x,y,z) with w >= 0 (Shepperd's method)."""
    R = np.asarray(R, dtype=np.float64)
    flat = R.reshape(-1, 3, 3)
    out = np.empty((flat.shape[0], 4))
    for n, m in enumerate(flat):
        t = m[0, 0] + m[1, 1] + m[2, 2]
        if t > 0:
            s = np.sqrt(t + 1.0)
            w = 0.5 * s
            s = 0.5 / s
            x = (m[2, 1] - m[1, 2]) * s
            y = (m[0, 2] - m[2, 0]) * s
            z = (m[1, 0] - m[0, 1]) * s
        else:
            i = int(np.argmax([m[0, 0], m[1, 1], m[2, 2]]))
            j = (i + 1) % 3
            k = (j + 1) % 3
            s = np.sqrt(m[i, i] - m[j, j] - m[k, k] + 1.0)
            v = [0.0, 0.0, 0.0]
            v[i] = 0.5 * s
            s = 0.5 / s
            w = (m[k, j] - m[j, k]) * s
            v[j] = (m[j, i] + m[i, j]) * s
            v[k] = (m[k, i] + m[i, k]) * s
            x, y, z = v
        q = np.array([w, x, y, z])
        if q[0] < 0:
            q = -q
        out[n] = q / np.linalg.norm(q)
    return out.reshape(R.shape[:-2] + (4,))


def aa_to_rotmat(a: np.ndarray) -> np.ndarray:
    """Angle-axis [...,3] -> rotation matrices (Rodrigues)."""
    a = np.asarray(a, dtype=np.float64)
    th = np.linalg.norm(a, axis=-1)
    safe = np.where(th > 1e-12, th, 1.0)
    k = a / safe[..., None]
    K = np.zeros(a.shape[:-1] + (3, 3))
    K[..., 0, 1] = -k[..., 2]
    K[..., 0, 2] = k[..., 1]
    K[..., 1, 0] = k[..., 2]
    K[..., 1, 2] = -k[..., 0]
    K[..., 2, 0] = -k[..., 1]
    K[..., 2, 1] = k[..., 0]
    s = np.sin(th)[..., None, None]
    c = np.cos(th)[..., None, None]
    I = np.broadcast_to(np.eye(3), K.shape)
    R = I + s * K + (1 - c) * (K @ K)
    return R


def aa_to_quat(a: np.ndarray) -> np.ndarray:
    a = np.asarray(a, dtype=np.float64)
    th = np.linalg.norm(a, axis=-1, keepdims=True)
    half = 0.5 * th
    # sin(th/2)/th with the small-angle limit 1/2
    k = np.where(th > 1e-12, np.sin(half) / np.where(th > 1e-12, th, 1.0), 0.5)
    return np.concatenate([np.cos(half), k * a], axis=-1)


def quat_to_aa(q: np.ndarray) -> np.ndarray:
    q = np.asarray(q, dtype=np.float64)
    q = q / np.linalg.norm(q, axis=-1, keepdims=True)
    w = q[..., :1]
    v = q[..., 1:]
    n = np.linalg.norm(v, axis=-1, keepdims=True)
    ang = 2.0 * np.arctan2(n, np.abs(w))
    sgn = np.where(w < 0, -1.0, 1.0)
    safe = np.where(n > 0, n, 1.0)
    return np.where(n > 0, sgn * ang * v / safe, 0.0)


def quat_mul(a: np.ndarray, b: np.ndarray) -> np.ndarray:
    aw, ax, ay, az = a[..., 0], a[..., 1], a[..., 2], a[..., 3]
    bw, bx, by, bz = b[..., 0], b[..., 1], b[..., 2], b[..., 3]
    return np.stack(
        [
            aw * bw - ax * bx - ay * by - az * bz,
            aw * bx + ax * bw + ay * bz - az * by,
            aw * by - ax * bz + ay * bw + az * bx,
            aw * bz + ax * by - ay * bx + az * bw,
        ],
        axis=-1,
    )


def quat_conj(q: np.ndarray) -> np.ndarray:
    return q * np.array([1.0, -1.0, -1.0, -1.0])


def rotation_angle_deg(Ra: np.ndarray, Rb: np.ndarray) -> np.ndarray:
    """Angle (degrees) between rotation matrices, as CalcAngle (reference: glomap/math/rigid3d.cc:22-27)."""
    tr = np.einsum("...ij,...ij->...", Ra, Rb)
    c = np.clip((tr - 1.0) / 2.0, -1.0, 1.0)
    return np.degrees(np.arccos(c))
