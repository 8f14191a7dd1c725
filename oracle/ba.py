"""ORACLE (test infrastructure only — never imported by the product path).

CPU restatement of GLOMAP's global bundle adjustment for trivial rigs:

  problem build   BundleAdjuster::AddPointToCameraConstraints      ba.cc:115-190  (tracks >= min views, ba.cc:122)
  residual        colmap::ReprojErrorCostFunctor<CameraModel>       via ba.cc:137-146 — COLMAP @ b6b7b54e is
                  un-vendored; restated from its published definition:
                  x_c = R(q) X + t;  (u,v) = CameraModel::ImgFromCam(params, x_c);  r = (u,v) - obs   [pixels]
                  (residual and Jacobian zero when the point is not in front of the camera)
  camera models   SIMPLE_PINHOLE, PINHOLE, SIMPLE_RADIAL, RADIAL, OPENCV, OPENCV_FISHEYE, FOV, SIMPLE_RADIAL_FISHEYE,
                  RADIAL_FISHEYE (colmap/sensor/models.h; every COLMAP model with at most 8 parameters) in [K, 8] blocks, and
                  FULL_OPENCV, THIN_PRISM_FISHEYE (12 parameters), RAD_TAN_THIN_PRISM_FISHEYE (16) in [K, 16] blocks — the
                  width of an intrinsics block is the second dimension of the intr_params array
  parameterisation ba.cc:244-317: EigenQuaternionManifold (q <- [sin|d| d/|d|, cos|d|] * q), first
                  frame constant, optimize_rotations / optimize_translation flags, principal point
                  frozen by a SubsetManifold unless optimize_principal_point
  loss            Huber(1 px)                                       bundle_adjustment.h:30,34-36
  solver          Ceres LM (oracle/lm.py), points eliminated first (ba.cc:204-208)

PROBLEM BUILDER PINNED TO REFERENCE CODE (round 5): bundle_adjustment.cc compiles, unmodified, against the recording Ceres and
stand-ins for COLMAP's cost-function factory / manifold helpers (oracle/_ref/libref_glomap_ba.so); tests/test_oracle_ref_ba.py
holds build_problem() — residual blocks and their functor, constant blocks, principal-point subset, manifolds, elimination
order, initial cost, trivial frames and both rig modes — to BundleAdjuster::Solve as the reference wrote it.  The projection
functions (un-vendored COLMAP) and the minimiser (oracle/lm.py, Ceres' trust-region loop) are restatements: parity unpinned for
those parts, compared through converged solutions.
"""
from __future__ import annotations

from dataclasses import dataclass, field

import numpy as np
import scipy.sparse as sp

from . import lm

SIMPLE_PINHOLE, PINHOLE, SIMPLE_RADIAL, RADIAL, OPENCV = 0, 1, 2, 3, 4
OPENCV_FISHEYE, FOV, SIMPLE_RADIAL_FISHEYE, RADIAL_FISHEYE = 5, 7, 8, 9  # COLMAP's CameraModelId values
FULL_OPENCV, THIN_PRISM_FISHEYE, RAD_TAN_THIN_PRISM_FISHEYE = 6, 10, 11    # more than 8 parameters: [K, 16] blocks
NUM_PARAMS = {0: 3, 1: 4, 2: 4, 3: 5, 4: 8, 5: 8, 7: 5, 8: 4, 9: 5, 6: 12, 10: 12, 11: 16}
PP_IDXS = {0: (1, 2), 1: (2, 3), 2: (1, 2), 3: (1, 2), 4: (2, 3), 5: (2, 3), 7: (2, 3), 8: (1, 2), 9: (1, 2),
           6: (2, 3), 10: (2, 3), 11: (2, 3)}
MAXP = 8        # width of an intrinsics block unless a model needs more
MAXP_WIDE = 16  # FULL_OPENCV, THIN_PRISM_FISHEYE, RAD_TAN_THIN_PRISM_FISHEYE


@dataclass
class BundleAdjusterOptions:
    """bundle_adjustment.h:12-37 (+ optimization_base.h:18-23)."""

    optimize_rotations: bool = True
    optimize_translation: bool = True
    optimize_intrinsics: bool = True
    optimize_principal_point: bool = False
    optimize_points: bool = True
    optimize_rig_poses: bool = False  # bundle_adjustment.h:15
    min_num_view_per_track: int = 3
    thres_loss_function: float = 1.0
    lm: lm.LmOptions = field(default_factory=lambda: lm.LmOptions(max_num_iterations=200))


def quat_to_rot(q):
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


def quat_mul(a, b):
    aw, ax, ay, az = a[..., 0], a[..., 1], a[..., 2], a[..., 3]
    bw, bx, by, bz = b[..., 0], b[..., 1], b[..., 2], b[..., 3]
    return np.stack([aw * bw - ax * bx - ay * by - az * bz, aw * bx + ax * bw + ay * bz - az * by,
                     aw * by - ax * bz + ay * bw + az * bx, aw * bz + ax * by - ay * bx + az * bw], axis=-1)


def project(model, params, xc):
    """ImgFromCam for per-row model ids; params [m, W], W = 8 or 16.  Returns (uv [m,2], J_xc [m,2,3], J_par [m,2,W], valid [m])."""
    m = xc.shape[0]
    x, y, z = xc[:, 0], xc[:, 1], xc[:, 2]
    valid = z > np.finfo(np.float64).eps
    zs = np.where(valid, z, 1.0)
    u, v = x / zs, y / zs
    r2 = u * u + v * v
    uv = np.zeros((m, 2))
    Juv = np.zeros((m, 2, 2))  # d(pixel) / d(u, v)
    Jp = np.zeros((m, 2, params.shape[1]))
    p = params
    for mid in np.unique(model):
        s = model == mid
        us, vs, r2s = u[s], v[s], r2[s]
        ps = p[s]
        if mid == SIMPLE_PINHOLE:
            f, cx, cy = ps[:, 0], ps[:, 1], ps[:, 2]
            uv[s] = np.stack([f * us + cx, f * vs + cy], 1)
            Juv[s, 0, 0] = f
            Juv[s, 1, 1] = f
            Jp[s, 0, 0], Jp[s, 1, 0] = us, vs
            Jp[s, 0, 1] = 1
            Jp[s, 1, 2] = 1
        elif mid == PINHOLE:
            fx, fy, cx, cy = ps[:, 0], ps[:, 1], ps[:, 2], ps[:, 3]
            uv[s] = np.stack([fx * us + cx, fy * vs + cy], 1)
            Juv[s, 0, 0] = fx
            Juv[s, 1, 1] = fy
            Jp[s, 0, 0] = us
            Jp[s, 1, 1] = vs
            Jp[s, 0, 2] = 1
            Jp[s, 1, 3] = 1
        elif mid in (SIMPLE_RADIAL, RADIAL):
            f, cx, cy, k1 = ps[:, 0], ps[:, 1], ps[:, 2], ps[:, 3]
            k2 = ps[:, 4] if mid == RADIAL else np.zeros_like(k1)
            rad = k1 * r2s + k2 * r2s * r2s
            drad = k1 + 2 * k2 * r2s  # d rad / d r2
            ud, vd = us * (1 + rad), vs * (1 + rad)
            uv[s] = np.stack([f * ud + cx, f * vd + cy], 1)
            Juv[s, 0, 0] = f * (1 + rad + 2 * us * us * drad)
            Juv[s, 0, 1] = f * (2 * us * vs * drad)
            Juv[s, 1, 0] = f * (2 * us * vs * drad)
            Juv[s, 1, 1] = f * (1 + rad + 2 * vs * vs * drad)
            Jp[s, 0, 0], Jp[s, 1, 0] = ud, vd
            Jp[s, 0, 1] = 1
            Jp[s, 1, 2] = 1
            Jp[s, 0, 3], Jp[s, 1, 3] = f * us * r2s, f * vs * r2s
            if mid == RADIAL:
                Jp[s, 0, 4], Jp[s, 1, 4] = f * us * r2s * r2s, f * vs * r2s * r2s
        elif mid == OPENCV:
            fx, fy, cx, cy, k1, k2, p1, p2 = (ps[:, i] for i in range(8))
            rad = k1 * r2s + k2 * r2s * r2s
            drad = k1 + 2 * k2 * r2s
            du = us * rad + 2 * p1 * us * vs + p2 * (r2s + 2 * us * us)
            dv = vs * rad + 2 * p2 * us * vs + p1 * (r2s + 2 * vs * vs)
            uv[s] = np.stack([fx * (us + du) + cx, fy * (vs + dv) + cy], 1)
            ddu_du = rad + 2 * us * us * drad + 2 * p1 * vs + 6 * p2 * us
            ddu_dv = 2 * us * vs * drad + 2 * p1 * us + 2 * p2 * vs
            ddv_du = 2 * us * vs * drad + 2 * p2 * vs + 2 * p1 * us
            ddv_dv = rad + 2 * vs * vs * drad + 2 * p2 * us + 6 * p1 * vs
            Juv[s, 0, 0] = fx * (1 + ddu_du)
            Juv[s, 0, 1] = fx * ddu_dv
            Juv[s, 1, 0] = fy * ddv_du
            Juv[s, 1, 1] = fy * (1 + ddv_dv)
            Jp[s, 0, 0] = us + du
            Jp[s, 1, 1] = vs + dv
            Jp[s, 0, 2] = 1
            Jp[s, 1, 3] = 1
            Jp[s, 0, 4], Jp[s, 1, 4] = fx * us * r2s, fy * vs * r2s
            Jp[s, 0, 5], Jp[s, 1, 5] = fx * us * r2s * r2s, fy * vs * r2s * r2s
            Jp[s, 0, 6], Jp[s, 1, 6] = fx * 2 * us * vs, fy * (r2s + 2 * vs * vs)
            Jp[s, 0, 7], Jp[s, 1, 7] = fx * (r2s + 2 * us * us), fy * 2 * us * vs
        elif mid in (OPENCV_FISHEYE, SIMPLE_RADIAL_FISHEYE, RADIAL_FISHEYE):
            # equidistant fisheye: theta = atan(r), pixel = f (u, v) theta_d / r + c, theta_d = theta (1 + sum k_j theta^(2j+2))
            full = mid == OPENCV_FISHEYE
            fx = ps[:, 0]
            fy = ps[:, 1] if full else ps[:, 0]
            ic, ik0 = (2, 4) if full else (1, 3)
            nk = 4 if full else (2 if mid == RADIAL_FISHEYE else 1)
            rs = np.sqrt(r2s)
            big = rs > np.finfo(np.float64).eps
            th = np.where(big, np.arctan(rs), rs)
            th2 = th * th
            poly = np.ones_like(th)
            dpoly = np.ones_like(th)
            tpow = []
            tp = np.ones_like(th)
            for j in range(nk):
                tp = tp * th2
                tpow.append(tp)
                poly = poly + ps[:, ik0 + j] * tp
                dpoly = dpoly + (2 * j + 3) * ps[:, ik0 + j] * tp
            rsafe = np.where(big, rs, 1.0)
            mfac = np.where(big, th * poly / rsafe, poly)
            dm_r = np.where(big, (dpoly / (1 + r2s) - mfac) / np.where(big, r2s, 1.0), 0.0)
            uv[s] = np.stack([fx * us * mfac + ps[:, ic], fy * vs * mfac + ps[:, ic + 1]], 1)
            Juv[s, 0, 0] = fx * (mfac + us * us * dm_r)
            Juv[s, 0, 1] = fx * us * vs * dm_r
            Juv[s, 1, 0] = fy * us * vs * dm_r
            Juv[s, 1, 1] = fy * (mfac + vs * vs * dm_r)
            if full:
                Jp[s, 0, 0] = us * mfac
                Jp[s, 1, 1] = vs * mfac
            else:
                Jp[s, 0, 0], Jp[s, 1, 0] = us * mfac, vs * mfac
            Jp[s, 0, ic] = 1
            Jp[s, 1, ic + 1] = 1
            sfac = np.where(big, th / rsafe, 1.0)
            for j in range(nk):
                Jp[s, 0, ik0 + j] = fx * us * sfac * tpow[j]
                Jp[s, 1, ik0 + j] = fy * vs * sfac * tpow[j]
        elif mid == FOV:
            fx, fy, cx, cy, om = (ps[:, i] for i in range(5))
            om2 = om * om
            fac = np.empty_like(om)
            dfac_r2 = np.empty_like(om)
            dfac_om = np.empty_like(om)
            a_ = om2 < 1e-4                       # COLMAP's series branches (FOVCameraModel::Distortion)
            b_ = ~a_ & (r2s < 1e-4)
            c_ = ~a_ & ~b_
            fac[a_] = om2[a_] * r2s[a_] / 3 - om2[a_] / 12 + 1
            dfac_r2[a_] = om2[a_] / 3
            dfac_om[a_] = 2 * om[a_] * (r2s[a_] / 3 - 1 / 12)
            t = np.tan(0.5 * om)
            fac[b_] = (-2 * t[b_] * (4 * r2s[b_] * t[b_] ** 2 - 3)) / (3 * om[b_])
            dfac_r2[b_] = -8 * t[b_] ** 3 / (3 * om[b_])
            dfac_om[b_] = (-2 * 0.5 * (1 + t[b_] ** 2) * (12 * r2s[b_] * t[b_] ** 2 - 3)) / (3 * om[b_]) - fac[b_] / om[b_]
            rc_ = np.sqrt(r2s[c_])
            ac = 2 * rc_ * t[c_]
            num = np.arctan(ac)
            fac[c_] = num / (rc_ * om[c_])
            da = 1 / (1 + ac * ac)
            dfac_r2[c_] = (da * 2 * t[c_] * rc_ - num) / (r2s[c_] * om[c_]) / (2 * rc_)
            dfac_om[c_] = da * 2 * rc_ * 0.5 * (1 + t[c_] ** 2) / (rc_ * om[c_]) - fac[c_] / om[c_]
            uv[s] = np.stack([fx * us * fac + cx, fy * vs * fac + cy], 1)
            Juv[s, 0, 0] = fx * (fac + 2 * us * us * dfac_r2)
            Juv[s, 0, 1] = fx * 2 * us * vs * dfac_r2
            Juv[s, 1, 0] = fy * 2 * us * vs * dfac_r2
            Juv[s, 1, 1] = fy * (fac + 2 * vs * vs * dfac_r2)
            Jp[s, 0, 0] = us * fac
            Jp[s, 1, 1] = vs * fac
            Jp[s, 0, 2] = 1
            Jp[s, 1, 3] = 1
            Jp[s, 0, 4], Jp[s, 1, 4] = fx * us * dfac_om, fy * vs * dfac_om
        elif mid in (FULL_OPENCV, THIN_PRISM_FISHEYE, RAD_TAN_THIN_PRISM_FISHEYE):
            # (a, b) -> (ad, bd) distortions of (u, v) itself (FULL_OPENCV) or of the equidistant coordinates
            # (a, b) = (u, v) theta / r, theta = atan(r) (the thin-prism fisheye models); E = d(a, b) / d(u, v)
            assert ps.shape[1] >= MAXP_WIDE, "camera models with more than 8 parameters need [K, 16] intrinsics blocks"
            fx, fy, cx, cy = ps[:, 0], ps[:, 1], ps[:, 2], ps[:, 3]
            n = us.shape[0]
            a, b = us.copy(), vs.copy()
            E = np.zeros((n, 2, 2))
            E[:, 0, 0] = E[:, 1, 1] = 1.0
            if mid != FULL_OPENCV:
                rs = np.sqrt(r2s)
                big = rs > np.finfo(np.float64).eps
                rsafe = np.where(big, rs, 1.0)
                sfac = np.where(big, np.arctan(rsafe) / rsafe, 1.0)
                sp_ = np.where(big, (1 / (1 + r2s) - sfac) / np.where(big, r2s, 1.0), 0.0)  # s' / r
                a, b = sfac * us, sfac * vs
                E[:, 0, 0] = sfac + us * us * sp_
                E[:, 0, 1] = E[:, 1, 0] = us * vs * sp_
                E[:, 1, 1] = sfac + vs * vs * sp_
            a2, b2, ab = a * a, b * b, a * b
            q2 = a2 + b2
            D = np.zeros((n, 2, 2))  # d(ad, bd) / d(a, b)
            dpar = {}                # parameter index -> (d ad, d bd)
            zero = np.zeros(n)
            if mid == FULL_OPENCV:
                k1, k2, p1, p2, k3, k4, k5, k6 = (ps[:, i] for i in range(4, 12))
                q4, q6 = q2 * q2, q2 * q2 * q2
                num = 1 + k1 * q2 + k2 * q4 + k3 * q6
                den = 1 + k4 * q2 + k5 * q4 + k6 * q6
                rad = num / den
                drad = ((k1 + 2 * k2 * q2 + 3 * k3 * q4) - rad * (k4 + 2 * k5 * q2 + 3 * k6 * q4)) / den
                ad = a * rad + 2 * p1 * ab + p2 * (q2 + 2 * a2)
                bd = b * rad + 2 * p2 * ab + p1 * (q2 + 2 * b2)
                D[:, 0, 0] = rad + 2 * a2 * drad + 2 * p1 * b + 6 * p2 * a
                D[:, 0, 1] = 2 * ab * drad + 2 * p1 * a + 2 * p2 * b
                D[:, 1, 0] = 2 * ab * drad + 2 * p2 * b + 2 * p1 * a
                D[:, 1, 1] = rad + 2 * b2 * drad + 2 * p2 * a + 6 * p1 * b
                for idx, qq in ((4, q2), (5, q4), (8, q6)):
                    dpar[idx] = (a * qq / den, b * qq / den)
                for idx, qq in ((9, q2), (10, q4), (11, q6)):
                    dpar[idx] = (-a * rad * qq / den, -b * rad * qq / den)
                dpar[6] = (2 * ab, q2 + 2 * b2)
                dpar[7] = (q2 + 2 * a2, 2 * ab)
            elif mid == THIN_PRISM_FISHEYE:
                k1, k2, p1, p2, k3, k4, sx1, sy1 = (ps[:, i] for i in range(4, 12))
                q4 = q2 * q2
                q6, q8 = q4 * q2, q4 * q4
                rad = k1 * q2 + k2 * q4 + k3 * q6 + k4 * q8
                drad = k1 + 2 * k2 * q2 + 3 * k3 * q4 + 4 * k4 * q6
                ad = a + a * rad + 2 * p1 * ab + p2 * (q2 + 2 * a2) + sx1 * q2
                bd = b + b * rad + 2 * p2 * ab + p1 * (q2 + 2 * b2) + sy1 * q2
                D[:, 0, 0] = 1 + rad + 2 * a2 * drad + 2 * p1 * b + 6 * p2 * a + 2 * sx1 * a
                D[:, 0, 1] = 2 * ab * drad + 2 * p1 * a + 2 * p2 * b + 2 * sx1 * b
                D[:, 1, 0] = 2 * ab * drad + 2 * p2 * b + 2 * p1 * a + 2 * sy1 * a
                D[:, 1, 1] = 1 + rad + 2 * b2 * drad + 2 * p2 * a + 6 * p1 * b + 2 * sy1 * b
                for idx, qq in ((4, q2), (5, q4), (8, q6), (9, q8)):
                    dpar[idx] = (a * qq, b * qq)
                dpar[6] = (2 * ab, q2 + 2 * b2)
                dpar[7] = (q2 + 2 * a2, 2 * ab)
                dpar[10] = (q2, zero)
                dpar[11] = (zero, q2)
            else:
                # radial stage (uh, vh) = (a, b) (1 + k0 q2 + ... + k5 q2^6); tangential (p0, p1) and thin prism (s0 .. s3)
                # on the radially distorted coordinates
                kk = [ps[:, 4 + i] for i in range(6)]
                p0, p1, s0, s1, s2, s3 = (ps[:, i] for i in range(10, 16))
                qp = [q2]
                for i in range(1, 6):
                    qp.append(qp[-1] * q2)
                Rr = 1 + sum(kk[i] * qp[i] for i in range(6))
                dR = kk[0] + sum((i + 1) * kk[i] * qp[i - 1] for i in range(1, 6))
                uh, vh = Rr * a, Rr * b
                uh2, vh2, uhvh = uh * uh, vh * vh, uh * vh
                h2 = uh2 + vh2
                h4 = h2 * h2
                ad = uh + p0 * (2 * uh2 + h2) + 2 * p1 * uhvh + s0 * h2 + s1 * h4
                bd = vh + p1 * (2 * vh2 + h2) + 2 * p0 * uhvh + s2 * h2 + s3 * h4
                sx, sy = s0 + 2 * s1 * h2, s2 + 2 * s3 * h2
                T = np.zeros((n, 2, 2))
                T[:, 0, 0] = 1 + 6 * p0 * uh + 2 * p1 * vh + 2 * uh * sx
                T[:, 0, 1] = 2 * p0 * vh + 2 * p1 * uh + 2 * vh * sx
                T[:, 1, 0] = 2 * p1 * uh + 2 * p0 * vh + 2 * uh * sy
                T[:, 1, 1] = 1 + 6 * p1 * vh + 2 * p0 * uh + 2 * vh * sy
                Rj = np.zeros((n, 2, 2))
                Rj[:, 0, 0] = Rr + 2 * a2 * dR
                Rj[:, 0, 1] = Rj[:, 1, 0] = 2 * ab * dR
                Rj[:, 1, 1] = Rr + 2 * b2 * dR
                D = T @ Rj
                ta = T[:, 0, 0] * a + T[:, 0, 1] * b
                tb = T[:, 1, 0] * a + T[:, 1, 1] * b
                for i in range(6):
                    dpar[4 + i] = (ta * qp[i], tb * qp[i])
                dpar[10] = (2 * uh2 + h2, 2 * uhvh)
                dpar[11] = (2 * uhvh, 2 * vh2 + h2)
                dpar[12] = (h2, zero)
                dpar[13] = (h4, zero)
                dpar[14] = (zero, h2)
                dpar[15] = (zero, h4)
            uv[s] = np.stack([fx * ad + cx, fy * bd + cy], 1)
            DE = D @ E
            Juv[s, 0, :] = fx[:, None] * DE[:, 0, :]
            Juv[s, 1, :] = fy[:, None] * DE[:, 1, :]
            Jp[s, 0, 0] = ad
            Jp[s, 1, 1] = bd
            Jp[s, 0, 2] = 1
            Jp[s, 1, 3] = 1
            for idx, (da_, db_) in dpar.items():
                Jp[s, 0, idx], Jp[s, 1, idx] = fx * da_, fy * db_
        else:
            raise ValueError(f"camera model {mid} not supported")
    # d(u,v)/d x_c
    Jn = np.zeros((m, 2, 3))
    Jn[:, 0, 0] = 1 / zs
    Jn[:, 0, 2] = -u / zs
    Jn[:, 1, 1] = 1 / zs
    Jn[:, 1, 2] = -v / zs
    Jx = Juv @ Jn
    return uv, Jx, Jp, valid


def free_param_mask(model, opt: BundleAdjusterOptions, width: int = MAXP):
    """Which entries of each intrinsics block are optimised (ba.cc:273-293)."""
    K = model.shape[0]
    mask = np.zeros((K, width), dtype=bool)
    for k in range(K):
        n = NUM_PARAMS[int(model[k])]
        if not opt.optimize_intrinsics and not opt.optimize_principal_point:
            continue  # SetParameterBlockConstant
        mask[k, :n] = True
        if opt.optimize_intrinsics and not opt.optimize_principal_point:
            mask[k, list(PP_IDXS[int(model[k])])] = False  # SubsetManifold on the principal point
    return mask


class _BaProblem:
    def __init__(self, N, cam, pt, xy, cam_intr, model, fixed_cam, P, opt, obs_ik=None, Rs=None, ts=None, obs_sens=None,
                 num_sensors=0, width=MAXP):
        self.W = int(width)  # doubles per intrinsics block (8, or 16 with a 12 / 16-parameter model)
        # N counts ALL pose blocks: the frames and, behind them, the `num_sensors` optimised cam_from_rig blocks
        self.N, self.P, self.M = N, P, cam.shape[0]
        self.cam, self.pt, self.xy = cam, pt, xy
        self.cam_intr, self.model = cam_intr, model
        # known rigs: RigReprojErrorConstantRigCostFunctor (ba.cc:147-160): x_c = cam_from_rig * (rig_from_world * X)
        # with a CONSTANT cam_from_rig per observation (Rs, ts) and the image's own intrinsics block (obs_ik)
        self.obs_ik, self.Rs, self.ts = obs_ik, Rs, ts
        # optimize_rig_poses: RigReprojErrorCostFunctor (ba.cc:161-179): the cam_from_rig of observation m is the pose
        # block obs_sens[m] (an index into the pose blocks, -1 = the constant of the tables above)
        self.obs_sens = obs_sens
        self.has_sens = None if obs_sens is None else obs_sens >= 0
        self.K = model.shape[0]
        self.opt = opt
        self.loss = lm.HuberLoss(opt.thres_loss_function)
        self.fmask = free_param_mask(model, opt, self.W)
        # column layout: [6 per pose | free intrinsics | 3 per point]
        self.intr_col = -np.ones((self.K, self.W), dtype=np.int64)
        nfree = int(self.fmask.sum())
        self.intr_col[self.fmask] = 6 * N + np.arange(nfree)
        self.pt_col0 = 6 * N + nfree
        self.n = self.pt_col0 + 3 * P
        self.rot_free = np.full(N, bool(opt.optimize_rotations))
        self.trn_free = np.full(N, bool(opt.optimize_translation))
        if fixed_cam >= 0:
            self.rot_free[fixed_cam] = False  # ba.cc:261-266
            self.trn_free[fixed_cam] = False
        if num_sensors:  # ba.cc:296-309: the cam_from_rig blocks only get their manifold, never a constant flag
            self.rot_free[N - num_sensors:] = True
            self.trn_free[N - num_sensors:] = True
        self.elimination = [(self.pt_col0, P, 3)] if opt.optimize_points else []

    def unpack(self, x):
        N, P, K = self.N, self.P, self.K
        o = 0
        q = x[o : o + 4 * N].reshape(N, 4); o += 4 * N
        t = x[o : o + 3 * N].reshape(N, 3); o += 3 * N
        X = x[o : o + 3 * P].reshape(P, 3); o += 3 * P
        intr = x[o : o + self.W * K].reshape(K, self.W)
        return q, t, X, intr

    @staticmethod
    def pack(q, t, X, intr):
        return np.concatenate([q.ravel(), t.ravel(), X.ravel(), intr.ravel()])

    def _geom(self, x):
        q, t, X, intr = self.unpack(x)
        R = quat_to_rot(q)
        RX = np.einsum("mij,mj->mi", R[self.cam], X[self.pt])
        xc = RX + t[self.cam]
        ik = self.cam_intr[self.cam] if self.obs_ik is None else self.obs_ik
        Rs, ts = self.Rs, self.ts
        if self.obs_sens is not None:
            h = self.has_sens
            Rs, ts = Rs.copy(), ts.copy()
            Rs[h], ts[h] = R[self.obs_sens[h]], t[self.obs_sens[h]]
        self._Jcam = self._a_rig = None
        if Rs is not None:
            a_rig = np.einsum("mij,mj->mi", Rs, xc)  # R_s x_rig: what the sensor's rotation acts on
            xc = a_rig + ts
        uv, Jx, Jp, valid = project(self.model[ik], intr[ik], xc)
        if Rs is not None:
            self._Jcam, self._a_rig = Jx, a_rig
            Jx = Jx @ Rs  # d(uv)/d(x_rig): everything downstream differentiates through the rig-frame point
        r = np.where(valid[:, None], uv - self.xy, 0.0)
        return R, RX, ik, r, Jx, Jp, valid

    def cost(self, x):
        r = self._geom(x)[3]
        rho0, _ = self.loss.evaluate((r * r).sum(1))
        return 0.5 * float(rho0.sum())

    def evaluate(self, x):
        N, M = self.N, self.M
        R, RX, ik, r, Jx, Jp, valid = self._geom(x)
        rho0, rho1 = self.loss.evaluate((r * r).sum(1))
        sw = np.sqrt(rho1) * valid
        Jx = Jx * sw[:, None, None]
        Jp = Jp * sw[:, None, None]
        # d x_c / d delta_rot = -2 [R X]_x  (EigenQuaternionManifold: rotation by 2|delta| on the left)
        a = RX
        skew = np.zeros((M, 3, 3))
        skew[:, 0, 1], skew[:, 0, 2] = -a[:, 2], a[:, 1]
        skew[:, 1, 0], skew[:, 1, 2] = a[:, 2], -a[:, 0]
        skew[:, 2, 0], skew[:, 2, 1] = -a[:, 1], a[:, 0]
        Jrot = -2.0 * (Jx @ skew) * self.rot_free[self.cam][:, None, None]
        Jtrn = Jx * self.trn_free[self.cam][:, None, None]
        Jpt = (Jx @ R[self.cam]) * (1.0 if self.opt.optimize_points else 0.0)
        rows = np.arange(2 * M).reshape(M, 2)
        ri, ci, vi = [], [], []

        def add(block, col0):  # block [M,2,w], col0 [M]
            w = block.shape[2]
            ri.append(np.repeat(rows[:, :, None], w, axis=2).ravel())
            ci.append(np.broadcast_to((col0[:, None] + np.arange(w))[:, None, :], (M, 2, w)).ravel())
            vi.append(block.ravel())

        add(Jrot, 6 * self.cam)
        add(Jtrn, 6 * self.cam + 3)
        if self.obs_sens is not None and self.has_sens.any():
            # x_c = Exp(2 d_rot) (R_s x_rig) + t_s + d_trn for the sensor block
            h = self.has_sens
            Jc = self._Jcam[h] * sw[h, None, None]
            b = self._a_rig[h]
            sk = np.zeros((b.shape[0], 3, 3))
            sk[:, 0, 1], sk[:, 0, 2] = -b[:, 2], b[:, 1]
            sk[:, 1, 0], sk[:, 1, 2] = b[:, 2], -b[:, 0]
            sk[:, 2, 0], sk[:, 2, 1] = -b[:, 1], b[:, 0]
            w_ = np.concatenate([-2.0 * (Jc @ sk), Jc], axis=2)  # [m,2,6]
            rr = rows[h]
            ri.append(np.repeat(rr[:, :, None], 6, axis=2).ravel())
            ci.append(np.broadcast_to((6 * self.obs_sens[h][:, None] + np.arange(6))[:, None, :], (rr.shape[0], 2, 6)).ravel())
            vi.append(w_.ravel())
        add(Jpt, self.pt_col0 + 3 * self.pt)
        cols = self.intr_col[ik]  # [M,8], -1 where constant
        for j in range(self.W):
            sel = cols[:, j] >= 0
            if not sel.any():
                continue
            ri.append(rows[sel].ravel())
            ci.append(np.repeat(cols[sel, j], 2))
            vi.append(Jp[sel, :, j].ravel())
        J = sp.csr_matrix((np.concatenate(vi), (np.concatenate(ri), np.concatenate(ci))), shape=(2 * M, self.n))
        return 0.5 * float(rho0.sum()), (sw[:, None] * r).ravel(), J

    def plus(self, x, delta):
        N = self.N
        q, t, X, intr = (a.copy() for a in self.unpack(x))
        d = delta[: 6 * N].reshape(N, 6)
        dr = d[:, :3]
        nrm = np.linalg.norm(dr, axis=1)
        safe = np.where(nrm > 0, nrm, 1.0)
        k = np.where(nrm > 0, np.sin(nrm) / safe, 1.0)
        qd = np.concatenate([np.cos(nrm)[:, None], k[:, None] * dr], axis=1)
        q = quat_mul(qd, q)
        t = t + d[:, 3:]
        X = X + delta[self.pt_col0 :].reshape(-1, 3)
        free = self.fmask
        intr[free] += delta[6 * N : self.pt_col0]
        return self.pack(q, t, X, intr)

    def x_norm(self, x):
        return float(np.linalg.norm(x))

    def step_norm(self, x, cand):
        return float(np.linalg.norm(cand - x))


def solve(num_cams, pt_offset, obs_cam, obs_xy, cam_intr, intr_model, fixed_cam, cam_q, cam_t, pt_xyz,
          intr_params, options: BundleAdjusterOptions | None = None, image_frame=None, image_cam_from_rig=None,
          image_intr=None, image_sensor=None, sensor_cam_from_rig=None):
    """Returns (ok, q [N,4], t [N,3], X [P,3], intr [K,8], LmSummary); arrays as glomap_amd.flat.BaProblem.
    Known rigs: with `image_frame` [I], `image_cam_from_rig` [I,7] (qw,qx,qy,qz,tx,ty,tz) and `image_intr` [I] given,
    obs_cam indexes IMAGES; the pose blocks are the frames' rig_from_world.
    With `image_sensor` [I] (-1 = reference sensor) and `sensor_cam_from_rig` [S,7] the cam_from_rig of image i is the
    sensor's entry: constant unless options.optimize_rig_poses, else S more pose blocks (RigReprojErrorCostFunctor,
    ba.cc:161-179) whose result is returned as summary.sensor_cam_from_rig."""
    b = build_problem(num_cams, pt_offset, obs_cam, obs_xy, cam_intr, intr_model, fixed_cam, cam_q, cam_t, pt_xyz, intr_params, options,
                      image_frame, image_cam_from_rig, image_intr, image_sensor, sensor_cam_from_rig)
    opt, N, S, used, X_all = b["opt"], int(num_cams), b["num_sensors"], b["used"], b["X_all"]
    if b["problem"] is None:
        return False, b["q0"][:N], b["t0"][:N], X_all, b["intr0"], lm.LmSummary(usable=False)
    prob = b["problem"]
    x, summ = lm.solve(prob, b["x0"], opt.lm)
    q, t, X, intr = prob.unpack(x)
    X_all[used] = X
    if S:
        summ.sensor_cam_from_rig = np.concatenate([q[N:], t[N:]], axis=1)
    return summ.usable, q[:N].copy(), t[:N].copy(), X_all, intr.copy(), summ


def build_problem(num_cams, pt_offset, obs_cam, obs_xy, cam_intr, intr_model, fixed_cam, cam_q, cam_t, pt_xyz, intr_params,
                  options: BundleAdjusterOptions | None = None, image_frame=None, image_cam_from_rig=None, image_intr=None,
                  image_sensor=None, sensor_cam_from_rig=None):
    """The problem BundleAdjuster::Solve poses (ba.cc:115-317), without minimising it: dict with `problem` (None when no
    observation is left), `x0`, `used` (tracks in the problem), the start arrays.  tests/test_oracle_ref_ba.py holds it to
    the reference's own builder."""
    opt = options or BundleAdjusterOptions()
    N = int(num_cams)
    pt_offset = np.asarray(pt_offset, dtype=np.int64)
    lens = np.diff(pt_offset)
    P_all = lens.shape[0]
    used = lens >= opt.min_num_view_per_track  # ba.cc:122
    obs_pt_all = np.repeat(np.arange(P_all), lens)
    keep = used[obs_pt_all]
    remap = -np.ones(P_all, dtype=np.int64)
    remap[used] = np.arange(int(used.sum()))
    cam = np.asarray(obs_cam, dtype=np.int64)[keep]
    obs_ik = Rs = ts = obs_sens = None
    S = 0
    q0 = np.array(cam_q, dtype=np.float64, copy=True)
    t0 = np.array(cam_t, dtype=np.float64, copy=True)
    if image_frame is not None:
        cfr = np.array(image_cam_from_rig, dtype=np.float64, copy=True)
        if image_sensor is not None:
            isen = np.asarray(image_sensor, dtype=np.int64)
            scfr = np.asarray(sensor_cam_from_rig, dtype=np.float64)
            cfr[isen >= 0] = scfr[isen[isen >= 0]]
            if opt.optimize_rig_poses:
                S = scfr.shape[0]
                obs_sens = np.where(isen[cam] >= 0, N + isen[cam], -1)
                q0 = np.concatenate([q0, scfr[:, :4]])
                t0 = np.concatenate([t0, scfr[:, 4:7]])
        obs_ik = np.asarray(image_intr, dtype=np.int64)[cam]
        Rs = quat_to_rot(cfr[cam, :4])
        ts = cfr[cam, 4:7]
        cam = np.asarray(image_frame, dtype=np.int64)[cam]
    pt = remap[obs_pt_all[keep]]
    xy = np.asarray(obs_xy, dtype=np.float64)[keep]
    X_all = np.array(pt_xyz, dtype=np.float64, copy=True)
    intr0 = np.array(intr_params, dtype=np.float64, copy=True)
    out = dict(opt=opt, num_sensors=S, used=used, X_all=X_all, q0=q0, t0=t0, intr0=intr0, problem=None, x0=None)
    if cam.shape[0] == 0:
        return out
    prob = _BaProblem(N + S, cam, pt, xy, None if cam_intr is None else np.asarray(cam_intr, dtype=np.int64),
                      np.asarray(intr_model, dtype=np.int64), int(fixed_cam), int(used.sum()), opt, obs_ik, Rs, ts,
                      obs_sens, S, width=intr0.shape[1])
    out["problem"], out["x0"] = prob, prob.pack(q0, t0, X_all[used], intr0)
    return out
