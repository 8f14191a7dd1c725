"""ORACLE (test infrastructure only — never imported by the product path).

CPU restatement (numpy/scipy, sparse direct solves) of the reference's rotation averaging,
3-DoF path with trivial rigs — the path `glomap mapper` exercises (use_gravity = false,
glomap/estimators/global_rotation_averaging.h:74).

Follows, function by function:
  maximum_spanning_tree_init : gra.cc:87-138, math/tree.cc:26-153
  setup_linear_system        : gra.cc:141-477   (A rows: -I3 @ image_id1, +I3 @ image_id2; 3 gauge rows)
  compute_residuals          : gra.cc:696-756
  update_global_rotations    : gra.cc:627-644
  average_step_size          : gra.cc:758-772
  solve_l1_regression        : gra.cc:479-541 + colmap::LeastAbsoluteDeviationSolver
                               (COLMAP @ b6b7b54e, src/colmap/optim/least_absolute_deviations.cc —
                               un-vendored; restated from its published ADMM algorithm)
  solve_irls                 : gra.cc:543-625
  estimate_rotations         : gra.cc:40-85

PINNED TO REFERENCE CODE (round 5): global_rotation_averaging.cc, rotation_initializer.cc, math/rigid3d.cc and math/tree.cc
compile, unmodified, against the Eigen / COLMAP / Boost stand-ins of oracle/ref_shim_ra/ (oracle/_ref/libref_glomap_ra.so);
tests/test_oracle_ref_ra.py holds all three variants below to RotationEstimator::EstimateRotations as the reference wrote it:
same L1 / IRLS iteration counts, rotations equal to 1e-14 rad.  What stays a restatement is the ADMM inside
colmap::LeastAbsoluteDeviationSolver (un-vendored COLMAP; the stand-in and this file state it identically) — "parity unpinned"
for that part.  The reference stores no numeric vectors for RA; its tests pin ground-truth recovery tolerances on synthetic
scenes (rotation_averager_test.cc:166-167, 309-310), which tests/test_oracle_ra.py reproduces with glomap_amd.synthetic.
"""
from __future__ import annotations

from dataclasses import dataclass, field

import numpy as np
import scipy.sparse as sp
import scipy.sparse.linalg as spla

from . import so3

GEMAN_MCCLURE = 0
HALF_NORM = 1


@dataclass
class RotationEstimatorOptions:
    """Mirror of glomap/estimators/global_rotation_averaging.h:39-75."""

    max_num_l1_iterations: int = 5
    l1_step_convergence_threshold: float = 0.001
    max_num_irls_iterations: int = 100
    irls_step_convergence_threshold: float = 0.001
    irls_loss_parameter_sigma: float = 5.0  # degrees
    weight_type: int = GEMAN_MCCLURE
    skip_initialization: bool = False
    use_weight: bool = False
    # colmap::LeastAbsoluteDeviationSolver::Options as set at gra.cc:483-486 (+ upstream defaults)
    l1_admm_max_num_iterations: int = 10
    l1_admm_rho: float = 1.0
    l1_admm_alpha: float = 1.0
    l1_admm_absolute_tolerance: float = 1e-4
    l1_admm_relative_tolerance: float = 1e-2


@dataclass
class RaTrace:
    l1_iterations: int = 0
    irls_iterations: int = 0
    l1_steps: list = field(default_factory=list)
    irls_steps: list = field(default_factory=list)


# ------------------------------------------------------------------------------------------
def maximum_spanning_tree_init(num_nodes, edge_i, edge_j, edge_R, edge_ninl, aa0, root=0):
    """Kruskal maximum spanning tree on #inliers (tree.cc:78-153), BFS from node 0, then
    R_child = R_rel * R_parent or R_rel^T * R_parent (gra.cc:125-134).  The root is never
    assigned in the reference loop (gra.cc:120 `continue`), so `cam_from_worlds[root]` is the
    default-constructed Rigid3d, i.e. the IDENTITY rotation, whatever the comment there says.
    Ties between equal inlier counts are broken by edge index (stable sort) — boost's order
    among ties is unspecified, the product uses the same rule."""
    E = edge_i.shape[0]
    max_w = float(edge_ninl.max()) if E else 0.0
    cost = max_w - edge_ninl.astype(np.float64)
    order = np.argsort(cost, kind="stable")  # boost kruskal sorts by weight
    parent = np.arange(num_nodes)

    def find(x):
        while parent[x] != x:
            parent[x] = parent[parent[x]]
            x = parent[x]
        return x

    adj = [[] for _ in range(num_nodes)]
    for e in order:
        a, b = int(edge_i[e]), int(edge_j[e])
        ra, rb = find(a), find(b)
        if ra != rb:
            parent[ra] = rb
            adj[a].append((b, int(e)))
            adj[b].append((a, int(e)))
    R = np.tile(np.eye(3), (num_nodes, 1, 1))
    visited = np.zeros(num_nodes, dtype=bool)
    visited[root] = True  # the reference's root is the first image its unordered_map yields (tree.cc:84-88, 141); 0 in the flat form
    queue = [root]
    head = 0
    while head < len(queue):
        cur = queue[head]
        head += 1
        for nb, e in adj[cur]:
            if visited[nb]:
                continue
            visited[nb] = True
            if int(edge_i[e]) == nb:
                # curr is image_id1: 1_R_w = 2_R_1^T * 2_R_w
                R[nb] = edge_R[e].T @ R[cur]
            else:
                R[nb] = edge_R[e] @ R[cur]
            queue.append(nb)
    # ConvertRotationsFromImageToRig for trivial rigs stores the quaternion of the matrix
    # (rotation_initializer.cc:7-125); SetupLinearSystem re-reads it as angle-axis (gra.cc:223-224).
    aa = so3.log_rot(R)
    # unreached nodes (other components) keep their input value
    aa[~visited] = aa0[~visited]
    return aa


def setup_linear_system(num_nodes, edge_i, edge_j, fixed_node):
    """A (3E+3) x 3N with -1 / +1 entries (gra.cc:396-421) and gauge rows (gra.cc:455-460)."""
    E = edge_i.shape[0]
    rows = np.arange(3 * E)
    comp = rows % 3
    e = rows // 3
    r = np.concatenate([rows, rows, 3 * E + np.arange(3)])
    c = np.concatenate([3 * edge_i[e] + comp, 3 * edge_j[e] + comp, 3 * fixed_node + np.arange(3)])
    v = np.concatenate([-np.ones(3 * E), np.ones(3 * E), np.ones(3)])
    return sp.csr_matrix((v, (r, c)), shape=(3 * E + 3, 3 * num_nodes))


def compute_residuals(rot, edge_i, edge_j, edge_R, fixed_node, fixed_rot):
    """b_e = -Log(R_j^T R_rel R_i) (gra.cc:741-742); gauge rows Log(R_fix0^T R_fix) (gra.cc:751-755)."""
    Rn = so3.exp_aa(rot)
    Ri = Rn[edge_i]
    Rj = Rn[edge_j]
    M = np.transpose(Rj, (0, 2, 1)) @ edge_R @ Ri
    b = -so3.log_rot(M)
    g = so3.log_rot(so3.exp_aa(fixed_rot).T @ Rn[fixed_node])
    return np.concatenate([b.reshape(-1), g])


def update_global_rotations(rot, step):
    """r <- Log(Exp(r) * Exp(-delta)) (gra.cc:635-640)."""
    return so3.log_rot(so3.exp_aa(rot) @ so3.exp_aa(-step))


def average_step_size(step):
    return float(np.linalg.norm(step, axis=1).sum() / step.shape[0])


def _shrink(v, k):
    return np.maximum(0.0, v - k) - np.maximum(0.0, -v - k)


class LeastAbsoluteDeviationSolver:
    """ADMM for min |A x - b|_1 (COLMAP's restatement of Theia's L1Solver; Boyd et al. §6.1)."""

    def __init__(self, A, opt: RotationEstimatorOptions):
        self.A = A.tocsr()
        self.At = A.T.tocsr()
        self.opt = opt
        self.lu = spla.splu((self.At @ self.A).tocsc())  # LLT in the reference; SPD system

    def solve(self, b):
        A, At, o = self.A, self.At, self.opt
        m, n = A.shape
        z = np.zeros(m)
        u = np.zeros(m)
        x = np.zeros(n)
        rhs_norm = np.linalg.norm(b)
        primal_abs = np.sqrt(m) * o.l1_admm_absolute_tolerance
        dual_abs = np.sqrt(n) * o.l1_admm_absolute_tolerance
        for _ in range(o.l1_admm_max_num_iterations):
            x = self.lu.solve(At @ (b + z - u))
            Ax = A @ x
            Ax_hat = o.l1_admm_alpha * Ax + (1.0 - o.l1_admm_alpha) * (z + b)
            z_old = z
            z = _shrink(Ax_hat - b + u, 1.0 / o.l1_admm_rho)
            u = u + Ax_hat - z - b
            r_norm = np.linalg.norm(Ax - z - b)
            s_norm = np.linalg.norm(-o.l1_admm_rho * (At @ (z - z_old)))
            max_norm = max(np.linalg.norm(Ax), np.linalg.norm(z), rhs_norm)
            primal_eps = primal_abs + o.l1_admm_relative_tolerance * max_norm
            dual_eps = dual_abs + o.l1_admm_relative_tolerance * np.linalg.norm(o.l1_admm_rho * (At @ u))
            if r_norm < primal_eps and s_norm < dual_eps:
                break
        return x


def estimate_rotations(
    num_nodes,
    edge_i,
    edge_j,
    edge_q,
    edge_weight,
    edge_ninl,
    node_aa0,
    fixed_node=0,
    options: RotationEstimatorOptions | None = None,
    trace: RaTrace | None = None,
    tree_root: int = 0,
):
    """Returns (ok, rot_aa[N,3])."""
    opt = options or RotationEstimatorOptions()
    edge_i = np.asarray(edge_i, dtype=np.int64)
    edge_j = np.asarray(edge_j, dtype=np.int64)
    edge_R = so3.quat_wxyz_to_rotmat(np.asarray(edge_q, dtype=np.float64))
    E = edge_i.shape[0]
    rot = np.array(node_aa0, dtype=np.float64, copy=True)
    if not opt.skip_initialization:
        rot = maximum_spanning_tree_init(num_nodes, edge_i, edge_j, edge_R, edge_ninl, rot, tree_root)
    # fixed camera keeps its initial (post-MST) rotation (gra.cc:248-257)
    fixed_rot = rot[fixed_node].copy()

    A = setup_linear_system(num_nodes, edge_i, edge_j, fixed_node)
    if opt.use_weight:
        ew = np.where(edge_weight >= 0, edge_weight, 1.0)  # gra.cc:417-420
        weights = np.concatenate([np.repeat(ew, 3), np.ones(3)])
    else:
        weights = np.ones(3 * E + 3)

    def residuals(r):
        return compute_residuals(r, edge_i, edge_j, edge_R, fixed_node, fixed_rot)

    # ---- L1 (gra.cc:479-541)
    if opt.max_num_l1_iterations > 0:
        WA = sp.diags(weights) @ A
        l1 = LeastAbsoluteDeviationSolver(WA, opt)
        last_norm = 0.0
        curr_norm = 0.0
        b = residuals(rot)
        it = 0
        for it in range(opt.max_num_l1_iterations):
            last_norm = curr_norm
            step = l1.solve(weights * b)
            if np.isnan(step).any():
                return False, rot
            curr_norm = float(np.linalg.norm(step))
            step3 = step.reshape(-1, 3)
            rot = update_global_rotations(rot, step3)
            b = residuals(rot)
            avg = average_step_size(step3)
            if trace is not None:
                trace.l1_steps.append(avg)
                trace.l1_iterations = it + 1
            if avg < opt.l1_step_convergence_threshold or abs(last_norm - curr_norm) < so3.EPS:
                break

    # ---- IRLS (gra.cc:543-625)
    if opt.max_num_irls_iterations > 0:
        sigma = np.radians(opt.irls_loss_parameter_sigma)
        At = A.T.tocsr()
        b = residuals(rot)
        w_irls = np.ones(3 * E + 3)
        for it in range(opt.max_num_irls_iterations):
            e2 = (b[: 3 * E].reshape(-1, 3) ** 2).sum(axis=1)
            if opt.weight_type == GEMAN_MCCLURE:
                tmp = e2 + sigma * sigma
                w = sigma * sigma / (tmp * tmp)
            else:
                with np.errstate(divide="ignore"):
                    w = np.power(e2, (0.5 - 2) / 2)
            if np.isnan(w).any():
                return False, rot
            w_irls[: 3 * E] = np.repeat(w, 3)
            at_weight = At @ sp.diags(w_irls * weights)
            H = (at_weight @ A).tocsc()
            step = spla.splu(H).solve(at_weight @ b)
            step3 = step.reshape(-1, 3)
            rot = update_global_rotations(rot, step3)
            b = residuals(rot)
            avg = average_step_size(step3)
            if trace is not None:
                trace.irls_steps.append(avg)
                trace.irls_iterations = it + 1
            if avg < opt.irls_step_convergence_threshold:
                break
    return True, rot


# ------------------------------------------------------------------------------------------
# Rigs with unknown cam_from_rig rotations (gra.cc:173-191, 396-446, 646-693, 718-739)
# ------------------------------------------------------------------------------------------
def average_quaternions(q):
    """colmap::AverageQuaternions with unit weights: principal eigenvector of sum q q^T (sign-free)."""
    q = np.asarray(q, dtype=np.float64)
    w, v = np.linalg.eigh(q.T @ q)
    return v[:, -1]


def estimate_rotations_rig(num_frames, num_cams, image_frame, image_cam, edge_i, edge_j, edge_q, edge_weight, frame_aa0,
                           cam_aa0, fixed_frame=0, options: RotationEstimatorOptions | None = None,
                           trace: RaTrace | None = None):
    """RotationEstimator::EstimateRotations with cam_from_rig rotations among the unknowns, skip_initialization = true
    (the spanning-tree start and ConvertRotationsFromImageToRig are the caller's, rotation_averager.cc:66-172).

    Nodes of the view graph are IMAGES: image i belongs to frame image_frame[i] and, when its sensor's cam_from_rig is to be
    estimated, to block image_cam[i] (else -1: reference sensor, or a calibrated sensor whose cam_from_rig the caller has
    folded into the relative rotations, gra.cc:306-309).  cam_from_world(i) = Exp(cam) Exp(frame) (gra.cc:729-738); the
    row of an edge carries -1 / +1 at the frame columns and at the cam columns of its two images (gra.cc:396-446; equal
    columns cancel).  Unknowns: [frames | cams]; gauge rows on `fixed_frame`.  Returns (ok, frame_aa [N,3], cam_aa [C,3])."""
    opt = options or RotationEstimatorOptions()
    N, C = int(num_frames), int(num_cams)
    imf = np.asarray(image_frame, dtype=np.int64)
    imc = np.asarray(image_cam, dtype=np.int64)
    I = imf.shape[0]
    edge_i = np.asarray(edge_i, dtype=np.int64)
    edge_j = np.asarray(edge_j, dtype=np.int64)
    edge_R = so3.quat_wxyz_to_rotmat(np.asarray(edge_q, dtype=np.float64))
    E = edge_i.shape[0]
    rot_f = np.array(frame_aa0, dtype=np.float64, copy=True).reshape(N, 3)
    rot_c = np.array(cam_aa0, dtype=np.float64, copy=True).reshape(C, 3)
    fixed_rot = rot_f[fixed_frame].copy()

    # V: image tangent = frame tangent + cam tangent
    has = imc >= 0
    Vs = sp.csr_matrix((np.ones(I + int(has.sum())), (np.concatenate([np.arange(I), np.arange(I)[has]]),
                                                      np.concatenate([imf, N + imc[has]]))), shape=(I, N + C))
    V3 = sp.kron(Vs, sp.eye(3), format="csr")
    A_img = setup_linear_system(I, edge_i, edge_j, 0)[: 3 * E]
    gauge = sp.csr_matrix((np.ones(3), (np.arange(3), 3 * fixed_frame + np.arange(3))), shape=(3, 3 * (N + C)))
    A = sp.vstack([A_img @ V3, gauge]).tocsr()
    A.eliminate_zeros()
    if opt.use_weight:
        ew = np.where(np.asarray(edge_weight) >= 0, edge_weight, 1.0)
        weights = np.concatenate([np.repeat(ew, 3), np.ones(3)])
    else:
        weights = np.ones(3 * E + 3)
    cam_images = [np.nonzero(imc == c)[0] for c in range(C)]

    def residuals(rf, rc):
        Rf, Rc = so3.exp_aa(rf), so3.exp_aa(rc)
        Ri = Rf[imf].copy()
        Ri[has] = Rc[imc[has]] @ Ri[has]  # gra.cc:729-738
        M = np.transpose(Ri[edge_j], (0, 2, 1)) @ edge_R @ Ri[edge_i]
        b = -so3.log_rot(M)
        g = so3.log_rot(so3.exp_aa(fixed_rot).T @ Rf[fixed_frame])
        return np.concatenate([b.reshape(-1), g])

    def update(rf, rc, step):
        sf, sc = step[:N], step[N:]
        rf = so3.log_rot(so3.exp_aa(rf) @ so3.exp_aa(-sf))  # gra.cc:632-645
        Rf = so3.exp_aa(rf)  # the UPDATED frames enter the cam update (gra.cc:651-671)
        rc_new = rc.copy()
        for c in range(C):
            if cam_images[c].shape[0] == 0:
                continue
            R_ori = so3.exp_aa(rc[c][None])[0]
            R_upd = so3.exp_aa(-sc[c][None])[0]
            Rs = Rf[imf[cam_images[c]]]
            qs = so3.rotmat_to_quat_eigen(R_ori @ Rs @ R_upd @ np.transpose(Rs, (0, 2, 1)))  # gra.cc:676-686
            rc_new[c] = so3.log_rot(so3.quat_wxyz_to_rotmat(average_quaternions(qs)[None]))[0]
        return rf, rc_new

    def avg_step(step):  # frames only, over the number of frames (gra.cc:758-772)
        return float(np.linalg.norm(step[:N], axis=1).sum() / N)

    if opt.max_num_l1_iterations > 0:
        l1 = LeastAbsoluteDeviationSolver(sp.diags(weights) @ A, opt)
        last_norm = curr_norm = 0.0
        b = residuals(rot_f, rot_c)
        for it in range(opt.max_num_l1_iterations):
            last_norm = curr_norm
            step = l1.solve(weights * b)
            if np.isnan(step).any():
                return False, rot_f, rot_c
            curr_norm = float(np.linalg.norm(step))
            step3 = step.reshape(-1, 3)
            rot_f, rot_c = update(rot_f, rot_c, step3)
            b = residuals(rot_f, rot_c)
            avg = avg_step(step3)
            if trace is not None:
                trace.l1_steps.append(avg)
                trace.l1_iterations = it + 1
            if avg < opt.l1_step_convergence_threshold or abs(last_norm - curr_norm) < so3.EPS:
                break
    if opt.max_num_irls_iterations > 0:
        sigma = np.radians(opt.irls_loss_parameter_sigma)
        At = A.T.tocsr()
        b = residuals(rot_f, rot_c)
        w_irls = np.ones(3 * E + 3)
        for it in range(opt.max_num_irls_iterations):
            e2 = (b[: 3 * E].reshape(-1, 3) ** 2).sum(axis=1)
            if opt.weight_type == GEMAN_MCCLURE:
                tmp = e2 + sigma * sigma
                w = sigma * sigma / (tmp * tmp)
            else:
                with np.errstate(divide="ignore"):
                    w = np.power(e2, (0.5 - 2) / 2)
            if np.isnan(w).any():
                return False, rot_f, rot_c
            w_irls[: 3 * E] = np.repeat(w, 3)
            at_weight = At @ sp.diags(w_irls * weights)
            step = spla.splu((at_weight @ A).tocsc()).solve(at_weight @ b)
            step3 = step.reshape(-1, 3)
            rot_f, rot_c = update(rot_f, rot_c, step3)
            b = residuals(rot_f, rot_c)
            avg = avg_step(step3)
            if trace is not None:
                trace.irls_steps.append(avg)
                trace.irls_iterations = it + 1
            if avg < opt.irls_step_convergence_threshold:
                break
    return True, rot_f, rot_c


# ------------------------------------------------------------------------------------------
# Gravity-aligned frames: 1-DoF unknowns (gra.cc:19-36, 207-217, 312-341, 376-446, 455-468, 639-645, 709-713, 746-749)
# ------------------------------------------------------------------------------------------
def rel_angle_error(angle_12, angle_1, angle_2):
    """RelAngleError (gra.cc:19-36) without the rand() jitter near +-pi (not reproducible by construction)."""
    est = (angle_2 - angle_1) - angle_12
    return (est + np.pi) % (2.0 * np.pi) - np.pi  # [-pi, pi)


def estimate_rotations_gravity(num_nodes, edge_i, edge_j, edge_q, edge_weight, node_gravity, node_aa0, fixed_node=0,
                               options: RotationEstimatorOptions | None = None, trace: RaTrace | None = None):
    """RotationEstimator::EstimateRotations with use_gravity: a frame with gravity has ONE unknown, the angle about the
    (aligned) vertical, stored here as node_aa0[n] = (0, angle, 0) — AngleToRotUp(angle) = Exp((0, angle, 0)),
    math/gravity.cc:30-33 —, the others three.  edge_q are the relative rotations ALREADY aligned by the caller
    (R_align2^T R_rel R_align1, gra.cc:315-327).  No spanning-tree start in this mode (gra.cc:60-62).

    Rows exactly as SetupLinearSystem writes them: a pair of two gravity frames has one row, -1 / +1 on the two angles
    (gra.cc:390-397), residual RelAngleError of the y components, constant xz_error in the IRLS weight (gra.cc:329-338,
    571-573); other pairs have three rows where a gravity frame only appears in the y row (gra.cc:399-418); the gauge is
    one row when the fixed frame has gravity (gra.cc:455-460).  Returns (ok, rot_aa [N,3])."""
    opt = options or RotationEstimatorOptions()
    N = int(num_nodes)
    grav = np.asarray(node_gravity).astype(bool)
    edge_i = np.asarray(edge_i, dtype=np.int64)
    edge_j = np.asarray(edge_j, dtype=np.int64)
    edge_R = so3.quat_wxyz_to_rotmat(np.asarray(edge_q, dtype=np.float64))
    E = edge_i.shape[0]
    rot = np.array(node_aa0, dtype=np.float64, copy=True)
    # unknown layout: one column per gravity frame, three per other frame
    col0 = np.concatenate([[0], np.cumsum(np.where(grav, 1, 3))])
    ncol = int(col0[-1])
    aa_rel = so3.log_rot(edge_R)
    both = grav[edge_i] & grav[edge_j]
    xz_err = aa_rel[:, 0] ** 2 + aa_rel[:, 2] ** 2
    angle_rel = aa_rel[:, 1]
    row0 = np.concatenate([[0], np.cumsum(np.where(both, 1, 3))])
    nrow_e = int(row0[-1])
    gauge_rows = 1 if grav[fixed_node] else 3
    # rows (vectorised): one-row pairs, then per endpoint of the three-row pairs either its y entry or all three
    e1 = np.nonzero(both)[0]
    e3 = np.nonzero(~both)[0]
    ri = [row0[e1], row0[e1]]
    ci = [col0[edge_i[e1]], col0[edge_j[e1]]]
    vi = [-np.ones(e1.size), np.ones(e1.size)]
    for node_arr, sgn in ((edge_i[e3], -1.0), (edge_j[e3], 1.0)):
        g1 = grav[node_arr]
        ri.append(row0[e3][g1] + 1)
        ci.append(col0[node_arr[g1]])
        vi.append(np.full(int(g1.sum()), sgn))
        for c in range(3):
            ri.append(row0[e3][~g1] + c)
            ci.append(col0[node_arr[~g1]] + c)
            vi.append(np.full(int((~g1).sum()), sgn))
    ri.append(nrow_e + np.arange(gauge_rows))
    ci.append(col0[fixed_node] + np.arange(gauge_rows))
    vi.append(np.ones(gauge_rows))
    A = sp.csr_matrix((np.concatenate(vi), (np.concatenate(ri), np.concatenate(ci))), shape=(nrow_e + gauge_rows, ncol))
    ew = np.where(np.asarray(edge_weight) >= 0, edge_weight, 1.0) if opt.use_weight else np.ones(E)
    weights = np.concatenate([np.repeat(ew, np.where(both, 1, 3)), np.ones(gauge_rows)])
    fixed_rot = rot[fixed_node].copy()
    rows3 = (row0[e3][:, None] + np.arange(3)[None, :]).ravel()
    gidx = np.nonzero(grav)[0]
    nidx = np.nonzero(~grav)[0]
    ncols3 = (col0[nidx][:, None] + np.arange(3)[None, :]).ravel()

    def residuals(r):
        Rn = so3.exp_aa(r)
        b = np.empty(nrow_e + gauge_rows)
        if e3.size:
            M = np.transpose(Rn[edge_j[e3]], (0, 2, 1)) @ edge_R[e3] @ Rn[edge_i[e3]]
            b[rows3] = -so3.log_rot(M).ravel()
        b[row0[e1]] = rel_angle_error(angle_rel[e1], r[edge_i[e1], 1], r[edge_j[e1], 1])
        if grav[fixed_node]:
            b[nrow_e] = r[fixed_node, 1] - fixed_rot[1]  # gra.cc:746-749
        else:
            b[nrow_e:] = so3.log_rot(so3.exp_aa(fixed_rot).T @ Rn[fixed_node])
        return b

    def update(r, step):
        r = r.copy()
        r[gidx, 1] -= step[col0[gidx]]  # gra.cc:643-644
        if nidx.size:
            s3 = step[ncols3].reshape(-1, 3)
            r[nidx] = so3.log_rot(so3.exp_aa(r[nidx]) @ so3.exp_aa(-s3))
        return r

    def avg_step(step):
        tot = np.abs(step[col0[gidx]]).sum()
        if nidx.size:
            tot += np.linalg.norm(step[ncols3].reshape(-1, 3), axis=1).sum()
        return float(tot / N)

    def irls_weights(b):
        e2 = np.empty(E)
        e2[e1] = b[row0[e1]] ** 2 + xz_err[e1]
        if e3.size:
            e2[e3] = (b[rows3].reshape(-1, 3) ** 2).sum(axis=1)
        if opt.weight_type == GEMAN_MCCLURE:
            tmp = e2 + sigma * sigma
            w = sigma * sigma / (tmp * tmp)
        else:
            with np.errstate(divide="ignore"):
                w = np.power(e2, (0.5 - 2) / 2)
        return np.concatenate([np.repeat(w, np.where(both, 1, 3)), np.ones(gauge_rows)])

    sigma = np.radians(opt.irls_loss_parameter_sigma)
    if opt.max_num_l1_iterations > 0:
        l1 = LeastAbsoluteDeviationSolver(sp.diags(weights) @ A, opt)
        last_norm = curr_norm = 0.0
        b = residuals(rot)
        for it in range(opt.max_num_l1_iterations):
            last_norm = curr_norm
            step = l1.solve(weights * b)
            if np.isnan(step).any():
                return False, rot
            curr_norm = float(np.linalg.norm(step))
            rot = update(rot, step)
            b = residuals(rot)
            avg = avg_step(step)
            if trace is not None:
                trace.l1_steps.append(avg)
                trace.l1_iterations = it + 1
            if avg < opt.l1_step_convergence_threshold or abs(last_norm - curr_norm) < so3.EPS:
                break
    if opt.max_num_irls_iterations > 0:
        sigma = np.radians(opt.irls_loss_parameter_sigma)
        At = A.T.tocsr()
        b = residuals(rot)
        for it in range(opt.max_num_irls_iterations):
            w_irls = irls_weights(b)
            if np.isnan(w_irls).any():
                return False, rot
            at_weight = At @ sp.diags(w_irls * weights)
            step = spla.splu((at_weight @ A).tocsc()).solve(at_weight @ b)
            rot = update(rot, step)
            b = residuals(rot)
            avg = avg_step(step)
            if trace is not None:
                trace.irls_steps.append(avg)
                trace.irls_iterations = it + 1
            if avg < opt.irls_step_convergence_threshold:
                break
    return True, rot
