"""Host-side mirror of the reference's estimator interface on top of the C ABI (libgsfm.so).

Two levels:
  * flat level  — ra_solve / gp_solve / ba_solve take the SoA problems of glomap_amd.flat (numpy
    arrays on the host, or DeviceArrays already resident in HBM) and call the C ABI directly;
  * scene level — RotationEstimator / GlobalPositioner / BundleAdjuster keep the reference's
    class names, option names and bool-returning methods
    (global_rotation_averaging.h:79-87, global_positioning.h:58-68, bundle_adjustment.h:40-51)
    over dict-based scene containers (glomap_amd.scene), doing the pack / unpack the C++
    adapter include/gsfm_glomap_adapter.hpp does for the real GLOMAP types.

Nothing here computes on the CPU: without libgsfm.so and a HIP device every call raises.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass, field
from typing import Dict, Optional

import numpy as np

from . import _lib, so3
from .flat import CAMERA_MAX_PARAMS, CAMERA_MAX_PARAMS_WIDE, BaProblem, GpProblem, RaProblem

_default_ctx: Optional[_lib.Context] = None


def default_context() -> _lib.Context:
    global _default_ctx
    if _default_ctx is None:
        _default_ctx = _lib.Context(-1)
    return _default_ctx


def _is_dev(a) -> bool:
    return isinstance(a, _lib.DeviceArray)


def _mem_of(*arrays) -> int:
    kinds = {_is_dev(a) for a in arrays if a is not None}
    if len(kinds) != 1:
        raise ValueError("all problem arrays must live in the same memory space")
    return _lib.GSFM_MEM_DEVICE if kinds.pop() else _lib.GSFM_MEM_HOST


def _h(a, dtype):
    """numpy: contiguous array of dtype; DeviceArray: dtype checked, returned as-is."""
    if a is None:
        return None
    if isinstance(a, np.ndarray):
        return np.ascontiguousarray(a, dtype=dtype)
    if not _is_dev(a):
        raise TypeError("problem arrays must be numpy arrays (host) or glomap_amd DeviceArrays (HBM)")
    assert a.dtype == np.dtype(dtype), f"device array has dtype {a.dtype}, expected {np.dtype(dtype)}"
    return a


# ============================================================================================
# Options (field names and defaults = the reference's structs)
# ============================================================================================
@dataclass
class RotationEstimatorOptions:
    """glomap/estimators/global_rotation_averaging.h:39-75."""

    max_num_l1_iterations: int = 5
    l1_step_convergence_threshold: float = 0.001
    max_num_irls_iterations: int = 100
    irls_step_convergence_threshold: float = 0.001
    irls_loss_parameter_sigma: float = 5.0
    weight_type: int = 0  # GEMAN_MCCLURE
    skip_initialization: bool = False
    use_weight: bool = False
    use_gravity: bool = False
    # linear solver (replaces CHOLMOD)
    pcg_relative_tolerance: float = 1e-10
    pcg_max_iterations: int = 2000
    force_iterative: bool = False  # True: PCG even where the dense direct solve applies (N <= 2048)
    pcg_relative_tolerance_admm: float = 1e-10  # x-updates inside the ADMM loop (warm-started corrections)

    GEMAN_MCCLURE = 0
    HALF_NORM = 1

    def to_c(self) -> _lib.RaOptions:
        o = _lib.RaOptions()
        _lib.load().gsfm_ra_options_default(C.byref(o))
        for name in (
            "max_num_l1_iterations l1_step_convergence_threshold max_num_irls_iterations "
            "irls_step_convergence_threshold irls_loss_parameter_sigma weight_type "
            "pcg_relative_tolerance pcg_max_iterations pcg_relative_tolerance_admm"
        ).split():
            setattr(o, name, getattr(self, name))
        o.skip_initialization = int(self.skip_initialization)
        o.use_weight = int(self.use_weight)
        o.use_gravity = int(self.use_gravity)
        o.force_iterative = int(self.force_iterative)
        return o


@dataclass
class SolverOptions:
    """The ceres::Solver::Options fields the reference touches (optimization_base.h:18-23) plus
    the linear-solver knobs of the implicit-Schur PCG that replaces SPARSE_SCHUR."""

    max_num_iterations: int = 100
    function_tolerance: float = 1e-5
    # None = what gsfm_{gp,ba}_options_default choose for the estimator (they differ: include/gsfm.h), so that a
    # SolverOptions built for another field does not silently replace them
    pcg_relative_tolerance: Optional[float] = None
    pcg_max_iterations: int = 1000
    # Ceres' projected line search on bounds-constrained problems (global positioning); None = the library default (20), 0 = off
    max_num_line_search_step_size_iterations: Optional[int] = None


@dataclass
class GlobalPositionerOptions:
    """glomap/estimators/global_positioning.h:9-54."""

    generate_random_positions: bool = True
    generate_random_points: bool = True
    generate_scales: bool = True
    optimize_positions: bool = True
    optimize_points: bool = True
    optimize_scales: bool = True
    min_num_view_per_track: int = 3
    seed: int = 1
    constraint_type: int = 0  # ONLY_POINTS, 1 = ONLY_CAMERAS, 2 = POINTS_AND_CAMERAS_BALANCED, 3 = POINTS_AND_CAMERAS
    constraint_reweight_scale: float = 1.0  # POINTS_AND_CAMERAS_BALANCED only (global_positioning.h:40-41)
    thres_loss_function: float = 1e-1
    # first of the three draws of a random start vector -> x (0: clang-built reference) or -> z (1: g++-built); include/gsfm.h
    rand_vector_order: int = 0
    solver_options: SolverOptions = field(default_factory=lambda: SolverOptions(max_num_iterations=100))

    def to_c(self) -> _lib.GpOptions:
        o = _lib.GpOptions()
        _lib.load().gsfm_gp_options_default(C.byref(o))
        for name in (
            "generate_random_positions generate_random_points generate_scales optimize_positions "
            "optimize_points optimize_scales min_num_view_per_track seed constraint_type rand_vector_order"
        ).split():
            setattr(o, name, int(getattr(self, name)))
        o.thres_loss_function = self.thres_loss_function
        o.constraint_reweight_scale = self.constraint_reweight_scale
        _fill_lm(o.lm, self.solver_options)
        return o


@dataclass
class BundleAdjusterOptions:
    """glomap/estimators/bundle_adjustment.h:12-37."""

    optimize_rig_poses: bool = False
    optimize_rotations: bool = True
    optimize_translation: bool = True
    optimize_intrinsics: bool = True
    optimize_principal_point: bool = False
    optimize_points: bool = True
    min_num_view_per_track: int = 3
    thres_loss_function: float = 1.0
    # reduced solves to 1e-6 (gsfm_ba_options_default): 3.3e-7 rad / 3e-6 from the exact-solve trajectory at configs[3]
    solver_options: SolverOptions = field(default_factory=lambda: SolverOptions(max_num_iterations=200))

    def to_c(self) -> _lib.BaOptions:
        o = _lib.BaOptions()
        _lib.load().gsfm_ba_options_default(C.byref(o))
        for name in (
            "optimize_rotations optimize_translation optimize_intrinsics optimize_principal_point "
            "optimize_points min_num_view_per_track optimize_rig_poses"
        ).split():
            setattr(o, name, int(getattr(self, name)))
        o.thres_loss_function = self.thres_loss_function
        _fill_lm(o.lm, self.solver_options)
        return o


def _fill_lm(lm: _lib.LmOptions, so: SolverOptions):
    lm.max_num_iterations = so.max_num_iterations
    lm.function_tolerance = so.function_tolerance
    if so.pcg_relative_tolerance is not None:
        lm.pcg_relative_tolerance = so.pcg_relative_tolerance
    lm.pcg_max_iterations = so.pcg_max_iterations
    if so.max_num_line_search_step_size_iterations is not None:
        lm.max_num_line_search_step_size_iterations = int(so.max_num_line_search_step_size_iterations)


# ============================================================================================
# Flat level
# ============================================================================================
def _ra_problem_c(p: RaProblem, keep: list) -> _lib.RaProblemC:
    ei, ej = _h(p.edge_i, np.int32), _h(p.edge_j, np.int32)
    eq, ew = _h(p.edge_q, np.float64), _h(p.edge_weight, np.float64)
    en = _h(p.edge_ninl, np.int32)
    keep += [ei, ej, eq, ew, en]
    c = _lib.RaProblemC()
    c.mem = _mem_of(ei, ej, eq, ew, en)
    c.num_nodes = int(p.num_nodes)
    c.num_edges = int(ei.shape[0])
    c.edge_i, c.edge_j, c.edge_q = _lib.ptr(ei), _lib.ptr(ej), _lib.ptr(eq)
    c.edge_weight, c.edge_ninl = _lib.ptr(ew), _lib.ptr(en)
    c.fixed_node = int(p.fixed_node)
    if getattr(p, "node_gravity", None) is not None:
        ng = _h(p.node_gravity, np.uint8)
        assert _mem_of(ng) == c.mem
        keep.append(ng)
        c.node_gravity = _lib.ptr(ng)
    return c


def ra_solve(p: RaProblem, options: Optional[RotationEstimatorOptions] = None, ctx=None, rot_inout=None):
    """gsfm_ra_solve.  Returns (status, rot_aa [N,3] (same kind as the inputs), report dict)."""
    ctx = ctx or default_context()
    opt = (options or RotationEstimatorOptions()).to_c()
    keep: list = []
    c = _ra_problem_c(p, keep)
    if rot_inout is None:
        rot_inout = p.node_aa0.copy() if isinstance(p.node_aa0, np.ndarray) else p.node_aa0.clone()
    rot_inout = _h(rot_inout, np.float64)
    assert _mem_of(rot_inout) == c.mem
    cams = None
    if getattr(p, "image_frame", None) is not None:  # cam_from_rig rotations among the unknowns: nodes are images
        imf, imc = _h(p.image_frame, np.int32), _h(p.image_cam, np.int32)
        assert _mem_of(imf, imc) == c.mem
        keep += [imf, imc]
        cams = np.array(p.cam_aa0, dtype=np.float64, order="C", copy=True).reshape(-1, 3)  # host table, in/out
        c.num_images, c.image_frame, c.image_cam = int(imf.shape[0]), _lib.ptr(imf), _lib.ptr(imc)
        c.num_cams, c.cam_rot_aa = int(cams.shape[0]), _lib.ptr(cams)
    rep = _lib.Report()
    rc = ctx.lib.gsfm_ra_solve(ctx.handle, C.byref(c), C.byref(opt), _lib.ptr(rot_inout), C.byref(rep))
    report = rep.as_dict()
    if cams is not None:
        report["cam_rot_aa"] = cams
    return rc, rot_inout, report


def ra_residuals(p: RaProblem, rot_aa: np.ndarray, options: Optional[RotationEstimatorOptions] = None, ctx=None):
    ctx = ctx or default_context()
    opt = (options or RotationEstimatorOptions()).to_c()
    keep: list = []
    c = _ra_problem_c(p, keep)
    rot = _h(rot_aa, np.float64)
    E = c.num_edges
    res = np.empty((E, 3))
    w = np.empty(E)
    assert c.mem == _lib.GSFM_MEM_HOST
    rc = ctx.lib.gsfm_ra_residuals(ctx.handle, C.byref(c), C.byref(opt), _lib.ptr(rot), _lib.ptr(res), _lib.ptr(w))
    if rc != 0:
        raise _lib.GsfmError(rc, "gsfm_ra_residuals")
    return res, w


def ra_residuals_timed(p: RaProblem, rot_aa, repeat: int = 10, options: Optional[RotationEstimatorOptions] = None, ctx=None):
    """Average kernel milliseconds of the per-edge residual + IRLS-weight sweep (gsfm_ra_residuals_timed)."""
    ctx = ctx or default_context()
    opt = (options or RotationEstimatorOptions()).to_c()
    keep: list = []
    c = _ra_problem_c(p, keep)
    rot = _h(rot_aa, np.float64)
    assert _mem_of(rot) == c.mem
    ms = C.c_double(0.0)
    rc = ctx.lib.gsfm_ra_residuals_timed(ctx.handle, C.byref(c), C.byref(opt), _lib.ptr(rot), repeat, C.byref(ms))
    if rc != 0:
        raise _lib.GsfmError(rc, "gsfm_ra_residuals_timed")
    return ms.value


def ra_laplacian_apply(p: RaProblem, w, x, repeat: int = 1, ctx=None):
    ctx = ctx or default_context()
    keep: list = []
    c = _ra_problem_c(p, keep)
    w, x = _h(w, np.float64), _h(x, np.float64)
    y = np.empty_like(x) if c.mem == _lib.GSFM_MEM_HOST else x.clone()
    ms = C.c_double(0.0)
    rc = ctx.lib.gsfm_ra_laplacian_apply(ctx.handle, C.byref(c), _lib.ptr(w), _lib.ptr(x), _lib.ptr(y), repeat, C.byref(ms))
    if rc != 0:
        raise _lib.GsfmError(rc, "gsfm_ra_laplacian_apply")
    return y, ms.value


def average_quaternions(q: np.ndarray) -> np.ndarray:
    """colmap::AverageQuaternions with unit weights: principal eigenvector of sum q q^T (w,x,y,z rows)."""
    A = q.T @ q
    w, v = np.linalg.eigh(A)
    out = v[:, -1]
    return out if out[0] >= 0 else -out


def ra_solve_known_rigs(num_frames: int, image_frame, image_cam_from_rig, pair_i, pair_j, pair_q, pair_ninl,
                        pair_weight=None, options: Optional[RotationEstimatorOptions] = None, ctx=None, fixed_frame: int = 0):
    """Rotation averaging of FRAMES (rigs in time) from IMAGE pairs when every cam_from_rig is known — what
    RotationEstimator::EstimateRotations does for calibrated rigs:

      * initialisation (unless skip_initialization): maximum spanning tree over the IMAGES (gra.cc:87-138), then
        ConvertRotationsFromImageToRig (rotation_initializer.cc:86-121): rig_from_world of a frame = quaternion average of
        cam_from_rig^-1 * cam_from_world over its images;
      * system: one node per frame, R_rel = R_cam2_from_rig2^T * R_cam2_from_cam1 * R_cam1_from_rig1 (gra.cc:306-309), pairs
        inside one frame skipped (gra.cc:300-304);
    both on the C ABI (gsfm_ra_solve): the first as a zero-iteration solve of the image-level graph, the second as a
    normal solve with skip_initialization.  Returns (status, frame rotations as angle-axis [F,3], report)."""
    from . import so3

    ctx = ctx or default_context()
    opt = options or RotationEstimatorOptions()
    imf = np.asarray(image_frame, np.int64)
    cfr = np.asarray(image_cam_from_rig, np.float64)
    R_s = so3.quat_to_rotmat(cfr[:, :4])
    pi, pj = np.asarray(pair_i, np.int64), np.asarray(pair_j, np.int64)
    E = pi.shape[0]
    w = np.ones(E) if pair_weight is None else np.asarray(pair_weight, np.float64)
    ninl = np.asarray(pair_ninl, np.int32)
    aa0 = np.zeros((num_frames, 3))
    if not opt.skip_initialization:
        img = RaProblem(int(imf.shape[0]), pi.astype(np.int32), pj.astype(np.int32), np.asarray(pair_q, np.float64), w, ninl,
                        np.zeros((imf.shape[0], 3)), 0)
        o0 = RotationEstimatorOptions(**{**vars(opt), "max_num_l1_iterations": 0, "max_num_irls_iterations": 0})
        rc, rot_img, _ = ra_solve(img, o0, ctx=ctx)
        if rc != 0:
            return rc, aa0, {}
        R_img = so3.aa_to_rotmat(rot_img)
        q_rig = so3.rotmat_to_quat(np.transpose(R_s, (0, 2, 1)) @ R_img)  # cam_from_rig^-1 * cam_from_world
        for f in range(num_frames):
            sel = np.nonzero(imf == f)[0]
            if sel.size:
                qs = q_rig[sel]
                qs = qs * np.where(qs @ qs[0] < 0, -1.0, 1.0)[:, None]
                aa0[f] = so3.quat_to_aa(average_quaternions(qs)[None])[0]
    keep = imf[pi] != imf[pj]
    R_rel = np.transpose(R_s[pj[keep]], (0, 2, 1)) @ so3.quat_to_rotmat(np.asarray(pair_q, np.float64)[keep]) @ R_s[pi[keep]]
    p = RaProblem(int(num_frames), imf[pi[keep]].astype(np.int32), imf[pj[keep]].astype(np.int32), so3.rotmat_to_quat(R_rel),
                  w[keep], ninl[keep], aa0, int(fixed_frame))
    o1 = RotationEstimatorOptions(**{**vars(opt), "skip_initialization": True})
    return ra_solve(p, o1, ctx=ctx)


def convert_rotations_from_image_to_rig(R_img: np.ndarray, image_frame, image_cam, num_frames: int, num_cams: int):
    """ConvertRotationsFromImageToRig (rotation_initializer.cc:7-125) in the flat representation, where an image without
    a cam block carries a rig-level rotation already (reference sensor, or calibrated sensor folded by the caller):
      * cam block c: quaternion average over its images of R_image * R_ref(frame)^T, R_ref = the frame's first image
        without a block; frames that have none are skipped (:26-73);
      * frame f: quaternion average over its images of R_image (no block) or R_cam^T R_image (:86-121).
    Returns (R_frame [F,3,3], R_cam [C,3,3]); frames / blocks without images stay the identity."""
    from . import so3

    imf = np.asarray(image_frame, np.int64)
    imc = np.asarray(image_cam, np.int64)
    R_frame = np.tile(np.eye(3), (num_frames, 1, 1))
    R_cam = np.tile(np.eye(3), (num_cams, 1, 1))
    ref = -np.ones(num_frames, np.int64)
    for i in range(imf.shape[0] - 1, -1, -1):
        if imc[i] < 0:
            ref[imf[i]] = i
    for c in range(num_cams):
        sel = np.nonzero((imc == c) & (ref[imf] >= 0))[0]
        if sel.size:
            qs = so3.rotmat_to_quat(R_img[sel] @ np.transpose(R_img[ref[imf[sel]]], (0, 2, 1)))
            R_cam[c] = so3.quat_to_rotmat(average_quaternions(qs)[None])[0]
    for f in range(num_frames):
        sel = np.nonzero(imf == f)[0]
        if sel.size:
            Rr = np.where((imc[sel] >= 0)[:, None, None], np.transpose(R_cam[np.maximum(imc[sel], 0)], (0, 2, 1)) @ R_img[sel],
                          R_img[sel])
            R_frame[f] = so3.quat_to_rotmat(average_quaternions(so3.rotmat_to_quat(Rr))[None])[0]
    return R_frame, R_cam


def ra_solve_rigs(num_frames: int, image_frame, image_cam, num_cams: int, pair_i, pair_j, pair_q, pair_ninl,
                  pair_weight=None, options: Optional[RotationEstimatorOptions] = None, ctx=None, fixed_frame: int = 0,
                  frame_aa0=None, cam_aa0=None):
    """RotationEstimator::EstimateRotations for rigs whose cam_from_rig ROTATIONS are (partly) unknown
    (gra.cc:40-85 with the cam blocks of :173-191): image pairs as edges — relative rotations of calibrated sensors
    already folded (gra.cc:306-309), pairs inside one frame between two images without a block dropped (:300-304) —,

      * initialisation (unless skip_initialization): maximum spanning tree over the images (gra.cc:87-138, a
        zero-iteration gsfm_ra_solve), then convert_rotations_from_image_to_rig;
      * the solve: gsfm_ra_solve with the image tables (frames + cam blocks as unknowns).

    Returns (status, frame_aa [F,3], cam_aa [C,3], report)."""
    from . import so3

    ctx = ctx or default_context()
    opt = options or RotationEstimatorOptions()
    imf = np.asarray(image_frame, np.int32)
    imc = np.asarray(image_cam, np.int32)
    pi, pj = np.asarray(pair_i, np.int32), np.asarray(pair_j, np.int32)
    E = pi.shape[0]
    w = np.ones(E) if pair_weight is None else np.asarray(pair_weight, np.float64)
    ninl = np.asarray(pair_ninl, np.int32)
    q = np.asarray(pair_q, np.float64)
    aa_f = np.zeros((num_frames, 3)) if frame_aa0 is None else np.array(frame_aa0, np.float64)
    aa_c = np.zeros((num_cams, 3)) if cam_aa0 is None else np.array(cam_aa0, np.float64)
    if not opt.skip_initialization:
        img = RaProblem(int(imf.shape[0]), pi, pj, q, w, ninl, np.zeros((imf.shape[0], 3)), 0)
        o0 = RotationEstimatorOptions(**{**vars(opt), "max_num_l1_iterations": 0, "max_num_irls_iterations": 0})
        rc, rot_img, _ = ra_solve(img, o0, ctx=ctx)
        if rc != 0:
            return rc, aa_f, aa_c, {}
        R_f, R_c = convert_rotations_from_image_to_rig(so3.aa_to_rotmat(rot_img), imf, imc, num_frames, num_cams)
        aa_f = so3.quat_to_aa(so3.rotmat_to_quat(R_f))
        aa_c = so3.quat_to_aa(so3.rotmat_to_quat(R_c)) if num_cams else np.zeros((0, 3))
    p = RaProblem(int(num_frames), pi, pj, q, w, ninl, aa_f, int(fixed_frame), image_frame=imf, image_cam=imc, cam_aa0=aa_c)
    o1 = RotationEstimatorOptions(**{**vars(opt), "skip_initialization": True})
    rc, rot, rep = ra_solve(p, o1, ctx=ctx)
    return rc, rot, rep.get("cam_rot_aa", aa_c), rep


def gp_solve(p: GpProblem, options: Optional[GlobalPositionerOptions] = None, ctx=None):
    """gsfm_gp_solve.  Returns (status, cam_center [N,3], pt_xyz [P,3], report dict); with unknown cam_from_rig centres
    (p.sensor_center) the report carries the estimates as report["sensor_center"] [S,3]."""
    ctx = ctx or default_context()
    opt = (options or GlobalPositionerOptions()).to_c()
    off, oc = _h(p.pt_offset, np.int64), _h(p.obs_cam, np.int32)
    od, cal = _h(p.obs_dir, np.float64), _h(p.obs_calibrated, np.uint8)
    c = _lib.GpProblemC()
    c.mem = _mem_of(off, oc, od, cal)
    c.num_cams, c.num_pts, c.num_obs = int(p.num_cams), int(p.num_pts), int(oc.shape[0])
    c.pt_offset, c.obs_cam, c.obs_dir, c.obs_calibrated = _lib.ptr(off), _lib.ptr(oc), _lib.ptr(od), _lib.ptr(cal)
    cen = _h(p.cam_center, np.float64)
    xyz = _h(p.pt_xyz, np.float64)
    cen = cen.copy() if isinstance(cen, np.ndarray) else cen.clone()
    xyz = xyz.copy() if isinstance(xyz, np.ndarray) else xyz.clone()
    if getattr(p, "image_frame", None) is not None:  # calibrated rigs: obs_cam indexes images
        imf, imo = _h(p.image_frame, np.int32), _h(p.image_offset, np.float64)
        assert _mem_of(imf, imo) == c.mem
        c.num_images, c.image_frame, c.image_offset = int(imf.shape[0]), _lib.ptr(imf), _lib.ptr(imo)
    sens = None
    if getattr(p, "sensor_center", None) is not None:  # unknown cam_from_rig: centre blocks (host table, in/out)
        ims, imr = _h(p.image_sensor, np.int32), _h(p.image_sensor_rot, np.float64)
        assert _mem_of(ims, imr) == c.mem
        sens = np.array(p.sensor_center, dtype=np.float64, order="C", copy=True)
        c.num_sensors, c.image_sensor, c.image_sensor_rot = int(sens.shape[0]), _lib.ptr(ims), _lib.ptr(imr)
        c.sensor_center = _lib.ptr(sens)
    if getattr(p, "pair_i", None) is not None:  # camera-to-camera constraints (constraint_type != ONLY_POINTS)
        pi, pj, pd = _h(p.pair_i, np.int32), _h(p.pair_j, np.int32), _h(p.pair_dir, np.float64)
        assert _mem_of(pi, pj, pd) == c.mem
        c.num_pairs, c.pair_i, c.pair_j, c.pair_dir = int(pi.shape[0]), _lib.ptr(pi), _lib.ptr(pj), _lib.ptr(pd)
    orders = []  # host permutations: the order of the random draws
    for name in ("cam_draw_order", "pt_draw_order"):
        v = getattr(p, name, None)
        if v is not None:
            a = np.ascontiguousarray(np.asarray(v.cpu() if hasattr(v, "cpu") else v), dtype=np.int32)
            want = int(c.num_cams) if name == "cam_draw_order" else int(c.num_pts)
            if a.shape != (want,):  # the library reads exactly that many entries from a raw pointer
                raise ValueError(f"{name} must have shape ({want},), got {a.shape}")
            orders.append(a)
            setattr(c, name, a.ctypes.data)
    rep = _lib.Report()
    rc = ctx.lib.gsfm_gp_solve(ctx.handle, C.byref(c), C.byref(opt), _lib.ptr(cen), _lib.ptr(xyz), C.byref(rep))
    report = rep.as_dict()
    if sens is not None:
        report["sensor_center"] = sens
    return rc, cen, xyz, report


def ba_solve(p: BaProblem, options: Optional[BundleAdjusterOptions] = None, ctx=None):
    """gsfm_ba_solve.  Returns (status, cam_q, cam_t, pt_xyz, intr_params, report dict); with sensor blocks
    (p.sensor_cam_from_rig) the report carries their final values as report["sensor_cam_from_rig"] [S,7]."""
    ctx = ctx or default_context()
    opt = (options or BundleAdjusterOptions()).to_c()
    off, oc, oxy = _h(p.pt_offset, np.int64), _h(p.obs_cam, np.int32), _h(p.obs_xy, np.float64)
    ci, im = _h(p.cam_intr, np.int32), _h(p.intr_model, np.int32)
    c = _lib.BaProblemC()
    c.mem = _mem_of(off, oc, oxy, ci, im)
    c.num_cams, c.num_intr, c.fixed_cam = int(p.num_cams), int(p.num_intr), int(p.fixed_cam)
    c.num_pts, c.num_obs = int(p.num_pts), int(oc.shape[0])
    c.pt_offset, c.obs_cam, c.obs_xy = _lib.ptr(off), _lib.ptr(oc), _lib.ptr(oxy)
    c.cam_intr, c.intr_model = _lib.ptr(ci), _lib.ptr(im)
    if getattr(p, "image_frame", None) is not None:  # calibrated rigs: obs_cam indexes images
        imf, imc, imi = _h(p.image_frame, np.int32), _h(p.image_cam_from_rig, np.float64), _h(p.image_intr, np.int32)
        assert _mem_of(imf, imc, imi) == c.mem
        c.num_images = int(imf.shape[0])
        c.image_frame, c.image_cam_from_rig, c.image_intr = _lib.ptr(imf), _lib.ptr(imc), _lib.ptr(imi)
    sens = None
    if getattr(p, "sensor_cam_from_rig", None) is not None:
        ims = _h(p.image_sensor, np.int32)
        assert _mem_of(ims) == c.mem
        sens = np.array(p.sensor_cam_from_rig, dtype=np.float64, order="C", copy=True)  # host, in/out
        c.num_sensors, c.image_sensor, c.sensor_cam_from_rig = int(sens.shape[0]), _lib.ptr(ims), _lib.ptr(sens)
    outs = []
    for a in (p.cam_q, p.cam_t, p.pt_xyz, p.intr_params):
        a = _h(a, np.float64)
        outs.append(a.copy() if isinstance(a, np.ndarray) else a.clone())
    # doubles per intrinsics row: 8, or 16 when a camera model has more than eight parameters (gsfm_ba_problem::intr_stride)
    c.intr_stride = int(outs[3].shape[1]) if len(outs[3].shape) == 2 else 0
    rep = _lib.Report()
    rc = ctx.lib.gsfm_ba_solve(ctx.handle, C.byref(c), C.byref(opt), *[_lib.ptr(a) for a in outs], C.byref(rep))
    report = rep.as_dict()
    if sens is not None:
        report["sensor_cam_from_rig"] = sens
    return (rc, *outs, report)


# ============================================================================================
# Scene level (reference class names)
# ============================================================================================
class RotationEstimator:
    """RotationEstimator (global_rotation_averaging.h:77-141), trivial rigs, 3-DoF."""

    def __init__(self, options: RotationEstimatorOptions, ctx=None):
        self.options_ = options
        self.ctx = ctx
        self.report = None

    def EstimateRotations(self, view_graph, rigs, frames, images) -> bool:
        if self.options_.use_gravity:
            return False  # 1-DoF path not provided; reference also refuses some rigs here (gra.cc:47-58)
        # frame -> dense node index over registered frames (gra.cc:193-227); first one is the gauge
        node_of_frame: Dict[int, int] = {}
        for fid, fr in frames.items():
            if fr.is_registered:
                node_of_frame[fid] = len(node_of_frame)
        if not node_of_frame:
            return False
        N = len(node_of_frame)
        ei, ej, eq, ew, en = [], [], [], [], []
        for pair in view_graph.image_pairs.values():
            if not pair.is_valid:
                continue
            i1, i2 = images[pair.image_id1], images[pair.image_id2]
            if not (frames[i1.frame_id].is_registered and frames[i2.frame_id].is_registered):
                continue
            ei.append(node_of_frame[i1.frame_id])
            ej.append(node_of_frame[i2.frame_id])
            eq.append(pair.cam2_from_cam1.rotation)
            ew.append(pair.weight)
            en.append(pair.inlier_count())
        aa0 = np.zeros((N, 3))
        for fid, n in node_of_frame.items():
            aa0[n] = so3.quat_to_aa(np.asarray(frames[fid].rig_from_world.rotation, dtype=np.float64))
        prob = RaProblem(
            num_nodes=N,
            edge_i=np.asarray(ei, dtype=np.int32),
            edge_j=np.asarray(ej, dtype=np.int32),
            edge_q=np.asarray(eq, dtype=np.float64).reshape(-1, 4),
            edge_weight=np.asarray(ew, dtype=np.float64),
            edge_ninl=np.asarray(en, dtype=np.int32),
            node_aa0=aa0,
            fixed_node=0,
        )
        rc, rot, self.report = ra_solve(prob, self.options_, self.ctx)
        if rc != 0:
            return False
        # ConvertResults (gra.cc:774-816): rotation written, translation zeroed
        quats = so3.aa_to_quat(rot)
        for fid, n in node_of_frame.items():
            frames[fid].rig_from_world.rotation = quats[n]
            frames[fid].rig_from_world.translation = np.zeros(3)
        return True


def _pack_tracks(images, frames, tracks, min_views, need_registered, node_of_frame):
    """Track-major observation lists as both GP and BA need them (gp.cc:257-292, ba.cc:121-127)."""
    tids, off, ocam, ofeat, oimg = [], [0], [], [], []
    for tid, tr in tracks.items():
        if len(tr.observations) < min_views:
            continue
        cnt = 0
        for image_id, feat in tr.observations:
            if image_id not in images:
                continue
            im = images[image_id]
            if need_registered and not im.IsRegistered():
                continue
            if im.frame_id not in node_of_frame:
                continue
            ocam.append(node_of_frame[im.frame_id])
            ofeat.append(feat)
            oimg.append(image_id)
            cnt += 1
        tids.append(tid)
        off.append(off[-1] + cnt)
    return tids, np.asarray(off, dtype=np.int64), np.asarray(ocam, dtype=np.int32), ofeat, oimg


class GlobalPositioner:
    """GlobalPositioner (global_positioning.h:56-133), trivial rigs.  ONLY_POINTS is handled here; the constraint types with
    camera-to-camera pairs (and every rig configuration) by mapper_estimators.GlobalPositioner, to which Solve delegates
    them."""

    def __init__(self, options: GlobalPositionerOptions, ctx=None):
        self.options_ = options
        self.ctx = ctx
        self.report = None

    def GetOptions(self) -> GlobalPositionerOptions:
        return self.options_

    def Solve(self, view_graph, rigs, cameras, frames, images, tracks) -> bool:
        if self.options_.constraint_type != 0:
            from . import mapper_estimators as mest

            eng = mest.GlobalPositioner(self.options_, mest.GpuBackend(self.ctx))
            ok = eng.Solve(view_graph, rigs, cameras, frames, images, tracks)
            self.report = eng.report
            return ok
        if not images:
            return False  # gp.cc:37-40
        if not tracks:
            return False  # gp.cc:46-50 (ONLY_POINTS)
        node_of_frame = {fid: n for n, fid in enumerate(frames.keys())}
        N = len(node_of_frame)
        tids, off, ocam, ofeat, oimg = _pack_tracks(
            images, frames, tracks, self.options_.min_num_view_per_track, True, node_of_frame
        )
        M = ocam.shape[0]
        R = np.zeros((N, 3, 3))
        cen = np.zeros((N, 3))
        for fid, n in node_of_frame.items():
            rw = frames[fid].rig_from_world
            R[n] = so3.quat_to_rotmat(np.asarray(rw.rotation, dtype=np.float64))
            cen[n] = -R[n].T @ np.asarray(rw.translation, dtype=np.float64)  # CenterFromPose (rigid3d.cc:65-67)
        odir = np.zeros((M, 3))
        ocal = np.ones(M, dtype=np.uint8)
        keep = np.ones(M, dtype=bool)
        for k in range(M):
            im = images[oimg[k]]
            ray = im.features_undist[ofeat[k]]
            if np.isnan(ray).any():  # gp.cc:286-292
                keep[k] = False
                continue
            odir[k] = R[ocam[k]].T @ ray
            ocal[k] = 1 if cameras[im.camera_id].has_prior_focal_length else 0
        if not keep.all():
            cnt = np.add.reduceat(keep.astype(np.int64), off[:-1]) if M else np.zeros(0, dtype=np.int64)
            cnt[np.diff(off) == 0] = 0
            off = np.concatenate([[0], np.cumsum(cnt)]).astype(np.int64)
            ocam, odir, ocal = ocam[keep], odir[keep], ocal[keep]
        xyz = np.array([tracks[t].xyz for t in tids], dtype=np.float64).reshape(-1, 3)
        prob = GpProblem(
            num_cams=N, num_pts=len(tids), pt_offset=off, obs_cam=ocam, obs_dir=odir,
            obs_calibrated=ocal, cam_center=cen, pt_xyz=xyz,
        )
        rc, cen, xyz, self.report = gp_solve(prob, self.options_, self.ctx)
        # ConvertResults (gp.cc:562-572): t = -R c for every frame
        for fid, n in node_of_frame.items():
            frames[fid].rig_from_world.translation = -(R[n] @ cen[n])
        for t, x in zip(tids, xyz):
            tracks[t].xyz = x
            if self.options_.optimize_points and self.options_.generate_random_points:
                tracks[t].is_initialized = True
        return rc == 0


class BundleAdjuster:
    """BundleAdjuster (bundle_adjustment.h:38-82), trivial rigs."""

    def __init__(self, options: BundleAdjusterOptions, ctx=None):
        self.options_ = options
        self.ctx = ctx
        self.report = None

    def GetOptions(self) -> BundleAdjusterOptions:
        return self.options_

    def Solve(self, rigs, cameras, frames, images, tracks) -> bool:
        if not images or not tracks:
            return False  # ba.cc:17-24
        node_of_frame = {fid: n for n, fid in enumerate(frames.keys())}
        N = len(node_of_frame)
        # BA does not re-check registration (ba.cc:124-127)
        tids, off, ocam, ofeat, oimg = _pack_tracks(
            images, frames, tracks, self.options_.min_num_view_per_track, False, node_of_frame
        )
        M = ocam.shape[0]
        intr_of_cam = {cid: k for k, cid in enumerate(cameras.keys())}
        K = len(intr_of_cam)
        cam_intr = np.zeros(N, dtype=np.int32)
        for im in images.values():
            if im.frame_id in node_of_frame:
                cam_intr[node_of_frame[im.frame_id]] = intr_of_cam[im.camera_id]
        oxy = np.zeros((M, 2))
        for k in range(M):
            oxy[k] = images[oimg[k]].features[ofeat[k]]
        q = np.zeros((N, 4))
        t = np.zeros((N, 3))
        for fid, n in node_of_frame.items():
            q[n] = frames[fid].rig_from_world.rotation
            t[n] = frames[fid].rig_from_world.translation
        model = np.zeros(K, dtype=np.int32)
        width = CAMERA_MAX_PARAMS_WIDE if any(len(c.params) > CAMERA_MAX_PARAMS for c in cameras.values()) else CAMERA_MAX_PARAMS
        params = np.zeros((K, width))  # 16-wide rows when a camera model has more than 8 parameters
        for cid, k in intr_of_cam.items():
            model[k] = cameras[cid].model_id
            params[k, : len(cameras[cid].params)] = cameras[cid].params
        xyz = np.array([tracks[tid].xyz for tid in tids], dtype=np.float64).reshape(-1, 3)
        # first frame that owns a residual is the constant one (ba.cc:253-269)
        seen = set(ocam.tolist())
        fixed = next((n for n in range(N) if n in seen), -1)
        prob = BaProblem(
            num_cams=N, num_pts=len(tids), num_intr=K, pt_offset=off, obs_cam=ocam, obs_xy=oxy,
            cam_intr=cam_intr, cam_q=q, cam_t=t, pt_xyz=xyz, intr_model=model, intr_params=params,
            fixed_cam=fixed,
        )
        rc, q, t, xyz, params, self.report = ba_solve(prob, self.options_, self.ctx)
        for fid, n in node_of_frame.items():
            frames[fid].rig_from_world.rotation = q[n]
            frames[fid].rig_from_world.translation = t[n]
        for tid, x in zip(tids, xyz):
            tracks[tid].xyz = x
        for cid, k in intr_of_cam.items():
            cameras[cid].params = params[k, : len(cameras[cid].params)].copy()
        return rc == 0
