"""Host-side mirror of the reference's rotation-averaging entry points for ALL rig / gravity configurations:

  RotationEstimator.EstimateRotations   glomap/estimators/global_rotation_averaging.cc:40-85 (+ :141-477 for what goes where)
  ConvertRotationsFromImageToRig        glomap/estimators/rotation_initializer.cc:7-125
  KeepLargestConnectedComponents        glomap/scene/view_graph.cc:56-97
  SolveRotationAveraging                glomap/controllers/rotation_averager.cc:8-198  (stratified 1-DoF pre-solve for
                                        gravity, trivial-rig pre-pass for unknown cam_from_rig)

on the scene containers of glomap_amd.scene.  Everything numerical goes through a *backend* with two calls —
`ra_solve(RaProblem, RotationEstimatorOptions)` and `keep_largest_cc(...)` —, by default the C ABI (gsfm_ra_solve with its
image tables / node_gravity, gsfm_keep_largest_connected_component).  The scene <-> flat logic here is the same the C++
adapter (include/gsfm_glomap_adapter.hpp) carries; tests/test_rotation_averager_policy.py runs it on the CPU with the
oracle as backend (the flat calls themselves are pinned to the oracle on the GPU by tests/test_ra_*.py).

`estimators.RotationEstimator` stays the minimal trivial-rig class the pipeline tests use; this module is the complete one.
"""
from __future__ import annotations

from typing import Dict, List, Optional

import numpy as np

from . import so3
from .flat import RaProblem
from .scene import Frame, Image, Rigid3d, ViewGraph


# ---------------------------------------------------------------------------------------------
# backend
# ---------------------------------------------------------------------------------------------
class GpuBackend:
    """The product path: libgsfm through glomap_amd.estimators / glomap_amd.tracks."""

    def __init__(self, ctx=None):
        self.ctx = ctx

    def ra_solve(self, p: RaProblem, opt):
        from . import estimators

        return estimators.ra_solve(p, opt, ctx=self.ctx)

    def keep_largest_cc(self, num_nodes, edge_i, edge_j, edge_valid, node_num_images):
        from . import tracks

        return tracks.KeepLargestConnectedComponents(num_nodes, edge_i, edge_j, edge_valid, node_num_images, ctx=self.ctx)


# ---------------------------------------------------------------------------------------------
# small scene helpers (image.h:55-99)
# ---------------------------------------------------------------------------------------------
def _rig_of(image: Image, frames, rigs):
    fr = frames[image.frame_id]
    return None if fr.rig_id is None else rigs.get(fr.rig_id)


def has_trivial_frame(image: Image, frames, rigs) -> bool:
    rig = _rig_of(image, frames, rigs)
    return rig is None or rig.IsRefSensor(image.camera_id)


def is_registered(image: Image, frames) -> bool:
    return image.frame_id in frames and frames[image.frame_id].is_registered and image.is_registered


def image_has_gravity(image: Image, frames, rigs) -> bool:
    """Image::HasGravity (image.h:78-84)."""
    fr = frames[image.frame_id]
    if not fr.HasGravity():
        return False
    return has_trivial_frame(image, frames, rigs) or _rig_of(image, frames, rigs).MaybeSensorFromRig(image.camera_id) is not None


def _cam_from_rig_state(image: Image, frames, rigs):
    """(state, q wxyz): 0 known (identity for reference sensors), 1 value with NaN translation, 2 no value."""
    ident = np.array([1.0, 0.0, 0.0, 0.0])
    if has_trivial_frame(image, frames, rigs):
        return 0, ident
    cfr = _rig_of(image, frames, rigs).MaybeSensorFromRig(image.camera_id)
    if cfr is None:
        return 2, ident
    q = np.asarray(cfr.rotation, dtype=np.float64)
    return (1 if np.isnan(np.asarray(cfr.translation, dtype=np.float64)).any() else 0), q


def _avg_quat(qs: np.ndarray) -> np.ndarray:
    """colmap::AverageQuaternions, unit weights."""
    qs = np.asarray(qs, dtype=np.float64).reshape(-1, 4)
    w, v = np.linalg.eigh(qs.T @ qs)
    out = v[:, -1]
    return out if out[0] >= 0 else -out


# ---------------------------------------------------------------------------------------------
# KeepLargestConnectedComponents (view_graph.cc:56-97)
# ---------------------------------------------------------------------------------------------
def KeepLargestConnectedComponents(view_graph: ViewGraph, frames: Dict[int, Frame], images: Dict[int, Image], backend) -> int:
    """Frames are the nodes, valid pairs the edges; the largest component (by frame count) stays registered, pairs that
    touch an unregistered image become invalid.  Returns the number of registered images."""
    fids = list(frames.keys())
    node = {f: n for n, f in enumerate(fids)}
    keys = list(view_graph.image_pairs.keys())
    if not keys:
        return 0
    ei = np.array([node[images[view_graph.image_pairs[k].image_id1].frame_id] for k in keys], dtype=np.int32)
    ej = np.array([node[images[view_graph.image_pairs[k].image_id2].frame_id] for k in keys], dtype=np.int32)
    valid = np.array([1 if view_graph.image_pairs[k].is_valid else 0 for k in keys], dtype=np.uint8)
    nimg = np.zeros(len(fids), dtype=np.int32)
    for im in images.values():
        if im.frame_id in node:
            nimg[node[im.frame_id]] += 1
    node_reg, edge_keep, num = backend.keep_largest_cc(len(fids), ei, ej, valid, nimg)
    if num == 0:
        return 0
    for n, f in enumerate(fids):
        frames[f].is_registered = bool(node_reg[n])
    for k, keep in zip(keys, edge_keep):
        if not keep:
            view_graph.image_pairs[k].is_valid = False
    return int(num)


# ---------------------------------------------------------------------------------------------
# ConvertRotationsFromImageToRig (rotation_initializer.cc:7-125)
# ---------------------------------------------------------------------------------------------
def ConvertRotationsFromImageToRig(cam_from_worlds: Dict[int, np.ndarray], images, rigs, frames) -> bool:
    """cam_from_worlds: image_id -> quaternion (w,x,y,z).  Sensors WITHOUT a cam_from_rig get the quaternion average of
    R_image R_refimage^T over the frames that have a reference image (NaN translation); then every frame gets the average
    over its images of R_image (reference image) / R_cam_from_rig^T R_image (others with a cam_from_rig).  Frames none of
    whose images were estimated keep their pose."""
    nan3 = np.full(3, np.nan)
    cam_to_rig = {}
    for rid, rig in rigs.items():
        for cam_id, _ in rig.NonRefSensors():
            cam_to_rig[cam_id] = rid
    frame_ref: Dict[int, int] = {}
    acc: Dict[int, List[np.ndarray]] = {}
    for fid, fr in frames.items():
        rig = None if fr.rig_id is None else rigs.get(fr.rig_id)
        ref_img = None
        for iid in fr.image_ids:
            if iid in images and is_registered(images[iid], frames) and (rig is None or images[iid].camera_id == rig.ref_camera_id):
                ref_img = iid
                break
        if ref_img is None or ref_img not in cam_from_worlds:
            continue
        frame_ref[fid] = ref_img
        if rig is None:
            continue
        R_ref = so3.quat_to_rotmat(cam_from_worlds[ref_img][None])[0]
        for iid in fr.image_ids:
            if iid not in images or not is_registered(images[iid], frames):
                continue
            im = images[iid]
            if im.camera_id == rig.ref_camera_id or rig.MaybeSensorFromRig(im.camera_id) is not None:
                continue
            if iid not in cam_from_worlds:
                continue
            R = so3.quat_to_rotmat(cam_from_worlds[iid][None])[0]
            acc.setdefault(im.camera_id, []).append(so3.rotmat_to_quat((R @ R_ref.T)[None])[0])
    for cam_id, qs in acc.items():
        rigs[cam_to_rig[cam_id]].SetSensorFromRig(cam_id, Rigid3d(_avg_quat(np.array(qs)), nan3.copy()))
    for fid, fr in frames.items():
        rig = None if fr.rig_id is None else rigs.get(fr.rig_id)
        qs = []
        for iid in fr.image_ids:
            if iid not in images or not is_registered(images[iid], frames) or iid not in cam_from_worlds:
                continue
            im = images[iid]
            R = so3.quat_to_rotmat(cam_from_worlds[iid][None])[0]
            if frame_ref.get(fid) == iid or rig is None:
                qs.append(so3.rotmat_to_quat(R[None])[0])
            else:
                cfr = rig.MaybeSensorFromRig(im.camera_id)
                if cfr is None:
                    continue
                Rc = so3.quat_to_rotmat(np.asarray(cfr.rotation, dtype=np.float64)[None])[0]
                qs.append(so3.rotmat_to_quat((Rc.T @ R)[None])[0])
        if qs:
            fr.rig_from_world = Rigid3d(_avg_quat(np.array(qs)), nan3.copy())
    return True


# ---------------------------------------------------------------------------------------------
# RotationEstimator, every configuration
# ---------------------------------------------------------------------------------------------
class RotationEstimator:
    """RotationEstimator (global_rotation_averaging.h:77-141): trivial rigs, calibrated rigs (cam_from_rig folded into
    the relative rotations, gra.cc:306-309), uncalibrated sensors (cam blocks, gra.cc:173-191) and gravity-aligned
    frames (use_gravity, gra.cc:207-217, 315-327)."""

    def __init__(self, options, backend=None):
        self.options_ = options
        self.backend = backend or GpuBackend()
        self.report = None

    # -- helpers -----------------------------------------------------------------------------
    def _solve(self, p: RaProblem, **overrides):
        from .estimators import RotationEstimatorOptions

        opt = RotationEstimatorOptions(**{**vars(self.options_), **overrides}) if overrides else self.options_
        return self.backend.ra_solve(p, opt)

    def EstimateRotations(self, view_graph: ViewGraph, rigs, frames, images) -> bool:
        o = self.options_
        if o.use_gravity:  # gra.cc:47-58
            for rig in rigs.values():
                for _, cfr in rig.NonRefSensors():
                    if cfr is None:
                        return False
        fids = [f for f, fr in frames.items() if fr.is_registered]  # gra.cc:193-227, map order
        if not fids:
            return False
        node = {f: n for n, f in enumerate(fids)}
        N = len(fids)
        pairs = [p for p in view_graph.image_pairs.values()
                 if p.is_valid and is_registered(images[p.image_id1], frames) and is_registered(images[p.image_id2], frames)]
        state = {}
        for p in pairs:
            for iid in (p.image_id1, p.image_id2):
                if iid not in state:
                    state[iid] = _cam_from_rig_state(images[iid], frames, rigs)
        unknown = (not o.use_gravity) and any(st != 0 for st, _ in state.values())
        if unknown:
            return self._estimate_with_cam_blocks(pairs, state, rigs, frames, images, fids, node)

        grav = np.zeros(N, dtype=np.uint8)
        R_align = {}
        aa0 = np.zeros((N, 3))
        first_gravity = -1
        for f, n in node.items():
            fr = frames[f]
            q = np.asarray(fr.rig_from_world.rotation, dtype=np.float64)
            if o.use_gravity and fr.HasGravity():
                Ra = fr.GetRAlign()
                R_align[f] = Ra
                # RotUpToAngle(R_align^T R_rig_from_world), gra.cc:207-211
                aa = so3.quat_to_aa(so3.rotmat_to_quat((Ra.T @ so3.quat_to_rotmat(q[None])[0])[None]))[0]
                aa0[n] = (0.0, aa[1], 0.0)
                grav[n] = 1
                if first_gravity < 0:
                    first_gravity = n  # gra.cc:212-216: the first gravity frame takes the gauge
            else:
                aa0[n] = so3.quat_to_aa(q[None])[0]
        rigged = any(not has_trivial_frame(images[i], frames, rigs) for i in state)
        ei, ej, eq, ew, en = [], [], [], [], []
        ii, ij, iq, iw, in_ = [], [], [], [], []  # image-level copy for the spanning-tree start of calibrated rigs
        img_idx: Dict[int, int] = {}

        def image_index(iid):
            return img_idx.setdefault(iid, len(img_idx))

        for p in pairs:
            i1, i2 = images[p.image_id1], images[p.image_id2]
            R21 = so3.quat_to_rotmat(np.asarray(p.cam2_from_cam1.rotation, dtype=np.float64)[None])[0]
            R_rel = R21
            if rigged:
                ii.append(image_index(p.image_id1))
                ij.append(image_index(p.image_id2))
                iq.append(np.asarray(p.cam2_from_cam1.rotation, dtype=np.float64))
                iw.append(p.weight)
                in_.append(p.inlier_count())
                if i1.frame_id == i2.frame_id:
                    continue  # gra.cc:300-304
                R1 = so3.quat_to_rotmat(state[p.image_id1][1][None])[0]
                R2 = so3.quat_to_rotmat(state[p.image_id2][1][None])[0]
                R_rel = R2.T @ R21 @ R1  # gra.cc:306-309
            if o.use_gravity:  # gra.cc:315-327
                if image_has_gravity(i1, frames, rigs):
                    R_rel = R_rel @ R_align[i1.frame_id]
                if image_has_gravity(i2, frames, rigs):
                    R_rel = R_align[i2.frame_id].T @ R_rel
            ei.append(node[i1.frame_id])
            ej.append(node[i2.frame_id])
            eq.append(so3.rotmat_to_quat(R_rel[None])[0])
            ew.append(p.weight)
            en.append(p.inlier_count())
        skip_init = bool(o.skip_initialization) or bool(o.use_gravity)  # gra.cc:60-62
        if rigged and not skip_init and img_idx:
            # spanning tree over the images, then rig_from_world = average of cam_from_rig^-1 cam_from_world (:86-121)
            NI = len(img_idx)
            img = RaProblem(NI, np.asarray(ii, np.int32), np.asarray(ij, np.int32), np.asarray(iq, np.float64).reshape(-1, 4),
                            np.asarray(iw, np.float64), np.asarray(in_, np.int32), np.zeros((NI, 3)), 0)
            rc, rot_img, _ = self._solve(img, max_num_l1_iterations=0, max_num_irls_iterations=0, skip_initialization=False)
            if rc != 0:
                return False
            R_img = so3.aa_to_rotmat(rot_img)
            per_frame: Dict[int, List[np.ndarray]] = {}
            for iid, k in img_idx.items():
                Rc = so3.quat_to_rotmat(state[iid][1][None])[0]
                per_frame.setdefault(node[images[iid].frame_id], []).append(so3.rotmat_to_quat((Rc.T @ R_img[k])[None])[0])
            for n, qs in per_frame.items():
                aa0[n] = so3.quat_to_aa(_avg_quat(np.array(qs))[None])[0]
            skip_init = True
        prob = RaProblem(N, np.asarray(ei, np.int32), np.asarray(ej, np.int32), np.asarray(eq, np.float64).reshape(-1, 4),
                         np.asarray(ew, np.float64), np.asarray(en, np.int32), aa0, first_gravity if first_gravity >= 0 else 0,
                         node_gravity=grav if first_gravity >= 0 else None)
        rc, rot, self.report = self._solve(prob, skip_initialization=skip_init)
        if rc != 0:
            return False
        quats = so3.aa_to_quat(rot)
        for f, n in node.items():  # ConvertResults (gra.cc:774-799)
            q = quats[n]
            if grav[n]:
                q = so3.rotmat_to_quat((R_align[f] @ so3.quat_to_rotmat(q[None])[0])[None])[0]
            frames[f].rig_from_world = Rigid3d(q, np.zeros(3))
        return True

    def _estimate_with_cam_blocks(self, pairs, state, rigs, frames, images, fids, node) -> bool:
        o = self.options_
        N = len(fids)
        img_idx: Dict[int, int] = {}
        image_frame, image_cam, image_ref, fold = [], [], [], []
        cam_of: Dict[int, int] = {}
        cam_ids, cam_has_start, cam_aa = [], [], []

        def image_index(iid):
            if iid in img_idx:
                return img_idx[iid]
            im = images[iid]
            st, q = state[iid]
            block = -1
            if st != 0:
                if im.camera_id not in cam_of:
                    cam_of[im.camera_id] = len(cam_ids)
                    cam_ids.append(im.camera_id)
                    cam_has_start.append(st == 1)
                    cam_aa.append(so3.quat_to_aa(q[None])[0] if st == 1 else np.zeros(3))  # gra.cc:231-242
                block = cam_of[im.camera_id]
            img_idx[iid] = len(image_frame)
            image_frame.append(node[im.frame_id])
            image_cam.append(block)
            image_ref.append(has_trivial_frame(im, frames, rigs))
            fold.append(q if st == 0 else np.array([1.0, 0.0, 0.0, 0.0]))
            return img_idx[iid]

        ii, ij, iq, iw, in_ = [], [], [], [], []
        for p in pairs:
            a, b = image_index(p.image_id1), image_index(p.image_id2)
            if image_frame[a] == image_frame[b] and image_cam[a] < 0 and image_cam[b] < 0:
                continue  # gra.cc:300-304
            R21 = so3.quat_to_rotmat(np.asarray(p.cam2_from_cam1.rotation, dtype=np.float64)[None])[0]
            R_rel = so3.quat_to_rotmat(fold[b][None])[0].T @ R21 @ so3.quat_to_rotmat(fold[a][None])[0]
            ii.append(a)
            ij.append(b)
            iq.append(so3.rotmat_to_quat(R_rel[None])[0])
            iw.append(p.weight)
            in_.append(p.inlier_count())
        NI, C = len(image_frame), len(cam_ids)
        if NI == 0:
            return False
        aa_f = np.zeros((N, 3))
        for f, n in node.items():
            aa_f[n] = so3.quat_to_aa(np.asarray(frames[f].rig_from_world.rotation, dtype=np.float64)[None])[0]
        aa_c = np.array(cam_aa, dtype=np.float64).reshape(C, 3)
        imf, imc = np.asarray(image_frame, np.int32), np.asarray(image_cam, np.int32)
        edges = (np.asarray(ii, np.int32), np.asarray(ij, np.int32), np.asarray(iq, np.float64).reshape(-1, 4),
                 np.asarray(iw, np.float64), np.asarray(in_, np.int32))
        if not o.skip_initialization:
            rc, rot_img, _ = self._solve(RaProblem(NI, *edges, np.zeros((NI, 3)), 0), max_num_l1_iterations=0,
                                         max_num_irls_iterations=0)
            if rc != 0:
                return False
            R_img = so3.aa_to_rotmat(rot_img)
            ref = -np.ones(N, dtype=np.int64)
            for i in range(NI - 1, -1, -1):
                if image_ref[i]:
                    ref[imf[i]] = i
            R_cam = so3.aa_to_rotmat(aa_c) if C else np.zeros((0, 3, 3))
            for c in range(C):
                if cam_has_start[c]:
                    continue  # a stored rotation is kept (rotation_initializer.cc:52-57)
                sel = [i for i in range(NI) if imc[i] == c and ref[imf[i]] >= 0]
                if sel:
                    qs = so3.rotmat_to_quat(R_img[sel] @ np.transpose(R_img[ref[imf[sel]]], (0, 2, 1)))
                    R_cam[c] = so3.quat_to_rotmat(_avg_quat(qs)[None])[0]
            for n in range(N):
                sel = np.nonzero(imf == n)[0]
                if sel.size:
                    Rr = np.where((imc[sel] >= 0)[:, None, None], np.transpose(R_cam[np.maximum(imc[sel], 0)], (0, 2, 1)) @ R_img[sel],
                                  R_img[sel]) if C else R_img[sel]
                    aa_f[n] = so3.quat_to_aa(_avg_quat(so3.rotmat_to_quat(Rr))[None])[0]
            if C:
                aa_c = so3.quat_to_aa(so3.rotmat_to_quat(R_cam))
        prob = RaProblem(N, *edges, aa_f, 0, image_frame=imf, image_cam=imc, cam_aa0=aa_c)
        rc, rot, self.report = self._solve(prob, skip_initialization=True)
        if rc != 0:
            return False
        quats = so3.aa_to_quat(rot)
        for f, n in node.items():
            frames[f].rig_from_world = Rigid3d(quats[n], np.zeros(3))
        cams = np.asarray(self.report.get("cam_rot_aa", aa_c), dtype=np.float64).reshape(C, 3)
        cq = so3.aa_to_quat(cams) if C else np.zeros((0, 4))
        cam_to_rig = {cam_id: rid for rid, rig in rigs.items() for cam_id, _ in rig.NonRefSensors()}
        for c, cam_id in enumerate(cam_ids):  # gra.cc:801-815: rotation set, "no translation yet"
            rigs[cam_to_rig[cam_id]].SetSensorFromRig(cam_id, Rigid3d(cq[c], np.full(3, np.nan)))
        return True


# ---------------------------------------------------------------------------------------------
# SolveRotationAveraging (controllers/rotation_averager.cc:8-198)
# ---------------------------------------------------------------------------------------------
def SolveRotationAveraging(view_graph: ViewGraph, rigs, frames, images, options, use_stratified: bool = True,
                           backend=None) -> bool:
    """The policy around the estimator: largest component; with use_gravity and use_stratified a 1-DoF solve of the pairs
    whose two images both have gravity first (unless there are none, or more than 95 % are: then one mixed solve does);
    for sensors without cam_from_rig a pre-pass in which every such image is a trivial frame of its own, followed by
    ConvertRotationsFromImageToRig and the real solve with skip_initialization."""
    import copy

    from .estimators import RotationEstimatorOptions
    from .scene import Rig

    backend = backend or GpuBackend()
    KeepLargestConnectedComponents(view_graph, frames, images, backend)
    solve_1dof = bool(options.use_gravity) and use_stratified
    grav_graph = ViewGraph()
    total_pairs = 0
    if solve_1dof:
        for key, p in view_graph.image_pairs.items():
            if not p.is_valid:
                continue
            i1, i2 = images[p.image_id1], images[p.image_id2]
            if not (is_registered(i1, frames) and is_registered(i2, frames)):
                continue
            total_pairs += 1
            if image_has_gravity(i1, frames, rigs) and image_has_gravity(i2, frames, rigs):
                q = copy.deepcopy(p)
                q.is_valid, q.weight = True, -1.0  # ImagePair(id1, id2, cam2_from_cam1): defaults otherwise (:34-38)
                grav_graph.image_pairs[key] = q
    grav_pairs = len(grav_graph.image_pairs)
    solve_1dof = solve_1dof and not (grav_pairs == 0 or grav_pairs > total_pairs * 0.95)
    if solve_1dof:
        KeepLargestConnectedComponents(grav_graph, frames, images, backend)
        if not RotationEstimator(options, backend).EstimateRotations(grav_graph, rigs, frames, images):
            return False
        KeepLargestConnectedComponents(view_graph, frames, images, backend)
    unknown_cams = set()
    for rig in rigs.values():
        for cam_id, cfr in rig.NonRefSensors():
            if cfr is None:
                unknown_cams.add(cam_id)
    if unknown_cams and not options.skip_initialization:
        # trivial rotation averaging for the cameras without cam_from_rig (:66-172)
        rigs_t: Dict[int, Rig] = {}
        max_rig = max(rigs.keys(), default=0)
        for rid, rig in rigs.items():
            rigs_t[rid] = Rig(rid, rig.ref_camera_id, {c: copy.deepcopy(v) for c, v in rig.NonRefSensors() if v is not None})
        cam_rig = {}
        for cam_id in sorted(unknown_cams):
            max_rig += 1
            rigs_t[max_rig] = Rig(max_rig, cam_id, {})
            cam_rig[cam_id] = max_rig
        frames_t: Dict[int, Frame] = {}
        images_t: Dict[int, Image] = {}
        next_frame = max(frames.keys(), default=0) + 1
        for fid, fr in frames.items():
            frames_t[fid] = Frame(fid, Rigid3d(), fr.is_registered, fr.rig_id if fr.rig_id in rigs_t else None, [])
            for iid in fr.image_ids:
                im = images[iid]
                if not is_registered(im, frames):
                    continue
                it = Image(iid, im.camera_id, fid)
                images_t[iid] = it
                if im.camera_id not in unknown_cams:
                    frames_t[fid].image_ids.append(iid)
                else:  # a trivial frame of its own (CreateFrameForImage, colmap_converter.cc:440-462)
                    frames_t[next_frame] = Frame(next_frame, Rigid3d(), True, cam_rig[im.camera_id], [iid])
                    it.frame_id = next_frame
                    next_frame += 1
        KeepLargestConnectedComponents(view_graph, frames_t, images_t, backend)
        RotationEstimator(options, backend).EstimateRotations(view_graph, rigs_t, frames_t, images_t)
        cams_from_world = {}
        for iid, it in images_t.items():
            if not is_registered(it, frames_t):
                continue
            st, q = _cam_from_rig_state(it, frames_t, rigs_t)
            Rf = so3.quat_to_rotmat(np.asarray(frames_t[it.frame_id].rig_from_world.rotation, dtype=np.float64)[None])[0]
            Rc = so3.quat_to_rotmat(q[None])[0]
            cams_from_world[iid] = so3.rotmat_to_quat((Rc @ Rf)[None])[0]  # image.CamFromWorld()
        ConvertRotationsFromImageToRig(cams_from_world, images, rigs, frames)
        o2 = RotationEstimatorOptions(**{**vars(options), "skip_initialization": True})
        ok = RotationEstimator(o2, backend).EstimateRotations(view_graph, rigs, frames, images)
        KeepLargestConnectedComponents(view_graph, frames, images, backend)
        return ok
    o2 = options
    if unknown_cams:  # (:179-182) the estimator must build its own start then
        o2 = RotationEstimatorOptions(**{**vars(options), "skip_initialization": False})
    ok = RotationEstimator(o2, backend).EstimateRotations(view_graph, rigs, frames, images)
    KeepLargestConnectedComponents(view_graph, frames, images, backend)
    return ok
