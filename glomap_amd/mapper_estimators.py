"""Scene-level GlobalPositioner and BundleAdjuster for every rig configuration — the Python counterpart of the two
classes in include/gsfm_glomap_adapter.hpp, next to rotation_averager.RotationEstimator:

  GlobalPositioner.Solve   glomap/estimators/global_positioning.cc:28-93: trivial frames (BATA), calibrated rigs (RigBATA,
                           :318-350), sensors whose cam_from_rig translation is NaN (RigUnknownBATA, :354-368); the constraint
                           types with camera-to-camera pairs (:167-210; trivial frames only)
  BundleAdjuster.Solve     glomap/estimators/bundle_adjustment.cc:11-106: trivial frames, calibrated rigs (:147-160),
                           optimize_rig_poses (:161-179)

They flatten the containers of glomap_amd.scene into flat.GpProblem / flat.BaProblem (image tables, sensor blocks), call a
backend — by default the C ABI through glomap_amd.estimators.gp_solve / ba_solve — and write the results back in place.
tests/test_mapper_rigs_cpu.py runs RA -> GP -> BA on rig scenes with the oracle as backend (the reference's two rig
mapper tests, global_mapper_test.cc:89-175); the flat calls themselves are pinned to the oracle on the GPU by
tests/test_rigs.py.  `estimators.GlobalPositioner / BundleAdjuster` stay the minimal trivial-rig classes."""
from __future__ import annotations

from typing import Dict, List

import numpy as np

from . import so3
from .flat import CAMERA_MAX_PARAMS, CAMERA_MAX_PARAMS_WIDE, BaProblem, GpProblem
from .rotation_averager import _cam_from_rig_state, has_trivial_frame, is_registered
from .scene import Rigid3d


class GpuBackend:
    def __init__(self, ctx=None):
        self.ctx = ctx

    def gp_solve(self, p: GpProblem, opt):
        from . import estimators

        return estimators.gp_solve(p, opt, ctx=self.ctx)

    def ba_solve(self, p: BaProblem, opt):
        from . import estimators

        return estimators.ba_solve(p, opt, ctx=self.ctx)


def _R(q) -> np.ndarray:
    return so3.quat_to_rotmat(np.asarray(q, dtype=np.float64)[None])[0]


def _q(R) -> np.ndarray:
    return so3.rotmat_to_quat(np.asarray(R)[None])[0]


def _pack_tracks(images, frames, tracks, node, min_views, keep, keep_empty=False):
    """Track-major observations; `min_views` is tested on the RAW observation count (gp.cc:258, ba.cc:122).  keep_empty
    (global positioning): a track that passes the raw count without one usable observation stays as a zero-length track —
    the reference still draws its random start and marks it initialised (gp.cc:258-264)."""
    tids, off, oimg, ofeat = [], [0], [], []
    for tid, tr in tracks.items():
        if len(tr.observations) < min_views:
            continue
        n0 = len(oimg)
        for iid, feat in tr.observations:
            if iid not in images or images[iid].frame_id not in node or not keep(images[iid], feat):
                continue
            oimg.append(iid)
            ofeat.append(feat)
        if len(oimg) == n0 and not keep_empty:
            continue
        tids.append(tid)
        off.append(len(oimg))
    return tids, np.asarray(off, dtype=np.int64), oimg, ofeat


class GlobalPositioner:
    def __init__(self, options, backend=None):
        self.options_ = options
        self.backend = backend or GpuBackend()
        self.report = None

    def GetOptions(self):
        return self.options_

    def Solve(self, view_graph, rigs, cameras, frames, images, tracks) -> bool:
        from .estimators import GlobalPositionerOptions

        o = self.options_
        ctype = int(o.constraint_type)
        pairs_in = getattr(view_graph, "image_pairs", None) or {}
        if not images or (not pairs_in and ctype != 0) or (not tracks and ctype != 1):
            return False  # gp.cc:37-50
        fids = list(frames.keys())  # every frame: ConvertResults rewrites them all (gp.cc:566-572)
        node = {f: n for n, f in enumerate(fids)}
        N = len(fids)

        def keep(im, feat):  # gp.cc:279-292
            return is_registered(im, frames) and not np.isnan(im.features_undist[feat]).any()

        tids, off, oimg, ofeat = _pack_tracks(images, frames, tracks, node, o.min_num_view_per_track, keep, keep_empty=True)
        if (not tids or not oimg) and ctype != 1:
            return False
        rigged = any(not has_trivial_frame(images[i], frames, rigs) for i in set(oimg))
        # AddCameraToCameraConstraints (gp.cc:167-210): one pair per valid image pair whose two images are known, in the
        # view graph's iteration order (the first one's scale is the constant one, gp.cc:484-489)
        pair_i, pair_j, pair_dir = [], [], []
        if ctype != 0:
            if rigged or any(is_registered(im, frames) and not has_trivial_frame(im, frames, rigs) for im in images.values()):
                return False  # "only trivial frames are supported for the camera to camera constraints" (gp.cc:169-176)
            for pair in pairs_in.values():
                if not pair.is_valid or pair.image_id1 not in images or pair.image_id2 not in images:
                    continue
                im1, im2 = images[pair.image_id1], images[pair.image_id2]
                R_cw2 = _R(frames[im2.frame_id].rig_from_world.rotation)  # trivial frame: cam_from_world = rig_from_world
                pair_i.append(node[im1.frame_id])
                pair_j.append(node[im2.frame_id])
                pair_dir.append(-(R_cw2.T @ np.asarray(pair.cam2_from_cam1.translation, dtype=np.float64)))  # gp.cc:195-197
            if not pair_i:
                return False
        R_f = {f: _R(frames[f].rig_from_world.rotation) for f in fids}
        img_idx: Dict[int, int] = {}
        image_frame, image_offset, image_rot, image_key, image_state = [], [], [], [], []
        M = len(oimg)
        dirs = np.zeros((M, 3))
        cal = np.zeros(M, dtype=np.uint8)
        obs_cam = np.zeros(M, dtype=np.int32)
        for k, (iid, feat) in enumerate(zip(oimg, ofeat)):
            im = images[iid]
            st, q = _cam_from_rig_state(im, frames, rigs)
            if st == 2:
                return False  # no cam_from_rig at all: the reference dereferences the empty optional (gp.cc:323)
            R_cw = _R(q) @ R_f[im.frame_id]
            dirs[k] = R_cw.T @ np.asarray(im.features_undist[feat], dtype=np.float64)  # gp.cc:294-296
            cal[k] = 1 if cameras[im.camera_id].has_prior_focal_length else 0  # gp.cc:313-316
            if not rigged:
                obs_cam[k] = node[im.frame_id]
                continue
            if iid not in img_idx:
                img_idx[iid] = len(image_frame)
                image_frame.append(node[im.frame_id])
                t_cfr = np.zeros(3)
                if st == 0 and not has_trivial_frame(im, frames, rigs):
                    t_cfr = np.asarray(rigs[frames[im.frame_id].rig_id].MaybeSensorFromRig(im.camera_id).translation, dtype=np.float64)
                image_offset.append(R_cw.T @ t_cfr)  # translation_rig, gp.cc:329-333 (zero for the unknown ones)
                image_rot.append(R_f[im.frame_id])
                image_key.append((frames[im.frame_id].rig_id, im.camera_id))
                image_state.append(st)
            obs_cam[k] = img_idx[iid]
        # centre blocks in the order ParameterizeVariables draws their start values (gp.cc:442-456)
        wanted = {image_key[i] for i in range(len(image_key)) if image_state[i] == 1}
        sensor_ids = [(rid, cam_id) for rid, rig in rigs.items() for cam_id, _ in rig.NonRefSensors() if (rid, cam_id) in wanted]
        block = {key: b for b, key in enumerate(sensor_ids)}
        cen = np.zeros((N, 3))
        for f, n in node.items():  # c = -R^T t
            cen[n] = -R_f[f].T @ np.asarray(frames[f].rig_from_world.translation, dtype=np.float64)
        xyz = np.array([tracks[t].xyz for t in tids], dtype=np.float64).reshape(-1, 3)
        p = GpProblem(num_cams=N, num_pts=len(tids), pt_offset=off, obs_cam=obs_cam, obs_dir=dirs, obs_calibrated=cal,
                      cam_center=cen, pt_xyz=xyz)
        if rigged:
            p.image_frame = np.asarray(image_frame, np.int32)
            p.image_offset = np.asarray(image_offset, np.float64).reshape(-1, 3)
            if sensor_ids:
                p.image_sensor = np.array([block[image_key[i]] if image_state[i] == 1 else -1 for i in range(len(image_key))], np.int32)
                p.image_sensor_rot = np.asarray(image_rot, np.float64).reshape(-1, 3, 3)
                p.sensor_center = np.zeros((len(sensor_ids), 3))
        # the raw-count rule is applied above; with 0 every packed track — the zero-length ones too — takes its random draw
        opt = GlobalPositionerOptions(**{**vars(o), "min_num_view_per_track": 0})
        if pair_i:
            p.pair_i, p.pair_j = np.asarray(pair_i, np.int32), np.asarray(pair_j, np.int32)
            p.pair_dir = np.asarray(pair_dir, np.float64).reshape(-1, 3)
            # POINTS_AND_CAMERAS_BALANCED weighs the point losses by reweight * #pairs / tracks.size() — every track, not
            # only the packed ones (gp.cc:220-233); the library divides by the tracks it is given
            if ctype == 2 and len(tracks) > 0:
                opt.constraint_reweight_scale = o.constraint_reweight_scale * len(tids) / len(tracks)
        rc, cen_out, xyz_out, self.report = self.backend.gp_solve(p, opt)
        if rc != 0:
            return False
        for f, n in node.items():  # ConvertResults: t = -R c (gp.cc:566-572)
            frames[f].rig_from_world = Rigid3d(np.asarray(frames[f].rig_from_world.rotation), -R_f[f] @ cen_out[n])
        for b, (rid, cam_id) in enumerate(sensor_ids):  # centre -> translation (gp.cc:576-582)
            cfr = rigs[rid].MaybeSensorFromRig(cam_id)
            rigs[rid].SetSensorFromRig(cam_id, Rigid3d(np.asarray(cfr.rotation), -_R(cfr.rotation) @ self.report["sensor_center"][b]))
        for t, x in zip(tids, xyz_out):
            if ctype == 1:
                break  # ONLY_CAMERAS: AddPointToCameraConstraints never ran (gp.cc:69-71)
            tracks[t].xyz = np.array(x)
            if o.optimize_points and o.generate_random_points:
                tracks[t].is_initialized = True  # gp.cc:261-264
        return True


class BundleAdjuster:
    def __init__(self, options, backend=None):
        self.options_ = options
        self.backend = backend or GpuBackend()
        self.report = None

    def GetOptions(self):
        return self.options_

    def Solve(self, rigs, cameras, frames, images, tracks) -> bool:
        from .estimators import BundleAdjusterOptions

        o = self.options_
        if not images or not tracks:
            return False  # ba.cc:17-24
        fids = list(frames.keys())
        node = {f: n for n, f in enumerate(fids)}
        N = len(fids)
        tids, off, oimg, ofeat = _pack_tracks(images, frames, tracks, node, o.min_num_view_per_track, lambda im, feat: True)
        if not tids:
            return False
        rigged = any(not has_trivial_frame(images[i], frames, rigs) for i in set(oimg))
        opt_rig = rigged and bool(o.optimize_rig_poses)
        intr_of: Dict[int, int] = {}
        intr_ids: List[int] = []

        def intr_index(cid):
            if cid not in intr_of:
                intr_of[cid] = len(intr_ids)
                intr_ids.append(cid)
            return intr_of[cid]

        M = len(oimg)
        xy = np.zeros((M, 2))
        obs_cam = np.zeros(M, dtype=np.int32)
        cam_intr = np.zeros(N, dtype=np.int32)
        img_idx: Dict[int, int] = {}
        image_frame, image_intr, image_cfr, image_sensor = [], [], [], []
        sensor_of: Dict[tuple, int] = {}
        sensor_ids, sensor_cfr = [], []
        for k, (iid, feat) in enumerate(zip(oimg, ofeat)):
            im = images[iid]
            xy[k] = im.features[feat]  # distorted pixels (ba.cc:139)
            if not rigged:
                obs_cam[k] = node[im.frame_id]
                cam_intr[node[im.frame_id]] = intr_index(im.camera_id)
                continue
            if iid not in img_idx:
                st, q = _cam_from_rig_state(im, frames, rigs)
                if st != 0:
                    return False  # BA needs every cam_from_rig (it dereferences SensorFromRig, ba.cc:148,163)
                t_cfr = np.zeros(3)
                triv = has_trivial_frame(im, frames, rigs)
                if not triv:
                    t_cfr = np.asarray(rigs[frames[im.frame_id].rig_id].MaybeSensorFromRig(im.camera_id).translation, dtype=np.float64)
                img_idx[iid] = len(image_frame)
                image_frame.append(node[im.frame_id])
                image_intr.append(intr_index(im.camera_id))
                image_cfr.append(np.concatenate([q, t_cfr]))
                sb = -1
                if opt_rig and not triv:
                    key = (frames[im.frame_id].rig_id, im.camera_id)
                    if key not in sensor_of:
                        sensor_of[key] = len(sensor_ids)
                        sensor_ids.append(key)
                        sensor_cfr.append(np.concatenate([q, t_cfr]))
                    sb = sensor_of[key]
                image_sensor.append(sb)
            obs_cam[k] = img_idx[iid]
        in_problem = np.zeros(N, dtype=bool)
        in_problem[np.asarray(image_frame, np.int64)[obs_cam] if rigged else obs_cam] = True
        fixed = int(np.nonzero(in_problem)[0][0])  # first frame that owns a parameter block (ba.cc:253-269)
        K = len(intr_ids)
        model = np.array([cameras[c].model_id for c in intr_ids], dtype=np.int32)
        # 16-wide rows when a camera model has more than 8 parameters (FULL_OPENCV, THIN_PRISM_FISHEYE, RAD_TAN_THIN_PRISM_FISHEYE)
        width = CAMERA_MAX_PARAMS_WIDE if any(len(cameras[c].params) > CAMERA_MAX_PARAMS for c in intr_ids) else CAMERA_MAX_PARAMS
        params = np.zeros((K, width))
        for k, c in enumerate(intr_ids):
            params[k, : len(cameras[c].params)] = cameras[c].params
        q0 = np.array([frames[f].rig_from_world.rotation for f in fids], dtype=np.float64)
        t0 = np.array([frames[f].rig_from_world.translation for f in fids], dtype=np.float64)
        xyz = np.array([tracks[t].xyz for t in tids], dtype=np.float64)
        p = BaProblem(num_cams=N, num_pts=len(tids), num_intr=K, pt_offset=off, obs_cam=obs_cam, obs_xy=xy, cam_intr=cam_intr,
                      cam_q=q0, cam_t=t0, pt_xyz=xyz, intr_model=model, intr_params=params, fixed_cam=fixed)
        if rigged:
            p.image_frame = np.asarray(image_frame, np.int32)
            p.image_cam_from_rig = np.asarray(image_cfr, np.float64).reshape(-1, 7)
            p.image_intr = np.asarray(image_intr, np.int32)
            if sensor_ids:
                p.image_sensor = np.asarray(image_sensor, np.int32)
                p.sensor_cam_from_rig = np.asarray(sensor_cfr, np.float64).reshape(-1, 7)
        opt = BundleAdjusterOptions(**{**vars(o), "min_num_view_per_track": 1})
        rc, q, t, X, intr, self.report = self.backend.ba_solve(p, opt)
        if rc != 0:
            return False
        for f, n in node.items():
            frames[f].rig_from_world = Rigid3d(np.array(q[n]), np.array(t[n]))
        for tid, x in zip(tids, X):
            tracks[tid].xyz = np.array(x)
        for k, c in enumerate(intr_ids):
            cameras[c].params = np.array(intr[k, : len(cameras[c].params)])
        for b, (rid, cam_id) in enumerate(sensor_ids):  # the blocks are the rigs' own storage (ba.cc:163-175)
            s = self.report["sensor_cam_from_rig"][b]
            rigs[rid].SetSensorFromRig(cam_id, Rigid3d(np.array(s[:4]), np.array(s[4:])))
        return True
