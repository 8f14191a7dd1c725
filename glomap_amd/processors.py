"""Host-side mirror of the reference's processors that run between the estimator calls, on the C ABI
(include/gsfm.h, glomap_amd/csrc/filters.hip):

  TrackFilter.FilterTracksByReprojection / FilterTracksByAngle / FilterTrackTriangulationAngle
      (glomap/processors/track_filter.h:9-31), NormalizeReconstruction
      (glomap/processors/reconstruction_normalizer.h), RelPoseFilter.FilterRotations
      (glomap/processors/relpose_filter.h), UndistortFeatures (glomap/processors/image_undistorter.cc:7-46).

Flat level only: arrays as in gsfm_scene_view (numpy on the host or DeviceArrays in HBM); results are keep
masks + the counter the reference returns.  Nothing here computes on the CPU."""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass
from typing import Optional

import numpy as np

from . import _lib
from .estimators import _h, _is_dev, _mem_of, default_context


@dataclass
class SceneView:
    """Camera poses + tracks (track-major observations), the read-only input of the track filters."""

    num_cams: int
    pt_offset: np.ndarray  # [P+1] int64
    obs_cam: np.ndarray  # [M] int32
    cam_q: np.ndarray  # [N,4] (w,x,y,z) cam_from_world
    cam_t: np.ndarray  # [N,3]
    pt_xyz: np.ndarray  # [P,3]
    obs_undist: Optional[np.ndarray] = None  # [M,3] image.features_undist
    obs_xy: Optional[np.ndarray] = None  # [M,2] image.features (pixels)
    cam_calibrated: Optional[np.ndarray] = None  # [N] uint8
    cam_intr: Optional[np.ndarray] = None  # [N] int32
    intr_model: Optional[np.ndarray] = None  # [K] int32
    intr_params: Optional[np.ndarray] = None  # [K,8]; [K,16] when a camera model has more than 8 parameters


def _view_c(v: SceneView, keep: list) -> _lib.SceneViewC:
    off, oc = _h(v.pt_offset, np.int64), _h(v.obs_cam, np.int32)
    q, t, X = _h(v.cam_q, np.float64), _h(v.cam_t, np.float64), _h(v.pt_xyz, np.float64)
    und, xy = _h(v.obs_undist, np.float64), _h(v.obs_xy, np.float64)
    cal = _h(v.cam_calibrated, np.uint8)
    ci, im, ip_ = _h(v.cam_intr, np.int32), _h(v.intr_model, np.int32), _h(v.intr_params, np.float64)
    keep.extend([off, oc, q, t, X, und, xy, cal, ci, im, ip_])
    c = _lib.SceneViewC()
    c.mem = _mem_of(off, oc, q, t, X)
    c.num_cams = int(v.num_cams)
    c.num_pts = int(off.shape[0]) - 1
    c.num_obs = int(oc.shape[0])
    c.pt_offset, c.obs_cam = _lib.ptr(off), _lib.ptr(oc)
    c.obs_undist, c.obs_xy = _lib.ptr(und), _lib.ptr(xy)
    c.cam_q, c.cam_t, c.pt_xyz = _lib.ptr(q), _lib.ptr(t), _lib.ptr(X)
    c.cam_calibrated = _lib.ptr(cal)
    c.num_intr = 0 if im is None else int(im.shape[0])
    c.cam_intr, c.intr_model, c.intr_params = _lib.ptr(ci), _lib.ptr(im), _lib.ptr(ip_)
    c.intr_stride = 0 if ip_ is None or len(ip_.shape) != 2 else int(ip_.shape[1])  # 8, or 16 (wide camera models)
    return c


def _out_mask(ctx, mem, n):
    return np.zeros(n, dtype=np.uint8) if mem == _lib.GSFM_MEM_HOST else _lib.DeviceArray(ctx, (n,), np.uint8)


class TrackFilter:
    """glomap/processors/track_filter.h:9-31 — static methods, flat arrays; each returns (keep mask, counter)."""

    @staticmethod
    def FilterTracksByReprojection(view: SceneView, max_reprojection_error: float = 1e-2, in_normalized_image: bool = True,
                                   ctx=None):
        ctx = ctx or default_context()
        keep: list = []
        c = _view_c(view, keep)
        out = _out_mask(ctx, c.mem, c.num_obs)
        n = C.c_int64(0)
        rc = ctx.lib.gsfm_filter_tracks_by_reprojection(ctx.handle, C.byref(c), max_reprojection_error,
                                                        int(in_normalized_image), _lib.ptr(out), C.byref(n))
        if rc != 0:
            raise _lib.GsfmError(rc, "gsfm_filter_tracks_by_reprojection")
        return out, n.value

    @staticmethod
    def FilterTracksByAngle(view: SceneView, max_angle_error: float = 1.0, ctx=None):
        ctx = ctx or default_context()
        keep: list = []
        c = _view_c(view, keep)
        out = _out_mask(ctx, c.mem, c.num_obs)
        n = C.c_int64(0)
        rc = ctx.lib.gsfm_filter_tracks_by_angle(ctx.handle, C.byref(c), max_angle_error, _lib.ptr(out), C.byref(n))
        if rc != 0:
            raise _lib.GsfmError(rc, "gsfm_filter_tracks_by_angle")
        return out, n.value

    @staticmethod
    def FilterTrackTriangulationAngle(view: SceneView, min_angle: float = 1.0, ctx=None):
        ctx = ctx or default_context()
        keep: list = []
        c = _view_c(view, keep)
        out = _out_mask(ctx, c.mem, c.num_pts)
        n = C.c_int64(0)
        rc = ctx.lib.gsfm_filter_tracks_triangulation_angle(ctx.handle, C.byref(c), min_angle, _lib.ptr(out), C.byref(n))
        if rc != 0:
            raise _lib.GsfmError(rc, "gsfm_filter_tracks_triangulation_angle")
        return out, n.value


def UndistortFeatures(feat_xy, feat_intr, intr_model, intr_params, ctx=None):
    """gsfm_undistort_features — what UndistortImages (image_undistorter.cc:7-46) stores in Image::features_undist: the unit
    bearing camera.CamFromImg(xy).value_or(Zero).homogeneous().normalized() of every pixel feat_xy [F,2] seen through intrinsics
    row feat_intr [F] of intr_model [K] / intr_params [K,8] or [K,16].  numpy arrays (host) or DeviceArrays (the rays stay in
    HBM).  Returns rays [F,3]."""
    ctx = ctx or default_context()
    xy, fi = _h(feat_xy, np.float64), _h(feat_intr, np.int32)
    im, ip_ = _h(intr_model, np.int32), _h(intr_params, np.float64)
    mem = _mem_of(xy, fi, im, ip_)
    F = int(xy.shape[0])
    out = np.zeros((F, 3)) if mem == _lib.GSFM_MEM_HOST else _lib.DeviceArray(ctx, (F, 3), np.float64)
    rc = ctx.lib.gsfm_undistort_features(ctx.handle, mem, F, _lib.ptr(xy), _lib.ptr(fi), int(im.shape[0]), _lib.ptr(im), _lib.ptr(ip_),
                                         int(ip_.shape[1]), _lib.ptr(out))
    if rc != 0:
        raise _lib.GsfmError(rc, "gsfm_undistort_features")
    return out


def UndistortImages(cameras, images, clean_points: bool = True, ctx=None):
    """UndistortImages (glomap/processors/image_undistorter.cc:7-46) on the scene containers of glomap_amd.scene: for every image
    whose bearings are missing (or for all of them with clean_points) features_undist [F,3] is recomputed from features [F,2]
    through the image's camera — ONE gsfm_undistort_features sweep over all of them.  `cameras`: {camera_id: scene.Camera},
    `images`: {image_id: scene.Image}; modified in place."""
    from .flat import CAMERA_MAX_PARAMS, CAMERA_MAX_PARAMS_WIDE

    todo = [im for im in images.values()
            if im.features is not None and (clean_points or im.features_undist is None or len(im.features_undist) != len(im.features))]
    if not todo:
        return
    cam_ids = sorted({im.camera_id for im in todo})
    row = {cid: k for k, cid in enumerate(cam_ids)}
    width = CAMERA_MAX_PARAMS_WIDE if any(len(cameras[c].params) > CAMERA_MAX_PARAMS for c in cam_ids) else CAMERA_MAX_PARAMS
    par = np.zeros((len(cam_ids), width))
    for c, k in row.items():
        par[k, : len(cameras[c].params)] = cameras[c].params
    model = np.array([cameras[c].model_id for c in cam_ids], dtype=np.int32)
    xy = np.concatenate([np.asarray(im.features, dtype=np.float64).reshape(-1, 2) for im in todo])
    fi = np.concatenate([np.full(len(im.features), row[im.camera_id], dtype=np.int32) for im in todo])
    rays = UndistortFeatures(xy, fi, model, par, ctx=ctx)
    o = 0
    for im in todo:
        n = len(im.features)
        im.features_undist = rays[o : o + n].copy()
        o += n


def CompactObservations(pt_offset, arrays, obs_keep=None, track_keep=None, ctx=None):
    """gsfm_tracks_compact: drops the observations a filter flagged (obs_keep [M] and / or track_keep [P], 0 = drop) from a
    track-major observation set, IN PLACE — what the reference does by erasing from Track::observations
    (track_filter.cc:36-44, 75-83, 120-123).  pt_offset [P+1] int64 is rewritten; every array of `arrays` ([M] or [M, k],
    4- or 8-byte items) is compacted.  numpy arrays (host) or DeviceArrays (HBM, nothing but the new count leaves the
    device).  Returns (new observation count, [the arrays cut to that length: numpy slices / DeviceArray.prefix views])."""
    ctx = ctx or default_context()
    if isinstance(pt_offset, np.ndarray):
        assert pt_offset.dtype == np.int64 and pt_offset.flags["C_CONTIGUOUS"], "pt_offset is rewritten in place: contiguous int64"
    off = _h(pt_offset, np.int64)
    arrs = list(arrays)
    for a in arrs:
        if isinstance(a, np.ndarray):
            assert a.flags["C_CONTIGUOUS"] and a.flags["WRITEABLE"], "compaction is in place: contiguous writable arrays"
    mem = _mem_of(off, *arrs)
    ok_, tk_ = _h(obs_keep, np.uint8), _h(track_keep, np.uint8)
    M = int(arrs[0].shape[0]) if arrs else int(off.numpy()[-1] if not isinstance(off, np.ndarray) else off[-1])
    P = int(off.shape[0]) - 1
    # the library trusts these lengths: check them here
    assert all(int(a.shape[0]) == M for a in arrs), "every per-observation array needs M rows"
    assert ok_ is None or int(ok_.shape[0]) == M, "obs_keep needs one byte per observation"
    assert tk_ is None or int(tk_.shape[0]) == P, "track_keep needs one byte per track"
    assert len({_is_dev(a) for a in (off, ok_, tk_, *arrs) if a is not None}) == 1, "host and device arrays mixed"
    nbytes = [int(np.prod(a.shape[1:], dtype=np.int64)) * np.dtype(a.dtype).itemsize for a in arrs]
    ptrs = (C.c_void_p * max(1, len(arrs)))(*[_lib.ptr(a) for a in arrs])
    eb = (C.c_int32 * max(1, len(arrs)))(*nbytes)
    n = C.c_int64(0)
    rc = ctx.lib.gsfm_tracks_compact(ctx.handle, mem, P, M, _lib.ptr(off), _lib.ptr(ok_), _lib.ptr(tk_), len(arrs), ptrs, eb, C.byref(n))
    if rc != 0:
        raise _lib.GsfmError(rc, "gsfm_tracks_compact")
    cut = [a[: n.value] if isinstance(a, np.ndarray) else a.prefix(n.value) for a in arrs]
    return n.value, cut


def NormalizeReconstruction(cam_q, cam_t, pt_xyz, cam_registered=None, fixed_scale=False, extent=10.0, p0=0.1, p1=0.9,
                            ctx=None):
    """reconstruction_normalizer.cc:5-85.  Returns (cam_t', pt_xyz', (scale, translation[3])); inputs untouched."""
    ctx = ctx or default_context()
    q, t, X = _h(cam_q, np.float64), _h(cam_t, np.float64), _h(pt_xyz, np.float64)
    reg = _h(cam_registered, np.uint8)
    mem = _mem_of(q, t, X)
    t2 = t.copy() if isinstance(t, np.ndarray) else t.clone()
    X2 = X.copy() if isinstance(X, np.ndarray) else X.clone()
    sim = (C.c_double * 4)()
    rc = ctx.lib.gsfm_normalize_reconstruction(ctx.handle, mem, int(q.shape[0]), _lib.ptr(reg), _lib.ptr(q), _lib.ptr(t2),
                                               int(X.shape[0]), _lib.ptr(X2), int(fixed_scale), extent, p0, p1, sim)
    if rc != 0:
        raise _lib.GsfmError(rc, "gsfm_normalize_reconstruction")
    return t2, X2, (sim[0], np.array([sim[1], sim[2], sim[3]]))


class RelPoseFilter:
    """glomap/processors/relpose_filter.h — FilterRotations on flat arrays: (keep mask [E], num_invalid)."""

    @staticmethod
    def FilterRotations(node_q, edge_i, edge_j, edge_q, max_angle: float = 5.0, ctx=None):
        ctx = ctx or default_context()
        nq, ei, ej, eq = _h(node_q, np.float64), _h(edge_i, np.int32), _h(edge_j, np.int32), _h(edge_q, np.float64)
        mem = _mem_of(nq, ei, ej, eq)
        E = int(ei.shape[0])
        out = _out_mask(ctx, mem, E)
        n = C.c_int64(0)
        rc = ctx.lib.gsfm_filter_rotations(ctx.handle, mem, int(nq.shape[0]), _lib.ptr(nq), E, _lib.ptr(ei), _lib.ptr(ej),
                                           _lib.ptr(eq), max_angle, _lib.ptr(out), C.byref(n))
        if rc != 0:
            raise _lib.GsfmError(rc, "gsfm_filter_rotations")
        return out, n.value
