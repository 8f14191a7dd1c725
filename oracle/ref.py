"""ctypes loader of oracle/_ref/libref_glomap.so: the REFERENCE'S OWN glomap/scene/view_graph.cc and
glomap/controllers/track_establishment.cc, compiled from /root/reference by `make -C oracle ref` against the stand-in
scene types of oracle/ref_shim/ (flat entry points: oracle/ref_glue.cc).  Test infrastructure: it pins oracle/tracks.py —
and through it csrc/tracks.hip, which is bit-exact against oracle/tracks.py — to reference code.  Only tests/ may import
this.  load() builds the library when the reference tree is present and returns None when neither the tree nor a prebuilt
library exists (the GPU box has the prebuilt one: oracle/_ref/ travels with the snapshot)."""
from __future__ import annotations

import ctypes as C
import os
import subprocess
import sys
from pathlib import Path

import numpy as np

HERE = Path(__file__).resolve().parent
LIB = HERE / "_ref" / "libref_glomap.so"
REFERENCE = Path(os.environ.get("GSFM_REFERENCE_DIR", "/root/reference"))
_lib = None


def load():
    global _lib
    if _lib is not None:
        return _lib
    if (REFERENCE / "glomap" / "scene" / "view_graph.cc").exists():
        r = subprocess.run(["make", "-C", str(HERE), "-s", "ref", "ref_solve", f"REF={REFERENCE}"], capture_output=True, text=True)
        if r.returncode != 0:  # a shim compile error skips the tests that need the library instead of breaking collection
            print("[oracle/ref] make ref failed:\n" + r.stderr[-2000:], file=sys.stderr)
    if not LIB.exists():
        return None
    lib = C.CDLL(str(LIB))
    vp, ip, lp = C.c_void_p, C.c_int, C.c_long
    lib.ref_keep_largest_connected_components.restype = ip
    lib.ref_keep_largest_connected_components.argtypes = [ip, vp, ip, lp, vp, vp, vp, vp]
    lib.ref_establish_full_tracks.restype = lp
    lib.ref_establish_full_tracks.argtypes = [ip, vp, vp, lp, vp, vp, vp, vp, vp, vp, C.c_double, lp, lp, vp, vp, vp, vp]
    lib.ref_find_tracks_for_problem.restype = lp
    lib.ref_find_tracks_for_problem.argtypes = [ip, vp, lp, vp, vp, vp, vp, ip, ip, ip, ip, vp, vp]
    _lib = lib
    return lib


def _p(a):
    return None if a is None else a.ctypes.data


def keep_largest_connected_components(num_images, image_frame, num_frames, pair_image1, pair_image2, pair_valid):
    """ViewGraph::KeepLargestConnectedComponents (view_graph.cc:56-97).  Returns (frame_registered [F] bool, pair_valid' [E]
    bool, the reference's return value).  NOTE: when nothing is valid the reference returns 0 before it touches any flag —
    frame_registered then holds the flags the frames were created with (all False here)."""
    lib = load()
    imf = np.ascontiguousarray(image_frame, dtype=np.int32)
    p1, p2 = np.ascontiguousarray(pair_image1, dtype=np.int32), np.ascontiguousarray(pair_image2, dtype=np.int32)
    pv = np.ascontiguousarray(pair_valid, dtype=np.uint8).copy()
    reg = np.zeros(num_frames, dtype=np.uint8)
    n = lib.ref_keep_largest_connected_components(int(num_images), _p(imf), int(num_frames), len(p1), _p(p1), _p(p2), _p(pv), _p(reg))
    return reg.astype(bool), pv.astype(bool), int(n)


def establish_full_tracks(pair_image1, pair_image2, pair_valid, pair_offset, match_feat1, match_feat2, feat_offset, feat_xy,
                          thres_inconsistency=10.0):
    """TrackEngine::EstablishFullTracks (track_establishment.cc:5-152) with every match an inlier.  Returns
    ({reference track id: [(image, feature), ...]} — empty list = discarded —, number of discarded tracks)."""
    lib = load()
    fo = np.ascontiguousarray(feat_offset, dtype=np.int64)
    xy = np.ascontiguousarray(feat_xy, dtype=np.float64)
    p1, p2 = np.ascontiguousarray(pair_image1, dtype=np.int32), np.ascontiguousarray(pair_image2, dtype=np.int32)
    pv = None if pair_valid is None else np.ascontiguousarray(pair_valid, dtype=np.uint8)
    po = np.ascontiguousarray(pair_offset, dtype=np.int64)
    f1, f2 = np.ascontiguousarray(match_feat1, dtype=np.uint32), np.ascontiguousarray(match_feat2, dtype=np.uint32)
    cap = 2 * len(f1) + 8
    tid = np.zeros(cap, dtype=np.uint64)
    off = np.zeros(cap + 1, dtype=np.int64)
    obs = np.zeros(cap, dtype=np.uint64)
    disc = C.c_long(0)
    T = lib.ref_establish_full_tracks(len(fo) - 1, _p(fo), _p(xy), len(p1), _p(p1), _p(p2), _p(pv), _p(po), _p(f1), _p(f2),
                                      float(thres_inconsistency), cap, cap, _p(tid), _p(off), _p(obs), C.byref(disc))
    assert T >= 0
    tracks = {}
    for t in range(T):
        g = obs[off[t] : off[t + 1]]
        tracks[int(tid[t])] = [(int(v >> np.uint64(32)), int(v & np.uint64(0xFFFFFFFF))) for v in g]
    return tracks, int(disc.value)


def find_tracks_for_problem(num_images, image_registered, track_id, track_offset, obs_image, obs_feature,
                            min_num_tracks_per_view=-1, min_num_view_per_track=3, max_num_view_per_track=100, max_num_tracks=10000000):
    """TrackEngine::FindTracksForProblem (track_establishment.cc:154-227).  Returns (selected [T] bool, kept_obs [M] bool, the
    reference's return value)."""
    lib = load()
    reg = np.ascontiguousarray(image_registered, dtype=np.uint8)
    tid = np.ascontiguousarray(track_id, dtype=np.uint64)
    off = np.ascontiguousarray(track_offset, dtype=np.int64)
    oi, of_ = np.ascontiguousarray(obs_image, dtype=np.int32), np.ascontiguousarray(obs_feature, dtype=np.uint32)
    sel = np.zeros(len(tid), dtype=np.uint8)
    kept = np.zeros(max(1, len(oi)), dtype=np.uint8)
    n = lib.ref_find_tracks_for_problem(int(num_images), _p(reg), len(tid), _p(tid), _p(off), _p(oi), _p(of_), int(min_num_tracks_per_view),
                                        int(min_num_view_per_track), int(max_num_view_per_track), int(max_num_tracks), _p(sel), _p(kept))
    assert n >= 0
    return sel.astype(bool), kept[: len(oi)].astype(bool), int(n)


def _load_processors():
    lib = load()
    if lib is not None and not getattr(lib, "_proc_ready", False):
        vp, ip, lp, d = C.c_void_p, C.c_int, C.c_long, C.c_double
        lib.ref_filter_tracks.restype = ip
        lib.ref_filter_tracks.argtypes = [ip, ip, vp, vp, vp, lp, vp, vp, vp, vp, d, vp]
        lib.ref_normalize_reconstruction.restype = ip
        lib.ref_normalize_reconstruction.argtypes = [ip, vp, vp, vp, lp, vp, ip, d, d, d, vp]
        lib._proc_ready = True
    return lib


def filter_tracks(mode, cam_q, cam_t, pt_offset, obs_cam, obs_undist, pt_xyz, threshold, cam_calibrated=None):
    """TrackFilter::FilterTracksByReprojection (mode 0, normalised image coordinates) / FilterTracksByAngle (1) /
    FilterTrackTriangulationAngle (2), track_filter.cc:7-127, trivial frames.  Returns (keep [M] bool: the observation is still
    in its track, the reference's return value = tracks changed / removed)."""
    lib = _load_processors()
    q, t = np.ascontiguousarray(cam_q, dtype=np.float64), np.ascontiguousarray(cam_t, dtype=np.float64)
    off, oc = np.ascontiguousarray(pt_offset, dtype=np.int64), np.ascontiguousarray(obs_cam, dtype=np.int32)
    und = None if obs_undist is None else np.ascontiguousarray(obs_undist, dtype=np.float64)
    X = np.ascontiguousarray(pt_xyz, dtype=np.float64)
    cal = None if cam_calibrated is None else np.ascontiguousarray(cam_calibrated, dtype=np.uint8)
    keep = np.zeros(max(1, len(oc)), dtype=np.uint8)
    n = lib.ref_filter_tracks(int(mode), len(q), _p(q), _p(t), _p(cal), len(off) - 1, _p(off), _p(oc), _p(und), _p(X), float(threshold), _p(keep))
    assert n >= 0
    return keep[: len(oc)].astype(bool), int(n)


def normalize_reconstruction(cam_q, cam_t, pt_xyz, cam_registered=None, fixed_scale=False, extent=10.0, p0=0.1, p1=0.9):
    """NormalizeReconstruction (reconstruction_normalizer.cc:5-85).  Returns (cam_t', pt_xyz', (scale, translation))."""
    lib = _load_processors()
    q = np.ascontiguousarray(cam_q, dtype=np.float64)
    t = np.array(cam_t, dtype=np.float64, order="C", copy=True)
    X = np.array(pt_xyz, dtype=np.float64, order="C", copy=True)
    reg = None if cam_registered is None else np.ascontiguousarray(cam_registered, dtype=np.uint8)
    sim = np.zeros(4)
    rc = lib.ref_normalize_reconstruction(len(q), _p(q), _p(t), _p(reg), len(X), _p(X), int(fixed_scale), float(extent), float(p0), float(p1), _p(sim))
    assert rc == 0
    return t, X, (float(sim[0]), sim[1:].copy())


def filter_rotations(node_q, edge_i, edge_j, edge_q, max_angle_deg, node_registered=None, edge_valid=None):
    """RelPoseFilter::FilterRotations (relpose_filter.cc:7-33).  Returns (edge_valid' [E] bool, number invalidated)."""
    lib = _load_processors()
    if not getattr(lib, "_rot_ready", False):
        lib.ref_filter_rotations.restype = C.c_long
        lib.ref_filter_rotations.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_long, C.c_void_p, C.c_void_p, C.c_void_p, C.c_double, C.c_void_p]
        lib._rot_ready = True
    nq, eq = np.ascontiguousarray(node_q, dtype=np.float64), np.ascontiguousarray(edge_q, dtype=np.float64)
    ei, ej = np.ascontiguousarray(edge_i, dtype=np.int32), np.ascontiguousarray(edge_j, dtype=np.int32)
    reg = None if node_registered is None else np.ascontiguousarray(node_registered, dtype=np.uint8)
    ev = np.ones(len(ei), dtype=np.uint8) if edge_valid is None else np.ascontiguousarray(edge_valid, dtype=np.uint8).copy()
    n = lib.ref_filter_rotations(len(nq), _p(nq), _p(reg), len(ei), _p(ei), _p(ej), _p(eq), float(max_angle_deg), _p(ev))
    return ev.astype(bool), int(n)


# ---------------------------------------------------------------------------------------------------------------
# the reference's global positioning PROBLEM BUILDER (global_positioning.cc + cost_function.h) on a recording Ceres
# ---------------------------------------------------------------------------------------------------------------
LIB_GP = HERE / "_ref" / "libref_glomap_gp.so"
_lib_gp = None


class _GpOptions(C.Structure):
    _fields_ = [("generate_random_positions", C.c_int), ("generate_random_points", C.c_int), ("generate_scales", C.c_int),
                ("optimize_positions", C.c_int), ("optimize_points", C.c_int), ("optimize_scales", C.c_int),
                ("min_num_view_per_track", C.c_int), ("seed", C.c_uint), ("constraint_type", C.c_int),
                ("constraint_reweight_scale", C.c_double), ("thres_loss_function", C.c_double)]


def load_gp():
    global _lib_gp
    if _lib_gp is None:
        load()  # (runs `make ref` where the reference tree exists)
        if LIB_GP.exists():
            _lib_gp = C.CDLL(str(LIB_GP))
            _lib_gp.ref_gp_build.restype = C.c_long
    return _lib_gp


def gp_build(cam_q, cam_t, pt_offset, obs_cam, obs_undist, pt_xyz, cam_calibrated=None, cam_registered=None, pt_initialized=None,
             pair_i=None, pair_j=None, pair_valid=None, pair_t=None, **options):
    """GlobalPositioner::Solve of the reference with a Ceres that records instead of minimising (global_positioning.cc:28-93).
    Returns a dict: frame_order / track_order (the reference's container walks = its draw order), center_start [N,3] and
    xyz_start [P,3] (the start point handed to Ceres), cam_t_after [N,3] (after ConvertResults), initial_cost, and per
    residual block: cam, cam2, pt, scale, loss_scale, lower, scale_const, dir."""
    lib = load_gp()
    pre, keep, (N, P, M, E), o = _gp_prefix(cam_q, cam_t, pt_offset, obs_cam, obs_undist, pt_xyz, cam_calibrated, cam_registered, pt_initialized,
                                             pair_i, pair_j, pair_valid, pair_t, options)
    return _gp_build_call(lib, pre, N, P, M, E)


LIB_GP_SOLVE = HERE / "_ref" / "libref_glomap_gp_solve.so"
_lib_gp_solve = None


def load_gp_solve():
    global _lib_gp_solve
    if _lib_gp_solve is None:
        load()
        if LIB_GP_SOLVE.exists():
            _lib_gp_solve = C.CDLL(str(LIB_GP_SOLVE))
            _lib_gp_solve.ref_gp_solve.restype = C.c_long
    return _lib_gp_solve


def gp_solve(cam_q, cam_t, pt_offset, obs_cam, obs_undist, pt_xyz, cam_calibrated=None, cam_registered=None, pt_initialized=None,
             pair_i=None, pair_j=None, pair_valid=None, pair_t=None, max_num_iterations=0, **options):
    """GlobalPositioner::Solve of the reference (global_positioning.cc:28-93, compiled unmodified) run TO ITS END POINT on the
    solving Ceres stand-in (oracle/ref_shim_solve/ceres/ceres.h: Jacobians = dual-number derivatives of the reference's own
    functors; the trust-region loop incl. the projected line search restated from Ceres' sources).  Returns a dict:
    frame_order / track_order (the reference's draw order), center [N,3] (camera centres after ConvertResults), xyz [P,3],
    initial_cost, final_cost, iterations, successful_steps, line_search_trials, line_search_shrunk, termination, constrained,
    trace [iterations,7] (columns of gsfm_ctx_lm_trace)."""
    lib = load_gp_solve()
    pre, keep, (N, P, M, E), o = _gp_prefix(cam_q, cam_t, pt_offset, obs_cam, obs_undist, pt_xyz, cam_calibrated, cam_registered, pt_initialized,
                                             pair_i, pair_j, pair_valid, pair_t, options)
    cap = 512
    out = dict(frame_order=np.zeros(N, np.int32), track_order=np.zeros(max(P, 1), np.int64), center=np.zeros((N, 3)),
               xyz=np.zeros((max(P, 1), 3)))
    summ, trace = np.zeros(8), np.zeros((cap, 7))
    args = pre + [C.c_int(int(max_num_iterations)), _p(out["frame_order"]), _p(out["track_order"]), _p(out["center"]), _p(out["xyz"]),
                  _p(summ), C.c_long(cap), _p(trace)]
    rows = lib.ref_gp_solve(*_cv(args))
    out["ok"] = rows >= 0
    out["track_order"], out["xyz"] = out["track_order"][:P], out["xyz"][:P]
    out.update(initial_cost=float(summ[0]), final_cost=float(summ[1]), iterations=int(summ[2]), successful_steps=int(summ[3]),
               line_search_trials=int(summ[4]), line_search_shrunk=int(summ[5]), termination=int(summ[6]), constrained=bool(summ[7]),
               trace=trace[:max(int(rows), 0)].copy())
    return out


def _gp_prefix(cam_q, cam_t, pt_offset, obs_cam, obs_undist, pt_xyz, cam_calibrated, cam_registered, pt_initialized, pair_i, pair_j, pair_valid,
               pair_t, options):
    """The flat arguments shared by ref_gp_build and ref_gp_adapter_solve (oracle/ref_glue_gp_scene.h)."""
    q, t = np.ascontiguousarray(cam_q, dtype=np.float64), np.ascontiguousarray(cam_t, dtype=np.float64)
    off, oc = np.ascontiguousarray(pt_offset, dtype=np.int64), np.ascontiguousarray(obs_cam, dtype=np.int32)
    und, X = np.ascontiguousarray(obs_undist, dtype=np.float64), np.ascontiguousarray(pt_xyz, dtype=np.float64)
    u8 = lambda a: None if a is None else np.ascontiguousarray(a, dtype=np.uint8)  # noqa: E731
    cal, reg, ini, pv = u8(cam_calibrated), u8(cam_registered), u8(pt_initialized), u8(pair_valid)
    E = 0 if pair_i is None else len(pair_i)
    pi = None if pair_i is None else np.ascontiguousarray(pair_i, dtype=np.int32)
    pj = None if pair_j is None else np.ascontiguousarray(pair_j, dtype=np.int32)
    pt = None if pair_t is None else np.ascontiguousarray(pair_t, dtype=np.float64)
    o = _GpOptions(1, 1, 1, 1, 1, 1, 3, 1, 0, 1.0, 0.1)
    for k, v in options.items():
        setattr(o, k, type(getattr(o, k))(v))
    N, P, M = len(q), len(off) - 1, len(oc)
    pre = [C.c_int(N), _p(q), _p(t), _p(cal), _p(reg), C.c_long(P), _p(off), _p(oc), _p(und), _p(X), _p(ini), C.c_long(E), _p(pi), _p(pj), _p(pv),
           _p(pt), C.byref(o)]
    return pre, (q, t, off, oc, und, X, cal, reg, ini, pv, pi, pj, pt, o), (N, P, M, E), o


def _cv(args):
    return [a if not isinstance(a, int) and a is not None else C.c_void_p(a) for a in args]


def _gp_build_call(lib, pre, N, P, M, E):
    cap = M + E + 8
    out = dict(frame_order=np.zeros(N, np.int32), track_order=np.zeros(max(P, 1), np.int64), center_start=np.zeros((N, 3)),
               xyz_start=np.zeros((max(P, 1), 3)), cam_t_after=np.zeros((N, 3)), cam=np.zeros(cap, np.int32), cam2=np.zeros(cap, np.int32),
               pt=np.zeros(cap, np.int64), scale=np.zeros(cap), loss_scale=np.zeros(cap), lower=np.zeros(cap), scale_const=np.zeros(cap, np.uint8),
               dir=np.zeros((cap, 3)))
    cost = C.c_double(0.0)
    args = pre + [_p(out["frame_order"]), _p(out["track_order"]), _p(out["center_start"]), _p(out["xyz_start"]),
                  _p(out["cam_t_after"]), C.c_long(cap), _p(out["cam"]), _p(out["cam2"]), _p(out["pt"]), _p(out["scale"]), _p(out["loss_scale"]),
                  _p(out["lower"]), _p(out["scale_const"]), _p(out["dir"]), C.byref(cost)]
    R = lib.ref_gp_build(*_cv(args))
    assert R >= 0, R
    for k in ("cam", "cam2", "pt", "scale", "loss_scale", "lower", "scale_const", "dir"):
        out[k] = out[k][:R]
    out["track_order"], out["xyz_start"] = out["track_order"][:P], out["xyz_start"][:P]
    out["initial_cost"] = float(cost.value)
    out["num_residual_blocks"] = int(R)
    return out


# ---------------------------------------------------------------------------------------------------------------
# the reference's ROTATION AVERAGING (global_rotation_averaging.cc, rotation_initializer.cc, rigid3d.cc, tree.cc)
# ---------------------------------------------------------------------------------------------------------------
LIB_RA = HERE / "_ref" / "libref_glomap_ra.so"
_lib_ra = None


class _RaOptions(C.Structure):
    _fields_ = [("max_num_l1_iterations", C.c_int), ("l1_step_convergence_threshold", C.c_double),
                ("max_num_irls_iterations", C.c_int), ("irls_step_convergence_threshold", C.c_double),
                ("irls_loss_parameter_sigma", C.c_double), ("weight_type", C.c_int), ("skip_initialization", C.c_int),
                ("use_weight", C.c_int), ("use_gravity", C.c_int)]


def load_ra():
    global _lib_ra
    if _lib_ra is None:
        load()
        if LIB_RA.exists():
            _lib_ra = C.CDLL(str(LIB_RA))
            _lib_ra.ref_ra_estimate.restype = C.c_int
    return _lib_ra


def ra_estimate(rig_ref_cam, frame_rig, image_frame, image_cam, pair_i, pair_j, pair_q, pair_weight=None, pair_ninl=None,
                pair_valid=None, sensor_rig=(), sensor_cam=(), sensor_state=(), sensor_q=None, frame_q=None, frame_R_align=None,
                frame_registered=None, **options):
    """RotationEstimator::EstimateRotations of the reference (global_rotation_averaging.cc:40-85) on containers built from flat
    arrays (ids = indices).  frame_q None = frames without a pose; frame_R_align [F,3,3] with NaN rows = no gravity.  Returns a
    dict: ok, frame_q [F,4] (rig_from_world, wxyz), sensor_q [S,4] / sensor_has [S], fixed_image, tree_root (-1: no tree),
    l1_iterations, admm_iterations, irls_iterations, first_frame."""
    lib = load_ra()
    i32 = lambda a: np.ascontiguousarray(a, dtype=np.int32)  # noqa: E731
    f64 = lambda a: np.ascontiguousarray(a, dtype=np.float64)  # noqa: E731
    rr, fr, imf, imc = i32(rig_ref_cam), i32(frame_rig), i32(image_frame), i32(image_cam)
    pi, pj, pq = i32(pair_i), i32(pair_j), f64(pair_q)
    E, F, S = len(pi), len(fr), len(sensor_rig)
    pw = f64(np.full(E, -1.0) if pair_weight is None else pair_weight)
    pn = i32(np.full(E, 100) if pair_ninl is None else pair_ninl)
    pv = np.ascontiguousarray(np.ones(E) if pair_valid is None else pair_valid, dtype=np.uint8)
    sr, sc, ss = i32(sensor_rig), i32(sensor_cam), i32(sensor_state)
    sq = f64(np.tile([1.0, 0, 0, 0], (max(S, 1), 1)) if sensor_q is None else sensor_q)
    has_pose = np.ascontiguousarray(np.zeros(F) if frame_q is None else np.ones(F), dtype=np.uint8)
    fq = f64(np.tile([1.0, 0, 0, 0], (F, 1)) if frame_q is None else frame_q)
    if frame_R_align is None:
        has_g, Ra = np.zeros(F, np.uint8), np.zeros((F, 9))
    else:
        Ra = f64(frame_R_align).reshape(F, 9).copy()
        has_g = np.ascontiguousarray(~np.isnan(Ra).any(axis=1), dtype=np.uint8)
        Ra[has_g == 0] = 0.0
    reg = np.ascontiguousarray(np.ones(F) if frame_registered is None else frame_registered, dtype=np.uint8)
    o = _RaOptions(5, 1e-3, 100, 1e-3, 5.0, 0, 0, 0, 0)
    for k, v in options.items():
        setattr(o, k, type(getattr(o, k))(v))
    out_fq, out_sq, out_sh, info = np.zeros((F, 4)), np.zeros((max(S, 1), 4)), np.zeros(max(S, 1), np.uint8), np.zeros(8, np.int64)
    vp = C.c_void_p
    ok = lib.ref_ra_estimate(C.c_int(len(rr)), vp(_p(rr)), C.c_int(S), vp(_p(sr)), vp(_p(sc)), vp(_p(ss)), vp(_p(sq)), C.c_int(F), vp(_p(fr)),
                             vp(_p(has_pose)), vp(_p(fq)), vp(_p(has_g)), vp(_p(Ra)), vp(_p(reg)), C.c_int(len(imf)), vp(_p(imf)), vp(_p(imc)),
                             C.c_long(E), vp(_p(pi)), vp(_p(pj)), vp(_p(pq)), vp(_p(pw)), vp(_p(pn)), vp(_p(pv)), C.byref(o), vp(_p(out_fq)),
                             vp(_p(out_sq)), vp(_p(out_sh)), vp(_p(info)))
    return dict(ok=bool(ok), frame_q=out_fq, sensor_q=out_sq[:S], sensor_has=out_sh[:S].astype(bool), fixed_image=int(info[0]),
                tree_root=int(info[1]), l1_iterations=int(info[2]), admm_iterations=int(info[3]), irls_iterations=int(info[4]),
                first_frame=int(info[5]))


# ---------------------------------------------------------------------------------------------------------------
# the reference's bundle adjustment PROBLEM BUILDER (bundle_adjustment.cc) on the recording Ceres
# ---------------------------------------------------------------------------------------------------------------
LIB_BA = HERE / "_ref" / "libref_glomap_ba.so"
_lib_ba = None


class _BaOptions(C.Structure):
    _fields_ = [("optimize_rig_poses", C.c_int), ("optimize_rotations", C.c_int), ("optimize_translation", C.c_int),
                ("optimize_intrinsics", C.c_int), ("optimize_principal_point", C.c_int), ("optimize_points", C.c_int),
                ("min_num_view_per_track", C.c_int), ("thres_loss_function", C.c_double)]


def load_ba():
    global _lib_ba
    if _lib_ba is None:
        load()
        if LIB_BA.exists():
            _lib_ba = C.CDLL(str(LIB_BA))
            _lib_ba.ref_ba_build.restype = C.c_long
    return _lib_ba


def ba_build(cam_model, cam_params, frame_q, frame_t, image_frame, image_cam, pt_offset, obs_image, obs_xy, pt_xyz, rig_ref_cam=(0,),
             frame_rig=None, sensor_rig=(), sensor_cam=(), sensor_pose=None, frame_has_pose=None, image_present=None, **options):
    """BundleAdjuster::Solve of the reference with a Ceres that records instead of minimising (bundle_adjustment.cc:9-113).
    Every observation becomes its own feature of its image.  Returns a dict: per residual block kind / frame / track / camera /
    sensor; frame_flags, camera_flags, sensor_flags, track_flags (bit 0 in the problem, 1 rotation-or-block constant,
    2 translation constant, 3 quaternion manifold, 4 ordering group 0, 5 in no group), camera_subset [K,16], frame_order,
    linear_solver_type, preconditioner_type, initial_cost."""
    lib = load_ba()
    pre, keep, (F, I, K, S, P, M) = _ba_prefix(cam_model, cam_params, frame_q, frame_t, image_frame, image_cam, pt_offset, obs_image, obs_xy, pt_xyz,
                                               rig_ref_cam, frame_rig, sensor_rig, sensor_cam, sensor_pose, frame_has_pose, image_present, options)
    return _ba_build_call(lib, pre, F, K, S, P, M)


LIB_BA_SOLVE = HERE / "_ref" / "libref_glomap_ba_solve.so"
_lib_ba_solve = None


def load_ba_solve():
    global _lib_ba_solve
    if _lib_ba_solve is None:
        load()
        if LIB_BA_SOLVE.exists():
            _lib_ba_solve = C.CDLL(str(LIB_BA_SOLVE))
            _lib_ba_solve.ref_ba_solve.restype = C.c_long
    return _lib_ba_solve


def ba_solve(cam_model, cam_params, frame_q, frame_t, image_frame, image_cam, pt_offset, obs_image, obs_xy, pt_xyz, rig_ref_cam=(0,),
             frame_rig=None, sensor_rig=(), sensor_cam=(), sensor_pose=None, frame_has_pose=None, image_present=None, max_num_iterations=0,
             **options):
    """BundleAdjuster::Solve of the reference (bundle_adjustment.cc, compiled unmodified) run TO ITS END POINT on the solving
    Ceres stand-in (oracle/ref_shim_solve/ceres/ceres.h).  Returns a dict: frame_q [F,4] (w, x, y, z), frame_t [F,3], xyz [P,3],
    cam_params [K,16], sensor_pose [S,7], frame_order, frame_const [F], initial_cost, final_cost, iterations, successful_steps,
    termination, trace [iterations,7]."""
    lib = load_ba_solve()
    pre, keep, (F, I, K, S, P, M) = _ba_prefix(cam_model, cam_params, frame_q, frame_t, image_frame, image_cam, pt_offset, obs_image, obs_xy, pt_xyz,
                                               rig_ref_cam, frame_rig, sensor_rig, sensor_cam, sensor_pose, frame_has_pose, image_present, options)
    cap = 512
    out = dict(frame_q=np.zeros((F, 4)), frame_t=np.zeros((F, 3)), xyz=np.zeros((max(P, 1), 3)), cam_params=np.zeros((K, 16)),
               sensor_pose=np.zeros((max(S, 1), 7)), frame_order=np.zeros(F, np.int32), frame_const=np.zeros(F, np.uint8))
    summ, trace = np.zeros(8), np.zeros((cap, 7))
    vp = C.c_void_p
    rows = lib.ref_ba_solve(*pre, C.c_int(int(max_num_iterations)), vp(_p(out["frame_q"])), vp(_p(out["frame_t"])), vp(_p(out["xyz"])),
                            vp(_p(out["cam_params"])), vp(_p(out["sensor_pose"])), vp(_p(out["frame_order"])), vp(_p(out["frame_const"])),
                            vp(_p(summ)), C.c_long(cap), vp(_p(trace)))
    out["ok"] = rows >= 0
    out["xyz"], out["sensor_pose"] = out["xyz"][:P], out["sensor_pose"][:S]
    out.update(initial_cost=float(summ[0]), final_cost=float(summ[1]), iterations=int(summ[2]), successful_steps=int(summ[3]),
               termination=int(summ[6]), constrained=bool(summ[7]), trace=trace[:max(int(rows), 0)].copy())
    return out


def _ba_prefix(cam_model, cam_params, frame_q, frame_t, image_frame, image_cam, pt_offset, obs_image, obs_xy, pt_xyz, rig_ref_cam, frame_rig,
               sensor_rig, sensor_cam, sensor_pose, frame_has_pose, image_present, options):
    """The flat arguments shared by ref_ba_build and ref_ba_adapter_solve (oracle/ref_glue_ba_scene.h)."""
    i32 = lambda a: np.ascontiguousarray(a, dtype=np.int32)  # noqa: E731
    f64 = lambda a: np.ascontiguousarray(a, dtype=np.float64)  # noqa: E731
    cm, cp = i32(cam_model), np.zeros((len(cam_model), 16))  # rows of 16 doubles (ref_glue_ba_scene.h: kCamRow)
    cp[:, : np.shape(cam_params)[1]] = cam_params
    fq, ft = f64(frame_q), f64(frame_t)
    F, I, K, S = len(fq), len(image_frame), len(cm), len(sensor_rig)
    fr = i32(np.zeros(F) if frame_rig is None else frame_rig)
    hp = np.ascontiguousarray(np.ones(F) if frame_has_pose is None else frame_has_pose, dtype=np.uint8)
    imf, imc = i32(image_frame), i32(image_cam)
    pres = np.ascontiguousarray(np.ones(I) if image_present is None else image_present, dtype=np.uint8)
    off, oi = np.ascontiguousarray(pt_offset, dtype=np.int64), i32(obs_image)
    M, P = len(oi), len(off) - 1
    # features: the observations of an image, in observation order
    order = np.argsort(oi, kind="stable")
    feat_off = np.zeros(I + 1, dtype=np.int64)
    feat_off[1:] = np.cumsum(np.bincount(oi, minlength=I))
    feat_xy = f64(np.asarray(obs_xy, dtype=np.float64)[order])
    obs_feature = np.empty(M, dtype=np.int32)
    obs_feature[order] = (np.arange(M) - feat_off[oi[order]]).astype(np.int32)
    X = f64(pt_xyz)
    rr, sr, sc = i32(rig_ref_cam), i32(sensor_rig), i32(sensor_cam)
    sp = f64(np.zeros((max(S, 1), 7)) if sensor_pose is None else sensor_pose)
    o = _BaOptions(0, 1, 1, 1, 0, 1, 3, 1.0)
    for k, v in options.items():
        setattr(o, k, type(getattr(o, k))(v))
    vp = C.c_void_p
    pre = [C.c_int(K), vp(_p(cm)), vp(_p(cp)), C.c_int(len(rr)), vp(_p(rr)), C.c_int(S), vp(_p(sr)), vp(_p(sc)), vp(_p(sp)), C.c_int(F), vp(_p(fr)),
           vp(_p(hp)), vp(_p(fq)), vp(_p(ft)), C.c_int(I), vp(_p(imf)), vp(_p(imc)), vp(_p(pres)), vp(_p(feat_off)), vp(_p(feat_xy)), C.c_long(P),
           vp(_p(off)), vp(_p(oi)), vp(_p(obs_feature)), vp(_p(X)), C.byref(o)]
    return pre, (cm, cp, rr, sr, sc, sp, fr, hp, fq, ft, imf, imc, pres, feat_off, feat_xy, off, oi, obs_feature, X, o), (F, I, K, S, P, M)


def _ba_build_call(lib, pre, F, K, S, P, M):
    cap = M + 8
    out = dict(kind=np.zeros(cap, np.int32), frame=np.zeros(cap, np.int32), track=np.zeros(cap, np.int64), camera=np.zeros(cap, np.int32),
               sensor=np.zeros(cap, np.int32), frame_flags=np.zeros(F, np.uint8), camera_flags=np.zeros(K, np.uint8),
               camera_subset=np.zeros((K, 16), np.uint8), sensor_flags=np.zeros(max(S, 1), np.uint8), track_flags=np.zeros(max(P, 1), np.uint8),
               frame_order=np.zeros(F, np.int32))
    info, cost = np.zeros(4, np.int64), C.c_double(0.0)
    vp = C.c_void_p
    R = lib.ref_ba_build(*pre, C.c_long(cap), vp(_p(out["kind"])), vp(_p(out["frame"])), vp(_p(out["track"])), vp(_p(out["camera"])),
                         vp(_p(out["sensor"])), vp(_p(out["frame_flags"])), vp(_p(out["camera_flags"])), vp(_p(out["camera_subset"])),
                         vp(_p(out["sensor_flags"])), vp(_p(out["track_flags"])), vp(_p(out["frame_order"])), vp(_p(info)), C.byref(cost))
    out["num_residual_blocks"] = int(R)
    if R < 0:
        return out
    for k in ("kind", "frame", "track", "camera", "sensor"):
        out[k] = out[k][:R]
    out["sensor_flags"], out["track_flags"] = out["sensor_flags"][:S], out["track_flags"][:P]
    out["linear_solver_type"], out["preconditioner_type"] = int(info[0]), int(info[1])
    out["initial_cost"] = float(cost.value)
    return out


# ---------------------------------------------------------------------------------------------------------------
# the reference's rotation-averaging CONTROLLER on either estimator: its own, or libgsfm's adapter class (the drop-in)
# ---------------------------------------------------------------------------------------------------------------
LIB_DROPIN = HERE / "_ref" / "libref_dropin_ra.so"
_lib_dropin = None


def load_dropin():
    """oracle/_ref/libref_dropin_ra.so (links glomap_amd/csrc/libgsfm.so): built where the reference tree and the built
    libgsfm.so exist, else the prebuilt file, else None."""
    global _lib_dropin
    if _lib_dropin is None:
        load()
        if (REFERENCE / "glomap" / "controllers" / "rotation_averager.cc").exists() and (HERE.parent / "glomap_amd" / "csrc" / "libgsfm.so").exists():
            subprocess.run(["make", "-C", str(HERE), "-s", "ref_dropin", f"REF={REFERENCE}"], check=True)
        if LIB_DROPIN.exists():
            _lib_dropin = C.CDLL(str(LIB_DROPIN))
            _lib_dropin.ref_ra_policy.restype = C.c_int
    return _lib_dropin


def ra_policy(which, rig_ref_cam, frame_rig, image_frame, image_cam, pair_i, pair_j, pair_q, pair_weight=None, pair_ninl=None,
              pair_valid=None, sensor_rig=(), sensor_cam=(), sensor_state=(), sensor_q=None, frame_q=None, frame_R_align=None,
              frame_registered=None, use_stratified=True, images_reversed=False, **options):
    """glomap::SolveRotationAveraging (controllers/rotation_averager.cc:8-198), the reference's source compiled unmodified:
    which = 0 on the reference's own RotationEstimator (CPU), which = 1 on include/gsfm_glomap_adapter.hpp's class (libgsfm, GPU).
    Arguments as ra_estimate; images_reversed fills the images map in descending id order (another iteration order than the
    frames map: tree root and gauge frame differ).  Returns a dict: ok, frame_q [F,4], sensor_q [S,4], sensor_has [S], frame_registered [F],
    pair_valid [E]."""
    lib = load_dropin()
    i32 = lambda a: np.ascontiguousarray(a, dtype=np.int32)  # noqa: E731
    f64 = lambda a: np.ascontiguousarray(a, dtype=np.float64)  # noqa: E731
    rr, fr, imf, imc = i32(rig_ref_cam), i32(frame_rig), i32(image_frame), i32(image_cam)
    pi, pj, pq = i32(pair_i), i32(pair_j), f64(pair_q)
    E, F, S = len(pi), len(fr), len(sensor_rig)
    pw = f64(np.full(E, -1.0) if pair_weight is None else pair_weight)
    pn = i32(np.full(E, 100) if pair_ninl is None else pair_ninl)
    pv = np.ascontiguousarray(np.ones(E) if pair_valid is None else pair_valid, dtype=np.uint8)
    sr, sc, ss = i32(sensor_rig), i32(sensor_cam), i32(sensor_state)
    sq = f64(np.tile([1.0, 0, 0, 0], (max(S, 1), 1)) if sensor_q is None else sensor_q)
    has_pose = np.ascontiguousarray(np.zeros(F) if frame_q is None else np.ones(F), dtype=np.uint8)
    fq = f64(np.tile([1.0, 0, 0, 0], (F, 1)) if frame_q is None else frame_q)
    if frame_R_align is None:
        has_g, Ra = np.zeros(F, np.uint8), np.zeros((F, 9))
    else:
        Ra = f64(frame_R_align).reshape(F, 9).copy()
        has_g = np.ascontiguousarray(~np.isnan(Ra).any(axis=1), dtype=np.uint8)
        Ra[has_g == 0] = 0.0
    reg = np.ascontiguousarray(np.ones(F) if frame_registered is None else frame_registered, dtype=np.uint8)
    o = _RaOptions(5, 1e-3, 100, 1e-3, 5.0, 0, 0, 0, 0)
    for k, v in options.items():
        setattr(o, k, type(getattr(o, k))(v))
    out_fq, out_sq, out_sh = np.zeros((F, 4)), np.zeros((max(S, 1), 4)), np.zeros(max(S, 1), np.uint8)
    out_reg, out_pv = np.zeros(F, np.uint8), np.zeros(max(E, 1), np.uint8)
    vp = C.c_void_p
    ok = lib.ref_ra_policy(C.c_int(int(which)), C.c_int(len(rr)), vp(_p(rr)), C.c_int(S), vp(_p(sr)), vp(_p(sc)), vp(_p(ss)), vp(_p(sq)), C.c_int(F),
                           vp(_p(fr)), vp(_p(has_pose)), vp(_p(fq)), vp(_p(has_g)), vp(_p(Ra)), vp(_p(reg)), C.c_int(len(imf)), vp(_p(imf)),
                           vp(_p(imc)), C.c_long(E), vp(_p(pi)), vp(_p(pj)), vp(_p(pq)), vp(_p(pw)), vp(_p(pn)), vp(_p(pv)), C.byref(o),
                           C.c_int(int(bool(use_stratified))), C.c_int(int(bool(images_reversed))), vp(_p(out_fq)), vp(_p(out_sq)), vp(_p(out_sh)), vp(_p(out_reg)), vp(_p(out_pv)))
    return dict(ok=bool(ok), frame_q=out_fq, sensor_q=out_sq[:S], sensor_has=out_sh[:S].astype(bool), frame_registered=out_reg.astype(bool),
                pair_valid=out_pv[:E].astype(bool))


def gp_adapter_solve(cam_q, cam_t, pt_offset, obs_cam, obs_undist, pt_xyz, cam_calibrated=None, cam_registered=None, pt_initialized=None,
                     pair_i=None, pair_j=None, pair_valid=None, pair_t=None, **options):
    """include/gsfm_glomap_adapter.hpp's GlobalPositioner::Solve (libgsfm, GPU) on the containers gp_build hands to the reference's
    class — same arguments.  Returns a dict: ok, initial_cost, final_cost, iterations, center [N,3], xyz [P,3], initialized [P]."""
    lib = load_dropin()
    pre, keep, (N, P, M, E), o = _gp_prefix(cam_q, cam_t, pt_offset, obs_cam, obs_undist, pt_xyz, cam_calibrated, cam_registered, pt_initialized,
                                             pair_i, pair_j, pair_valid, pair_t, options)
    rep, cen, xyz, ini = np.zeros(4), np.zeros((N, 3)), np.zeros((max(P, 1), 3)), np.zeros(max(P, 1), np.uint8)
    lib.ref_gp_adapter_solve.restype = C.c_int
    ok = lib.ref_gp_adapter_solve(*_cv(pre + [_p(rep), _p(cen), _p(xyz), _p(ini)]))
    return dict(ok=bool(ok), initial_cost=float(rep[0]), final_cost=float(rep[1]), iterations=int(rep[2]), center=cen, xyz=xyz[:P],
                initialized=ini[:P].astype(bool))


def ba_adapter_solve(cam_model, cam_params, frame_q, frame_t, image_frame, image_cam, pt_offset, obs_image, obs_xy, pt_xyz, rig_ref_cam=(0,),
                     frame_rig=None, sensor_rig=(), sensor_cam=(), sensor_pose=None, frame_has_pose=None, image_present=None, **options):
    """include/gsfm_glomap_adapter.hpp's BundleAdjuster::Solve (libgsfm, GPU) on the containers ba_build hands to the reference's
    class — same arguments.  Returns a dict: ok, initial_cost, final_cost, iterations, frame_q [F,4], frame_t [F,3], cam_params
    [K,8], xyz [P,3]."""
    lib = load_dropin()
    pre, keep, (F, I, K, S, P, M) = _ba_prefix(cam_model, cam_params, frame_q, frame_t, image_frame, image_cam, pt_offset, obs_image, obs_xy, pt_xyz,
                                               rig_ref_cam, frame_rig, sensor_rig, sensor_cam, sensor_pose, frame_has_pose, image_present, options)
    rep, fq, ft, cp, xyz = np.zeros(4), np.zeros((F, 4)), np.zeros((F, 3)), np.zeros((K, 16)), np.zeros((max(P, 1), 3))
    vp = C.c_void_p
    lib.ref_ba_adapter_solve.restype = C.c_int
    ok = lib.ref_ba_adapter_solve(*pre, vp(_p(rep)), vp(_p(fq)), vp(_p(ft)), vp(_p(cp)), vp(_p(xyz)))
    return dict(ok=bool(ok), initial_cost=float(rep[0]), final_cost=float(rep[1]), iterations=int(rep[2]), frame_q=fq, frame_t=ft,
                cam_params=cp, xyz=xyz[:P])


# ---- the reference's top-level controller, glomap/controllers/global_mapper.cc, on either set of estimators ----------------
LIB_MAPPER = HERE / "_ref" / "libref_dropin_mapper.so"
_lib_mapper = None


class _MapperOptions(C.Structure):
    _fields_ = [("num_iteration_bundle_adjustment", C.c_int), ("skip_rotation_averaging", C.c_int), ("skip_track_establishment", C.c_int),
                ("skip_global_positioning", C.c_int), ("skip_bundle_adjustment", C.c_int), ("min_num_view_per_track", C.c_int),
                ("optimize_intrinsics", C.c_int), ("gp_seed", C.c_uint), ("max_angle_error", C.c_double),
                ("max_reprojection_error", C.c_double), ("min_triangulation_angle", C.c_double), ("max_rotation_error", C.c_double),
                ("thres_inconsistency", C.c_double)]


def load_mapper():
    """oracle/_ref/libref_dropin_mapper.so (links glomap_amd/csrc/libgsfm.so): built where the reference tree and the built
    libgsfm.so exist, else the prebuilt file, else None."""
    global _lib_mapper
    if _lib_mapper is None:
        if (REFERENCE / "glomap" / "controllers" / "global_mapper.cc").exists() and (HERE.parent / "glomap_amd" / "csrc" / "libgsfm.so").exists():
            r = subprocess.run(["make", "-C", str(HERE), "-s", "ref_mapper", f"REF={REFERENCE}"], capture_output=True, text=True)
            if r.returncode != 0:
                print("[oracle/ref] make ref_mapper failed:\n" + r.stderr[-2000:], file=sys.stderr)
        if LIB_MAPPER.exists():
            _lib_mapper = C.CDLL(str(LIB_MAPPER))
            _lib_mapper.ref_mapper_solve.restype = C.c_int
    return _lib_mapper


def mapper_solve(which, cam_model, cam_params, image_cam, feat_offset, feat_xy, pair_i, pair_j, pair_q, pair_t, match_offset, match_f1,
                 match_f2, frame_q=None, frame_t=None, pair_weight=None, pair_valid=None, cam_has_prior=None, cap_tracks=None, **options):
    """glomap::GlobalMapper::Solve (controllers/global_mapper.cc:18-356), the reference's source compiled unmodified, with the stages
    outside SURVEY section 8 skipped by its own skip_* options: which = 0 on the reference's own estimators and processors (CPU),
    1 = RotationEstimator / GlobalPositioner / BundleAdjuster / UndistortImages of include/gsfm_glomap_adapter.hpp (libgsfm, GPU),
    2 = the track filters, the normaliser and the rotation filter on libgsfm as well.  Trivial frames (image i = frame i).
    cam_params [K,12] padded.  Returns a dict: ok, frame_q [N,4] (wxyz), frame_t [N,3], frame_registered [N], cam_params [K,12],
    pair_valid [E], num_tracks, num_observations, num_initialized, track_id / track_len / track_xyz (sorted by id)."""
    lib = load_mapper()
    i32 = lambda a: np.ascontiguousarray(a, dtype=np.int32)  # noqa: E731
    f64 = lambda a: np.ascontiguousarray(a, dtype=np.float64)  # noqa: E731
    i64 = lambda a: np.ascontiguousarray(a, dtype=np.int64)  # noqa: E731
    cm, cp, ic = i32(cam_model), f64(cam_params), i32(image_cam)
    K, N = len(cm), len(ic)
    assert cp.shape == (K, 12)
    fo, fxy = i64(feat_offset), f64(feat_xy)
    pi, pj, pq, pt = i32(pair_i), i32(pair_j), f64(pair_q), f64(pair_t)
    E = len(pi)
    mo, m1, m2 = i64(match_offset), i32(match_f1), i32(match_f2)
    fq = f64(np.tile([1.0, 0, 0, 0], (N, 1)) if frame_q is None else frame_q)
    ft = f64(np.zeros((N, 3)) if frame_t is None else frame_t)
    pw = f64(np.full(E, -1.0) if pair_weight is None else pair_weight)
    pv = np.ascontiguousarray(np.ones(E) if pair_valid is None else pair_valid, dtype=np.uint8)
    hp = np.ascontiguousarray(np.ones(K) if cam_has_prior is None else cam_has_prior, dtype=np.uint8)
    o = _MapperOptions(3, 0, 0, 0, 0, 3, 1, 1, -1.0, -1.0, -1.0, -1.0, -1.0)
    for k, v in options.items():
        setattr(o, k, type(getattr(o, k))(v))
    cap = int(len(m1) + 1 if cap_tracks is None else cap_tracks)
    out_q, out_t, out_reg = np.zeros((N, 4)), np.zeros((N, 3)), np.zeros(N, np.uint8)
    out_cp, out_pv, counts = np.zeros((K, 12)), np.zeros(max(E, 1), np.uint8), np.zeros(4, np.int64)
    tid, tlen, txyz = np.zeros(cap, np.uint64), np.zeros(cap, np.int32), np.zeros((cap, 3))
    vp = C.c_void_p
    ok = lib.ref_mapper_solve(C.c_int(int(which)), C.c_int(K), vp(_p(cm)), vp(_p(cp)), vp(_p(hp)), C.c_int(N), vp(_p(ic)), vp(_p(fo)), vp(_p(fxy)),
                              vp(_p(fq)), vp(_p(ft)), C.c_long(E), vp(_p(pi)), vp(_p(pj)), vp(_p(pq)), vp(_p(pt)), vp(_p(pw)), vp(_p(pv)),
                              vp(_p(mo)), vp(_p(m1)), vp(_p(m2)), C.byref(o), vp(_p(out_q)), vp(_p(out_t)), vp(_p(out_reg)), vp(_p(out_cp)),
                              vp(_p(out_pv)), vp(_p(counts)), C.c_long(cap), vp(_p(tid)), vp(_p(tlen)), vp(_p(txyz)))
    T = int(min(counts[0], cap))
    buf = C.create_string_buffer(4096)
    lib.ref_mapper_timings.restype = C.c_long
    lib.ref_mapper_timings(buf, C.c_long(4096))
    timings = {}
    for ln in buf.value.decode().splitlines():  # adapter entry point -> calls, seconds packing / inside libgsfm / unpacking
        name, calls, pk, cl, up, its, lin = ln.split()
        timings[name] = dict(calls=int(calls), pack=float(pk), call=float(cl), unpack=float(up), iterations=int(its), linear_iterations=int(lin))
    return dict(ok=ok == 1, rc=int(ok), seconds=counts[3] * 1e-6, adapter_timings=timings, frame_q=out_q, frame_t=out_t, frame_registered=out_reg.astype(bool), cam_params=out_cp,
                pair_valid=out_pv[:E].astype(bool), num_tracks=int(counts[0]), num_observations=int(counts[1]),
                num_initialized=int(counts[2]), track_id=tid[:T], track_len=tlen[:T], track_xyz=txyz[:T])
