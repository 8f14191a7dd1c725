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
        subprocess.run(["make", "-C", str(HERE), "-s", "ref", f"REF={REFERENCE}"], check=True)
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
