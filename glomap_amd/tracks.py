"""Host-side mirror of the reference's track establishment and view-graph component selection, on the C ABI
(include/gsfm.h, glomap_amd/csrc/tracks.hip):

  TrackEngine(view_graph, images, options).EstablishFullTracks / FindTracksForProblem
      (glomap/controllers/track_establishment.h:26-61), ViewGraph.KeepLargestConnectedComponents
      (glomap/scene/view_graph.h).

Flat level only: the match graph as in gsfm_match_graph (numpy on the host or DeviceArrays in HBM), tracks as CSR
(`TrackSet`).  Nothing here computes on the CPU."""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass
from typing import Optional

import numpy as np

from . import _lib
from .estimators import _h, _mem_of, default_context


@dataclass
class TrackEstablishmentOptions:
    """glomap/controllers/track_establishment.h:9-24."""

    thres_inconsistency: float = 10.0
    min_num_tracks_per_view: int = -1
    min_num_view_per_track: int = 3
    max_num_view_per_track: int = 100
    max_num_tracks: int = 10000000

    def to_c(self) -> _lib.TrackOptionsC:
        c = _lib.TrackOptionsC()
        c.thres_inconsistency = float(self.thres_inconsistency)
        c.min_num_tracks_per_view = int(self.min_num_tracks_per_view)
        c.min_num_view_per_track = int(self.min_num_view_per_track)
        c.max_num_view_per_track = int(self.max_num_view_per_track)
        c.max_num_tracks = int(self.max_num_tracks)
        return c


@dataclass
class MatchGraph:
    """Inlier matches of the valid image pairs + the feature positions (layout of gsfm_match_graph)."""

    num_images: int
    feat_offset: np.ndarray  # [I+1] int64
    feat_xy: np.ndarray  # [F,2] float64
    pair_image1: np.ndarray  # [P] int32
    pair_image2: np.ndarray  # [P] int32
    pair_offset: np.ndarray  # [P+1] int64
    match_feat1: np.ndarray  # [M] uint32
    match_feat2: np.ndarray  # [M] uint32
    pair_valid: Optional[np.ndarray] = None  # [P] uint8

    @classmethod
    def from_dict(cls, d):
        return cls(**{k: d[k] for k in ("num_images", "feat_offset", "feat_xy", "pair_image1", "pair_image2", "pair_offset",
                                        "match_feat1", "match_feat2", "pair_valid") if k in d})

    def to_device(self, ctx):
        f = lambda a, dt: None if a is None else _lib.DeviceArray.from_numpy(ctx, np.ascontiguousarray(a, dtype=dt))
        return MatchGraph(self.num_images, f(self.feat_offset, np.int64), f(self.feat_xy, np.float64),
                          f(self.pair_image1, np.int32), f(self.pair_image2, np.int32), f(self.pair_offset, np.int64),
                          f(self.match_feat1, np.uint32), f(self.match_feat2, np.uint32), f(self.pair_valid, np.uint8))


@dataclass
class TrackSet:
    """CSR set of tracks (layout of gsfm_track_set)."""

    track_id: np.ndarray  # [T] int64: image << 32 | feature of the smallest member
    track_offset: np.ndarray  # [T+1] int64
    obs_image: np.ndarray  # [Mo] int32
    obs_feature: np.ndarray  # [Mo] uint32

    @property
    def num_tracks(self):
        return int(self.track_offset.shape[0]) - 1

    @property
    def num_obs(self):
        return int(self.obs_image.shape[0])


def _track_set_c(ts: TrackSet, keep: list) -> _lib.TrackSetC:
    tid, off = _h(ts.track_id, np.int64), _h(ts.track_offset, np.int64)
    img, ft = _h(ts.obs_image, np.int32), _h(ts.obs_feature, np.uint32)
    keep.extend([tid, off, img, ft])
    c = _lib.TrackSetC()
    c.mem = _mem_of(tid, off, img, ft)
    c.num_tracks = int(off.shape[0]) - 1
    c.num_obs = int(img.shape[0])
    c.track_id, c.track_offset, c.obs_image, c.obs_feature = _lib.ptr(tid), _lib.ptr(off), _lib.ptr(img), _lib.ptr(ft)
    return c


class TrackEngine:
    """glomap/controllers/track_establishment.h:26-61.  The established tracks stay in HBM inside the context;
    `fetch=False` skips the copy back (a device-resident pipeline only needs the counts)."""

    def __init__(self, graph: MatchGraph, options: Optional[TrackEstablishmentOptions] = None, ctx=None):
        self.graph = graph
        self.options = options or TrackEstablishmentOptions()
        self.ctx = ctx or default_context()
        self.num_discarded = 0
        self._counts = {}

    def _fetch(self, which: int, device: bool = False) -> TrackSet:
        T, Mo = self._counts[which]
        if device:
            mk = lambda n, dt: _lib.DeviceArray(self.ctx, (max(n, 1),), dt)
            mem = _lib.GSFM_MEM_DEVICE
        else:
            mk = lambda n, dt: np.zeros(max(n, 1), dtype=dt)
            mem = _lib.GSFM_MEM_HOST
        tid, off, img, ft = mk(T, np.int64), mk(T + 1, np.int64), mk(Mo, np.int32), mk(Mo, np.uint32)
        c = _lib.TrackSetC()
        c.mem = mem
        c.track_id, c.track_offset, c.obs_image, c.obs_feature = _lib.ptr(tid), _lib.ptr(off), _lib.ptr(img), _lib.ptr(ft)
        rc = self.ctx.lib.gsfm_tracks_fetch(self.ctx.handle, which, C.byref(c))
        if rc != 0:
            raise _lib.GsfmError(rc, "gsfm_tracks_fetch")
        if device:
            return TrackSet(tid, off, img, ft)
        return TrackSet(tid[:T], off[:T + 1], img[:Mo], ft[:Mo])

    def EstablishFullTracks(self, fetch: bool = True):
        """Returns the TrackSet (or, with fetch=False, the number of tracks like the reference does)."""
        g = self.graph
        arrs = [_h(g.feat_offset, np.int64), _h(g.feat_xy, np.float64), _h(g.pair_image1, np.int32), _h(g.pair_image2, np.int32),
                _h(g.pair_valid, np.uint8), _h(g.pair_offset, np.int64), _h(g.match_feat1, np.uint32), _h(g.match_feat2, np.uint32)]
        c = _lib.MatchGraphC()
        c.mem = _mem_of(*[a for a in arrs if a is not None])
        c.num_images = int(g.num_images)
        c.num_pairs = int(arrs[2].shape[0])
        (c.feat_offset, c.feat_xy, c.pair_image1, c.pair_image2, c.pair_valid, c.pair_offset, c.match_feat1,
         c.match_feat2) = [_lib.ptr(a) for a in arrs]
        o = self.options.to_c()
        nt, no, nd = C.c_int64(0), C.c_int64(0), C.c_int64(0)
        rc = self.ctx.lib.gsfm_tracks_establish(self.ctx.handle, C.byref(c), C.byref(o), C.byref(nt), C.byref(no), C.byref(nd))
        if rc != 0:
            raise _lib.GsfmError(rc, "gsfm_tracks_establish")
        self.num_discarded = nd.value
        self._counts[_lib.GSFM_TRACKS_FULL] = (nt.value, no.value)
        return self._fetch(_lib.GSFM_TRACKS_FULL) if fetch else nt.value

    def FindTracksForProblem(self, image_registered, tracks_full: Optional[TrackSet] = None, fetch: bool = True):
        """tracks_full=None selects from the tracks EstablishFullTracks left in HBM."""
        reg = _h(image_registered, np.uint8)
        keep: list = []
        full_c = _track_set_c(tracks_full, keep) if tracks_full is not None else None
        o = self.options.to_c()
        nt, no = C.c_int64(0), C.c_int64(0)
        rc = self.ctx.lib.gsfm_tracks_select(self.ctx.handle, C.byref(full_c) if full_c is not None else None,
                                             int(reg.shape[0]), _lib.ptr(reg), _mem_of(reg), C.byref(o), C.byref(nt), C.byref(no))
        if rc != 0:
            raise _lib.GsfmError(rc, "gsfm_tracks_select")
        self._counts[_lib.GSFM_TRACKS_SELECTED] = (nt.value, no.value)
        return self._fetch(_lib.GSFM_TRACKS_SELECTED) if fetch else nt.value


def KeepLargestConnectedComponents(num_nodes, edge_i, edge_j, edge_valid, node_num_images=None, ctx=None):
    """view_graph.cc:56-97 on dense node indices.  Returns (node_registered [N] uint8, edge_valid' [E] uint8,
    number of registered images); inputs are left untouched.  0 images = no valid edge (masks are then None / a copy)."""
    ctx = ctx or default_context()
    ei, ej = _h(edge_i, np.int32), _h(edge_j, np.int32)
    ev = _h(edge_valid, np.uint8)
    ni = _h(node_num_images, np.int32)
    mem = _mem_of(ei, ej, ev)
    ev2 = ev.copy() if isinstance(ev, np.ndarray) else ev.clone()
    reg = np.zeros(num_nodes, dtype=np.uint8) if mem == _lib.GSFM_MEM_HOST else _lib.DeviceArray(ctx, (num_nodes,), np.uint8)
    n = C.c_int64(0)
    rc = ctx.lib.gsfm_keep_largest_connected_component(ctx.handle, mem, int(num_nodes), int(ei.shape[0]), _lib.ptr(ei), _lib.ptr(ej),
                                                       _lib.ptr(ev2), _lib.ptr(ni), _lib.ptr(reg), C.byref(n))
    if rc != 0:
        raise _lib.GsfmError(rc, "gsfm_keep_largest_connected_component")
    if n.value == 0:  # no valid edge at all (nothing written), or a component whose frames hold no image
        any_reg = bool((reg if isinstance(reg, np.ndarray) else reg.numpy()).any())
        if not any_reg:
            return None, ev2, 0
    return reg, ev2, n.value
