"""Multi-GPU sharding of the hot path (one process per GPU).

* GP / BA: observations are sharded by TRACK — every rank owns a contiguous, observation-balanced
  range of tracks with all their observations, so each 3x3 point block is eliminated locally; the
  camera-side vectors are replicated and the partial reduced-system products are summed with one
  all-reduce per PCG iteration (glomap_amd/csrc/cg.hpp).
* RA: edges are sharded in contiguous ranges; node vectors are replicated.

The transport is RCCL over xGMI (gsfm_comm_init).  `host_allreduce` builds the validation transport
(gsfm_comm_init_host) on top of a torch.distributed process group (gloo), used by the tests to run
several ranks on one device.
"""
from __future__ import annotations

import numpy as np


def shard_tracks(pt_offset: np.ndarray, rank: int, world: int):
    """Track range [lo, hi) of `rank`: contiguous, balanced by observation count."""
    M = int(pt_offset[-1])
    P = len(pt_offset) - 1
    lo = int(np.searchsorted(pt_offset, (M * rank) // world, side="left")) if rank > 0 else 0
    hi = int(np.searchsorted(pt_offset, (M * (rank + 1)) // world, side="left")) if rank + 1 < world else P
    return min(lo, P), min(max(hi, lo), P)


def shard_edges(num_edges: int, rank: int, world: int):
    """Edge range [lo, hi) of `rank`: contiguous, balanced by count."""
    return (num_edges * rank) // world, (num_edges * (rank + 1)) // world


def shard_gp_problem(p, rank: int, world: int):
    """This rank's shard of a flat GpProblem (host arrays): its tracks, all cameras."""
    from .flat import GpProblem

    lo, hi = shard_tracks(p.pt_offset, rank, world)
    o0, o1 = int(p.pt_offset[lo]), int(p.pt_offset[hi])
    s = GpProblem(
        num_cams=p.num_cams, num_pts=hi - lo, pt_offset=(p.pt_offset[lo : hi + 1] - o0).astype(np.int64),
        obs_cam=p.obs_cam[o0:o1].copy(), obs_dir=p.obs_dir[o0:o1].copy(), obs_calibrated=p.obs_calibrated[o0:o1].copy(),
        cam_center=p.cam_center.copy(), pt_xyz=p.pt_xyz[lo:hi].copy(),
        # camera-to-camera constraints live in camera space: every rank carries all of them (rank 0 adds their terms)
        pair_i=getattr(p, "pair_i", None), pair_j=getattr(p, "pair_j", None), pair_dir=getattr(p, "pair_dir", None),
    )
    # rig tables are per image / per sensor block: replicated like the cameras
    for name in ("image_frame", "image_offset", "image_sensor", "image_sensor_rot", "sensor_center"):
        if getattr(p, name, None) is not None:
            setattr(s, name, getattr(p, name))
    return s, (lo, hi)


def shard_ba_problem(p, rank: int, world: int):
    """This rank's shard of a flat BaProblem (host arrays): its tracks, all cameras and intrinsics."""
    import copy

    lo, hi = shard_tracks(p.pt_offset, rank, world)
    o0, o1 = int(p.pt_offset[lo]), int(p.pt_offset[hi])
    s = copy.copy(p)
    s.num_pts = hi - lo
    s.pt_offset = (p.pt_offset[lo : hi + 1] - o0).astype(np.int64)
    s.obs_cam = p.obs_cam[o0:o1].copy()
    s.obs_xy = p.obs_xy[o0:o1].copy()
    s.pt_xyz = p.pt_xyz[lo:hi].copy()
    return s, (lo, hi)


def shard_ra_problem(p, rank: int, world: int):
    """This rank's shard of a flat RaProblem (host arrays): its edges, all nodes."""
    import copy

    lo, hi = shard_edges(p.num_edges, rank, world)
    s = copy.copy(p)
    s.edge_i = p.edge_i[lo:hi].copy()
    s.edge_j = p.edge_j[lo:hi].copy()
    s.edge_q = p.edge_q[lo:hi].copy()
    s.edge_weight = p.edge_weight[lo:hi].copy()
    s.edge_ninl = p.edge_ninl[lo:hi].copy()
    return s, (lo, hi)


def host_allreduce(dist):
    """All-reduce callback for Context.comm_init_host on a torch.distributed (gloo) group."""
    import torch

    def fn(a: np.ndarray, op: int):
        t = torch.from_numpy(a)  # shares memory with the library's staging buffer
        dist.all_reduce(t, op=dist.ReduceOp.SUM if op == 0 else dist.ReduceOp.MAX)

    return fn
