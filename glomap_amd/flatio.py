"""Reader / writer of the flat on-disk problem format (`*.gsfm`, layout in glomap_amd/csrc/dump.hpp) that libgsfm
writes at every gsfm_{ra,gp,ba}_solve when a dump directory is set (`Context.set_dump_dir`, or GSFM_DUMP_DIR).

    rec = flatio.load("dump/ba_0003.gsfm")
    rec.kind, rec.scalars, rec.options, rec.report, rec.status, rec.arrays["obs_xy"], rec.arrays["out_cam_q"]
    problem, options = flatio.to_problem(rec)          # flat.RaProblem / GpProblem / BaProblem + option dataclass

`save` writes the same layout from Python (tests, fixtures, problems produced elsewhere)."""
from __future__ import annotations

import json
import struct
from dataclasses import dataclass, field
from typing import Dict

import numpy as np

from . import estimators
from .flat import BaProblem, GpProblem, RaProblem

MAGIC = b"GSFMFLT1"
_DTYPES = {"f64": np.float64, "i32": np.int32, "i64": np.int64, "u8": np.uint8, "u32": np.uint32}
_NAMES = {np.dtype(v): k for k, v in _DTYPES.items()}


@dataclass
class FlatRecord:
    kind: str
    scalars: Dict[str, float] = field(default_factory=dict)
    options: Dict[str, float] = field(default_factory=dict)
    report: Dict[str, float] = field(default_factory=dict)
    status: int = 0
    arrays: Dict[str, np.ndarray] = field(default_factory=dict)


def load(path) -> FlatRecord:
    with open(path, "rb") as f:
        blob = f.read()
    if blob[:8] != MAGIC:
        raise ValueError(f"{path}: not a gsfm flat file")
    (hlen,) = struct.unpack("<Q", blob[8:16])
    hdr = json.loads(blob[16:16 + hlen].decode("utf-8"))
    base = 16 + hlen
    arrays = {}
    for a in hdr["arrays"]:
        dt = np.dtype(_DTYPES[a["dtype"]])
        n = int(np.prod(a["shape"], dtype=np.int64))
        arrays[a["name"]] = np.frombuffer(blob, dtype=dt, count=n, offset=base + a["offset"]).reshape(a["shape"]).copy()
    return FlatRecord(hdr["kind"], hdr.get("scalars", {}), hdr.get("options", {}), hdr.get("report", {}),
                      int(hdr.get("status", 0)), arrays)


def save(path, rec: FlatRecord) -> None:
    descr, blobs, off = [], [], 0
    for name, a in rec.arrays.items():
        a = np.ascontiguousarray(a)
        descr.append({"name": name, "dtype": _NAMES[a.dtype], "shape": list(a.shape), "offset": off, "nbytes": a.nbytes})
        pad = (-a.nbytes) % 64
        blobs.append(a.tobytes() + b"\0" * pad)
        off += a.nbytes + pad
    hdr = json.dumps({"format": "gsfm-flat", "version": 1, "kind": rec.kind, "status": rec.status, "scalars": rec.scalars,
                      "options": rec.options, "report": rec.report, "arrays": descr}).encode("utf-8")
    hdr += b" " * ((-(16 + len(hdr))) % 64)
    with open(path, "wb") as f:
        f.write(MAGIC + struct.pack("<Q", len(hdr)) + hdr + b"".join(blobs))


def _fill(obj, values: dict):
    for k, v in values.items():
        if hasattr(obj, k):
            cur = getattr(obj, k)
            setattr(obj, k, type(cur)(v) if isinstance(cur, (bool, int, float)) else v)
    return obj


def _lm(so: estimators.SolverOptions, o: dict):
    return _fill(so, {k: o[k] for k in ("max_num_iterations", "function_tolerance", "pcg_relative_tolerance", "pcg_max_iterations",
                                        "max_num_line_search_step_size_iterations")
                      if k in o})


def to_problem(rec: FlatRecord):
    """(flat problem, options dataclass) of a record, ready for estimators.{ra,gp,ba}_solve."""
    a, s, o = rec.arrays, rec.scalars, rec.options
    if rec.kind == "ra":
        E = len(a["edge_i"])
        p = RaProblem(int(s["num_nodes"]), a["edge_i"], a["edge_j"], a["edge_q"], a.get("edge_weight", np.ones(E)),
                      a.get("edge_ninl", np.ones(E, np.int32)), a["node_aa0"], int(s["fixed_node"]),
                      image_frame=a.get("image_frame"), image_cam=a.get("image_cam"),  # cam_from_rig rotations unknown
                      cam_aa0=a.get("cam_aa0", np.zeros((0, 3)) if "image_frame" in a else None),
                      node_gravity=a.get("node_gravity"))
        return p, _fill(estimators.RotationEstimatorOptions(), o)
    if rec.kind == "gp":
        M = len(a["obs_cam"])
        p = GpProblem(int(s["num_cams"]), len(a["pt_offset"]) - 1, a["pt_offset"], a["obs_cam"], a["obs_dir"],
                      a.get("obs_calibrated", np.ones(M, np.uint8)), a["cam_center"], a["pt_xyz"],
                      image_frame=a.get("image_frame"), image_offset=a.get("image_offset"),  # calibrated rigs
                      image_sensor=a.get("image_sensor"), image_sensor_rot=a.get("image_sensor_rot"),
                      sensor_center=a.get("sensor_center"),  # unknown cam_from_rig centres
                      pair_i=a.get("pair_i"), pair_j=a.get("pair_j"), pair_dir=a.get("pair_dir"),  # camera-to-camera constraints
                      cam_draw_order=a.get("cam_draw_order"), pt_draw_order=a.get("pt_draw_order"))
        opt = _fill(estimators.GlobalPositionerOptions(), o)
        _lm(opt.solver_options, o)
        return p, opt
    if rec.kind == "ba":
        p = BaProblem(num_cams=int(s["num_cams"]), num_pts=len(a["pt_offset"]) - 1, num_intr=int(s["num_intr"]),
                      pt_offset=a["pt_offset"], obs_cam=a["obs_cam"], obs_xy=a["obs_xy"],
                      cam_intr=a.get("cam_intr", np.zeros(int(s["num_cams"]), np.int32)), cam_q=a["cam_q"],
                      cam_t=a["cam_t"], pt_xyz=a["pt_xyz"], intr_model=a["intr_model"], intr_params=a["intr_params"],
                      fixed_cam=int(s["fixed_cam"]), image_frame=a.get("image_frame"),
                      image_cam_from_rig=a.get("image_cam_from_rig"), image_intr=a.get("image_intr"),
                      image_sensor=a.get("image_sensor"), sensor_cam_from_rig=a.get("sensor_cam_from_rig"))
        opt = _fill(estimators.BundleAdjusterOptions(), o)
        _lm(opt.solver_options, o)
        return p, opt
    raise ValueError(f"unknown kind {rec.kind!r}")


def from_problem(p, options=None) -> FlatRecord:
    """The record libgsfm would write for this problem (inputs only)."""
    if isinstance(p, RaProblem):
        arrs = dict(edge_i=np.asarray(p.edge_i, np.int32), edge_j=np.asarray(p.edge_j, np.int32), edge_q=np.asarray(p.edge_q, np.float64),
                    edge_weight=np.asarray(p.edge_weight, np.float64), edge_ninl=np.asarray(p.edge_ninl, np.int32),
                    node_aa0=np.asarray(p.node_aa0, np.float64))
        if p.image_frame is not None:
            arrs.update(image_frame=np.asarray(p.image_frame, np.int32), image_cam=np.asarray(p.image_cam, np.int32),
                        cam_aa0=np.asarray(p.cam_aa0, np.float64).reshape(-1, 3))
        if p.node_gravity is not None:
            arrs.update(node_gravity=np.asarray(p.node_gravity, np.uint8))
        return FlatRecord("ra", {"num_nodes": p.num_nodes, "fixed_node": p.fixed_node}, _opts(options), arrays=arrs)
    if isinstance(p, GpProblem):
        arrs = dict(pt_offset=np.asarray(p.pt_offset, np.int64), obs_cam=np.asarray(p.obs_cam, np.int32),
                    obs_dir=np.asarray(p.obs_dir, np.float64), obs_calibrated=np.asarray(p.obs_calibrated, np.uint8),
                    cam_center=np.asarray(p.cam_center, np.float64), pt_xyz=np.asarray(p.pt_xyz, np.float64))
        if p.image_frame is not None:
            arrs.update(image_frame=np.asarray(p.image_frame, np.int32), image_offset=np.asarray(p.image_offset, np.float64))
        if p.sensor_center is not None:
            arrs.update(image_sensor=np.asarray(p.image_sensor, np.int32), image_sensor_rot=np.asarray(p.image_sensor_rot, np.float64),
                        sensor_center=np.asarray(p.sensor_center, np.float64))
        if p.pair_i is not None:
            arrs.update(pair_i=np.asarray(p.pair_i, np.int32), pair_j=np.asarray(p.pair_j, np.int32),
                        pair_dir=np.asarray(p.pair_dir, np.float64))
        for name in ("cam_draw_order", "pt_draw_order"):
            if getattr(p, name, None) is not None:
                arrs[name] = np.asarray(getattr(p, name), np.int32)
        return FlatRecord("gp", {"num_cams": p.num_cams}, _opts(options), arrays=arrs)
    if isinstance(p, BaProblem):
        arrs = dict(pt_offset=np.asarray(p.pt_offset, np.int64), obs_cam=np.asarray(p.obs_cam, np.int32), obs_xy=np.asarray(p.obs_xy, np.float64),
                    cam_intr=np.asarray(p.cam_intr, np.int32), intr_model=np.asarray(p.intr_model, np.int32),
                    cam_q=np.asarray(p.cam_q, np.float64), cam_t=np.asarray(p.cam_t, np.float64), pt_xyz=np.asarray(p.pt_xyz, np.float64),
                    intr_params=np.asarray(p.intr_params, np.float64))
        if p.image_frame is not None:
            arrs.update(image_frame=np.asarray(p.image_frame, np.int32), image_intr=np.asarray(p.image_intr, np.int32),
                        image_cam_from_rig=np.asarray(p.image_cam_from_rig, np.float64))
        if p.sensor_cam_from_rig is not None:
            arrs.update(image_sensor=np.asarray(p.image_sensor, np.int32),
                        sensor_cam_from_rig=np.asarray(p.sensor_cam_from_rig, np.float64))
        return FlatRecord("ba", {"num_cams": p.num_cams, "num_intr": p.num_intr, "fixed_cam": p.fixed_cam}, _opts(options), arrays=arrs)
    raise TypeError(type(p))


def _opts(options) -> dict:
    if options is None:
        return {}
    out = {}
    for k, v in vars(options).items():
        if isinstance(v, (bool, int, float)):
            out[k] = float(v)
        elif isinstance(v, estimators.SolverOptions):
            out.update({kk: float(vv) for kk, vv in vars(v).items() if vv is not None})
    return out
