"""ORACLE (test infrastructure only — never imported by the product path).

ctypes front end of oracle/liboracle_cpu.so, the multithreaded C++ restatement of the three
estimators (oracle/csrc/orc_{ra,gp,ba}.cc; built by oracle/Makefile, g++ + OpenMP, no third-party
code).  It exists for two jobs the numpy oracle is too slow for:

  * full-size oracle: the same algorithm as oracle/{ra,gp,ba}.py (LM decisions identical to
    oracle/lm.py, exact block elimination, reduced system solved to 1e-14) at the sizes of
    BASELINE.json configs[2] / configs[3], so that the `-m gpu` parity tests can compare poses at
    the sizes the tolerance is stated for;
  * cpu_baseline of bench.py: timed on all host cores of the GPU box, labelled "restated CPU
    oracle — not Ceres" (kind = "port").

tests/test_oracle_cpu.py cross-validates it against the numpy oracle on small problems.
Signatures follow oracle.gp.solve / oracle.ba.solve / oracle.ra.estimate_rotations.

parity unpinned: see oracle/__init__.py.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from dataclasses import dataclass
from pathlib import Path

import numpy as np

from . import ba as _ba
from . import gp as _gp
from . import lm as _lm
from . import ra as _ra

_DIR = Path(__file__).resolve().parent
_LIB = None


class _LmFields(C.Structure):
    _fields_ = [
        ("max_num_iterations", C.c_int32),
        ("function_tolerance", C.c_double),
        ("gradient_tolerance", C.c_double),
        ("parameter_tolerance", C.c_double),
        ("initial_trust_region_radius", C.c_double),
        ("max_trust_region_radius", C.c_double),
        ("min_trust_region_radius", C.c_double),
        ("min_relative_decrease", C.c_double),
        ("min_lm_diagonal", C.c_double),
        ("max_lm_diagonal", C.c_double),
        ("jacobi_scaling", C.c_int32),
        ("max_num_consecutive_invalid_steps", C.c_int32),
        ("pcg_relative_tolerance", C.c_double),
        ("pcg_max_iterations", C.c_int32),
        ("order", C.c_int32),
        ("verbose", C.c_int32),
        ("line_search", C.c_int32),
    ]


class _GpOptions(C.Structure):
    _fields_ = _LmFields._fields_ + [
        ("thres_loss_function", C.c_double),
        ("generate_random_positions", C.c_int32),
        ("generate_random_points", C.c_int32),
        ("generate_scales", C.c_int32),
        ("optimize_positions", C.c_int32),
        ("optimize_points", C.c_int32),
        ("optimize_scales", C.c_int32),
        ("min_num_view_per_track", C.c_int32),
        ("seed", C.c_uint32),
    ]


class _BaOptions(C.Structure):
    _fields_ = _LmFields._fields_ + [
        ("thres_loss_function", C.c_double),
        ("optimize_rotations", C.c_int32),
        ("optimize_translation", C.c_int32),
        ("optimize_intrinsics", C.c_int32),
        ("optimize_principal_point", C.c_int32),
        ("optimize_points", C.c_int32),
        ("min_num_view_per_track", C.c_int32),
        ("optimize_rig_poses", C.c_int32),
    ]


class _Report(C.Structure):
    _fields_ = [
        ("iterations", C.c_int32),
        ("successful_steps", C.c_int32),
        ("termination", C.c_int32),
        ("usable", C.c_int32),
        ("linear_iterations", C.c_int64),
        ("initial_cost", C.c_double),
        ("final_cost", C.c_double),
        ("max_linear_residual", C.c_double),
        ("seconds_total", C.c_double),
        ("seconds_linear", C.c_double),
        ("threads", C.c_int32),
        ("line_search_shrunk", C.c_int32),
    ]


class _RaOptions(C.Structure):
    _fields_ = [
        ("max_num_l1_iterations", C.c_int32),
        ("l1_step_convergence_threshold", C.c_double),
        ("max_num_irls_iterations", C.c_int32),
        ("irls_step_convergence_threshold", C.c_double),
        ("irls_loss_parameter_sigma", C.c_double),
        ("weight_type", C.c_int32),
        ("skip_initialization", C.c_int32),
        ("use_weight", C.c_int32),
        ("l1_admm_max_num_iterations", C.c_int32),
        ("l1_admm_rho", C.c_double),
        ("l1_admm_alpha", C.c_double),
        ("l1_admm_absolute_tolerance", C.c_double),
        ("l1_admm_relative_tolerance", C.c_double),
    ]


class _RaReport(C.Structure):
    _fields_ = [
        ("l1_iterations", C.c_int32),
        ("irls_iterations", C.c_int32),
        ("factorizations", C.c_int32),
        ("threads", C.c_int32),
        ("profile_entries", C.c_int64),
        ("seconds_total", C.c_double),
        ("seconds_factor", C.c_double),
    ]


@dataclass
class CpuSummary:
    """LmSummary-compatible result of the C++ solves (+ timing)."""

    iterations: int = 0
    successful_steps: int = 0
    linear_iterations: int = 0
    initial_cost: float = 0.0
    final_cost: float = 0.0
    termination: int = 1
    usable: bool = True
    max_linear_residual: float = 0.0
    seconds_total: float = 0.0
    seconds_linear: float = 0.0
    threads: int = 0
    line_search_shrunk: int = 0  # LM iterations whose step Ceres' projected line search shortened (GP)


def lib_path() -> Path:
    return _DIR / "liboracle_cpu.so"


def build(force: bool = False) -> Path:
    """make -C oracle (g++ -O3 -fopenmp); a prebuilt .so is used as is when the sources are not newer."""
    so = lib_path()
    srcs = list((_DIR / "csrc").glob("*"))
    stale = (not so.exists()) or any(s.stat().st_mtime > so.stat().st_mtime for s in srcs)
    if force or stale:
        subprocess.run(["make", "-C", str(_DIR), "-s"] + (["-B"] if force else []), check=True)
    return so


BUILD_FLAGS = "g++ -O3 -fopenmp, baseline x86-64"   # of the library load() returned (load_native() rewrites it)


def load_native():
    """bench.py's cpu_baseline legs only: rebuild the restatement with -march=native ON THIS BOX (oracle/_native/, never
    shipped: a native build of the build container may not run on the GPU box's host CPU) and make it the library every
    later call uses.  Falls back to the portable prebuilt library when g++ / make are missing."""
    global _LIB, BUILD_FLAGS
    try:
        subprocess.run(["make", "-C", str(_DIR), "-s", "native"], check=True, capture_output=True, timeout=300)
        flags = subprocess.run(["make", "-C", str(_DIR), "-s", "flags"], check=True, capture_output=True, text=True).stdout.strip()
        _LIB = None
        load(_DIR / "_native" / "liboracle_cpu.so")
        BUILD_FLAGS = flags + " -march=native (built on this box)"
    except Exception as e:  # noqa: BLE001
        BUILD_FLAGS = "g++ -O3 -fopenmp, baseline x86-64 (prebuilt; native rebuild failed: %s)" % type(e).__name__
        load()
    return _LIB


def load(path=None):
    global _LIB
    if _LIB is None:
        lib = C.CDLL(str(path or build()))
        lib.orc_gp_solve.restype = C.c_int
        lib.orc_gp_solve_pairs.restype = C.c_int
        lib.orc_ba_solve.restype = C.c_int
        lib.orc_ba_solve_wide.restype = C.c_int  # orc_ba_wide.cc: [K,16] intrinsics rows
        lib.orc_ra_solve.restype = C.c_int
        lib.orc_num_threads.restype = C.c_int
        _LIB = lib
    return _LIB


def lm_trace() -> np.ndarray:
    """[iterations, 7]: the LM iterations of the last gp_solve / ba_solve of this process — cost | radius | model change |
    candidate cost | line-search step size | accepted | linear iterations (the columns of the product's gsfm_ctx_lm_trace)."""
    lib = load()
    lib.orc_lm_trace.restype = C.c_int32
    rows = lib.orc_lm_trace(None, C.c_int32(0))
    out = np.zeros((max(rows, 0), 7))
    if rows > 0:
        lib.orc_lm_trace(out.ctypes.data_as(C.POINTER(C.c_double)), C.c_int32(rows))
    return out


def effective_cores() -> int:
    """Host cores this process may actually use: the scheduler affinity capped by the cgroup CPU quota (the GPU boxes
    expose 256 hardware threads but run inside a 16-CPU quota — an OpenMP team of 256 spinning threads in there is
    ~100x slower than a team of 16)."""
    import os

    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    quota = None
    try:  # cgroup v2
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            quota = float(q) / float(per)
    except Exception:
        try:  # cgroup v1
            q = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                quota = q / per
        except Exception:
            pass
    if quota is not None:
        n = min(n, max(1, int(quota)))
    return max(1, n)


def num_threads() -> int:
    """Threads the solves use by default (threads=0)."""
    return effective_cores()


def _threads(t: int) -> int:
    return int(t) if t and t > 0 else effective_cores()


def _p(a, t):
    return a.ctypes.data_as(C.POINTER(t))


def _fill_lm(o, lmo: _lm.LmOptions, pcg_tol, pcg_max, order, verbose):
    for name in ("max_num_iterations", "function_tolerance", "gradient_tolerance", "parameter_tolerance",
                 "initial_trust_region_radius", "max_trust_region_radius", "min_trust_region_radius",
                 "min_relative_decrease", "min_lm_diagonal", "max_lm_diagonal", "max_num_consecutive_invalid_steps"):
        setattr(o, name, getattr(lmo, name))
    o.jacobi_scaling = int(lmo.jacobi_scaling)
    o.pcg_relative_tolerance = pcg_tol
    o.pcg_max_iterations = pcg_max
    o.order = order
    o.verbose = int(verbose)
    o.line_search = int(getattr(lmo, "line_search", True))


class _deflate_env:
    """`deflate` of gp_solve / ba_solve: the gauge modes deflated from the reduced solves (ORC_DEFLATE, read per solve by
    orc_gp.cc / orc_ba.cc) — same systems, same tolerance, fewer operator applications; what libgsfm does by default."""

    def __init__(self, mode):
        self.mode = mode

    def __enter__(self):
        self.prev = os.environ.get("ORC_DEFLATE")
        if self.mode:
            os.environ["ORC_DEFLATE"] = str(int(self.mode))
        elif self.mode is not None:
            os.environ.pop("ORC_DEFLATE", None)

    def __exit__(self, *a):
        if self.mode is None:
            return
        if self.prev is None:
            os.environ.pop("ORC_DEFLATE", None)
        else:
            os.environ["ORC_DEFLATE"] = self.prev


def _summary(rep) -> CpuSummary:
    return CpuSummary(rep.iterations, rep.successful_steps, rep.linear_iterations, rep.initial_cost, rep.final_cost,
                      rep.termination, bool(rep.usable), rep.max_linear_residual, rep.seconds_total, rep.seconds_linear,
                      rep.threads, rep.line_search_shrunk)


def gp_solve(num_cams, pt_offset, obs_cam, obs_dir, obs_calibrated, cam_center, pt_xyz,
             options: _gp.GlobalPositionerOptions | None = None, threads: int = 0, pcg_tol: float = 1e-14,
             pcg_max: int = 20000, order: int = 0, verbose: bool = False, image_frame=None, image_offset=None,
             image_sensor=None, image_sensor_rot=None, sensor_center=None, deflate=None, pair_i=None, pair_j=None,
             pair_dir=None):
    """Same contract as oracle.gp.solve: returns (ok, cam_center [N,3], pt_xyz [P,3], CpuSummary); with unknown
    cam_from_rig centres the summary carries the estimates as summary.sensor_center.  Camera-to-camera constraints
    (options.constraint_type != ONLY_POINTS): pair_i / pair_j [E], pair_dir [E,3] as in oracle.gp.solve."""
    opt = options or _gp.GlobalPositionerOptions()
    lib = load()
    o = _GpOptions()
    _fill_lm(o, opt.lm, pcg_tol, pcg_max, order, verbose)
    o.thres_loss_function = opt.thres_loss_function
    for name in ("generate_random_positions", "generate_random_points", "generate_scales", "optimize_positions",
                 "optimize_points", "optimize_scales", "min_num_view_per_track", "seed"):
        setattr(o, name, int(getattr(opt, name)))
    off = np.ascontiguousarray(pt_offset, dtype=np.int64)
    cam = np.ascontiguousarray(obs_cam, dtype=np.int32)
    v = np.ascontiguousarray(obs_dir, dtype=np.float64)
    cal = None if obs_calibrated is None else np.ascontiguousarray(obs_calibrated, dtype=np.uint8)
    c = np.array(cam_center, dtype=np.float64, copy=True, order="C")
    X = np.array(pt_xyz, dtype=np.float64, copy=True, order="C")
    rep = _Report()
    imf = None if image_frame is None else np.ascontiguousarray(image_frame, dtype=np.int32)
    imo = None if image_frame is None else np.ascontiguousarray(image_offset, dtype=np.float64)
    ims = None if image_sensor is None else np.ascontiguousarray(image_sensor, dtype=np.int32)
    imr = None if image_sensor is None else np.ascontiguousarray(image_sensor_rot, dtype=np.float64)
    sc = None if image_sensor is None else np.array(sensor_center, dtype=np.float64, copy=True, order="C")
    ctype = int(getattr(opt, "constraint_type", 0))
    pi = pj = pd = None
    if ctype != 0:
        pi = np.ascontiguousarray(np.zeros(0) if pair_i is None else pair_i, dtype=np.int32)
        pj = np.ascontiguousarray(np.zeros(0) if pair_j is None else pair_j, dtype=np.int32)
        pd = np.ascontiguousarray(np.zeros((0, 3)) if pair_dir is None else pair_dir, dtype=np.float64)
    with _deflate_env(deflate):
        rc = lib.orc_gp_solve_pairs(
            C.c_int32(int(num_cams)), C.c_int64(off.shape[0] - 1), _p(off, C.c_int64), _p(cam, C.c_int32),
            _p(v, C.c_double), None if cal is None else _p(cal, C.c_uint8), C.byref(o), _p(c, C.c_double),
            _p(X, C.c_double), C.byref(rep), C.c_int32(_threads(threads)),
            None if imf is None else _p(imf, C.c_int32), None if imo is None else _p(imo, C.c_double),
            C.c_int32(0 if sc is None else sc.shape[0]), None if ims is None else _p(ims, C.c_int32),
            None if imr is None else _p(imr, C.c_double), None if sc is None else _p(sc, C.c_double),
            C.c_int32(ctype), C.c_double(float(getattr(opt, "constraint_reweight_scale", 1.0))),
            C.c_int64(0 if pi is None else pi.shape[0]), None if pi is None or pi.size == 0 else _p(pi, C.c_int32),
            None if pj is None or pj.size == 0 else _p(pj, C.c_int32), None if pd is None or pd.size == 0 else _p(pd, C.c_double))
    s = _summary(rep)
    if rc == -5:
        s.usable = False
    if sc is not None:
        s.sensor_center = sc
    return rc == 0, c, X, s


def ba_solve(num_cams, pt_offset, obs_cam, obs_xy, cam_intr, intr_model, fixed_cam, cam_q, cam_t, pt_xyz, intr_params,
             options: _ba.BundleAdjusterOptions | None = None, threads: int = 0, pcg_tol: float = 1e-14,
             pcg_max: int = 20000, order: int = 0, verbose: bool = False, image_frame=None, image_cam_from_rig=None,
             image_intr=None, image_sensor=None, sensor_cam_from_rig=None, deflate=None):
    """Same contract as oracle.ba.solve: returns (ok, q [N,4], t [N,3], X [P,3], intr [K,8], CpuSummary); with sensor
    blocks (options.optimize_rig_poses) the summary carries their result as summary.sensor_cam_from_rig."""
    opt = options or _ba.BundleAdjusterOptions()
    lib = load()
    o = _BaOptions()
    _fill_lm(o, opt.lm, pcg_tol, pcg_max, order, verbose)
    o.thres_loss_function = opt.thres_loss_function
    for name in ("optimize_rotations", "optimize_translation", "optimize_intrinsics", "optimize_principal_point",
                 "optimize_points", "min_num_view_per_track", "optimize_rig_poses"):
        setattr(o, name, int(getattr(opt, name)))
    off = np.ascontiguousarray(pt_offset, dtype=np.int64)
    cam = np.ascontiguousarray(obs_cam, dtype=np.int32)
    xy = np.ascontiguousarray(obs_xy, dtype=np.float64)
    ci = np.zeros(int(num_cams), np.int32) if cam_intr is None else np.ascontiguousarray(cam_intr, dtype=np.int32)
    imf = None if image_frame is None else np.ascontiguousarray(image_frame, dtype=np.int32)
    imc = None if image_frame is None else np.ascontiguousarray(image_cam_from_rig, dtype=np.float64)
    imi = None if image_frame is None else np.ascontiguousarray(image_intr, dtype=np.int32)
    ims = None if image_sensor is None else np.ascontiguousarray(image_sensor, dtype=np.int32)
    sen = None if image_sensor is None else np.array(sensor_cam_from_rig, dtype=np.float64, copy=True, order="C")
    mdl = np.ascontiguousarray(intr_model, dtype=np.int32)
    q = np.array(cam_q, dtype=np.float64, copy=True, order="C")
    t = np.array(cam_t, dtype=np.float64, copy=True, order="C")
    X = np.array(pt_xyz, dtype=np.float64, copy=True, order="C")
    intr = np.array(intr_params, dtype=np.float64, copy=True, order="C")
    rep = _Report()
    if intr.ndim != 2 or intr.shape[1] not in (_ba.MAXP, _ba.MAXP_WIDE):
        raise ValueError("intr_params must be [K, 8] or [K, 16]")
    # the width of the intrinsics rows picks the unit: orc_ba.cc (8) or orc_ba_wide.cc (16: the 12 / 16-parameter camera models)
    entry = lib.orc_ba_solve if intr.shape[1] == _ba.MAXP else lib.orc_ba_solve_wide
    with _deflate_env(deflate):
        rc = entry(C.c_int32(int(num_cams)), C.c_int32(mdl.shape[0]), C.c_int32(int(fixed_cam)),
                              C.c_int64(off.shape[0] - 1), _p(off, C.c_int64), _p(cam, C.c_int32), _p(xy, C.c_double),
                              _p(ci, C.c_int32), _p(mdl, C.c_int32), C.byref(o), _p(q, C.c_double), _p(t, C.c_double),
                              _p(X, C.c_double), _p(intr, C.c_double), C.byref(rep), C.c_int32(_threads(threads)),
                              None if imf is None else _p(imf, C.c_int32), None if imc is None else _p(imc, C.c_double),
                              None if imi is None else _p(imi, C.c_int32), C.c_int32(0 if sen is None else sen.shape[0]),
                              None if ims is None else _p(ims, C.c_int32), None if sen is None else _p(sen, C.c_double))
    s = _summary(rep)
    if rc == -5:
        s.usable = False
    if sen is not None and opt.optimize_rig_poses:
        s.sensor_cam_from_rig = sen
    return rc == 0, q, t, X, intr, s


def ra_estimate_rotations(num_nodes, edge_i, edge_j, edge_q, edge_weight, edge_ninl, node_aa0, fixed_node=0,
                          options: _ra.RotationEstimatorOptions | None = None, threads: int = 0, report: dict | None = None):
    """Same contract as oracle.ra.estimate_rotations: returns (ok, rot_aa [N,3]).  Raises NotImplementedError
    when the graph's reverse-Cuthill-McKee profile is too large for the skyline factor (use oracle.ra then)."""
    opt = options or _ra.RotationEstimatorOptions()
    lib = load()
    o = _RaOptions()
    for name, _ in _RaOptions._fields_:
        setattr(o, name, type(getattr(o, name))(getattr(opt, name)))
    ei = np.ascontiguousarray(edge_i, dtype=np.int32)
    ej = np.ascontiguousarray(edge_j, dtype=np.int32)
    eq = np.ascontiguousarray(edge_q, dtype=np.float64)
    E = ei.shape[0]
    ew = np.ones(E) if edge_weight is None else np.ascontiguousarray(edge_weight, dtype=np.float64)
    ninl = np.ones(E, np.int32) if edge_ninl is None else np.ascontiguousarray(edge_ninl, dtype=np.int32)
    rot = np.array(node_aa0, dtype=np.float64, copy=True, order="C")
    rep = _RaReport()
    rc = lib.orc_ra_solve(C.c_int32(int(num_nodes)), C.c_int64(E), _p(ei, C.c_int32), _p(ej, C.c_int32), _p(eq, C.c_double),
                          _p(ew, C.c_double), _p(ninl, C.c_int32), C.c_int32(int(fixed_node)), C.byref(o),
                          _p(rot, C.c_double), C.byref(rep), C.c_int32(_threads(threads)))
    if report is not None:
        report.update(l1_iterations=rep.l1_iterations, irls_iterations=rep.irls_iterations,
                      factorizations=rep.factorizations, threads=rep.threads, profile_entries=rep.profile_entries,
                      seconds_total=rep.seconds_total, seconds_factor=rep.seconds_factor)
    if rc == -7:
        raise NotImplementedError(f"skyline profile too large ({rep.profile_entries} entries)")
    return rc == 0, rot
