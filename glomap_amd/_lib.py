"""ctypes binding of libgsfm.so (include/gsfm.h).  No fallback: a missing library or a missing GPU is
a hard error — the product path never runs on the CPU."""
from __future__ import annotations

import ctypes as C
import os
from pathlib import Path

import numpy as np

LIB_PATH = Path(__file__).resolve().parent / "csrc" / "libgsfm.so"

GSFM_MEM_HOST = 0
GSFM_MEM_DEVICE = 1
GSFM_COMM_ID_BYTES = 128
GSFM_CAMERA_MAX_PARAMS = 8
GSFM_CAMERA_MAX_PARAMS_WIDE = 16

STATUS_NAMES = {
    0: "GSFM_OK",
    -1: "GSFM_ERR_INVALID_ARGUMENT",
    -2: "GSFM_ERR_HIP",
    -3: "GSFM_ERR_NO_DEVICE",
    -4: "GSFM_ERR_NUMERICAL",
    -5: "GSFM_ERR_EMPTY_PROBLEM",
    -6: "GSFM_ERR_NOT_USABLE",
    -7: "GSFM_ERR_UNSUPPORTED",
    -8: "GSFM_ERR_COMM",
}


class GsfmError(RuntimeError):
    def __init__(self, status: int, where: str):
        self.status = status
        super().__init__(f"{where}: {STATUS_NAMES.get(status, status)}")


class Report(C.Structure):
    _fields_ = [
        ("iterations", C.c_int32),
        ("iterations_l1", C.c_int32),
        ("iterations_irls", C.c_int32),
        ("successful_steps", C.c_int32),
        ("linear_iterations", C.c_int64),
        ("initial_cost", C.c_double),
        ("final_cost", C.c_double),
        ("termination", C.c_int32),
        ("hip_error", C.c_int32),
        ("seconds_total", C.c_double),
        ("seconds_solve", C.c_double),
        ("last_step_norm", C.c_double),
        ("line_search_trials", C.c_int32),
        ("line_search_shrunk", C.c_int32),
    ]

    def as_dict(self):
        return {f: getattr(self, f) for f, _ in self._fields_}


class RaOptions(C.Structure):
    _fields_ = [
        ("max_num_l1_iterations", C.c_int32),
        ("l1_step_convergence_threshold", C.c_double),
        ("max_num_irls_iterations", C.c_int32),
        ("irls_step_convergence_threshold", C.c_double),
        ("irls_loss_parameter_sigma", C.c_double),
        ("weight_type", C.c_int32),
        ("skip_initialization", C.c_int32),
        ("use_weight", C.c_int32),
        ("use_gravity", C.c_int32),
        ("l1_admm_max_num_iterations", C.c_int32),
        ("l1_admm_rho", C.c_double),
        ("l1_admm_alpha", C.c_double),
        ("l1_admm_absolute_tolerance", C.c_double),
        ("l1_admm_relative_tolerance", C.c_double),
        ("pcg_relative_tolerance", C.c_double),
        ("pcg_max_iterations", C.c_int32),
        ("force_iterative", C.c_int32),
        ("pcg_relative_tolerance_admm", C.c_double),
    ]


class RaProblemC(C.Structure):
    _fields_ = [
        ("mem", C.c_int32),
        ("num_nodes", C.c_int32),
        ("num_edges", C.c_int64),
        ("edge_i", C.c_void_p),
        ("edge_j", C.c_void_p),
        ("edge_q", C.c_void_p),
        ("edge_weight", C.c_void_p),
        ("edge_ninl", C.c_void_p),
        ("fixed_node", C.c_int32),
        ("num_images", C.c_int32),
        ("image_frame", C.c_void_p),
        ("image_cam", C.c_void_p),
        ("num_cams", C.c_int32),
        ("cam_rot_aa", C.c_void_p),
        ("node_gravity", C.c_void_p),
    ]


class LmOptions(C.Structure):
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
        ("max_num_line_search_step_size_iterations", C.c_int32),
    ]


class GpOptions(C.Structure):
    _fields_ = [
        ("lm", LmOptions),
        ("thres_loss_function", C.c_double),
        ("generate_random_positions", C.c_int32),
        ("generate_random_points", C.c_int32),
        ("generate_scales", C.c_int32),
        ("optimize_positions", C.c_int32),
        ("optimize_points", C.c_int32),
        ("optimize_scales", C.c_int32),
        ("min_num_view_per_track", C.c_int32),
        ("seed", C.c_uint32),
        ("constraint_type", C.c_int32),
        ("constraint_reweight_scale", C.c_double),
        ("rand_vector_order", C.c_int32),
    ]


class GpProblemC(C.Structure):
    _fields_ = [
        ("mem", C.c_int32),
        ("num_cams", C.c_int32),
        ("num_pts", C.c_int64),
        ("num_obs", C.c_int64),
        ("pt_offset", C.c_void_p),
        ("obs_cam", C.c_void_p),
        ("obs_dir", C.c_void_p),
        ("obs_calibrated", C.c_void_p),
        ("num_images", C.c_int32),
        ("image_frame", C.c_void_p),
        ("image_offset", C.c_void_p),
        ("num_sensors", C.c_int32),
        ("image_sensor", C.c_void_p),
        ("image_sensor_rot", C.c_void_p),
        ("sensor_center", C.c_void_p),
        ("num_pairs", C.c_int64),
        ("pair_i", C.c_void_p),
        ("pair_j", C.c_void_p),
        ("pair_dir", C.c_void_p),
        ("cam_draw_order", C.c_void_p),
        ("pt_draw_order", C.c_void_p),
    ]


class BaOptions(C.Structure):
    _fields_ = [
        ("lm", LmOptions),
        ("thres_loss_function", C.c_double),
        ("optimize_rotations", C.c_int32),
        ("optimize_translation", C.c_int32),
        ("optimize_intrinsics", C.c_int32),
        ("optimize_principal_point", C.c_int32),
        ("optimize_points", C.c_int32),
        ("min_num_view_per_track", C.c_int32),
        ("optimize_rig_poses", C.c_int32),
    ]


class BaProblemC(C.Structure):
    _fields_ = [
        ("mem", C.c_int32),
        ("num_cams", C.c_int32),
        ("num_intr", C.c_int32),
        ("fixed_cam", C.c_int32),
        ("num_pts", C.c_int64),
        ("num_obs", C.c_int64),
        ("pt_offset", C.c_void_p),
        ("obs_cam", C.c_void_p),
        ("obs_xy", C.c_void_p),
        ("cam_intr", C.c_void_p),
        ("intr_model", C.c_void_p),
        ("num_images", C.c_int32),
        ("image_frame", C.c_void_p),
        ("image_cam_from_rig", C.c_void_p),
        ("image_intr", C.c_void_p),
        ("num_sensors", C.c_int32),
        ("image_sensor", C.c_void_p),
        ("sensor_cam_from_rig", C.c_void_p),
        ("intr_stride", C.c_int32),
    ]


HOST_ALLREDUCE_FN = C.CFUNCTYPE(C.c_int, C.POINTER(C.c_double), C.c_int64, C.c_int, C.c_void_p)

class SceneViewC(C.Structure):
    _fields_ = [
        ("mem", C.c_int32),
        ("num_cams", C.c_int32),
        ("num_pts", C.c_int64),
        ("num_obs", C.c_int64),
        ("pt_offset", C.c_void_p),
        ("obs_cam", C.c_void_p),
        ("obs_undist", C.c_void_p),
        ("obs_xy", C.c_void_p),
        ("cam_q", C.c_void_p),
        ("cam_t", C.c_void_p),
        ("pt_xyz", C.c_void_p),
        ("cam_calibrated", C.c_void_p),
        ("num_intr", C.c_int32),
        ("cam_intr", C.c_void_p),
        ("intr_model", C.c_void_p),
        ("intr_params", C.c_void_p),
        ("intr_stride", C.c_int32),
    ]


class MatchGraphC(C.Structure):
    _fields_ = [
        ("mem", C.c_int32),
        ("num_images", C.c_int32),
        ("feat_offset", C.c_void_p),
        ("feat_xy", C.c_void_p),
        ("num_pairs", C.c_int64),
        ("pair_image1", C.c_void_p),
        ("pair_image2", C.c_void_p),
        ("pair_valid", C.c_void_p),
        ("pair_offset", C.c_void_p),
        ("match_feat1", C.c_void_p),
        ("match_feat2", C.c_void_p),
    ]


class TrackOptionsC(C.Structure):
    _fields_ = [
        ("thres_inconsistency", C.c_double),
        ("min_num_tracks_per_view", C.c_int32),
        ("min_num_view_per_track", C.c_int32),
        ("max_num_view_per_track", C.c_int32),
        ("max_num_tracks", C.c_int32),
    ]


class TrackSetC(C.Structure):
    _fields_ = [
        ("mem", C.c_int32),
        ("num_tracks", C.c_int64),
        ("num_obs", C.c_int64),
        ("track_id", C.c_void_p),
        ("track_offset", C.c_void_p),
        ("obs_image", C.c_void_p),
        ("obs_feature", C.c_void_p),
    ]


GSFM_TRACKS_FULL, GSFM_TRACKS_SELECTED = 0, 1

_lib = None


def load():
    """Loads libgsfm.so (building nothing: run __graft_entry__.build() / glomap_amd.build first)."""
    global _lib
    if _lib is not None:
        return _lib
    if not LIB_PATH.exists():
        raise RuntimeError(
            f"{LIB_PATH} is missing — build it with `python -m glomap_amd.build` (hipcc, gfx950). "
            "There is no CPU fallback."
        )
    lib = C.CDLL(str(LIB_PATH))
    vp, ip, dp = C.c_void_p, C.c_int, C.POINTER(C.c_double)
    lib.gsfm_version.restype = ip
    lib.gsfm_status_string.restype = C.c_char_p
    lib.gsfm_status_string.argtypes = [ip]
    lib.gsfm_ctx_create.restype = ip
    lib.gsfm_ctx_create.argtypes = [ip, C.POINTER(vp)]
    lib.gsfm_ctx_destroy.restype = None
    lib.gsfm_ctx_destroy.argtypes = [vp]
    lib.gsfm_ctx_stream.restype = vp
    lib.gsfm_ctx_stream.argtypes = [vp]
    lib.gsfm_ctx_device_name.restype = ip
    lib.gsfm_ctx_device_name.argtypes = [vp, C.c_char_p, C.c_size_t]
    lib.gsfm_device_alloc.restype = ip
    lib.gsfm_device_alloc.argtypes = [vp, C.c_size_t, C.POINTER(vp)]
    lib.gsfm_device_free.restype = ip
    lib.gsfm_device_free.argtypes = [vp, vp]
    for name in ("gsfm_memcpy_h2d", "gsfm_memcpy_d2h", "gsfm_memcpy_d2d"):
        getattr(lib, name).restype = ip
        getattr(lib, name).argtypes = [vp, vp, vp, C.c_size_t]
    lib.gsfm_ctx_synchronize.restype = ip
    lib.gsfm_ctx_synchronize.argtypes = [vp]
    lib.gsfm_ctx_profile_enable.restype = ip
    lib.gsfm_ctx_profile_enable.argtypes = [vp, ip]
    lib.gsfm_ctx_profile_read.restype = ip
    lib.gsfm_ctx_profile_read.argtypes = [vp, ip, C.POINTER(C.c_int64), dp]
    lib.gsfm_ctx_set_knob.restype = ip
    lib.gsfm_ctx_set_knob.argtypes = [vp, ip, ip]
    lib.gsfm_ctx_stats.restype = ip
    lib.gsfm_ctx_stats.argtypes = [vp, C.POINTER(C.c_int64), ip, ip]
    lib.gsfm_ctx_lm_trace.restype = ip
    lib.gsfm_ctx_lm_trace.argtypes = [vp, dp, C.c_int32]
    lib.gsfm_comm_unique_id.restype = ip
    lib.gsfm_comm_unique_id.argtypes = [C.c_char_p]
    lib.gsfm_comm_init.restype = ip
    lib.gsfm_comm_init.argtypes = [vp, C.c_char_p, ip, ip]
    lib.gsfm_comm_selftest.restype = ip
    lib.gsfm_comm_selftest.argtypes = [vp, dp]
    lib.gsfm_selftest_mt19937.restype = ip
    lib.gsfm_selftest_mt19937.argtypes = [C.c_uint32, C.c_uint64, C.c_int64, C.c_double, dp, dp]
    lib.gsfm_comm_destroy.restype = ip
    lib.gsfm_comm_destroy.argtypes = [vp]
    lib.gsfm_comm_init_host.restype = ip
    lib.gsfm_comm_init_host.argtypes = [vp, HOST_ALLREDUCE_FN, vp, ip, ip]
    lib.gsfm_ctx_last_error.restype = C.c_char_p
    lib.gsfm_ctx_last_error.argtypes = [vp]
    lib.gsfm_comm_peer_open.restype = ip
    lib.gsfm_comm_peer_open.argtypes = [vp, ip, ip, C.c_int64, C.c_char_p]
    lib.gsfm_comm_peer_connect.restype = ip
    lib.gsfm_comm_peer_connect.argtypes = [vp, C.c_char_p]
    lib.gsfm_comm_allreduce_bench.restype = ip
    lib.gsfm_comm_allreduce_bench.argtypes = [vp, C.c_int64, ip, dp]
    lib.gsfm_comm_peer_selftest.restype = ip
    lib.gsfm_comm_peer_selftest.argtypes = [vp, dp]
    lib.gsfm_ra_options_default.restype = None
    lib.gsfm_ra_options_default.argtypes = [C.POINTER(RaOptions)]
    lib.gsfm_ra_solve.restype = ip
    lib.gsfm_ra_solve.argtypes = [vp, C.POINTER(RaProblemC), C.POINTER(RaOptions), vp, C.POINTER(Report)]
    lib.gsfm_ra_residuals.restype = ip
    lib.gsfm_ra_residuals.argtypes = [vp, C.POINTER(RaProblemC), C.POINTER(RaOptions), vp, vp, vp]
    lib.gsfm_ra_residuals_timed.restype = ip
    lib.gsfm_ra_residuals_timed.argtypes = [vp, C.POINTER(RaProblemC), C.POINTER(RaOptions), vp, ip, dp]
    lib.gsfm_ra_laplacian_apply.restype = ip
    lib.gsfm_ra_laplacian_apply.argtypes = [vp, C.POINTER(RaProblemC), vp, vp, vp, ip, dp]
    if hasattr(lib, "gsfm_gp_solve"):
        lib.gsfm_gp_options_default.restype = None
        lib.gsfm_gp_options_default.argtypes = [C.POINTER(GpOptions)]
        lib.gsfm_gp_solve.restype = ip
        lib.gsfm_gp_solve.argtypes = [vp, C.POINTER(GpProblemC), C.POINTER(GpOptions), vp, vp, C.POINTER(Report)]
    if hasattr(lib, "gsfm_ba_solve"):
        lib.gsfm_ba_options_default.restype = None
        lib.gsfm_ba_options_default.argtypes = [C.POINTER(BaOptions)]
        lib.gsfm_ba_solve.restype = ip
        lib.gsfm_ba_solve.argtypes = [vp, C.POINTER(BaProblemC), C.POINTER(BaOptions), vp, vp, vp, vp, C.POINTER(Report)]
    i64p = C.POINTER(C.c_int64)
    lib.gsfm_filter_tracks_by_reprojection.restype = ip
    lib.gsfm_filter_tracks_by_reprojection.argtypes = [vp, C.POINTER(SceneViewC), C.c_double, ip, vp, i64p]
    lib.gsfm_undistort_features.restype = ip
    lib.gsfm_undistort_features.argtypes = [vp, ip, C.c_int64, vp, vp, ip, vp, vp, ip, vp]
    lib.gsfm_filter_tracks_by_angle.restype = ip
    lib.gsfm_filter_tracks_by_angle.argtypes = [vp, C.POINTER(SceneViewC), C.c_double, vp, i64p]
    lib.gsfm_filter_tracks_triangulation_angle.restype = ip
    lib.gsfm_filter_tracks_triangulation_angle.argtypes = [vp, C.POINTER(SceneViewC), C.c_double, vp, i64p]
    lib.gsfm_normalize_reconstruction.restype = ip
    lib.gsfm_normalize_reconstruction.argtypes = [vp, C.c_int32, C.c_int32, vp, vp, vp, C.c_int64, vp, C.c_int32, C.c_double,
                                                  C.c_double, C.c_double, dp]
    lib.gsfm_filter_rotations.restype = ip
    lib.gsfm_filter_rotations.argtypes = [vp, C.c_int32, C.c_int32, vp, C.c_int64, vp, vp, vp, C.c_double, vp, i64p]
    lib.gsfm_tracks_compact.restype = ip
    lib.gsfm_tracks_compact.argtypes = [vp, C.c_int32, C.c_int64, C.c_int64, vp, vp, vp, C.c_int32, C.POINTER(vp), C.POINTER(C.c_int32), i64p]
    lib.gsfm_ctx_set_dump_dir.restype = ip
    lib.gsfm_ctx_set_dump_dir.argtypes = [vp, C.c_char_p]
    lib.gsfm_track_options_default.restype = None
    lib.gsfm_track_options_default.argtypes = [C.POINTER(TrackOptionsC)]
    lib.gsfm_tracks_establish.restype = ip
    lib.gsfm_tracks_establish.argtypes = [vp, C.POINTER(MatchGraphC), C.POINTER(TrackOptionsC), i64p, i64p, i64p]
    lib.gsfm_tracks_select.restype = ip
    lib.gsfm_tracks_select.argtypes = [vp, C.POINTER(TrackSetC), C.c_int32, vp, C.c_int32, C.POINTER(TrackOptionsC), i64p, i64p]
    lib.gsfm_tracks_fetch.restype = ip
    lib.gsfm_tracks_fetch.argtypes = [vp, C.c_int32, C.POINTER(TrackSetC)]
    lib.gsfm_keep_largest_connected_component.restype = ip
    lib.gsfm_keep_largest_connected_component.argtypes = [vp, C.c_int32, C.c_int32, C.c_int64, vp, vp, vp, vp, vp, i64p]
    _lib = lib
    return lib


def ptr(a):
    """Raw address of a numpy array (host) or DeviceArray (HBM); None -> NULL."""
    if a is None:
        return None
    if isinstance(a, np.ndarray):
        assert a.flags["C_CONTIGUOUS"], "array must be C-contiguous"
        if a.size == 0:
            return None  # what std::vector<T>(0).data() hands the C ABI from the C++ adapter
        return a.ctypes.data
    return a.data_ptr()


class DeviceArray:
    """A dense array resident in the HBM of a Context's GPU, allocated by libgsfm's own HIP
    runtime (gsfm_device_alloc).  Deliberately not a torch tensor: PyTorch wheels bundle a second
    copy of the HIP runtime, and pointers must not cross runtimes."""

    def __init__(self, ctx: "Context", shape, dtype):
        self.ctx = ctx
        self.shape = tuple(int(s) for s in (shape if isinstance(shape, (tuple, list)) else (shape,)))
        self.dtype = np.dtype(dtype)
        self.nbytes = int(np.prod(self.shape, dtype=np.int64)) * self.dtype.itemsize
        p = C.c_void_p()
        rc = ctx.lib.gsfm_device_alloc(ctx.handle, self.nbytes, C.byref(p))
        if rc != 0:
            raise GsfmError(rc, "gsfm_device_alloc")
        self._ptr = p.value

    @classmethod
    def from_numpy(cls, ctx: "Context", a: np.ndarray) -> "DeviceArray":
        a = np.ascontiguousarray(a)
        d = cls(ctx, a.shape, a.dtype)
        rc = ctx.lib.gsfm_memcpy_h2d(ctx.handle, d._ptr, a.ctypes.data, d.nbytes)
        if rc != 0:
            raise GsfmError(rc, "gsfm_memcpy_h2d")
        return d

    def numpy(self) -> np.ndarray:
        out = np.empty(self.shape, dtype=self.dtype)
        if self.nbytes == 0:
            return out
        rc = self.ctx.lib.gsfm_memcpy_d2h(self.ctx.handle, out.ctypes.data, self._ptr, self.nbytes)
        if rc != 0:
            raise GsfmError(rc, "gsfm_memcpy_d2h")
        return out

    def copy_from(self, other: "DeviceArray"):
        assert other.nbytes == self.nbytes
        rc = self.ctx.lib.gsfm_memcpy_d2d(self.ctx.handle, self._ptr, other._ptr, self.nbytes)
        if rc != 0:
            raise GsfmError(rc, "gsfm_memcpy_d2d")

    def clone(self) -> "DeviceArray":
        d = DeviceArray(self.ctx, self.shape, self.dtype)
        d.copy_from(self)
        return d

    def copy(self) -> "DeviceArray":
        return self.clone()

    def data_ptr(self) -> int:
        return self._ptr

    def contiguous(self) -> "DeviceArray":
        return self

    def prefix(self, n: int) -> "DeviceArray":
        """The first n rows as a DeviceArray that shares this one's memory (no copy; this array must outlive the view)."""
        assert 0 <= n <= self.shape[0]
        v = DeviceArray.__new__(DeviceArray)
        v.ctx, v.dtype, v.shape = self.ctx, self.dtype, (int(n),) + self.shape[1:]
        v.nbytes = int(np.prod(v.shape, dtype=np.int64)) * v.dtype.itemsize
        v._ptr, v._base = self._ptr, self  # keeps the owner alive
        return v

    def free(self):
        if getattr(self, "_base", None) is not None:  # a prefix() view does not own its memory
            self._ptr = None
            return
        if getattr(self, "_ptr", None) and getattr(self.ctx, "handle", None):
            self.ctx.lib.gsfm_device_free(self.ctx.handle, self._ptr)
        self._ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class Context:
    """One GPU, one stream (gsfm_ctx).  Raises if no HIP device is present."""

    def __init__(self, device_id: int = -1):
        self.lib = load()
        h = C.c_void_p()
        rc = self.lib.gsfm_ctx_create(device_id, C.byref(h))
        if rc != 0:
            raise GsfmError(rc, "gsfm_ctx_create")
        self.handle = h
        self.rank = 0
        self.world = 1
        for item in filter(None, os.environ.get("GSFM_KNOBS", "").split(",")):  # experiments: "name=value,..."
            name, _, val = item.partition("=")
            self.set_knob(name.strip(), int(val or 1))

    def close(self):
        if getattr(self, "handle", None):
            self.lib.gsfm_ctx_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def stream(self) -> int:
        return int(self.lib.gsfm_ctx_stream(self.handle) or 0)

    def device_name(self) -> str:
        buf = C.create_string_buffer(256)
        self.lib.gsfm_ctx_device_name(self.handle, buf, 256)
        return buf.value.decode()

    def synchronize(self):
        rc = self.lib.gsfm_ctx_synchronize(self.handle)
        if rc != 0:
            raise GsfmError(rc, "gsfm_ctx_synchronize")

    def to_device(self, a: np.ndarray) -> "DeviceArray":
        return DeviceArray.from_numpy(self, a)

    def set_dump_dir(self, directory):
        """Every solve on this context writes its flat problem + result to `directory` (None disables); glomap_amd/flatio.py
        reads the files back."""
        rc = self.lib.gsfm_ctx_set_dump_dir(self.handle, None if directory is None else str(directory).encode())
        if rc != 0:
            raise GsfmError(rc, "gsfm_ctx_set_dump_dir")

    def profile_enable(self, on: bool):
        self.lib.gsfm_ctx_profile_enable(self.handle, int(on))

    def profile_read(self, kernel_id: int):
        n = C.c_int64(0)
        ms = C.c_double(0.0)
        rc = self.lib.gsfm_ctx_profile_read(self.handle, kernel_id, C.byref(n), C.byref(ms))
        if rc != 0:
            raise GsfmError(rc, "gsfm_ctx_profile_read")
        return n.value, ms.value

    KNOBS = ("ba_aw_by_application", "ba_aw_check", "ba_separate_blocks", "ba_no_nontemporal", "ra_no_blockdense",
             "ra_no_substructure", "ra_dense_refactor", "gp_coarse_cluster", "seg_len", "chunked_sweeps", "experiment",
             "gp_no_recycle", "gp_recycle_min_iters", "gp_recycle_cut_percent", "gp_dense")

    def set_knob(self, name: str, value: int = 1):
        """Diagnostic / A-B knobs (gsfm_ctx_set_knob); 0 restores the default.  The library reads no environment variable
        for these: experiments that want one set GSFM_KNOBS="name=value,..." and the Python mirror applies it here."""
        rc = self.lib.gsfm_ctx_set_knob(self.handle, self.KNOBS.index(name), int(value))
        if rc != 0:
            raise GsfmError(rc, "gsfm_ctx_set_knob")

    STAT_NAMES = ("pcg_solves", "pcg_deflated", "pcg_closed_form_aw", "pcg_single_workgroup", "pcg_joint_blocks",
                  "pcg_second_level", "allreduces", "pcg_iterations", "pcg_chunked_sweeps", "pcg_recycled", "ritz_harvested", "dense_solves")

    def stats(self, reset: bool = False) -> dict:
        """Which solver paths ran on this context (gsfm_ctx_stats)."""
        buf = (C.c_int64 * len(self.STAT_NAMES))()
        rc = self.lib.gsfm_ctx_stats(self.handle, buf, len(self.STAT_NAMES), int(reset))
        if rc != 0:
            raise GsfmError(rc, "gsfm_ctx_stats")
        return dict(zip(self.STAT_NAMES, (int(v) for v in buf)))

    LM_TRACE_COLS = ("cost", "radius", "model_change", "candidate_cost", "step_size", "accepted", "linear_iterations")

    def lm_trace(self):
        """[iterations, 7] array: the LM iterations of the last gp / ba solve on this context (gsfm_ctx_lm_trace; columns
        LM_TRACE_COLS)."""
        import numpy as np

        rows = self.lib.gsfm_ctx_lm_trace(self.handle, None, 0)
        out = np.zeros((max(rows, 0), len(self.LM_TRACE_COLS)))
        if rows > 0:
            self.lib.gsfm_ctx_lm_trace(self.handle, out.ctypes.data_as(C.POINTER(C.c_double)), rows)
        return out

    def comm_destroy(self):
        """Detach whatever transport is attached (collective in effect: every rank must be done with its solves)."""
        rc = self.lib.gsfm_comm_destroy(self.handle)
        if rc != 0:
            raise GsfmError(rc, "gsfm_comm_destroy")
        self.rank, self.world = 0, 1

    def last_error(self) -> str:
        return (self.lib.gsfm_ctx_last_error(self.handle) or b"").decode(errors="replace")

    def comm_init(self, unique_id: bytes, rank: int, world: int):
        rc = self.lib.gsfm_comm_init(self.handle, unique_id, rank, world)
        if rc != 0:
            raise GsfmError(rc, "gsfm_comm_init")
        self.rank, self.world = rank, world

    def comm_selftest(self) -> float:
        """One checked RCCL all-reduce on the context's stream; returns the world size it observed."""
        out = C.c_double(0.0)
        rc = self.lib.gsfm_comm_selftest(self.handle, C.byref(out))
        if rc != 0:
            raise GsfmError(rc, "gsfm_comm_selftest")
        return out.value


def _comm_init_host(self, allreduce, rank: int, world: int):
    """Validation transport: `allreduce(np_array, op)` must reduce the array in place across ranks
    (op 0 = sum, 1 = max), e.g. with torch.distributed on gloo.  See gsfm_comm_init_host."""

    def _cb(buf, n, op, _user):
        try:
            a = np.ctypeslib.as_array(buf, shape=(n,))
            allreduce(a, op)
            return 0
        except Exception as e:  # never let an exception cross the C boundary
            print(f"[gsfm] host all-reduce callback failed: {e!r}")
            return 1

    self._host_cb = HOST_ALLREDUCE_FN(_cb)  # keep alive
    rc = self.lib.gsfm_comm_init_host(self.handle, self._host_cb, None, rank, world)
    if rc != 0:
        raise GsfmError(rc, "gsfm_comm_init_host")
    self.rank, self.world = rank, world


Context.comm_init_host = _comm_init_host

GSFM_PEER_HANDLE_BYTES = 64


def _comm_init_peer(self, allgather, rank: int, world: int, capacity_doubles: int = 1 << 18):
    """Peer-mailbox transport (gsfm_comm_peer_*): `allgather(bytes) -> list of the bytes of every rank, in rank order`
    carries the 64-byte memory handles (e.g. torch.distributed.all_gather_object on gloo).  At most 8 ranks of one node."""
    buf = C.create_string_buffer(GSFM_PEER_HANDLE_BYTES)
    rc = self.lib.gsfm_comm_peer_open(self.handle, rank, world, capacity_doubles, buf)
    if rc != 0:
        raise GsfmError(rc, "gsfm_comm_peer_open: " + self.last_error())
    handles = allgather(buf.raw)
    if len(handles) != world or any(len(h) != GSFM_PEER_HANDLE_BYTES for h in handles):
        raise ValueError("peer transport: all-gather must return one 64-byte handle per rank")
    rc = self.lib.gsfm_comm_peer_connect(self.handle, b"".join(handles))
    if rc != 0:
        raise GsfmError(rc, "gsfm_comm_peer_connect: " + self.last_error())
    self.rank, self.world = rank, world


def _comm_peer_selftest(self) -> float:
    out = C.c_double(0.0)
    rc = self.lib.gsfm_comm_peer_selftest(self.handle, C.byref(out))
    if rc != 0:
        raise GsfmError(rc, "gsfm_comm_peer_selftest: " + self.last_error())
    return out.value


def _comm_allreduce_bench(self, n: int, repeats: int = 200) -> float:
    """Microseconds per all-reduce of n doubles through the attached transport (collective)."""
    out = C.c_double(0.0)
    rc = self.lib.gsfm_comm_allreduce_bench(self.handle, n, repeats, C.byref(out))
    if rc != 0:
        raise GsfmError(rc, "gsfm_comm_allreduce_bench: " + self.last_error())
    return out.value


Context.comm_allreduce_bench = _comm_allreduce_bench
Context.comm_init_peer = _comm_init_peer
Context.comm_peer_selftest = _comm_peer_selftest


def comm_unique_id() -> bytes:
    lib = load()
    buf = C.create_string_buffer(GSFM_COMM_ID_BYTES)
    rc = lib.gsfm_comm_unique_id(buf)
    if rc != 0:
        raise GsfmError(rc, "gsfm_comm_unique_id")
    return buf.raw
