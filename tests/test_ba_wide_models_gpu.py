"""GPU parity of the 16-wide bundle-adjustment unit (glomap_amd/csrc/ba_wide.hip): the camera models with more than eight
parameters — FULL_OPENCV, THIN_PRISM_FISHEYE, RAD_TAN_THIN_PRISM_FISHEYE — which the reference reaches through
colmap::CreateCameraCostFunction on any CameraModelId (glomap/estimators/bundle_adjustment.cc:136-139,149-152,167-170).

Three kinds of evidence:
  * the 16-wide unit against the 8-wide unit on the SAME problems (8-parameter models, intrinsics rows zero-padded to 16):
    both are ba_impl.hpp, so every intrinsics width that did not scale shows up here;
  * the three wide models against the numpy oracle (oracle/ba.py, Jacobians finite-difference-checked in
    tests/test_oracle_ba.py) — rotations 1e-4 rad, centres 1e-3 of the extent (north_star);
  * the boundary: a wide model with 8-wide rows is refused, the pixel-space reprojection filter reads 16-wide rows.
"""
import numpy as np
import pytest

from glomap_amd import estimators, so3, synthetic
from oracle import ba as oba
from oracle import filters as of

pytestmark = pytest.mark.gpu

WIDE = {
    # fx, fy, cx, cy, k1, k2, p1, p2, k3, k4, k5, k6
    "full_opencv": (6, [1200, 1190, 640, 480, 0.02, -0.01, 0.001, -0.002, 0.003, 0.01, -0.004, 0.002]),
    # fx, fy, cx, cy, k1, k2, p1, p2, k3, k4, sx1, sy1
    "thin_prism_fisheye": (10, [1200, 1190, 640, 480, 0.02, -0.01, 0.001, -0.002, 0.004, -0.002, 0.0015, -0.001]),
    # fx, fy, cx, cy, k0 .. k5, p0, p1, s0 .. s3
    "rad_tan_thin_prism_fisheye": (11, [1200, 1190, 640, 480, 0.02, -0.01, 0.004, -0.002, 0.001, -0.0005, 0.001, -0.002,
                                        0.0015, -0.0008, -0.001, 0.0005]),
}


def _oracle(p, **kw):
    return oba.solve(p.num_cams, p.pt_offset, p.obs_cam, p.obs_xy, p.cam_intr, p.intr_model, p.fixed_cam, p.cam_q, p.cam_t,
                     p.pt_xyz, p.intr_params, oba.BundleAdjusterOptions(**kw))


def _gpu_options(**kw):
    """Reduced solves to 1e-10: the high-order distortion coefficients are weakly determined on a 60-degree scene, and the
    comparison is with an oracle that solves the normal equations exactly."""
    so = estimators.SolverOptions(max_num_iterations=200, pcg_relative_tolerance=1e-10)
    return estimators.BundleAdjusterOptions(solver_options=so, **kw)


def _pose_diff(qa, ta, qb, tb):
    Ra, Rb = so3.quat_to_rotmat(qa), so3.quat_to_rotmat(qb)
    ca, cb = -np.einsum("nji,nj->ni", Ra, ta), -np.einsum("nji,nj->ni", Rb, tb)
    ang = np.radians(so3.rotation_angle_deg(Ra, Rb)).max()
    extent = np.linalg.norm(cb - cb.mean(0), axis=1).max()
    return ang, (np.linalg.norm(ca - cb, axis=1) / extent).max()


def _widen(p, width=16):
    w = p.copy()
    for name in ("intr_params", "gt_intr"):
        a = getattr(w, name)
        if a is not None:
            b = np.zeros((a.shape[0], width))
            b[:, : a.shape[1]] = a
            setattr(w, name, b)
    return w


def _observe(p, noise=0.3, seed=0):
    """Observations of the ground-truth scene through p.intr_model / p.intr_params (the oracle's projection) + noise."""
    lens = np.diff(p.pt_offset)
    obs_pt = np.repeat(np.arange(p.num_pts), lens)
    R = so3.quat_to_rotmat(p.gt_q)
    xc = np.einsum("mij,mj->mi", R[p.obs_cam], p.gt_xyz[obs_pt]) + p.gt_t[p.obs_cam]
    ik = p.cam_intr[p.obs_cam]
    uv, _, _, valid = oba.project(p.intr_model[ik], p.intr_params[ik], xc)
    assert valid.all()
    p.obs_xy = uv + np.random.default_rng(seed).normal(0, noise, uv.shape)


def _wide_problem(name, shared, seed=5, ncam=15, npts=300):
    p = _widen(synthetic.make_ba_problem(num_cams=ncam, num_pts=npts, seed=seed, pixel_noise=0.0, outlier_ratio=0.0,
                                         shared_intrinsics=shared))
    mid, vals = WIDE[name]
    p.intr_model[:] = mid
    p.intr_params[:] = 0
    p.intr_params[:, : len(vals)] = vals
    _observe(p)
    return p


@pytest.mark.parametrize("shared", [True, False])
@pytest.mark.parametrize("model,params", [(2, None), (4, [1200, 1190, 640, 480, 0.02, -0.01, 0.001, -0.002]),
                                          (5, [1200, 1190, 640, 480, 0.02, -0.01, 0.004, -0.002])])
def test_wide_unit_equals_narrow_unit_on_8_parameter_models(gsfm_ctx, model, params, shared):
    """Same problem, [K,8] rows (ba.hip) and zero-padded [K,16] rows (ba_wide.hip): the 8-wide unit with separate pose /
    intrinsics blocks (the only layout the 16-wide unit has) and 30 cameras (no deflation) runs the same algorithm."""
    p = synthetic.make_ba_problem(num_cams=30, num_pts=800, seed=3, pixel_noise=0.5, outlier_ratio=0.01, shared_intrinsics=shared,
                                  intr_noise=0.01)
    if params is not None:
        p.intr_model[:] = model
        p.intr_params[:] = 0
        p.intr_params[:, : len(params)] = params
        _observe(p, noise=0.5)
    gsfm_ctx.set_knob("ba_separate_blocks", 1)
    try:
        rc8, q8, t8, X8, i8, rep8 = estimators.ba_solve(p, ctx=gsfm_ctx)
    finally:
        gsfm_ctx.set_knob("ba_separate_blocks", 0)
    rc16, q16, t16, X16, i16, rep16 = estimators.ba_solve(_widen(p), ctx=gsfm_ctx)
    assert rc8 == 0 and rc16 == 0
    ang, pos = _pose_diff(q16, t16, q8, t8)
    print(f"[parity] wide vs narrow unit, model {model}, shared={shared}: iterations {rep16['iterations']} / {rep8['iterations']}, "
          f"initial cost {rep16['initial_cost']:.12e} / {rep8['initial_cost']:.12e}, final {rep16['final_cost']:.12e} / "
          f"{rep8['final_cost']:.12e}, rotations {ang:.2e} rad, centres {pos:.2e}, intrinsics {np.abs(i16[:, :8] - i8).max():.2e}")
    assert i16.shape == (p.num_intr, 16) and np.array_equal(i16[:, 8:], np.zeros((p.num_intr, 8)))
    assert abs(rep16["initial_cost"] - rep8["initial_cost"]) <= 1e-12 * rep8["initial_cost"]
    assert rep16["iterations"] == rep8["iterations"]
    # (the two units stop their reduced solves at 1e-6 on sums that may differ in the last bit: not bit-identical by design)
    assert abs(rep16["final_cost"] - rep8["final_cost"]) <= 1e-7 * rep8["final_cost"]
    assert ang < 1e-6 and pos < 1e-6
    assert np.abs(i16[:, :8] - i8).max() < 1e-3
    assert np.abs(X16 - X8).max() < 1e-5 * np.abs(X8).max()


@pytest.mark.parametrize("shared", [True, False])
@pytest.mark.parametrize("name", list(WIDE))
def test_ba_wide_camera_models_match_oracle(gsfm_ctx, name, shared):
    p = _wide_problem(name, shared)
    ok, q_o, t_o, X_o, intr_o, summ = _oracle(p)
    rc, q_g, t_g, X_g, intr_g, rep = estimators.ba_solve(p, _gpu_options(), ctx=gsfm_ctx)
    assert ok and rc == 0
    ang, pos = _pose_diff(q_g, t_g, q_o, t_o)
    npar = oba.NUM_PARAMS[WIDE[name][0]]
    print(f"[parity] BA {name} shared={shared}: LM {rep['iterations']} / {summ.iterations}, initial cost {rep['initial_cost']:.9e} / "
          f"{summ.initial_cost:.9e}, final {rep['final_cost']:.9e} / {summ.final_cost:.9e}, rotations {ang:.2e} rad, centres {pos:.2e}, "
          f"intrinsics {np.abs(intr_g - intr_o).max():.2e}")
    assert abs(rep["initial_cost"] - summ.initial_cost) <= 1e-9 * summ.initial_cost
    assert abs(rep["final_cost"] - summ.final_cost) <= 1e-4 * summ.final_cost
    assert ang < 1e-4 and pos < 1e-3
    # principal point frozen (SubsetManifold, ba.cc:273-287), the unused tail of the rows untouched
    assert np.array_equal(intr_g[:, 2:4], p.intr_params[:, 2:4])
    assert np.array_equal(intr_g[:, npar:], p.intr_params[:, npar:])
    assert not np.array_equal(intr_g[:, 4:npar], p.intr_params[:, 4:npar])  # (the distortion parameters did move)


@pytest.mark.parametrize("kw", [dict(optimize_intrinsics=False), dict(optimize_principal_point=True),
                                dict(optimize_rotations=False), dict(optimize_points=False)])
def test_ba_wide_option_flags_match_oracle(gsfm_ctx, kw):
    """The option branches of ba.cc:244-293 on a 16-parameter model."""
    p = _wide_problem("rad_tan_thin_prism_fisheye", shared=True, seed=7)
    ok, q_o, t_o, X_o, intr_o, summ = _oracle(p, **kw)
    rc, q_g, t_g, X_g, intr_g, rep = estimators.ba_solve(p, _gpu_options(**kw), ctx=gsfm_ctx)
    assert ok and rc == 0
    ang, pos = _pose_diff(q_g, t_g, q_o, t_o)
    print(f"[parity] BA rad_tan_thin_prism_fisheye {kw}: LM {rep['iterations']} / {summ.iterations}, final {rep['final_cost']:.9e} / "
          f"{summ.final_cost:.9e}, rotations {ang:.2e} rad, centres {pos:.2e}")
    assert abs(rep["final_cost"] - summ.final_cost) <= 1e-4 * summ.final_cost
    assert ang < 1e-4 and pos < 1e-3
    if kw.get("optimize_intrinsics") is False:
        assert np.array_equal(intr_g, p.intr_params)
    if kw.get("optimize_principal_point"):
        assert not np.array_equal(intr_g[:, 2:4], p.intr_params[:, 2:4])
    if kw.get("optimize_rotations") is False:
        assert np.array_equal(q_g, p.cam_q)
    if kw.get("optimize_points") is False:
        assert np.array_equal(X_g, p.pt_xyz)


def test_ba_wide_model_with_narrow_rows_is_refused(gsfm_ctx):
    p = synthetic.make_ba_problem(num_cams=8, num_pts=100, seed=1)
    p.intr_model[:] = 6  # FULL_OPENCV in [K,8] rows
    rc = estimators.ba_solve(p, ctx=gsfm_ctx)[0]
    assert rc == -7, rc  # GSFM_ERR_UNSUPPORTED
    assert "intr_stride" in gsfm_ctx.last_error()


def test_pixel_reprojection_filter_reads_wide_rows(gsfm_ctx):
    from glomap_amd import processors as pr

    p = _wide_problem("thin_prism_fisheye", shared=False, seed=6, ncam=25, npts=1500)
    rng = np.random.default_rng(2)
    bad = rng.random(p.obs_xy.shape[0]) < 0.05
    p.obs_xy[bad] += rng.normal(0, 30, (int(bad.sum()), 2))
    view = pr.SceneView(p.num_cams, p.pt_offset, p.obs_cam, p.gt_q, p.gt_t, p.gt_xyz, obs_xy=p.obs_xy, cam_intr=p.cam_intr,
                        intr_model=p.intr_model, intr_params=p.intr_params)
    k_o, c_o = of.filter_tracks_by_reprojection(p.pt_offset, p.obs_cam, p.gt_q, p.gt_t, p.gt_xyz, 4.0, False, obs_xy=p.obs_xy,
                                                cam_intr=p.cam_intr, intr_model=p.intr_model, intr_params=p.intr_params)
    k_g, c_g = pr.TrackFilter.FilterTracksByReprojection(view, 4.0, False, ctx=gsfm_ctx)
    assert np.array_equal(k_g.astype(bool), k_o) and c_g == c_o
    assert 0.02 < 1 - k_o.mean() < 0.1


@pytest.mark.parametrize("sensors", [False, True])
def test_wide_unit_equals_narrow_unit_on_rigs(gsfm_ctx, sensors):
    """The rig branches of ba_impl.hpp (constant cam_from_rig, bundle_adjustment.cc:147-160; optimised cam_from_rig blocks,
    :161-179) in the 16-wide unit: same rig problem with [K,8] and zero-padded [K,16] intrinsics rows."""
    _, ba, info = synthetic.make_rig_problems(20, 3, 1500, seed=4, pixel_noise=0.5)
    opt = estimators.BundleAdjusterOptions(optimize_rig_poses=sensors)
    if sensors:
        rng = np.random.default_rng(9)
        s0 = info["sensor_cam_from_rig"].copy()
        dq = so3.rotmat_to_quat(so3.aa_to_rotmat(rng.normal(0, np.radians(0.5), (s0.shape[0], 3))))
        s0[:, :4] = oba.quat_mul(dq, s0[:, :4])
        s0[:, 4:] += rng.normal(0, 0.03, (s0.shape[0], 3))
        ba = ba.copy()
        ba.image_sensor = info["sensor_block"].copy()
        ba.sensor_cam_from_rig = s0
    rc8, q8, t8, X8, i8, rep8 = estimators.ba_solve(ba, opt, ctx=gsfm_ctx)
    rc16, q16, t16, X16, i16, rep16 = estimators.ba_solve(_widen(ba), opt, ctx=gsfm_ctx)
    assert rc8 == 0 and rc16 == 0
    ang, pos = _pose_diff(q16, t16, q8, t8)
    print(f"[parity] wide vs narrow unit, rigs (sensor blocks: {sensors}): iterations {rep16['iterations']} / {rep8['iterations']}, "
          f"initial cost {rep16['initial_cost']:.12e} / {rep8['initial_cost']:.12e}, final {rep16['final_cost']:.12e} / "
          f"{rep8['final_cost']:.12e}, rotations {ang:.2e} rad, centres {pos:.2e}")
    assert abs(rep16["initial_cost"] - rep8["initial_cost"]) <= 1e-12 * rep8["initial_cost"]
    assert rep16["iterations"] == rep8["iterations"]
    assert abs(rep16["final_cost"] - rep8["final_cost"]) <= 1e-7 * rep8["final_cost"]
    assert ang < 1e-6 and pos < 1e-6 and np.abs(i16[:, :8] - i8).max() < 1e-3
    if sensors:
        assert np.abs(rep16["sensor_cam_from_rig"] - rep8["sensor_cam_from_rig"]).max() < 1e-6
