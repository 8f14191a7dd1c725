"""Oracle (CPU restatement) of global bundle adjustment: analytic reprojection Jacobians checked by
finite differences for every camera model, and ground-truth recovery with the reference's
tolerances (glomap/controllers/global_mapper_test.cc:82-86: 1e-2 deg, 1e-4 centre, noise-free)."""
import numpy as np
import pytest

from glomap_amd import so3, synthetic
from oracle import ba


def _set_model(p, name):
    if name == "opencv":
        p.intr_model[:] = ba.OPENCV
        p.intr_params[:, :8] = [1200, 1190, 640, 480, 0.02, -0.01, 0.001, -0.002]
    elif name == "pinhole":
        p.intr_model[:] = ba.PINHOLE
        p.intr_params[:] = 0
        p.intr_params[:, :4] = [1200, 1190, 640, 480]
    elif name == "radial":
        p.intr_model[:] = ba.RADIAL
        p.intr_params[:] = 0
        p.intr_params[:, :5] = [1200, 640, 480, 0.02, -0.01]
    elif name == "simple_pinhole":
        p.intr_model[:] = ba.SIMPLE_PINHOLE
        p.intr_params[:] = 0
        p.intr_params[:, :3] = [1200, 640, 480]
    elif name == "opencv_fisheye":
        p.intr_model[:] = ba.OPENCV_FISHEYE
        p.intr_params[:, :8] = [1200, 1190, 640, 480, 0.02, -0.01, 0.004, -0.002]
    elif name == "fov":
        p.intr_model[:] = ba.FOV
        p.intr_params[:] = 0
        p.intr_params[:, :5] = [1200, 1190, 640, 480, 0.6]
    elif name == "simple_radial_fisheye":
        p.intr_model[:] = ba.SIMPLE_RADIAL_FISHEYE
        p.intr_params[:] = 0
        p.intr_params[:, :4] = [1200, 640, 480, 0.02]
    elif name == "radial_fisheye":
        p.intr_model[:] = ba.RADIAL_FISHEYE
        p.intr_params[:] = 0
        p.intr_params[:, :5] = [1200, 640, 480, 0.02, -0.01]


ALL_MODELS = ["simple_radial", "opencv", "pinhole", "radial", "simple_pinhole", "opencv_fisheye", "fov", "simple_radial_fisheye",
              "radial_fisheye"]


def _problem(p, opt, fixed=-1):
    lens = np.diff(p.pt_offset)
    used = lens >= opt.min_num_view_per_track
    sel = np.repeat(used, lens)
    pt = (np.cumsum(used) - 1)[np.repeat(np.arange(p.num_pts), lens)][sel]
    prob = ba._BaProblem(p.num_cams, p.obs_cam.astype(np.int64)[sel], pt, p.obs_xy[sel], p.cam_intr.astype(np.int64),
                         p.intr_model.astype(np.int64), fixed, int(used.sum()), opt, width=p.intr_params.shape[1])
    return prob, prob.pack(p.cam_q, p.cam_t, p.pt_xyz[used], p.intr_params)


@pytest.mark.parametrize("mid,par,pts", [
    # every model on generic rays, plus the special branches: FOV's two series (omega^2 < 1e-4, r^2 < 1e-4) and the fisheye
    # models on the optical axis (r <= eps)
    (ba.SIMPLE_PINHOLE, [1200, 640, 480], "generic"), (ba.PINHOLE, [1200, 1190, 640, 480], "generic"),
    (ba.SIMPLE_RADIAL, [1200, 640, 480, 0.02], "generic"), (ba.RADIAL, [1200, 640, 480, 0.02, -0.01], "generic"),
    (ba.OPENCV, [1200, 1190, 640, 480, 0.02, -0.01, 0.001, -0.002], "generic"),
    (ba.OPENCV_FISHEYE, [1200, 1190, 640, 480, 0.02, -0.01, 0.004, -0.002], "generic"),
    (ba.SIMPLE_RADIAL_FISHEYE, [1200, 640, 480, 0.02], "generic"), (ba.RADIAL_FISHEYE, [1200, 640, 480, 0.02, -0.01], "generic"),
    (ba.FOV, [1200, 1190, 640, 480, 0.6], "generic"), (ba.FOV, [1200, 1190, 640, 480, 5e-3], "generic"),
    (ba.FOV, [1200, 1190, 640, 480, 0.6], "near_axis"), (ba.OPENCV_FISHEYE, [1200, 1190, 640, 480, 0.02, -0.01, 0.004, -0.002], "near_axis"),
    # the models with more than 8 parameters ([K, 16] intrinsics rows)
    (ba.FULL_OPENCV, [1200, 1190, 640, 480, 0.02, -0.01, 0.001, -0.002, 0.003, 0.01, -0.004, 0.002], "generic"),
    (ba.THIN_PRISM_FISHEYE, [1200, 1190, 640, 480, 0.02, -0.01, 0.001, -0.002, 0.004, -0.002, 0.0015, -0.001], "generic"),
    (ba.THIN_PRISM_FISHEYE, [1200, 1190, 640, 480, 0.02, -0.01, 0.001, -0.002, 0.004, -0.002, 0.0015, -0.001], "near_axis"),
    (ba.RAD_TAN_THIN_PRISM_FISHEYE, [1200, 1190, 640, 480, 0.02, -0.01, 0.004, -0.002, 0.001, -0.0005, 0.001, -0.002, 0.0015, -0.0008,
                                     -0.001, 0.0005], "generic"),
    (ba.RAD_TAN_THIN_PRISM_FISHEYE, [1200, 1190, 640, 480, 0.02, -0.01, 0.004, -0.002, 0.001, -0.0005, 0.001, -0.002, 0.0015, -0.0008,
                                     -0.001, 0.0005], "near_axis"),
])
def test_projection_jacobians_match_finite_differences(mid, par, pts):
    """oracle.ba.project (ImgFromCam + analytic derivatives) against central differences, ray by ray."""
    rng = np.random.default_rng(3)
    m = 64
    if pts == "generic":
        xc = np.column_stack([rng.uniform(-0.8, 0.8, m), rng.uniform(-0.6, 0.6, m), rng.uniform(1.0, 3.0, m)])
    else:
        xc = np.column_stack([rng.uniform(-4e-3, 4e-3, m), rng.uniform(-4e-3, 4e-3, m), rng.uniform(1.0, 3.0, m)])
    W = ba.MAXP_WIDE if len(par) > ba.MAXP else ba.MAXP
    P = np.zeros((m, W))
    P[:, : len(par)] = par
    model = np.full(m, mid)
    uv, Jx, Jp, valid = ba.project(model, P, xc)
    assert valid.all() and Jp.shape == (m, 2, W)
    h = 1e-6
    for j in range(3):
        d = np.zeros(3)
        d[j] = h
        num = (ba.project(model, P, xc + d)[0] - ba.project(model, P, xc - d)[0]) / (2 * h)
        assert np.abs(num - Jx[:, :, j]).max() < 1e-6 * max(1.0, np.abs(Jx).max())
    for j in range(len(par)):
        hj = h * max(1.0, abs(par[j]))
        d = np.zeros(W)
        d[j] = hj
        num = (ba.project(model, P + d, xc)[0] - ba.project(model, P - d, xc)[0]) / (2 * hj)
        assert np.abs(num - Jp[:, :, j]).max() < 1e-6 * max(1.0, np.abs(Jp[:, :, j]).max())
    assert np.all(Jp[:, :, len(par):] == 0)


def test_fisheye_on_the_optical_axis():
    """r <= eps: the equidistant mapping is the identity there and the derivative stays finite."""
    xc = np.array([[0.0, 0.0, 2.0]])
    for mid, par in ((ba.OPENCV_FISHEYE, [1200, 1190, 640, 480, 0.02, -0.01, 0.004, -0.002]), (ba.RADIAL_FISHEYE, [1200, 640, 480, 0.02, -0.01])):
        P = np.zeros((1, ba.MAXP))
        P[0, : len(par)] = par
        uv, Jx, Jp, valid = ba.project(np.array([mid]), P, xc)
        assert valid[0] and np.allclose(uv[0], [640, 480]) and np.isfinite(Jx).all()
        assert np.isclose(Jx[0, 0, 0], 1200 / 2.0)


@pytest.mark.parametrize("model", ALL_MODELS)
def test_analytic_jacobian_matches_finite_differences(model):
    p = synthetic.make_ba_problem(num_cams=8, num_pts=60, seed=1, pixel_noise=0.0, outlier_ratio=0.0,
                                  shared_intrinsics=(model != "simple_radial"))
    _set_model(p, model)
    opt = ba.BundleAdjusterOptions(thres_loss_function=1e9, optimize_principal_point=True)
    prob, x0 = _problem(p, opt)
    _, _, J = prob.evaluate(x0)
    d = np.random.default_rng(0).normal(size=prob.n) * 1e-6
    num = (prob.evaluate(prob.plus(x0, d))[1] - prob.evaluate(prob.plus(x0, -d))[1]) / 2
    ana = J @ d
    assert np.abs(num - ana).max() < 1e-8 * np.abs(ana).max()


def test_parameterisation_masks():
    p = synthetic.make_ba_problem(num_cams=6, num_pts=40, seed=2)
    # default: principal point frozen (ba.cc:273-285), first frame constant (ba.cc:261-266)
    prob, x0 = _problem(p, ba.BundleAdjusterOptions(), fixed=0)
    _, _, J = prob.evaluate(x0)
    colsq = np.asarray(J.multiply(J).sum(0)).ravel()
    assert np.all(colsq[:6] == 0) and np.all(colsq[6:12] > 0)
    assert prob.fmask[0].tolist() == [True, False, False, True, False, False, False, False]
    # optimize_rotations = false (stage 1 of each BA round, global_mapper.cc:208)
    prob, x0 = _problem(p, ba.BundleAdjusterOptions(optimize_rotations=False), fixed=0)
    colsq = np.asarray(prob.evaluate(x0)[2].multiply(prob.evaluate(x0)[2]).sum(0)).ravel()
    assert np.all(colsq[: 6 * p.num_cams].reshape(-1, 6)[:, :3] == 0)
    # intrinsics constant
    assert not ba.free_param_mask(p.intr_model, ba.BundleAdjusterOptions(optimize_intrinsics=False)).any()


def _solve(p, **kw):
    opt = ba.BundleAdjusterOptions(**kw)
    return ba.solve(p.num_cams, p.pt_offset, p.obs_cam, p.obs_xy, p.cam_intr, p.intr_model, p.fixed_cam, p.cam_q,
                    p.cam_t, p.pt_xyz, p.intr_params, opt)


def _centers(q, t):
    R = so3.quat_to_rotmat(q)
    return -np.einsum("nji,nj->ni", R, t), R


@pytest.mark.parametrize("shared", [False, True])
def test_without_noise_recovers_ground_truth(shared):
    p = synthetic.make_ba_problem(num_cams=20, num_pts=400, seed=0, pixel_noise=0.0, outlier_ratio=0.0,
                                  shared_intrinsics=shared, intr_noise=0.01)
    ok, q, t, X, intr, s = _solve(p)
    assert ok and s.final_cost < 1e-8
    c, R = _centers(q, t)
    cg, Rg = _centers(p.gt_q, p.gt_t)
    assert synthetic.center_errors_after_sim3(c, cg).max() < 1e-4
    assert synthetic.rotation_errors_deg(R, Rg).max() < 1e-2
    assert np.abs(intr[:, 0] - 1200).max() < 1e-3


def test_with_noise_and_outliers_two_stage():
    # the controller runs positions-only then full BA (global_mapper.cc:201-223)
    p = synthetic.make_ba_problem(num_cams=25, num_pts=600, seed=3)
    ok, q, t, X, intr, s1 = _solve(p, optimize_rotations=False)
    assert ok and np.allclose(q, p.cam_q)
    p.cam_t, p.pt_xyz, p.intr_params = t, X, intr
    ok, q, t, X, intr, s2 = _solve(p)
    assert ok and s2.final_cost <= s1.final_cost
    c, R = _centers(q, t)
    cg, Rg = _centers(p.gt_q, p.gt_t)
    assert synthetic.center_errors_after_sim3(c, cg).max() < 0.1 and synthetic.rotation_errors_deg(R, Rg).max() < 0.1


def test_wide_models_reduce_to_their_narrow_relatives():
    """Definitions cross-checked against the 8-parameter models they extend: FULL_OPENCV with k3 .. k6 = 0 is OPENCV;
    THIN_PRISM_FISHEYE without tangential / thin-prism terms and RAD_TAN_THIN_PRISM_FISHEYE with k4 = k5 = p = s = 0 are the
    equidistant fisheye with a polynomial in theta^2, i.e. OPENCV_FISHEYE with the same k1 .. k4 (theta_d = theta (1 + k1 theta^2
    + ...), models.h)."""
    rng = np.random.default_rng(5)
    m = 50
    xc = np.column_stack([rng.uniform(-0.8, 0.8, m), rng.uniform(-0.6, 0.6, m), rng.uniform(1.0, 3.0, m)])
    base = [1200, 1190, 640, 480]
    P8 = np.zeros((m, 8))
    P8[:] = base + [0.02, -0.01, 0.001, -0.002]
    P16 = np.zeros((m, 16))
    P16[:, :8] = P8
    a = ba.project(np.full(m, ba.OPENCV), P8, xc)
    b = ba.project(np.full(m, ba.FULL_OPENCV), P16, xc)
    assert np.allclose(a[0], b[0], rtol=0, atol=1e-9) and np.allclose(a[1], b[1], rtol=1e-12, atol=1e-9)
    assert np.allclose(a[2], b[2][:, :, :8], rtol=1e-12, atol=1e-9)
    P8[:] = base + [0.02, -0.01, 0.004, -0.002]           # OPENCV_FISHEYE k1 .. k4
    a = ba.project(np.full(m, ba.OPENCV_FISHEYE), P8, xc)
    P16[:] = 0
    P16[:, :4] = base
    P16[:, [4, 5, 8, 9]] = [0.02, -0.01, 0.004, -0.002]   # THIN_PRISM_FISHEYE k1, k2, (p1, p2,) k3, k4
    b = ba.project(np.full(m, ba.THIN_PRISM_FISHEYE), P16, xc)
    assert np.allclose(a[0], b[0], rtol=0, atol=1e-9) and np.allclose(a[1], b[1], rtol=1e-12, atol=1e-9)
    assert np.allclose(a[2][:, :, 4:8], b[2][:, :, [4, 5, 8, 9]], rtol=1e-12, atol=1e-9)
    P16[:] = 0
    P16[:, :4] = base
    P16[:, 4:8] = [0.02, -0.01, 0.004, -0.002]            # RAD_TAN_THIN_PRISM_FISHEYE k0 .. k3
    b = ba.project(np.full(m, ba.RAD_TAN_THIN_PRISM_FISHEYE), P16, xc)
    assert np.allclose(a[0], b[0], rtol=0, atol=1e-9) and np.allclose(a[1], b[1], rtol=1e-12, atol=1e-9)
    assert np.allclose(a[2][:, :, 4:8], b[2][:, :, 4:8], rtol=1e-12, atol=1e-9)


@pytest.mark.parametrize("mid,par", [
    (ba.FULL_OPENCV, [1200, 1190, 640, 480, 0.02, -0.01, 0.001, -0.002, 0.003, 0.01, -0.004, 0.002]),
    (ba.RAD_TAN_THIN_PRISM_FISHEYE, [1200, 1190, 640, 480, 0.02, -0.01, 0.004, -0.002, 0.001, -0.0005, 0.001, -0.002, 0.0015, -0.0008,
                                     -0.001, 0.0005])])
def test_wide_model_problem_jacobian_and_masks(mid, par):
    """[K, 16] intrinsics blocks through the whole problem: free-parameter mask (12 / 16 parameters, principal point frozen by
    the SubsetManifold, ba.cc:273-287), column layout, J d against central differences."""
    p = synthetic.make_ba_problem(num_cams=8, num_pts=60, seed=1, pixel_noise=0.0, outlier_ratio=0.0, shared_intrinsics=False)
    p.intr_model[:] = mid
    wide = np.zeros((p.num_intr, 16))
    wide[:, : len(par)] = par
    p.intr_params = wide
    prob, x0 = _problem(p, ba.BundleAdjusterOptions(thres_loss_function=1e9))
    assert prob.fmask.shape == (p.num_intr, 16)
    assert np.array_equal(prob.fmask.sum(1), np.full(p.num_intr, len(par) - 2)) and not prob.fmask[:, 2:4].any()
    assert not prob.fmask[:, len(par):].any()
    _, _, J = prob.evaluate(x0)
    d = np.random.default_rng(0).normal(size=prob.n) * 1e-6
    num = (prob.evaluate(prob.plus(x0, d))[1] - prob.evaluate(prob.plus(x0, -d))[1]) / 2
    ana = J @ d
    assert np.abs(num - ana).max() < 1e-8 * np.abs(ana).max()
