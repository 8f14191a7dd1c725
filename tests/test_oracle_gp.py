"""Oracle (CPU restatement) of global positioning: Ceres-style LM over BATA residuals, checked
against ground-truth recovery as the reference's own tests do after Sim(3) alignment
(glomap/controllers/global_mapper_test.cc:26-38, 82-86, 211-215)."""
import numpy as np

from glomap_amd import synthetic
from oracle import gp, lm


def test_mt19937_uniform_matches_libstdcxx():
    # std::mt19937 g(1); std::uniform_real_distribution<double> d(-1,1); d(g) x3 (checked with g++ 11)
    u = gp.mt19937_uniform(1, 3, -1.0, 1.0)
    assert np.allclose(u, [0.99436961646053112, 0.86511472273633094, -0.74375110445538795], rtol=0, atol=1e-16)


def test_huber_matches_ceres_definition():
    h = lm.HuberLoss(0.1)
    rho0, rho1 = h.evaluate(np.array([0.0, 0.005, 0.01, 0.04]))
    assert np.allclose(rho0, [0.0, 0.005, 0.01, 2 * 0.1 * 0.2 - 0.01])
    assert np.allclose(rho1, [1.0, 1.0, 1.0, 0.1 / 0.2])
    hs = lm.HuberLoss(0.1, 0.5)  # ScaledLoss(Huber, 0.5), gp.cc:242-247
    assert np.allclose(hs.evaluate(np.array([0.04]))[0], 0.5 * (2 * 0.1 * 0.2 - 0.01))


def test_schur_solve_equals_direct():
    import scipy.sparse as sp

    rng = np.random.default_rng(0)
    n_cam, n_pt, n_s = 4, 7, 20
    n = 3 * n_cam + 3 * n_pt + n_s
    J = sp.random(3 * n_s, n, density=0.0, format="lil", random_state=1)
    for k in range(n_s):
        c, p = rng.integers(n_cam), rng.integers(n_pt)
        for j in range(3):
            J[3 * k + j, 3 * c + j] = rng.normal()
            J[3 * k + j, 3 * n_cam + 3 * p + j] = rng.normal()
            J[3 * k + j, 3 * n_cam + 3 * n_pt + k] = rng.normal()
    J = J.tocsr()
    A = (J.T @ J + sp.identity(n) * 0.3).tocsr()
    b = rng.normal(size=n)
    x = lm.schur_solve(A, b, [(3 * n_cam + 3 * n_pt, n_s, 1), (3 * n_cam, n_pt, 3)])
    assert np.allclose(A @ x, b, atol=1e-10)


def _solve(p, **kw):
    opt = gp.GlobalPositionerOptions(**kw)
    return gp.solve(p.num_cams, p.pt_offset, p.obs_cam, p.obs_dir, p.obs_calibrated, p.cam_center, p.pt_xyz, opt)


def test_without_noise_recovers_ground_truth():
    # global_mapper_test.cc:82-86 pins 1e-4 on the centres for noise-free data
    p = synthetic.make_gp_problem(num_cams=30, num_pts=400, seed=0, dir_noise=0.0, outlier_ratio=0.0)
    ok, c, X, s = _solve(p)
    assert ok and s.final_cost < 1e-12
    assert synthetic.center_errors_after_sim3(c, p.gt_center).max() < 1e-4


def test_with_noise_and_outliers():
    # global_mapper_test.cc:211-215 pins 0.1 on the centres for the noisy scene
    p = synthetic.make_gp_problem(num_cams=40, num_pts=800, seed=1, uncalibrated_ratio=0.2)
    ok, c, X, s = _solve(p)
    assert ok
    assert s.termination.startswith("CONVERGENCE")
    assert synthetic.center_errors_after_sim3(c, p.gt_center).max() < 0.1


def test_short_tracks_are_left_untouched():
    p = synthetic.make_gp_problem(num_cams=20, num_pts=100, seed=2)
    # cut track 0 down to 2 views: skipped (gp.cc:258), xyz untouched
    lens = np.diff(p.pt_offset)
    drop = int(lens[0] - 2)
    keep = np.ones(p.num_obs, dtype=bool)
    keep[2 : 2 + drop] = False
    p.obs_cam, p.obs_dir, p.obs_calibrated = p.obs_cam[keep], p.obs_dir[keep], p.obs_calibrated[keep]
    p.pt_offset = np.concatenate([[0], np.cumsum(np.concatenate([[2], lens[1:]]))])
    p.pt_xyz[0] = [7.0, 8.0, 9.0]
    ok, c, X, s = _solve(p)
    assert ok and np.all(X[0] == [7.0, 8.0, 9.0])


def _pairs(p, rng, num_succ=4, noise=0.0):
    """Camera-to-camera directions as GlobalPositioner::AddCameraToCameraConstraints sees them (gp.cc:195-197):
    -R_cw2^T t_21 = c_2 - c_1 in the scale of the two-view geometry (unit translation)."""
    N = p.num_cams
    i = np.repeat(np.arange(N), num_succ)
    j = (i + np.tile(np.arange(1, num_succ + 1), N)) % N
    d = p.gt_center[j] - p.gt_center[i]
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    d += noise * rng.normal(size=d.shape)
    return i, j, d / np.linalg.norm(d, axis=1, keepdims=True)


def test_constraint_types_recover_ground_truth():
    # the estimator's other constraint types (gp.cc:42-64, 167-210, 223-253); the mapper only runs ONLY_POINTS
    p = synthetic.make_gp_problem(num_cams=24, num_pts=300, seed=3, dir_noise=0.0, outlier_ratio=0.0)
    pi, pj, pd = _pairs(p, np.random.default_rng(0))
    for ctype in (gp.ONLY_CAMERAS, gp.POINTS_AND_CAMERAS_BALANCED, gp.POINTS_AND_CAMERAS):
        opt = gp.GlobalPositionerOptions(constraint_type=ctype)
        ok, c, X, s = gp.solve(p.num_cams, p.pt_offset, p.obs_cam, p.obs_dir, p.obs_calibrated, p.cam_center, p.pt_xyz,
                               opt, pair_i=pi, pair_j=pj, pair_dir=pd)
        assert ok and s.final_cost < 1e-10, (ctype, s.final_cost)
        assert synthetic.center_errors_after_sim3(c, p.gt_center).max() < 1e-4, ctype
        if ctype == gp.ONLY_CAMERAS:  # points are not part of the problem: left as they came in
            assert np.array_equal(X, p.pt_xyz)


def test_only_cameras_needs_pairs_and_points_need_tracks():
    p = synthetic.make_gp_problem(num_cams=10, num_pts=60, seed=4)
    opt = gp.GlobalPositionerOptions(constraint_type=gp.ONLY_CAMERAS)
    e = np.zeros(0, dtype=np.int64)
    ok, *_ = gp.solve(p.num_cams, p.pt_offset, p.obs_cam, p.obs_dir, p.obs_calibrated, p.cam_center, p.pt_xyz, opt,
                      pair_i=e, pair_j=e, pair_dir=np.zeros((0, 3)))
    assert not ok  # gp.cc:41-45


def test_balanced_weight_scales_point_losses():
    # POINTS_AND_CAMERAS_BALANCED: rho_pt = w * huber with w = reweight * #pairs / #tracks (gp.cc:223-253); at the start
    # point the cost must be  sum_pairs huber + w * sum_obs huber
    p = synthetic.make_gp_problem(num_cams=12, num_pts=80, seed=5)
    pi, pj, pd = _pairs(p, np.random.default_rng(1), noise=0.05)
    common = dict(generate_random_positions=False, generate_random_points=False, generate_scales=True)
    costs = {}
    for ctype, rw in ((gp.POINTS_AND_CAMERAS, 1.0), (gp.POINTS_AND_CAMERAS_BALANCED, 3.0), (gp.ONLY_CAMERAS, 1.0), (gp.ONLY_POINTS, 1.0)):
        opt = gp.GlobalPositionerOptions(constraint_type=ctype, constraint_reweight_scale=rw, **common)
        opt.lm.max_num_iterations = 0
        kw = {} if ctype == gp.ONLY_POINTS else dict(pair_i=pi, pair_j=pj, pair_dir=pd)
        _, _, _, s = gp.solve(p.num_cams, p.pt_offset, p.obs_cam, p.obs_dir, p.obs_calibrated, p.gt_center.copy(), p.pt_xyz, opt, **kw)
        costs[ctype] = s.initial_cost
    w = 3.0 * len(pi) / (p.pt_offset.shape[0] - 1)
    assert np.isclose(costs[gp.POINTS_AND_CAMERAS], costs[gp.ONLY_CAMERAS] + costs[gp.ONLY_POINTS], rtol=1e-12)
    assert np.isclose(costs[gp.POINTS_AND_CAMERAS_BALANCED], costs[gp.ONLY_CAMERAS] + w * costs[gp.ONLY_POINTS], rtol=1e-12)
