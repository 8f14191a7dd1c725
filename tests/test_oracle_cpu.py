"""The multithreaded C++ restatement (oracle/csrc, oracle/cpu.py) against the numpy oracle it restates.

Both are test infrastructure.  The numpy oracle solves every LM step with a dense / SuperLU factorisation;
the C++ one eliminates the same blocks in closed form and solves the reduced camera system by PCG to
1e-14 — same LM decisions (oracle/lm.py <-> oracle/csrc/orc_lm.hpp), so iteration counts must be identical
and the solutions equal to solver precision.  This is what lets the C++ oracle stand in for the numpy one
at the sizes of BASELINE.json configs[2] / configs[3] (tests/test_fullsize_gpu.py).
"""
import numpy as np
import pytest

from glomap_amd import synthetic
from oracle import ba as oba
from oracle import cpu
from oracle import gp as ogp
from oracle import ra as ora


def _gp_args(p):
    return (p.num_cams, p.pt_offset, p.obs_cam, p.obs_dir, p.obs_calibrated, p.cam_center, p.pt_xyz)


def _ba_args(p):
    return (p.num_cams, p.pt_offset, p.obs_cam, p.obs_xy, p.cam_intr, p.intr_model, p.fixed_cam, p.cam_q, p.cam_t,
            p.pt_xyz, p.intr_params)


@pytest.mark.parametrize("seed", [0, 3])
def test_gp_cpp_matches_numpy_oracle(seed):
    p = synthetic.make_gp_problem(40, 900, seed=seed)
    if seed == 3:  # uncalibrated cameras take the ScaledLoss branch (gp.cc:242-255)
        p.obs_calibrated = (p.obs_cam % 3 != 0).astype(np.uint8)
    ok, c, X, s = ogp.solve(*_gp_args(p))
    ok2, c2, X2, s2 = cpu.gp_solve(*_gp_args(p))
    assert ok and ok2
    assert (s2.iterations, s2.successful_steps) == (s.iterations, s.successful_steps)
    assert abs(s2.final_cost - s.final_cost) <= 1e-9 * s.final_cost
    assert s2.max_linear_residual < 1e-8
    scale = np.abs(c).max()
    assert np.abs(c - c2).max() <= 1e-7 * scale
    assert np.abs(X - X2).max() <= 1e-7 * scale


def test_gp_cpp_option_flags():
    p = synthetic.make_gp_problem(30, 500, seed=1)
    for kw in (dict(optimize_scales=False), dict(optimize_points=False, generate_random_points=False),
               dict(generate_scales=False, generate_random_positions=False, generate_random_points=False)):
        opt = ogp.GlobalPositionerOptions(**kw)
        pp = synthetic.make_gp_problem(30, 500, seed=1)
        pp.cam_center = pp.gt_center + 0.5
        pp.pt_xyz = pp.gt_xyz + 0.1
        ok, c, X, s = ogp.solve(*_gp_args(pp), options=opt)
        ok2, c2, X2, s2 = cpu.gp_solve(*_gp_args(pp), options=opt)
        assert ok == ok2
        assert s2.iterations == s.iterations, kw
        assert np.abs(c - c2).max() <= 1e-6 * np.abs(c).max(), kw


def test_gp_cpp_is_thread_count_independent():
    p = synthetic.make_gp_problem(40, 900, seed=0)
    a = cpu.gp_solve(*_gp_args(p), threads=1)
    b = cpu.gp_solve(*_gp_args(p), threads=4)
    assert np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2])
    assert a[3].final_cost == b[3].final_cost


@pytest.mark.parametrize("shared", [False, True])
def test_ba_cpp_matches_numpy_oracle(shared):
    p = synthetic.make_ba_problem(25, 600, seed=2, shared_intrinsics=shared)
    r = oba.solve(*_ba_args(p))
    r2 = cpu.ba_solve(*_ba_args(p))
    assert r[0] and r2[0]
    assert (r2[5].iterations, r2[5].successful_steps) == (r[5].iterations, r[5].successful_steps)
    assert abs(r2[5].final_cost - r[5].final_cost) <= 1e-9 * r[5].final_cost
    assert np.abs(r[1] - r2[1]).max() < 1e-9      # quaternions
    assert np.abs(r[2] - r2[2]).max() < 1e-6      # translations
    assert np.abs(r[3] - r2[3]).max() < 1e-6      # points
    assert np.abs(r[4] - r2[4]).max() < 1e-6      # intrinsics


@pytest.mark.parametrize("model", [0, 1, 3, 4, 5, 7, 8, 9])
def test_ba_cpp_camera_models(model):
    from glomap_amd.flat import CAMERA_NUM_PARAMS

    p = synthetic.make_ba_problem(12, 250, seed=model)
    K = p.intr_params.shape[0]
    f, cx, cy = 1200.0, 640.0, 480.0
    par = {0: [f, cx, cy], 1: [f, f * 1.01, cx, cy], 3: [f, cx, cy, 0.02, 0.0], 4: [f, f, cx, cy, 0.02, 0.0, 0.0, 0.0],
           5: [f, f, cx, cy, 0.3, 0.0, 0.0, 0.0], 7: [f, f, cx, cy, 0.3], 8: [f, cx, cy, 0.3], 9: [f, cx, cy, 0.3, 0.0]}[model]
    p.intr_model[:] = model
    p.intr_params[:] = 0.0
    p.intr_params[:, : CAMERA_NUM_PARAMS[model]] = par
    # observations were generated with SIMPLE_RADIAL k = 0.02: every model here starts from a slightly wrong
    # projection, which is what BA is for.  (The fisheye / FOV models cannot imitate that projection at all — the solve
    # then wanders for 50 ill-conditioned iterations in which rounding differences between the two oracles grow — so
    # their observations are regenerated through the model itself.)
    if model >= 5:
        from glomap_amd import so3

        obs_pt = np.repeat(np.arange(p.num_pts), np.diff(p.pt_offset))
        xc = np.einsum("mij,mj->mi", so3.quat_to_rotmat(p.gt_q)[p.obs_cam], p.gt_xyz[obs_pt]) + p.gt_t[p.obs_cam]
        ik = p.cam_intr[p.obs_cam]
        uv, _, _, valid = oba.project(p.intr_model[ik], p.intr_params[ik], xc)
        assert valid.all()
        p.obs_xy = uv + np.random.default_rng(model).normal(0, 0.3, uv.shape)
    for staged in (dict(optimize_rotations=False), dict()):
        opt = oba.BundleAdjusterOptions(**staged)
        r = oba.solve(*_ba_args(p), options=opt)
        r2 = cpu.ba_solve(*_ba_args(p), options=opt)
        assert r[0] == r2[0]
        assert r2[5].iterations == r[5].iterations
        # (OPENCV_FISHEYE: theta^6 / theta^8 coefficients on a 40-degree field of view are barely determined — the trust
        # region runs to its 1e16 cap — and the two implementations' rounding shows at 3e-8 in the final cost)
        tol = 1e-8 if model < 5 else 1e-6
        assert abs(r2[5].final_cost - r[5].final_cost) <= tol * r[5].final_cost
        assert np.abs(r[1] - r2[1]).max() < tol
        if model != 5:
            assert np.abs(r[4] - r2[4]).max() < 1e-5 * 1200.0
        else:
            assert np.abs(r[4][:, :4] - r2[4][:, :4]).max() < 1e-5 * 1200.0  # focal lengths, principal point


@pytest.mark.parametrize("shared", [True, False])
@pytest.mark.parametrize("model,par", [
    (6, [1200, 1190, 640, 480, 0.02, -0.01, 0.001, -0.002, 0.003, 0.01, -0.004, 0.002]),                      # FULL_OPENCV
    (10, [1200, 1190, 640, 480, 0.02, -0.01, 0.001, -0.002, 0.004, -0.002, 0.0015, -0.001]),                   # THIN_PRISM_FISHEYE
    (11, [1200, 1190, 640, 480, 0.02, -0.01, 0.004, -0.002, 0.001, -0.0005, 0.001, -0.002, 0.0015, -0.0008, -0.001, 0.0005]),
])
def test_ba_cpp_wide_camera_models(model, par, shared):
    """The 16-wide unit of the C++ oracle (orc_ba_wide.cc: orc_ba.cc compiled with ORC_BA_MAXP 16) against the numpy oracle on
    the camera models with more than eight parameters: same LM decisions, solutions to solver precision."""
    from glomap_amd import so3

    p = synthetic.make_ba_problem(15, 300, seed=5, pixel_noise=0.0, outlier_ratio=0.0, shared_intrinsics=shared)
    p.intr_model[:] = model
    p.intr_params = np.zeros((p.num_intr, 16))
    p.intr_params[:, : len(par)] = par
    obs_pt = np.repeat(np.arange(p.num_pts), np.diff(p.pt_offset))
    xc = np.einsum("mij,mj->mi", so3.quat_to_rotmat(p.gt_q)[p.obs_cam], p.gt_xyz[obs_pt]) + p.gt_t[p.obs_cam]
    ik = p.cam_intr[p.obs_cam]
    uv, _, _, valid = oba.project(p.intr_model[ik], p.intr_params[ik], xc)
    assert valid.all()
    p.obs_xy = uv + np.random.default_rng(0).normal(0, 0.3, uv.shape)
    r = oba.solve(*_ba_args(p))
    r2 = cpu.ba_solve(*_ba_args(p))
    assert r[0] and r2[0] and r2[4].shape == (p.num_intr, 16)
    assert (r2[5].iterations, r2[5].successful_steps) == (r[5].iterations, r[5].successful_steps)
    assert abs(r2[5].initial_cost - r[5].initial_cost) <= 1e-12 * r[5].initial_cost
    assert abs(r2[5].final_cost - r[5].final_cost) <= 1e-8 * r[5].final_cost
    assert np.abs(r[1] - r2[1]).max() < 1e-7 and np.abs(r[2] - r2[2]).max() < 1e-5
    assert np.array_equal(r2[4][:, len(par):], np.zeros((p.num_intr, 16 - len(par))))  # the unused tail of the rows


def test_ba_cpp_wide_unit_equals_the_narrow_unit_on_padded_rows():
    """orc_ba_solve on [K, 8] rows and orc_ba_solve_wide on the same rows zero-padded to 16: the same arithmetic, bit for bit
    (every width of orc_ba.cc scales with ORC_BA_MAXP)."""
    for shared in (True, False):
        p = synthetic.make_ba_problem(30, 800, seed=3, pixel_noise=0.5, outlier_ratio=0.01, shared_intrinsics=shared, intr_noise=0.01)
        a = cpu.ba_solve(*_ba_args(p), threads=4)
        w = p.copy()
        w.intr_params = np.zeros((p.num_intr, 16))
        w.intr_params[:, :8] = p.intr_params
        b = cpu.ba_solve(*_ba_args(w), threads=4)
        assert a[0] and b[0] and a[5].iterations == b[5].iterations and a[5].final_cost == b[5].final_cost
        assert np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2]) and np.array_equal(a[3], b[3])
        assert np.array_equal(a[4], b[4][:, :8]) and not b[4][:, 8:].any()


def test_ba_cpp_order_switch_changes_only_rounding():
    p = synthetic.make_ba_problem(25, 600, seed=2)
    a = cpu.ba_solve(*_ba_args(p))
    b = cpu.ba_solve(*_ba_args(p), order=1)
    assert a[5].iterations == b[5].iterations
    assert np.abs(a[1] - b[1]).max() < 1e-9


@pytest.mark.parametrize("n,succ,kw", [(200, 10, {}), (150, 8, dict(weight_type=ora.HALF_NORM)), (120, 6, dict(use_weight=True))])
def test_ra_cpp_matches_numpy_oracle(n, succ, kw):
    p = synthetic.make_ring_view_graph(n, succ, seed=n)
    if kw.get("use_weight"):
        p.edge_weight = np.random.default_rng(0).uniform(0.5, 1.5, p.num_edges)
    opt = ora.RotationEstimatorOptions(**kw)
    tr = ora.RaTrace()
    args = (p.num_nodes, p.edge_i, p.edge_j, p.edge_q, p.edge_weight, p.edge_ninl, p.node_aa0, p.fixed_node)
    ok, rot = ora.estimate_rotations(*args, options=opt, trace=tr)
    rep = {}
    ok2, rot2 = cpu.ra_estimate_rotations(*args, options=opt, report=rep)
    assert ok and ok2
    assert (rep["l1_iterations"], rep["irls_iterations"]) == (tr.l1_iterations, tr.irls_iterations)
    # HALF_NORM weights e2^-0.75 span many orders of magnitude against the unit-weight gauge rows: the two
    # factorisations (SuperLU there, skyline Cholesky here) then agree to ~1e-8 only
    assert np.abs(rot - rot2).max() < (1e-7 if kw.get("weight_type") else 1e-9)


def test_ra_cpp_shuffled_node_ids():
    """Node numbering must not matter for the skyline factor (reverse Cuthill-McKee finds the band again)."""
    p = synthetic.make_ring_view_graph(400, 10, seed=5)
    perm = np.random.default_rng(1).permutation(p.num_nodes)
    ei, ej = perm[p.edge_i].astype(np.int32), perm[p.edge_j].astype(np.int32)
    aa0 = np.zeros_like(p.node_aa0)
    aa0[perm] = p.node_aa0
    rep = {}
    ok, rot = cpu.ra_estimate_rotations(p.num_nodes, ei, ej, p.edge_q, p.edge_weight, p.edge_ninl, aa0, int(perm[0]),
                                        report=rep)
    ok2, rot2 = ora.estimate_rotations(p.num_nodes, ei, ej, p.edge_q, p.edge_weight, p.edge_ninl, aa0, int(perm[0]))
    assert ok and ok2
    assert rep["profile_entries"] < 400 * 45
    assert np.abs(rot - rot2).max() < 1e-9


@pytest.mark.parametrize("kind", ["geometric", "hub", "chords"])
def test_ra_cpp_matches_numpy_oracle_on_non_ring_graphs(kind):
    p = synthetic.make_view_graph(kind, 600, 16, seed=2)
    args = (p.num_nodes, p.edge_i, p.edge_j, p.edge_q, p.edge_weight, p.edge_ninl, p.node_aa0, p.fixed_node)
    tr = ora.RaTrace()
    ok, rot = ora.estimate_rotations(*args, trace=tr)
    rep = {}
    ok2, rot2 = cpu.ra_estimate_rotations(*args, report=rep)
    assert ok and ok2
    assert (rep["l1_iterations"], rep["irls_iterations"]) == (tr.l1_iterations, tr.irls_iterations)
    assert np.abs(rot - rot2).max() < 1e-9


@pytest.mark.parametrize("ctype", [1, 2, 3])  # ONLY_CAMERAS, POINTS_AND_CAMERAS_BALANCED, POINTS_AND_CAMERAS
def test_cpp_gp_constraint_types_match_the_numpy_oracle(ctype):
    """The camera-to-camera constraint types (gp.cc:42-71, 167-255) in the C++ oracle against their numpy restatement: same
    start, same LM path, same result."""
    from test_oracle_gp import _pairs

    p = synthetic.make_gp_problem(num_cams=60, num_pts=1500, seed=7, dir_noise=2e-3, outlier_ratio=0.02)
    pi, pj, pd = _pairs(p, np.random.default_rng(7), noise=2e-3)
    opt = ogp.GlobalPositionerOptions(constraint_type=ctype, constraint_reweight_scale=2.0)
    args = (p.num_cams, p.pt_offset, p.obs_cam, p.obs_dir, p.obs_calibrated, p.cam_center, p.pt_xyz, opt)
    ok_n, c_n, X_n, s_n = ogp.solve(*args, pair_i=pi, pair_j=pj, pair_dir=pd)
    ok_c, c_c, X_c, s_c = cpu.gp_solve(*args, pair_i=pi, pair_j=pj, pair_dir=pd)
    assert ok_n and ok_c and s_n.iterations == s_c.iterations
    assert abs(s_n.initial_cost - s_c.initial_cost) <= 1e-12 * s_n.initial_cost
    assert abs(s_n.final_cost - s_c.final_cost) <= 1e-9 * s_n.final_cost
    assert synthetic.center_errors_after_sim3(c_c, c_n).max() < 1e-9
    assert np.abs(X_c - X_n).max() <= 1e-7 * np.abs(X_n).max()
    if ctype == 1:
        assert np.array_equal(X_c, p.pt_xyz)
    ok, *_ = cpu.gp_solve(*args)  # no pairs: gp.cc:41-45
    assert not ok
