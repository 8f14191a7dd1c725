"""Calibrated (known) multi-camera rigs — the shape of the reference's WithoutNoiseWithNonTrivialKnownRig mapper test
(glomap/controllers/global_mapper_test.cc:89-126): frames of 2-3 cameras with known cam_from_rig.

  GP: RigBATAPairwiseDirectionError with the rig scale constant (cost_function.h:49-82, global_positioning.cc:318-350, 470-478)
  BA: colmap::RigReprojErrorConstantRigCostFunctor (bundle_adjustment.cc:147-160, optimize_rig_poses = false)

  BA: colmap::RigReprojErrorCostFunctor (bundle_adjustment.cc:161-179, optimize_rig_poses = true): cam_from_rig blocks
  GP: RigUnknownBATAPairwiseDirectionError (cost_function.h:90-136, global_positioning.cc:354-368): unknown translations

CPU: the two oracles against each other and against ground truth.  GPU: the HIP path (sweeps over images, LM / PCG state
per frame and per sensor block) through the C ABI against the oracles."""
import numpy as np
import pytest

from glomap_amd import so3, synthetic
from oracle import ba as oba
from oracle import cpu
from oracle import gp as ogp


def _gp_kw(p):
    return dict(image_frame=p.image_frame, image_offset=p.image_offset)


def _ba_kw(p):
    return dict(image_frame=p.image_frame, image_cam_from_rig=p.image_cam_from_rig, image_intr=p.image_intr)


def _gp_args(p):
    return (p.num_cams, p.pt_offset, p.obs_cam, p.obs_dir, p.obs_calibrated, p.cam_center, p.pt_xyz)


def _ba_args(p):
    return (p.num_cams, p.pt_offset, p.obs_cam, p.obs_xy, None, p.intr_model, p.fixed_cam, p.cam_q, p.cam_t, p.pt_xyz,
            p.intr_params)


def test_oracles_agree_on_rig_problems_and_recover_ground_truth():
    gp, ba, info = synthetic.make_rig_problems(14, 2, 400, seed=0)
    a = ogp.solve(*_gp_args(gp), **_gp_kw(gp))
    b = cpu.gp_solve(*_gp_args(gp), **_gp_kw(gp))
    assert a[0] and b[0] and a[3].iterations == b[3].iterations
    assert np.abs(a[1] - b[1]).max() < 1e-7
    # the rig translations are metric: the solution is fixed up to a RIGID motion, the Sim(3) fit must find scale 1
    scale, _, _ = synthetic.align_sim3(a[1], gp.gt_center)
    assert abs(scale - 1.0) < 2e-2
    assert synthetic.center_errors_after_sim3(a[1], gp.gt_center).max() < 0.1
    a = oba.solve(*_ba_args(ba), **_ba_kw(ba))
    b = cpu.ba_solve(*_ba_args(ba), **_ba_kw(ba))
    assert a[0] and b[0] and a[5].iterations == b[5].iterations
    assert a[5].final_cost < 1e-9 * a[5].initial_cost  # noise-free: zero reprojection error
    assert np.abs(a[1] - b[1]).max() < 1e-9 and np.abs(a[2] - b[2]).max() < 1e-7


def test_identity_rigs_reduce_to_the_trivial_problem():
    """One reference sensor per frame with identity cam_from_rig: bit-for-bit the plain problems in the oracle."""
    p = synthetic.make_ba_problem(12, 250, seed=1)
    ident = np.zeros((12, 7))
    ident[:, 0] = 1
    a = oba.solve(p.num_cams, p.pt_offset, p.obs_cam, p.obs_xy, p.cam_intr, p.intr_model, p.fixed_cam, p.cam_q, p.cam_t,
                  p.pt_xyz, p.intr_params)
    b = oba.solve(p.num_cams, p.pt_offset, p.obs_cam, p.obs_xy, None, p.intr_model, p.fixed_cam, p.cam_q, p.cam_t, p.pt_xyz,
                  p.intr_params, image_frame=np.arange(12), image_cam_from_rig=ident, image_intr=p.cam_intr)
    assert a[5].iterations == b[5].iterations and np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2])


def _miscalibrated(info, seed, rot_deg=1.0, trans=0.05):
    """Sensor blocks of make_rig_problems perturbed: the start values of optimize_rig_poses."""
    rng = np.random.default_rng(seed)
    sg = info["sensor_cam_from_rig"]
    s0 = sg.copy()
    dq = so3.rotmat_to_quat(so3.aa_to_rotmat(rng.normal(0, np.radians(rot_deg), (sg.shape[0], 3))))
    s0[:, :4] = oba.quat_mul(dq, sg[:, :4])
    s0[:, 4:] += rng.normal(0, trans, (sg.shape[0], 3))
    return s0


def _with_sensors(ba, info, s0):
    p = ba.copy()
    p.image_sensor = info["sensor_block"].copy()
    p.sensor_cam_from_rig = s0.copy()
    return p


def _sens_kw(p):
    return dict(_ba_kw(p), image_sensor=p.image_sensor, sensor_cam_from_rig=p.sensor_cam_from_rig)


def test_oracle_refines_miscalibrated_rigs():
    """optimize_rig_poses in the oracle: from cam_from_rig blocks that are 1 degree / 5 cm off the noise-free problem goes
    to zero reprojection error and the (observable) cam_from_rig rotations come back; with the option off the same tables
    are constants and the error stays."""
    _, ba, info = synthetic.make_rig_problems(14, 3, 500, seed=0)
    p = _with_sensors(ba, info, _miscalibrated(info, 1))
    r = oba.solve(*_ba_args(p), **_sens_kw(p))
    assert r[0] and r[5].final_cost > 0.1 * r[5].initial_cost and not hasattr(r[5], "sensor_cam_from_rig")
    r = oba.solve(*_ba_args(p), options=oba.BundleAdjusterOptions(optimize_rig_poses=True), **_sens_kw(p))
    assert r[0] and r[5].final_cost < 1e-9 * r[5].initial_cost
    sc, sg = r[5].sensor_cam_from_rig, info["sensor_cam_from_rig"]
    assert so3.rotation_angle_deg(so3.quat_to_rotmat(sc[:, :4]), so3.quat_to_rotmat(sg[:, :4])).max() < 1e-4
    # the constant frame stays (ba.cc:261-266)
    assert np.array_equal(r[1][p.fixed_cam], p.cam_q[p.fixed_cam]) and np.array_equal(r[2][p.fixed_cam], p.cam_t[p.fixed_cam])


def _unk_kw(p):
    return dict(image_frame=p.image_frame, image_offset=p.image_offset, image_sensor=p.image_sensor,
                image_sensor_rot=p.image_sensor_rot, sensor_center=p.sensor_center)


def test_oracle_estimates_unknown_rig_translations():
    """RigUnknownBATAPairwiseDirectionError in the oracle (cost_function.h:90-136, gp.cc:354-368): with every non-reference
    cam_from_rig translation unknown the noise-free problem still goes to zero cost, and frame centres and sensor centres
    come out in ONE common scale (the scale itself is free again: no metric baseline is left)."""
    gp, _, info = synthetic.make_rig_problems(14, 3, 500, seed=0)
    p = synthetic.forget_rig_translations(gp, info)
    ok, cen, X, summ = ogp.solve(*_gp_args(p), **_unk_kw(p))
    assert ok and summ.final_cost < 1e-12 * summ.initial_cost
    scale, _, _ = synthetic.align_sim3(cen, gp.gt_center)
    assert synthetic.center_errors_after_sim3(cen, gp.gt_center).max() < 1e-6
    assert np.abs(summ.sensor_center * scale - info["sensor_center"]).max() < 1e-5


@pytest.mark.parametrize("frames,cams,pts,noise", [(14, 3, 500, 0.0), (30, 3, 3000, 1.0)])
def test_cpp_oracle_agrees_on_unknown_rig_translations(frames, cams, pts, noise):
    gp, _, info = synthetic.make_rig_problems(frames, cams, pts, seed=4, dir_noise=1e-3 * noise, outlier_ratio=0.01 * noise)
    p = synthetic.forget_rig_translations(gp, info)
    a = ogp.solve(*_gp_args(p), **_unk_kw(p))
    b = cpu.gp_solve(*_gp_args(p), **_unk_kw(p))
    assert a[0] and b[0] and a[3].iterations == b[3].iterations
    assert abs(a[3].initial_cost - b[3].initial_cost) <= 1e-12 * a[3].initial_cost  # same draws, sensor centres included
    assert np.abs(a[1] - b[1]).max() < 1e-7 and np.abs(a[3].sensor_center - b[3].sensor_center).max() < 1e-9


@pytest.mark.gpu
@pytest.mark.parametrize("frames,cams,pts,noise", [(14, 3, 500, 0.0), (30, 3, 3000, 1.0)])
def test_gp_with_unknown_rig_translations_matches_oracle(gsfm_ctx, frames, cams, pts, noise):
    """The same through the C ABI: centre blocks of the unknown sensors behind the frames, same random start (the sensor
    draws come last in the stream, gp.cc:442-456), same LM path as the exact-solve oracle."""
    from glomap_amd import estimators

    gp, _, info = synthetic.make_rig_problems(frames, cams, pts, seed=4, dir_noise=1e-3 * noise, outlier_ratio=0.01 * noise)
    p = synthetic.forget_rig_translations(gp, info)
    opt = estimators.GlobalPositionerOptions()
    opt.solver_options.pcg_relative_tolerance = 1e-10
    rc, cen, xyz, rep = estimators.gp_solve(p, opt, ctx=gsfm_ctx)
    assert rc == 0
    ok, c_o, X_o, s = ogp.solve(*_gp_args(p), **_unk_kw(p))
    assert ok
    assert abs(rep["initial_cost"] - s.initial_cost) <= 1e-12 * s.initial_cost  # same random start, sensor centres included
    assert abs(rep["iterations"] - s.iterations) <= 2
    assert abs(rep["final_cost"] - s.final_cost) <= 1e-3 * s.final_cost + 1e-9
    assert synthetic.center_errors_after_sim3(cen, c_o).max() < 1e-3
    scale, _, _ = synthetic.align_sim3(cen, c_o)
    cs, cs_o = rep["sensor_center"], s.sensor_center
    assert np.abs(cs * scale - cs_o).max() < 1e-3 * np.abs(cs_o).max()
    if noise == 0.0:
        sc, _, _ = synthetic.align_sim3(cen, gp.gt_center)
        assert np.abs(cs * sc - info["sensor_center"]).max() < 1e-4


@pytest.mark.parametrize("frames,cams,pts,noise,rot", [(14, 2, 400, 0.0, True), (30, 3, 3000, 0.5, True)])
def test_cpp_oracle_agrees_on_optimised_rig_poses(frames, cams, pts, noise, rot):
    """The two restatements of RigReprojErrorCostFunctor against each other (numpy: sparse direct solves; C++: dense or PCG
    solves of the reduced system with sensor blocks behind the frames)."""
    _, ba, info = synthetic.make_rig_problems(frames, cams, pts, seed=11, pixel_noise=noise)
    p = _with_sensors(ba, info, _miscalibrated(info, 12, rot_deg=0.5, trans=0.03))
    opt = oba.BundleAdjusterOptions(optimize_rig_poses=True, optimize_rotations=rot)
    a = oba.solve(*_ba_args(p), options=opt, **_sens_kw(p))
    b = cpu.ba_solve(*_ba_args(p), options=opt, **_sens_kw(p))
    assert a[0] and b[0] and a[5].iterations == b[5].iterations
    assert abs(a[5].final_cost - b[5].final_cost) <= 1e-9 * max(a[5].final_cost, 1e-6)
    assert np.abs(a[1] - b[1]).max() < 1e-9 and np.abs(a[2] - b[2]).max() < 1e-7
    assert np.abs(a[5].sensor_cam_from_rig - b[5].sensor_cam_from_rig).max() < 1e-8
    # and a table without the option is a constant in both
    c = cpu.ba_solve(*_ba_args(p), **_sens_kw(p))
    d = oba.solve(*_ba_args(p), **_sens_kw(p))
    assert c[0] and d[0] and c[5].iterations == d[5].iterations and np.abs(c[1] - d[1]).max() < 1e-8


@pytest.mark.gpu
@pytest.mark.parametrize("frames,cams,pts,noise,rot", [(14, 2, 400, 0.0, True), (30, 3, 3000, 0.5, True), (16, 3, 800, 0.3, False)])
def test_ba_with_optimised_rig_poses_matches_oracle(gsfm_ctx, frames, cams, pts, noise, rot):
    """RigReprojErrorCostFunctor through the C ABI: frames, cam_from_rig blocks, points and intrinsics against the numpy
    oracle's exact-solve LM.  rot=False: optimize_rotations off — the frame rotations stay bit for bit, the cam_from_rig
    rotations are still optimised (ba.cc:296-309 sets their manifold only)."""
    from glomap_amd import estimators

    _, ba, info = synthetic.make_rig_problems(frames, cams, pts, seed=11, pixel_noise=noise)
    p = _with_sensors(ba, info, _miscalibrated(info, 12, rot_deg=0.5, trans=0.03))
    opt = estimators.BundleAdjusterOptions(optimize_rig_poses=True, optimize_rotations=rot)
    opt.solver_options.pcg_relative_tolerance = 1e-10
    rc, q, t, X, intr, rep = estimators.ba_solve(p, opt, ctx=gsfm_ctx)
    assert rc == 0
    r = oba.solve(*_ba_args(p), options=oba.BundleAdjusterOptions(optimize_rig_poses=True, optimize_rotations=rot), **_sens_kw(p))
    assert r[0]
    assert abs(rep["initial_cost"] - r[5].initial_cost) <= 1e-10 * r[5].initial_cost
    assert abs(rep["iterations"] - r[5].iterations) <= (1 if rot else 3)
    if noise == 0.0:
        assert rep["final_cost"] < 1e-9 * rep["initial_cost"]
    else:
        assert abs(rep["final_cost"] - r[5].final_cost) <= 1e-6 * r[5].final_cost
    sc, so = rep["sensor_cam_from_rig"], r[5].sensor_cam_from_rig
    assert np.radians(so3.rotation_angle_deg(so3.quat_to_rotmat(sc[:, :4]), so3.quat_to_rotmat(so[:, :4]))).max() < 1e-5
    if rot:
        ang = np.radians(so3.rotation_angle_deg(so3.quat_to_rotmat(q), so3.quat_to_rotmat(r[1])))
        assert ang.max() < 1e-5
    else:
        assert np.array_equal(q, p.cam_q)
    if noise > 0.0:  # (noise-free: the free scale of the minimum makes translations path dependent)
        assert np.abs(t - r[2]).max() < 1e-3 * 50.0 and np.abs(sc[:, 4:] - so[:, 4:]).max() < 1e-3
    assert np.array_equal(q[p.fixed_cam], p.cam_q[p.fixed_cam]) and np.array_equal(t[p.fixed_cam], p.cam_t[p.fixed_cam])


@pytest.mark.gpu
def test_sensor_tables_are_constants_without_the_option(gsfm_ctx):
    """optimize_rig_poses off: a sensor table is only another way to write image_cam_from_rig — same solve, table untouched."""
    from glomap_amd import estimators

    _, ba, info = synthetic.make_rig_problems(14, 2, 400, seed=3, pixel_noise=0.5)
    a = estimators.ba_solve(ba, ctx=gsfm_ctx)
    p = _with_sensors(ba, info, info["sensor_cam_from_rig"])
    p.image_cam_from_rig = p.image_cam_from_rig.copy()
    p.image_cam_from_rig[p.image_sensor >= 0] = np.array([1.0, 0, 0, 0, 0, 0, 0])  # superseded by the table
    b = estimators.ba_solve(p, ctx=gsfm_ctx)
    assert a[0] == 0 and b[0] == 0 and a[5]["iterations"] == b[5]["iterations"]
    assert np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2]) and np.array_equal(a[4], b[4])
    assert np.array_equal(b[5]["sensor_cam_from_rig"], info["sensor_cam_from_rig"])


@pytest.mark.gpu
@pytest.mark.parametrize("frames,cams,pts,noise", [(14, 2, 400, 0.0), (40, 3, 4000, 1.0)])
def test_gp_with_known_rigs_matches_oracle(gsfm_ctx, frames, cams, pts, noise):
    from glomap_amd import estimators

    gp, _, _ = synthetic.make_rig_problems(frames, cams, pts, seed=2, dir_noise=1e-3 * noise, outlier_ratio=0.01 * noise)
    opt = estimators.GlobalPositionerOptions()
    opt.solver_options.pcg_relative_tolerance = 1e-10
    rc, cen, xyz, rep = estimators.gp_solve(gp, opt, ctx=gsfm_ctx)
    assert rc == 0
    ok, c_o, X_o, s = cpu.gp_solve(*_gp_args(gp), **_gp_kw(gp))
    assert ok
    assert abs(rep["initial_cost"] - s.initial_cost) <= 1e-12 * s.initial_cost  # same random start
    assert abs(rep["iterations"] - s.iterations) <= 2
    assert abs(rep["final_cost"] - s.final_cost) <= 1e-3 * s.final_cost + 1e-9
    assert synthetic.center_errors_after_sim3(cen, c_o).max() < 1e-3  # relative to the extent (the helper divides)
    scale, _, _ = synthetic.align_sim3(cen, gp.gt_center)
    assert abs(scale - 1.0) < 3e-2  # metric rig offsets fix the scale


@pytest.mark.gpu
@pytest.mark.parametrize("frames,cams,pts,noise", [(14, 2, 400, 0.0), (40, 3, 4000, 0.5)])
def test_ba_with_known_rigs_matches_oracle(gsfm_ctx, frames, cams, pts, noise):
    from glomap_amd import estimators

    _, ba, _ = synthetic.make_rig_problems(frames, cams, pts, seed=3, pixel_noise=noise)
    rc, q, t, X, intr, rep = estimators.ba_solve(ba, ctx=gsfm_ctx)
    assert rc == 0
    r = cpu.ba_solve(*_ba_args(ba), **_ba_kw(ba))
    assert r[0]
    assert abs(rep["initial_cost"] - r[5].initial_cost) <= 1e-10 * r[5].initial_cost
    assert rep["iterations"] == r[5].iterations
    if noise == 0.0:
        assert rep["final_cost"] < 1e-9 * rep["initial_cost"]
    else:
        assert abs(rep["final_cost"] - r[5].final_cost) <= 1e-6 * r[5].final_cost
    ang = np.radians(so3.rotation_angle_deg(so3.quat_to_rotmat(q), so3.quat_to_rotmat(r[1])))
    assert ang.max() < 1e-6
    assert np.abs(t - r[2]).max() < 1e-5 * 50.0
    assert np.abs(intr - r[4]).max() < 1e-4
    # the constant frame is untouched (ba.cc:261-266)
    assert np.array_equal(q[ba.fixed_cam], ba.cam_q[ba.fixed_cam]) and np.array_equal(t[ba.fixed_cam], ba.cam_t[ba.fixed_cam])


@pytest.mark.gpu
def test_identity_rig_tables_give_the_trivial_solution(gsfm_ctx):
    """The rig path with one identity sensor per frame solves the same problem as the trivial path (other block-Jacobi
    blocks — no joint pose + intrinsics blocks — so equal to solver tolerance, not bit for bit)."""
    from glomap_amd import estimators

    p = synthetic.make_ba_problem(20, 600, seed=5, shared_intrinsics=True)
    rc, q, t, X, intr, rep = estimators.ba_solve(p, ctx=gsfm_ctx)
    pr = p.copy()
    ident = np.zeros((p.num_cams, 7))
    ident[:, 0] = 1
    pr.image_frame, pr.image_cam_from_rig, pr.image_intr = np.arange(p.num_cams, dtype=np.int32), ident, p.cam_intr.copy()
    rc2, q2, t2, X2, intr2, rep2 = estimators.ba_solve(pr, ctx=gsfm_ctx)
    assert rc == 0 and rc2 == 0 and rep["iterations"] == rep2["iterations"]
    assert np.abs(q - q2).max() < 1e-8 and np.abs(t - t2).max() < 1e-6 and np.abs(intr - intr2).max() < 1e-5
    g = synthetic.make_gp_problem(20, 600, seed=5)
    rc, cen, xyz, rep = estimators.gp_solve(g, ctx=gsfm_ctx)
    g.image_frame, g.image_offset = np.arange(20, dtype=np.int32), np.zeros((20, 3))
    rc2, cen2, xyz2, rep2 = estimators.gp_solve(g, ctx=gsfm_ctx)
    assert rc == 0 and rc2 == 0 and rep["iterations"] == rep2["iterations"]
    assert np.abs(cen - cen2).max() < 1e-6 * np.abs(cen).max()


@pytest.mark.gpu
def test_known_rig_pipeline_recovers_the_scene(gsfm_ctx):
    """RA -> GP -> BA chained on a noise-free scene of 2 calibrated rigs x 2 cameras x 7 frames each (the configuration of
    global_mapper_test.cc:89-126), every stage through the C ABI; pins of that test: rotations 1e-2 deg, projection
    centres 1e-4 (:121-125)."""
    from glomap_amd import estimators
    from glomap_amd.flat import BaProblem, GpProblem

    gp, ba, info = synthetic.make_rig_problems(14, 2, 600, seed=7)
    N, I = gp.num_cams, gp.num_images
    R_cw, R_s, t_s = info["R_cw"], info["R_s"], info["t_s"]
    imf = gp.image_frame.astype(np.int64)
    # view graph: image pairs whose frames are at most 3 apart on the ring (pairs inside one frame included: the
    # estimator must skip them), exact relative rotations
    ii, jj = np.triu_indices(I, 1)
    d = np.abs(imf[ii] - imf[jj])
    near = np.minimum(d, N - d) <= 3
    ii, jj = ii[near], jj[near]
    q_rel = so3.rotmat_to_quat(R_cw[jj] @ np.transpose(R_cw[ii], (0, 2, 1)))
    ninl = np.random.default_rng(0).integers(30, 300, ii.shape[0]).astype(np.int32)
    rc, rot, rep = estimators.ra_solve_known_rigs(N, imf, ba.image_cam_from_rig, ii, jj, q_rel, ninl, ctx=gsfm_ctx)
    assert rc == 0
    R_est = so3.aa_to_rotmat(rot)
    assert synthetic.rotation_errors_deg(R_est, gp.cam_R).max() < 1e-2
    # global positioning with the ESTIMATED rotations (rays and rig offsets rotated by them, gp.cc:294-296, 329-333)
    Rcw_est = R_s @ R_est[imf]
    ray_cam = np.einsum("mij,mj->mi", R_cw[gp.obs_cam], gp.obs_dir)  # back to camera-frame rays (features_undist)
    gp2 = GpProblem(num_cams=N, num_pts=gp.num_pts, pt_offset=gp.pt_offset, obs_cam=gp.obs_cam,
                    obs_dir=np.ascontiguousarray(np.einsum("mji,mj->mi", Rcw_est[gp.obs_cam], ray_cam)),
                    obs_calibrated=gp.obs_calibrated, cam_center=np.zeros((N, 3)), pt_xyz=np.zeros((gp.num_pts, 3)),
                    image_frame=gp.image_frame, image_offset=np.ascontiguousarray(np.einsum("iba,ib->ia", Rcw_est, t_s)))
    rc, cen, xyz, rep = estimators.gp_solve(gp2, ctx=gsfm_ctx)
    assert rc == 0
    # bundle adjustment from the positioning result (positions only first, then everything: global_mapper.cc:201-223)
    q0 = so3.rotmat_to_quat(R_est)
    t0 = -np.einsum("nij,nj->ni", R_est, cen)
    ba2 = BaProblem(num_cams=N, num_pts=ba.num_pts, num_intr=ba.num_intr, pt_offset=ba.pt_offset, obs_cam=ba.obs_cam,
                    obs_xy=ba.obs_xy, cam_intr=ba.cam_intr, cam_q=q0, cam_t=t0, pt_xyz=xyz, intr_model=ba.intr_model,
                    intr_params=ba.intr_params, fixed_cam=0, image_frame=ba.image_frame,
                    image_cam_from_rig=ba.image_cam_from_rig, image_intr=ba.image_intr)
    rc, q, t, X, intr, rep = estimators.ba_solve(ba2, estimators.BundleAdjusterOptions(optimize_rotations=False), ctx=gsfm_ctx)
    assert rc == 0
    ba2.cam_t, ba2.pt_xyz, ba2.intr_params = t, X, intr
    rc, q, t, X, intr, rep = estimators.ba_solve(ba2, ctx=gsfm_ctx)
    assert rc == 0 and rep["final_cost"] < 1e-8 * max(1.0, rep["initial_cost"])
    R_fin = so3.quat_to_rotmat(q)
    c_fin = -np.einsum("nji,nj->ni", R_fin, t)
    assert synthetic.rotation_errors_deg(R_fin, gp.cam_R).max() < 1e-2
    err = synthetic.center_errors_after_sim3(c_fin, gp.gt_center)
    assert err.max() < 1e-4
    scale, _, _ = synthetic.align_sim3(c_fin, gp.gt_center)
    assert abs(scale - 1.0) < 1e-4  # the metric rig baselines fix the scale of the whole reconstruction
