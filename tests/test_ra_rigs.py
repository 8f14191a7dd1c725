"""Rotation averaging with cam_from_rig ROTATIONS among the unknowns — the estimator side of the reference's
WithoutNoiseWithNoneTrivialUnknownRig test (glomap/controllers/rotation_averager_test.cc:214-263):
global_rotation_averaging.cc:173-191 (cam blocks), :396-446 (their columns), :646-690 (their quaternion-average update),
:718-739 (composed residuals), rotation_initializer.cc:7-125 (ConvertRotationsFromImageToRig).

CPU: the numpy oracle and the flat restatement of ConvertRotationsFromImageToRig on noise-free data (exact start, stays
exact — the property the reference test pins at 1e-2 degrees).  GPU: the HIP path (image-level sweeps, PCG on frames + cam
blocks) against the oracle, iteration for iteration."""
import numpy as np
import pytest

from glomap_amd import estimators, so3, synthetic
from oracle import ra as ora


def make_rig_view_graph(frames=14, cams=3, seed=0, noise_deg=0.5, outlier=0.05, reach=3):
    """Image pairs of a scene of make_rig_problems: frames at most `reach` apart on the ring, all non-reference sensors
    with unknown cam_from_rig (one block per rig and sensor); pairs inside a frame between two images without a block are
    dropped like the reference does (gra.cc:300-304)."""
    gp, _, info = synthetic.make_rig_problems(frames, cams, 50, seed=seed)
    rng = np.random.default_rng(seed + 100)
    N, I = gp.num_cams, gp.num_images
    R_cw = info["R_cw"]
    imf = gp.image_frame.astype(np.int32)
    imc = info["sensor_block"].astype(np.int32)
    ii, jj = np.triu_indices(I, 1)
    d = np.abs(imf[ii].astype(np.int64) - imf[jj])
    near = np.minimum(d, N - d) <= reach
    near &= ~((imf[ii] == imf[jj]) & (imc[ii] < 0) & (imc[jj] < 0))
    ii, jj = ii[near].astype(np.int32), jj[near].astype(np.int32)
    R_rel = R_cw[jj] @ np.transpose(R_cw[ii], (0, 2, 1))
    if noise_deg:
        R_rel = so3.aa_to_rotmat(rng.normal(0, np.radians(noise_deg), (ii.size, 3))) @ R_rel
    out = rng.random(ii.size) < outlier
    if out.any():
        R_rel[out] = so3.aa_to_rotmat(rng.normal(0, 1.0, (int(out.sum()), 3)))
    return dict(N=N, C=int(imc.max()) + 1, imf=imf, imc=imc, ii=ii, jj=jj, q=so3.rotmat_to_quat(R_rel),
                ninl=rng.integers(30, 300, ii.size).astype(np.int32), R_f=gp.cam_R,
                R_c=so3.quat_to_rotmat(info["sensor_cam_from_rig"][:, :4]))


def _start(s):
    """Spanning tree over the images (gra.cc:87-138), then ConvertRotationsFromImageToRig."""
    aa_img = ora.maximum_spanning_tree_init(s["imf"].size, s["ii"], s["jj"], so3.quat_to_rotmat(s["q"]), s["ninl"],
                                            np.zeros((s["imf"].size, 3)))
    R_f, R_c = estimators.convert_rotations_from_image_to_rig(so3.aa_to_rotmat(aa_img), s["imf"], s["imc"], s["N"], s["C"])
    return so3.quat_to_aa(so3.rotmat_to_quat(R_f)), so3.quat_to_aa(so3.rotmat_to_quat(R_c))


def test_noise_free_unknown_rig_is_recovered_by_the_oracle():
    s = make_rig_view_graph(noise_deg=0.0, outlier=0.0)
    aa_f, aa_c = _start(s)
    assert so3.rotation_angle_deg(so3.aa_to_rotmat(aa_c), s["R_c"]).max() < 1e-5  # the start is exact already
    ok, rf, rc = ora.estimate_rotations_rig(s["N"], s["C"], s["imf"], s["imc"], s["ii"], s["jj"], s["q"], np.ones(s["ii"].size),
                                            aa_f, aa_c, 0)
    assert ok
    assert synthetic.rotation_errors_deg(so3.aa_to_rotmat(rf), s["R_f"]).max() < 1e-2  # rotation_averager_test.cc:258-259
    assert so3.rotation_angle_deg(so3.aa_to_rotmat(rc), s["R_c"]).max() < 1e-2


def test_rig_system_without_cam_blocks_is_the_plain_one():
    """Every image without a block, one image per frame: the rig formulation of the oracle reproduces the plain solver."""
    p = synthetic.make_ring_view_graph(40, 6, seed=3)
    opt = ora.RotationEstimatorOptions(skip_initialization=True)
    ok, rot = ora.estimate_rotations(p.num_nodes, p.edge_i, p.edge_j, p.edge_q, p.edge_weight, p.edge_ninl, p.node_aa0, 0, opt)
    ok2, rf, rc = ora.estimate_rotations_rig(p.num_nodes, 0, np.arange(p.num_nodes), -np.ones(p.num_nodes, np.int64), p.edge_i,
                                             p.edge_j, p.edge_q, p.edge_weight, p.node_aa0, np.zeros((0, 3)), 0, opt)
    assert ok and ok2 and np.abs(rot - rf).max() < 1e-9


@pytest.mark.gpu
@pytest.mark.parametrize("frames,cams,noise,outlier", [(14, 3, 0.0, 0.0), (14, 3, 0.5, 0.05), (60, 2, 1.0, 0.1)])
def test_rig_rotation_averaging_matches_oracle(gsfm_ctx, frames, cams, noise, outlier):
    """Fixed iteration counts (the thresholds are set out of reach) so that both sides run the same iterations: the
    quaternion-average cam update makes this iteration converge slowly from an inexact start, which is the reference's
    behaviour and not the point here."""
    s = make_rig_view_graph(frames, cams, seed=2, noise_deg=noise, outlier=outlier)
    aa_f, aa_c = _start(s)
    kw = dict(max_num_l1_iterations=3, max_num_irls_iterations=6, l1_step_convergence_threshold=0.0,
              irls_step_convergence_threshold=0.0, skip_initialization=True)
    tr = ora.RaTrace()
    ok, rf, rc = ora.estimate_rotations_rig(s["N"], s["C"], s["imf"], s["imc"], s["ii"], s["jj"], s["q"], np.ones(s["ii"].size),
                                            aa_f, aa_c, 0, ora.RotationEstimatorOptions(**kw), trace=tr)
    assert ok
    rc_, rot, cam, rep = estimators.ra_solve_rigs(s["N"], s["imf"], s["imc"], s["C"], s["ii"], s["jj"], s["q"], s["ninl"],
                                                   options=estimators.RotationEstimatorOptions(**kw), ctx=gsfm_ctx,
                                                   frame_aa0=aa_f, cam_aa0=aa_c)
    assert rc_ == 0
    if noise == 0.0:  # |last_norm - curr_norm| < EPS ends the L1 stage at once on exact data (gra.cc:529-535)
        assert rep["iterations_l1"] == tr.l1_iterations
    else:
        assert rep["iterations_l1"] == tr.l1_iterations == 3 and rep["iterations_irls"] == tr.irls_iterations == 6
    assert np.radians(so3.rotation_angle_deg(so3.aa_to_rotmat(rot), so3.aa_to_rotmat(rf))).max() < 1e-6
    assert np.radians(so3.rotation_angle_deg(so3.aa_to_rotmat(cam), so3.aa_to_rotmat(rc))).max() < 1e-6
    if noise == 0.0:
        assert synthetic.rotation_errors_deg(so3.aa_to_rotmat(rot), s["R_f"]).max() < 1e-2
        assert so3.rotation_angle_deg(so3.aa_to_rotmat(cam), s["R_c"]).max() < 1e-2


@pytest.mark.gpu
def test_rig_rotation_averaging_initialises_from_the_image_spanning_tree(gsfm_ctx):
    """skip_initialization = false: spanning tree over the images through the C ABI, ConvertRotationsFromImageToRig, solve;
    noise-free data come out exact (the reference's pin, 1e-2 degrees)."""
    s = make_rig_view_graph(20, 3, seed=5, noise_deg=0.0, outlier=0.0)
    rc_, rot, cam, rep = estimators.ra_solve_rigs(s["N"], s["imf"], s["imc"], s["C"], s["ii"], s["jj"], s["q"], s["ninl"],
                                                   ctx=gsfm_ctx)
    assert rc_ == 0
    assert synthetic.rotation_errors_deg(so3.aa_to_rotmat(rot), s["R_f"]).max() < 1e-2
    assert so3.rotation_angle_deg(so3.aa_to_rotmat(cam), s["R_c"]).max() < 1e-2


@pytest.mark.gpu
def test_rig_mode_needs_an_initial_estimate(gsfm_ctx):
    s = make_rig_view_graph(10, 2, seed=1)
    from glomap_amd.flat import RaProblem

    p = RaProblem(s["N"], s["ii"], s["jj"], s["q"], np.ones(s["ii"].size), s["ninl"], np.zeros((s["N"], 3)), 0,
                  image_frame=s["imf"], image_cam=s["imc"], cam_aa0=np.zeros((s["C"], 3)))
    rc_, _, _ = estimators.ra_solve(p, estimators.RotationEstimatorOptions(), ctx=gsfm_ctx)
    assert rc_ != 0  # GSFM_ERR_UNSUPPORTED: the library does not build the image-level start itself


@pytest.mark.gpu
def test_unknown_rig_pipeline_recovers_the_scene(gsfm_ctx):
    """RA -> GP -> BA on a noise-free scene of 2 rigs x 3 cameras x 7 frames whose non-reference cam_from_rig are all
    UNKNOWN — the configuration and the pins (rotations 1e-2 degrees, projection centres 1e-4) of the reference's
    GlobalMapper.WithoutNoiseWithNonTrivialUnknownRig (global_mapper_test.cc:128-175): rotation averaging estimates the
    cam_from_rig rotations, global positioning their translations (RigUnknownBATA), bundle adjustment then runs with
    the estimated rigs held constant (optimize_rig_poses = false, as the mapper leaves it)."""
    from glomap_amd.flat import BaProblem, GpProblem

    gp, ba, info = synthetic.make_rig_problems(14, 3, 600, seed=9)
    N, I = gp.num_cams, gp.num_images
    R_cw = info["R_cw"]
    imf = gp.image_frame.astype(np.int32)
    imc = info["sensor_block"].astype(np.int32)
    C = int(imc.max()) + 1
    ii, jj = np.triu_indices(I, 1)
    d = np.abs(imf[ii].astype(np.int64) - imf[jj])
    near = (np.minimum(d, N - d) <= 3) & ~((imf[ii] == imf[jj]) & (imc[ii] < 0) & (imc[jj] < 0))
    ii, jj = ii[near].astype(np.int32), jj[near].astype(np.int32)
    q_rel = so3.rotmat_to_quat(R_cw[jj] @ np.transpose(R_cw[ii], (0, 2, 1)))
    ninl = np.random.default_rng(0).integers(30, 300, ii.shape[0]).astype(np.int32)
    rc, rot, cam, rep = estimators.ra_solve_rigs(N, imf, imc, C, ii, jj, q_rel, ninl, ctx=gsfm_ctx)
    assert rc == 0
    R_f, R_c = so3.aa_to_rotmat(rot), so3.aa_to_rotmat(cam)
    assert synthetic.rotation_errors_deg(R_f, gp.cam_R).max() < 1e-2
    # global positioning: rays rotated by the ESTIMATED cam_from_world rotations, every cam_from_rig translation unknown
    R_s_est = np.where((imc >= 0)[:, None, None], R_c[np.maximum(imc, 0)], np.eye(3))
    Rcw_est = R_s_est @ R_f[imf]
    ray_cam = np.einsum("mij,mj->mi", R_cw[gp.obs_cam], gp.obs_dir)  # back to camera-frame rays (features_undist)
    gp2 = GpProblem(num_cams=N, num_pts=gp.num_pts, pt_offset=gp.pt_offset, obs_cam=gp.obs_cam,
                    obs_dir=np.ascontiguousarray(np.einsum("mji,mj->mi", Rcw_est[gp.obs_cam], ray_cam)),
                    obs_calibrated=gp.obs_calibrated, cam_center=np.zeros((N, 3)), pt_xyz=np.zeros((gp.num_pts, 3)),
                    image_frame=gp.image_frame, image_offset=np.zeros((I, 3)), image_sensor=imc,
                    image_sensor_rot=np.ascontiguousarray(R_f[imf]), sensor_center=np.zeros((C, 3)))
    rc, cen, xyz, rep = estimators.gp_solve(gp2, ctx=gsfm_ctx)
    assert rc == 0
    t_s_est = -np.einsum("cij,cj->ci", R_c, rep["sensor_center"])  # t = -R c (gp.cc:576-582)
    # bundle adjustment from the positioning result with the estimated rigs as constants
    cfr = np.zeros((I, 7))
    cfr[:, 0] = 1.0
    has = imc >= 0
    cfr[has] = np.concatenate([so3.rotmat_to_quat(R_c[imc[has]]), t_s_est[imc[has]]], axis=1)
    ba2 = BaProblem(num_cams=N, num_pts=ba.num_pts, num_intr=ba.num_intr, pt_offset=ba.pt_offset, obs_cam=ba.obs_cam,
                    obs_xy=ba.obs_xy, cam_intr=ba.cam_intr, cam_q=so3.rotmat_to_quat(R_f), cam_t=-np.einsum("nij,nj->ni", R_f, cen),
                    pt_xyz=xyz, intr_model=ba.intr_model, intr_params=ba.intr_params, fixed_cam=0, image_frame=ba.image_frame,
                    image_cam_from_rig=cfr, image_intr=ba.image_intr)
    rc, q, t, X, intr, rep = estimators.ba_solve(ba2, estimators.BundleAdjusterOptions(optimize_rotations=False), ctx=gsfm_ctx)
    assert rc == 0
    ba2.cam_t, ba2.pt_xyz, ba2.intr_params = t, X, intr
    rc, q, t, X, intr, rep = estimators.ba_solve(ba2, ctx=gsfm_ctx)
    assert rc == 0
    R_fin = so3.quat_to_rotmat(q)
    # image level, like ComputeImageAlignmentError: cam_from_world = cam_from_rig * rig_from_world
    R_img = so3.quat_to_rotmat(cfr[:, :4]) @ R_fin[imf]
    t_img = np.einsum("iab,ib->ia", so3.quat_to_rotmat(cfr[:, :4]), t[imf]) + cfr[:, 4:]
    c_img = -np.einsum("iba,ib->ia", R_img, t_img)
    c_gt = -np.einsum("iba,ib->ia", R_cw, info["t_cw"])
    assert synthetic.rotation_errors_deg(R_img, R_cw).max() < 1e-2
    assert synthetic.center_errors_after_sim3(c_img, c_gt).max() < 1e-4
