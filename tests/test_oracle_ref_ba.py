"""The problem oracle/ba.py poses against the problem the REFERENCE'S OWN bundle adjustment builder poses.

oracle/_ref/libref_glomap_ba.so is glomap/estimators/bundle_adjustment.cc compiled from /root/reference, unmodified
(`make -C oracle ref`; flat entry point oracle/ref_glue_ba.cc), against a RECORDING Ceres (oracle/ref_shim/ceres/ceres.h: it
stores residual blocks, constants, manifolds and the ordering, evaluates the robustified cost at the start point and does not
minimise) and stand-ins for COLMAP's cost-function factory and manifold helpers (oracle/ref_shim_ba/; Ceres and COLMAP are not in
this image).  What is pinned is the reference's builder logic (ba.cc:115-317): which tracks and observations enter, which of
the three reprojection functors an observation gets, which blocks are held constant — the FIRST frame the frames map yields,
rotations / translations / points by option, the intrinsics through the principal-point subset or as a whole, and nothing when
optimize_principal_point is set without optimize_intrinsics —, the manifolds, the elimination ordering, the linear solver it
asks for; and, through the initial cost, that every block is wired to the right parameters.  The trust-region loop stays a
restatement (oracle/lm.py), the projection functions are restated on both sides (un-vendored COLMAP)."""
import numpy as np
import pytest

from glomap_amd import synthetic
from oracle import ba as oba
from oracle import ref

pytestmark = pytest.mark.skipif(ref.load_ba() is None, reason="oracle/_ref: neither /root/reference nor a prebuilt oracle/_ref/libref_glomap_ba.so")

IN, ROT_CONST, TRN_CONST, QUAT, GROUP0, NO_GROUP = 1, 2, 4, 8, 16, 32
SPARSE_SCHUR, CLUSTER_TRIDIAGONAL = 4, 4  # ceres/types.h order, ba.cc:98-99


def _first_in_problem(r):
    order = r["frame_order"]
    return int(order[(r["frame_flags"][order] & IN) != 0][0])


def _check_blocks(r, prob, used, opt, num_frames):
    """Frames, cameras and tracks of the reference's recorded problem against the oracle's free / fixed bookkeeping."""
    ff = r["frame_flags"]
    inp = (ff & IN) != 0
    assert set(np.nonzero(inp)[0]) == set(np.unique(prob.cam))
    assert ((ff[inp] & QUAT) != 0).all() and ((ff[inp] & (GROUP0 | NO_GROUP)) == 0).all()  # quaternion manifold, group 1
    assert np.array_equal((ff[inp] & ROT_CONST) != 0, ~prob.rot_free[:num_frames][inp])
    assert np.array_equal((ff[inp] & TRN_CONST) != 0, ~prob.trn_free[:num_frames][inp])
    # intrinsics: constant block, principal-point subset, or free — the oracle's mask of optimised entries
    for k in range(len(r["camera_flags"])):
        if not r["camera_flags"][k] & IN:
            continue
        n = oba.NUM_PARAMS[int(prob.model[k])]
        free_ref = np.ones(n, bool)
        if r["camera_flags"][k] & ROT_CONST:
            free_ref[:] = False
        free_ref[r["camera_subset"][k, :n] != 0] = False
        assert np.array_equal(free_ref, prob.fmask[k, :n]), (k, free_ref, prob.fmask[k])
        assert (r["camera_flags"][k] & (GROUP0 | NO_GROUP)) == 0
    tf = r["track_flags"]
    assert np.array_equal((tf & IN) != 0, used)  # ba.cc:122
    assert ((tf[used] & GROUP0) != 0).all()       # points are eliminated first (ba.cc:204-208)
    assert (((tf[used] & ROT_CONST) != 0) == (not opt.optimize_points)).all()
    assert (r["linear_solver_type"], r["preconditioner_type"]) == (SPARSE_SCHUR, CLUSTER_TRIDIAGONAL)


OPTION_SETS = [
    dict(),
    dict(optimize_rotations=False),                                   # the mapper's positions-only stage (global_mapper.cc:201-212)
    dict(optimize_translation=False),
    dict(optimize_intrinsics=False),
    dict(optimize_principal_point=True),
    dict(optimize_intrinsics=False, optimize_principal_point=True),   # neither branch of ba.cc:273-293: everything stays free
    dict(optimize_points=False),
    dict(min_num_view_per_track=5),
]


@pytest.mark.parametrize("kw", OPTION_SETS, ids=lambda kw: ",".join(f"{k}={int(v)}" for k, v in kw.items()) or "default")
def test_trivial_frames_problem_equals_the_reference(kw):
    p = synthetic.make_ba_problem(num_cams=12, num_pts=160, seed=3, pixel_noise=0.7, outlier_ratio=0.03, intr_noise=0.01)
    # two camera models: the second half of the cameras becomes RADIAL (f, cx, cy, k1, k2)
    model = p.intr_model.copy()
    params = p.intr_params.copy()
    model[p.num_intr // 2:] = oba.RADIAL
    params[p.num_intr // 2:, 4] = -0.003
    r = ref.ba_build(model, params, p.cam_q, p.cam_t, np.arange(p.num_cams), p.cam_intr, p.pt_offset, p.obs_cam, p.obs_xy, p.pt_xyz,
                     rig_ref_cam=np.arange(p.num_intr), frame_rig=p.cam_intr, **kw)  # one rig per camera, the camera its reference sensor
    assert r["num_residual_blocks"] > 0
    fixed = _first_in_problem(r)
    opt = oba.BundleAdjusterOptions(**kw)
    b = oba.build_problem(p.num_cams, p.pt_offset, p.obs_cam, p.obs_xy, p.cam_intr, model, fixed, p.cam_q, p.cam_t, p.pt_xyz, params, opt)
    prob, used = b["problem"], b["used"]
    assert (r["kind"] == 0).all() and (r["sensor"] == -1).all()  # trivial frames: ReprojErrorCostFunctor with the one Huber loss
    tracks_used = np.nonzero(used)[0]
    ref_pairs = sorted(zip(r["frame"].tolist(), r["track"].tolist(), r["camera"].tolist()))
    ora_pairs = sorted(zip(prob.cam.tolist(), tracks_used[prob.pt].tolist(), prob.cam_intr[prob.cam].tolist()))
    assert ref_pairs == ora_pairs
    _check_blocks(r, prob, used, opt, p.num_cams)
    assert abs(prob.cost(b["x0"]) - r["initial_cost"]) <= 1e-12 * r["initial_cost"]


@pytest.mark.parametrize("kw", [dict(), dict(optimize_intrinsics=False), dict(optimize_principal_point=True)],
                         ids=lambda kw: ",".join(f"{k}={int(v)}" for k, v in kw.items()) or "default")
def test_a_camera_model_with_more_than_eight_parameters(kw):
    """FULL_OPENCV (12 parameters) through BundleAdjuster::Solve as the reference wrote it: the parameter block it adds for such a
    camera (all 12 entries of Camera::params), the SubsetManifold it puts on the principal point (indices 2, 3 of 12,
    bundle_adjustment.cc:273-287) and the initial cost — against oracle/ba.py on [K, 16] intrinsics rows, the layout the 16-wide
    units of libgsfm and of the C++ oracle take.  Half of the cameras stay SIMPLE_RADIAL: both widths in one problem."""
    p = synthetic.make_ba_problem(num_cams=12, num_pts=160, seed=3, pixel_noise=0.7, outlier_ratio=0.03, intr_noise=0.01)
    model = p.intr_model.copy()
    params = np.zeros((p.num_intr, 16))
    params[:, :8] = p.intr_params
    h = p.num_intr // 2
    model[h:] = oba.FULL_OPENCV
    params[h:, :12] = [1200, 1190, 640, 480, 0.02, -0.01, 0.001, -0.002, 0.003, 0.01, -0.004, 0.002]
    r = ref.ba_build(model, params, p.cam_q, p.cam_t, np.arange(p.num_cams), p.cam_intr, p.pt_offset, p.obs_cam, p.obs_xy, p.pt_xyz,
                     rig_ref_cam=np.arange(p.num_intr), frame_rig=p.cam_intr, **kw)
    assert r["num_residual_blocks"] > 0
    fixed = _first_in_problem(r)
    opt = oba.BundleAdjusterOptions(**kw)
    b = oba.build_problem(p.num_cams, p.pt_offset, p.obs_cam, p.obs_xy, p.cam_intr, model, fixed, p.cam_q, p.cam_t, p.pt_xyz, params, opt)
    prob, used = b["problem"], b["used"]
    assert prob.fmask.shape == (p.num_intr, 16)
    _check_blocks(r, prob, used, opt, p.num_cams)
    if not kw:  # the default: 10 of the 12 FULL_OPENCV parameters are optimised, the principal point is held by the subset manifold
        assert np.array_equal(prob.fmask[h:].sum(1), np.full(p.num_intr - h, 10)) and np.array_equal(r["camera_subset"][h:, :12].sum(1),
                                                                                                   np.full(p.num_intr - h, 2))
        assert r["camera_subset"][h:, 2:4].all()
    assert abs(prob.cost(b["x0"]) - r["initial_cost"]) <= 1e-12 * r["initial_cost"]


def test_observations_of_absent_images_and_frames_without_pose():
    """ba.cc:125: an observation whose image is not in the map is skipped (the track stays); a frame that then has no residual
    is not part of the problem and the FIRST frame that is becomes the constant one (ba.cc:252-270)."""
    p = synthetic.make_ba_problem(num_cams=10, num_pts=120, seed=5, pixel_noise=0.5)
    present = np.ones(p.num_cams, np.uint8)
    r0 = ref.ba_build(p.intr_model, p.intr_params, p.cam_q, p.cam_t, np.arange(p.num_cams), p.cam_intr, p.pt_offset, p.obs_cam, p.obs_xy, p.pt_xyz,
                      rig_ref_cam=np.arange(p.num_intr), frame_rig=p.cam_intr)
    first = _first_in_problem(r0)
    present[first] = 0  # drop the image of the frame that was constant
    r = ref.ba_build(p.intr_model, p.intr_params, p.cam_q, p.cam_t, np.arange(p.num_cams), p.cam_intr, p.pt_offset, p.obs_cam, p.obs_xy, p.pt_xyz,
                     rig_ref_cam=np.arange(p.num_intr), frame_rig=p.cam_intr, image_present=present)
    assert not r["frame_flags"][first] & IN
    fixed = _first_in_problem(r)
    assert fixed != first and (r["frame_flags"][fixed] & (ROT_CONST | TRN_CONST)) == (ROT_CONST | TRN_CONST)
    # the oracle's flat form of the same thing: the observations dropped, the track lengths counted BEFORE the drop
    keep = p.obs_cam != first
    lens = np.diff(p.pt_offset)
    long_enough = np.repeat(lens >= 3, lens)
    assert r["num_residual_blocks"] == int((keep & long_enough).sum())


def test_rig_problems_equal_the_reference():
    """Non-trivial frames: RigReprojErrorConstantRigCostFunctor with the calibration held (ba.cc:147-160), or — with
    optimize_rig_poses — RigReprojErrorCostFunctor on the sensors' cam_from_rig blocks, which get their manifold and no constant
    flag (ba.cc:161-179, 296-309)."""
    gp, ba, info = synthetic.make_rig_problems(10, 3, 200, seed=4, pixel_noise=0.5)
    S, R = 3, 2
    I = ba.image_frame.size
    sensor = info["image_sensor"]
    rig_of_frame = info["rig_of_frame"]
    image_camera = ba.image_intr  # one camera per (rig, sensor)
    sensor_rig = np.repeat(np.arange(R), S - 1)
    sensor_cam = np.concatenate([r * S + np.arange(1, S) for r in range(R)])
    for rig_poses in (False, True):
        kw = dict(optimize_rig_poses=rig_poses)
        r = ref.ba_build(ba.intr_model, ba.intr_params, ba.cam_q, ba.cam_t, ba.image_frame, image_camera, ba.pt_offset, ba.obs_cam, ba.obs_xy,
                         ba.pt_xyz, rig_ref_cam=np.arange(R) * S, frame_rig=rig_of_frame, sensor_rig=sensor_rig, sensor_cam=sensor_cam,
                         sensor_pose=info["sensor_cam_from_rig"], **kw)
        fixed = _first_in_problem(r)
        opt = oba.BundleAdjusterOptions(**kw)
        b = oba.build_problem(ba.num_cams, ba.pt_offset, ba.obs_cam, ba.obs_xy, None, ba.intr_model, fixed, ba.cam_q, ba.cam_t, ba.pt_xyz,
                              ba.intr_params, opt, image_frame=ba.image_frame, image_cam_from_rig=ba.image_cam_from_rig,
                              image_intr=ba.image_intr, image_sensor=info["sensor_block"], sensor_cam_from_rig=info["sensor_cam_from_rig"])
        prob, used = b["problem"], b["used"]
        tracks_used = np.nonzero(used)[0]
        # per observation: frame, track, camera, functor
        lens = np.diff(ba.pt_offset)
        obs_pt = np.repeat(np.arange(lens.size), lens)
        keep = used[obs_pt]
        img = ba.obs_cam[keep]
        kind_expected = np.where(sensor[img] == 0, 0, 2 if rig_poses else 1)
        sens_expected = np.where((sensor[img] > 0) & rig_poses, info["sensor_block"][img], -1)
        ora_rows = sorted(zip(prob.cam.tolist(), tracks_used[prob.pt].tolist(), ba.image_intr[img].tolist(), kind_expected.tolist(),
                              sens_expected.tolist()))
        ref_rows = sorted(zip(r["frame"].tolist(), r["track"].tolist(), r["camera"].tolist(), r["kind"].tolist(), r["sensor"].tolist()))
        assert ref_rows == ora_rows
        if rig_poses:
            assert np.array_equal(np.sort(np.unique(prob.obs_sens[prob.obs_sens >= 0])) - ba.num_cams, np.nonzero(r["sensor_flags"] & IN)[0])
            sf = r["sensor_flags"]
            assert ((sf & IN) != 0).all() and ((sf & QUAT) != 0).all() and ((sf & (ROT_CONST | TRN_CONST)) == 0).all()
            assert prob.rot_free[ba.num_cams:].all() and prob.trn_free[ba.num_cams:].all()
        else:
            assert ((r["sensor_flags"] & IN) == 0).all()
        _check_blocks(r, prob, used, opt, ba.num_cams)
        assert abs(prob.cost(b["x0"]) - r["initial_cost"]) <= 1e-12 * r["initial_cost"]


# ---------------------------------------------------------------------------------------------------------------
# round 6: the reference's BundleAdjuster::Solve run to its END POINT (oracle/_ref/libref_glomap_ba_solve.so)
# ---------------------------------------------------------------------------------------------------------------
@pytest.mark.skipif(ref.load_ba_solve() is None, reason="oracle/_ref/libref_glomap_ba_solve.so not built")
@pytest.mark.parametrize("N,P,seed,kw", [(15, 300, 23, dict(shared_intrinsics=True, intr_noise=0.01)),
                                         (40, 1500, 2, dict(pixel_noise=0.7, outlier_ratio=0.02, intr_noise=0.01)),
                                         (60, 3000, 5, dict(pixel_noise=0.5, outlier_ratio=0.02))])
def test_bundle_adjustment_end_point_equals_the_reference(N, P, seed, kw):
    """bundle_adjustment.cc, unmodified, on the SOLVING Ceres stand-in (oracle/ref_shim_solve/ceres/ceres.h; COLMAP's reprojection
    functor restated in ref_shim_ba/ and differentiated by dual numbers; quaternion / subset manifolds, the constant frame the
    reference's hash map yields, points eliminated exactly, dense Cholesky on the rest) against oracle/ba.py + oracle/lm.py
    (analytic Jacobians, left-multiplicative tangent, sparse Schur complement): the same LM iterations and accepted steps, the
    same costs to nine digits, the same poses."""
    from glomap_amd import so3, synthetic
    from oracle import ba as oba

    p = synthetic.make_ba_problem(num_cams=N, num_pts=P, seed=seed, **kw)
    r = ref.ba_solve(p.intr_model, p.intr_params, p.cam_q, p.cam_t, np.arange(p.num_cams), p.cam_intr, p.pt_offset, p.obs_cam, p.obs_xy, p.pt_xyz,
                     rig_ref_cam=np.arange(p.num_intr), frame_rig=p.cam_intr)
    assert r["ok"] and not r["constrained"] and r["frame_const"].sum() == 1
    fixed = int(np.nonzero(r["frame_const"])[0][0])
    assert fixed == int(r["frame_order"][0])  # the first frame the map yields (ba.cc:252-270)
    ok, q, t, X, intr, s = oba.solve(p.num_cams, p.pt_offset, p.obs_cam, p.obs_xy, p.cam_intr, p.intr_model, fixed, p.cam_q, p.cam_t, p.pt_xyz,
                                     p.intr_params)
    assert ok and (s.iterations, s.successful_steps) == (r["iterations"], r["successful_steps"])
    assert abs(s.initial_cost - r["initial_cost"]) <= 1e-12 * r["initial_cost"]
    assert abs(s.final_cost - r["final_cost"]) <= 1e-8 * r["final_cost"]
    ang = np.radians(so3.rotation_angle_deg(so3.quat_to_rotmat(q), so3.quat_to_rotmat(r["frame_q"])))
    assert ang.max() < 1e-6 and np.abs(t - r["frame_t"]).max() < 1e-3 * (1 + np.abs(t).max())
    assert np.abs(intr - r["cam_params"][:, : intr.shape[1]]).max() < 1e-6 * 1200
    assert np.array_equal(q[fixed], p.cam_q[fixed]) and np.allclose(r["frame_q"][fixed], p.cam_q[fixed], atol=1e-15)
