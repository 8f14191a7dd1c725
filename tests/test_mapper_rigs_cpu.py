"""The estimator stage of the mapper on multi-camera rigs at SCENE level — rotation_averager.SolveRotationAveraging ->
mapper_estimators.GlobalPositioner -> mapper_estimators.BundleAdjuster on the containers of glomap_amd.scene — with the
ORACLE as numerical backend (CPU).  Configuration and pins of the reference's two rig mapper tests
(glomap/controllers/global_mapper_test.cc:89-126 known rig, :128-175 unknown rig): noise-free synthetic scene, recovered
rotations within 1e-2 degrees and projection centres within 1e-4 after a similarity alignment.

The same flat calls run on the GPU in tests/test_rigs.py / test_ra_rigs.py (pinned to the same oracle functions)."""
import numpy as np
import pytest

from glomap_amd import estimators, mapper_estimators as mest, rotation_averager as rav, so3, synthetic
from glomap_amd.scene import Camera, Frame, Image, ImagePair, Rig, Rigid3d, Track, ViewGraph
from oracle import ba as oba
from oracle import cpu
from oracle import gp as ogp
from test_rotation_averager_policy import OracleBackend as RaOracleBackend


class OracleBackend(RaOracleBackend):
    """+ the GP / BA flat calls answered by the multithreaded C++ oracle."""

    def gp_solve(self, p, opt):
        oo = ogp.GlobalPositionerOptions(**{k: getattr(opt, k) for k in ogp.GlobalPositionerOptions.__dataclass_fields__
                                            if k != "lm" and hasattr(opt, k)})
        oo.lm.max_num_iterations = opt.solver_options.max_num_iterations
        oo.lm.function_tolerance = opt.solver_options.function_tolerance
        kw = {}
        if p.image_frame is not None:
            kw.update(image_frame=p.image_frame, image_offset=p.image_offset)
        if p.sensor_center is not None:
            kw.update(image_sensor=p.image_sensor, image_sensor_rot=p.image_sensor_rot, sensor_center=p.sensor_center)
        if p.pair_i is not None:  # camera-to-camera constraints: the numpy oracle restates them (the C++ one does not)
            ok, c, X, s = ogp.solve(p.num_cams, p.pt_offset, p.obs_cam, p.obs_dir, p.obs_calibrated, p.cam_center, p.pt_xyz, oo,
                                    pair_i=p.pair_i, pair_j=p.pair_j, pair_dir=p.pair_dir)
        else:
            ok, c, X, s = cpu.gp_solve(p.num_cams, p.pt_offset, p.obs_cam, p.obs_dir, p.obs_calibrated, p.cam_center, p.pt_xyz, oo, **kw)
        rep = {"iterations": s.iterations, "final_cost": s.final_cost}
        if p.sensor_center is not None:
            rep["sensor_center"] = s.sensor_center
        return (0 if ok else -6), c, X, rep

    def ba_solve(self, p, opt):
        oo = oba.BundleAdjusterOptions(**{k: getattr(opt, k) for k in oba.BundleAdjusterOptions.__dataclass_fields__
                                          if k != "lm" and hasattr(opt, k)})
        oo.lm.max_num_iterations = opt.solver_options.max_num_iterations
        oo.lm.function_tolerance = opt.solver_options.function_tolerance
        kw = {}
        if p.image_frame is not None:
            kw.update(image_frame=p.image_frame, image_cam_from_rig=p.image_cam_from_rig, image_intr=p.image_intr)
        if p.sensor_cam_from_rig is not None:
            kw.update(image_sensor=p.image_sensor, sensor_cam_from_rig=p.sensor_cam_from_rig)
        r = cpu.ba_solve(p.num_cams, p.pt_offset, p.obs_cam, p.obs_xy, None if p.image_frame is not None else p.cam_intr,
                         p.intr_model, p.fixed_cam, p.cam_q, p.cam_t, p.pt_xyz, p.intr_params, oo, **kw)
        rep = {"iterations": r[5].iterations, "final_cost": r[5].final_cost, "initial_cost": r[5].initial_cost}
        if p.sensor_cam_from_rig is not None:
            rep["sensor_cam_from_rig"] = getattr(r[5], "sensor_cam_from_rig", np.array(p.sensor_cam_from_rig))
        return (0 if r[0] else -6), r[1], r[2], r[3], r[4], rep


@pytest.fixture
def make_backend():
    return OracleBackend


def make_rig_scene(unknown, frames_n=14, cams=3, pts=600, seed=9):
    """2 rigs x `cams` cameras x 7 frames each (the reference's configuration), as scene containers."""
    gp, ba, info = synthetic.make_rig_problems(frames_n, cams, pts, seed=seed)
    S = cams
    imf = gp.image_frame.astype(np.int64)
    sens = info["image_sensor"]
    rig_of_frame = info["rig_of_frame"]
    R_s, t_s, R_cw, t_cw = info["R_s"], info["t_s"], info["R_cw"], info["t_cw"]
    I = imf.shape[0]
    rigs, frames, images, cameras, tracks = {}, {}, {}, {}, {}
    for r in np.unique(rig_of_frame):
        img0 = {s_: next(i for i in range(I) if rig_of_frame[imf[i]] == r and sens[i] == s_) for s_ in range(S)}
        rigs[int(r) + 1] = Rig(int(r) + 1, 100 + int(r) * S,
                               {100 + int(r) * S + s_: (None if unknown else Rigid3d(so3.rotmat_to_quat(R_s[img0[s_]][None])[0], t_s[img0[s_]].copy()))
                                for s_ in range(1, S)})
    for k in range(ba.num_intr):
        cameras[100 + k] = Camera(100 + k, 2, ba.gt_intr[k, :4].copy())  # SIMPLE_RADIAL f, cx, cy, k
    for f in range(frames_n):
        frames[f] = Frame(f, Rigid3d(), True, int(rig_of_frame[f]) + 1, [])
    feats = {i: ([], []) for i in range(I)}
    obs_pt = np.repeat(np.arange(gp.num_pts), np.diff(gp.pt_offset))
    for p in range(gp.num_pts):
        tracks[p] = Track(p, np.zeros(3), [])
    for k in range(gp.num_obs):
        i = int(gp.obs_cam[k])
        feats[i][0].append(ba.obs_xy[k])
        feats[i][1].append(R_cw[i] @ gp.obs_dir[k])  # back to the camera-frame ray (features_undist)
        tracks[int(obs_pt[k])].observations.append((i, len(feats[i][0]) - 1))
    for i in range(I):
        images[i] = Image(i, 100 + int(ba.image_intr[i]), int(imf[i]), np.array(feats[i][0]).reshape(-1, 2),
                          np.array(feats[i][1]).reshape(-1, 3))
        frames[int(imf[i])].image_ids.append(i)
    vg = ViewGraph()
    rng = np.random.default_rng(0)
    for a in range(I):
        for b in range(a + 1, I):
            d = abs(int(imf[a]) - int(imf[b]))
            if min(d, frames_n - d) > 3:
                continue
            pr = ImagePair(a, b, Rigid3d(so3.rotmat_to_quat((R_cw[b] @ R_cw[a].T)[None])[0], np.zeros(3)))
            pr.num_inliers = int(rng.integers(30, 300))
            vg.image_pairs[(a, b)] = pr
    c_gt = -np.einsum("iba,ib->ia", R_cw, t_cw)
    return vg, rigs, cameras, frames, images, tracks, R_cw, c_gt


def _image_poses(rigs, frames, images):
    R, c = [], []
    for i in sorted(images):
        im = images[i]
        fr = frames[im.frame_id]
        Rw = so3.quat_to_rotmat(np.asarray(fr.rig_from_world.rotation)[None])[0]
        tw = np.asarray(fr.rig_from_world.translation)
        if not rav.has_trivial_frame(im, frames, rigs):
            cfr = rigs[fr.rig_id].MaybeSensorFromRig(im.camera_id)
            Rc = so3.quat_to_rotmat(np.asarray(cfr.rotation)[None])[0]
            Rw, tw = Rc @ Rw, Rc @ tw + np.asarray(cfr.translation)
        R.append(Rw)
        c.append(-Rw.T @ tw)
    return np.array(R), np.array(c)


@pytest.mark.parametrize("unknown,cams,seed", [(False, 2, 7), (True, 3, 9)])
def test_rig_scene_through_ra_gp_ba(unknown, cams, seed, make_backend):
    """Known rigs: 2 rigs x 2 cameras (global_mapper_test.cc:95-97); unknown rigs: 2 rigs x 3 cameras (:134-136).  (The
    calibrated 3-camera scene of seed 9 is not used: positioning with metric rig offsets stops at its function tolerance
    1 % away, and this bare chain — no track filtering, no retriangulation, one BA round — then parks bundle adjustment in
    a local minimum; the oracle and the HIP path agree on that too.)"""
    vg, rigs, cameras, frames, images, tracks, R_cw, c_gt = make_rig_scene(unknown, cams=cams, seed=seed)
    be = make_backend()
    assert rav.SolveRotationAveraging(vg, rigs, frames, images, estimators.RotationEstimatorOptions(), backend=be)
    R_est, _ = _image_poses(rigs, frames, images)
    assert synthetic.rotation_errors_deg(R_est, R_cw).max() < 1e-2
    if unknown:  # rotation averaging leaves "no translation yet" (gra.cc:801-815) for global positioning to fill
        assert all(np.isnan(cfr.translation).all() for rig in rigs.values() for _, cfr in rig.NonRefSensors())
    gp_engine = mest.GlobalPositioner(estimators.GlobalPositionerOptions(), be)
    assert gp_engine.Solve(vg, rigs, cameras, frames, images, tracks)
    assert all(not np.isnan(cfr.translation).any() for rig in rigs.values() for _, cfr in rig.NonRefSensors())
    assert all(tr.is_initialized for tr in tracks.values() if len(tr.observations) >= 3)
    # bundle adjustment like the mapper: positions only first, then everything (global_mapper.cc:201-223)
    assert mest.BundleAdjuster(estimators.BundleAdjusterOptions(optimize_rotations=False), be).Solve(rigs, cameras, frames, images, tracks)
    ba_engine = mest.BundleAdjuster(estimators.BundleAdjusterOptions(), be)
    assert ba_engine.Solve(rigs, cameras, frames, images, tracks)
    if unknown:  # and the refinement of the estimated rigs (optimize_rig_poses), which the calibrated scene does not need
        ba_rig = mest.BundleAdjuster(estimators.BundleAdjusterOptions(optimize_rig_poses=True), be)
        assert ba_rig.Solve(rigs, cameras, frames, images, tracks)
        assert ba_rig.report["final_cost"] < 1e-6
    R_fin, c_fin = _image_poses(rigs, frames, images)
    assert synthetic.rotation_errors_deg(R_fin, R_cw).max() < 1e-2          # global_mapper_test.cc:121-125, 170-174
    assert synthetic.center_errors_after_sim3(c_fin, c_gt).max() < 1e-4
    if not unknown:
        scale, _, _ = synthetic.align_sim3(c_fin, c_gt)
        assert abs(scale - 1.0) < 1e-4  # metric rig baselines fix the scale


def test_bundle_adjuster_refuses_uncalibrated_sensors(make_backend):
    vg, rigs, cameras, frames, images, tracks, _, _ = make_rig_scene(True, pts=100)
    assert not mest.BundleAdjuster(estimators.BundleAdjusterOptions(), make_backend()).Solve(rigs, cameras, frames, images, tracks)
    assert not mest.GlobalPositioner(estimators.GlobalPositionerOptions(), make_backend()).Solve(vg, rigs, cameras, frames, images, tracks)


@pytest.mark.parametrize("ctype", [1, 2, 3])  # ONLY_CAMERAS, POINTS_AND_CAMERAS_BALANCED, POINTS_AND_CAMERAS
def test_global_positioner_constraint_types_on_a_trivial_scene(ctype, make_backend):
    """GlobalPositioner::Solve with camera-to-camera constraints (gp.cc:55-71, 167-210) on scene containers: trivial frames,
    rotations known, every valid image pair a BATA constraint built from cam2_from_cam1; noise-free, so the centres come
    back exactly (same pin as the mapper tests: 1e-4 after a similarity alignment)."""
    vg, rigs, cameras, frames, images, tracks, R_cw, c_gt = make_rig_scene(False, frames_n=12, cams=1, pts=300, seed=5)
    assert all(rav.has_trivial_frame(im, frames, rigs) for im in images.values())
    for i, im in images.items():  # rotations as rotation averaging would leave them
        frames[im.frame_id].rig_from_world = Rigid3d(so3.rotmat_to_quat(R_cw[i][None])[0], np.zeros(3))
    for (a, b), pr in vg.image_pairs.items():  # relative poses of the two-view geometry: unit translations
        t = R_cw[b] @ (c_gt[a] - c_gt[b])
        pr.cam2_from_cam1 = Rigid3d(np.asarray(pr.cam2_from_cam1.rotation), t / np.linalg.norm(t))
    xyz0 = {t: np.array(tr.xyz) for t, tr in tracks.items()}
    opt = estimators.GlobalPositionerOptions(constraint_type=ctype, constraint_reweight_scale=2.0)
    eng = mest.GlobalPositioner(opt, make_backend())
    assert eng.Solve(vg, rigs, cameras, frames, images, tracks)
    assert eng.report["final_cost"] < 1e-10
    _, c_fin = _image_poses(rigs, frames, images)
    assert synthetic.center_errors_after_sim3(c_fin, c_gt).max() < 1e-4
    if ctype == 1:  # the tracks are not part of the problem
        assert all(np.array_equal(tracks[t].xyz, xyz0[t]) for t in tracks)
    # what the reference refuses: no image pairs at all (gp.cc:41-45)
    assert not mest.GlobalPositioner(opt, make_backend()).Solve(ViewGraph(), rigs, cameras, frames, images, tracks)
