"""The whole hot path chained the way `GlobalMapper::Solve` chains it (glomap/controllers/global_mapper.cc:85-333),
every stage through the C ABI on flat arrays:

    KeepLargestConnectedComponents -> RotationEstimator -> RelPoseFilter::FilterRotations -> (again)      :92-110
    TrackEngine::EstablishFullTracks / FindTracksForProblem                                                :127-134
    GlobalPositioner (random init) -> TrackFilter x3 -> NormalizeReconstruction                            :152-186
    [BundleAdjuster positions-only, BundleAdjuster full, Normalize, FilterTracksByReprojection] x 3        :201-275

The controller itself is out of scope (control plane); this driver lives in tests/ only.  Pins: the tolerances of the
reference's own end-to-end tests on synthetic scenes — rotations 1e-2 deg and projection centres 1e-4 noise-free
(global_mapper_test.cc:82-86), 0.1 deg / 0.1 with noise and outliers (:211-215) — with our own scene generator
(colmap::SynthesizeDataset is un-vendored)."""
import numpy as np
import pytest

from glomap_amd import _lib, estimators, processors, so3, synthetic
from glomap_amd.flat import BaProblem, GpProblem, RaProblem
from glomap_amd.tracks import KeepLargestConnectedComponents, MatchGraph, TrackEngine, TrackEstablishmentOptions


def _compact(mask):
    idx = np.full(len(mask), -1, dtype=np.int32)
    idx[mask] = np.arange(int(mask.sum()), dtype=np.int32)
    return idx


def _drop_observations(off, keep, *arrays):
    lens = np.diff(off)
    trk = np.repeat(np.arange(len(lens)), lens)
    new_len = np.bincount(trk[keep], minlength=len(lens))
    off2 = np.zeros(len(lens) + 1, dtype=np.int64)
    off2[1:] = np.cumsum(new_len)
    return (off2, *[a[keep] for a in arrays])


def run_pipeline(s, ctx, log=None, ba_on_device=False, ba_max_reprojection_error=1e-2):
    """Returns (registered image mask, R_est [Nr,3,3], centre_est [Nr,3], stats)."""
    N = s["num_images"]
    stats = {}
    ei, ej = s["pair_image1"], s["pair_image2"]
    ev = np.ones(len(ei), dtype=np.uint8)
    ninl = np.diff(s["pair_offset"]).astype(np.int32)

    # ---- 3. rotation averaging, twice, with the rotation filter in between (global_mapper.cc:92-110)
    reg, ev, nimg = KeepLargestConnectedComponents(N, ei, ej, ev, ctx=ctx)
    assert nimg > 0
    R = None
    for _ in range(2):
        regb = reg.astype(bool)
        node = _compact(regb)
        use = ev.astype(bool)
        p = RaProblem(int(regb.sum()), node[ei[use]], node[ej[use]], s["pair_q"][use], np.ones(int(use.sum())), ninl[use],
                      np.zeros((int(regb.sum()), 3)), 0)
        rc, rot, rep = estimators.ra_solve(p, estimators.RotationEstimatorOptions(), ctx=ctx)
        assert rc == 0
        keep, ninv = processors.RelPoseFilter.FilterRotations(so3.aa_to_quat(rot), p.edge_i, p.edge_j, p.edge_q, 10.0, ctx=ctx)
        stats.setdefault("rotation_filtered", []).append(int(ninv))
        ev2 = ev.copy()
        ev2[np.nonzero(use)[0][keep == 0]] = 0
        reg, ev, nimg = KeepLargestConnectedComponents(N, ei, ej, ev2, ctx=ctx)
        assert nimg > 0
        R = np.zeros((N, 3, 3))
        R[regb] = so3.aa_to_rotmat(rot)
    regb = reg.astype(bool)
    stats["registered"] = int(regb.sum())

    # ---- 4. track establishment and selection (:127-134)
    g = MatchGraph(N, s["feat_offset"], s["feat_xy"], ei, ej, s["pair_offset"], s["match_feat1"], s["match_feat2"], pair_valid=ev)
    eng = TrackEngine(g, TrackEstablishmentOptions(), ctx=ctx)
    full = eng.EstablishFullTracks()
    sel = eng.FindTracksForProblem(reg)
    stats["tracks_full"], stats["tracks_discarded"], stats["tracks_selected"] = full.num_tracks, eng.num_discarded, sel.num_tracks
    node = _compact(regb)
    Nr = int(regb.sum())
    Rr = R[regb]
    off = sel.track_offset.copy()
    ocam = node[sel.obs_image]
    ofeat = s["feat_offset"][sel.obs_image] + sel.obs_feature.astype(np.int64)
    assert (ocam >= 0).all()

    # ---- 5. global positioning (:152-160) and the filters after it (:164-186)
    undist = s["feat_undist"][ofeat]
    obs_dir = np.einsum("mji,mj->mi", Rr[ocam], undist)  # R^T v
    gp = GpProblem(Nr, len(off) - 1, off, ocam, obs_dir, np.ones(len(ocam), np.uint8), np.zeros((Nr, 3)), np.zeros((len(off) - 1, 3)))
    rc, cen, X, rep = estimators.gp_solve(gp, estimators.GlobalPositionerOptions(), ctx=ctx)
    assert rc == 0, rc
    q = so3.rotmat_to_quat(Rr)
    t = -np.einsum("nij,nj->ni", Rr, cen)

    def view():
        return processors.SceneView(Nr, off, ocam, q, t, X, obs_undist=s["feat_undist"][ofeat])

    def apply_obs(keep):
        nonlocal off, ocam, ofeat
        off, ocam, ofeat = _drop_observations(off, np.asarray(keep, bool), ocam, ofeat)

    keep, _ = processors.TrackFilter.FilterTracksByAngle(view(), 1.0, ctx=ctx)
    apply_obs(keep)
    tkeep, _ = processors.TrackFilter.FilterTrackTriangulationAngle(view(), 1.0, ctx=ctx)
    apply_obs(np.repeat(np.asarray(tkeep, bool), np.diff(off)))
    keep, _ = processors.TrackFilter.FilterTracksByReprojection(view(), 10 * 1e-2, True, ctx=ctx)
    apply_obs(keep)
    t, X, _ = processors.NormalizeReconstruction(q, t, X, ctx=ctx)

    # ---- 6. bundle adjustment, staged, three rounds (:201-275)
    q, t, X, intr, off, ocam, ofeat, ba_stats = ba_outer_loop(ctx, Nr, off, ocam, ofeat, q, t, X, s["intr_params"].copy(), s["cam_intr"][regb],
                                                              s["intr_model"], s["feat_xy"], s["feat_undist"], device=ba_on_device,
                                                              max_reprojection_error=ba_max_reprojection_error)
    stats.update(ba_stats)
    stats["observations"] = int(len(ocam))
    Rf = so3.quat_to_rotmat(q)
    return regb, Rf, -np.einsum("nji,nj->ni", Rf, t), stats


def ba_outer_loop(ctx, Nr, off, ocam, ofeat, q, t, X, intr, ci, intr_model, feat_xy, feat_undist, device, max_reprojection_error=1e-2):
    """[BundleAdjuster positions-only, BundleAdjuster full, NormalizeReconstruction, FilterTracksByReprojection] x 3
    (global_mapper.cc:201-275).

    device = False: every call takes host arrays (the library stages them, solves, copies back) and the dropped observations
    are removed with numpy — the per-call pack / unpack pattern of the C++ adapter.
    device = True : the state is uploaded ONCE; bundle adjustment, normaliser and filter run on DeviceArrays, the filter's keep
    mask stays in HBM and gsfm_tracks_compact removes the dropped observations there (processors.CompactObservations); one
    download at the end.  Same kernels on the same numbers in the same order: the two must agree bit for bit."""
    stats = {}
    xy = np.ascontiguousarray(feat_xy[ofeat])
    und = np.ascontiguousarray(feat_undist[ofeat])
    ofeat = np.ascontiguousarray(ofeat, dtype=np.int64)
    if device:
        up = lambda a: _lib.DeviceArray.from_numpy(ctx, np.ascontiguousarray(a))  # noqa: E731
        off, ocam, xy, und, ofeat_d = up(off.astype(np.int64)), up(ocam.astype(np.int32)), up(xy), up(und), up(ofeat)
        q, t, X, intr_d, ci_d, model_d = up(q), up(t), up(X), up(intr), up(ci.astype(np.int32)), up(intr_model.astype(np.int32))
    else:
        ofeat_d, intr_d, ci_d, model_d = ofeat, intr, ci, intr_model
    P = int(off.shape[0]) - 1
    rep = None
    for ite in range(3):
        for optimize_rotations in (False, True):
            ba = BaProblem(num_cams=Nr, num_pts=P, num_intr=int(intr_d.shape[0]), pt_offset=off, obs_cam=ocam, obs_xy=xy, cam_intr=ci_d,
                           cam_q=q, cam_t=t, pt_xyz=X, intr_model=model_d, intr_params=intr_d, fixed_cam=0)
            rc, q, t, X, intr_d, rep = estimators.ba_solve(ba, estimators.BundleAdjusterOptions(optimize_rotations=optimize_rotations), ctx=ctx)
            assert rc == 0, rc
        t, X, _ = processors.NormalizeReconstruction(q, t, X, ctx=ctx)
        view = processors.SceneView(Nr, off, ocam, q, t, X, obs_undist=und)
        keep, changed = processors.TrackFilter.FilterTracksByReprojection(view, max(3 - ite, 1) * max_reprojection_error, True, ctx=ctx)
        if device:
            assert isinstance(keep, _lib.DeviceArray)  # the mask never leaves the device
            n, (ocam, xy, und, ofeat_d) = processors.CompactObservations(off, [ocam, xy, und, ofeat_d], obs_keep=keep, ctx=ctx)
        else:
            off, ocam, xy, und, ofeat_d = _drop_observations(off, np.asarray(keep, bool), ocam, xy, und, ofeat_d)
        stats.setdefault("ba_filtered_tracks", []).append(int(changed))
    stats["final_cost"] = rep["final_cost"]
    if device:
        q, t, X, intr_d, off, ocam, ofeat_d = (a.numpy() for a in (q, t, X, intr_d, off, ocam, ofeat_d))
    return q, t, X, intr_d, off, ocam, ofeat_d, stats


@pytest.mark.gpu
def test_pipeline_without_noise(gsfm_ctx):
    """global_mapper_test.cc:56-86 (2 x 7 images, 50 points, no noise): 1e-2 deg, 1e-4 centre error."""
    s = synthetic.make_pipeline_scene(14, 50, seed=0)
    regb, R, C, stats = run_pipeline(s, gsfm_ctx)
    assert regb.all() and stats["tracks_selected"] == 50 and stats["tracks_discarded"] == 0
    rot_err = synthetic.rotation_errors_deg(R, s["gt_R"])
    cen_err = synthetic.center_errors_after_sim3(C, s["gt_center"])
    print(stats, rot_err.max(), cen_err.max())
    assert rot_err.max() < 1e-2
    assert cen_err.max() < 1e-4
    assert stats["observations"] == 14 * 50  # num_obs_tolerance = 0


@pytest.mark.gpu
def test_pipeline_with_noise_outliers_and_a_stray_component(gsfm_ctx):
    """global_mapper_test.cc:176-215 in spirit: pixel noise, wrong relative rotations, false matches, plus two images
    that only see each other.  0.1 deg / 0.1 centre error (relative to the ring's extent here)."""
    s = synthetic.make_pipeline_scene(40, 1500, seed=1, pixel_noise=0.5, rot_outlier_pairs=6, false_match_frac=2e-4,
                                      isolated_pair=True)
    regb, R, C, stats = run_pipeline(s, gsfm_ctx)
    print(stats)
    assert regb[:40].all() and not regb[40:].any()  # the stray pair is gone
    assert stats["rotation_filtered"][0] == 6  # exactly the corrupted pairs
    assert stats["tracks_discarded"] > 0 and stats["tracks_selected"] > 1000
    rot_err = synthetic.rotation_errors_deg(R, s["gt_R"][:40])
    cen_err = synthetic.center_errors_after_sim3(C, s["gt_center"][:40])
    print(rot_err.max(), cen_err.max())
    assert rot_err.max() < 0.1
    assert cen_err.max() < 1e-2


@pytest.mark.gpu
def test_ba_outer_loop_device_resident(gsfm_ctx):
    """The BA outer loop of GlobalMapper::Solve (global_mapper.cc:201-275: solve, solve, normalise, filter, three rounds) with
    the whole state resident in HBM — one upload, DeviceArray problems, the filter's keep mask consumed on the device by
    gsfm_tracks_compact, one download — against the same loop through host arrays: identical poses, points, intrinsics and
    surviving observations, bit for bit ("keeping poses/points device-resident across the BA outer loop")."""
    s = synthetic.make_pipeline_scene(40, 1500, seed=1, pixel_noise=0.5, rot_outlier_pairs=6, false_match_frac=2e-4, isolated_pair=True)
    # a threshold of 1.5 sigma of the pixel noise (0.5 px / f = 1200) in the last round: every round of the loop drops
    # observations, so the compaction has work to do (the reference's 1e-2 drops nothing on this scene)
    kw = dict(ba_max_reprojection_error=6e-4)
    regb_h, R_h, C_h, st_h = run_pipeline(s, gsfm_ctx, ba_on_device=False, **kw)
    regb_d, R_d, C_d, st_d = run_pipeline(s, gsfm_ctx, ba_on_device=True, **kw)
    assert np.array_equal(regb_h, regb_d)
    print(st_h["ba_filtered_tracks"], st_h["observations"])
    assert st_h["ba_filtered_tracks"] == st_d["ba_filtered_tracks"] and min(st_h["ba_filtered_tracks"]) > 0  # every round dropped something
    assert st_h["observations"] == st_d["observations"] and st_h["final_cost"] == st_d["final_cost"]
    assert np.array_equal(R_h, R_d) and np.array_equal(C_h, C_d)
