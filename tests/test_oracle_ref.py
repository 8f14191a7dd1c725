"""oracle/tracks.py against the REFERENCE'S OWN code.

oracle/_ref/libref_glomap.so is glomap/scene/view_graph.cc and glomap/controllers/track_establishment.cc compiled from
/root/reference, unmodified, against the stand-in scene types of oracle/ref_shim/ (recipe: `make -C oracle ref`; flat entry
points: oracle/ref_glue.cc).  These two translation units are pure container / integer logic, which is why they — unlike the
Ceres-backed estimators — can be built in this image.  What is pinned here:

  ViewGraph::KeepLargestConnectedComponents     view_graph.cc:56-97       <->  oracle.tracks.keep_largest_connected_component[_literal]
  TrackEngine::EstablishFullTracks              track_establishment.cc:5-152  <->  oracle.tracks.establish_full_tracks[_literal]
  TrackEngine::FindTracksForProblem             track_establishment.cc:154-227 <->  oracle.tracks.find_tracks_for_problem[_literal]

csrc/tracks.hip is compared bit for bit with oracle/tracks.py on the GPU (tests/test_tracks.py, test_fullsize_gpu.py), so this
chain ends at reference code.  Third-party boundary that remains: colmap::UnionFind (un-vendored) is restated in
oracle/ref_shim/colmap/math/union_find.h; it decides which member names a track (the reference's track ids), not the tracks."""
import numpy as np
import pytest

from glomap_amd import synthetic
from oracle import ref, tracks as ot

pytestmark = pytest.mark.skipif(ref.load() is None, reason="oracle/_ref: neither /root/reference nor a prebuilt oracle/_ref/libref_glomap.so")


def _random_view_graph(rng, num_frames, images_per_frame, comps, p_invalid):
    """Frames split into components of the given sizes (distinct: no tie for the largest), random spanning edges + extras
    inside each component, a share of extra pairs already invalid; images of one frame share it (rigs)."""
    assert sum(comps) <= num_frames and len(set(comps)) == len(comps)
    frame_of_image = np.repeat(np.arange(num_frames), images_per_frame).astype(np.int32)
    images_of = [np.nonzero(frame_of_image == f)[0] for f in range(num_frames)]
    perm = rng.permutation(num_frames)
    e1, e2, valid = [], [], []
    start = 0
    for size in comps:
        members = perm[start : start + size]
        start += size
        for k in range(1, size):  # a random spanning tree of the component
            a, b = members[k], members[rng.integers(0, k)]
            e1.append(rng.choice(images_of[a]))
            e2.append(rng.choice(images_of[b]))
            valid.append(1)
        for _ in range(2 * size):  # extra pairs inside the component, some invalid
            a, b = rng.choice(members, 2, replace=True)
            if a == b:
                continue
            e1.append(rng.choice(images_of[a]))
            e2.append(rng.choice(images_of[b]))
            valid.append(0 if rng.random() < p_invalid else 1)
    # invalid pairs BETWEEN components must not connect anything
    for _ in range(10):
        a, b = rng.integers(0, num_frames, 2)
        e1.append(rng.choice(images_of[a]))
        e2.append(rng.choice(images_of[b]))
        valid.append(0)
    order = rng.permutation(len(e1))
    return (frame_of_image, np.array(e1, np.int32)[order], np.array(e2, np.int32)[order], np.array(valid, np.uint8)[order])


@pytest.mark.parametrize("seed", range(6))
def test_keep_largest_connected_components_equals_the_reference(seed):
    rng = np.random.default_rng(seed)
    num_frames = 60
    images_per_frame = 1 if seed % 2 == 0 else int(rng.integers(2, 4))
    comps = [27, 14, 9, 5, 1][: int(rng.integers(2, 6))]
    imf, p1, p2, pv = _random_view_graph(rng, num_frames, images_per_frame, comps, p_invalid=0.3)
    reg_r, pv_r, n_r = ref.keep_largest_connected_components(len(imf), imf, num_frames, p1, p2, pv)
    per_frame = np.bincount(imf, minlength=num_frames)
    for fn in (ot.keep_largest_connected_component_literal, ot.keep_largest_connected_component):
        reg_o, pv_o, n_o = fn(num_frames, imf[p1], imf[p2], pv, node_num_images=per_frame)
        assert n_o == n_r == comps[0] * images_per_frame
        assert np.array_equal(reg_o, reg_r) and np.array_equal(pv_o, pv_r)
    # a pair that was valid and joins two registered frames stays valid, everything else is invalid now (view_graph.cc:86-91)
    assert np.array_equal(pv_r, pv.astype(bool) & reg_r[imf[p1]] & reg_r[imf[p2]])


def test_keep_largest_connected_components_without_a_valid_pair():
    imf = np.arange(5, dtype=np.int32)
    p1, p2, pv = np.array([0, 1], np.int32), np.array([1, 2], np.int32), np.zeros(2, np.uint8)
    reg_r, pv_r, n_r = ref.keep_largest_connected_components(5, imf, 5, p1, p2, pv)
    reg_o, pv_o, n_o = ot.keep_largest_connected_component_literal(5, p1, p2, pv)
    assert n_r == n_o == 0 and reg_o is None and not pv_r.any()  # the reference returns before it touches a flag (:70)


def _args(g):
    return (g["pair_image1"], g["pair_image2"], g["pair_valid"], g["pair_offset"], g["match_feat1"], g["match_feat2"],
            g["feat_offset"], g["feat_xy"])


def _as_sets(tracks):
    return {frozenset(o) for o in tracks.values() if o}


@pytest.mark.parametrize("seed", range(4))
def test_establish_full_tracks_equals_the_reference(seed):
    """Same partition into tracks, same observations per kept track, same number of tracks discarded by the same-image
    consistency test — against the literal restatement AND the vectorised oracle the GPU path is pinned to."""
    g = synthetic.make_match_graph(60, 700, seed=seed, false_match_frac=0.02, twin_frac=0.02)
    tr_r, disc_r = ref.establish_full_tracks(*_args(g))
    tr_l, disc_l, members = ot.establish_full_tracks_literal(*_args(g))
    assert disc_r == disc_l > 0 and len(tr_r) == len(tr_l)
    assert _as_sets(tr_r) == _as_sets(tr_l)
    # the reference's track ids: union-find roots — the literal restatement reproduces them for the same pair order only
    # when it walks the pairs like the reference's unordered_map does; what must agree is the canonical form
    tid, off, img, ft, disc_v = ot.establish_full_tracks(*_args(g))
    assert disc_v == disc_r and len(tid) == len(tr_r)
    kept = {frozenset(zip(img[off[t] : off[t + 1]].tolist(), ft[off[t] : off[t + 1]].tolist())) for t in range(len(tid)) if off[t + 1] > off[t]}
    assert kept == _as_sets(tr_r)
    # canonical id of a kept track = its smallest member (image << 32 | feature): DESIGN.md 4.7 / INTEGRATION.md
    by_min = {min((i << 32) | f for i, f in o): o for o in tr_r.values() if o}
    for t in range(len(tid)):
        if off[t + 1] > off[t]:
            assert int(tid[t]) in by_min


@pytest.mark.parametrize("kw", [dict(), dict(min_num_tracks_per_view=5), dict(min_num_tracks_per_view=0),
                                dict(min_num_tracks_per_view=20, max_num_tracks=30), dict(max_num_tracks=10), dict(max_num_tracks=0),
                                dict(min_num_view_per_track=2, max_num_view_per_track=6), dict(min_num_tracks_per_view=3, min_num_view_per_track=4)])
def test_find_tracks_for_problem_equals_the_reference(kw):
    """The greedy selection on the oracle's canonical full tracks (ids = smallest member): the reference's own loop must pick
    the same tracks and keep the same observations as both oracle forms, for every option combination of tests/test_tracks.py."""
    g = synthetic.make_match_graph(50, 600, seed=3)
    tid, off, img, ft, _ = ot.establish_full_tracks(*_args(g))
    reg = np.ones(50, np.uint8)
    reg[::7] = 0
    sel_r, kept_r, n_r = ref.find_tracks_for_problem(50, reg, tid, off, img, ft, **kw)
    for fn in (ot.find_tracks_for_problem_literal, ot.find_tracks_for_problem):
        s_tid, s_off, s_img, s_ft = fn(tid, off, img, ft, reg, **kw)
        assert len(s_tid) == n_r == int(sel_r.sum())
        assert set(s_tid.tolist()) == set(tid[sel_r].tolist())
        # observations of the selected tracks: the registered-image subsequence of the full track
        want = {}
        for t in np.nonzero(sel_r)[0]:
            k = np.arange(off[t], off[t + 1])[kept_r[off[t] : off[t + 1]]]
            want[int(tid[t])] = list(zip(img[k].tolist(), ft[k].tolist()))
        got = {int(s_tid[j]): list(zip(s_img[s_off[j] : s_off[j + 1]].tolist(), s_ft[s_off[j] : s_off[j + 1]].tolist())) for j in range(len(s_tid))}
        assert got == want


# ---------------------------------------------------------------------------------------------------------------
# the processors between the solves: track_filter.cc, reconstruction_normalizer.cc
# ---------------------------------------------------------------------------------------------------------------
def _posed_scene(seed, ncam=40, npts=3000, noise=2e-3, outliers=0.05):
    from glomap_amd import so3

    p = synthetic.make_gp_problem(ncam, npts, seed=seed, dir_noise=noise, outlier_ratio=outliers)
    q = so3.rotmat_to_quat(p.cam_R)
    t = -np.einsum("nij,nj->ni", p.cam_R, p.gt_center)
    undist = np.einsum("mij,mj->mi", p.cam_R[p.obs_cam], p.obs_dir)  # world rays -> camera rays
    rng = np.random.default_rng(seed)
    X = p.gt_xyz + rng.normal(0, 0.02, p.gt_xyz.shape)
    X[rng.random(npts) < 0.02] *= 400.0  # a few far points: no triangulation angle left
    X[rng.random(npts) < 0.01] *= -1.0   # and a few behind most of their cameras
    return p, q, t, undist, X


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_track_filters_equal_the_reference(seed):
    """Keep masks and counters of the three filters, bit for bit, against the reference's own track_filter.cc (compiled against
    the stand-in vector types of oracle/ref_shim: every dot product / norm it takes is evaluated in plain doubles in the order
    its source writes them).  csrc/filters.hip is bit-exact against oracle/filters.py on the GPU (tests/test_filters.py)."""
    from oracle import filters as of

    p, q, t, undist, X = _posed_scene(seed)
    cal = (np.random.default_rng(seed).random(p.num_cams) > 0.3).astype(np.uint8)
    for thr in (1e-2, 2e-3):
        k_r, c_r = ref.filter_tracks(0, q, t, p.pt_offset, p.obs_cam, undist, X, thr)
        k_o, c_o = of.filter_tracks_by_reprojection(p.pt_offset, p.obs_cam, q, t, X, thr, True, obs_undist=undist)
        assert np.array_equal(k_r, k_o) and c_r == c_o and 0 < c_r < p.num_pts
    for ang in (1.0, 0.2):
        k_r, c_r = ref.filter_tracks(1, q, t, p.pt_offset, p.obs_cam, undist, X, ang, cam_calibrated=cal)
        k_o, c_o = of.filter_tracks_by_angle(p.pt_offset, p.obs_cam, q, t, X, undist, ang, cam_calibrated=cal)
        assert np.array_equal(k_r, k_o) and c_r == c_o and 0 < c_r < p.num_pts
    for ang in (1.0, 5.0):
        k_r, c_r = ref.filter_tracks(2, q, t, p.pt_offset, p.obs_cam, None, X, ang)
        for fn in (of.filter_tracks_triangulation_angle, of.filter_tracks_triangulation_angle_grouped):
            tk_o, c_o = fn(p.pt_offset, p.obs_cam, q, t, X, ang)
            # the reference clears the observation list of a removed track; an EMPTY track has no pair either: removed as well
            assert c_r == c_o > 0
            assert np.array_equal(k_r, np.repeat(tk_o, np.diff(p.pt_offset)))


@pytest.mark.parametrize("seed", [0, 1, 2])
@pytest.mark.parametrize("kw", [dict(), dict(fixed_scale=True), dict(extent=3.0, p0=0.2, p1=0.7)])
def test_normalize_reconstruction_equals_the_reference(seed, kw):
    """The robust bounding box of reconstruction_normalizer.cc — float copies of the registered centres, one sort per axis, the
    p0 / p1 order statistics, the mean accumulated over the sorted floats in order — and the similarity built from it."""
    from oracle import filters as of

    p, q, t, undist, X = _posed_scene(seed, ncam=37 + seed)
    reg = (np.random.default_rng(10 + seed).random(p.num_cams) > 0.2).astype(np.uint8)
    t_r, X_r, (s_r, tr_r) = ref.normalize_reconstruction(q, t, X, cam_registered=reg, **kw)
    t_o, X_o, (s_o, tr_o) = of.normalize_reconstruction(q, t, X, cam_registered=reg, **kw)
    assert s_r == s_o and np.array_equal(tr_r, tr_o)  # the selection logic: bit for bit
    assert np.abs(t_r - t_o).max() <= 1e-12 * (1 + np.abs(t_o).max()) and np.abs(X_r - X_o).max() <= 1e-12 * (1 + np.abs(X_o).max())
    # three cameras or fewer: the whole range (reconstruction_normalizer.cc:35-38)
    t_r, X_r, (s_r, tr_r) = ref.normalize_reconstruction(q[:3], t[:3], X[:10], **kw)
    t_o, X_o, (s_o, tr_o) = of.normalize_reconstruction(q[:3], t[:3], X[:10], **kw)
    assert s_r == s_o and np.array_equal(tr_r, tr_o)


@pytest.mark.parametrize("seed", [0, 1])
@pytest.mark.parametrize("max_angle", [5.0, 10.0, 1.5])
def test_filter_rotations_equals_the_reference(seed, max_angle):
    """RelPoseFilter::FilterRotations, relpose_filter.cc:7-33: the reference measures the angle as Eigen's
    Quaternion::angularDistance (2 atan2(|vec|, |w|)), the oracle through the trace of the rotation matrix — two formulas for
    the same angle, so the keep masks are compared (identical on these graphs) and the angles are not."""
    from glomap_amd import so3
    from oracle import filters as of

    g = synthetic.make_ring_view_graph(300, 10, seed=seed, noise_deg=2.0, outlier_ratio=0.1)
    nq = so3.rotmat_to_quat(g.gt_R)
    ev_r, n_r = ref.filter_rotations(nq, g.edge_i, g.edge_j, g.edge_q, max_angle)
    keep_o, n_o = of.filter_rotations(nq, g.edge_i, g.edge_j, g.edge_q, max_angle)
    assert n_r == n_o > 0 and np.array_equal(ev_r, keep_o)
    # pairs with an unregistered image and pairs that are already invalid are left alone (:13-21)
    reg = np.ones(300, np.uint8)
    reg[::9] = 0
    ev0 = (np.random.default_rng(seed).random(len(g.edge_i)) > 0.2).astype(np.uint8)
    ev_r, n_r = ref.filter_rotations(nq, g.edge_i, g.edge_j, g.edge_q, max_angle, node_registered=reg, edge_valid=ev0)
    touch = ev0.astype(bool) & reg[g.edge_i].astype(bool) & reg[g.edge_j].astype(bool)
    assert np.array_equal(ev_r, np.where(touch, keep_o, ev0.astype(bool)))


# ---------------------------------------------------------------------------------------------------------------
# the problem global positioning poses: the reference's own builder (global_positioning.cc + cost_function.h on a recording
# Ceres) against oracle/gp.py
# ---------------------------------------------------------------------------------------------------------------
needs_gp = pytest.mark.skipif(ref.load_gp() is None, reason="oracle/_ref/libref_glomap_gp.so not available")


def _renumbered(p, frame_order, track_order):
    """The flat problem with cameras and tracks numbered in the order the REFERENCE walks its unordered_maps — the oracle (like
    the C ABI without *_draw_order) draws its random start in index order, so this makes the two draw orders coincide."""
    from glomap_amd.flat import GpProblem

    new_cam = np.empty(p.num_cams, dtype=np.int64)
    new_cam[frame_order] = np.arange(p.num_cams)
    lens = np.diff(p.pt_offset)[track_order]
    off = np.zeros(p.num_pts + 1, dtype=np.int64)
    off[1:] = np.cumsum(lens)
    idx = np.concatenate([np.arange(p.pt_offset[t], p.pt_offset[t + 1]) for t in track_order]) if p.num_obs else np.zeros(0, np.int64)
    return GpProblem(num_cams=p.num_cams, num_pts=p.num_pts, pt_offset=off, obs_cam=new_cam[p.obs_cam[idx]].astype(np.int32),
                     obs_dir=np.ascontiguousarray(p.obs_dir[idx]), obs_calibrated=p.obs_calibrated[idx], cam_center=p.cam_center[frame_order],
                     pt_xyz=p.pt_xyz[track_order]), new_cam, idx


def _gp_scene(seed, uncal=0.3):
    from glomap_amd import so3

    p = synthetic.make_gp_problem(num_cams=14, num_pts=90, seed=seed, uncalibrated_ratio=uncal, dir_noise=2e-3, outlier_ratio=0.05)
    # tracks below min_num_view_per_track and a camera nobody constrains are part of the walk but not of the problem
    keep = np.ones(p.num_obs, bool)
    lens = np.diff(p.pt_offset)
    for t in (3, 17, 40):
        keep[p.pt_offset[t] + 2 : p.pt_offset[t + 1]] = False  # cut to two views
    keep[p.obs_cam == 5] = False  # camera 5 sees nothing
    trk = np.repeat(np.arange(p.num_pts), lens)
    off = np.zeros(p.num_pts + 1, dtype=np.int64)
    off[1:] = np.cumsum(np.bincount(trk[keep], minlength=p.num_pts))
    p.pt_offset, p.obs_cam, p.obs_dir, p.obs_calibrated = off, p.obs_cam[keep], np.ascontiguousarray(p.obs_dir[keep]), p.obs_calibrated[keep]
    q = so3.rotmat_to_quat(p.cam_R)
    t = -np.einsum("nij,nj->ni", p.cam_R, p.gt_center)
    und = np.einsum("mij,mj->mi", p.cam_R[p.obs_cam], p.obs_dir)
    cal = np.ones(p.num_cams, np.uint8)
    cal[p.obs_cam] = p.obs_calibrated
    p.cam_center = p.gt_center.copy()  # CenterFromPose of the input poses (global_positioning.cc:149,161)
    p.pt_xyz = np.random.default_rng(seed).normal(size=(p.num_pts, 3))
    return p, q, t, und, cal


@needs_gp
@pytest.mark.parametrize("seed", [0, 1, 2])
def test_global_positioning_problem_equals_the_reference(seed):
    """GlobalPositioner::Solve of the reference, run as written up to (and through) ceres::Solve on a Ceres that records and
    does not minimise, against oracle/gp.py stopped before its first iteration: the SAME random start bit for bit (which frames
    and tracks are drawn, in which order, 100 x U(-1, 1) from std::mt19937 seeded with options.seed), the same residual blocks
    (one per observation of a track with >= min_num_view_per_track views), the same observed directions R^T v, Huber(0.1) bare
    for calibrated cameras and scaled by 0.5 for uncalibrated ones, every scale bounded below by 1e-5, the FIRST scale constant,
    and the same initial cost."""
    from oracle import gp as ogp

    p, q, t, und, cal = _gp_scene(seed)
    r = ref.gp_build(q, t, p.pt_offset, p.obs_cam, und, p.pt_xyz, cam_calibrated=cal)
    lens = np.diff(p.pt_offset)
    used = lens >= 3
    assert r["num_residual_blocks"] == int(lens[used].sum())
    # --- the oracle on the same problem, numbered in the reference's walk order, zero LM iterations
    pr, new_cam, idx = _renumbered(p, r["frame_order"], r["track_order"])
    # oracle/_ref is built with g++, which evaluates the three draws of RandVector3d(...) right to left: the first draw of a
    # start vector is its z (rand_vector_order = 1; a clang-built reference gives 0 — the language leaves it open).  This test
    # is what found that: with the oracle's historical x-first order every start vector came out with x and z swapped.
    opt = ogp.GlobalPositionerOptions(rand_vector_order=1)
    opt.lm.max_num_iterations = 0
    ok, c0, X0, summ = ogp.solve(pr.num_cams, pr.pt_offset, pr.obs_cam, pr.obs_dir, pr.obs_calibrated, pr.cam_center, pr.pt_xyz, opt)
    assert summ.iterations == 0
    ok, c0_x, _, _ = ogp.solve(pr.num_cams, pr.pt_offset, pr.obs_cam, pr.obs_dir, pr.obs_calibrated, pr.cam_center, pr.pt_xyz,
                               ogp.GlobalPositionerOptions(rand_vector_order=0, lm=opt.lm))
    drawn = np.arange(p.num_cams) != new_cam[5]
    assert np.array_equal(c0_x[drawn], c0[drawn][:, ::-1])  # the same draws, the other argument order
    # start point: bit for bit (camera 5 is unconstrained: it keeps its centre and takes no draw)
    ref_c0 = r["center_start"][r["frame_order"]]
    assert np.array_equal(c0[drawn], ref_c0[drawn])
    # the unconstrained camera keeps CenterFromPose of its input pose (global_positioning.cc:161): -R^T t, equal to rounding
    assert np.array_equal(c0[new_cam[5]], p.gt_center[5]) and np.abs(ref_c0[new_cam[5]] - p.gt_center[5]).max() < 1e-12
    used_r = used[r["track_order"]]
    assert np.array_equal(X0[used_r], r["xyz_start"][r["track_order"]][used_r])
    assert np.array_equal(r["xyz_start"][~used], p.pt_xyz[~used])  # short tracks keep their input
    assert np.abs(r["center_start"][np.arange(p.num_cams) != 5]).max() <= 100.0 and np.abs(r["center_start"]).max() > 50.0
    # initial cost
    assert abs(summ.initial_cost - r["initial_cost"]) <= 1e-12 * r["initial_cost"]
    # residual blocks, in the order the reference added them = track walk order, observation order inside a track
    want_pt = np.repeat(r["track_order"][used_r], lens[r["track_order"]][used_r])
    assert np.array_equal(r["pt"], want_pt) and (r["cam2"] == -1).all()
    obs_of_block = np.concatenate([np.arange(p.pt_offset[tt], p.pt_offset[tt + 1]) for tt in r["track_order"][used_r]])
    assert np.array_equal(r["cam"], p.obs_cam[obs_of_block])
    assert np.abs(r["dir"] - p.obs_dir[obs_of_block]).max() < 1e-15  # R^T feature_undist (global_positioning.cc:292-294)
    assert np.array_equal(r["loss_scale"], np.where(p.obs_calibrated[obs_of_block] != 0, 1.0, 0.5))  # :244-246, 312-315
    assert (r["lower"] == 1e-5).all() and (r["scale"] == 1.0).all()  # :204 / :376, generate_scales
    assert r["scale_const"][0] == 1 and not r["scale_const"][1:].any()  # :484-489
    # ConvertResults without a minimisation in between: t = -R c (global_positioning.cc:557-559)
    from glomap_amd import so3

    R = so3.quat_to_rotmat(q)
    assert np.abs(r["cam_t_after"] + np.einsum("nij,nj->ni", R, r["center_start"])).max() < 1e-12


@needs_gp
@pytest.mark.parametrize("ctype", [1, 2, 3])
def test_global_positioning_pair_constraints_equal_the_reference(ctype):
    """ONLY_CAMERAS / POINTS_AND_CAMERAS_BALANCED / POINTS_AND_CAMERAS: camera-to-camera blocks first (one per valid pair, the
    first PAIR's scale constant), the point blocks re-weighted by constraint_reweight_scale * #pairs / #TRACKS in the balanced
    mode (global_positioning.cc:230-252), the same initial cost as oracle/gp.py."""
    from glomap_amd import so3
    from oracle import gp as ogp

    p, q, t, und, cal = _gp_scene(4, uncal=0.2)
    rng = np.random.default_rng(ctype)
    allp = np.array([(a, b) for a in range(p.num_cams) for b in range(p.num_cams) if a != b])
    sel = rng.choice(len(allp), 30, replace=False)  # distinct ordered pairs
    E = len(sel)
    pi, pj = allp[sel, 0].astype(np.int32), allp[sel, 1].astype(np.int32)
    pv = (rng.random(E) > 0.2).astype(np.uint8)
    pt = rng.normal(size=(E, 3))
    pt /= np.linalg.norm(pt, axis=1, keepdims=True)
    kw = dict(constraint_type=ctype, constraint_reweight_scale=2.0)
    r = ref.gp_build(q, t, p.pt_offset, p.obs_cam, und, p.pt_xyz, cam_calibrated=cal, pair_i=pi, pair_j=pj, pair_valid=pv, pair_t=pt, **kw)
    npair = int(pv.sum())
    is_pair = r["pt"] < 0
    # the pairs come first, in the view graph's walk order; their direction: -R_2^T t_21 (:195-197)
    assert is_pair[:npair].all() and not is_pair[npair:].any()
    R = so3.quat_to_rotmat(q)
    got = {(int(a), int(b)): d for a, b, d in zip(r["cam"][:npair], r["cam2"][:npair], r["dir"][:npair])}
    for e in np.nonzero(pv)[0]:
        assert np.abs(got[(int(pi[e]), int(pj[e]))] + R[pj[e]].T @ pt[e]).max() < 1e-15
    assert (r["loss_scale"][:npair] == 1.0).all() and r["scale_const"][0] == 1 and not r["scale_const"][1:].any()
    if ctype == 1:
        assert len(r["pt"]) == npair  # ONLY_CAMERAS: no point blocks
    else:
        w = 2.0 * npair / p.num_pts if ctype == 2 else 1.0  # (#TRACKS in the map, not the used ones: :222)
        cal_blk = p.obs_calibrated[np.concatenate([np.arange(p.pt_offset[tt], p.pt_offset[tt + 1])
                                                   for tt in r["track_order"] if p.pt_offset[tt + 1] - p.pt_offset[tt] >= 3])] != 0
        want = np.where(cal_blk, w if ctype == 2 else 1.0, 0.5 * w)
        assert np.allclose(r["loss_scale"][npair:], want, rtol=1e-15, atol=0)
    # the oracle poses the same problem: same initial cost from the same start
    pr, new_cam, idx = _renumbered(p, r["frame_order"], r["track_order"])
    walk = [e for e in range(E)]  # the view graph is walked in the unordered_map's order too; the glue inserts pairs 0..E-1
    opt = ogp.GlobalPositionerOptions(rand_vector_order=1, **kw)  # (g++-built reference: see the test above)
    opt.lm.max_num_iterations = 0
    # pairs in the reference's walk order = the order of its first npair residual blocks
    order = [next(e for e in np.nonzero(pv)[0] if (int(pi[e]), int(pj[e])) == (int(a), int(b)) and np.abs(d + R[pj[e]].T @ pt[e]).max() < 1e-15)
             for a, b, d in zip(r["cam"][:npair], r["cam2"][:npair], r["dir"][:npair])]
    pdir = np.array([-(R[pj[e]].T @ pt[e]) for e in order])
    ok, c0, X0, summ = ogp.solve(pr.num_cams, pr.pt_offset, pr.obs_cam, pr.obs_dir, pr.obs_calibrated, pr.cam_center, pr.pt_xyz, opt,
                                 pair_i=new_cam[pi[order]], pair_j=new_cam[pj[order]], pair_dir=pdir)
    assert np.abs(c0 - r["center_start"][r["frame_order"]]).max() < 1e-12  # (bit-equal where drawn: test above)
    assert abs(summ.initial_cost - r["initial_cost"]) <= 1e-12 * r["initial_cost"]


# ---------------------------------------------------------------------------------------------------------------
# round 6: the reference's GlobalPositioner::Solve run to its END POINT (oracle/_ref/libref_glomap_gp_solve.so)
# ---------------------------------------------------------------------------------------------------------------
needs_gp_solve = pytest.mark.skipif(ref.load_gp_solve() is None, reason="oracle/_ref/libref_glomap_gp_solve.so not built")


@needs_gp_solve
@pytest.mark.parametrize("N,P,seed", [(14, 90, 0), (40, 800, 1), (60, 2000, 2)])
def test_global_positioning_end_point_equals_the_reference(N, P, seed):
    """global_positioning.cc + cost_function.h, unmodified, on the SOLVING Ceres stand-in (oracle/ref_shim_solve/ceres/ceres.h:
    dual-number Jacobians of the reference's own BATA functors; Ceres' trust-region loop with the projected line search of
    bounds-constrained programs, restated a third time on small dense blocks with exact variable elimination) against
    oracle/gp.py + oracle/lm.py (numpy, sparse Schur complements) from the same std::mt19937 start: the same LM iterations,
    accepted steps and line-search contractions, the same costs, the same camera centres — what pins the oracle's MINIMISER
    to something that shares no code with it and whose residuals / Jacobians come from the reference's source."""
    from glomap_amd import so3
    from oracle import gp as ogp

    p = synthetic.make_gp_problem(num_cams=N, num_pts=P, seed=seed, uncalibrated_ratio=0.2)
    q = so3.rotmat_to_quat(p.cam_R)
    t = -np.einsum("nij,nj->ni", p.cam_R, p.gt_center)
    und = np.einsum("mij,mj->mi", p.cam_R[p.obs_cam], p.obs_dir)
    cal = np.ones(p.num_cams, np.uint8)
    cal[p.obs_cam] = p.obs_calibrated
    p.cam_center = p.gt_center.copy()
    r = ref.gp_solve(q, t, p.pt_offset, p.obs_cam, und, p.pt_xyz, cam_calibrated=cal)
    assert r["ok"] and r["constrained"] and r["line_search_shrunk"] > 0
    pr, new_cam, idx = _renumbered(p, r["frame_order"], r["track_order"])
    ok, c, X, s = ogp.solve(pr.num_cams, pr.pt_offset, pr.obs_cam, pr.obs_dir, pr.obs_calibrated, pr.cam_center, pr.pt_xyz,
                            ogp.GlobalPositionerOptions(rand_vector_order=1))
    assert ok
    assert (s.iterations, s.successful_steps, s.line_search_shrunk) == (r["iterations"], r["successful_steps"], r["line_search_shrunk"])
    assert abs(s.initial_cost - r["initial_cost"]) <= 1e-12 * r["initial_cost"]
    assert abs(s.final_cost - r["final_cost"]) <= 1e-8 * r["final_cost"]
    assert np.allclose(np.array(s.step_sizes), r["trace"][:, 4], rtol=1e-6)  # the line search's step sizes, iteration by iteration
    c_ref = r["center"][r["frame_order"]]
    assert np.abs(c - c_ref).max() <= 1e-7 * np.abs(c_ref).max()  # same start, same gauge: no alignment
