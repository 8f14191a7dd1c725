"""Processors between the estimator calls: track filters, reconstruction
normaliser, relative-rotation filter.  CPU: the oracle restatement against hand-checkable properties.
GPU: the HIP kernels through the C ABI against the oracle — masks and counters bit-exact, transformed
coordinates to 1e-12 relative."""
import numpy as np
import pytest

from glomap_amd import so3, synthetic
from oracle import filters as of


def _scene(seed=0, ncam=40, npts=3000, noise=2e-3, outliers=0.05):
    p = synthetic.make_gp_problem(ncam, npts, seed=seed, dir_noise=noise, outlier_ratio=outliers)
    q = so3.rotmat_to_quat(p.cam_R)
    t = -np.einsum("nij,nj->ni", p.cam_R, p.gt_center)
    undist = np.einsum("mij,mj->mi", p.cam_R[p.obs_cam], p.obs_dir)  # world rays -> camera rays
    X = p.gt_xyz + np.random.default_rng(seed).normal(0, 0.02, p.gt_xyz.shape)
    return p, q, t, undist, X


def test_oracle_reprojection_and_angle_filters_drop_outliers():
    p, q, t, undist, X = _scene()
    keep, changed = of.filter_tracks_by_reprojection(p.pt_offset, p.obs_cam, q, t, X, 2e-2, True, obs_undist=undist)
    frac = 1.0 - keep.mean()
    assert 0.03 < frac < 0.15 and changed > 0  # ~5 % outlier rays
    keep_a, changed_a = of.filter_tracks_by_angle(p.pt_offset, p.obs_cam, q, t, X, undist, 1.0)
    assert abs(keep_a.mean() - keep.mean()) < 0.03
    # uncalibrated cameras get twice the angle: never fewer observations kept
    keep_u, _ = of.filter_tracks_by_angle(p.pt_offset, p.obs_cam, q, t, X, undist, 1.0, cam_calibrated=np.zeros(p.num_cams, np.uint8))
    assert (keep_u | ~keep_a).all()
    # points behind a camera are dropped whatever the threshold
    t2 = t.copy()
    t2[:, 2] -= 1e3
    k3, _ = of.filter_tracks_by_reprojection(p.pt_offset, p.obs_cam, q, t2, X, 1e9, True, obs_undist=undist)
    assert not k3.any()


def test_oracle_triangulation_angle():
    p, q, t, undist, X = _scene()
    keep, removed = of.filter_tracks_triangulation_angle(p.pt_offset, p.obs_cam, q, t, X, 1.0)
    assert keep.mean() > 0.9 and removed == (~keep).sum()
    # a point very far away subtends no angle
    X2 = X.copy()
    X2[0] = X[0] * 1e6
    keep2, _ = of.filter_tracks_triangulation_angle(p.pt_offset, p.obs_cam, q, t, X2, 1.0)
    assert not keep2[0]
    # the batched form used on millions of tracks (tests/chain_util.py) is the same test
    for pts in (X, X2, X * 40.0):
        a = of.filter_tracks_triangulation_angle(p.pt_offset, p.obs_cam, q, t, pts, 1.0)
        b = of.filter_tracks_triangulation_angle_grouped(p.pt_offset, p.obs_cam, q, t, pts, 1.0)
        assert np.array_equal(a[0], b[0]) and a[1] == b[1]


def test_oracle_normalizer_properties():
    p, q, t, undist, X = _scene(seed=3)
    t2, X2, (scale, trans) = of.normalize_reconstruction(q, t, X)
    R = so3.quat_to_rotmat(q)
    c1 = -np.einsum("nji,nj->ni", R, t)
    c2 = -np.einsum("nji,nj->ni", R, t2)
    assert np.allclose(c2, scale * c1 + trans, atol=1e-9)  # centres follow the similarity
    assert np.allclose(X2, scale * X + trans, atol=1e-9)
    # robust extent of the normalised centres is 10 (reconstruction_normalizer.cc defaults)
    cs = np.sort(c2, axis=0)
    n = cs.shape[0]
    ext = np.linalg.norm(cs[int(0.9 * (n - 1))] - cs[int(0.1 * (n - 1))])
    assert abs(ext - 10.0) < 1e-3
    # reprojection is invariant: x_c scales by `scale`
    pc1 = np.einsum("nij,nj->ni", R[p.obs_cam[:50]], X[np.repeat(np.arange(p.num_pts), np.diff(p.pt_offset))[:50]]) + t[p.obs_cam[:50]]
    pc2 = np.einsum("nij,nj->ni", R[p.obs_cam[:50]], X2[np.repeat(np.arange(p.num_pts), np.diff(p.pt_offset))[:50]]) + t2[p.obs_cam[:50]]
    assert np.allclose(pc2, scale * pc1, atol=1e-9)


def test_oracle_filter_rotations():
    g = synthetic.make_ring_view_graph(100, 8, seed=5)
    nq = so3.rotmat_to_quat(g.gt_R)
    keep, ninv = of.filter_rotations(nq, g.edge_i, g.edge_j, g.edge_q, 5.0)
    assert ninv == (~keep).sum()
    assert (~keep[g.outlier]).mean() > 0.9 and keep[~g.outlier].mean() > 0.99


# ---- GPU parity --------------------------------------------------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("seed", [0, 1])
def test_gpu_track_filters_match_oracle(gsfm_ctx, seed):
    from glomap_amd import processors as pr

    p, q, t, undist, X = _scene(seed=seed)
    cal = (np.random.default_rng(seed).random(p.num_cams) > 0.3).astype(np.uint8)
    view = pr.SceneView(p.num_cams, p.pt_offset, p.obs_cam, q, t, X, obs_undist=undist, cam_calibrated=cal)
    k_o, c_o = of.filter_tracks_by_reprojection(p.pt_offset, p.obs_cam, q, t, X, 1e-2, True, obs_undist=undist)
    k_g, c_g = pr.TrackFilter.FilterTracksByReprojection(view, 1e-2, True, ctx=gsfm_ctx)
    assert np.array_equal(k_g.astype(bool), k_o) and c_g == c_o
    k_o, c_o = of.filter_tracks_by_angle(p.pt_offset, p.obs_cam, q, t, X, undist, 1.0, cam_calibrated=cal)
    k_g, c_g = pr.TrackFilter.FilterTracksByAngle(view, 1.0, ctx=gsfm_ctx)
    assert np.array_equal(k_g.astype(bool), k_o) and c_g == c_o
    k_o, c_o = of.filter_tracks_triangulation_angle(p.pt_offset, p.obs_cam, q, t, X, 1.0)
    k_g, c_g = pr.TrackFilter.FilterTrackTriangulationAngle(view, 1.0, ctx=gsfm_ctx)
    assert np.array_equal(k_g.astype(bool), k_o) and c_g == c_o


@pytest.mark.gpu
def test_gpu_pixel_reprojection_filter_matches_oracle(gsfm_ctx):
    from glomap_amd import processors as pr

    b = synthetic.make_ba_problem(num_cams=25, num_pts=1500, seed=6, shared_intrinsics=False, outlier_ratio=0.05)
    view = pr.SceneView(b.num_cams, b.pt_offset, b.obs_cam, b.gt_q, b.gt_t, b.gt_xyz, obs_xy=b.obs_xy, cam_intr=b.cam_intr,
                        intr_model=b.intr_model, intr_params=b.gt_intr)
    k_o, c_o = of.filter_tracks_by_reprojection(b.pt_offset, b.obs_cam, b.gt_q, b.gt_t, b.gt_xyz, 4.0, False, obs_xy=b.obs_xy,
                                                cam_intr=b.cam_intr, intr_model=b.intr_model, intr_params=b.gt_intr)
    k_g, c_g = pr.TrackFilter.FilterTracksByReprojection(view, 4.0, False, ctx=gsfm_ctx)
    assert np.array_equal(k_g.astype(bool), k_o) and c_g == c_o
    assert 0.02 < 1 - k_o.mean() < 0.1


@pytest.mark.gpu
def test_gpu_normalizer_and_rotation_filter_match_oracle(gsfm_ctx):
    from glomap_amd import processors as pr

    p, q, t, undist, X = _scene(seed=4)
    reg = np.ones(p.num_cams, np.uint8)
    reg[::7] = 0
    t_o, X_o, (s_o, tr_o) = of.normalize_reconstruction(q, t, X, cam_registered=reg)
    t_g, X_g, (s_g, tr_g) = pr.NormalizeReconstruction(q, t, X, cam_registered=reg, ctx=gsfm_ctx)
    assert abs(s_g - s_o) <= 1e-12 * s_o and np.allclose(tr_g, tr_o, rtol=1e-12, atol=1e-12)
    assert np.allclose(t_g, t_o, rtol=1e-12, atol=1e-10) and np.allclose(X_g, X_o, rtol=1e-12, atol=1e-10)
    g = synthetic.make_ring_view_graph(300, 10, seed=7)
    nq = so3.rotmat_to_quat(g.gt_R)
    k_o, n_o = of.filter_rotations(nq, g.edge_i, g.edge_j, g.edge_q, 5.0)
    k_g, n_g = pr.RelPoseFilter.FilterRotations(nq, g.edge_i, g.edge_j, g.edge_q, 5.0, ctx=gsfm_ctx)
    assert np.array_equal(k_g.astype(bool), k_o) and n_g == n_o


@pytest.mark.gpu
def test_gpu_filters_reject_malformed_views(gsfm_ctx):
    """Out-of-range indices must come back as GSFM_ERR_INVALID_ARGUMENT, not as out-of-bounds device reads."""
    from glomap_amd import _lib
    from glomap_amd import processors as pr

    p, q, t, undist, X = _scene(seed=0, ncam=20, npts=400)
    bad_cam = p.obs_cam.copy()
    bad_cam[17] = p.num_cams  # one past the end
    bad_off = p.pt_offset.copy()
    bad_off[5], bad_off[6] = bad_off[6], bad_off[5] - 1  # not monotone
    short_off = p.pt_offset.copy()
    short_off[-1] -= 1  # does not end at num_obs
    for off, cam in ((p.pt_offset, bad_cam), (bad_off, p.obs_cam), (short_off, p.obs_cam)):
        view = pr.SceneView(p.num_cams, off, cam, q, t, X, obs_undist=undist)
        with pytest.raises(_lib.GsfmError) as e:
            pr.TrackFilter.FilterTracksByAngle(view, 1.0, ctx=gsfm_ctx)
        assert e.value.status == -1  # GSFM_ERR_INVALID_ARGUMENT
    g = synthetic.make_ring_view_graph(50, 5, seed=1)
    ej = g.edge_j.copy()
    ej[3] = -1
    with pytest.raises(_lib.GsfmError) as e:
        pr.RelPoseFilter.FilterRotations(so3.rotmat_to_quat(g.gt_R), g.edge_i, ej, g.edge_q, 5.0, ctx=gsfm_ctx)
    assert e.value.status == -1  # GSFM_ERR_INVALID_ARGUMENT
    # the context is still usable afterwards
    view = pr.SceneView(p.num_cams, p.pt_offset, p.obs_cam, q, t, X, obs_undist=undist)
    k, c = pr.TrackFilter.FilterTracksByAngle(view, 1.0, ctx=gsfm_ctx)
    assert k.shape[0] == p.num_obs


def _compact_reference(off, obs_keep, track_keep, arrays):
    """What gsfm_tracks_compact has to produce (numpy): survivors in order, new offsets."""
    lens = np.diff(off)
    trk = np.repeat(np.arange(len(lens)), lens)
    keep = np.ones(int(off[-1]), bool)
    if obs_keep is not None:
        keep &= obs_keep.astype(bool)
    if track_keep is not None:
        keep &= track_keep.astype(bool)[trk]
    new_off = np.zeros(len(lens) + 1, dtype=np.int64)
    new_off[1:] = np.cumsum(np.bincount(trk[keep], minlength=len(lens)))
    return new_off, [a[keep] for a in arrays]


@pytest.mark.gpu
@pytest.mark.parametrize("case", ["mixed", "obs_only", "tracks_only", "all_dropped", "none_dropped", "empty"])
def test_gpu_tracks_compact_matches_numpy(gsfm_ctx, case):
    """gsfm_tracks_compact (the erase of track_filter.cc:36-44, 75-83, 120-123 in the flat layout), host and device memory,
    ragged tracks with empty ones among them: offsets and every per-observation array equal to the numpy compaction."""
    from glomap_amd import _lib
    from glomap_amd import processors as pr

    rng = np.random.default_rng(7)
    P = 0 if case == "empty" else 5000
    lens = rng.integers(0, 9, P)
    lens[rng.random(P) < 0.1] = 0
    off = np.zeros(P + 1, dtype=np.int64)
    off[1:] = np.cumsum(lens)
    M = int(off[-1])
    cam = rng.integers(0, 100, M).astype(np.int32)
    xy = rng.normal(size=(M, 2))
    und = rng.normal(size=(M, 3))
    ids = rng.integers(0, 1 << 40, M).astype(np.int64)
    obs_keep = (rng.random(M) < 0.7).astype(np.uint8)
    track_keep = (rng.random(P) < 0.8).astype(np.uint8)
    if case == "obs_only":
        track_keep = None
    elif case == "tracks_only":
        obs_keep = None
    elif case == "all_dropped":
        obs_keep[:] = 0
    elif case == "none_dropped":
        obs_keep, track_keep = None, np.ones(P, np.uint8)
    want_off, want = _compact_reference(off, obs_keep, track_keep, [cam, xy, und, ids])
    # host memory
    h_off, h_arr = off.copy(), [cam.copy(), xy.copy(), und.copy(), ids.copy()]
    n, cut = pr.CompactObservations(h_off, h_arr, obs_keep=obs_keep, track_keep=track_keep, ctx=gsfm_ctx)
    assert n == int(want_off[-1]) and np.array_equal(h_off, want_off)
    for a, b in zip(cut, want):
        assert a.shape == b.shape and np.array_equal(a, b)
    # device memory: nothing but the count comes back until we ask for it
    if M == 0:
        return  # zero-byte device allocations are not part of the contract
    up = lambda a: None if a is None else _lib.DeviceArray.from_numpy(gsfm_ctx, a)  # noqa: E731
    d_off, d_arr = up(off), [up(cam), up(xy), up(und), up(ids)]
    n, cut = pr.CompactObservations(d_off, d_arr, obs_keep=up(obs_keep), track_keep=up(track_keep), ctx=gsfm_ctx)
    assert n == int(want_off[-1]) and np.array_equal(d_off.numpy(), want_off)
    for a, b in zip(cut, want):
        assert a.shape == b.shape and np.array_equal(a.numpy(), b)


# every COLMAP camera model of include/gsfm.h with parameters that bend the image noticeably
_MODELS = {
    0: [1200, 640, 480], 1: [1200, 1190, 640, 480], 2: [1200, 640, 480, 0.02], 3: [1200, 640, 480, 0.02, -0.01],
    4: [1200, 1190, 640, 480, 0.02, -0.01, 0.001, -0.002], 5: [1200, 1190, 640, 480, 0.02, -0.01, 0.004, -0.002],
    7: [1200, 1190, 640, 480, 0.8], 8: [1200, 640, 480, 0.02], 9: [1200, 640, 480, 0.02, -0.01],
    6: [1200, 1190, 640, 480, 0.02, -0.01, 0.001, -0.002, 0.003, 0.01, -0.004, 0.002],
    10: [1200, 1190, 640, 480, 0.02, -0.01, 0.001, -0.002, 0.004, -0.002, 0.0015, -0.001],
    11: [1200, 1190, 640, 480, 0.02, -0.01, 0.004, -0.002, 0.001, -0.0005, 0.001, -0.002, 0.0015, -0.0008, -0.001, 0.0005],
}


def _undistort_case(models, F, width, seed=0):
    """F pixels = projections of random rays (|u|, |v| <= 0.5) through cameras of the given models, round-robin."""
    from oracle import ba as oba

    rng = np.random.default_rng(seed)
    K = len(models)
    par = np.zeros((K, width))
    for k, m in enumerate(models):
        par[k, : len(_MODELS[m])] = _MODELS[m]
        par[k, 0] *= 1.0 + 0.01 * k  # (not all cameras alike)
    model = np.array(models, dtype=np.int32)
    fi = (np.arange(F) % K).astype(np.int32)
    ray = np.concatenate([rng.uniform(-0.5, 0.5, (F, 2)), np.ones((F, 1))], 1)
    uv, _, _, valid = oba.project(model[fi], np.pad(par, ((0, 0), (0, 16 - width)))[fi], ray)
    assert valid.all()
    return uv, fi, model, par, ray / np.linalg.norm(ray, axis=1, keepdims=True)


def test_oracle_undistortion_inverts_the_projection_of_every_camera_model():
    """oracle.filters.undistort_features (UndistortImages, image_undistorter.cc:7-46, COLMAP's IterativeUndistortion restated):
    project(undistort(pixel)) = pixel for the twelve models."""
    for m in _MODELS:
        uv, fi, model, par, ray = _undistort_case([m], 300, 16, seed=m)
        out = of.undistort_features(uv, fi, model, par)
        assert np.allclose(np.linalg.norm(out, axis=1), 1.0, atol=1e-14)
        assert np.abs(out - ray).max() < 1e-9, (m, np.abs(out - ray).max())


@pytest.mark.gpu
@pytest.mark.parametrize("width", [8, 16])
def test_gpu_undistortion_matches_oracle(gsfm_ctx, width):
    """gsfm_undistort_features against the oracle on a mix of camera models (8-wide rows: the nine models with at most eight
    parameters; 16-wide: all twelve), host arrays and device arrays; a pixel far outside any image converges or comes back
    as (0, 0, 1) on both sides."""
    from glomap_amd import processors

    models = [m for m in sorted(_MODELS) if width == 16 or len(_MODELS[m]) <= 8]
    uv, fi, model, par, ray = _undistort_case(models, 20_000, width, seed=3)
    want = of.undistort_features(uv, fi, model, par)
    got = processors.UndistortFeatures(uv, fi, model, par, ctx=gsfm_ctx)
    d = np.abs(got - want).max()
    print(f"[parity] undistortion, {len(models)} camera models, {width}-wide rows: max |ray - oracle| {d:.2e}, max |ray - truth| {np.abs(got - ray).max():.2e}")
    assert d < 1e-9 and np.abs(got - ray).max() < 1e-9
    dev = processors.UndistortFeatures(gsfm_ctx.to_device(uv), gsfm_ctx.to_device(fi), gsfm_ctx.to_device(model),
                                       gsfm_ctx.to_device(par), ctx=gsfm_ctx)
    assert np.array_equal(dev.numpy(), got)


@pytest.mark.gpu
def test_gpu_undistortion_rejects_malformed_input_and_scales(gsfm_ctx):
    from glomap_amd import _lib, processors

    uv, fi, model, par, ray = _undistort_case([2], 3_000_000, 8, seed=5)  # configs[2]-size feature set, SIMPLE_RADIAL
    got = processors.UndistortFeatures(uv, fi, model, par, ctx=gsfm_ctx)
    assert np.abs(got - ray).max() < 1e-9
    bad = fi.copy()
    bad[17] = 4
    with pytest.raises(_lib.GsfmError):
        processors.UndistortFeatures(uv[:100], bad[:100], model, par, ctx=gsfm_ctx)
    with pytest.raises(_lib.GsfmError):  # a 12-parameter model in 8-wide rows
        processors.UndistortFeatures(uv[:100], fi[:100], np.array([6], dtype=np.int32), par, ctx=gsfm_ctx)
    assert processors.UndistortFeatures(uv[:0], fi[:0], model, par, ctx=gsfm_ctx).shape == (0, 3)


@pytest.mark.gpu
def test_gpu_undistort_images_on_scene_containers(gsfm_ctx):
    """processors.UndistortImages — the reference's UndistortImages(cameras, images, clean_points) on the containers of
    glomap_amd.scene: two cameras of different models (8- and 12-parameter: 16-wide rows), clean_points semantics."""
    from glomap_amd import processors, scene

    uv, fi, model, par, ray = _undistort_case([4, 6], 4000, 16, seed=9)
    cams = {10: scene.Camera(10, 4, par[0, :8].copy()), 20: scene.Camera(20, 6, par[1, :12].copy())}
    imgs = {}
    for i in range(8):
        sel = np.nonzero(fi == (i % 2))[0][i // 2 :: 4]
        imgs[i] = scene.Image(i, 10 if i % 2 == 0 else 20, i, features=uv[sel].copy())
        imgs[i]._truth = ray[sel]
    imgs[3].features_undist = np.tile([0.0, 0.6, 0.8], (len(imgs[3].features), 1))  # complete, stale
    processors.UndistortImages(cams, imgs, clean_points=False, ctx=gsfm_ctx)
    assert np.array_equal(imgs[3].features_undist[0], [0.0, 0.6, 0.8])  # image_undistorter.cc:13-15: already undistorted
    for i in (0, 1, 2, 4, 5, 6, 7):
        assert np.abs(imgs[i].features_undist - imgs[i]._truth).max() < 1e-9
    processors.UndistortImages(cams, imgs, clean_points=True, ctx=gsfm_ctx)
    assert np.abs(imgs[3].features_undist - imgs[3]._truth).max() < 1e-9
