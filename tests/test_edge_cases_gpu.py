"""Edge cases through the C ABI on the GPU: long tracks (> 64 observations: the multi-round path of the
wave tiles), ragged tracks incl. empty and too-short ones, empty problems and bad indices (status codes
instead of crashes), a single camera."""
import copy

import numpy as np
import pytest

from glomap_amd import _lib, estimators, so3, synthetic
from glomap_amd.flat import BaProblem, GpProblem
from oracle import ba as oba
from oracle import gp as ogp

pytestmark = pytest.mark.gpu


def _dense_visibility_gp(ncam=96, npts=60, seed=0, short=25):
    """Every point is seen by EVERY camera (track length 96 > 64) plus `short` ragged tracks (0..4 views)."""
    rng = np.random.default_rng(seed)
    p = synthetic.make_gp_problem(ncam, npts, seed=seed, outlier_ratio=0.0)
    centers, R = p.gt_center, p.cam_R
    X = rng.normal(0, 3.0, (npts + short, 3))
    lens = np.concatenate([np.full(npts, ncam), rng.integers(0, 5, short)])
    off = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    cam = np.concatenate([np.arange(ncam)] * npts + [rng.choice(ncam, int(l), replace=False) for l in lens[npts:]]).astype(np.int32)
    pt = np.repeat(np.arange(npts + short), lens)
    d = X[pt] - centers[cam]
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    return GpProblem(ncam, npts + short, off, cam, d, np.ones(len(cam), np.uint8), np.zeros((ncam, 3)), np.zeros((npts + short, 3)),
                     cam_R=R, gt_center=centers, gt_xyz=X), lens


def test_gp_long_and_ragged_tracks_match_oracle(gsfm_ctx):
    p, lens = _dense_visibility_gp()
    ok, c_o, X_o, s = ogp.solve(p.num_cams, p.pt_offset, p.obs_cam, p.obs_dir, p.obs_calibrated, p.cam_center, p.pt_xyz)
    rc, c_g, X_g, rep = estimators.gp_solve(p, ctx=gsfm_ctx)
    assert ok and rc == 0
    assert synthetic.center_errors_after_sim3(c_g, c_o).max() < 1e-3  # relative to the extent (the helper divides)
    assert synthetic.center_errors_after_sim3(c_g, p.gt_center).max() < 1e-6  # noise-free
    # tracks shorter than min_num_view_per_track (gp.cc:258), the empty ones included, are left untouched
    untouched = lens < 3
    assert untouched.any() and np.array_equal(X_g[untouched], p.pt_xyz[untouched])


def test_ba_long_and_ragged_tracks_match_oracle(gsfm_ctx):
    b = synthetic.make_ba_problem(num_cams=80, num_pts=40, seed=2, pixel_noise=0.3, outlier_ratio=0.0, shared_intrinsics=True)
    rng = np.random.default_rng(2)
    ncam, npts, short = 80, 40, 20
    R = so3.quat_to_rotmat(b.gt_q)
    X = np.concatenate([b.gt_xyz, rng.normal(0, 5.0, (short, 3))])
    lens = np.concatenate([np.full(npts, ncam), rng.integers(0, 5, short)])
    off = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    cam = np.concatenate([np.arange(ncam)] * npts + [rng.choice(ncam, int(l), replace=False) for l in lens[npts:]]).astype(np.int32)
    pt = np.repeat(np.arange(npts + short), lens)
    xc = np.einsum("mij,mj->mi", R[cam], X[pt]) + b.gt_t[cam]
    front = xc[:, 2] > 1.0  # keep the synthetic scene physical: drop views from behind
    keep_obs = front | (pt >= npts)
    xy = synthetic.project_simple_radial(b.gt_intr[np.zeros(len(cam), int)], np.where(xc[:, 2:3] > 0.1, xc, [0, 0, 1.0]))
    xy += rng.normal(0, 0.3, xy.shape)
    # rebuild ragged arrays after dropping back-facing views of the long tracks
    cam, xy, pt = cam[keep_obs], xy[keep_obs], pt[keep_obs]
    lens = np.bincount(pt, minlength=npts + short)
    off = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    assert lens.max() > 64
    X0 = X + rng.normal(0, 0.05, X.shape)
    p = BaProblem(num_cams=ncam, num_pts=npts + short, num_intr=1, pt_offset=off, obs_cam=cam, obs_xy=xy,
                  cam_intr=b.cam_intr, cam_q=b.cam_q, cam_t=b.cam_t, pt_xyz=X0, intr_model=b.intr_model,
                  intr_params=b.intr_params, fixed_cam=0)
    ok, q_o, t_o, X_o, i_o, s = oba.solve(p.num_cams, p.pt_offset, p.obs_cam, p.obs_xy, p.cam_intr, p.intr_model, p.fixed_cam,
                                          p.cam_q, p.cam_t, p.pt_xyz, p.intr_params)
    rc, q_g, t_g, X_g, i_g, rep = estimators.ba_solve(p, ctx=gsfm_ctx)
    assert ok and rc == 0
    assert abs(rep["final_cost"] - s.final_cost) <= 1e-4 * s.final_cost
    ang = np.radians(so3.rotation_angle_deg(so3.quat_to_rotmat(q_g), so3.quat_to_rotmat(q_o)))
    assert ang.max() < 1e-4
    untouched = lens < 3
    assert untouched.any() and np.array_equal(X_g[untouched], X0[untouched])  # ba.cc:122


def test_empty_and_invalid_inputs_return_status_codes(gsfm_ctx):
    # no tracks: reference returns false (gp.cc:46-50, ba.cc:21-24)
    p = GpProblem(4, 0, np.zeros(1, np.int64), np.zeros(0, np.int32), np.zeros((0, 3)), np.zeros(0, np.uint8),
                  np.zeros((4, 3)), np.zeros((0, 3)))
    rc, *_ = estimators.gp_solve(p, ctx=gsfm_ctx)
    assert rc == -5  # GSFM_ERR_EMPTY_PROBLEM
    # only too-short tracks
    p = GpProblem(4, 2, np.array([0, 2, 4], np.int64), np.array([0, 1, 2, 3], np.int32), np.ones((4, 3)) / np.sqrt(3),
                  np.ones(4, np.uint8), np.zeros((4, 3)), np.zeros((2, 3)))
    rc, *_ = estimators.gp_solve(p, ctx=gsfm_ctx)
    assert rc == -5
    # camera index out of range
    p = GpProblem(4, 1, np.array([0, 3], np.int64), np.array([0, 1, 7], np.int32), np.ones((3, 3)) / np.sqrt(3),
                  np.ones(3, np.uint8), np.zeros((4, 3)), np.zeros((1, 3)))
    rc, *_ = estimators.gp_solve(p, ctx=gsfm_ctx)
    assert rc == -1  # GSFM_ERR_INVALID_ARGUMENT
    # unsupported camera model
    b = synthetic.make_ba_problem(num_cams=5, num_pts=40, seed=1, shared_intrinsics=True)
    b.intr_model = np.array([17], np.int32)
    rc, *_ = estimators.ba_solve(b, ctx=gsfm_ctx)
    assert rc == -7  # GSFM_ERR_UNSUPPORTED
    # rotation averaging: gravity together with cam_from_rig unknowns is refused like the reference refuses it (gra.cc:47-58)
    g = synthetic.make_ring_view_graph(20, 3, seed=0)
    gr = copy.deepcopy(g)
    gr.image_frame, gr.image_cam = np.arange(20, dtype=np.int32), np.where(np.arange(20) % 2 == 1, 0, -1).astype(np.int32)
    gr.cam_aa0 = np.zeros((1, 3))
    rc, *_ = estimators.ra_solve(gr, estimators.RotationEstimatorOptions(use_gravity=True, skip_initialization=True), ctx=gsfm_ctx)
    assert rc == -7
    # the context is still usable afterwards
    rc, rot, rep = estimators.ra_solve(g, ctx=gsfm_ctx)
    assert rc == 0


def test_ra_two_nodes_one_edge(gsfm_ctx):
    """Smallest possible view graph."""
    q = so3.aa_to_quat(np.array([[0.0, 0.3, 0.0]]))
    from glomap_amd.flat import RaProblem

    p = RaProblem(2, np.array([0], np.int32), np.array([1], np.int32), q, np.array([1.0]), np.array([50], np.int32),
                  np.zeros((2, 3)), 0)
    rc, rot, rep = estimators.ra_solve(p, ctx=gsfm_ctx)
    assert rc == 0
    R = so3.aa_to_rotmat(rot)
    rel = R[1] @ R[0].T
    assert np.allclose(rel, so3.quat_to_rotmat(q)[0], atol=1e-9)


# ---- skewed visibility: a few images see 10-100x more tracks than the median image ------------------------------
def test_gp_skewed_visibility_matches_oracle(gsfm_ctx):
    """Cameras with > 1024 observations are cut into slices handled by different waves and combined in slice order by a
    second pass of the same kernel (obsgraph.hpp).  Parity with the exact-solve oracle and run-to-run bit-identity (no
    floating-point atomics) on a problem where the busiest camera holds 10 x the median.  Round 6: with Ceres' line search in
    the loop this input is one of the chaotic ones — the oracle summed backwards takes 40 LM iterations where the oracle
    summed forwards takes 42 and ends 3.1e-3 (p99 1.9e-3, median 4.3e-5) away — so the comparison is the one of
    tests/test_fullsize_gpu.py::_gp_parity: same trajectory while the reference follows its own, end point inside its scatter."""
    from test_fullsize_gpu import _assert_gp_parity, _gp_parity

    p = synthetic.make_gp_problem(num_cams=80, num_pts=25_000, seed=4, zipf=1.3)
    per_cam = np.bincount(p.obs_cam, minlength=p.num_cams)
    assert per_cam.max() > 4 * 1024 and np.median(per_cam) < 1024  # cut and whole cameras side by side
    res = _gp_parity("skewed visibility, 80 cameras / 25 000 tracks", p, gsfm_ctx)
    _assert_gp_parity(res)
    opt = None
    rc, cen, xyz, rep = estimators.gp_solve(p, opt, ctx=gsfm_ctx)
    assert rc == 0 and np.array_equal(cen, res["cen"])
    rc, cen2, xyz2, rep2 = estimators.gp_solve(p, opt, ctx=gsfm_ctx)
    assert np.array_equal(cen, cen2) and np.array_equal(xyz, xyz2) and rep2["final_cost"] == rep["final_cost"]


@pytest.mark.parametrize("shared", [False, True])
def test_ba_skewed_visibility_matches_oracle(gsfm_ctx, shared):
    """Same for BA.  One shared camera: 80 images with 50 ... 8 000 observations, 1 % gross outliers.  One camera per image
    (the joint 14 x 14 block path): every image needs enough observations to pin its own focal length, and no gross
    outliers — otherwise the oracle itself is not reproducible (its translations move by > 1 unit along the free scale
    gauge when only the summation order changes)."""
    from oracle import cpu

    if shared:
        p = synthetic.make_ba_problem(num_cams=80, num_pts=25_000, seed=4, zipf=1.3, shared_intrinsics=True)
    else:
        p = synthetic.make_ba_problem(num_cams=60, num_pts=40_000, seed=4, zipf=1.0, shared_intrinsics=False, outlier_ratio=0.0)
    per_cam = np.bincount(p.obs_cam, minlength=p.num_cams)
    assert per_cam.max() > 4 * 1024 and per_cam.min() < 2048
    rc, q, t, X, intr, rep = estimators.ba_solve(p, ctx=gsfm_ctx)
    assert rc == 0
    r = cpu.ba_solve(p.num_cams, p.pt_offset, p.obs_cam, p.obs_xy, p.cam_intr, p.intr_model, p.fixed_cam, p.cam_q, p.cam_t,
                     p.pt_xyz, p.intr_params)
    assert r[0] and (rep["iterations"], rep["successful_steps"]) == (r[5].iterations, r[5].successful_steps)
    assert abs(rep["final_cost"] - r[5].final_cost) <= 1e-6 * r[5].final_cost
    ang = np.radians(so3.rotation_angle_deg(so3.quat_to_rotmat(q), so3.quat_to_rotmat(r[1])))
    assert ang.max() < 1e-6
    assert np.abs(t - r[2]).max() < 1e-5 * 50.0
    rc, q2, t2, X2, intr2, rep2 = estimators.ba_solve(p, ctx=gsfm_ctx)
    assert np.array_equal(q, q2) and np.array_equal(t, t2) and rep2["final_cost"] == rep["final_cost"]
