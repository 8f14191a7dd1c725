"""GPU parity: HIP global positioning (through the C ABI) against the CPU oracle on the same seeded
inputs and the same (std::mt19937) random initialisation.  Tolerance (north_star): camera
positions within 1e-3 relative after Sim(3) alignment."""
import numpy as np
import pytest

from glomap_amd import estimators, synthetic
from oracle import gp as ogp

pytestmark = pytest.mark.gpu
TOL_REL = 1e-3


def _oracle(p, **kw):
    opt = ogp.GlobalPositionerOptions(**kw)
    return ogp.solve(p.num_cams, p.pt_offset, p.obs_cam, p.obs_dir, p.obs_calibrated, p.cam_center, p.pt_xyz, opt)


def _rel_diff(a, b):
    """Centre difference after Sim(3) alignment of a onto b, relative to the extent of b."""
    return synthetic.center_errors_after_sim3(a, b).max()


@pytest.mark.parametrize(
    "ncam,npts,noise,outl,unc,seed",
    [(30, 400, 0.0, 0.0, 0.0, 0), (40, 800, 1e-3, 0.02, 0.2, 1), (80, 3000, 1e-3, 0.02, 0.0, 3),
     (300, 20000, 1e-3, 0.02, 0.1, 5)],  # the last one: ~120k observations, the largest the oracle solves in seconds
)
def test_gp_matches_oracle(gsfm_ctx, ncam, npts, noise, outl, unc, seed):
    p = synthetic.make_gp_problem(num_cams=ncam, num_pts=npts, seed=seed, dir_noise=noise, outlier_ratio=outl,
                                  uncalibrated_ratio=unc)
    ok, c_o, X_o, summ = _oracle(p)
    assert ok
    rc, c_g, X_g, rep = estimators.gp_solve(p, ctx=gsfm_ctx)
    assert rc == 0
    print("oracle", summ.iterations, summ.successful_steps, summ.final_cost, "gpu", rep)
    assert abs(rep["initial_cost"] - summ.initial_cost) <= 1e-9 * summ.initial_cost
    if noise == 0.0:
        assert rep["final_cost"] < 1e-12
    else:
        assert abs(rep["final_cost"] - summ.final_cost) <= 1e-3 * summ.final_cost
    assert _rel_diff(c_g, c_o) < TOL_REL
    # ground-truth recovery with the reference's tolerances (global_mapper_test.cc:82-86, 211-215)
    assert _rel_diff(c_g, p.gt_center) < (1e-4 if noise == 0.0 else 0.1)


def test_gp_short_tracks_untouched_and_flags(gsfm_ctx):
    p = synthetic.make_gp_problem(num_cams=25, num_pts=300, seed=5)
    lens = np.diff(p.pt_offset)
    drop = int(lens[0] - 2)
    keep = np.ones(p.num_obs, dtype=bool)
    keep[2 : 2 + drop] = False
    p.obs_cam, p.obs_dir, p.obs_calibrated = p.obs_cam[keep], p.obs_dir[keep], p.obs_calibrated[keep]
    p.pt_offset = np.concatenate([[0], np.cumsum(np.concatenate([[2], lens[1:]]))]).astype(np.int64)
    p.pt_xyz[0] = [7.0, 8.0, 9.0]
    rc, c_g, X_g, rep = estimators.gp_solve(p, ctx=gsfm_ctx)
    assert rc == 0 and np.all(X_g[0] == [7.0, 8.0, 9.0])
    ok, c_o, X_o, _ = _oracle(p)
    assert _rel_diff(c_g, c_o) < TOL_REL


def test_gp_fixed_positions_points_only(gsfm_ctx):
    """optimize_positions = false: cameras stay, only points and scales move (gp.cc:146-152, 456-464)."""
    p = synthetic.make_gp_problem(num_cams=25, num_pts=300, seed=6, dir_noise=0.0, outlier_ratio=0.0)
    p.cam_center = p.gt_center.copy()
    opt = estimators.GlobalPositionerOptions(optimize_positions=False)
    rc, c_g, X_g, rep = estimators.gp_solve(p, opt, ctx=gsfm_ctx)
    assert rc == 0
    assert np.array_equal(c_g, p.gt_center)
    used = np.diff(p.pt_offset) >= 3
    # the first used track carries the constant (gauge) scale s = 1 (gp.cc:484-489), which with
    # fixed cameras pins that point at unit distance: exclude it
    first = int(np.argmax(used))
    used[first] = False
    assert np.abs(X_g[used] - p.gt_xyz[used]).max() < 1e-3  # stops on function_tolerance 1e-5


def test_gp_empty_inputs_fail_like_reference(gsfm_ctx):
    p = synthetic.make_gp_problem(num_cams=10, num_pts=20, seed=0)
    p.obs_cam, p.obs_dir, p.obs_calibrated = p.obs_cam[:0], p.obs_dir[:0], p.obs_calibrated[:0]
    p.pt_offset = np.zeros(1, dtype=np.int64)
    p.num_pts = 0
    p.pt_xyz = np.zeros((0, 3))
    rc, *_ = estimators.gp_solve(p, ctx=gsfm_ctx)
    assert rc == -5  # GSFM_ERR_EMPTY_PROBLEM (gp.cc:46-50 returns false)


def test_gp_config3_scaled_properties(gsfm_ctx):
    """C3-shaped problem (cameras on a ring, ball of points) at 1/10 scale: 500 cameras / 50k tracks.
    Size-independent checks: converged, ground truth recovered within the reference's noisy
    tolerance, re-running from the same seed reproduces the solution to solver tolerance."""
    p = synthetic.make_gp_problem(num_cams=500, num_pts=50_000, seed=0)
    rc, c1, X1, rep1 = estimators.gp_solve(p, ctx=gsfm_ctx)
    assert rc == 0 and rep1["termination"] == 0
    assert _rel_diff(c1, p.gt_center) < 0.1
    assert np.median(synthetic.center_errors_after_sim3(c1, p.gt_center)) < 5e-3
    rc, c2, X2, rep2 = estimators.gp_solve(p, ctx=gsfm_ctx)
    assert _rel_diff(c2, c1) < 1e-6


def _pairs(p, rng, num_succ=4, noise=0.0):
    """Camera-to-camera directions as GlobalPositioner::AddCameraToCameraConstraints sees them (gp.cc:195-197):
    -R_cw2^T t_21 = c_2 - c_1 in the scale of the two-view geometry (unit translation); each camera with its next
    `num_succ` neighbours on the ring."""
    N = p.num_cams
    i = np.repeat(np.arange(N), num_succ)
    j = (i + np.tile(np.arange(1, num_succ + 1), N)) % N
    d = p.gt_center[j] - p.gt_center[i]
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    d += noise * rng.normal(size=d.shape)
    return i.astype(np.int32), j.astype(np.int32), d / np.linalg.norm(d, axis=1, keepdims=True)


@pytest.mark.parametrize("ctype", [ogp.ONLY_CAMERAS, ogp.POINTS_AND_CAMERAS_BALANCED, ogp.POINTS_AND_CAMERAS])
@pytest.mark.parametrize("ncam,npts,noise,seed", [(24, 300, 0.0, 3), (60, 1500, 2e-3, 7)])
def test_gp_constraint_types_match_oracle(gsfm_ctx, ctype, ncam, npts, noise, seed):
    """The estimator's other constraint types (gp.cc:42-64, 167-210, 223-253: camera-to-camera BATA pairs with a scale per
    pair, alone or next to the tracks, the point losses re-weighted in the BALANCED type) against the numpy oracle on the
    same seeded inputs: same start (same draws, same cost to 1e-9), same final cost, centres within 1e-3 after Sim(3)."""
    p = synthetic.make_gp_problem(num_cams=ncam, num_pts=npts, seed=seed, dir_noise=noise, outlier_ratio=0.02 if noise else 0.0)
    p.pair_i, p.pair_j, p.pair_dir = _pairs(p, np.random.default_rng(seed), noise=noise)
    kw = dict(constraint_type=int(ctype), constraint_reweight_scale=2.0)
    ok, c_o, X_o, summ = _oracle_pairs(p, **kw)
    assert ok
    rc, c_g, X_g, rep = estimators.gp_solve(p, estimators.GlobalPositionerOptions(**kw), ctx=gsfm_ctx)
    assert rc == 0
    print("oracle", summ.iterations, summ.successful_steps, summ.initial_cost, summ.final_cost, "gpu", rep)
    assert abs(rep["initial_cost"] - summ.initial_cost) <= 1e-9 * summ.initial_cost
    if noise == 0.0:
        assert rep["final_cost"] < 1e-10
    else:
        assert abs(rep["final_cost"] - summ.final_cost) <= 1e-3 * summ.final_cost
    assert _rel_diff(c_g, c_o) < TOL_REL
    assert _rel_diff(c_g, p.gt_center) < (1e-4 if noise == 0.0 else 0.1)
    if ctype == ogp.ONLY_CAMERAS:  # points are not part of the problem: left as they came in
        assert np.array_equal(X_g, p.pt_xyz)


def test_gp_only_cameras_without_any_track(gsfm_ctx):
    """ONLY_CAMERAS positions the cameras from the view graph alone: the reference asks for tracks only when the constraint
    type uses them (gp.cc:46-50).  An empty track set reaches the C ABI as null arrays (std::vector<T>(0).data())."""
    p = synthetic.make_gp_problem(num_cams=24, num_pts=300, seed=3, dir_noise=0.0, outlier_ratio=0.0)
    p.pair_i, p.pair_j, p.pair_dir = _pairs(p, np.random.default_rng(3), noise=0.0)
    kw = dict(constraint_type=int(ogp.ONLY_CAMERAS))
    rc, c_ref, _, rep_ref = estimators.gp_solve(p, estimators.GlobalPositionerOptions(**kw), ctx=gsfm_ctx)
    assert rc == 0
    e = type(p)(num_cams=p.num_cams, num_pts=0, pt_offset=np.zeros(1, np.int64), obs_cam=np.zeros(0, np.int32),
                obs_dir=np.zeros((0, 3)), obs_calibrated=np.zeros(0, np.uint8), cam_center=p.cam_center, pt_xyz=np.zeros((0, 3)))
    e.pair_i, e.pair_j, e.pair_dir = p.pair_i, p.pair_j, p.pair_dir
    rc, c_e, X_e, rep = estimators.gp_solve(e, estimators.GlobalPositionerOptions(**kw), ctx=gsfm_ctx)
    assert rc == 0 and X_e.shape == (0, 3)
    assert rep["final_cost"] < 1e-10
    assert _rel_diff(c_e, p.gt_center) < 1e-4
    # with tracks present the frames the kept tracks touch are re-drawn too (gp.cc:121-163), so the random starts differ;
    # both recover the noise-free scene
    assert _rel_diff(c_ref, p.gt_center) < 1e-4
    # the other types still refuse an empty track set (gp.cc:46-50)
    rc, *_ = estimators.gp_solve(e, estimators.GlobalPositionerOptions(constraint_type=int(ogp.POINTS_AND_CAMERAS)), ctx=gsfm_ctx)
    assert rc == _lib_status("GSFM_ERR_EMPTY_PROBLEM")


def _oracle_pairs(p, **kw):
    opt = ogp.GlobalPositionerOptions(**kw)
    return ogp.solve(p.num_cams, p.pt_offset, p.obs_cam, p.obs_dir, p.obs_calibrated, p.cam_center, p.pt_xyz, opt,
                     pair_i=p.pair_i, pair_j=p.pair_j, pair_dir=p.pair_dir)


def test_gp_constraint_types_reject_what_the_reference_rejects(gsfm_ctx):
    p = synthetic.make_gp_problem(num_cams=10, num_pts=60, seed=4)
    rc, *_ = estimators.gp_solve(p, estimators.GlobalPositionerOptions(constraint_type=1), ctx=gsfm_ctx)
    assert rc == _lib_status("GSFM_ERR_EMPTY_PROBLEM")  # no pairs: gp.cc:41-45


def _lib_status(name):
    from glomap_amd import _lib

    return {v: k for k, v in _lib.STATUS_NAMES.items()}[name]


@pytest.mark.parametrize("case", ["uniform", "skewed", "short_tracks", "pairs"])
def test_gp_chunked_camera_side_sweep_equals_camera_major(gsfm_ctx, case):
    """The PCG's camera-side sweep has two layouts: one wave per camera over the camera-major lists (k_gp_phaseB) and one lane
    per observation over the (point chunk, camera) order with every XCD on its own chunks (k_gp_phaseB_x + k_gp_wsum,
    obsgraph.hpp ObsX) — chosen by the library when the point records overflow an XCD's L2.  Same operator, another
    summation order: forced on (default chunk count, and 8 / 64 chunks) on problems far below that size, the solve must
    follow the camera-major one — same accept / reject decisions over 8 LM iterations, centres equal to 1e-5 of the extent
    (the reduced solves stop at 1e-12, near the floor of what the recurrence residual reaches, so the two summation orders may
    stop one PCG iteration apart; eight LM iterations from a random start amplify that ~1e4-fold: 2.5e-8 on the cost of the
    `pairs` case) — use the chunked kernels (stats), leave what the camera-major solve leaves untouched, and repeat bit for
    bit."""
    kw = {}
    if case == "skewed":  # busiest camera far above the median: many pieces per camera, pieces longer than a tile
        p = synthetic.make_gp_problem(num_cams=80, num_pts=25_000, seed=4, zipf=1.3)
    elif case == "short_tracks":  # tracks below min_num_view_per_track are not part of any order
        p = synthetic.make_gp_problem(num_cams=120, num_pts=6_000, seed=2)
        kw = dict(min_num_view_per_track=6)
    else:
        p = synthetic.make_gp_problem(num_cams=300, num_pts=20_000, seed=5, uncalibrated_ratio=0.1)
    opt = estimators.GlobalPositionerOptions(**kw)
    if case == "pairs":  # camera-to-camera terms on top of the sweep (their delta slots follow the sweep's)
        p.pair_i, p.pair_j, p.pair_dir = _pairs(p, np.random.default_rng(0), noise=1e-3)
        opt.constraint_type = 3  # POINTS_AND_CAMERAS
    # a fixed number of LM iterations: both layouts stop at the same iterate (function_tolerance would end the two
    # trajectories wherever rounding puts them)
    opt.solver_options.max_num_iterations = 8
    try:
        gsfm_ctx.set_knob("chunked_sweeps", 2)
        rc, c0, X0, rep0 = estimators.gp_solve(p, opt, ctx=gsfm_ctx)
        assert rc == 0
        for knob in (1, 8, 64):
            gsfm_ctx.stats(reset=True)
            gsfm_ctx.set_knob("chunked_sweeps", knob)
            rc, c1, X1, rep1 = estimators.gp_solve(p, opt, ctx=gsfm_ctx)
            assert rc == 0
            st = gsfm_ctx.stats()
            assert st["pcg_chunked_sweeps"] == st["pcg_solves"] > 0
            print(case, knob, rep0["iterations"], rep1["iterations"], rep0["final_cost"], rep1["final_cost"])
            assert rep1["iterations"] == rep0["iterations"] and rep1["successful_steps"] == rep0["successful_steps"]
            assert abs(rep1["final_cost"] - rep0["final_cost"]) <= 1e-6 * rep0["final_cost"]
            ext = np.linalg.norm(c0 - c0.mean(0), axis=1).max()
            assert np.abs(c1 - c0).max() <= 1e-5 * ext
            assert np.array_equal(X1[np.diff(p.pt_offset) < opt.min_num_view_per_track],
                                  X0[np.diff(p.pt_offset) < opt.min_num_view_per_track])
            rc, c2, X2, rep2 = estimators.gp_solve(p, opt, ctx=gsfm_ctx)
            assert np.array_equal(c1, c2) and np.array_equal(X1, X2)
    finally:
        gsfm_ctx.set_knob("chunked_sweeps", 0)


def test_gp_draw_orders_keep_the_start_under_renumbering(gsfm_ctx):
    """gsfm_gp_problem::cam_draw_order / pt_draw_order: a caller that numbers frames and tracks for locality (the C++ adapter:
    ascending ids instead of hash-map order) hands over the order in which the reference's containers are walked, and the
    random start is the one the reference draws — here: a problem with cameras and tracks renumbered at random plus the
    draw orders of the original numbering starts from the SAME cost (1e-12: other summation order) and ends at the same
    cameras; without the draw orders it is another start."""
    import copy

    p = synthetic.make_gp_problem(num_cams=60, num_pts=1500, seed=7)
    rc, c0, X0, rep0 = estimators.gp_solve(p, ctx=gsfm_ctx)
    assert rc == 0
    rng = np.random.default_rng(1)
    N, P = p.num_cams, p.num_pts
    cam_new = rng.permutation(N).astype(np.int32)      # new index of old camera n
    trk_old = rng.permutation(P)                       # new track i = old track trk_old[i]
    trk_new = np.empty(P, np.int64)
    trk_new[trk_old] = np.arange(P)                    # new index of old track p
    lens = np.diff(p.pt_offset)
    q = copy.copy(p)
    q.pt_offset = np.concatenate([[0], np.cumsum(lens[trk_old])]).astype(np.int64)
    idx = np.concatenate([np.arange(p.pt_offset[t], p.pt_offset[t + 1]) for t in trk_old])
    q.obs_cam = cam_new[p.obs_cam[idx]].astype(np.int32)
    q.obs_dir = np.ascontiguousarray(p.obs_dir[idx])
    q.obs_calibrated = np.ascontiguousarray(p.obs_calibrated[idx])
    q.cam_center = np.zeros_like(p.cam_center)
    q.pt_xyz = np.zeros_like(p.pt_xyz)
    q.cam_draw_order = cam_new.copy()                  # the i-th camera draw goes to old camera i = new index cam_new[i]
    q.pt_draw_order = trk_new.astype(np.int32)         # the i-th track draw goes to old track i = new index trk_new[i]
    rc, c1, X1, rep1 = estimators.gp_solve(q, ctx=gsfm_ctx)
    assert rc == 0
    assert abs(rep1["initial_cost"] - rep0["initial_cost"]) <= 1e-12 * rep0["initial_cost"]
    assert _rel_diff(c1[cam_new], c0) < TOL_REL
    q.cam_draw_order = q.pt_draw_order = None
    rc, c2, X2, rep2 = estimators.gp_solve(q, ctx=gsfm_ctx)
    assert rc == 0 and abs(rep2["initial_cost"] - rep0["initial_cost"]) > 1e-6 * rep0["initial_cost"]
    q.cam_draw_order = np.zeros(N, np.int32)           # not a permutation
    rc, *_ = estimators.gp_solve(q, ctx=gsfm_ctx)
    assert rc != 0


def test_gp_rand_vector_order_of_a_gcc_built_reference(gsfm_ctx):
    """RandVector3d's three draws are constructor ARGUMENTS (global_positioning.cc:21-23): a g++-built GLOMAP evaluates them
    right to left, so the first draw of each vector lands in z (tests/test_oracle_ref.py pins this against the reference's
    own translation unit).  rand_vector_order = 1 reproduces that start on both sides; the default stays x-first."""
    p = synthetic.make_gp_problem(num_cams=40, num_pts=800, seed=1, dir_noise=1e-3, outlier_ratio=0.02, uncalibrated_ratio=0.2)
    ok, c_o, X_o, summ = _oracle(p, rand_vector_order=1)
    assert ok
    rc, c_g, X_g, rep = estimators.gp_solve(p, estimators.GlobalPositionerOptions(rand_vector_order=1), ctx=gsfm_ctx)
    assert rc == 0
    assert abs(rep["initial_cost"] - summ.initial_cost) <= 1e-9 * summ.initial_cost
    assert abs(rep["final_cost"] - summ.final_cost) <= 1e-3 * summ.final_cost
    assert _rel_diff(c_g, c_o) < TOL_REL
    # and it IS a different start from the default order
    rc, _, _, rep0 = estimators.gp_solve(p, ctx=gsfm_ctx)
    assert rc == 0 and abs(rep0["initial_cost"] - rep["initial_cost"]) > 1e-6 * rep["initial_cost"]


@pytest.mark.parametrize("ncam,npts,min_iters,chunk_knob", [(1500, 90_000, 8, 1), (1500, 90_000, 8, 2), (5000, 500_000, 0, 0)])
def test_gp_recycled_ritz_vectors_same_solution_fewer_iterations(gsfm_ctx, ncam, npts, min_iters, chunk_knob):
    """The recycled-Ritz-vector preconditioner (cg.hpp CgRecycle, ritz.hpp).  The SAME reduced systems are
    solved to the SAME tolerance, so with the knob gp_no_recycle set and cleared the LM paths agree to solver precision while
    they are stable — the first iterations' costs, radii and decisions — and both end at a point of the same quality; the
    counters prove the path ran.  Case 1: 1 500 cameras / 90 k tracks, camera-side sweep forced into the chunked order (the
    library picks it above ~130 k tracks; the dot products with the recycled vectors ride on its k_gp_wsum) and the harvest
    threshold lowered to 8 iterations (solves of this size take ~12) — the mechanics on a small, stable problem.  Case 2: the
    same with the chunked order switched OFF (knob 2): the dot products come from k_cgr_dots_w, one more launch per iteration
    (what eight ranks on configs[3] run).  Case 3: configs[2] as the library runs it — solves of 25 ... 45 iterations in the
    middle of the trajectory, where it pays."""
    p = synthetic.make_gp_problem(num_cams=ncam, num_pts=npts, seed=4 if ncam < 5000 else 0)
    runs = {}
    gsfm_ctx.set_knob("chunked_sweeps", chunk_knob)
    gsfm_ctx.set_knob("gp_recycle_min_iters", min_iters)
    try:
        for off in (1, 0):
            gsfm_ctx.set_knob("gp_no_recycle", off)
            gsfm_ctx.stats(reset=True)
            rc, cen, _, rep = estimators.gp_solve(p, ctx=gsfm_ctx)
            assert rc == 0
            runs[off] = (cen, rep, gsfm_ctx.stats(reset=True), gsfm_ctx.lm_trace())
    finally:
        gsfm_ctx.set_knob("gp_no_recycle", 0)
        gsfm_ctx.set_knob("chunked_sweeps", 0)
        gsfm_ctx.set_knob("gp_recycle_min_iters", 0)
    (c0, r0, s0, t0), (c1, r1, s1, t1) = runs[1], runs[0]
    med = float(np.median(synthetic.center_errors_after_sim3(c1, c0)))
    print("[parity] gp recycle off/on %d cams: lm %d/%d pcg %d/%d recycled solves %d harvested %d cost %.9g/%.9g apart max %.2e median %.2e" % (
        ncam, r0["iterations"], r1["iterations"], r0["linear_iterations"], r1["linear_iterations"], s1["pcg_recycled"],
        s1["ritz_harvested"], r0["final_cost"], r1["final_cost"], _rel_diff(c1, c0), med))
    assert s0["pcg_recycled"] == 0 and s0["ritz_harvested"] == 0
    assert (s1["pcg_chunked_sweeps"] == s1["pcg_solves"] > 0) if chunk_knob != 2 else s1["pcg_chunked_sweeps"] == 0
    assert s1["pcg_recycled"] >= 5 and s1["ritz_harvested"] >= 5
    if min_iters == 0:
        assert r1["linear_iterations"] < 0.9 * r0["linear_iterations"]  # (measured 1 067 against 1 318)
    else:
        assert r1["linear_iterations"] <= r0["linear_iterations"]
    # until the first harvest the two runs are the same program; after it they solve the same systems to 1e-10
    n = min(10, len(t0), len(t1))
    assert np.allclose(t0[:n, 0], t1[:n, 0], rtol=1e-6) and np.allclose(t0[:n, 1], t1[:n, 1], rtol=1e-5)
    assert np.array_equal(t0[:n, 5], t1[:n, 5])
    assert abs(r1["final_cost"] - r0["final_cost"]) <= 1e-3 * r0["final_cost"]
    assert med < 1e-4
    assert _rel_diff(c1, p.gt_center) < 0.1 and _rel_diff(c0, p.gt_center) < 0.1


@pytest.mark.parametrize("ncam,npts,outliers", [(200, 12_000, 0.0), (800, 40_000, 0.0), (200, 12_000, 0.02)])
def test_gp_dense_reduced_system_on_a_capture_like_scene(gsfm_ctx, ncam, npts, outliers):
    """The dense direct path of the reduced camera system (gp.hip k_gp_dense_assemble + the block sweep of ra_dense.hpp): on a
    sequential capture — every point seen by a run of neighbouring cameras, the reduced system a long chain — block-Jacobi PCG
    needs hundreds of iterations per solve on a few hundred unknowns, and up to 1 024 cameras the library switches to assembling
    and inverting the system once a solve runs past 100 iterations.  Knob gp_dense: 1 = never (PCG all the way, as before), 2 =
    every solve dense, 0 = the shipped rule.  The same systems, solved exactly instead of to 1e-10: without outlier bearings the
    three runs make the same LM decisions and end at the same cameras; with the benchmark's 2 % outliers the trajectory is the
    chaotic one of DESIGN.md section 2 (two exact solvers end apart as well) and the end points are compared by their cost.
    The counters prove which path ran."""
    p = synthetic.make_gp_problem(num_cams=ncam, num_pts=npts, seed=3, capture="sequential", outlier_ratio=outliers)
    runs = {}
    try:
        for knob in (1, 2, 0):
            gsfm_ctx.set_knob("gp_dense", knob)
            gsfm_ctx.stats(reset=True)
            rc, cen, _, rep = estimators.gp_solve(p, ctx=gsfm_ctx)
            assert rc == 0
            runs[knob] = (cen, rep, gsfm_ctx.stats(reset=True), gsfm_ctx.lm_trace())
    finally:
        gsfm_ctx.set_knob("gp_dense", 0)
    (c1, r1, s1, t1), (c2, r2, s2, t2), (c0, r0, s0, t0) = runs[1], runs[2], runs[0]
    print("[parity] gp dense never/always/auto %d cams, outliers %.2f: lm %d/%d/%d pcg %d/%d/%d dense solves %d/%d/%d seconds %.3f/%.3f/%.3f cost %.9g/%.9g/%.9g "
          "always vs never %.2e, auto vs never %.2e" % (ncam, outliers, r1["iterations"], r2["iterations"], r0["iterations"], r1["linear_iterations"],
                                                         r2["linear_iterations"], r0["linear_iterations"], s1["dense_solves"], s2["dense_solves"],
                                                         s0["dense_solves"], r1["seconds_solve"], r2["seconds_solve"], r0["seconds_solve"],
                                                         r1["final_cost"], r2["final_cost"], r0["final_cost"], _rel_diff(c2, c1), _rel_diff(c0, c1)))
    assert s1["dense_solves"] == 0 and r1["linear_iterations"] > 100 * 10  # the scene IS chain-like for block-Jacobi
    assert s2["dense_solves"] >= r2["iterations"] - 1 and r2["linear_iterations"] == 0
    assert s0["dense_solves"] >= r0["iterations"] - 3 and r0["linear_iterations"] <= 3 * 100  # switched on by the first long solve
    if outliers == 0.0:
        n = min(8, len(t1), len(t2), len(t0))
        for t in (t2, t0):
            assert np.allclose(t1[:n, 0], t[:n, 0], rtol=1e-6) and np.array_equal(t1[:n, 5], t[:n, 5])
        for c, r in ((c2, r2), (c0, r0)):
            assert abs(r["final_cost"] - r1["final_cost"]) <= 1e-6 * r1["final_cost"]
            assert _rel_diff(c, c1) < 1e-5
    else:
        for r in (r2, r0):
            assert abs(r["final_cost"] - r1["final_cost"]) <= 1e-2 * r1["final_cost"]  # (the chaotic trajectory: three exact solvers, three end points)
    assert r0["seconds_solve"] < 0.5 * r1["seconds_solve"]
