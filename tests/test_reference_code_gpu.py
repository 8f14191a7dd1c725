"""The HIP path against REFERENCE CODE directly — no oracle in between.

oracle/_ref/ holds the reference's own rotation averaging and its global positioning / bundle adjustment problem builders,
compiled unmodified from /root/reference (in the build container; the libraries travel to the GPU box with the snapshot).  The
CPU tests hold the oracle to them (tests/test_oracle_ref*.py) and the GPU tests hold the HIP path to the oracle; these tests
close the triangle on the GPU box itself, same inputs on both sides:

  RotationEstimator::EstimateRotations (reference code on the CPU)  vs  gsfm_ra_solve: rotations and iteration counts, for
      trivial frames, cam_from_rig rotations among the unknowns, and gravity-aligned frames
  GlobalPositioner::Solve up to the first cost evaluation (reference code on a recording Ceres)  vs  gsfm_gp_solve: the
      initial cost of the reference's random start — draws in the reference's container walk, g++'s argument order
  BundleAdjuster::Solve up to the first cost evaluation  vs  gsfm_ba_solve: the initial cost, with the reference's constant frame
  GlobalPositioner::Solve / BundleAdjuster::Solve TO THEIR END POINTS (round 6: reference code on a SOLVING Ceres stand-in,
      oracle/ref_shim_solve/)  vs  gsfm_gp_solve / gsfm_ba_solve: final camera centres / poses and the LM trajectory"""
import numpy as np
import pytest

from glomap_amd import estimators, so3, synthetic
from glomap_amd.flat import BaProblem, GpProblem, RaProblem
from oracle import ref

pytestmark = [pytest.mark.gpu,
              pytest.mark.skipif(ref.load_ra() is None or ref.load_gp() is None or ref.load_ba() is None,
                                 reason="oracle/_ref: libraries not built (they come with the snapshot)")]


def _dist(q_a, q_b):
    d = so3.quat_mul(so3.quat_conj(np.atleast_2d(q_a)), np.atleast_2d(q_b))
    return 2.0 * np.arcsin(np.minimum(1.0, np.linalg.norm(d[:, 1:], axis=1)))


@pytest.mark.parametrize("N,deg,skip,use_weight", [(120, 6, 0, 0), (300, 8, 0, 1), (300, 8, 1, 0)])
def test_rotation_averaging_equals_the_reference_code(gsfm_ctx, N, deg, skip, use_weight):
    p = synthetic.make_ring_view_graph(N, deg, seed=11)
    E = len(p.edge_i)
    rng = np.random.default_rng(N)
    ninl = (rng.permutation(E) + 30).astype(np.int32)  # distinct: the spanning tree does not depend on tie-breaking
    w = rng.uniform(0.3, 1.0, E)
    r = ref.ra_estimate([0], np.zeros(N), np.arange(N), np.zeros(N), p.edge_i, p.edge_j, p.edge_q, pair_weight=w, pair_ninl=ninl,
                        frame_q=so3.aa_to_quat(p.node_aa0), skip_initialization=skip, use_weight=use_weight)
    assert r["ok"]
    # the reference roots its spanning tree at, and fixes, the first image its hash map yields; the flat problem's node 0 and
    # fixed_node play those roles: swap the labels 0 <-> that image
    f = r["fixed_image"]
    assert skip or r["tree_root"] == f
    lab = np.arange(N)
    lab[[0, f]] = lab[[f, 0]]
    aa0 = p.node_aa0[lab]  # node k of the GPU problem is image lab[k]
    inv = np.empty(N, np.int64)
    inv[lab] = np.arange(N)
    pg = RaProblem(N, inv[p.edge_i].astype(np.int32), inv[p.edge_j].astype(np.int32), p.edge_q, w, ninl, aa0, 0)
    opt = estimators.RotationEstimatorOptions(skip_initialization=bool(skip), use_weight=bool(use_weight))
    rc, rot, rep = estimators.ra_solve(pg, opt, ctx=gsfm_ctx)
    assert rc == 0
    assert (rep["iterations_l1"], rep["iterations_irls"]) == (r["l1_iterations"], r["irls_iterations"])
    d = _dist(so3.aa_to_quat(rot), r["frame_q"][lab])
    print(f"[parity] RA vs REFERENCE CODE N={N} skip={skip} weight={use_weight}: L1 {rep['iterations_l1']} IRLS {rep['iterations_irls']} "
          f"max {d.max():.2e} rad")
    assert d.max() < 1e-6


def test_rig_rotation_averaging_equals_the_reference_code(gsfm_ctx):
    from test_oracle_ref_ra import _rig_scene

    s = _rig_scene(20, 3, 2, 0.5, 0.05)
    rng = np.random.default_rng(3)
    aa_f = so3.quat_to_aa(so3.rotmat_to_quat(so3.aa_to_rotmat(rng.normal(0, 0.05, (s["N"], 3))) @ s["R_f"]))
    aa_c = so3.quat_to_aa(so3.rotmat_to_quat(so3.aa_to_rotmat(rng.normal(0, 0.05, (s["C"], 3))) @ s["R_c"]))
    kw = dict(max_num_l1_iterations=3, max_num_irls_iterations=6, l1_step_convergence_threshold=0.0, irls_step_convergence_threshold=0.0)
    r = ref.ra_estimate(s["rig_ref_cam"], s["frame_rig"], s["imf"], s["image_camera"], s["ii"], s["jj"], s["q"], pair_ninl=s["ninl"],
                        sensor_rig=s["sensor_rig"], sensor_cam=s["sensor_cam"], sensor_state=np.full(s["C"], 2),
                        sensor_q=so3.aa_to_quat(aa_c), frame_q=so3.aa_to_quat(aa_f), skip_initialization=1, **kw)
    assert r["ok"] and (r["l1_iterations"], r["irls_iterations"]) == (3, 6)
    # the gauge: the frame of the reference's fixed image -> frame 0 of the flat problem
    ff = int(s["imf"][r["fixed_image"]])
    lab = np.arange(s["N"])
    lab[[0, ff]] = lab[[ff, 0]]
    inv = np.empty(s["N"], np.int64)
    inv[lab] = np.arange(s["N"])
    rc_, rot, cam, rep = estimators.ra_solve_rigs(s["N"], inv[s["imf"]].astype(np.int32), s["imc"], s["C"], s["ii"], s["jj"], s["q"], s["ninl"],
                                                   options=estimators.RotationEstimatorOptions(skip_initialization=True, **kw), ctx=gsfm_ctx,
                                                   frame_aa0=aa_f[lab], cam_aa0=aa_c)
    assert rc_ == 0 and (rep["iterations_l1"], rep["iterations_irls"]) == (3, 6)
    df, dc = _dist(so3.aa_to_quat(rot), r["frame_q"][lab]), _dist(so3.aa_to_quat(cam), r["sensor_q"])
    print(f"[parity] rig RA vs REFERENCE CODE: frames {df.max():.2e} cams {dc.max():.2e} rad")
    assert df.max() < 1e-6 and dc.max() < 1e-6


def test_gravity_rotation_averaging_equals_the_reference_code(gsfm_ctx):
    from test_ra_gravity import make_gravity_graph

    p = make_gravity_graph(60, 6, seed=3, frac=0.6)
    N = p.num_nodes
    g = p.node_gravity.astype(bool)
    Ra = np.tile(np.eye(3), (N, 1, 1))
    Ra[~g] = np.nan
    r = ref.ra_estimate([0], np.zeros(N), np.arange(N), np.zeros(N), p.edge_i, p.edge_j, p.edge_q, frame_q=so3.aa_to_quat(p.node_aa0),
                        frame_R_align=Ra, use_gravity=1)
    assert r["ok"]
    p.fixed_node = r["fixed_image"]
    rc, rot, rep = estimators.ra_solve(p, estimators.RotationEstimatorOptions(use_gravity=True), ctx=gsfm_ctx)
    assert rc == 0 and (rep["iterations_l1"], rep["iterations_irls"]) == (r["l1_iterations"], r["irls_iterations"])
    d = _dist(so3.aa_to_quat(rot), r["frame_q"])
    print(f"[parity] gravity RA vs REFERENCE CODE: max {d.max():.2e} rad")
    assert d.max() < 1e-6


@pytest.mark.parametrize("seed", [0, 1])
def test_global_positioning_start_equals_the_reference_code(gsfm_ctx, seed):
    """The reference's GlobalPositioner::Solve builds its problem from ITS random start (std::mt19937, draws in the walk order of
    its hash maps, g++ argument order) and a recording Ceres evaluates the cost there; the HIP path, given the two walk orders and
    rand_vector_order = 1, reports the same initial cost."""
    p = synthetic.make_gp_problem(num_cams=60, num_pts=2000, seed=seed, uncalibrated_ratio=0.2, dir_noise=1e-3, outlier_ratio=0.02)
    q = so3.rotmat_to_quat(p.cam_R)
    t = -np.einsum("nij,nj->ni", p.cam_R, p.gt_center)
    und = np.einsum("mij,mj->mi", p.cam_R[p.obs_cam], p.obs_dir)
    cal = np.ones(p.num_cams, np.uint8)
    cal[p.obs_cam] = p.obs_calibrated
    r = ref.gp_build(q, t, p.pt_offset, p.obs_cam, und, p.pt_xyz, cam_calibrated=cal)
    pg = GpProblem(num_cams=p.num_cams, num_pts=p.num_pts, pt_offset=p.pt_offset, obs_cam=p.obs_cam, obs_dir=p.obs_dir,
                   obs_calibrated=p.obs_calibrated, cam_center=p.gt_center.copy(), pt_xyz=p.pt_xyz.copy(),
                   cam_draw_order=r["frame_order"].astype(np.int32), pt_draw_order=r["track_order"].astype(np.int32))
    rc, c, X, rep = estimators.gp_solve(pg, estimators.GlobalPositionerOptions(rand_vector_order=1), ctx=gsfm_ctx)
    assert rc == 0
    rel = abs(rep["initial_cost"] - r["initial_cost"]) / r["initial_cost"]
    print(f"[parity] GP start vs REFERENCE CODE seed={seed}: initial cost {rep['initial_cost']:.12e} vs {r['initial_cost']:.12e} (rel {rel:.1e})")
    assert rel < 1e-12
    assert synthetic.center_errors_after_sim3(c, p.gt_center).max() < 0.1  # and the solve from that start recovers the scene


@pytest.mark.skipif(ref.load_gp_solve() is None, reason="oracle/_ref/libref_glomap_gp_solve.so not built (it comes with the snapshot)")
@pytest.mark.parametrize("N,P,seed,uncal", [(40, 800, 1, 0.2), (60, 2000, 2, 0.2), (60, 2000, 3, 0.0), (150, 6000, 0, 0.0)])
def test_global_positioning_end_point_equals_the_reference_code(gsfm_ctx, N, P, seed, uncal):
    """GlobalPositioner::Solve of the reference run TO ITS END POINT (global_positioning.cc compiled unmodified on the solving
    Ceres stand-in of oracle/ref_shim_solve/: the reference's own BATA functors differentiated by dual numbers, its losses, its
    bounds and constant blocks, its ordering; the trust-region loop with the projected line search restated from Ceres' sources
    — a third writing that shares no code with the oracle's or the product's) against gsfm_gp_solve from the same random start
    (the reference's container walk, g++ argument order): FINAL camera centres, LM iteration for LM iteration."""
    p = synthetic.make_gp_problem(num_cams=N, num_pts=P, seed=seed, uncalibrated_ratio=uncal)
    q = so3.rotmat_to_quat(p.cam_R)
    t = -np.einsum("nij,nj->ni", p.cam_R, p.gt_center)
    und = np.einsum("mij,mj->mi", p.cam_R[p.obs_cam], p.obs_dir)
    cal = np.ones(p.num_cams, np.uint8)
    cal[p.obs_cam] = p.obs_calibrated
    r = ref.gp_solve(q, t, p.pt_offset, p.obs_cam, und, p.pt_xyz, cam_calibrated=cal)
    assert r["ok"] and r["constrained"] and r["termination"] == 0
    pg = GpProblem(num_cams=p.num_cams, num_pts=p.num_pts, pt_offset=p.pt_offset, obs_cam=p.obs_cam, obs_dir=p.obs_dir,
                   obs_calibrated=p.obs_calibrated, cam_center=p.gt_center.copy(), pt_xyz=p.pt_xyz.copy(),
                   cam_draw_order=r["frame_order"].astype(np.int32), pt_draw_order=r["track_order"].astype(np.int32))
    rc, c, X, rep = estimators.gp_solve(pg, estimators.GlobalPositionerOptions(rand_vector_order=1), ctx=gsfm_ctx)
    assert rc == 0
    tr, tr_r = gsfm_ctx.lm_trace(), r["trace"]
    n = min(len(tr), len(tr_r))
    bad = np.abs(tr[:n, 0] - tr_r[:n, 0]) > 1e-6 * np.abs(tr_r[:n, 0])
    same = int(np.argmax(bad)) if bad.any() else n
    st = synthetic.center_distance_stats(c, r["center"])
    raw = np.abs(c - r["center"]).max() / synthetic.scene_extent(r["center"])  # same start, same gauge: no alignment needed
    print(f"[parity] GP END POINT vs REFERENCE CODE {N} cameras / {P} tracks, seed {seed}: LM {rep['iterations']} ({rep['successful_steps']} accepted, "
          f"{rep['line_search_shrunk']} shortened) vs {r['iterations']} ({r['successful_steps']}, {r['line_search_shrunk']}), initial cost rel "
          f"{abs(rep['initial_cost'] - r['initial_cost']) / r['initial_cost']:.1e}, final cost {rep['final_cost']:.9f} vs {r['final_cost']:.9f}, same cost "
          f"to 1e-6 for the first {same} LM iterations, centres / extent: max {st['max']:.3e} p99 {st['p99']:.3e} median {st['median']:.3e} "
          f"(without alignment: {raw:.3e})")
    assert abs(rep["initial_cost"] - r["initial_cost"]) <= 1e-12 * r["initial_cost"]
    assert same >= min(10, n)
    assert abs(rep["iterations"] - r["iterations"]) <= 2 and abs(rep["successful_steps"] - r["successful_steps"]) <= 2
    assert abs(rep["final_cost"] - r["final_cost"]) <= 1e-4 * r["final_cost"]
    assert st["max"] < 1e-3  # north_star's bar on the worst camera


def test_bundle_adjustment_start_equals_the_reference_code(gsfm_ctx):
    p = synthetic.make_ba_problem(num_cams=40, num_pts=1500, seed=2, pixel_noise=0.7, outlier_ratio=0.02, intr_noise=0.01)
    r = ref.ba_build(p.intr_model, p.intr_params, p.cam_q, p.cam_t, np.arange(p.num_cams), p.cam_intr, p.pt_offset, p.obs_cam, p.obs_xy, p.pt_xyz,
                     rig_ref_cam=np.arange(p.num_intr), frame_rig=p.cam_intr)
    order = r["frame_order"]
    p.fixed_cam = int(order[(r["frame_flags"][order] & 1) != 0][0])  # the constant frame of the reference (ba.cc:252-270)
    rc, q, t, X, intr, rep = estimators.ba_solve(p, ctx=gsfm_ctx)
    assert rc == 0
    rel = abs(rep["initial_cost"] - r["initial_cost"]) / r["initial_cost"]
    print(f"[parity] BA start vs REFERENCE CODE: initial cost {rep['initial_cost']:.12e} vs {r['initial_cost']:.12e} (rel {rel:.1e})")
    assert rel < 1e-12
    assert np.array_equal(q[p.fixed_cam], p.cam_q[p.fixed_cam]) and np.array_equal(t[p.fixed_cam], p.cam_t[p.fixed_cam])


@pytest.mark.skipif(ref.load_ba_solve() is None, reason="oracle/_ref/libref_glomap_ba_solve.so not built (it comes with the snapshot)")
@pytest.mark.parametrize("N,P,seed,kw", [(15, 300, 23, dict(shared_intrinsics=True, intr_noise=0.01)),
                                         (40, 1500, 2, dict(pixel_noise=0.7, outlier_ratio=0.02, intr_noise=0.01)),
                                         (60, 3000, 5, dict(pixel_noise=0.5, outlier_ratio=0.02)),
                                         (120, 8000, 7, dict(pixel_noise=0.5, outlier_ratio=0.01, intr_noise=0.005))])
def test_bundle_adjustment_end_point_equals_the_reference_code(gsfm_ctx, N, P, seed, kw):
    """BundleAdjuster::Solve of the reference run TO ITS END POINT (bundle_adjustment.cc compiled unmodified on the solving Ceres
    stand-in, oracle/ref_shim_solve/) against gsfm_ba_solve with the reference's constant frame: FINAL rotations (bar 1e-4 rad)
    and camera centres (bar 1e-3 of the extent; no alignment, the constant frame fixes the gauge), equal LM iteration counts."""
    p = synthetic.make_ba_problem(num_cams=N, num_pts=P, seed=seed, **kw)
    r = ref.ba_solve(p.intr_model, p.intr_params, p.cam_q, p.cam_t, np.arange(p.num_cams), p.cam_intr, p.pt_offset, p.obs_cam, p.obs_xy, p.pt_xyz,
                     rig_ref_cam=np.arange(p.num_intr), frame_rig=p.cam_intr)
    assert r["ok"] and r["frame_const"].sum() == 1
    p.fixed_cam = int(np.nonzero(r["frame_const"])[0][0])
    rc, q, t, X, intr, rep = estimators.ba_solve(p, ctx=gsfm_ctx)
    assert rc == 0
    ang = _dist(q, r["frame_q"])
    Rg, Rr = so3.quat_to_rotmat(q), so3.quat_to_rotmat(r["frame_q"])
    cg, cr = -np.einsum("nji,nj->ni", Rg, t), -np.einsum("nji,nj->ni", Rr, r["frame_t"])
    dc_raw = np.linalg.norm(cg - cr, axis=1).max() / synthetic.scene_extent(cr)
    # one constant frame fixes six of the seven gauge freedoms: the SCALE is held by the LM damping alone, and where the steps
    # creep along it two solvers that stop their linear solves differently (the library: PCG to 1e-6)
    # end at slightly different scales — compared after Sim(3) alignment, as north_star's translation bar is, next to the raw figure
    dc = synthetic.center_distance_stats(cg, cr)["max"]
    print(f"[parity] BA END POINT vs REFERENCE CODE {N} cameras / {P} tracks: LM {rep['iterations']} ({rep['successful_steps']} accepted) vs "
          f"{r['iterations']} ({r['successful_steps']}), final cost {rep['final_cost']:.9f} vs {r['final_cost']:.9f}, rotations max {ang.max():.3e} rad "
          f"(bar 1e-4), centres / extent max {dc:.3e} after Sim(3) (bar 1e-3; {dc_raw:.3e} without alignment), focal max "
          f"{np.abs(intr[:, 0] - r['cam_params'][:, 0]).max():.3e}")
    assert abs(rep["initial_cost"] - r["initial_cost"]) <= 1e-12 * r["initial_cost"]
    assert abs(rep["iterations"] - r["iterations"]) <= 1 and abs(rep["final_cost"] - r["final_cost"]) <= 1e-5 * r["final_cost"]
    assert ang.max() < 1e-4 and dc < 1e-3 and dc_raw < 5e-3


# ---------------------------------------------------------------------------------------------------------------
# BASELINE sizes, against results the reference code produced in the build container (tests/golden/make_reference_code_golden.py)
# ---------------------------------------------------------------------------------------------------------------
def _golden(name):
    from pathlib import Path

    return np.load(Path(__file__).resolve().parent / "golden" / name)


def test_ra_config2_equals_the_reference_code(gsfm_ctx):
    """BASELINE configs[1] (1 000 cameras / 50 000 relative rotations): the HIP solve against the rotations the reference's own
    RotationEstimator::EstimateRotations returned for the same view graph."""
    g = _golden("ra_c2_reference_code.npz")
    p = synthetic.make_ring_view_graph(1000, 50, seed=0)
    N = p.num_nodes
    f = int(g["fixed_image"])
    assert int(g["tree_root"]) == f
    lab = np.arange(N)
    lab[[0, f]] = lab[[f, 0]]
    inv = np.empty(N, np.int64)
    inv[lab] = np.arange(N)
    pg = RaProblem(N, inv[p.edge_i].astype(np.int32), inv[p.edge_j].astype(np.int32), p.edge_q, p.edge_weight, p.edge_ninl, p.node_aa0[lab], 0)
    rc, rot, rep = estimators.ra_solve(pg, ctx=gsfm_ctx)
    assert rc == 0
    assert (rep["iterations_l1"], rep["iterations_irls"]) == (int(g["l1_iterations"]), int(g["irls_iterations"]))
    d = _dist(so3.aa_to_quat(rot), g["frame_q"][lab])
    print(f"[parity] RA configs[1] vs REFERENCE CODE: L1 {rep['iterations_l1']} IRLS {rep['iterations_irls']}, max {d.max():.2e} rad (bar 1e-4)")
    assert d.max() < 1e-6


def test_ra_config4_equals_the_reference_code(gsfm_ctx):
    """BASELINE configs[3]'s view graph (10 000 cameras / 500 000 relative rotations) — the size at which the HIP path runs its
    substructured PCG instead of the dense direct solve: against the rotations the reference's own
    RotationEstimator::EstimateRotations returned (global_rotation_averaging.cc:543-625 on 30 000-unknown systems; round 6: the
    CHOLMOD stand-in factors an envelope, tests/golden/make_reference_code_golden.py ra_c4, 17 s on one thread)."""
    g = _golden("ra_c4_reference_code.npz")
    p = synthetic.make_ring_view_graph(10_000, 50, seed=0)
    N = p.num_nodes
    f = int(g["fixed_image"])
    assert int(g["tree_root"]) == f
    lab = np.arange(N)
    lab[[0, f]] = lab[[f, 0]]
    inv = np.empty(N, np.int64)
    inv[lab] = np.arange(N)
    pg = RaProblem(N, inv[p.edge_i].astype(np.int32), inv[p.edge_j].astype(np.int32), p.edge_q, p.edge_weight, p.edge_ninl, p.node_aa0[lab], 0)
    rc, rot, rep = estimators.ra_solve(pg, ctx=gsfm_ctx)
    assert rc == 0
    assert (rep["iterations_l1"], rep["iterations_irls"]) == (int(g["l1_iterations"]), int(g["irls_iterations"]))
    d = _dist(so3.aa_to_quat(rot), g["frame_q"][lab])
    print(f"[parity] RA configs[3] vs REFERENCE CODE: L1 {rep['iterations_l1']} IRLS {rep['iterations_irls']} both, max {d.max():.2e} rad (bar 1e-4); "
          f"reference code took {float(g['seconds_one_thread']):.1f} s on one thread of the build container")
    assert d.max() < 1e-6


def test_gp_config3_start_equals_the_reference_code(gsfm_ctx):
    """BASELINE configs[2] size: the cost of the reference's random start, 3.0 M BATA residuals summed by reference code on the
    recording Ceres, against the HIP path's initial cost for the same draw orders."""
    g = _golden("gp_start_reference_code.npz")
    p = synthetic.make_gp_problem(5000, 500_000, seed=0)
    p.cam_draw_order, p.pt_draw_order = g["frame_order"], g["track_order"]
    rc, c, X, rep = estimators.gp_solve(p, estimators.GlobalPositionerOptions(rand_vector_order=1), ctx=gsfm_ctx)
    assert rc == 0
    rel = abs(rep["initial_cost"] - float(g["initial_cost"])) / float(g["initial_cost"])
    print(f"[parity] GP configs[2] start vs REFERENCE CODE: initial cost {rep['initial_cost']:.12e} vs {float(g['initial_cost']):.12e} (rel {rel:.1e})")
    assert rel < 1e-11
    assert synthetic.center_errors_after_sim3(c, p.gt_center).max() < 0.1


def test_ba_config4_start_equals_the_reference_code(gsfm_ctx):
    """BASELINE configs[3] size: 5.0 M reprojection residuals at the start point, reference builder vs HIP path, with the frame
    the reference holds constant."""
    g = _golden("ba_start_reference_code.npz")
    p = synthetic.make_ba_problem(10_000, 1_000_000, seed=0, shared_intrinsics=False)
    p.fixed_cam = int(g["fixed_frame"])
    opt = estimators.BundleAdjusterOptions()
    opt.solver_options.max_num_iterations = 2  # the start is what is compared
    rc, q, t, X, intr, rep = estimators.ba_solve(p, opt, ctx=gsfm_ctx)
    assert rc == 0
    rel = abs(rep["initial_cost"] - float(g["initial_cost"])) / float(g["initial_cost"])
    print(f"[parity] BA configs[3] start vs REFERENCE CODE: initial cost {rep['initial_cost']:.12e} vs {float(g['initial_cost']):.12e} (rel {rel:.1e})")
    lens = np.diff(p.pt_offset)
    assert rel < 1e-11 and int(g["num_residual_blocks"]) == int(lens[lens >= 3].sum())  # ba.cc:122
