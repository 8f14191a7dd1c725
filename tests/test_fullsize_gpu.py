"""BASELINE.json's full-size configurations through size-independent properties (the CPU oracle cannot
finish these sizes in test time): convergence, ground-truth recovery at the synthetic noise level,
monotone cost, idempotence (a second solve started from the solution stops at once and does not move
it), and agreement of the two RA linear solvers."""
import numpy as np
import pytest

from glomap_amd import estimators, so3, synthetic

pytestmark = pytest.mark.gpu


def test_ra_config2_full(gsfm_ctx):
    """configs[1]: 1k cameras / 50k edges, 5 % outlier edges, 1 degree noise."""
    p = synthetic.make_ring_view_graph(1000, 50, seed=0)
    rc, rot, rep = estimators.ra_solve(p, ctx=gsfm_ctx)
    assert rc == 0 and rep["iterations_l1"] >= 1 and rep["iterations_irls"] >= 1
    err = synthetic.rotation_errors_deg(so3.aa_to_rotmat(rot), p.gt_R)
    assert np.median(err) < 0.5 and err.max() < 3.0  # rotation_averager_test.cc:309-310 allows 3 degrees
    # dense direct solver (N <= 2048) and the PCG solver large graphs use agree
    rc, rot_it, _ = estimators.ra_solve(p, estimators.RotationEstimatorOptions(force_iterative=True), ctx=gsfm_ctx)
    assert rc == 0
    d = np.radians(so3.rotation_angle_deg(so3.aa_to_rotmat(rot), so3.aa_to_rotmat(rot_it)))
    assert d.max() < 1e-4
    # idempotence: restarted from the solution, IRLS stops after its first step (mean step < 1e-3 rad)
    p2 = type(p)(**{**p.__dict__, "node_aa0": rot})
    rc, rot2, rep2 = estimators.ra_solve(
        p2, estimators.RotationEstimatorOptions(skip_initialization=True, max_num_l1_iterations=0), ctx=gsfm_ctx)
    assert rc == 0 and rep2["iterations_irls"] == 1
    d = np.radians(so3.rotation_angle_deg(so3.aa_to_rotmat(rot2), so3.aa_to_rotmat(rot)))
    assert np.mean(d) < 1e-3


def test_gp_config3_full(gsfm_ctx):
    """configs[2]: 5k cameras / 500k tracks / ~3M observations, random initialisation (seed 1)."""
    p = synthetic.make_gp_problem(5000, 500_000, seed=0)
    rc, cen, xyz, rep = estimators.gp_solve(p, ctx=gsfm_ctx)
    assert rc == 0 and rep["termination"] == 0
    assert rep["final_cost"] < 1e-3 * rep["initial_cost"]
    err = synthetic.center_errors_after_sim3(cen, p.gt_center)
    extent = np.linalg.norm(p.gt_center - p.gt_center.mean(0), axis=1).max()
    assert np.median(err) / extent < 1e-3  # ray noise 1e-3
    # idempotence: from the solution (no random re-draw; the per-observation scales are re-derived from the
    # geometry, gp.cc:300-305, so this is not a bit-exact restart) the solver stops quickly at the same cost
    p2 = type(p)(**{**p.__dict__, "cam_center": cen, "pt_xyz": xyz})
    opt = estimators.GlobalPositionerOptions(generate_random_positions=False, generate_random_points=False,
                                            generate_scales=False)
    rc, cen2, xyz2, rep2 = estimators.gp_solve(p2, opt, ctx=gsfm_ctx)
    assert rc == 0 and rep2["iterations"] <= 12
    assert rep2["final_cost"] <= rep["final_cost"] * (1 + 1e-4)
    assert synthetic.center_errors_after_sim3(cen2, cen).max() / extent < 1e-3


def test_ba_config4_full(gsfm_ctx):
    """configs[3] on one GPU: 10k cameras / 1M tracks / ~5M observations, one SIMPLE_RADIAL camera per image."""
    p = synthetic.make_ba_problem(10_000, 1_000_000, seed=0, shared_intrinsics=False)
    rc, q, t, X, intr, rep = estimators.ba_solve(p, ctx=gsfm_ctx)
    assert rc == 0 and rep["termination"] == 0
    assert rep["final_cost"] < 0.5 * rep["initial_cost"]
    rot_err = synthetic.rotation_errors_deg(so3.quat_to_rotmat(q), so3.quat_to_rotmat(p.gt_q))
    assert np.median(rot_err) < 0.05  # start: 0.5 degree noise per camera
    # the constant frame is untouched (ba.cc:261-266)
    assert np.array_equal(q[p.fixed_cam], p.cam_q[p.fixed_cam]) and np.array_equal(t[p.fixed_cam], p.cam_t[p.fixed_cam])
    # restart from the solution: the cost is reproduced exactly, never goes up, and only creeps further
    # (the first solve stopped on function_tolerance 1e-5; a fresh trust region makes a little more progress)
    p2 = type(p)(**{**p.__dict__, "cam_q": q, "cam_t": t, "pt_xyz": X, "intr_params": intr})
    rc, q2, t2, X2, intr2, rep2 = estimators.ba_solve(p2, ctx=gsfm_ctx)
    assert rc == 0
    assert abs(rep2["initial_cost"] - rep["final_cost"]) <= 1e-9 * rep["final_cost"]
    assert rep2["final_cost"] <= rep["final_cost"] * (1 + 1e-9)
    assert rep2["final_cost"] >= 0.97 * rep["final_cost"]
