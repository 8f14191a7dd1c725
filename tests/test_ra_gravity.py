"""Gravity-aligned rotation averaging, use_gravity = true (global_rotation_averaging.cc:19-36, 207-217, 312-341, 376-418,
455-460, 639-645, 709-713, 746-749): frames with gravity have ONE unknown (the angle about the aligned vertical), pairs
of two such frames one row.  The reference exercises this through rotation_averager_test.cc:171-212, 314-364
(`for use_gravity : {true, false}`), pinning recovered rotations.

CPU: the oracle builds the reference's mixed 1-DoF / 3-DoF matrix row by row.  GPU: the HIP path keeps its scalar
Laplacian machinery and masks the x / z components of gravity frames — two independent formulations of the same system."""
import numpy as np
import pytest

from glomap_amd import so3, synthetic
from glomap_amd.flat import RaProblem
from oracle import ra as ora


def make_gravity_graph(N=40, deg=6, seed=0, frac=0.6, noise_deg=0.5, outlier=0.05, start_sigma=0.2):
    """Ring-like view graph; a fraction of the frames is gravity aligned (rotation about y only, R_align = I, so the
    relative rotations are already 'aligned'), the others arbitrary.  Start: ground truth perturbed by `start_sigma` rad
    (no spanning-tree start exists in this mode)."""
    rng = np.random.default_rng(seed)
    grav = rng.random(N) < frac
    R = so3.aa_to_rotmat(rng.normal(0, 0.8, (N, 3)))
    ang = rng.uniform(-3.0, 3.0, N)
    Ry = so3.aa_to_rotmat(np.stack([np.zeros(N), ang, np.zeros(N)], 1))
    R[grav] = Ry[grav]
    ei = np.concatenate([np.arange(N) for _ in range(deg)]).astype(np.int32)
    ej = np.concatenate([(np.arange(N) + k) % N for k in range(1, deg + 1)]).astype(np.int32)
    Rrel = R[ej] @ np.transpose(R[ei], (0, 2, 1))
    if noise_deg:
        Rrel = so3.aa_to_rotmat(rng.normal(0, np.radians(noise_deg), (ei.size, 3))) @ Rrel
    o = rng.random(ei.size) < outlier
    if o.any():
        Rrel[o] = so3.aa_to_rotmat(rng.normal(0, 1.0, (int(o.sum()), 3)))
    aa0 = so3.quat_to_aa(so3.rotmat_to_quat(so3.aa_to_rotmat(rng.normal(0, start_sigma, (N, 3))) @ R))
    aa0[grav] = np.stack([np.zeros(N), ang + rng.normal(0, start_sigma, N), np.zeros(N)], 1)[grav]
    p = RaProblem(N, ei, ej, so3.rotmat_to_quat(Rrel), np.ones(ei.size), np.ones(ei.size, np.int32), aa0, 0, gt_R=R,
                  node_gravity=grav.astype(np.uint8))
    return p


def _oracle(p, opt=None, trace=None):
    return ora.estimate_rotations_gravity(p.num_nodes, p.edge_i, p.edge_j, p.edge_q, p.edge_weight, p.node_gravity, p.node_aa0,
                                          p.fixed_node, opt, trace)


def test_oracle_recovers_mixed_gravity_scene():
    p = make_gravity_graph()
    ok, rot = _oracle(p)
    assert ok
    assert synthetic.rotation_errors_deg(so3.aa_to_rotmat(rot), p.gt_R).max() < 1.5  # 0.5 degree noise, 5 % outliers
    g = p.node_gravity.astype(bool)
    assert np.abs(rot[g][:, [0, 2]]).max() == 0.0  # a gravity frame only ever moves about the vertical
    p0 = make_gravity_graph(noise_deg=0.0, outlier=0.0)
    ok, rot = _oracle(p0)
    assert ok and synthetic.rotation_errors_deg(so3.aa_to_rotmat(rot), p0.gt_R).max() < 1e-2


def test_without_gravity_frames_the_oracle_is_the_plain_one():
    p = make_gravity_graph(frac=0.0)
    ok, a = _oracle(p)
    ok2, b = ora.estimate_rotations(p.num_nodes, p.edge_i, p.edge_j, p.edge_q, p.edge_weight, p.edge_ninl, p.node_aa0, 0,
                                    ora.RotationEstimatorOptions(skip_initialization=True))
    assert ok and ok2 and np.array_equal(a, b)


@pytest.mark.gpu
@pytest.mark.parametrize("N,deg,frac,fixed_grav", [(40, 6, 0.6, True), (40, 6, 0.6, False), (300, 10, 0.3, True), (120, 8, 1.0, True)])
def test_gravity_rotation_averaging_matches_oracle(gsfm_ctx, N, deg, frac, fixed_grav):
    from glomap_amd import estimators

    p = make_gravity_graph(N, deg, seed=3, frac=frac)
    g = p.node_gravity.astype(bool)
    cand = np.nonzero(g == fixed_grav)[0]
    p.fixed_node = int(cand[0]) if cand.size else 0  # one gauge row (gravity frame) or three
    tr = ora.RaTrace()
    ok, rot_o = _oracle(p, trace=tr)
    assert ok
    rc, rot, rep = estimators.ra_solve(p, estimators.RotationEstimatorOptions(use_gravity=True), ctx=gsfm_ctx)
    assert rc == 0
    assert rep["iterations_l1"] == tr.l1_iterations and rep["iterations_irls"] == tr.irls_iterations
    assert np.radians(so3.rotation_angle_deg(so3.aa_to_rotmat(rot), so3.aa_to_rotmat(rot_o))).max() < 1e-6
    assert np.abs(rot[g][:, [0, 2]]).max() == 0.0
    assert synthetic.rotation_errors_deg(so3.aa_to_rotmat(rot), p.gt_R).max() < 2.0


@pytest.mark.gpu
def test_use_gravity_without_gravity_frames_is_the_plain_solve_without_spanning_tree(gsfm_ctx):
    from glomap_amd import estimators

    p = make_gravity_graph(60, 6, seed=5, frac=0.0)
    a = estimators.ra_solve(p, estimators.RotationEstimatorOptions(use_gravity=True), ctx=gsfm_ctx)
    p.node_gravity = None
    b = estimators.ra_solve(p, estimators.RotationEstimatorOptions(skip_initialization=True, force_iterative=True), ctx=gsfm_ctx)
    assert a[0] == 0 and b[0] == 0 and np.abs(a[1] - b[1]).max() < 1e-9
