"""Oracle (CPU restatement) of rotation averaging against the reference's own pins: ground-truth
recovery tolerances of glomap/controllers/rotation_averager_test.cc on synthetic scenes."""
import numpy as np
import pytest

from glomap_amd import so3, synthetic
from oracle import ra
from oracle import so3 as oso3


def _solve(p, **kw):
    opt = ra.RotationEstimatorOptions(**kw)
    tr = ra.RaTrace()
    ok, rot = ra.estimate_rotations(
        p.num_nodes, p.edge_i, p.edge_j, p.edge_q, p.edge_weight, p.edge_ninl, p.node_aa0, p.fixed_node, opt, tr
    )
    assert ok
    return rot, tr


def test_exp_log_roundtrip():
    rng = np.random.default_rng(0)
    a = rng.normal(size=(2000, 3))
    a *= (rng.uniform(0, np.pi - 1e-3, (2000, 1))) / np.linalg.norm(a, axis=1, keepdims=True)
    back = oso3.log_rot(oso3.exp_aa(a))
    assert np.abs(back - a).max() < 1e-9
    # small-angle branch of AngleAxisToRotation (glomap/math/rigid3d.cc:50-62)
    tiny = np.array([[1e-13, -2e-13, 3e-14]])
    R = oso3.exp_aa(tiny)
    assert R[0, 0, 0] == 1.0 and R[0, 1, 0] == tiny[0, 2] and R[0, 0, 1] == -tiny[0, 2]
    # near-pi rotation goes through the largest-diagonal branch of the matrix->quaternion step
    a_pi = np.array([[np.pi - 1e-6, 0.0, 0.0], [0.0, 0.0, -(np.pi - 1e-7)]])
    assert np.abs(oso3.log_rot(oso3.exp_aa(a_pi)) - a_pi).max() < 1e-6


def test_agrees_with_independent_so3():
    rng = np.random.default_rng(1)
    a = rng.normal(size=(500, 3))
    assert np.abs(oso3.exp_aa(a) - so3.aa_to_rotmat(a)).max() < 1e-12


def test_without_noise():
    # rotation_averager_test.cc:126-169 — noise-free scene, tolerance 1e-2 degrees
    p = synthetic.make_ring_view_graph(60, 8, noise_deg=0.0, outlier_ratio=0.0, seed=3)
    rot, tr = _solve(p)
    err = synthetic.rotation_errors_deg(so3.aa_to_rotmat(rot), p.gt_R)
    assert err.max() < 1e-2


def test_with_noise_and_outliers():
    # rotation_averager_test.cc:265-312 — noisy scene with outliers, tolerance 3 degrees
    p = synthetic.make_ring_view_graph(120, 15, noise_deg=1.0, outlier_ratio=0.1, seed=5)
    rot, tr = _solve(p)
    err = synthetic.rotation_errors_deg(so3.aa_to_rotmat(rot), p.gt_R)
    assert err.max() < 3.0
    assert tr.irls_iterations >= 1 and tr.l1_iterations >= 1


def test_skip_initialization_and_weights():
    p = synthetic.make_ring_view_graph(80, 10, noise_deg=0.5, outlier_ratio=0.05, seed=7, init="gt_noisy")
    p.edge_weight = np.random.default_rng(0).uniform(0.5, 1.0, p.num_edges)
    rot, _ = _solve(p, skip_initialization=True, use_weight=True)
    err = synthetic.rotation_errors_deg(so3.aa_to_rotmat(rot), p.gt_R)
    assert err.max() < 2.0


def test_half_norm_weights():
    p = synthetic.make_ring_view_graph(80, 10, noise_deg=0.5, outlier_ratio=0.05, seed=9)
    rot, _ = _solve(p, weight_type=ra.HALF_NORM)
    err = synthetic.rotation_errors_deg(so3.aa_to_rotmat(rot), p.gt_R)
    assert err.max() < 3.0


def test_mst_root_is_identity():
    # gra.cc:120: the root is skipped, so cam_from_worlds[root] stays the default (identity) pose
    p = synthetic.make_ring_view_graph(30, 4, noise_deg=0.0, outlier_ratio=0.0, seed=1)
    R = oso3.quat_wxyz_to_rotmat(p.edge_q)
    aa = ra.maximum_spanning_tree_init(p.num_nodes, p.edge_i.astype(np.int64), p.edge_j.astype(np.int64), R, p.edge_ninl, p.node_aa0 + 0.3)
    assert np.abs(aa[0]).max() == 0.0
    # noise-free: tree propagation already reproduces all relative rotations
    Rn = oso3.exp_aa(aa)
    rel = Rn[p.edge_j] @ np.transpose(Rn[p.edge_i], (0, 2, 1))
    assert np.abs(rel - R).max() < 1e-9
