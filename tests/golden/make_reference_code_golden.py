"""Golden vectors produced by REFERENCE CODE run in the build container (oracle/_ref: the reference's own sources compiled
from /root/reference against stand-in headers — `make -C oracle ref`).  The libraries travel to the GPU box, but the larger
cases take minutes of CPU there (dense Cholesky stand-in for CHOLMOD), so their results are frozen here:

  ra_c2_reference_code.npz   BASELINE configs[1]: ring view graph, 1 000 cameras / 50 000 relative rotations, seed 0, the
                             reference's RotationEstimator::EstimateRotations with its default options (spanning-tree start)
                             -> frame rotations (wxyz), fixed image, tree root, L1 / IRLS iteration counts.  The inlier counts
                             U{30..500} of the benchmark graph are full of ties, and among tied edges the spanning tree
                             depends on Boost's priority queue (a different tree moves the end point by 6e-5 rad: the IRLS
                             loop stops at a mean step of 1e-3 rad): the counts are made distinct in the order this
                             library breaks ties (synthetic.break_inlier_ties_by_index), which leaves its own result unchanged
  ra_c4_reference_code.npz   BASELINE configs[3]'s view graph (10 000 cameras / 500 000 relative rotations), the same way
  gp_start_reference_code.npz  BASELINE configs[2] size (5 000 cameras / 500 000 tracks / 3.0 M observations, seed 0):
                             GlobalPositioner::Solve on the recording Ceres -> the walk orders of its hash maps (= its draw
                             order) and the initial cost of its random start  (13 s, 2.2 GB)
  ba_start_reference_code.npz  BASELINE configs[3] size (10 000 cameras / 1 M tracks / 5.0 M observations, seed 0):
                             BundleAdjuster::Solve on the recording Ceres -> the constant frame and the initial cost

Usage (from the repository root):  python tests/golden/make_reference_code_golden.py
"""
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
from glomap_amd import so3, synthetic  # noqa: E402
from oracle import ref  # noqa: E402

OUT = Path(__file__).resolve().parent


def ra_c2():
    p = synthetic.make_ring_view_graph(1000, 50, seed=0)
    N = p.num_nodes
    t0 = time.time()
    r = ref.ra_estimate([0], np.zeros(N), np.arange(N), np.zeros(N), p.edge_i, p.edge_j, p.edge_q, pair_weight=p.edge_weight,
                        pair_ninl=synthetic.break_inlier_ties_by_index(p.edge_ninl), frame_q=so3.aa_to_quat(p.node_aa0))
    assert r["ok"]
    print("ra_c2", N, len(p.edge_i), f"{time.time() - t0:.1f} s", {k: v for k, v in r.items() if not hasattr(v, "shape")})
    np.savez_compressed(OUT / "ra_c2_reference_code.npz", frame_q=r["frame_q"], fixed_image=r["fixed_image"], tree_root=r["tree_root"],
                        l1_iterations=r["l1_iterations"], irls_iterations=r["irls_iterations"], admm_iterations=r["admm_iterations"])


def ra_c4():
    """BASELINE configs[3]'s view graph (10 000 cameras / 500 000 relative rotations, seed 0) through the reference's own
    RotationEstimator::EstimateRotations (round 6: the CHOLMOD stand-in factors an envelope after reverse Cuthill-McKee, so the
    30 000-unknown systems of gra.cc:543-625 are tractable here) -> ra_c4_reference_code.npz, with the wall time of the run."""
    p = synthetic.make_ring_view_graph(10_000, 50, seed=0)
    N = p.num_nodes
    t0 = time.time()
    r = ref.ra_estimate([0], np.zeros(N), np.arange(N), np.zeros(N), p.edge_i, p.edge_j, p.edge_q, pair_weight=p.edge_weight,
                        pair_ninl=synthetic.break_inlier_ties_by_index(p.edge_ninl), frame_q=so3.aa_to_quat(p.node_aa0))
    sec = time.time() - t0
    assert r["ok"]
    print("ra_c4", N, len(p.edge_i), f"{sec:.1f} s", {k: v for k, v in r.items() if not hasattr(v, "shape")})
    np.savez_compressed(OUT / "ra_c4_reference_code.npz", frame_q=r["frame_q"], fixed_image=r["fixed_image"], tree_root=r["tree_root"],
                        l1_iterations=r["l1_iterations"], irls_iterations=r["irls_iterations"], admm_iterations=r["admm_iterations"],
                        seconds_one_thread=sec)


def gp_start():
    p = synthetic.make_gp_problem(5000, 500_000, seed=0)  # tests/test_fullsize_gpu.py::_gp_full_size_problem(seed 0)
    q = so3.rotmat_to_quat(p.cam_R)
    t = -np.einsum("nij,nj->ni", p.cam_R, p.gt_center)
    und = np.einsum("mij,mj->mi", p.cam_R[p.obs_cam], p.obs_dir)
    cal = np.ones(p.num_cams, np.uint8)
    cal[p.obs_cam] = p.obs_calibrated
    t0 = time.time()
    r = ref.gp_build(q, t, p.pt_offset, p.obs_cam, und, p.pt_xyz, cam_calibrated=cal)
    print("gp_start", p.num_cams, p.num_pts, p.num_obs, f"{time.time() - t0:.1f} s", r["initial_cost"], r["num_residual_blocks"])
    np.savez_compressed(OUT / "gp_start_reference_code.npz", frame_order=r["frame_order"].astype(np.int32),
                        track_order=r["track_order"].astype(np.int32), initial_cost=r["initial_cost"],
                        num_residual_blocks=r["num_residual_blocks"], center_start=r["center_start"])


def ba_start():
    p = synthetic.make_ba_problem(10_000, 1_000_000, seed=0, shared_intrinsics=False)  # test_ba_config4_matches_cpu_oracle's input
    t0 = time.time()
    r = ref.ba_build(p.intr_model, p.intr_params, p.cam_q, p.cam_t, np.arange(p.num_cams), p.cam_intr, p.pt_offset, p.obs_cam, p.obs_xy, p.pt_xyz,
                     rig_ref_cam=np.arange(p.num_intr), frame_rig=p.cam_intr)
    order = r["frame_order"]
    fixed = int(order[(r["frame_flags"][order] & 1) != 0][0])
    print("ba_start", p.num_cams, p.num_pts, p.num_obs, f"{time.time() - t0:.1f} s", r["initial_cost"], fixed)
    np.savez_compressed(OUT / "ba_start_reference_code.npz", fixed_frame=fixed, initial_cost=r["initial_cost"],
                        num_residual_blocks=r["num_residual_blocks"])


if __name__ == "__main__":
    which = sys.argv[1:] or ["ra_c2", "gp_start", "ba_start"]
    for w in which:
        globals()[w]()
