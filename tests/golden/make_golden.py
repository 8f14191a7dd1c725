#!/usr/bin/env python
"""Generates the frozen fixtures tests/golden/*.npz.

The reference (colmap/glomap v1.1.0) ships no golden vectors for its estimators and cannot be built
in this image, so these fixtures are produced by the repository's own CPU
restatement (oracle/) on seeded synthetic problems: inputs + the oracle's outputs at the time of
generation.  They pin BOTH sides afterwards: tests/test_golden.py checks that today's oracle still
reproduces them (CPU) and that the HIP path matches them through the C ABI (GPU).

Usage: python tests/golden/make_golden.py        (rewrites the .npz files)
"""
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))

from glomap_amd import synthetic  # noqa: E402
from oracle import ba as oba  # noqa: E402
from oracle import gp as ogp  # noqa: E402
from oracle import ra as ora  # noqa: E402

OUT = Path(__file__).resolve().parent


def ra_fixture():
    p = synthetic.make_ring_view_graph(60, 8, seed=21)
    ok, rot = ora.estimate_rotations(p.num_nodes, p.edge_i, p.edge_j, p.edge_q, p.edge_weight, p.edge_ninl, p.node_aa0,
                                     p.fixed_node)
    assert ok
    np.savez_compressed(OUT / "ra_ring60.npz", num_nodes=p.num_nodes, edge_i=p.edge_i, edge_j=p.edge_j, edge_q=p.edge_q,
                        edge_weight=p.edge_weight, edge_ninl=p.edge_ninl, node_aa0=p.node_aa0, fixed_node=p.fixed_node,
                        out_rot_aa=rot)


def gp_fixture():
    p = synthetic.make_gp_problem(20, 300, seed=22)
    ok, c, X, s = ogp.solve(p.num_cams, p.pt_offset, p.obs_cam, p.obs_dir, p.obs_calibrated, p.cam_center, p.pt_xyz)
    assert ok
    np.savez_compressed(OUT / "gp_20x300.npz", num_cams=p.num_cams, pt_offset=p.pt_offset, obs_cam=p.obs_cam,
                        obs_dir=p.obs_dir, obs_calibrated=p.obs_calibrated, cam_center=p.cam_center, pt_xyz=p.pt_xyz,
                        out_center=c, out_xyz=X, out_final_cost=s.final_cost, out_iterations=s.iterations)


def ba_fixture():
    p = synthetic.make_ba_problem(num_cams=15, num_pts=300, seed=23, shared_intrinsics=True, intr_noise=0.01)
    ok, q, t, X, intr, s = oba.solve(p.num_cams, p.pt_offset, p.obs_cam, p.obs_xy, p.cam_intr, p.intr_model, p.fixed_cam,
                                     p.cam_q, p.cam_t, p.pt_xyz, p.intr_params)
    assert ok
    np.savez_compressed(OUT / "ba_15x300.npz", num_cams=p.num_cams, num_intr=p.num_intr, pt_offset=p.pt_offset,
                        obs_cam=p.obs_cam, obs_xy=p.obs_xy, cam_intr=p.cam_intr, intr_model=p.intr_model,
                        fixed_cam=p.fixed_cam, cam_q=p.cam_q, cam_t=p.cam_t, pt_xyz=p.pt_xyz, intr_params=p.intr_params,
                        out_q=q, out_t=t, out_xyz=X, out_intr=intr, out_initial_cost=s.initial_cost,
                        out_final_cost=s.final_cost, out_iterations=s.iterations)


def tracks_fixture():
    """Match graph -> established tracks -> selected tracks (integer results: frozen bit for bit)."""
    from oracle import tracks as ot

    g = synthetic.make_match_graph(40, 300, seed=24, false_match_frac=0.03, twin_frac=0.02)
    g["pair_valid"][::9] = 0
    full = ot.establish_full_tracks(g["pair_image1"], g["pair_image2"], g["pair_valid"], g["pair_offset"], g["match_feat1"],
                                    g["match_feat2"], g["feat_offset"], g["feat_xy"])
    reg = np.ones(40, dtype=np.uint8)
    reg[::6] = 0
    sel = ot.find_tracks_for_problem(*full[:4], reg, min_num_tracks_per_view=6)
    np.savez_compressed(OUT / "tracks_40x300.npz", **{k: g[k] for k in ("num_images", "feat_offset", "feat_xy", "pair_image1",
                        "pair_image2", "pair_valid", "pair_offset", "match_feat1", "match_feat2")},
                        image_registered=reg, min_num_tracks_per_view=6,
                        full_id=full[0], full_offset=full[1], full_image=full[2], full_feature=full[3], discarded=full[4],
                        sel_id=sel[0], sel_offset=sel[1], sel_image=sel[2], sel_feature=sel[3])


if __name__ == "__main__":
    tracks_fixture()
    ra_fixture()
    gp_fixture()
    ba_fixture()
    for f in sorted(OUT.glob("*.npz")):
        print(f.name, f.stat().st_size, "bytes")
