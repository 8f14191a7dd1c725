#!/usr/bin/env python
"""Generates tests/golden/chain_c4_oracle.npz: the CPU oracle's result for the WHOLE hot path chained on one configs[3]-size
scene — rotation averaging -> global positioning (bearings oriented by the rotations RA returned, random start) -> bundle
adjustment (started from GP's centres and points) — as GlobalMapper::Solve chains the three estimators
(global_mapper.cc:92-223).  Scene: synthetic.make_chained_scene(10_000, 1_000_000, seed=0).

Why a fixture: three exact-solve oracle stages at this size are ~15 minutes on 8 cores; every reduction of oracle/csrc runs
in a fixed order, so the numbers generated here ARE what the GPU box would compute.  The test regenerates the scene from the
seed (pinned by checksums) and runs the HIP chain; only the oracle's stage results travel (RA rotations, GP centres, final
poses).  Reduced systems solved to 1e-14, true residuals recorded.

Usage: python tests/golden/make_chain_golden.py [cams tracks name]"""
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))

from glomap_amd import so3, synthetic  # noqa: E402
from oracle import cpu  # noqa: E402


def oracle_chain(sc, verbose=False):
    """The three oracle stages chained; returns a dict of stage results and reports."""
    p = sc.ra
    rep_ra = {}
    ok, rot = cpu.ra_estimate_rotations(p.num_nodes, p.edge_i, p.edge_j, p.edge_q, p.edge_weight, p.edge_ninl, p.node_aa0,
                                        p.fixed_node, report=rep_ra)
    assert ok
    R = so3.aa_to_rotmat(rot)
    g = synthetic.chain_gp_problem(sc, R)
    ok, c, X, sg = cpu.gp_solve(g.num_cams, g.pt_offset, g.obs_cam, g.obs_dir, g.obs_calibrated, g.cam_center, g.pt_xyz, verbose=verbose)
    assert ok
    b = synthetic.chain_ba_problem(sc, R, c, X)
    r = cpu.ba_solve(b.num_cams, b.pt_offset, b.obs_cam, b.obs_xy, b.cam_intr, b.intr_model, b.fixed_cam, b.cam_q, b.cam_t, b.pt_xyz,
                     b.intr_params, verbose=verbose)
    assert r[0]
    sb = r[5]
    return dict(ra_rot=rot, ra_l1=rep_ra["l1_iterations"], ra_irls=rep_ra["irls_iterations"], gp_center=c,
                gp_iterations=sg.iterations, gp_initial_cost=sg.initial_cost, gp_final_cost=sg.final_cost,
                gp_max_linear_residual=sg.max_linear_residual, ba_q=r[1], ba_t=r[2], ba_intr_f=r[4][:, 0].copy(),
                ba_iterations=sb.iterations, ba_successful=sb.successful_steps, ba_initial_cost=sb.initial_cost,
                ba_final_cost=sb.final_cost, ba_max_linear_residual=sb.max_linear_residual)


def scene_checksums(sc):
    return dict(num_obs=int(sc.obs_cam.shape[0]), obs_xy_checksum=float(np.sum(sc.obs_xy)), edge_q_checksum=float(np.sum(sc.ra.edge_q)),
                obs_cam_checksum=int(np.sum(sc.obs_cam.astype(np.int64))))


def main():
    N, P, name = 10_000, 1_000_000, "chain_c4_oracle.npz"
    if len(sys.argv) > 3:
        N, P, name = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
    sc = synthetic.make_chained_scene(N, P, seed=0)
    t0 = time.time()
    out = oracle_chain(sc, verbose=True)
    print("RA %d+%d  GP LM %d cost %.6f (max true relres %.1e)  BA LM %d (%d accepted) cost %.3f -> %.3f (max true relres %.1e)  %.0f s"
          % (out["ra_l1"], out["ra_irls"], out["gp_iterations"], out["gp_final_cost"], out["gp_max_linear_residual"],
             out["ba_iterations"], out["ba_successful"], out["ba_initial_cost"], out["ba_final_cost"],
             out["ba_max_linear_residual"], time.time() - t0))
    np.savez_compressed(Path(__file__).resolve().parent / name, **out, **scene_checksums(sc))


if __name__ == "__main__":
    main()
