#!/usr/bin/env python
"""Generates tests/golden/chain_c4_oracle.npz (and chain_2k_oracle.npz): the CPU oracle's result for the WHOLE hot path
chained on one scene the way GlobalMapper::Solve chains it (global_mapper.cc:92-223; driver: tests/chain_util.py run_chain):

    rotation averaging -> global positioning (bearings oriented by RA's rotations, random start)
    -> FilterTracksByAngle / FilterTrackTriangulationAngle / FilterTracksByReprojection(10 x) -> NormalizeReconstruction
    -> bundle adjustment, positions only -> bundle adjustment with rotations.

Scene: synthetic.make_chained_scene(10_000, 1_000_000, seed=0).

Why a fixture: the exact-solve oracle stages at this size are ~15 minutes on 8 cores; every reduction of oracle/csrc runs
in a fixed order, so the numbers generated here ARE what the GPU box would compute.  The test regenerates the scene from the
seed (pinned by checksums) and runs the HIP chain; only the oracle's stage results travel (RA rotations, GP centres, the
observation counts after each filter, final poses).  Reduced systems solved to 1e-14, true residuals recorded.

Usage: python tests/golden/make_chain_golden.py [cams tracks name [seed]] [--reversed]"""
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))

from chain_util import OracleBackend, run_chain  # noqa: E402
from glomap_amd import synthetic  # noqa: E402


def scene_checksums(sc):
    return dict(num_obs=int(sc.obs_cam.shape[0]), obs_xy_checksum=float(np.sum(sc.obs_xy)), edge_q_checksum=float(np.sum(sc.ra.edge_q)),
                obs_cam_checksum=int(np.sum(sc.obs_cam.astype(np.int64))))


def main():
    N, P, name, seed = 10_000, 1_000_000, "chain_c4_oracle.npz", 0
    argv = [a for a in sys.argv if not a.startswith("--")]
    if len(argv) > 3:
        N, P, name = int(argv[1]), int(argv[2]), argv[3]
    if len(argv) > 4:
        seed = int(argv[4])
    sc = synthetic.make_chained_scene(N, P, seed=seed)
    t0 = time.time()
    be = OracleBackend(verbose=True)
    r = run_chain(sc, be)
    g, b1, b2 = r["rep_gp"], r["rep_ba1"], r["rep_ba2"]
    print("RA %d+%d | GP LM %d cost %.6f (max true relres %.1e) | observations kept %s | BA positions-only LM %d (%d accepted) cost "
          "%.3f -> %.3f (max true relres %.1e) | BA full LM %d (%d accepted) cost %.3f -> %.3f (max true relres %.1e) | %.0f s"
          % (r["rep_ra"]["l1"], r["rep_ra"]["irls"], g["iterations"], g["final_cost"], g["max_linear_residual"], r["observations_kept"],
             b1["iterations"], b1["successful"], b1["initial_cost"], b1["final_cost"], b1["max_linear_residual"],
             b2["iterations"], b2["successful"], b2["initial_cost"], b2["final_cost"], b2["max_linear_residual"], time.time() - t0))
    out = dict(ra_rot=r["ra_rot"], ra_l1=r["rep_ra"]["l1"], ra_irls=r["rep_ra"]["irls"], gp_center=r["gp_center"],
               observations_kept=np.array(r["observations_kept"], dtype=np.int64), ba_q=r["ba_q"], ba_t=r["ba_t"],
               ba_intr_f=r["ba_intr"][:, 0].copy())
    for stage, rep in (("gp", g), ("ba1", b1), ("ba2", b2)):
        for k, v in rep.items():
            out[f"{stage}_{k}"] = v
    out["gp_trace"] = be.gp_trace  # the LM iterations of global positioning (oracle.cpu.lm_trace)
    if "--reversed" in sys.argv:
        # the same chain with every owner-side reduction of GP / BA summed backwards: the reference algorithm at another
        # rounding — how well defined its own end points are
        t1 = time.time()
        be2 = OracleBackend(verbose=True, order=1)
        r2 = run_chain(sc, be2)
        out.update(rev_gp_center=r2["gp_center"], rev_ba_q=r2["ba_q"], rev_ba_t=r2["ba_t"], rev_gp_trace=be2.gp_trace,
                   rev_observations_kept=np.array(r2["observations_kept"], dtype=np.int64),
                   rev_gp_iterations=r2["rep_gp"]["iterations"], rev_gp_final_cost=r2["rep_gp"]["final_cost"])
        print("reversed sums: GP LM %d cost %.6f | kept %s | %.0f s" % (r2["rep_gp"]["iterations"], r2["rep_gp"]["final_cost"],
                                                                         r2["observations_kept"], time.time() - t1))
    np.savez_compressed(Path(__file__).resolve().parent / name, **out, seed=seed, **scene_checksums(sc))


if __name__ == "__main__":
    main()
