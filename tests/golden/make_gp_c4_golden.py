#!/usr/bin/env python
"""Generates tests/golden/gp_c4_s{0,1,2}_oracle.npz and gp_c3_s{1,2}_oracle.npz: the exact-solve C++ oracle's camera centres for
global positioning at configs[3] size (10 000 cameras / 1 M tracks / ~6.0 M observations) and configs[2] size (5 000 / 500 k /
~3.0 M), seeds of
tests/test_fullsize_gpu.py::_gp_full_size_problem, with the oracle's reductions summed forwards (order 0) AND backwards
(order 1) — two rounding-level variants of the same restatement.  Round 6: the oracle's LM loop carries
Ceres' projected line search (oracle/lm.py header); the LM iterations of each run are stored too (trace_{order}: the rows of
oracle.cpu.lm_trace), so that the GPU test can compare trajectories, not only end points.

Why fixtures: each run is 1 - 2 minutes on the GPU box's 16 cores and the GPU suite would need five of them at this size;
the oracle's reductions are thread-count independent, so what is generated here IS what the box would compute.  Seed 0 —
the GP problem of the headline — and all configs[2] seeds keep a LIVE oracle run in the test.  The test regenerates the
input from the seed and checks it against the checksums stored here.

Usage: python tests/golden/make_gp_c4_golden.py [cams tracks seed]...     (about 40 minutes on 8 cores for all five)"""
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))

from glomap_amd import synthetic  # noqa: E402
from oracle import cpu  # noqa: E402


def main():
    # configs[2] seeds 1, 2 (seed 0 runs live in the test) and configs[3] size seeds 0, 1, 2; or the cases named on the command line
    a = [int(v) for v in sys.argv[1:]]
    cases = [tuple(a[i:i + 3]) for i in range(0, len(a), 3)] or [(5000, 500_000, 1), (5000, 500_000, 2), (10_000, 1_000_000, 0),
                                                                 (10_000, 1_000_000, 1), (10_000, 1_000_000, 2)]
    for (N, P, seed) in cases:
        p = synthetic.make_gp_problem(N, P, seed=seed, uncalibrated_ratio=0.1 if seed == 1 else 0.0)
        out = dict(num_obs=p.num_obs, obs_dir_checksum=float(np.sum(p.obs_dir)), obs_cam_checksum=int(np.sum(p.obs_cam.astype(np.int64))),
                   calibrated_checksum=int(np.sum(p.obs_calibrated.astype(np.int64))))
        for order in (0, 1):
            t0 = time.time()
            ok, c, X, s = cpu.gp_solve(p.num_cams, p.pt_offset, p.obs_cam, p.obs_dir, p.obs_calibrated, p.cam_center, p.pt_xyz, order=order)
            assert ok
            print(f"seed {seed} order {order}: LM {s.iterations} final cost {s.final_cost:.6f} max true relres {s.max_linear_residual:.1e} "
                  f"{time.time() - t0:.0f} s", flush=True)
            out.update({f"trace_{order}": cpu.lm_trace(), f"successful_{order}": s.successful_steps,
                        f"line_search_shrunk_{order}": s.line_search_shrunk,
                        f"center_{order}": c, f"iterations_{order}": s.iterations, f"final_cost_{order}": s.final_cost,
                        f"initial_cost_{order}": s.initial_cost, f"max_linear_residual_{order}": s.max_linear_residual})
        np.savez_compressed(Path(__file__).resolve().parent / f"gp_c{3 if N == 5000 else 4}_s{seed}_oracle.npz", **out)


if __name__ == "__main__":
    main()
