#!/usr/bin/env python
"""Generates tests/golden/gp_c4_s{1,2}_oracle.npz: the exact-solve C++ oracle's camera centres for global positioning at
configs[3] size (10 000 cameras / 1 M tracks / ~6.0 M observations), seeds 1 and 2 of
tests/test_fullsize_gpu.py::_gp_full_size_problem, with the oracle's reductions summed forwards (order 0) AND backwards
(order 1) — two rounding-level variants of the same restatement ("GP parity, round 5").

Why fixtures: each run is 1 - 2 minutes on the GPU box's 16 cores and the GPU suite would need five of them at this size;
the oracle's reductions are thread-count independent, so what is generated here IS what the box would compute.  Seed 0 —
the GP problem of the headline — and all configs[2] seeds keep a LIVE oracle run in the test.  The test regenerates the
input from the seed and checks it against the checksums stored here.

Usage: python tests/golden/make_gp_c4_golden.py      (about 10 minutes on 8 cores)"""
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))

from glomap_amd import synthetic  # noqa: E402
from oracle import cpu  # noqa: E402


def main():
    for seed in (1, 2):
        p = synthetic.make_gp_problem(10_000, 1_000_000, seed=seed, uncalibrated_ratio=0.1 if seed == 1 else 0.0)
        out = dict(num_obs=p.num_obs, obs_dir_checksum=float(np.sum(p.obs_dir)), obs_cam_checksum=int(np.sum(p.obs_cam.astype(np.int64))),
                   calibrated_checksum=int(np.sum(p.obs_calibrated.astype(np.int64))))
        for order in (0, 1):
            t0 = time.time()
            ok, c, X, s = cpu.gp_solve(p.num_cams, p.pt_offset, p.obs_cam, p.obs_dir, p.obs_calibrated, p.cam_center, p.pt_xyz, order=order)
            assert ok
            print(f"seed {seed} order {order}: LM {s.iterations} final cost {s.final_cost:.6f} max true relres {s.max_linear_residual:.1e} "
                  f"{time.time() - t0:.0f} s", flush=True)
            out.update({f"center_{order}": c, f"iterations_{order}": s.iterations, f"final_cost_{order}": s.final_cost,
                        f"initial_cost_{order}": s.initial_cost, f"max_linear_residual_{order}": s.max_linear_residual})
        np.savez_compressed(Path(__file__).resolve().parent / f"gp_c4_s{seed}_oracle.npz", **out)


if __name__ == "__main__":
    main()
