"""Exact-solve C++ oracle results for the full-size GP problems, cached for GPU experiments (NOT used by the tests, which run
the oracle live): oracle/_cache/gp_{cams}_{tracks}_s{seed}.npz.  The oracle's reductions are thread-count independent, so
a result computed in the build container is the result the GPU box would compute — given the same input, which the file
pins with a checksum.  Usage: python tools/make_gp_oracle_cache.py cams tracks seed [seed ...]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from glomap_amd import synthetic  # noqa: E402
from oracle import cpu  # noqa: E402

N, P = int(sys.argv[1]), int(sys.argv[2])
for seed in map(int, sys.argv[3:]):
    out = f"oracle/_cache/gp_{N}_{P}_s{seed}.npz"
    if os.path.exists(out):
        continue
    p = synthetic.make_gp_problem(N, P, seed=seed)
    t0 = time.time()
    ok, c, X, s = cpu.gp_solve(p.num_cams, p.pt_offset, p.obs_cam, p.obs_dir, p.obs_calibrated, p.cam_center, p.pt_xyz)
    np.savez(out, center=c, iterations=s.iterations, final_cost=s.final_cost, initial_cost=s.initial_cost,
             linear_iterations=s.linear_iterations, max_linear_residual=s.max_linear_residual,
             obs_dir_checksum=float(np.sum(p.obs_dir)), num_obs=p.num_obs)
    print(out, ok, s.iterations, s.final_cost, s.linear_iterations, round(time.time() - t0, 1), flush=True)
