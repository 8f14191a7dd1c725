"""Saves the RA results of the block-preconditioned path (two ADMM inner tolerances) and of the Jacobi path for an
offline comparison with the oracle: python tools/exp_ra_save.py N"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from glomap_amd import _lib, estimators, synthetic
ctx = _lib.Context(0)
N = int(sys.argv[1])
p = synthetic.make_ring_view_graph(N, 50, seed=0)
out = {}
for name, opt in (("block_1e-6", estimators.RotationEstimatorOptions()),
                  ("block_1e-8", estimators.RotationEstimatorOptions(pcg_relative_tolerance_admm=1e-8)),
                  ("jacobi_1e-10", estimators.RotationEstimatorOptions(force_iterative=True, pcg_relative_tolerance_admm=1e-10, pcg_max_iterations=20000))):
    t0 = time.time(); rc, rot, rep = estimators.ra_solve(p, opt, ctx=ctx)
    print(name, rc, 'irls', rep['iterations_irls'], 'lin', rep['linear_iterations'], '%.0f ms' % ((time.time() - t0) * 1e3), flush=True)
    out[name] = rot
np.savez(os.path.join(os.environ.get('GRAFT_REPO_ROOT', '.'), 'gpurun_out', 'ra_%d.npz' % N), **out)
