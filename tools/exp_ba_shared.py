"""BA at configs[3] size with per-image intrinsics vs ONE shared camera (and the shared case with constant intrinsics:
how many PCG iterations the pose-intrinsics coupling costs the block-Jacobi preconditioner)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from glomap_amd import _lib, estimators, synthetic
ctx = _lib.Context(0)
scale = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
N, P = int(10_000 * scale), int(1_000_000 * scale)
for name, kw, opt in (("per-image", dict(shared_intrinsics=False), estimators.BundleAdjusterOptions()),
                      ("shared", dict(shared_intrinsics=True), estimators.BundleAdjusterOptions()),
                      ("shared, intrinsics constant", dict(shared_intrinsics=True), estimators.BundleAdjusterOptions(optimize_intrinsics=False))):
    p = synthetic.make_ba_problem(N, P, seed=0, **kw)
    for _ in range(2):
        t0 = time.time()
        rc, q, t, X, intr, rep = estimators.ba_solve(p, opt, ctx=ctx)
        dt = time.time() - t0
    print(name, 'K', p.num_intr, 'rc', rc, 'LM', rep['iterations'], 'pcg', rep['linear_iterations'], 'pcg/LM %.1f' % (rep['linear_iterations'] / rep['iterations']),
          'cost', rep['final_cost'], '%.1f ms' % (dt * 1e3), flush=True)
