import sys, time
sys.path.insert(0, '/root/repo')
import numpy as np
from glomap_amd import _lib, estimators, synthetic
ctx = _lib.Context(0)
p = synthetic.make_ba_problem(10_000, 1_000_000, seed=0, shared_intrinsics=False)
for name, opt in (("full", estimators.BundleAdjusterOptions()), ("no_intr", estimators.BundleAdjusterOptions(optimize_intrinsics=False))):
    t0 = time.time()
    rc, q, t, X, intr, rep = estimators.ba_solve(p, opt, ctx=ctx)
    print(name, rc, rep['iterations'], rep['successful_steps'], rep['linear_iterations'], rep['final_cost'], round(time.time()-t0,2), flush=True)
ps = synthetic.make_ba_problem(10_000, 1_000_000, seed=0, shared_intrinsics=True)
rc, q, t, X, intr, rep = estimators.ba_solve(ps, estimators.BundleAdjusterOptions(), ctx=ctx)
print("shared", rc, rep['iterations'], rep['successful_steps'], rep['linear_iterations'], rep['final_cost'], flush=True)
