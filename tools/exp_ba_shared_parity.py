"""BA configs[3] with ONE shared camera on the GPU; saves the result for comparison with the CPU oracle's
(tests/golden/make_ba_shared_golden.py).  gpurun -- 'python tools/exp_ba_shared_parity.py'"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from glomap_amd import _lib, estimators, synthetic
ctx = _lib.Context(0)
p = synthetic.make_ba_problem(10_000, 1_000_000, seed=0, shared_intrinsics=True)
for rep_i in range(2):
    t0 = time.time()
    rc, q, t, X, intr, rep = estimators.ba_solve(p, ctx=ctx)
    print('rc', rc, 'LM', rep['iterations'], 'acc', rep['successful_steps'], 'pcg', rep['linear_iterations'], 'cost', rep['final_cost'], '%.1f ms' % ((time.time() - t0) * 1e3), flush=True)
os.makedirs('gpurun_out', exist_ok=True)
np.savez('gpurun_out/ba_shared_gpu.npz', q=q, t=t, intr=intr, final_cost=rep['final_cost'], iterations=rep['iterations'])
