"""Global positioning on a sequential-capture scene (synthetic capture="sequential"): LM / PCG counts, time, error against
ground truth.  GSFM_KNOBS=gp_coarse_cluster=<cluster size> varies the coarse space; GSFM_VERBOSE=1 prints the PCG count per LM step."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from glomap_amd import _lib, estimators, synthetic
N = int(sys.argv[1]) if len(sys.argv) > 1 else 5000
P = int(sys.argv[2]) if len(sys.argv) > 2 else 500_000
ctx = _lib.Context(0)
p = synthetic.make_gp_problem(N, P, seed=0, capture="sequential")
for rep in range(2):
    t0 = time.perf_counter()
    rc, c, X, r = estimators.gp_solve(p, ctx=ctx)
    dt = time.perf_counter() - t0
err = synthetic.center_errors_after_sim3(c, p.gt_center)
print(f"sequential {N}/{P}: rc {rc} LM {r['iterations']} ({r['successful_steps']} accepted) applications {r['linear_iterations']} "
      f"final cost {r['final_cost']:.4f} {dt * 1e3:.1f} ms  median centre error / 50 = {np.median(err) / 50:.2e}")
np.save("gpurun_out/gp_seq_%s.npy" % os.environ.get("GSFM_KNOBS", "default").replace("=", "_"), c)
