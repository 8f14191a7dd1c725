"""One-off timing of a full RA solve: python tools/exp_ra_one.py [N succ]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from glomap_amd import _lib, estimators, synthetic
N = int(sys.argv[1]) if len(sys.argv) > 1 else 5000
succ = int(sys.argv[2]) if len(sys.argv) > 2 else 50
ctx = _lib.Context(0)
p = synthetic.make_ring_view_graph(N, succ, seed=0)
for _ in range(3):
    t0 = time.time(); rc, rot, rep = estimators.ra_solve(p, ctx=ctx)
    print(N, rc, 'l1', rep['iterations_l1'], 'irls', rep['iterations_irls'], 'lin', rep['linear_iterations'], '%.1f ms' % ((time.time() - t0) * 1e3), flush=True)
