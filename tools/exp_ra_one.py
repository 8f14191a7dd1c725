import sys, time
sys.path.insert(0, '/root/repo')
from glomap_amd import _lib, estimators, synthetic
ctx = _lib.Context(0)
p = synthetic.make_ring_view_graph(5000, 50, seed=0)
for _ in range(2):
    t0 = time.time(); rc, rot, rep = estimators.ra_solve(p, ctx=ctx); print(rc, rep['linear_iterations'], time.time() - t0)
