import sys, time
sys.path.insert(0, '/root/repo')
import numpy as np
from glomap_amd import _lib, estimators, synthetic, so3
ctx = _lib.Context(0)
for N, succ in ((2000, 50), (5000, 50), (10000, 50)):
    p = synthetic.make_ring_view_graph(N, succ, seed=0)
    for rep_i in range(2):
        t0 = time.time()
        rc, rot, rep = estimators.ra_solve(p, ctx=ctx)
        dt = time.time() - t0
    err = synthetic.rotation_errors_deg(so3.aa_to_rotmat(rot), p.gt_R)
    print(N, p.num_edges, rc, rep['iterations_l1'], rep['iterations_irls'], rep['linear_iterations'], round(dt*1e3,1), 'ms', round(rep['seconds_total']*1e3,1), round(rep['seconds_solve']*1e3,1), float(np.median(err)), flush=True)
