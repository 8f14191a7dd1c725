"""One-off timing of a full RA solve: python tools/exp_ra_one.py [N succ [compare]] — `compare` also runs the Jacobi-PCG
path and prints the largest rotation difference between the two."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from glomap_amd import _lib, estimators, so3, synthetic
N = int(sys.argv[1]) if len(sys.argv) > 1 else 5000
succ = int(sys.argv[2]) if len(sys.argv) > 2 else 50
ctx = _lib.Context(0)
p = synthetic.make_ring_view_graph(N, succ, seed=0)
for _ in range(2):
    t0 = time.time(); rc, rot, rep = estimators.ra_solve(p, ctx=ctx)
    print(N, rc, 'l1', rep['iterations_l1'], 'irls', rep['iterations_irls'], 'lin', rep['linear_iterations'], '%.1f ms' % ((time.time() - t0) * 1e3), flush=True)
if len(sys.argv) > 3:
    t0 = time.time(); rc, rot_j, rep = estimators.ra_solve(p, estimators.RotationEstimatorOptions(force_iterative=True), ctx=ctx)
    d = np.radians(so3.rotation_angle_deg(so3.aa_to_rotmat(rot), so3.aa_to_rotmat(rot_j))).max()
    print('jacobi', rc, 'l1', rep['iterations_l1'], 'irls', rep['iterations_irls'], 'lin', rep['linear_iterations'], '%.1f ms' % ((time.time() - t0) * 1e3), 'max diff %.2e rad' % d, flush=True)
    for name, r in (('block', rot), ('jacobi', rot_j)):
        err = synthetic.rotation_errors_deg(so3.aa_to_rotmat(r), p.gt_R)
        print(name, 'GT error deg: median %.3f max %.3f' % (np.median(err), err.max()))
