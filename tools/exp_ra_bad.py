import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from glomap_amd import _lib, estimators, so3, synthetic
ctx = _lib.Context(0)
for N in [int(a) for a in sys.argv[1:]]:
    p = synthetic.make_ring_view_graph(N, 50, seed=0)
    rc, rot, rep = estimators.ra_solve(p, ctx=ctx)
    err = synthetic.rotation_errors_deg(so3.aa_to_rotmat(rot), p.gt_R)
    bad = np.nonzero(err > 5)[0]
    print(N, 'irls', rep['iterations_irls'], 'median %.3f max %.3f' % (np.median(err), err.max()), 'bad nodes', len(bad), bad[:12], np.round(err[bad[:12]], 1), flush=True)
    # the same graph, IRLS only from the ground truth: is the bad node a property of the data?
    p2 = type(p)(**{**p.__dict__, "node_aa0": so3.quat_to_aa(so3.rotmat_to_quat(p.gt_R))})
    rc, rot2, rep2 = estimators.ra_solve(p2, estimators.RotationEstimatorOptions(skip_initialization=True, max_num_l1_iterations=0), ctx=ctx)
    err2 = synthetic.rotation_errors_deg(so3.aa_to_rotmat(rot2), p.gt_R)
    print('   from GT, IRLS only: irls', rep2['iterations_irls'], 'median %.3f max %.3f' % (np.median(err2), err2.max()), flush=True)
    # angle of the rotation vectors themselves (axis-angle norm close to pi?)
    nrm = np.linalg.norm(rot, axis=1)
    print('   |aa| max %.4f, nodes with |aa| > 3.1: %d' % (nrm.max(), (nrm > 3.1).sum()), flush=True)
