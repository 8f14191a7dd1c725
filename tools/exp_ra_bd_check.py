"""Deviation from the oracle of the iterative RA paths as a function of the ADMM inner tolerance:
python tools/exp_ra_bd_check.py N succ"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from glomap_amd import _lib, estimators, so3, synthetic
from oracle import ra as ora, so3 as oso3
ctx = _lib.Context(0)
n, succ = int(sys.argv[1]), int(sys.argv[2])
p = synthetic.make_ring_view_graph(n, succ, noise_deg=1.0, outlier_ratio=0.05, seed=11)
def ang(a, b):
    return np.radians(so3.rotation_angle_deg(oso3.exp_aa(a), oso3.exp_aa(b)))
opt = ora.RotationEstimatorOptions(); tr = ora.RaTrace()
ok, rot_o = ora.estimate_rotations(p.num_nodes, p.edge_i, p.edge_j, p.edge_q, p.edge_weight, p.edge_ninl, p.node_aa0, p.fixed_node, opt, tr)
print('oracle l1', tr.l1_iterations, 'irls', tr.irls_iterations)
for force in (False, True):
    for tol in (1e-3, 1e-4, 1e-5, 1e-6, 1e-8, 1e-10):
        t0 = time.time()
        rc, rot, rep = estimators.ra_solve(p, estimators.RotationEstimatorOptions(force_iterative=force, pcg_relative_tolerance_admm=tol), ctx=ctx)
        print('jacobi' if force else 'block ', 'admm tol %.0e' % tol, 'l1', rep['iterations_l1'], 'irls', rep['iterations_irls'], 'lin', rep['linear_iterations'],
              '%.1f ms' % ((time.time() - t0) * 1e3), 'max dev vs oracle %.3e rad' % ang(rot, rot_o).max(), flush=True)
