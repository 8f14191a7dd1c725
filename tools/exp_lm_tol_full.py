"""Self-consistency of the GP / BA solves at full configs[2] / configs[3] size as a function of the PCG tolerance (the
oracle's exact Schur solves are out of reach at this size): every result is compared with the tightest one."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from glomap_amd import _lib, estimators, so3, synthetic
ctx = _lib.Context(0)
which = sys.argv[1] if len(sys.argv) > 1 else "gp"
tols = (1e-8, 1e-9, 1e-10, 1e-11, 1e-12, 1e-13)
if which == "gp":
    scale = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
    p = synthetic.make_gp_problem(int(5000 * scale), int(500_000 * scale), seed=0)
    res = {}
    for tol in tols:
        o = estimators.GlobalPositionerOptions(); o.solver_options.pcg_relative_tolerance = tol; o.solver_options.pcg_max_iterations = 5000
        t0 = time.time(); rc, c, X, rep = estimators.gp_solve(p, o, ctx=ctx)
        print('GP tol %.0e' % tol, rc, 'lm', rep['iterations'], 'ok', rep['successful_steps'], 'pcg', rep['linear_iterations'], 'cost %.9e' % rep['final_cost'], '%.0f ms' % ((time.time() - t0) * 1e3), flush=True)
        res[tol] = c
    for tol in tols[:-1]:
        print('GP centre difference %.0e vs %.0e (Sim3-aligned, relative): %.3e' % (tol, tols[-1], synthetic.center_errors_after_sim3(res[tol], res[tols[-1]]).max()))
    print('vs ground truth: %.3e' % synthetic.center_errors_after_sim3(res[tols[-1]], p.gt_center).max())
