import sys, time
sys.path.insert(0, '/root/repo')
import numpy as np
from glomap_amd import _lib, estimators, synthetic
import bench
ctx = _lib.Context(0)
p = synthetic.make_gp_problem(10_000, 1_000_000, seed=0)
d = bench._dev_gp(ctx, p)
for i in range(3):
    t0 = time.perf_counter(); rc, c, X, r = estimators.gp_solve(d, ctx=ctx); ctx.synchronize(); dt = time.perf_counter() - t0
    print('GP wall %.1f ms  total %.1f  solve %.1f  setup %.1f' % (dt * 1e3, r['seconds_total'] * 1e3, r['seconds_solve'] * 1e3, (r['seconds_total'] - r['seconds_solve']) * 1e3), r['iterations'], r['linear_iterations'])
p = synthetic.make_ba_problem(10_000, 1_000_000, seed=0)
d = bench._dev_ba(ctx, p)
for i in range(3):
    t0 = time.perf_counter(); rc, q, t, X, intr, r = estimators.ba_solve(d, ctx=ctx); ctx.synchronize(); dt = time.perf_counter() - t0
    print('BA wall %.1f ms  total %.1f  solve %.1f  setup %.1f' % (dt * 1e3, r['seconds_total'] * 1e3, r['seconds_solve'] * 1e3, (r['seconds_total'] - r['seconds_solve']) * 1e3), r['iterations'], r['linear_iterations'])
p = synthetic.make_ring_view_graph(10_000, 50, seed=0)
d = bench._dev_ra(ctx, p)
for i in range(3):
    rot = d.node_aa0.clone()
    t0 = time.perf_counter(); rc, _, r = estimators.ra_solve(d, ctx=ctx, rot_inout=rot); ctx.synchronize(); dt = time.perf_counter() - t0
    print('RA wall %.1f ms  total %.1f  solve %.1f  setup %.1f' % (dt * 1e3, r['seconds_total'] * 1e3, r['seconds_solve'] * 1e3, (r['seconds_total'] - r['seconds_solve']) * 1e3))
