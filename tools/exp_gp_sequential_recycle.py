"""GPU: global positioning on the sequential capture of configs[2] size (the scene whose reduced solves switch the second-level
cluster preconditioner on) with and without the knob gp_no_recycle — the experiment behind `!coarse` in GpSolver::pcg's
`recycle` condition (recycling on top of the second level: 6 139 instead of 6 243 iterations, 698 instead of 628 ms).
Usage: python tools/exp_gp_sequential_recycle.py"""
import sys, time, json, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
from glomap_amd import estimators, synthetic
from glomap_amd._lib import Context
ctx = Context()
p = synthetic.make_gp_problem(5000, 500000, seed=0, capture="sequential")
for knobs in ({}, {"gp_no_recycle": 1}):
    for k, v in knobs.items(): ctx.set_knob(k, v)
    for rep_i in range(2):
        ctx.stats(reset=True)
        t0 = time.perf_counter()
        rc, cen, xyz, rep = estimators.gp_solve(p, ctx=ctx)
        ms = (time.perf_counter() - t0) * 1e3
    tr = ctx.lm_trace()
    print(json.dumps(dict(knobs=knobs, rc=rc, lm=rep["iterations"], pcg=rep["linear_iterations"], ms=round(ms, 1), stats=ctx.stats(), per=[int(r[-1]) for r in tr])), flush=True)
    for k in knobs: ctx.set_knob(k, 0)
