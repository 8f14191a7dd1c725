"""python tools/exp_ba_aw_check.py (sets the knob ba_aw_check): the library prints, per deflated BA solve and gauge mode, the largest
difference between the closed-form product A W (k_ba_aw_modes) and the one formed by an operator application — joint pose /
intrinsics blocks with 7 modes, frozen rotations (4 modes), no free intrinsics."""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
from glomap_amd import _lib, estimators, synthetic
ctx = _lib.Context(0)
ctx.set_knob("ba_aw_check", 1)
for kw, opts in [(dict(), dict()), (dict(), dict(optimize_rotations=False)), (dict(shared_intrinsics=True), dict(optimize_intrinsics=False))]:
    p = synthetic.make_ba_problem(num_cams=150, num_pts=6000, seed=11, **kw)
    o = estimators.BundleAdjusterOptions(**opts)
    o.solver_options.max_num_iterations = 3
    print("case", kw, opts, file=sys.stderr)
    rc, *_, rep = estimators.ba_solve(p, o, ctx=ctx)
    print(rc, rep["iterations"], rep["linear_iterations"], file=sys.stderr)
