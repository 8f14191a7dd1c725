"""GSFM_VERBOSE trace of one GP solve (per LM iteration: radius, PCG iterations, model / candidate cost, line-search step) and
the solver-path counters.  Usage: GSFM_VERBOSE=1 python tools/exp_gp_verbose.py cams tracks seed [pcg_tol]"""
import sys

sys.path.insert(0, ".")
from glomap_amd import estimators, synthetic  # noqa: E402

N, P, seed = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
p = synthetic.make_gp_problem(N, P, seed=seed)
opt = estimators.GlobalPositionerOptions()
if len(sys.argv) > 4:
    opt.solver_options.pcg_relative_tolerance = float(sys.argv[4])
ctx = estimators.default_context()
rc, c, X, rep = estimators.gp_solve(p, opt, ctx=ctx)
print(rc, {k: rep[k] for k in ("iterations", "successful_steps", "linear_iterations", "final_cost", "line_search_trials", "line_search_shrunk", "seconds_solve")})
print(ctx.stats())
tr = ctx.lm_trace()
for i, r in enumerate(tr):
    print(i + 1, "cost %.6e radius %.3e t %.4f acc %d pcg %d" % (r[0], r[1], r[4], r[5], r[6]))
