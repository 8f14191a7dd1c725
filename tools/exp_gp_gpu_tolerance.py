"""GPU: how tight must the reduced-system solves of global positioning be for the HIP solve to follow the exact-solve
oracle's LM trajectory at BASELINE sizes?  (CPU twin of this study: tools/exp_gp_same_minimiser.py.)

For each cached oracle result (tools/make_gp_oracle_cache.py) runs gp.hip with pcg_relative_tolerance 1e-8 ... 1e-13 and
prints LM / PCG iteration counts, milliseconds, final cost and max / p99 / median camera-centre distance to the oracle
(Sim(3)-aligned, relative to the extent — divided ONCE)."""
import glob
import json
import re
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from glomap_amd import estimators, synthetic  # noqa: E402

tols = [float(t) for t in sys.argv[1:]] or [1e-8, 1e-10, 1e-11, 1e-12, 1e-13]
from glomap_amd._lib import Context  # noqa: E402

ctx = Context()
for f in sorted(glob.glob("oracle/_cache/gp_*_s*.npz")):
    m = re.match(r".*gp_(\d+)_(\d+)_s(\d+)(_order1)?\.npz", f)
    N, P, seed = map(int, m.groups()[:3])
    g = np.load(f)
    if m.group(4):  # the oracle with its reductions summed in reverse order (rounding-level variant): same input
        g = dict(g, num_obs=None)
    p = synthetic.make_gp_problem(N, P, seed=seed)
    if g["num_obs"] is not None:
        assert p.num_obs == int(g["num_obs"]) and abs(float(np.sum(p.obs_dir)) - float(g["obs_dir_checksum"])) < 1e-6
    print(f"== {f}: oracle LM {int(g['iterations'])} final cost {float(g['final_cost']):.6f}", flush=True)
    for tol in tols:
        opt = estimators.GlobalPositionerOptions()
        opt.solver_options.pcg_relative_tolerance = tol
        best = None
        for rep_i in range(2):
            t0 = time.perf_counter()
            rc, cen, xyz, rep = estimators.gp_solve(p, opt, ctx=ctx)
            ms = (time.perf_counter() - t0) * 1e3
            best = ms if best is None else min(best, ms)
        st = synthetic.center_distance_stats(cen, g["center"])
        print(json.dumps(dict(cams=N, seed=seed, pcg_tol=tol, rc=rc, lm=rep["iterations"], pcg=rep["linear_iterations"],
                              final_cost=rep["final_cost"], ms_incl_h2d=round(best, 1), vs_oracle=st)), flush=True)
