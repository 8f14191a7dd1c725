"""Global positioning with Ceres' projected line search: the HIP solve against the exact-solve C++ oracle, TRAJECTORIES
and end points, and the oracle against itself (GPU box; round 6).

Per problem: the oracle with its reductions summed forwards and backwards (the same algorithm at two roundings), then the
HIP solve with the reduced systems stopped at 1e-12 / 1e-10 / 1e-8 / 1e-6.  Printed per run: LM iterations (accepted, steps
the line search shortened), final cost, PCG iterations, how many leading LM iterations have the same cost as the
forward-summed oracle to 1e-9 / 1e-6 relative and the same line-search step size to 1e-6, and the Sim(3)-aligned
camera-centre distance to the forward-summed oracle relative to the extent (max / p99 / median).

Usage: python tools/exp_gp_line_search_gpu.py [cams tracks seed]...   default: 150 6000 0  300 20000 1  1000 100000 0  5000 500000 0"""
import json
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from glomap_amd import estimators, synthetic  # noqa: E402
from oracle import cpu  # noqa: E402
from oracle import gp as ogp  # noqa: E402


def prefix(tr, ref, col, rtol):
    n = min(len(tr), len(ref))
    if n == 0:
        return 0
    a, b = tr[:n, col], ref[:n, col]
    bad = np.abs(a - b) > rtol * np.maximum(np.abs(b), 1e-300)
    return int(np.argmax(bad)) if bad.any() else n


def main():
    a = [int(v) for v in sys.argv[1:]]
    cases = [tuple(a[i:i + 3]) for i in range(0, len(a), 3)] or [(150, 6000, 0), (300, 20000, 1), (1000, 100000, 0), (5000, 500000, 0)]
    ctx = estimators.default_context()
    for (N, P, seed) in cases:
        p = synthetic.make_gp_problem(num_cams=N, num_pts=P, seed=seed)
        print(json.dumps(dict(cams=N, tracks=P, seed=seed, observations=int(p.num_obs))), flush=True)
        ref_c = ref_tr = None
        for order in (0, 1):
            t0 = time.time()
            ok, c, X, s = cpu.gp_solve(p.num_cams, p.pt_offset, p.obs_cam, p.obs_dir, p.obs_calibrated, p.cam_center, p.pt_xyz,
                                       ogp.GlobalPositionerOptions(), order=order)
            tr = cpu.lm_trace()
            if ref_c is None:
                ref_c, ref_tr = c, tr
            print(json.dumps(dict(run="oracle reversed" if order else "oracle forward", lm=int(s.iterations), accepted=int(s.successful_steps),
                                  shrunk=int(s.line_search_shrunk), cost=float(s.final_cost), pcg=int(s.linear_iterations),
                                  same_cost_1e9=prefix(tr, ref_tr, 0, 1e-9), same_cost_1e6=prefix(tr, ref_tr, 0, 1e-6),
                                  same_step_size_1e6=prefix(tr, ref_tr, 4, 1e-6),
                                  vs_forward=synthetic.center_distance_stats(c, ref_c), vs_gt=synthetic.center_distance_stats(c, p.gt_center),
                                  seconds=round(time.time() - t0, 1))), flush=True)
        for tol in (1e-12, 1e-10, 1e-8, 1e-6):
            opt = estimators.GlobalPositionerOptions()
            opt.solver_options.pcg_relative_tolerance = tol
            estimators.gp_solve(p, opt, ctx=ctx)  # warm-up (workspace growth)
            t0 = time.time()
            rc, c, X, rep = estimators.gp_solve(p, opt, ctx=ctx)
            sec = time.time() - t0
            tr = ctx.lm_trace()
            print(json.dumps(dict(run=f"gpu pcg {tol:g}", rc=rc, lm=rep["iterations"], accepted=rep["successful_steps"],
                                  shrunk=rep["line_search_shrunk"], trials=rep["line_search_trials"], cost=rep["final_cost"],
                                  pcg=rep["linear_iterations"], same_cost_1e9=prefix(tr, ref_tr, 0, 1e-9),
                                  same_cost_1e6=prefix(tr, ref_tr, 0, 1e-6), same_step_size_1e6=prefix(tr, ref_tr, 4, 1e-6),
                                  vs_forward=synthetic.center_distance_stats(c, ref_c), vs_gt=synthetic.center_distance_stats(c, p.gt_center),
                                  ms=round(1e3 * sec, 1))), flush=True)


if __name__ == "__main__":
    main()
