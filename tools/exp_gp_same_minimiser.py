"""Does the distance between two global-positioning runs come from WHERE they stop or from WHAT they solve? (CPU study.)

Global positioning stops on function_tolerance 1e-5 (optimization_base.h:22), from a random start, after ~40 LM
iterations: a handful
of poorly constrained cameras are still moving when the cost change drops below the tolerance.  Two runs that differ
in rounding (summation order, PCG tolerance of the reduced solves) then stop at visibly different places although they
minimise the same function.  This script separates the two effects on the C++ oracle alone:

    for function_tolerance in (1e-5 [reference default], 1e-10 [tight, max 400 LM iterations]):
        run PCG tolerance 1e-14 (exact), 1e-8 (what gp.hip uses), reversed summation order (order=1)
        print max / p99 / median centre distance to the exact run, Sim(3)-aligned, relative to the extent (ONCE)

Usage: python tools/exp_gp_same_minimiser.py [num_cams] [num_pts] [seed]"""
import json
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from glomap_amd import synthetic  # noqa: E402
from oracle import cpu  # noqa: E402
from oracle import gp as ogp  # noqa: E402


def main():
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
    P = int(sys.argv[2]) if len(sys.argv) > 2 else 100_000
    seed = int(sys.argv[3]) if len(sys.argv) > 3 else 0
    p = synthetic.make_gp_problem(num_cams=N, num_pts=P, seed=seed)
    print(f"cameras {N} tracks {P} observations {p.num_obs} seed {seed}", flush=True)
    for ftol, max_it in ((1e-5, 100), (1e-10, 400)):
        ref = None
        for pcg_tol, order in ((1e-14, 0), (1e-8, 0), (1e-14, 1)):
            opt = ogp.GlobalPositionerOptions()
            opt.lm.function_tolerance = ftol
            opt.lm.max_num_iterations = max_it
            t0 = time.time()
            ok, c, X, s = cpu.gp_solve(p.num_cams, p.pt_offset, p.obs_cam, p.obs_dir, p.obs_calibrated, p.cam_center, p.pt_xyz,
                                       opt, pcg_tol=pcg_tol, order=order)
            sec = time.time() - t0
            if ref is None:
                ref = c
            st = synthetic.center_distance_stats(c, ref)
            gt = synthetic.center_distance_stats(c, p.gt_center)
            print(json.dumps(dict(function_tolerance=ftol, pcg_tol=pcg_tol, order=order, ok=bool(ok), lm_iterations=int(s.iterations),
                                  final_cost=float(s.final_cost), termination=str(getattr(s, "termination", "")),
                                  extent=synthetic.scene_extent(c), vs_exact=st, vs_gt=gt, seconds=round(sec, 1))), flush=True)


if __name__ == "__main__":
    main()
