"""GPU: recycled Ritz vectors in the preconditioner of global positioning's reduced solves (cg.hpp CgRecycle) — A/B.

Runs gp.hip on synthetic.make_gp_problem inputs with the knob gp_no_recycle set (the block-Jacobi + gauge-deflation
preconditioner of rounds 3 - 6) and cleared (the shipped path), and prints LM / PCG iteration counts, milliseconds
(second of two runs), the solver-path counters, final cost, the per-LM-iteration PCG counts of both runs, and how far the
two end points are apart (Sim(3)-aligned camera centres relative to the extent: max / p99 / median).

Usage: python tools/exp_gp_recycle_gpu.py [--only on|off] [cams tracks seed]...     default: 10000 1000000 0   5000 500000 0"""
import json
import sys
import time

import numpy as np

import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from glomap_amd import estimators, synthetic  # noqa: E402
from glomap_amd._lib import Context  # noqa: E402


def main():
    modes = (1, 0)
    argv = sys.argv[1:]
    if argv[:1] == ["--only"]:  # --only on | off: one mode per process (for rocprofv3 runs)
        modes = (0,) if argv[1] == "on" else (1,)
        argv = argv[2:]
    a = [int(v) for v in argv]
    cases = [tuple(a[i:i + 3]) for i in range(0, len(a), 3)] or [(10000, 1000000, 0), (5000, 500000, 0)]
    ctx = Context()
    for (N, P, seed) in cases:
        p = synthetic.make_gp_problem(N, P, seed=seed)
        res = {}
        for off in modes:
            ctx.set_knob("gp_no_recycle", off)
            opt = estimators.GlobalPositionerOptions()
            best = None
            for _ in range(2):
                ctx.stats(reset=True)
                t0 = time.perf_counter()
                rc, cen, xyz, rep = estimators.gp_solve(p, opt, ctx=ctx)
                ms = (time.perf_counter() - t0) * 1e3
                best = ms if best is None else min(best, ms)
            st = ctx.stats()
            tr = ctx.lm_trace() if hasattr(ctx, "lm_trace") else None
            per = [int(r[-1]) for r in tr] if tr is not None else None
            res[off] = cen
            print(json.dumps(dict(cams=N, tracks=P, seed=seed, recycle=not off, rc=rc, lm=rep["iterations"],
                                  accepted=rep["successful_steps"], pcg=rep["linear_iterations"], final_cost=rep["final_cost"],
                                  ms_incl_h2d=round(best, 1), seconds_solve=rep.get("seconds_solve"),
                                  pcg_recycled=st.get("pcg_recycled"), ritz_harvested=st.get("ritz_harvested"),
                                  pcg_per_lm=per, vs_gt=synthetic.center_distance_stats(cen, p.gt_center))), flush=True)
        if len(res) == 2:
            print(json.dumps(dict(cams=N, seed=seed, end_points_apart=synthetic.center_distance_stats(res[0], res[1]))), flush=True)
    ctx.set_knob("gp_no_recycle", 0)


if __name__ == "__main__":
    main()
