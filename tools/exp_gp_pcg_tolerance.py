"""How far can the reduced-system solves of global positioning be loosened before the RESULT moves? (CPU study, no GPU.)

Runs the C++/OpenMP oracle (oracle/cpu.py, the exact-solve restatement) on one synthetic GP problem with its PCG stopped
at a relative residual of 1e-14 ("exact"), 1e-8 (what gp.hip uses), ... 1e-1 (the forcing term an inexact-Newton
solver would use) and prints, per tolerance: LM iterations, final cost, the largest camera-centre distance to the exact
run after Sim(3) alignment relative to the scene extent (the quantity the parity bar of 1e-3 is stated on), and the
error against ground truth.  Reduced systems of at most 1536 unknowns are factored densely by the oracle (no PCG): use more than 512 cameras.
Usage: python tools/exp_gp_pcg_tolerance.py [num_cams] [num_pts]    (configs[2]: 5000 500000)"""
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
    p = synthetic.make_gp_problem(num_cams=N, num_pts=P, seed=0)
    print(f"cameras {N} tracks {P} observations {p.num_obs}", flush=True)
    ref = None
    for tol in (1e-14, 1e-8, 1e-6, 1e-4, 1e-3, 1e-2, 1e-1):
        t0 = time.time()
        ok, c, X, s = cpu.gp_solve(p.num_cams, p.pt_offset, p.obs_cam, p.obs_dir, p.obs_calibrated, p.cam_center, p.pt_xyz,
                                   ogp.GlobalPositionerOptions(), pcg_tol=tol)
        sec = time.time() - t0
        if ref is None:
            ref = c
        extent = np.linalg.norm(ref - ref.mean(0), axis=1).max()
        gt = synthetic.center_errors_after_sim3(c, p.gt_center)
        print(json.dumps(dict(pcg_tol=tol, ok=bool(ok), lm_iterations=int(s.iterations), accepted=int(s.successful_steps),
                              final_cost=float(s.final_cost),
                              max_rel_vs_exact=float(synthetic.center_errors_after_sim3(c, ref).max() / extent),
                              median_err_vs_gt=float(np.median(gt)), seconds=round(sec, 1))), flush=True)


if __name__ == "__main__":
    main()
