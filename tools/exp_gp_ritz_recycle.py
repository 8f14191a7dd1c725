"""Recycled Ritz vectors as an additive coarse space for the reduced solves of global positioning  (CPU study, C++ oracle).

Round 6: with Ceres' projected line search in the loop a configs[3] GP solve takes 2 204 PCG iterations (block-Jacobi, the
four gauge modes deflated, tolerance 1e-10), 85 % of them in the thirty mid-trajectory LM iterations where the preconditioned
reduced system has a TAIL of eigenvalues 0.01 ... 0.1 under a bulk in [0.2, 2] (tools/ritz_from_cg_trace.py) — no closed-form
modes, no cluster structure (the second level of DESIGN.md 4.2 fails its definiteness check on this scene).  What is
available for free is the Lanczos process inside every PCG solve: its small Ritz pairs (theta, u) approximate exactly those
eigenvectors, and consecutive LM steps change the system slowly (most steps are shortened to a fifth by the line search).
This script runs the C++ oracle with its experimental ORC_RECYCLE switch (oracle/csrc/orc_lm.hpp, pcg): after a solve of at
least ORC_RECYCLE_MINIT iterations the converged Ritz pairs below ORC_RECYCLE_CUT are stored (at most ORC_RECYCLE vectors,
the largest Ritz value evicted first, dropped when the trust-region radius has moved by more than ORC_RECYCLE_RADIUS or after
ORC_RECYCLE_AGE solves) and the next solves run with  M2^-1 = M^-1 + sum u u^T / theta  — same system, same tolerance.

    python tools/exp_gp_ritz_recycle.py [cams tracks]            default 10000 1000000 (4 - 5 minutes per variant on 8 cores)

Prints, per variant, the PCG count of every LM iteration, the totals, and the distance of the end point from the plain run.
Measured (profiles/r06_gp_ritz_recycle_cpu.txt): 2 204 -> 1 646 iterations (k_max 16), 1 583 with 32 vectors; without the
staleness rule the last ten LM iterations get WORSE than plain (a stale, too small theta lifts a mode far above the bulk).
The product's version of this is cg.hpp CgRecycle / ritz.hpp / gp.hip harvest()."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r"""
import json, sys, time
import numpy as np
sys.path.insert(0, {root!r})
from glomap_amd import synthetic
from oracle import cpu, gp as ogp
N, P, out = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
p = synthetic.make_gp_problem(num_cams=N, num_pts=P, seed=0)
t0 = time.time()
ok, c, X, s = cpu.gp_solve(p.num_cams, p.pt_offset, p.obs_cam, p.obs_dir, p.obs_calibrated, p.cam_center, p.pt_xyz,
                           ogp.GlobalPositionerOptions(), pcg_tol=1e-10, verbose=True, deflate=True)
np.save(out, c)
print("RESULT " + json.dumps(dict(lm=int(s.iterations), accepted=int(s.successful_steps), pcg=int(s.linear_iterations),
                                   final_cost=float(s.final_cost), seconds=round(time.time() - t0, 1))))
"""

VARIANTS = [
    ("plain (block-Jacobi + gauge deflation)", {}),
    ("recycle 16, first in first out, no staleness rule", {"ORC_RECYCLE": "16", "ORC_RECYCLE_RADIUS": "1e300", "ORC_RECYCLE_AGE": "1000000",
                                                          "ORC_RECYCLE_MINIT": "4"}),
    ("recycle 16 (radius 3, age 6, >= 25 iterations)", {"ORC_RECYCLE": "16"}),
    ("recycle 16, radius ratio 10", {"ORC_RECYCLE": "16", "ORC_RECYCLE_RADIUS": "10"}),
    ("recycle 32, 12 per solve, cut 0.4, ratio 10, age 8", {"ORC_RECYCLE": "32", "ORC_RECYCLE_PER": "12", "ORC_RECYCLE_CUT": "0.4",
                                                             "ORC_RECYCLE_RADIUS": "10", "ORC_RECYCLE_AGE": "8"}),
]


def main():
    import numpy as np

    sys.path.insert(0, ROOT)
    from glomap_amd import synthetic

    N, P = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (10000, 1000000)
    base = None
    for name, env_add in VARIANTS:
        env = dict(os.environ)
        for k in list(env):
            if k.startswith("ORC_RECYCLE"):
                env.pop(k)
        env.update(env_add)
        out = f"/tmp/exp_ritz_{abs(hash(name)) % 10**8}.npy"
        pr = subprocess.run([sys.executable, "-c", WORKER.format(root=ROOT), str(N), str(P), out], env=env, capture_output=True, text=True)
        per = [ln.split()[7] for ln in pr.stderr.splitlines() if ln.startswith("[orc lm] it") and " pcg " in ln]
        res = [json.loads(ln[7:]) for ln in pr.stdout.splitlines() if ln.startswith("RESULT ")]
        if not res:
            raise SystemExit(pr.stdout[-2000:] + pr.stderr[-4000:])
        c = np.load(out)
        if base is None:
            base = c
        print(json.dumps(dict(variant=name, cams=N, tracks=P, **res[0], pcg_per_lm=[int(v) for v in per],
                              vs_plain=synthetic.center_distance_stats(c, base))), flush=True)


if __name__ == "__main__":
    main()
