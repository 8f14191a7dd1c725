"""GPU vs C++ oracle: bundle adjustment with a 12-parameter camera model (FULL_OPENCV, the 16-wide unit ba_wide.hip /
orc_ba_wide.cc) at BASELINE sizes — one camera per image, start = ground truth + noise, the library's default options.

Usage: python tools/exp_ba_wide_fullsize.py [cams tracks groups]...      default: 2000 200000 20   10000 1000000 100
(groups = physical cameras shared by the images round-robin; 0 = one camera per image)"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from glomap_amd import estimators, so3, synthetic  # noqa: E402
from glomap_amd._lib import Context  # noqa: E402
from oracle import cpu  # noqa: E402


def main():
    a = [int(v) for v in sys.argv[1:]]
    cases = [tuple(a[i:i + 3]) for i in range(0, len(a), 3)] or [(2000, 200000, 20), (10000, 1000000, 100)]
    ctx = Context()
    for (N, P, G) in cases:
        p = synthetic.make_ba_problem_wide(N, P, "full_opencv", seed=0, num_intr_groups=G)
        best = None
        for _ in range(2):
            ctx.stats(reset=True)
            t0 = time.perf_counter()
            rc, q, t, X, intr, rep = estimators.ba_solve(p, ctx=ctx)
            ms = (time.perf_counter() - t0) * 1e3
            best = ms if best is None else min(best, ms)
        st = ctx.stats()
        t0 = time.perf_counter()
        r = cpu.ba_solve(p.num_cams, p.pt_offset, p.obs_cam, p.obs_xy, p.cam_intr, p.intr_model, p.fixed_cam, p.cam_q, p.cam_t,
                         p.pt_xyz, p.intr_params)
        sec = time.perf_counter() - t0
        s = r[5]
        Rg, Ro = so3.quat_to_rotmat(q), so3.quat_to_rotmat(r[1])
        ang = np.radians(so3.rotation_angle_deg(Rg, Ro))
        cg, co = -np.einsum("nji,nj->ni", Rg, t), -np.einsum("nji,nj->ni", Ro, r[2])
        ext = np.linalg.norm(co - co.mean(0), axis=1).max()
        print(json.dumps(dict(cams=N, tracks=P, intrinsics_blocks=int(p.num_intr), observations=int(p.num_obs), rc=rc, gpu_lm=rep["iterations"], gpu_accepted=rep["successful_steps"],
                              gpu_pcg=rep["linear_iterations"], gpu_ms_incl_h2d=round(best, 1), gpu_final_cost=rep["final_cost"],
                              initial_cost=(rep["initial_cost"], s.initial_cost), oracle_lm=s.iterations, oracle_final_cost=s.final_cost,
                              oracle_seconds=round(sec, 1), oracle_max_linear_residual=s.max_linear_residual, solver_paths=st,
                              max_rotation_rad=float(ang.max()), max_centre_rel=float(np.linalg.norm(cg - co, axis=1).max() / ext),
                              max_intrinsics_abs=float(np.abs(intr - r[4]).max()))), flush=True)


if __name__ == "__main__":
    main()
