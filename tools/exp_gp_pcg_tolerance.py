"""How far can the reduced-system solves of global positioning be loosened before the RESULT moves? (CPU study, no GPU.)

Runs the C++/OpenMP oracle (oracle/cpu.py, the exact-solve restatement) on one synthetic GP problem with its PCG stopped
at a relative residual of 1e-14 ("exact"), 1e-8 (what gp.hip uses), ... 1e-1 (the forcing term an inexact-Newton
solver would use) and prints, per tolerance: LM iterations, final cost, the largest camera-centre distance to the exact
run after Sim(3) alignment relative to the scene extent (the quantity the parity bar of 1e-3 is stated on), and the
error against ground truth.  Reduced systems of at most 1536 unknowns are factored densely by the oracle (no PCG): use more than 512 cameras.
Usage: python tools/exp_gp_pcg_tolerance.py [num_cams] [num_pts] [gp|ba] [seed] [nols]   (configs[2]: 5000 500000 gp; configs[3]: 10000 1000000 ba)
Round 6: the LM loop now carries Ceres' projected line search (oracle/lm.py header); `nols` switches it off (the loop of
rounds 1 - 5).  The row "1e-14 reversed" is the oracle against itself with every reduction summed in the opposite order.
With `ba` the same sweep runs on bundle adjustment (first frame constant in every run, so no alignment: largest rotation
difference in rad and largest centre difference relative to the extent against the 1e-14 run)."""
import json
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from glomap_amd import synthetic  # noqa: E402
from oracle import cpu  # noqa: E402
from oracle import gp as ogp  # noqa: E402


def main_ba(N, P):
    from glomap_amd import so3

    p = synthetic.make_ba_problem(N, P, seed=0)
    print(f"BA: cameras {N} tracks {P} observations {p.num_obs}", flush=True)
    ref = None
    for tol in (1e-14, 1e-8, 1e-6, 1e-4, 1e-3, 1e-2, 1e-1):
        t0 = time.time()
        ok, q, t, X, intr, s = cpu.ba_solve(p.num_cams, p.pt_offset, p.obs_cam, p.obs_xy, p.cam_intr, p.intr_model, p.fixed_cam,
                                            p.cam_q, p.cam_t, p.pt_xyz, p.intr_params, pcg_tol=tol)
        sec = time.time() - t0
        R = so3.quat_to_rotmat(q)
        c = -np.einsum("nji,nj->ni", R, t)
        if ref is None:
            ref = (R, c)
        extent = np.linalg.norm(ref[1] - ref[1].mean(0), axis=1).max()
        ang = np.radians(so3.rotation_angle_deg(R, ref[0]))
        gt = synthetic.rotation_errors_deg(R, so3.quat_to_rotmat(p.gt_q))
        print(json.dumps(dict(pcg_tol=tol, ok=bool(ok), lm_iterations=int(s.iterations), accepted=int(s.successful_steps),
                              final_cost=float(s.final_cost), max_rot_vs_exact_rad=float(ang.max()),
                              max_center_rel_vs_exact=float(np.linalg.norm(c - ref[1], axis=1).max() / extent),
                              median_rot_err_vs_gt_deg=float(np.median(gt)), seconds=round(sec, 1))), flush=True)


def main():
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
    P = int(sys.argv[2]) if len(sys.argv) > 2 else 100_000
    if len(sys.argv) > 3 and sys.argv[3] == "ba":
        return main_ba(N, P)
    seed = int(sys.argv[4]) if len(sys.argv) > 4 else 0
    ls = not (len(sys.argv) > 5 and sys.argv[5] == "nols")
    p = synthetic.make_gp_problem(num_cams=N, num_pts=P, seed=seed)
    print(f"cameras {N} tracks {P} observations {p.num_obs} seed {seed} line_search {ls}", flush=True)
    ref = None
    for tol, order in ((1e-14, 0), (1e-14, 1), (1e-13, 0), (1e-12, 0), (1e-11, 0), (1e-10, 0), (1e-8, 0), (1e-6, 0), (1e-4, 0), (1e-2, 0)):
        t0 = time.time()
        opt = ogp.GlobalPositionerOptions()
        opt.lm.line_search = ls
        ok, c, X, s = cpu.gp_solve(p.num_cams, p.pt_offset, p.obs_cam, p.obs_dir, p.obs_calibrated, p.cam_center, p.pt_xyz,
                                   opt, pcg_tol=tol, order=order)
        sec = time.time() - t0
        if ref is None:
            ref = c
        gt = synthetic.center_errors_after_sim3(c, p.gt_center)
        # (round 5: this column used to be divided by the extent a second time — center_errors_after_sim3 already is relative)
        print(json.dumps(dict(pcg_tol=tol, order=order, ok=bool(ok), lm_iterations=int(s.iterations), accepted=int(s.successful_steps),
                              shrunk=int(s.line_search_shrunk),
                              pcg_iterations=int(s.linear_iterations), final_cost=float(s.final_cost),
                              vs_exact=synthetic.center_distance_stats(c, ref),
                              median_err_vs_gt=float(np.median(gt)), seconds=round(sec, 1))), flush=True)


if __name__ == "__main__":
    main()
