"""GPU chain RA -> GP -> BA against the frozen oracle chain (tests/golden/make_chain_golden.py), for several PCG tolerances
of the GP / BA reduced solves.  Usage: python tools/exp_chain_gpu.py golden.npz cams tracks [gp_tol:ba_tol ...]"""
import json
import sys
import time

import numpy as np

sys.path.insert(0, ".")
sys.path.insert(0, "tests")
from chain_util import final_pose_distance, gpu_chain  # noqa: E402
from glomap_amd import estimators, so3, synthetic  # noqa: E402
from glomap_amd._lib import Context  # noqa: E402

g = np.load(sys.argv[1])
N, P = int(sys.argv[2]), int(sys.argv[3])
pairs = [tuple(map(float, a.split(":"))) for a in sys.argv[4:]] or [(1e-8, 1e-6), (1e-12, 1e-6), (1e-12, 1e-8)]
sc = synthetic.make_chained_scene(N, P, seed=0)
assert sc.obs_cam.shape[0] == int(g["num_obs"]) and abs(float(np.sum(sc.obs_xy)) / float(g["obs_xy_checksum"]) - 1) < 1e-12
ctx = Context()
print(f"oracle: GP LM {int(g['gp_iterations'])} cost {float(g['gp_final_cost']):.6f}  BA LM {int(g['ba_iterations'])} "
      f"({int(g['ba_successful'])} accepted) cost {float(g['ba_final_cost']):.3f}", flush=True)
for gp_tol, ba_tol in pairs:
    go, bo = estimators.GlobalPositionerOptions(), estimators.BundleAdjusterOptions()
    go.solver_options.pcg_relative_tolerance = gp_tol
    bo.solver_options.pcg_relative_tolerance = ba_tol
    t0 = time.perf_counter()
    r = gpu_chain(sc, ctx, go, bo)
    sec = time.perf_counter() - t0
    d_ra = float(np.radians(so3.rotation_angle_deg(so3.aa_to_rotmat(r["ra_rot"]), so3.aa_to_rotmat(g["ra_rot"]))).max())
    ang, cen = final_pose_distance(r["ba_q"], r["ba_t"], g["ba_q"], g["ba_t"])
    print(json.dumps(dict(gp_tol=gp_tol, ba_tol=ba_tol, ra_rot_vs_oracle_rad=d_ra,
                          gp_lm=r["rep_gp"]["iterations"], gp_pcg=r["rep_gp"]["linear_iterations"], gp_cost=r["rep_gp"]["final_cost"],
                          gp_center_vs_oracle=synthetic.center_distance_stats(r["gp_center"], g["gp_center"]),
                          ba_lm=r["rep_ba"]["iterations"], ba_acc=r["rep_ba"]["successful_steps"], ba_pcg=r["rep_ba"]["linear_iterations"],
                          ba_cost=r["rep_ba"]["final_cost"], final_rot_vs_oracle_rad=ang, final_center_vs_oracle=cen,
                          seconds_incl_host=round(sec, 2))), flush=True)
