"""GPU chain (tests/chain_util.py run_chain: RA -> GP -> filters -> normalise -> staged BA) against the frozen oracle chain
(tests/golden/make_chain_golden.py), for several PCG tolerances of the GP / BA reduced solves.
Usage: python tools/exp_chain_gpu.py golden.npz cams tracks [gp_tol:ba_tol ...]   (no pairs: the library defaults)"""
import json
import sys
import time

import numpy as np

sys.path.insert(0, ".")
sys.path.insert(0, "tests")
from chain_util import GpuBackend, final_pose_distance, run_chain  # noqa: E402
from glomap_amd import estimators, so3, synthetic  # noqa: E402
from glomap_amd._lib import Context  # noqa: E402

g = np.load(sys.argv[1])
N, P = int(sys.argv[2]), int(sys.argv[3])
pairs = [tuple(map(float, a.split(":"))) for a in sys.argv[4:]] or [None]
sc = synthetic.make_chained_scene(N, P, seed=0)
assert sc.obs_cam.shape[0] == int(g["num_obs"]) and abs(float(np.sum(sc.obs_xy)) / float(g["obs_xy_checksum"]) - 1) < 1e-12
ctx = Context()
print(f"oracle: GP LM {int(g['gp_iterations'])} cost {float(g['gp_final_cost']):.6f} | kept {g['observations_kept'].tolist()} | BA1 LM "
      f"{int(g['ba1_iterations'])} ({int(g['ba1_successful'])}) cost {float(g['ba1_final_cost']):.3f} (max relres {float(g['ba1_max_linear_residual']):.1e})"
      f" | BA2 LM {int(g['ba2_iterations'])} ({int(g['ba2_successful'])}) cost {float(g['ba2_final_cost']):.3f} (max relres "
      f"{float(g['ba2_max_linear_residual']):.1e})", flush=True)
for pr in pairs:
    go = bo = None
    if pr is not None:
        go, bo = estimators.GlobalPositionerOptions(), estimators.BundleAdjusterOptions()
        go.solver_options.pcg_relative_tolerance = pr[0]
        bo.solver_options.pcg_relative_tolerance = pr[1]
    t0 = time.perf_counter()
    r = run_chain(sc, GpuBackend(ctx, go, bo))
    sec = time.perf_counter() - t0
    d_ra = float(np.radians(so3.rotation_angle_deg(so3.aa_to_rotmat(r["ra_rot"]), so3.aa_to_rotmat(g["ra_rot"]))).max())
    ang, cen = final_pose_distance(r["ba_q"], r["ba_t"], g["ba_q"], g["ba_t"])
    print(json.dumps(dict(tolerances=pr, ra_rot_vs_oracle_rad=d_ra, gp=r["rep_gp"],
                          gp_center_vs_oracle=synthetic.center_distance_stats(r["gp_center"], g["gp_center"]),
                          observations_kept=r["observations_kept"], ba1=r["rep_ba1"], ba2=r["rep_ba2"],
                          final_rot_vs_oracle_rad=ang, final_center_vs_oracle=cen, seconds_incl_host=round(sec, 2))), flush=True)
