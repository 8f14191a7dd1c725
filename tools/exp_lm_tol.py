"""Deviation from the oracle (exact Schur solves) of the GP / BA solves as a function of the PCG tolerance."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
from glomap_amd import _lib, estimators, so3, synthetic
from oracle import gp as ogp
import test_ba_gpu as TB
ctx = _lib.Context(0)
p = synthetic.make_gp_problem(num_cams=300, num_pts=20000, seed=5, dir_noise=1e-3, outlier_ratio=0.02, uncalibrated_ratio=0.1)
ok, c_o, X_o, summ = ogp.solve(p.num_cams, p.pt_offset, p.obs_cam, p.obs_dir, p.obs_calibrated, p.cam_center, p.pt_xyz, ogp.GlobalPositionerOptions())
print('GP oracle its', summ.iterations, 'cost', summ.final_cost)
for tol in (1e-10, 1e-8, 1e-6, 1e-4, 1e-3, 1e-2):
    o = estimators.GlobalPositionerOptions(); o.solver_options.pcg_relative_tolerance = tol
    t0 = time.time(); rc, c, X, rep = estimators.gp_solve(p, o, ctx=ctx)
    print('GP tol %.0e' % tol, 'its', rep['iterations'], 'pcg', rep['linear_iterations'], '%.1f ms' % ((time.time() - t0) * 1e3), 'cost', rep['final_cost'],
          'dev vs oracle %.3e' % synthetic.center_errors_after_sim3(c, c_o).max(), flush=True)
pb = synthetic.make_ba_problem(num_cams=120, num_pts=6000, seed=6, pixel_noise=0.5, outlier_ratio=0.01, shared_intrinsics=False, intr_noise=0.01)
ok, q_o, t_o, X_o, intr_o, summ = TB._oracle(pb)
print('BA oracle its', summ.iterations, 'cost', summ.final_cost)
for tol in (1e-10, 1e-8, 1e-6, 1e-4, 1e-3, 1e-2):
    o = estimators.BundleAdjusterOptions(); o.solver_options.pcg_relative_tolerance = tol
    t0 = time.time(); rc, q, t, X, intr, rep = estimators.ba_solve(pb, o, ctx=ctx)
    ang = np.radians(so3.rotation_angle_deg(so3.quat_to_rotmat(q), so3.quat_to_rotmat(q_o))).max()
    Cg = -np.einsum('nji,nj->ni', so3.quat_to_rotmat(q), t); Co = -np.einsum('nji,nj->ni', so3.quat_to_rotmat(q_o), t_o)
    print('BA tol %.0e' % tol, 'its', rep['iterations'], 'pcg', rep['linear_iterations'], '%.1f ms' % ((time.time() - t0) * 1e3), 'cost', rep['final_cost'],
          'rot dev %.3e rad' % ang, 'centre dev %.3e' % (np.linalg.norm(Cg - Co, axis=1).max() / np.linalg.norm(Co - Co.mean(0), axis=1).max()), flush=True)
