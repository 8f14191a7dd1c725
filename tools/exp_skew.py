"""Skewed-visibility problems of tests/test_edge_cases_gpu.py: GPU (camera lists cut into slices, or with the knob seg_len set
so large that nothing is cut) against the C++ CPU oracle — LM iterations, final cost, pose differences."""
import os, subprocess, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) < 2:
    for env in ({}, {"GSFM_KNOBS": "seg_len=1000000000"}, {"GSFM_KNOBS": "seg_len=256"}):
        print("=== env", env, flush=True)
        subprocess.run([sys.executable, __file__, "run"], env={**os.environ, **env})
    sys.exit(0)
import numpy as np
from glomap_amd import _lib, estimators, so3, synthetic
from oracle import cpu
ctx = _lib.Context(0)
p = synthetic.make_gp_problem(num_cams=80, num_pts=25_000, seed=4, zipf=1.3)
rc, cen, xyz, rep = estimators.gp_solve(p, ctx=ctx)
ok, c_o, X_o, s = cpu.gp_solve(p.num_cams, p.pt_offset, p.obs_cam, p.obs_dir, p.obs_calibrated, p.cam_center, p.pt_xyz)
ext = np.linalg.norm(c_o - c_o.mean(0), axis=1).max()
print("GP gpu lm %d ok %d pcg %d cost %.9e | oracle lm %d ok %d cost %.9e | centres rel %.3e | init cost rel %.2e" % (
    rep["iterations"], rep["successful_steps"], rep["linear_iterations"], rep["final_cost"], s.iterations, s.successful_steps,
    s.final_cost, synthetic.center_errors_after_sim3(cen, c_o).max(), abs(rep["initial_cost"] - s.initial_cost) / s.initial_cost), flush=True)
for tol in (1e-10, 1e-12):
    o = estimators.GlobalPositionerOptions(); o.solver_options.pcg_relative_tolerance = tol; o.solver_options.pcg_max_iterations = 5000
    rc, cen, xyz, rep = estimators.gp_solve(p, o, ctx=ctx)
    print("   pcg tol %.0e: lm %d ok %d cost %.9e centres rel %.3e" % (tol, rep["iterations"], rep["successful_steps"], rep["final_cost"],
          synthetic.center_errors_after_sim3(cen, c_o).max()), flush=True)
for shared in (False, True):
    b = synthetic.make_ba_problem(num_cams=80, num_pts=25_000, seed=4, zipf=1.3, shared_intrinsics=shared)
    r = cpu.ba_solve(b.num_cams, b.pt_offset, b.obs_cam, b.obs_xy, b.cam_intr, b.intr_model, b.fixed_cam, b.cam_q, b.cam_t, b.pt_xyz, b.intr_params)
    for tol in (1e-8, 1e-12):
        o = estimators.BundleAdjusterOptions(); o.solver_options.pcg_relative_tolerance = tol; o.solver_options.pcg_max_iterations = 5000
        rc, q, t, X, intr, rep = estimators.ba_solve(b, o, ctx=ctx)
        ang = np.radians(so3.rotation_angle_deg(so3.quat_to_rotmat(q), so3.quat_to_rotmat(r[1])))
        print("BA shared=%s tol %.0e gpu lm %d ok %d pcg %d cost %.9e | oracle lm %d ok %d cost %.9e | rot max %.3e rad t max %.3e" % (
            shared, tol, rep["iterations"], rep["successful_steps"], rep["linear_iterations"], rep["final_cost"], r[5].iterations,
            r[5].successful_steps, r[5].final_cost, ang.max(), np.abs(t - r[2]).max()), flush=True)
