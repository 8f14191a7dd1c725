"""GPU solves of BASELINE.json configs[2] (GP) / configs[3] (BA) and a 10k-camera RA against the multithreaded C++ CPU
oracle on the same inputs; prints pose differences, iteration counts and the CPU oracle's wall time on this box.
usage: python tools/exp_fullsize_parity.py [gp] [ba] [ra] [gp_tol]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from glomap_amd import _lib, estimators, so3, synthetic
from oracle import cpu

ctx = _lib.Context(0)
which = sys.argv[1:] or ["gp", "ba", "ra"]
print("host cores", os.cpu_count(), "omp threads", cpu.num_threads(), flush=True)


def rel_center(a, b):
    return synthetic.center_errors_after_sim3(a, b).max()  # already relative to the extent


if "gp" in which or "gp_tol" in which:
    p = synthetic.make_gp_problem(5000, 500_000, seed=0)
    t0 = time.time()
    ok, c_o, X_o, s = cpu.gp_solve(p.num_cams, p.pt_offset, p.obs_cam, p.obs_dir, p.obs_calibrated, p.cam_center, p.pt_xyz)
    print("GP cpu oracle: %.1f s" % (time.time() - t0), s, flush=True)
    tols = (1e-8, 1e-10, 1e-12) if "gp_tol" in which else (1e-8,)
    for tol in tols:
        o = estimators.GlobalPositionerOptions()
        o.solver_options.pcg_relative_tolerance = tol
        o.solver_options.pcg_max_iterations = 5000
        t0 = time.time()
        rc, c, X, rep = estimators.gp_solve(p, o, ctx=ctx)
        print("GP gpu tol %.0e: rc %d lm %d ok %d pcg %d cost %.9e %.0f ms | vs oracle: centres %.3e (relative, Sim3), cost rel %.2e | vs GT %.3e (oracle vs GT %.3e)"
              % (tol, rc, rep["iterations"], rep["successful_steps"], rep["linear_iterations"], rep["final_cost"],
                 (time.time() - t0) * 1e3, rel_center(c, c_o), abs(rep["final_cost"] - s.final_cost) / s.final_cost,
                 rel_center(c, p.gt_center), rel_center(c_o, p.gt_center)), flush=True)

if "ba" in which:
    for shared in (False, True):
        p = synthetic.make_ba_problem(10_000, 1_000_000, seed=0, shared_intrinsics=shared)
        t0 = time.time()
        r = cpu.ba_solve(p.num_cams, p.pt_offset, p.obs_cam, p.obs_xy, p.cam_intr, p.intr_model, p.fixed_cam, p.cam_q, p.cam_t,
                         p.pt_xyz, p.intr_params)
        print("BA shared=%s cpu oracle: %.1f s" % (shared, time.time() - t0), r[5], flush=True)
        t0 = time.time()
        rc, q, t, X, intr, rep = estimators.ba_solve(p, ctx=ctx)
        dt = time.time() - t0
        ang = np.radians(so3.rotation_angle_deg(so3.quat_to_rotmat(q), so3.quat_to_rotmat(r[1])))
        cg = -np.einsum("nji,nj->ni", so3.quat_to_rotmat(q), t)
        co = -np.einsum("nji,nj->ni", so3.quat_to_rotmat(r[1]), r[2])
        ext = np.linalg.norm(co - co.mean(0), axis=1).max()
        print("BA gpu: rc %d lm %d ok %d pcg %d cost %.9e %.0f ms | vs oracle: rot max %.3e rad, centres max %.3e (relative, no alignment) %.3e (Sim3), "
              "intr max %.3e, cost rel %.2e" % (rc, rep["iterations"], rep["successful_steps"], rep["linear_iterations"], rep["final_cost"],
                                               dt * 1e3, ang.max(), np.linalg.norm(cg - co, axis=1).max() / ext, rel_center(cg, co),
                                               np.abs(intr - r[4]).max(), abs(rep["final_cost"] - r[5].final_cost) / r[5].final_cost), flush=True)

if "ra" in which:
    p = synthetic.make_ring_view_graph(10_000, 50, seed=0)
    rep_o = {}
    t0 = time.time()
    ok, rot_o = cpu.ra_estimate_rotations(p.num_nodes, p.edge_i, p.edge_j, p.edge_q, p.edge_weight, p.edge_ninl, p.node_aa0,
                                          p.fixed_node, report=rep_o)
    print("RA cpu oracle: %.2f s" % (time.time() - t0), rep_o, flush=True)
    t0 = time.time()
    rc, rot, rep = estimators.ra_solve(p, ctx=ctx)
    d = np.radians(so3.rotation_angle_deg(so3.aa_to_rotmat(rot), so3.aa_to_rotmat(rot_o)))
    print("RA gpu: rc %d l1 %d irls %d pcg %d %.0f ms | vs oracle max %.3e rad" % (rc, rep["iterations_l1"], rep["iterations_irls"],
                                                                                   rep["linear_iterations"], (time.time() - t0) * 1e3, d.max()), flush=True)
