"""RA -> GP -> BA chained on ONE scene, CPU oracle only: how far apart do two chains end that differ only in how exactly the
reduced systems are solved?  (Sizes the parity bar of tests/test_fullsize_gpu.py::test_chain_* before the GPU sees it.)
Usage: python tools/exp_chain_oracle.py cams tracks [seed]"""
import json
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from glomap_amd import so3, synthetic  # noqa: E402
from oracle import cpu  # noqa: E402


def chain(sc, gp_tol, ba_tol, verbose=False):
    p = sc.ra
    t0 = time.time()
    ok, rot = cpu.ra_estimate_rotations(p.num_nodes, p.edge_i, p.edge_j, p.edge_q, p.edge_weight, p.edge_ninl, p.node_aa0, p.fixed_node)
    R = so3.aa_to_rotmat(rot)
    t1 = time.time()
    g = synthetic.chain_gp_problem(sc, R)
    ok, c, X, sg = cpu.gp_solve(g.num_cams, g.pt_offset, g.obs_cam, g.obs_dir, g.obs_calibrated, g.cam_center, g.pt_xyz, pcg_tol=gp_tol)
    t2 = time.time()
    b = synthetic.chain_ba_problem(sc, R, c, X)
    r = cpu.ba_solve(b.num_cams, b.pt_offset, b.obs_cam, b.obs_xy, b.cam_intr, b.intr_model, b.fixed_cam, b.cam_q, b.cam_t, b.pt_xyz,
                     b.intr_params, pcg_tol=ba_tol, verbose=verbose)
    t3 = time.time()
    sb = r[5]
    Rf = so3.quat_to_rotmat(r[1])
    cf = -np.einsum("nji,nj->ni", Rf, r[2])
    info = dict(gp_lm=sg.iterations, gp_cost=sg.final_cost, ba_lm=sb.iterations, ba_acc=sb.successful_steps, ba_cost0=sb.initial_cost,
                ba_cost=sb.final_cost, ba_maxres=sb.max_linear_residual, sec=[round(t1 - t0, 1), round(t2 - t1, 1), round(t3 - t2, 1)])
    return R, c, Rf, cf, info


def main():
    N, P = int(sys.argv[1]), int(sys.argv[2])
    seed = int(sys.argv[3]) if len(sys.argv) > 3 else 0
    sc = synthetic.make_chained_scene(N, P, seed=seed)
    print(f"cameras {N} tracks {P} observations {sc.obs_cam.shape[0]} edges {sc.ra.num_edges}", flush=True)
    ref = None
    for gp_tol, ba_tol in ((1e-14, 1e-14), (1e-12, 1e-12), (1e-12, 1e-6), (1e-8, 1e-6)):
        R, c, Rf, cf, info = chain(sc, gp_tol, ba_tol)
        if ref is None:
            ref = (R, c, Rf, cf)
        info.update(gp_tol=gp_tol, ba_tol=ba_tol,
                    ra_err_deg_median=float(np.median(synthetic.rotation_errors_deg(R, sc.gt_R))),
                    gp_vs_ref=synthetic.center_distance_stats(c, ref[1]),
                    final_rot_vs_ref_rad=float(np.radians(so3.rotation_angle_deg(Rf, ref[2])).max()),
                    final_center_vs_ref=synthetic.center_distance_stats(cf, ref[3]),
                    final_rot_err_deg_median=float(np.median(synthetic.rotation_errors_deg(Rf, sc.gt_R))),
                    final_center_vs_gt=synthetic.center_distance_stats(cf, sc.gt_center))
        print(json.dumps(info), flush=True)


if __name__ == "__main__":
    main()
