"""RA -> GP -> BA chained on one synthetic scene through the C ABI, the way GlobalMapper::Solve chains the three estimators
(global_mapper.cc:92-223): every stage starts from the PREVIOUS STAGE'S RESULT, never from ground truth.  Shared by
tests/test_fullsize_gpu.py and tools/exp_chain_gpu.py."""
import numpy as np

from glomap_amd import estimators, so3, synthetic


def gpu_chain(sc, ctx, gp_options=None, ba_options=None):
    rc, rot, rep_ra = estimators.ra_solve(sc.ra, ctx=ctx)
    assert rc == 0
    R = so3.aa_to_rotmat(rot)
    g = synthetic.chain_gp_problem(sc, R)
    rc, cen, xyz, rep_gp = estimators.gp_solve(g, gp_options, ctx=ctx)
    assert rc == 0
    b = synthetic.chain_ba_problem(sc, R, cen, xyz)
    rc, q, t, X, intr, rep_ba = estimators.ba_solve(b, ba_options, ctx=ctx)
    assert rc == 0
    return dict(ra_rot=rot, gp_center=cen, ba_q=q, ba_t=t, ba_intr=intr, rep_ra=rep_ra, rep_gp=rep_gp, rep_ba=rep_ba)


def final_pose_distance(q_a, t_a, q_b, t_b):
    """north_star's two numbers between two sets of final poses: largest rotation distance in rad (no alignment: node 0 is
    the gauge of RA and the constant frame of BA in both chains) and max / p99 / median camera-centre distance after Sim(3)
    alignment relative to the scene extent (BA inherits GP's free scale)."""
    Ra, Rb = so3.quat_to_rotmat(q_a), so3.quat_to_rotmat(q_b)
    ang = np.radians(so3.rotation_angle_deg(Ra, Rb))
    ca = -np.einsum("nji,nj->ni", Ra, t_a)
    cb = -np.einsum("nji,nj->ni", Rb, t_b)
    return float(ang.max()), synthetic.center_distance_stats(ca, cb)
