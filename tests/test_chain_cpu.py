"""The chained-pipeline driver of tests/chain_util.py (RA -> GP -> track filters -> normaliser -> staged BA, the order of
global_mapper.cc:92-223) on the CPU oracle backend, at a size that runs in seconds: the driver, the oracle's filters and the
scene generator are test infrastructure too, and the GPU suite only meets them at full size."""
import numpy as np

from chain_util import OracleBackend, final_pose_distance, run_chain
from glomap_amd import so3, synthetic


def test_oracle_chain_recovers_the_scene():
    sc = synthetic.make_chained_scene(120, 6000, seed=3, num_succ=12)
    r = run_chain(sc, OracleBackend())
    kept = r["observations_kept"]
    assert kept[0] == sc.obs_cam.shape[0] and kept[0] > kept[1] >= kept[2] >= kept[3] > 0.9 * kept[0]  # ~1 % gross outliers go
    # rotation averaging: 1 degree of noise per edge, 12 edges per node
    R_ra = so3.aa_to_rotmat(r["ra_rot"])
    assert np.median(synthetic.rotation_errors_deg(R_ra, sc.gt_R)) < 0.5
    # the chain ends at the scene: rotations to a few hundredths of a degree, centres to 1e-3 of the extent
    Rf = so3.quat_to_rotmat(r["ba_q"])
    cf = -np.einsum("nji,nj->ni", Rf, r["ba_t"])
    assert np.median(synthetic.rotation_errors_deg(Rf, sc.gt_R)) < 0.05
    assert np.median(synthetic.center_errors_after_sim3(cf, sc.gt_center)) < 2e-3
    assert r["rep_ba1"]["final_cost"] < r["rep_ba1"]["initial_cost"] and r["rep_ba2"]["final_cost"] < r["rep_ba2"]["initial_cost"]
    # the exact-solve oracle really solved its reduced systems
    assert r["rep_gp"]["max_linear_residual"] < 1e-8 and r["rep_ba2"]["max_linear_residual"] < 1e-8
    # and the comparison helper is a metric: zero on itself
    ang, st = final_pose_distance(r["ba_q"], r["ba_t"], r["ba_q"], r["ba_t"])
    assert ang < 1e-7 and st["max"] < 1e-12


def test_unproject_inverts_project_simple_radial():
    rng = np.random.default_rng(0)
    par = np.tile(np.array([1200.0, 640.0, 480.0, 0.02]), (1000, 1))
    xc = np.column_stack([rng.uniform(-0.5, 0.5, 1000), rng.uniform(-0.4, 0.4, 1000), np.ones(1000)])
    xy = synthetic.project_simple_radial(par, xc)
    b = synthetic.unproject_simple_radial(par, xy)
    assert np.abs(b / b[:, 2:3] - xc).max() < 1e-12
