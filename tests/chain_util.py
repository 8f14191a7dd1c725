"""RA -> GP -> track filters -> normalisation -> staged BA chained on ONE synthetic scene, the way GlobalMapper::Solve chains
the estimators and the processors between them (global_mapper.cc:92-110 rotation averaging, :152-186 global positioning +
FilterTracksByAngle / FilterTrackTriangulationAngle / FilterTracksByReprojection(10 x) / NormalizeReconstruction, :201-223
bundle adjustment positions-only then full): every stage starts from the PREVIOUS STAGE'S RESULT, never from ground truth.

One driver, two backends: the HIP library through the C ABI (GpuBackend) and the CPU oracle (OracleBackend; test
infrastructure — tests/golden/make_chain_golden.py freezes its result).  Shared by tests/test_fullsize_gpu.py,
tools/exp_chain_gpu.py and the golden generator."""
import numpy as np

from glomap_amd import so3, synthetic

MAX_ANGLE_ERROR = 1.0           # InlierThresholdOptions::max_angle_error (glomap/types.h), degrees
MIN_TRIANGULATION_ANGLE = 1.0   # InlierThresholdOptions::min_triangulation_angle, degrees
MAX_REPROJECTION_ERROR = 1e-2   # InlierThresholdOptions::max_reprojection_error, normalised image coordinates


class GpuBackend:
    def __init__(self, ctx, gp_options=None, ba_options=None):
        from glomap_amd import estimators, processors

        self.e, self.p, self.ctx = estimators, processors, ctx
        self.gp_options, self.ba_options = gp_options, ba_options

    def ra(self, prob):
        rc, rot, rep = self.e.ra_solve(prob, ctx=self.ctx)
        assert rc == 0
        return rot, dict(l1=rep["iterations_l1"], irls=rep["iterations_irls"])

    def gp(self, prob):
        rc, cen, xyz, rep = self.e.gp_solve(prob, self.gp_options, ctx=self.ctx)
        assert rc == 0
        self.gp_trace = self.ctx.lm_trace()
        return cen, xyz, dict(iterations=rep["iterations"], successful=rep["successful_steps"], initial_cost=rep["initial_cost"],
                              final_cost=rep["final_cost"], linear_iterations=rep["linear_iterations"],
                              line_search_shrunk=rep.get("line_search_shrunk", 0))

    def _view(self, N, off, ocam, q, t, X, und):
        return self.p.SceneView(N, off, ocam, q, t, X, obs_undist=und)

    def filter_angle(self, N, off, ocam, q, t, X, und):
        return np.asarray(self.p.TrackFilter.FilterTracksByAngle(self._view(N, off, ocam, q, t, X, und), MAX_ANGLE_ERROR, ctx=self.ctx)[0], bool)

    def filter_triangulation(self, N, off, ocam, q, t, X, und):
        return np.asarray(self.p.TrackFilter.FilterTrackTriangulationAngle(self._view(N, off, ocam, q, t, X, und), MIN_TRIANGULATION_ANGLE,
                                                                           ctx=self.ctx)[0], bool)

    def filter_reprojection(self, N, off, ocam, q, t, X, und, thr):
        return np.asarray(self.p.TrackFilter.FilterTracksByReprojection(self._view(N, off, ocam, q, t, X, und), thr, True, ctx=self.ctx)[0], bool)

    def normalize(self, q, t, X):
        t2, X2, _ = self.p.NormalizeReconstruction(q, t, X, ctx=self.ctx)
        return t2, X2

    def ba(self, prob, optimize_rotations):
        import copy

        opt = copy.deepcopy(self.ba_options) if self.ba_options is not None else self.e.BundleAdjusterOptions()
        opt.optimize_rotations = optimize_rotations
        rc, q, t, X, intr, rep = self.e.ba_solve(prob, opt, ctx=self.ctx)
        assert rc == 0
        return q, t, X, intr, dict(iterations=rep["iterations"], successful=rep["successful_steps"], initial_cost=rep["initial_cost"],
                                   final_cost=rep["final_cost"], linear_iterations=rep["linear_iterations"])


class OracleBackend:
    """oracle/cpu.py (exact reduced solves: PCG to 1e-14) + oracle/filters.py."""

    def __init__(self, verbose=False, order=0, gp_pcg_tol=1e-14):
        from oracle import ba as oba
        from oracle import cpu, filters

        self.cpu, self.f, self.oba, self.verbose = cpu, filters, oba, verbose
        # order = 1: every owner-side reduction of GP / BA summed in the opposite order — the same algorithm at another
        # rounding (tools/exp_chain_oracle_scatter.py: how far apart are two roundings of the REFERENCE algorithm?)
        self.order, self.gp_pcg_tol = order, gp_pcg_tol

    def ra(self, p):
        rep = {}
        ok, rot = self.cpu.ra_estimate_rotations(p.num_nodes, p.edge_i, p.edge_j, p.edge_q, p.edge_weight, p.edge_ninl, p.node_aa0,
                                                 p.fixed_node, report=rep)
        assert ok
        return rot, dict(l1=rep["l1_iterations"], irls=rep["irls_iterations"])

    def gp(self, g):
        ok, c, X, s = self.cpu.gp_solve(g.num_cams, g.pt_offset, g.obs_cam, g.obs_dir, g.obs_calibrated, g.cam_center, g.pt_xyz,
                                        verbose=self.verbose, order=self.order, pcg_tol=self.gp_pcg_tol)
        assert ok
        self.gp_trace = self.cpu.lm_trace()
        return c, X, dict(iterations=s.iterations, successful=s.successful_steps, initial_cost=s.initial_cost, final_cost=s.final_cost,
                          linear_iterations=s.linear_iterations, max_linear_residual=s.max_linear_residual,
                          line_search_shrunk=s.line_search_shrunk)

    def filter_angle(self, N, off, ocam, q, t, X, und):
        return self.f.filter_tracks_by_angle(off, ocam, q, t, X, und, MAX_ANGLE_ERROR)[0]

    def filter_triangulation(self, N, off, ocam, q, t, X, und):
        return self.f.filter_tracks_triangulation_angle_grouped(off, ocam, q, t, X, MIN_TRIANGULATION_ANGLE)[0]

    def filter_reprojection(self, N, off, ocam, q, t, X, und, thr):
        return self.f.filter_tracks_by_reprojection(off, ocam, q, t, X, thr, True, obs_undist=und)[0]

    def normalize(self, q, t, X):
        t2, X2, _ = self.f.normalize_reconstruction(q, t, X)
        return t2, X2

    def ba(self, b, optimize_rotations):
        opt = self.oba.BundleAdjusterOptions(optimize_rotations=optimize_rotations)
        r = self.cpu.ba_solve(b.num_cams, b.pt_offset, b.obs_cam, b.obs_xy, b.cam_intr, b.intr_model, b.fixed_cam, b.cam_q, b.cam_t,
                              b.pt_xyz, b.intr_params, options=opt, verbose=self.verbose, order=self.order)
        assert r[0]
        s = r[5]
        return r[1], r[2], r[3], r[4], dict(iterations=s.iterations, successful=s.successful_steps, initial_cost=s.initial_cost,
                                            final_cost=s.final_cost, linear_iterations=s.linear_iterations,
                                            max_linear_residual=s.max_linear_residual)


def _drop_observations(off, keep, *arrays):
    lens = np.diff(off)
    trk = np.repeat(np.arange(len(lens)), lens)
    new_len = np.bincount(trk[keep], minlength=len(lens))
    off2 = np.zeros(len(lens) + 1, dtype=np.int64)
    off2[1:] = np.cumsum(new_len)
    return (off2, *[a[keep] for a in arrays])


def run_chain(sc, be):
    """Returns a dict of stage results: ra_rot, gp_center, ba_q, ba_t, ba_intr, the kept-observation count after each
    filter, and the per-stage reports."""
    out = {}
    rot, out["rep_ra"] = be.ra(sc.ra)
    out["ra_rot"] = rot
    R = so3.aa_to_rotmat(rot)
    N = sc.num_cams
    # global positioning on the bearings oriented by THESE rotations (global_mapper.cc:157-160)
    g = synthetic.chain_gp_problem(sc, R)
    cen, X, out["rep_gp"] = be.gp(g)
    out["gp_center"] = cen
    q = so3.rotmat_to_quat(R)
    t = -np.einsum("nij,nj->ni", R, cen)
    # the three track filters and the normalisation between GP and BA (global_mapper.cc:164-186)
    und = synthetic.unproject_simple_radial(sc.intr[sc.obs_cam], sc.obs_xy)
    off, ocam, xy = sc.pt_offset, sc.obs_cam, sc.obs_xy
    kept = [int(ocam.shape[0])]
    keep = be.filter_angle(N, off, ocam, q, t, X, und)
    off, ocam, xy, und = _drop_observations(off, keep, ocam, xy, und)
    kept.append(int(ocam.shape[0]))
    tkeep = be.filter_triangulation(N, off, ocam, q, t, X, und)
    off, ocam, xy, und = _drop_observations(off, np.repeat(np.asarray(tkeep, bool), np.diff(off)), ocam, xy, und)
    kept.append(int(ocam.shape[0]))
    keep = be.filter_reprojection(N, off, ocam, q, t, X, und, 10 * MAX_REPROJECTION_ERROR)
    off, ocam, xy, und = _drop_observations(off, keep, ocam, xy, und)
    kept.append(int(ocam.shape[0]))
    out["observations_kept"] = kept
    t, X = be.normalize(q, t, X)
    # bundle adjustment, positions only and then with the rotations (global_mapper.cc:201-223), first frame constant
    intr = sc.intr.copy()
    from glomap_amd.flat import CAMERA_SIMPLE_RADIAL, BaProblem

    for stage, optimize_rotations in enumerate((False, True)):
        b = BaProblem(num_cams=N, num_pts=sc.num_pts, num_intr=N, pt_offset=off, obs_cam=ocam, obs_xy=xy,
                      cam_intr=np.arange(N, dtype=np.int32), cam_q=q, cam_t=t, pt_xyz=X,
                      intr_model=np.full(N, CAMERA_SIMPLE_RADIAL, dtype=np.int32), intr_params=intr, fixed_cam=0)
        q, t, X, intr, out[f"rep_ba{stage + 1}"] = be.ba(b, optimize_rotations)
    out["ba_q"], out["ba_t"], out["ba_intr"] = q, t, intr
    return out


def final_pose_distance(q_a, t_a, q_b, t_b):
    """north_star's two numbers between two sets of final poses: largest rotation distance in rad (no alignment: node 0 is
    the gauge of RA and the constant frame of BA in both chains) and max / p99 / median camera-centre distance after Sim(3)
    alignment relative to the scene extent (BA inherits GP's free scale)."""
    Ra, Rb = so3.quat_to_rotmat(q_a), so3.quat_to_rotmat(q_b)
    ang = np.radians(so3.rotation_angle_deg(Ra, Rb))
    ca = -np.einsum("nji,nj->ni", Ra, t_a)
    cb = -np.einsum("nji,nj->ni", Rb, t_b)
    return float(ang.max()), synthetic.center_distance_stats(ca, cb)
