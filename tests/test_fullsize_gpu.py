"""BASELINE.json's full-size configurations.

Two kinds of checks at configs[1] / configs[2] / configs[3] size:
  * against the ORACLE on the same inputs: the multithreaded C++ restatement (oracle/cpu.py; LM decisions and block
    elimination identical to the numpy oracle, reduced systems solved to 1e-14 — cross-validated in
    tests/test_oracle_cpu.py) finishes these sizes in seconds to minutes on the host cores of the GPU box, so the
    poses are compared at north_star's tolerance: rotations <= 1e-4 rad, camera centres <= 1e-3 relative;
  * size-independent properties: convergence, ground-truth recovery at the synthetic noise level, monotone cost,
    idempotence (a second solve started from the solution stops at once and does not move it), agreement of the
    RA linear solvers."""
import numpy as np
import pytest

from glomap_amd import estimators, so3, synthetic

pytestmark = pytest.mark.gpu


def test_ra_config2_full(gsfm_ctx):
    """configs[1]: 1k cameras / 50k edges, 5 % outlier edges, 1 degree noise."""
    p = synthetic.make_ring_view_graph(1000, 50, seed=0)
    rc, rot, rep = estimators.ra_solve(p, ctx=gsfm_ctx)
    assert rc == 0 and rep["iterations_l1"] >= 1 and rep["iterations_irls"] >= 1
    err = synthetic.rotation_errors_deg(so3.aa_to_rotmat(rot), p.gt_R)
    assert np.median(err) < 0.5 and err.max() < 3.0  # rotation_averager_test.cc:309-310 allows 3 degrees
    # dense direct solver (N <= 2048) and the PCG solver large graphs use agree
    rc, rot_it, _ = estimators.ra_solve(p, estimators.RotationEstimatorOptions(force_iterative=True), ctx=gsfm_ctx)
    assert rc == 0
    d = np.radians(so3.rotation_angle_deg(so3.aa_to_rotmat(rot), so3.aa_to_rotmat(rot_it)))
    assert d.max() < 1e-4
    # idempotence: restarted from the solution, IRLS stops after its first step (mean step < 1e-3 rad)
    p2 = type(p)(**{**p.__dict__, "node_aa0": rot})
    rc, rot2, rep2 = estimators.ra_solve(
        p2, estimators.RotationEstimatorOptions(skip_initialization=True, max_num_l1_iterations=0), ctx=gsfm_ctx)
    assert rc == 0 and rep2["iterations_irls"] == 1
    d = np.radians(so3.rotation_angle_deg(so3.aa_to_rotmat(rot2), so3.aa_to_rotmat(rot)))
    assert np.mean(d) < 1e-3


def test_gp_config3_full(gsfm_ctx):
    """configs[2]: 5k cameras / 500k tracks / ~3M observations, random initialisation (seed 1)."""
    p = synthetic.make_gp_problem(5000, 500_000, seed=0)
    rc, cen, xyz, rep = estimators.gp_solve(p, ctx=gsfm_ctx)
    assert rc == 0 and rep["termination"] == 0
    assert rep["final_cost"] < 1e-3 * rep["initial_cost"]
    err = synthetic.center_errors_after_sim3(cen, p.gt_center)  # relative to the extent of the ground truth
    # ray noise 1e-3: the numpy oracle's own end point on this input sits at a median of 1.028e-3 from the ground truth
    # (profiles/r06_gp_line_search_gpu_vs_oracle.txt, "oracle forward", 5 000 / 500 000 seed 0), and LM paths that differ in the
    # last bits end 1e-5 of that apart (the reference's own scatter) — the bar is the noise level, not 1e-3 flat
    assert np.median(err) < 1.2e-3
    # idempotence: from the solution (no random re-draw; the per-observation scales are re-derived from the
    # geometry, gp.cc:300-305, so this is not a bit-exact restart) the solver stops quickly at the same cost
    p2 = type(p)(**{**p.__dict__, "cam_center": cen, "pt_xyz": xyz})
    opt = estimators.GlobalPositionerOptions(generate_random_positions=False, generate_random_points=False,
                                            generate_scales=False)
    rc, cen2, xyz2, rep2 = estimators.gp_solve(p2, opt, ctx=gsfm_ctx)
    assert rc == 0 and rep2["iterations"] <= 12
    assert rep2["final_cost"] <= rep["final_cost"] * (1 + 1e-4)
    assert synthetic.center_errors_after_sim3(cen2, cen).max() < 1e-3


def test_ba_config4_full(gsfm_ctx):
    """configs[3] on one GPU: 10k cameras / 1M tracks / ~5M observations, one SIMPLE_RADIAL camera per image."""
    p = synthetic.make_ba_problem(10_000, 1_000_000, seed=0, shared_intrinsics=False)
    rc, q, t, X, intr, rep = estimators.ba_solve(p, ctx=gsfm_ctx)
    assert rc == 0 and rep["termination"] == 0
    assert rep["final_cost"] < 0.5 * rep["initial_cost"]
    rot_err = synthetic.rotation_errors_deg(so3.quat_to_rotmat(q), so3.quat_to_rotmat(p.gt_q))
    assert np.median(rot_err) < 0.05  # start: 0.5 degree noise per camera
    # the constant frame is untouched (ba.cc:261-266)
    assert np.array_equal(q[p.fixed_cam], p.cam_q[p.fixed_cam]) and np.array_equal(t[p.fixed_cam], p.cam_t[p.fixed_cam])
    # restart from the solution: the cost is reproduced exactly, never goes up, and only creeps further
    # (the first solve stopped on function_tolerance 1e-5; a fresh trust region makes a little more progress)
    p2 = type(p)(**{**p.__dict__, "cam_q": q, "cam_t": t, "pt_xyz": X, "intr_params": intr})
    rc, q2, t2, X2, intr2, rep2 = estimators.ba_solve(p2, ctx=gsfm_ctx)
    assert rc == 0
    assert abs(rep2["initial_cost"] - rep["final_cost"]) <= 1e-9 * rep["final_cost"]
    assert rep2["final_cost"] <= rep["final_cost"] * (1 + 1e-9)
    assert rep2["final_cost"] >= 0.97 * rep["final_cost"]


def _extent(c):
    return synthetic.scene_extent(c)


def _gp_full_size_problem(ncam, npts, seed):
    # seed 1 carries uncalibrated cameras (the half-weight loss branch of gp.cc:212-231) on top of the 2 % outlier rays
    return synthetic.make_gp_problem(ncam, npts, seed=seed, uncalibrated_ratio=0.1 if seed == 1 else 0.0)


class _FrozenSummary:
    def __init__(self, g, order):
        self.iterations, self.final_cost = int(g[f"iterations_{order}"]), float(g[f"final_cost_{order}"])
        self.initial_cost, self.max_linear_residual = float(g[f"initial_cost_{order}"]), float(g[f"max_linear_residual_{order}"])
        self.successful_steps, self.line_search_shrunk = int(g[f"successful_{order}"]), int(g[f"line_search_shrunk_{order}"])


def _same_prefix(tr, ref, col, rtol):
    """Number of leading LM iterations in which column `col` of two LM traces (gsfm_ctx_lm_trace / oracle.cpu.lm_trace: cost |
    radius | model change | candidate cost | line-search step size | accepted | linear iterations) agrees to rtol."""
    n = min(len(tr), len(ref))
    bad = np.abs(tr[:n, col] - ref[:n, col]) > rtol * np.maximum(np.abs(ref[:n, col]), 1e-300)
    return int(np.argmax(bad)) if bad.any() else n


def _gp_parity(tag, p, ctx, lm_kw=None, frozen=None):
    """HIP solve vs the exact-solve C++ oracle on the same input and the same std::mt19937 start — end points AND trajectories —
    next to the oracle against ITSELF with every reduction summed in the opposite order (the same algorithm at another
    rounding).  Distances: camera centres, Sim(3)-aligned, relative to the extent of the reference solution — divided ONCE.

    Round 6.  The reference's GP problem is bounds-constrained (every scale has a lower bound, gp.cc:204,373), so Ceres runs a
    projected Armijo line search on every LM step (oracle/lm.py header); rounds 1 - 5 had left it out of oracle and product
    (VERDICT r5).  With it in both, the question "how far is the HIP solve from the reference" has two different answers:
      * on inputs where the reference algorithm's end point is DEFINED — the oracle summed forwards and backwards ends in the
        same place — the HIP solve has to end there too (north_star's 1e-3 on the worst camera, same iteration counts);
      * on the full-size benchmark inputs the algorithm is chaotic for ANY implementation: the two oracle roundings agree on the
        first ~11 LM iterations to nine digits, then drift apart ~10 x per iteration and end 2e-3 (p99) / 2e-5 (median) /
        3e-2 (one camera) apart with different iteration counts (profiles/r06_gp_line_search_gpu_vs_oracle.txt;
        tools/exp_gp_pcg_tolerance.py: every solver tolerance from 1e-6 to 1e-14 lands inside that scatter).  There parity
        can only mean: the same trajectory for as long as the reference follows its own, and an end point inside the
        reference's own scatter.  The FINAL poses after bundle adjustment are defined again (the chain tests below).

    frozen: fixture under tests/golden/ with both oracle variants (tests/golden/make_gp_c4_golden.py; the oracle's reductions
    are thread-count independent, so the fixture is what the box would compute; the input is pinned by its checksums)."""
    from oracle import cpu
    from oracle import gp as ogp

    opt = estimators.GlobalPositionerOptions()
    oopt = ogp.GlobalPositionerOptions()
    for k, v in (lm_kw or {}).items():
        setattr(opt.solver_options, k, v)
        setattr(oopt.lm, k, v)
    g = None
    if frozen is not None:
        import os

        assert not lm_kw
        g = np.load(os.path.join(os.path.dirname(__file__), "golden", frozen))
        assert p.num_obs == int(g["num_obs"]) and int(np.sum(p.obs_cam.astype(np.int64))) == int(g["obs_cam_checksum"])
        assert abs(float(np.sum(p.obs_dir)) - float(g["obs_dir_checksum"])) < 1e-6
        assert int(np.sum(p.obs_calibrated.astype(np.int64))) == int(g["calibrated_checksum"])
    rc, cen, xyz, rep = estimators.gp_solve(p, opt, ctx=ctx)
    assert rc == 0
    tr = ctx.lm_trace()
    assert len(tr) == rep["iterations"]
    o = []
    for order in (0, 1):
        if g is not None:
            c_o, s, tr_o = g[f"center_{order}"], _FrozenSummary(g, order), g[f"trace_{order}"]
        else:
            ok, c_o, X_o, s = cpu.gp_solve(p.num_cams, p.pt_offset, p.obs_cam, p.obs_dir, p.obs_calibrated, p.cam_center, p.pt_xyz, oopt,
                                           order=order)
            assert ok
            tr_o = cpu.lm_trace()
        assert s.max_linear_residual < 1e-8  # the oracle's reduced solves really were exact (true residual)
        assert abs(rep["initial_cost"] - s.initial_cost) <= 1e-12 * s.initial_cost  # identical random start
        o.append((c_o, s, tr_o))
    res = dict(cen=cen, rep=rep, trace=tr, oracle=o, self=synthetic.center_distance_stats(o[1][0], o[0][0]),
               self_prefix=_same_prefix(o[1][2], o[0][2], 0, 1e-6),
               vs=[synthetic.center_distance_stats(cen, o[k][0]) for k in (0, 1)],
               prefix=[_same_prefix(tr, o[k][2], 0, 1e-6) for k in (0, 1)],
               prefix_step=[_same_prefix(tr, o[k][2], 4, 1e-5) for k in (0, 1)],
               gt=[float(np.median(synthetic.center_errors_after_sim3(c, p.gt_center))) for c in (cen, o[0][0], o[1][0])])
    src = " [frozen oracle results]" if g is not None else ""
    for k, name in ((0, "forwards"), (1, "backwards")):
        s, st = o[k][1], res["vs"][k]
        print(f"\n[parity] GP {tag} vs the oracle summed {name}{src}: LM {rep['iterations']} ({rep['successful_steps']} accepted, "
              f"{rep['line_search_shrunk']} shortened by the line search) vs {s.iterations} ({s.successful_steps}, {s.line_search_shrunk}), "
              f"final cost {rep['final_cost']:.6f} vs {s.final_cost:.6f}, PCG {rep['linear_iterations']}, same cost to 1e-6 for the first "
              f"{res['prefix'][k]} LM iterations (same step size: {res['prefix_step'][k]}), centre distance / extent: max {st['max']:.3e} "
              f"p99 {st['p99']:.3e} median {st['median']:.3e}")
    st = res["self"]
    print(f"[parity] GP {tag}: the ORACLE against itself, sums backwards vs forwards: LM {o[1][1].iterations} vs {o[0][1].iterations}, final "
          f"cost {o[1][1].final_cost:.6f} vs {o[0][1].final_cost:.6f}, same cost to 1e-6 for the first {res['self_prefix']} LM iterations, "
          f"centre distance / extent: max {st['max']:.3e} p99 {st['p99']:.3e} median {st['median']:.3e} | median error vs ground truth: "
          f"GPU {res['gt'][0]:.3e}, oracle {res['gt'][1]:.3e} / {res['gt'][2]:.3e}")
    return res


def _assert_gp_parity(res):
    """The two regimes of _gp_parity's docstring, decided by the oracle's own two roundings."""
    rep, o, me = res["rep"], res["oracle"], res["self"]
    if me["max"] < 1e-5:
        # the reference's end point is defined on this input: north_star's bar on the worst camera
        s = o[0][1]
        assert res["vs"][0]["max"] < 1e-3
        assert abs(rep["iterations"] - s.iterations) <= 1
        assert abs(rep["final_cost"] - s.final_cost) <= 1e-4 * s.final_cost
        return
    # chaotic input: (1) the same trajectory for (almost) as long as the reference follows its own ...
    assert max(res["prefix"]) >= min(8, res["self_prefix"] - 3), (res["prefix"], res["self_prefix"])
    # (2) ... an end point inside the reference's own scatter: typical cameras (median), the tail (p99) ...
    k = 0 if res["vs"][0]["p99"] <= res["vs"][1]["p99"] else 1
    assert res["vs"][k]["median"] <= 2.0 * me["median"] + 1e-5, (res["vs"], me)
    assert res["vs"][k]["p99"] <= 2.0 * me["p99"] + 1e-4, (res["vs"], me)
    # ... the final cost, and the quality against ground truth
    spread = abs(o[1][1].final_cost - o[0][1].final_cost)
    # (configs[2] seed 0 over nine oracle / GPU variants: 5119.7 ... 5126.7, i.e. +- 0.07 %; configs[3] seed 2: GPU 10172.8, oracle 10184.6 / 10184.7)
    assert min(abs(rep["final_cost"] - o[j][1].final_cost) for j in (0, 1)) <= 3.0 * spread + 3e-3 * o[0][1].final_cost
    assert res["gt"][0] <= 1.05 * max(res["gt"][1], res["gt"][2]) + 1e-5


def test_gp_stable_input_ends_where_the_oracle_ends(gsfm_ctx):
    """An input on which the reference algorithm's end point IS defined (150 cameras / 6 000 tracks, seed 0: the oracle summed
    forwards and backwards ends in the same place to 6e-10, 28 LM iterations / 27 accepted / 12 shortened by the line search on
    both): the HIP solve takes the same 28 / 27 / 12, the same line-search step sizes, and ends 3e-5 of the extent away with
    the default solver tolerance (1e-10; 7e-7 at 1e-12, 3.7e-4 at 1e-8, 2.4e-3 at 1e-6 —
    profiles/r06_gp_line_search_gpu_vs_oracle.txt).  Bar: north_star's 1e-3 on the worst camera."""
    p = synthetic.make_gp_problem(num_cams=150, num_pts=6000, seed=0)
    res = _gp_parity("150 cameras / 6 000 tracks, seed 0", p, gsfm_ctx)
    assert res["self"]["max"] < 1e-6  # the premise: a stable input
    _assert_gp_parity(res)
    rep, s = res["rep"], res["oracle"][0][1]
    assert (rep["iterations"], rep["successful_steps"], rep["line_search_shrunk"]) == (s.iterations, s.successful_steps, s.line_search_shrunk)
    assert res["vs"][0]["max"] < 2e-4
    # the line search ran and took the oracle's step sizes for the first dozen iterations at least
    tr, tr_o = res["trace"], res["oracle"][0][2]
    assert (tr[:, 4] < 1.0).sum() == s.line_search_shrunk and res["prefix_step"][0] >= 8
    dstep = np.abs(tr[:, 4] / tr_o[:, 4] - 1).max()
    print(f"[parity] GP stable input: line-search step sizes GPU vs oracle, all {len(tr)} iterations: max relative difference {dstep:.2e}")
    assert dstep < 5e-2
    assert np.array_equal(tr[:, 5], tr_o[:, 5])  # the same accept / reject decisions
    # ... and switched off (max_num_line_search_step_size_iterations = 0, as in Ceres) the library runs the loop of rounds 1 - 5
    off = estimators.GlobalPositionerOptions()
    off.solver_options.max_num_line_search_step_size_iterations = 0
    rc, c2, _, rep2 = estimators.gp_solve(p, off, ctx=gsfm_ctx)
    assert rc == 0 and rep2["line_search_trials"] == 0 and rep2["line_search_shrunk"] == 0
    assert (rep2["iterations"], rep2["successful_steps"]) == (34, 19)  # tools/exp_gp_line_search.py: the committed oracle of round 5


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_gp_config3_matches_cpu_oracle(gsfm_ctx, seed):
    """configs[2] (5k cameras / 500k tracks / ~3M observations), three seeds (seed 1 with 10 % uncalibrated cameras): the HIP
    solve against the exact-solve CPU oracle — see _gp_parity for what is compared and why.  Seed 0 runs the oracle live
    (both summation orders, 2 x 45 s on the box's 16 cores), seeds 1 and 2 use its frozen results."""
    p = _gp_full_size_problem(5000, 500_000, seed)
    res = _gp_parity(f"configs[2] seed {seed}", p, gsfm_ctx, frozen=None if seed == 0 else f"gp_c3_s{seed}_oracle.npz")
    _assert_gp_parity(res)


def test_gp_solver_tolerance_inside_the_reference_scatter(gsfm_ctx):
    """Why the reduced systems are solved to 1e-10 and not to round 5's 1e-12: on configs[2] the HIP solve at 1e-8 and at 1e-12
    are as far from each other as the oracle is from itself under another summation order — the tolerance is not what
    decides the end point here (tools/exp_gp_pcg_tolerance.py: the same on the CPU for 1e-6 ... 1e-14) — and the stable-input
    test above pins what it does decide.  Asserted: the median and p99 of the distance between the two HIP solves stay
    inside twice the documented scatter, and the looser solve does less linear work."""
    p = _gp_full_size_problem(5000, 500_000, 0)
    out = []
    for tol in (1e-12, 1e-8):
        o = estimators.GlobalPositionerOptions()
        o.solver_options.pcg_relative_tolerance = tol
        rc, c, _, rep = estimators.gp_solve(p, o, ctx=gsfm_ctx)
        assert rc == 0
        out.append((c, rep))
    st = synthetic.center_distance_stats(out[1][0], out[0][0])
    print(f"\n[parity] GP configs[2] seed 0, PCG 1e-8 vs 1e-12: LM {out[1][1]['iterations']} vs {out[0][1]['iterations']}, PCG "
          f"{out[1][1]['linear_iterations']} vs {out[0][1]['linear_iterations']}, centre distance / extent: max {st['max']:.3e} p99 "
          f"{st['p99']:.3e} median {st['median']:.3e} (the oracle against itself: p99 2.0e-3, median 1.8e-5)")
    assert st["median"] < 5e-5 and st["p99"] < 5e-3
    assert out[1][1]["linear_iterations"] < out[0][1]["linear_iterations"]


def test_ba_config4_matches_cpu_oracle(gsfm_ctx):
    """configs[3] on one GPU (10k cameras / 1M tracks / ~5M observations, one SIMPLE_RADIAL camera per image): the HIP
    solve against the exact-solve CPU oracle on the same inputs.  Bar = north_star: rotations <= 1e-4 rad, camera
    centres <= 1e-3 relative to the scene extent (no alignment: the first frame is constant in both)."""
    from oracle import cpu

    p = synthetic.make_ba_problem(10_000, 1_000_000, seed=0, shared_intrinsics=False)
    rc, q, t, X, intr, rep = estimators.ba_solve(p, ctx=gsfm_ctx)
    assert rc == 0
    r = cpu.ba_solve(p.num_cams, p.pt_offset, p.obs_cam, p.obs_xy, p.cam_intr, p.intr_model, p.fixed_cam, p.cam_q, p.cam_t,
                     p.pt_xyz, p.intr_params)
    assert r[0]
    s = r[5]
    assert s.max_linear_residual < 1e-7  # the oracle's linear solves really were exact (true residual, every LM step)
    assert abs(rep["initial_cost"] - s.initial_cost) <= 1e-10 * s.initial_cost
    assert abs(rep["iterations"] - s.iterations) <= 3
    assert abs(rep["final_cost"] - s.final_cost) <= 1e-3 * s.final_cost
    ang = np.radians(so3.rotation_angle_deg(so3.quat_to_rotmat(q), so3.quat_to_rotmat(r[1])))
    cg = -np.einsum("nji,nj->ni", so3.quat_to_rotmat(q), t)
    co = -np.einsum("nji,nj->ni", so3.quat_to_rotmat(r[1]), r[2])
    dc = np.linalg.norm(cg - co, axis=1).max() / _extent(co)
    print(f"\n[parity] BA configs[3]: LM {rep['iterations']} vs {s.iterations}, final cost {rep['final_cost']:.3f} vs {s.final_cost:.3f}, "
          f"max rotation distance {ang.max():.3e} rad (bar 1e-4), max centre distance / extent {dc:.3e} (bar 1e-3)")
    assert ang.max() < 1e-4
    assert dc < 1e-3
    assert np.abs(intr[:, 0] - r[4][:, 0]).max() < 1e-3 * 1200.0  # focal lengths


def test_ba_config4_shared_intrinsics_matches_cpu_oracle(gsfm_ctx):
    """configs[3] with ONE camera shared by all images (SURVEY 8(d) names both intrinsics variants; bench.py times this one
    as extra.ba_c4_shared_intrinsics): the HIP solve against the CPU oracle's poses, frozen in
    tests/golden/ba_c4_shared_oracle.npz (tests/golden/make_ba_shared_golden.py: ten minutes of oracle time, thread-count
    independent; that file also says how far the oracle can be trusted on this input — two oracle configurations agree to
    7e-7 rad / 1.5e-6).  The input has a free scale gauge along which LM creeps, so the two LM
    trajectories are NOT expected to have the same length (GPU 43, oracle 53 iterations; the oracle's own variants 52 / 53)
    and the final costs agree to 1e-4 only; the poses they end in are compared at north_star's bar."""
    import os

    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "ba_c4_shared_oracle.npz"))
    p = synthetic.make_ba_problem(10_000, 1_000_000, seed=0, shared_intrinsics=True)
    # same input (numpy's pairwise sums may group differently on another CPU: compare to rounding)
    assert p.num_obs == int(g["num_obs"]) and abs(float(np.sum(p.obs_xy)) / float(g["obs_xy_checksum"]) - 1) < 1e-12
    assert abs(float(np.sum(p.cam_t)) / float(g["cam_t_checksum"]) - 1) < 1e-12
    rc, q, t, X, intr, rep = estimators.ba_solve(p, ctx=gsfm_ctx)
    assert rc == 0
    assert abs(rep["initial_cost"] - float(g["out_initial_cost"])) <= 1e-10 * float(g["out_initial_cost"])
    ang = np.radians(so3.rotation_angle_deg(so3.quat_to_rotmat(q), so3.quat_to_rotmat(g["out_q"])))
    cg = -np.einsum("nji,nj->ni", so3.quat_to_rotmat(q), t)
    co = -np.einsum("nji,nj->ni", so3.quat_to_rotmat(g["out_q"]), g["out_t"])
    dc = np.linalg.norm(cg - co, axis=1).max() / _extent(co)
    print(f"\n[parity] BA configs[3], one shared camera: LM {rep['iterations']} vs {int(g['out_iterations'])}, final cost "
          f"{rep['final_cost']:.3f} vs {float(g['out_final_cost']):.3f}, max rotation distance {ang.max():.3e} rad (bar 1e-4), "
          f"max centre distance / extent {dc:.3e} (bar 1e-3)")
    assert abs(rep["final_cost"] - float(g["out_final_cost"])) <= 1e-3 * float(g["out_final_cost"])
    assert ang.max() < 1e-4
    assert dc < 1e-3
    assert abs(intr[0, 0] - g["out_intr"][0, 0]) < 1e-3 * 1200.0  # the shared focal length


def test_ba_config4_shared_intrinsics_follows_the_exact_oracle_trajectory(gsfm_ctx):
    """The same input stopped after 24 LM iterations.  Why: the end point of this problem sits in a nearly flat valley (one
    shared camera + one constant frame: the scale gauge is held by the damping alone), LM creeps along it with trust-region
    radii of 1e6 ... 1e8, and there NO iterative solve is exact — the oracle's verbose log (make_ba_shared_golden.py) shows
    true relative residuals of its reduced solves of <= 1e-8 up to LM iteration 24 and 1e-6 ... 1e-1 afterwards, border
    eliminated densely or not.  Up to iteration 24 the oracle IS what SPARSE_SCHUR would compute, so this is where the
    trajectories are compared: same accept / reject decisions, same cost, poses to far below the bar (the end-point test
    above keeps the bar itself).  Fixture: tests/golden/ba_c4_shared_oracle_it24.npz."""
    import os

    path = os.path.join(os.path.dirname(__file__), "golden", "ba_c4_shared_oracle_it24.npz")
    g = np.load(path)
    assert float(g["out_max_linear_residual"]) < 1e-7  # the oracle's solves were exact on this stretch
    p = synthetic.make_ba_problem(10_000, 1_000_000, seed=0, shared_intrinsics=True)
    assert p.num_obs == int(g["num_obs"]) and abs(float(np.sum(p.obs_xy)) / float(g["obs_xy_checksum"]) - 1) < 1e-12
    opt = estimators.BundleAdjusterOptions()
    opt.solver_options.max_num_iterations = 24
    rc, q, t, X, intr, rep = estimators.ba_solve(p, opt, ctx=gsfm_ctx)
    assert rc == 0
    ang = np.radians(so3.rotation_angle_deg(so3.quat_to_rotmat(q), so3.quat_to_rotmat(g["out_q"])))
    cg = -np.einsum("nji,nj->ni", so3.quat_to_rotmat(q), t)
    co = -np.einsum("nji,nj->ni", so3.quat_to_rotmat(g["out_q"]), g["out_t"])
    dc = np.linalg.norm(cg - co, axis=1).max() / _extent(co)
    print(f"\n[parity] BA configs[3], one shared camera, first 24 LM iterations (oracle solves exact to "
          f"{float(g['out_max_linear_residual']):.1e}): LM {rep['iterations']} vs {int(g['out_iterations'])}, cost "
          f"{rep['final_cost']:.3f} vs {float(g['out_final_cost']):.3f}, max rotation distance {ang.max():.3e} rad, max centre "
          f"distance / extent {dc:.3e}, focal {intr[0, 0]:.6f} vs {float(g['out_intr'][0, 0]):.6f}")
    assert rep["iterations"] == int(g["out_iterations"])
    assert abs(rep["final_cost"] - float(g["out_final_cost"])) <= 1e-6 * float(g["out_final_cost"])
    assert ang.max() < 1e-5
    assert dc < 1e-4
    assert abs(intr[0, 0] - g["out_intr"][0, 0]) < 1e-4 * 1200.0


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_gp_config4_matches_cpu_oracle(gsfm_ctx, seed):
    """Global positioning at the size the headline times it — configs[3]: 10k cameras / 1M tracks / ~6.0M observations —
    against the exact-solve CPU oracle on the same inputs and the same std::mt19937 start, three seeds (seed 0 is the
    headline's GP problem); oracle results frozen by tests/golden/make_gp_c4_golden.py (both summation orders, with their LM
    traces; six four-minute oracle runs otherwise).  _gp_parity says what is compared."""
    p = _gp_full_size_problem(10_000, 1_000_000, seed)
    res = _gp_parity(f"configs[3] size, seed {seed}", p, gsfm_ctx, frozen=f"gp_c4_s{seed}_oracle.npz")
    _assert_gp_parity(res)


def test_gp_sequential_capture_matches_cpu_oracle(gsfm_ctx):
    """A walk-around capture (every point seen by a run of consecutive cameras: the camera graph is a chain closed into a
    ring) is where block-Jacobi PCG needs thousands of iterations per solve; the library switches its second-level
    preconditioner on (cluster translations + local scale, gp.hip GpCoarseDev).  Same bars as the other GP parity tests
    against the C++ oracle (plain block-Jacobi PCG, capped at the reference's 1 000 iterations per solve — which it hits
    here, hence the looser bound on its linear residual), and the preconditioner has to show in the operator count."""
    from oracle import cpu

    p = synthetic.make_gp_problem(2_500, 125_000, seed=0, capture="sequential")
    rc, cen, xyz, rep = estimators.gp_solve(p, ctx=gsfm_ctx)
    assert rc == 0
    ok, c_o, X_o, s = cpu.gp_solve(p.num_cams, p.pt_offset, p.obs_cam, p.obs_dir, p.obs_calibrated, p.cam_center, p.pt_xyz)
    assert ok and s.max_linear_residual < 1e-4
    assert abs(rep["initial_cost"] - s.initial_cost) <= 1e-12 * s.initial_cost
    st = synthetic.center_distance_stats(cen, c_o)
    e_g = float(np.median(synthetic.center_errors_after_sim3(cen, p.gt_center)))
    e_o = float(np.median(synthetic.center_errors_after_sim3(c_o, p.gt_center)))
    print(f"\n[parity] GP sequential capture 2.5k / 125k: LM {rep['iterations']} vs {s.iterations}, final cost "
          f"{rep['final_cost']:.6f} vs {s.final_cost:.6f}, operator applications {rep['linear_iterations']} vs "
          f"{s.linear_iterations}, centre distance GPU-oracle / extent: max {st['max']:.3e} p99 {st['p99']:.3e} median {st['median']:.3e}; "
          f"median error vs ground truth {e_g:.3e} vs {e_o:.3e}")
    # Round 6: a chain-like scene is the most sensitive input there is for the LM path with Ceres' line search in it, and the
    # oracle's own reduced solves are not exact here (capped PCG): the two runs are two samples of the reference's scatter
    # (_gp_parity) — held together through the bulk of the cameras, the cost and the quality against ground truth.
    assert abs(rep["final_cost"] - s.final_cost) <= 5e-3 * s.final_cost
    assert st["median"] < 0.5 * e_o  # the two runs are closer to each other than either is to ground truth (7e-3 vs 3.6e-2)
    assert e_g < 1.1 * e_o + 1e-5
    assert rep["linear_iterations"] < 0.5 * s.linear_iterations


def test_gp_points_and_cameras_balanced_matches_cpu_oracle(gsfm_ctx):
    """Camera-to-camera constraints next to the tracks (POINTS_AND_CAMERAS_BALANCED, gp.cc:42-71, 167-255) above the size
    where one workgroup runs the PCG (1 024 cameras): the multi-workgroup vector update with the pair terms' delta slots.
    Against the exact-solve C++ oracle on the same seeded inputs; same bars as the other GP parity tests."""
    from oracle import cpu
    from test_gp_gpu import _pairs

    p = synthetic.make_gp_problem(1_200, 40_000, seed=2, dir_noise=1e-3, outlier_ratio=0.02)
    p.pair_i, p.pair_j, p.pair_dir = _pairs(p, np.random.default_rng(2), num_succ=4, noise=1e-3)
    kw = dict(constraint_type=2, constraint_reweight_scale=2.0)
    rc, cen, xyz, rep = estimators.gp_solve(p, estimators.GlobalPositionerOptions(**kw), ctx=gsfm_ctx)
    assert rc == 0
    from oracle import gp as ogp

    ok, c_o, X_o, s = cpu.gp_solve(p.num_cams, p.pt_offset, p.obs_cam, p.obs_dir, p.obs_calibrated, p.cam_center, p.pt_xyz,
                                   ogp.GlobalPositionerOptions(**kw), pair_i=p.pair_i, pair_j=p.pair_j, pair_dir=p.pair_dir)
    assert ok
    assert abs(rep["initial_cost"] - s.initial_cost) <= 1e-12 * s.initial_cost
    st = synthetic.center_distance_stats(cen, c_o)
    tr, tr_o = gsfm_ctx.lm_trace(), cpu.lm_trace()
    same = _same_prefix(tr, tr_o, 0, 1e-6)
    print(f"\n[parity] GP POINTS_AND_CAMERAS_BALANCED 1.2k / 40k / 4.8k pairs: LM {rep['iterations']} vs {s.iterations}, final cost "
          f"{rep['final_cost']:.6f} vs {s.final_cost:.6f}, same cost to 1e-6 for the first {same} LM iterations, centre distance GPU-oracle / "
          f"extent: max {st['max']:.3e} p99 {st['p99']:.3e} median {st['median']:.3e} (largest true residual of the oracle's solves: "
          f"{s.max_linear_residual:.1e})")
    # Round 6: with the line search the LM path of this input runs through trust-region radii at which the oracle's PCG (3 600
    # unknowns: above its dense limit, pairs: no deflation) stops converging — true relative residual 1e-2 in its worst
    # solve — so the oracle is exact only on the first stretch of the trajectory.  Asserted: that stretch, and the end point
    # through bulk, cost and iteration count.
    assert same >= 8
    assert abs(rep["final_cost"] - s.final_cost) <= 5e-3 * s.final_cost
    assert st["max"] < 1e-3  # (2.8e-5: the oracle runs 14 more LM iterations with its inexact late solves and gets nowhere else)


def test_ra_config4_matches_cpu_oracle(gsfm_ctx):
    """The rotation-averaging stage of configs[3] (10k cameras / 500k edges) against the C++ oracle (direct skyline
    Cholesky solves): same L1 / IRLS iteration counts, rotations to 1e-6 rad."""
    from oracle import cpu

    p = synthetic.make_ring_view_graph(10_000, 50, seed=0)
    rc, rot, rep = estimators.ra_solve(p, ctx=gsfm_ctx)
    assert rc == 0
    ro = {}
    ok, rot_o = cpu.ra_estimate_rotations(p.num_nodes, p.edge_i, p.edge_j, p.edge_q, p.edge_weight, p.edge_ninl, p.node_aa0,
                                          p.fixed_node, report=ro)
    assert ok and (rep["iterations_l1"], rep["iterations_irls"]) == (ro["l1_iterations"], ro["irls_iterations"])
    d = np.radians(so3.rotation_angle_deg(so3.aa_to_rotmat(rot), so3.aa_to_rotmat(rot_o)))
    print(f"\n[parity] RA configs[3]: {rep['iterations_l1']} L1 + {rep['iterations_irls']} IRLS iterations both, "
          f"max rotation distance GPU-oracle = {d.max():.3e} rad (bar 1e-6)")
    assert d.max() < 1e-6


@pytest.mark.parametrize("n", [5000, 16000])
def test_ra_large_graphs_match_the_oracle(gsfm_ctx, n):
    """5k cameras / 250k edges (the camera count of configs[2]) and 16k / 800k: the block-preconditioned PCG path against
    the oracle (its sparse LU is banded on the ring graphs: 7 s and 36 s on one host core), and against the Jacobi-PCG
    path (another preconditioner and node order solving the same systems).  At 16k the reference algorithm itself leaves
    a few nodes more than 100 degrees off — a sub-tree that the spanning-tree initialisation hung on an outlier edge and
    ten ADMM iterations per L1 solve do not repair — and the HIP path must reproduce exactly that."""
    from oracle import ra as ora

    p = synthetic.make_ring_view_graph(n, 50, seed=0)
    rc, rot, rep = estimators.ra_solve(p, ctx=gsfm_ctx)
    assert rc == 0
    tr = ora.RaTrace()
    ok, rot_o = ora.estimate_rotations(p.num_nodes, p.edge_i, p.edge_j, p.edge_q, p.edge_weight, p.edge_ninl, p.node_aa0,
                                       p.fixed_node, ora.RotationEstimatorOptions(), tr)
    assert ok and rep["iterations_l1"] == tr.l1_iterations and rep["iterations_irls"] == tr.irls_iterations
    d = np.radians(so3.rotation_angle_deg(so3.aa_to_rotmat(rot), so3.aa_to_rotmat(rot_o)))
    assert d.max() < 1e-6
    err = synthetic.rotation_errors_deg(so3.aa_to_rotmat(rot), p.gt_R)
    err_o = synthetic.rotation_errors_deg(so3.aa_to_rotmat(rot_o), p.gt_R)
    assert np.median(err) < 0.5 and np.array_equal(err > 5.0, err_o > 5.0)
    if n == 5000:
        assert err.max() < 3.0
    rc, rot_it, rep_it = estimators.ra_solve(p, estimators.RotationEstimatorOptions(force_iterative=True), ctx=gsfm_ctx)
    assert rc == 0 and rep_it["iterations_l1"] == rep["iterations_l1"] and rep_it["iterations_irls"] == rep["iterations_irls"]
    d = np.radians(so3.rotation_angle_deg(so3.aa_to_rotmat(rot_it), so3.aa_to_rotmat(rot_o)))
    assert d.max() < 1e-6
    assert rep["linear_iterations"] < 0.3 * rep_it["linear_iterations"]
    if n == 5000:  # idempotence
        p2 = type(p)(**{**p.__dict__, "node_aa0": rot})
        rc, rot2, rep2 = estimators.ra_solve(
            p2, estimators.RotationEstimatorOptions(skip_initialization=True, max_num_l1_iterations=0), ctx=gsfm_ctx)
        assert rc == 0 and rep2["iterations_irls"] == 1


def test_track_establishment_config3_full(gsfm_ctx):
    """Match graph of the size of configs[2] (5k images, ~5.7M inlier matches): integer work, so the full-size result
    is compared with the (vectorised) oracle bit for bit."""
    from glomap_amd.tracks import MatchGraph, TrackEngine
    from oracle import tracks as ot

    g = synthetic.make_match_graph(5000, 500_000, seed=0)
    ref = ot.establish_full_tracks(g["pair_image1"], g["pair_image2"], g["pair_valid"], g["pair_offset"], g["match_feat1"],
                                   g["match_feat2"], g["feat_offset"], g["feat_xy"])
    eng = TrackEngine(MatchGraph.from_dict(g), ctx=gsfm_ctx)
    full = eng.EstablishFullTracks()
    assert eng.num_discarded == ref[4] > 10000
    for a, b in zip((full.track_id, full.track_offset, full.obs_image, full.obs_feature), ref[:4]):
        assert np.array_equal(a, b)
    reg = np.ones(5000, np.uint8)
    reg[::11] = 0
    from glomap_amd.tracks import TrackEstablishmentOptions

    for kw in (dict(), dict(min_num_tracks_per_view=150, max_num_tracks=200000)):
        eng.options = TrackEstablishmentOptions(**kw)
        sel = eng.FindTracksForProblem(reg)
        want = ot.find_tracks_for_problem(*ref[:4], reg, **kw)
        for a, b in zip((sel.track_id, sel.track_offset, sel.obs_image, sel.obs_feature), want):
            assert np.array_equal(a, b)


def _chain_against_fixture(name, ncam, npts, ctx, seed=0):
    """GPU chain vs the frozen oracle chain (tests/golden/make_chain_golden.py): returns the stage distances."""
    import os

    from chain_util import GpuBackend, final_pose_distance, run_chain

    g = np.load(os.path.join(os.path.dirname(__file__), "golden", name))
    sc = synthetic.make_chained_scene(ncam, npts, seed=seed)
    # same scene as the one the oracle chain ran on (numpy's pairwise sums may group differently on another CPU)
    assert sc.obs_cam.shape[0] == int(g["num_obs"]) and int(np.sum(sc.obs_cam.astype(np.int64))) == int(g["obs_cam_checksum"])
    assert abs(float(np.sum(sc.obs_xy)) / float(g["obs_xy_checksum"]) - 1) < 1e-12
    assert abs(float(np.sum(sc.ra.edge_q)) / float(g["edge_q_checksum"]) - 1) < 1e-10
    assert float(g["ba1_max_linear_residual"]) < 1e-8 and float(g["ba2_max_linear_residual"]) < 1e-8  # the oracle's solves were exact
    r = run_chain(sc, GpuBackend(ctx))
    d_ra = float(np.radians(so3.rotation_angle_deg(so3.aa_to_rotmat(r["ra_rot"]), so3.aa_to_rotmat(g["ra_rot"]))).max())
    st_gp = synthetic.center_distance_stats(r["gp_center"], g["gp_center"])
    ang, st_ba = final_pose_distance(r["ba_q"], r["ba_t"], g["ba_q"], g["ba_t"])
    # how good the result is against ground truth (context for the distances): rotations and Sim(3)-aligned centres
    Rf = so3.quat_to_rotmat(r["ba_q"])
    cf = -np.einsum("nji,nj->ni", Rf, r["ba_t"])
    gt_rot = float(np.median(synthetic.rotation_errors_deg(Rf, sc.gt_R)))
    gt_cen = synthetic.center_distance_stats(cf, sc.gt_center)
    b1, b2 = r["rep_ba1"], r["rep_ba2"]
    print(f"\n[parity] chain RA->GP->filters->BA {ncam} cameras / {npts} tracks / {sc.obs_cam.shape[0]} observations: RA {r['rep_ra']['l1']}+"
          f"{r['rep_ra']['irls']} vs {int(g['ra_l1'])}+{int(g['ra_irls'])} iterations, rotations {d_ra:.3e} rad apart | GP LM "
          f"{r['rep_gp']['iterations']} vs {int(g['gp_iterations'])}, cost {r['rep_gp']['final_cost']:.6f} vs {float(g['gp_final_cost']):.6f}, "
          f"centres max {st_gp['max']:.3e} p99 {st_gp['p99']:.3e} | observations kept by the filters {r['observations_kept']} vs "
          f"{g['observations_kept'].tolist()} | BA positions-only LM {b1['iterations']} ({b1['successful']}) vs {int(g['ba1_iterations'])} "
          f"({int(g['ba1_successful'])}), BA full LM {b2['iterations']} ({b2['successful']}) vs {int(g['ba2_iterations'])} "
          f"({int(g['ba2_successful'])}), cost {b2['final_cost']:.3f} vs {float(g['ba2_final_cost']):.3f} | FINAL POSES GPU-oracle: rotations max "
          f"{ang:.3e} rad (bar 1e-4), centres / extent max {st_ba['max']:.3e} p99 {st_ba['p99']:.3e} median {st_ba['median']:.3e} (bar 1e-3) "
          f"| vs ground truth: median rotation error {gt_rot:.4f} deg, centres median {gt_cen['median']:.2e}")
    if "rev_ba_q" in g:  # the oracle chain with every GP / BA reduction summed backwards: the reference against itself
        ang_o, st_o = final_pose_distance(g["rev_ba_q"], g["rev_ba_t"], g["ba_q"], g["ba_t"])
        st_gpo = synthetic.center_distance_stats(g["rev_gp_center"], g["gp_center"])
        print(f"[parity] chain {ncam} / {npts}: the ORACLE chain against itself (sums backwards vs forwards): GP LM {int(g['rev_gp_iterations'])} vs "
              f"{int(g['gp_iterations'])}, GP centres max {st_gpo['max']:.3e} p99 {st_gpo['p99']:.3e} median {st_gpo['median']:.3e}, "
              f"observations kept {g['rev_observations_kept'].tolist()} vs {g['observations_kept'].tolist()} | FINAL POSES: rotations max "
              f"{ang_o:.3e} rad, centres / extent max {st_o['max']:.3e} p99 {st_o['p99']:.3e} median {st_o['median']:.3e}")
    return r, g, d_ra, st_gp, ang, st_ba


def _assert_chain(r, g, d_ra, st_gp, ang, st_ba):
    """What a chain test asserts (round 6).  Rotation averaging is a contraction: equal iteration counts, 1e-6 rad.  Global
    positioning with Ceres' line search is chaotic on these scenes for the reference algorithm itself (_gp_parity) — two
    roundings of the ORACLE chain end 1.6e-3 apart after GP, with 54 vs 38 LM iterations, on the 2 000-camera scene
    (tools/exp_chain_oracle_scatter.py) — so after GP only the bulk is held (p99, median).  The filters then keep a few dozen
    of a million observations differently, and bundle adjustment contracts again: the FINAL poses of the two oracle roundings
    are 1e-5 rad / 6e-5 apart, and north_star's bar — 1e-4 rad, 1e-3 of the extent — is asserted on the worst camera."""
    assert (r["rep_ra"]["l1"], r["rep_ra"]["irls"]) == (int(g["ra_l1"]), int(g["ra_irls"]))
    assert d_ra < 1e-6
    assert st_gp["p99"] < 5e-3 and st_gp["median"] < 1e-4
    for a, b in zip(r["observations_kept"], g["observations_kept"].tolist()):
        assert abs(a - b) <= max(2, int(2e-4 * b)), (r["observations_kept"], g["observations_kept"].tolist())
    assert ang < 1e-4
    assert st_ba["max"] < 1e-3


@pytest.mark.parametrize("seed", [0, 1])
def test_chain_config4_final_poses_match_the_oracle_chain(gsfm_ctx, seed):
    """north_star's bar is on the FINAL camera poses: rotation averaging -> global positioning (bearings oriented by the
    rotations RA returned, random start) -> the three track filters and the normalisation -> bundle adjustment (positions
    only, then with rotations; started from GP's centres and points), chained on ONE configs[3]-size scene as
    GlobalMapper::Solve chains estimators and processors (global_mapper.cc:92-223; driver tests/chain_util.py) — every stage
    through the C ABI on the GPU, against the same chain run by the exact-solve CPU oracle (frozen:
    tests/golden/chain_c4_oracle.npz, tests/golden/make_chain_golden.py).  Rotations <= 1e-4 rad (no alignment: node 0 is
    RA's gauge and BA's constant frame in both chains), camera centres <= 1e-3 of the scene extent after Sim(3) alignment
    (BA inherits the scale the normaliser set).  Two scenes (seeds 0, 1)."""
    name = "chain_c4_oracle.npz" if seed == 0 else f"chain_c4_s{seed}_oracle.npz"  # two scenes
    _assert_chain(*_chain_against_fixture(name, 10_000, 1_000_000, gsfm_ctx, seed=seed))


@pytest.mark.parametrize("seed", [0, 1])
def test_chain_2k_final_poses_match_the_oracle_chain(gsfm_ctx, seed):
    """The same chain on 2 000-camera / 200 000-track scenes (fixtures tests/golden/chain_2k_oracle.npz, chain_2k_s1_oracle.npz).
    Seed 2 of this generator is not a fixture: its positions-only BA creeps along the free scale gauge with trust-region radii
    of 1e8 ... 1e9, where the oracle's own reduced solves break down (true relative residual 5.0 in one LM step) — the
    situation of DESIGN.md section 2 "BA, one camera shared by all images"; there is no exact reference on that stretch."""
    name = "chain_2k_oracle.npz" if seed == 0 else f"chain_2k_s{seed}_oracle.npz"
    _assert_chain(*_chain_against_fixture(name, 2_000, 200_000, gsfm_ctx, seed=seed))


def test_ba_config4_full_opencv_matches_cpu_oracle(gsfm_ctx):
    """configs[3] size through a 12-parameter camera model (the 16-wide unit, csrc/ba_wide.hip; the reference dispatches any
    CameraModelId, bundle_adjustment.cc:136-139): 10k images / 1M tracks / ~5M observations, 100 FULL_OPENCV cameras shared
    round-robin, against the C++ oracle's result frozen in tests/golden/ba_c4_full_opencv_oracle.npz
    (tests/golden/make_ba_wide_golden.py says why a fixture and why 100 cameras).  Bar = north_star."""
    import os

    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "ba_c4_full_opencv_oracle.npz"))
    p = synthetic.make_ba_problem_wide(10_000, 1_000_000, "full_opencv", seed=0, num_intr_groups=100)
    assert p.num_obs == int(g["num_obs"]) and abs(float(np.sum(p.obs_xy)) / float(g["obs_xy_checksum"]) - 1) < 1e-12
    assert abs(float(np.sum(p.cam_t)) / float(g["cam_t_checksum"]) - 1) < 1e-12
    gsfm_ctx.stats(reset=True)
    rc, q, t, X, intr, rep = estimators.ba_solve(p, ctx=gsfm_ctx)
    st = gsfm_ctx.stats(reset=True)
    assert rc == 0 and intr.shape == (100, 16)
    assert st["pcg_deflated"] > 0 and st["pcg_joint_blocks"] == 0  # the wide unit: separate blocks, gauge deflated by application
    assert abs(rep["initial_cost"] - float(g["out_initial_cost"])) <= 1e-10 * float(g["out_initial_cost"])
    assert abs(rep["iterations"] - int(g["out_iterations"])) <= 3
    assert abs(rep["final_cost"] - float(g["out_final_cost"])) <= 1e-4 * float(g["out_final_cost"])
    ang = np.radians(so3.rotation_angle_deg(so3.quat_to_rotmat(q), so3.quat_to_rotmat(g["out_q"])))
    cg = -np.einsum("nji,nj->ni", so3.quat_to_rotmat(q), t)
    co = -np.einsum("nji,nj->ni", so3.quat_to_rotmat(g["out_q"]), g["out_t"])
    dc = np.linalg.norm(cg - co, axis=1).max() / _extent(co)
    print(f"\n[parity] BA configs[3] FULL_OPENCV (100 cameras): LM {rep['iterations']} vs {int(g['out_iterations'])}, final cost "
          f"{rep['final_cost']:.3f} vs {float(g['out_final_cost']):.3f}, max rotation distance {ang.max():.3e} rad (bar 1e-4), max centre "
          f"distance / extent {dc:.3e} (bar 1e-3), focal lengths {np.abs(intr[:, :2] - g['out_intr'][:, :2]).max():.3e} px")
    assert ang.max() < 1e-4
    assert dc < 1e-3
    assert np.array_equal(intr[:, 2:4], p.intr_params[:, 2:4]) and np.array_equal(intr[:, 12:], np.zeros((100, 4)))
