"""Sharded (N > 1) path on the GPU: two / four ranks share the one MI355X of the test box.  Two transports carry the
collectives: the host-staged validation transport (gsfm_comm_init_host over gloo) and the library's own peer-mailbox
all-reduce (gsfm_comm_peer_*, csrc/peer.hpp) — between processes on one GPU its mailboxes are mapped with the same
hipIpc* calls as between GPUs, and its push / flag / ordered-sum kernels are the ones a multi-GPU node runs.  RCCL cannot
be exercised with several ranks on one device.  Same sharding, same kernels, same reduction points: the sharded solves
must reproduce the single-rank solves of the same problems — same LM / IRLS iteration counts, poses equal to solver
precision, replicated state bit-identical on every rank."""
import os
import socket

import numpy as np
import pytest

from glomap_amd import estimators, sharding, so3, synthetic

pytestmark = pytest.mark.gpu



def _gp_follows(rep, cen, rep1, cen1):
    """A sharded GP solve against the single-rank solve of the same problem from the same random start.  The all-reduce sums
    the shards' contributions in another order than one rank sums its observations, and global positioning with Ceres' line
    search in the loop (round 6) amplifies rounding ~10 x per LM iteration (tests/test_fullsize_gpu.py::_gp_parity), so what a
    correct sharding guarantees is: the same start, the same first LM iterations to solver precision — cost, candidate cost,
    radius, line-search step size, accept / reject — and an end point of the same quality; NOT the same last digits."""
    assert abs(rep["initial_cost"] - rep1["initial_cost"]) <= 1e-12 * rep1["initial_cost"]
    tr, tr1 = rep["lm_trace"], rep1["lm_trace"]
    n = min(6, len(tr), len(tr1))
    assert n >= 4
    for col, rtol in ((0, 1e-7), (3, 1e-6), (1, 1e-6), (4, 1e-4)):  # (measured: 2e-11 at the third iteration, 1e-9 at the sixth)
        assert np.allclose(tr[:n, col], tr1[:n, col], rtol=rtol, atol=0), (col, tr[:n, col], tr1[:n, col])
    assert np.array_equal(tr[:n, 5], tr1[:n, 5])
    assert abs(rep["iterations"] - rep1["iterations"]) <= 3 and abs(rep["successful_steps"] - rep1["successful_steps"]) <= 3
    assert abs(rep["final_cost"] - rep1["final_cost"]) <= 1e-3 * rep1["final_cost"]
    d = np.linalg.norm(cen - cen1, axis=1)  # same start, same gauge: no alignment
    ext = np.linalg.norm(cen1 - cen1.mean(0), axis=1).max()
    assert np.median(d) <= 1e-3 * ext and d.max() <= 1e-1 * ext, (np.median(d) / ext, d.max() / ext)  # (measured 3e-4 / 2.3e-2)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _problems():
    ra = synthetic.make_ring_view_graph(200, 15, seed=3)
    gp = synthetic.make_gp_problem(40, 2000, seed=2)
    ba = synthetic.make_ba_problem(num_cams=30, num_pts=1500, seed=4, shared_intrinsics=True)
    return ra, gp, ba


def _gp_with_pairs():
    """Tracks + camera-to-camera constraints (POINTS_AND_CAMERAS_BALANCED: the point losses are weighed by #pairs / #tracks of
    the WHOLE problem, gp.cc:223-233): the tracks are sharded, the pairs replicated."""
    gp = synthetic.make_gp_problem(40, 2000, seed=2)
    rng = np.random.default_rng(5)
    i = np.repeat(np.arange(gp.num_cams), 3)
    j = (i + np.tile(np.arange(1, 4), gp.num_cams)) % gp.num_cams
    d = gp.gt_center[j] - gp.gt_center[i]
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    d += 1e-3 * rng.normal(size=d.shape)
    gp.pair_i, gp.pair_j = i.astype(np.int32), j.astype(np.int32)
    gp.pair_dir = d / np.linalg.norm(d, axis=1, keepdims=True)
    return gp, estimators.GlobalPositionerOptions(constraint_type=2, constraint_reweight_scale=2.0)


def _ba_wide():
    return synthetic.make_ba_problem_wide(30, 1500, "full_opencv", seed=4, num_intr_groups=3)


def _worker(rank, world, port, ra_init, q, transport="host"):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      GSFM_PEER_TIMEOUT_S="30")
    import torch.distributed as dist

    from glomap_amd import _lib

    dist.init_process_group("gloo", rank=rank, world_size=world)
    ctx = _lib.Context(0)
    if transport == "peer":
        def allgather(b):
            out = [None] * world
            dist.all_gather_object(out, b)
            return out

        # the smallest capacity: the per-iteration vectors fit (RA 600, GP 121, BA 189 doubles), BA's per-LM-step cross
        # blocks (48 N = 1 440 doubles) go in two pieces — both the one-piece and the chunked path run
        ctx.comm_init_peer(allgather, rank, world, 1024)
        assert ctx.comm_peer_selftest() == world
    else:
        ctx.comm_init_host(sharding.host_allreduce(dist), rank, world)
    ra, gp, ba = _problems()
    out = {}
    # RA: edges sharded, nodes replicated; initial rotations = the MST initialisation of the whole graph
    s, _ = sharding.shard_ra_problem(ra, rank, world)
    s.node_aa0 = ra_init
    rc, rot, rep = estimators.ra_solve(s, estimators.RotationEstimatorOptions(skip_initialization=True), ctx=ctx)
    out["ra"] = (rc, rot, rep)  # sharded => iterative linear solver
    # GP: tracks sharded, centres replicated
    s, (lo, hi) = sharding.shard_gp_problem(gp, rank, world)
    rc, cen, xyz, rep = estimators.gp_solve(s, ctx=ctx)
    rep["lm_trace"] = ctx.lm_trace()
    out["gp"] = (rc, cen, rep)
    gpp, gpp_opt = _gp_with_pairs()
    s, _ = sharding.shard_gp_problem(gpp, rank, world)
    rc, cen, xyz, rep = estimators.gp_solve(s, gpp_opt, ctx=ctx)
    rep["lm_trace"] = ctx.lm_trace()
    out["gp_pairs"] = (rc, cen, rep)
    # BA: tracks sharded, poses and intrinsics replicated
    s, (lo, hi) = sharding.shard_ba_problem(ba, rank, world)
    rc, q_, t_, X_, intr_, rep = estimators.ba_solve(s, ctx=ctx)
    out["ba"] = (rc, q_, t_, intr_, X_, (lo, hi), rep)
    # BA through a 12-parameter camera model: the 16-wide unit (csrc/ba_wide.hip), tracks sharded like the narrow one
    baw = _ba_wide()
    s, _ = sharding.shard_ba_problem(baw, rank, world)
    rc, q_, t_, X_, intr_, rep = estimators.ba_solve(s, ctx=ctx)
    out["ba_wide"] = (rc, q_, t_, intr_, rep)
    q.put((rank, out))
    dist.barrier()
    ctx.close()
    dist.destroy_process_group()


@pytest.mark.timeout(900)
@pytest.mark.parametrize("world,transport", [(2, "host"), (4, "host"), (2, "peer"), (4, "peer")])
def test_ranks_reproduce_single_rank(gsfm_ctx, world, transport):
    import multiprocessing as mp

    ra, gp, ba = _problems()
    # single-rank references (and the MST initialisation handed to the sharded RA)
    rc, ra_init, _ = estimators.ra_solve(
        ra, estimators.RotationEstimatorOptions(max_num_l1_iterations=0, max_num_irls_iterations=0), ctx=gsfm_ctx)
    assert rc == 0
    ra1 = type(ra)(**{**ra.__dict__, "node_aa0": ra_init})
    # same linear solver as the sharded run (PCG; a single rank would otherwise pick the dense direct solver)
    rc, rot1, rep_ra1 = estimators.ra_solve(
        ra1, estimators.RotationEstimatorOptions(skip_initialization=True, force_iterative=True), ctx=gsfm_ctx)
    assert rc == 0
    rc, cen1, xyz1, rep_gp1 = estimators.gp_solve(gp, ctx=gsfm_ctx)
    assert rc == 0
    rep_gp1["lm_trace"] = gsfm_ctx.lm_trace()
    rc, q1, t1, X1, intr1, rep_ba1 = estimators.ba_solve(ba, ctx=gsfm_ctx)
    assert rc == 0
    gpp, gpp_opt = _gp_with_pairs()
    rc, cenp1, _, rep_gpp1 = estimators.gp_solve(gpp, gpp_opt, ctx=gsfm_ctx)
    assert rc == 0
    rep_gpp1["lm_trace"] = gsfm_ctx.lm_trace()

    mpc = mp.get_context("spawn")
    queue = mpc.Queue()
    port = _free_port()
    procs = [mpc.Process(target=_worker, args=(r, world, port, ra_init, queue, transport)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict(queue.get(timeout=800) for _ in range(world))
    ranks = range(world)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0

    # --- RA: same iteration path, rotations equal to solver precision on every rank
    for r in ranks:
        rc, rot, rep = res[r]["ra"]
        assert rc == 0
        assert rep["iterations_l1"] == rep_ra1["iterations_l1"]
        assert abs(rep["iterations_irls"] - rep_ra1["iterations_irls"]) <= 1  # all-reduce changes the summation order
        ang = np.radians(so3.rotation_angle_deg(so3.aa_to_rotmat(rot), so3.aa_to_rotmat(rot1)))
        assert ang.max() < 1e-4, ang.max()
    assert all(np.array_equal(res[0]["ra"][1], res[r]["ra"][1]) for r in ranks)  # replicated state is bit-identical

    # --- GP: every shard draws its part of the ONE std::mt19937 stream of the unsharded problem (same random start),
    # so the sharded solve follows the single-rank solve (_gp_follows says how far "follows" can go)
    for r in ranks:
        rc, cen, rep = res[r]["gp"]
        assert rc == 0
        _gp_follows(rep, cen, rep_gp1, cen1)
    assert all(np.array_equal(res[0]["gp"][1], res[r]["gp"][1]) for r in ranks)
    # ... and with camera-to-camera constraints next to the tracks (pairs replicated, rank 0 adds their terms)
    for r in ranks:
        rc, cen, rep = res[r]["gp_pairs"]
        assert rc == 0
        _gp_follows(rep, cen, rep_gpp1, cenp1)
    assert all(np.array_equal(res[0]["gp_pairs"][1], res[r]["gp_pairs"][1]) for r in ranks)

    # --- BA: deterministic start => the sharded solve follows the single-rank solve
    for r in ranks:
        rc, q_, t_, intr_, X_, (lo, hi), rep = res[r]["ba"]
        assert rc == 0
        assert rep["iterations"] == rep_ba1["iterations"]
        assert abs(rep["final_cost"] - rep_ba1["final_cost"]) <= 1e-9 * rep_ba1["final_cost"]
        ang = np.radians(so3.rotation_angle_deg(so3.quat_to_rotmat(q_), so3.quat_to_rotmat(q1)))
        assert ang.max() < 1e-6  # arccos-based angle: resolution ~ sqrt(eps)
        assert np.abs(t_ - t1).max() < 1e-7 * (1 + np.abs(t1).max())
        assert np.abs(intr_ - intr1).max() < 1e-6
        assert np.abs(X_ - X1[lo:hi]).max() < 1e-6 * (1 + np.abs(X1).max())
    assert all(np.array_equal(res[0]["ba"][1], res[r]["ba"][1]) for r in ranks)

    # --- BA, 16-wide unit (FULL_OPENCV): the same sharding through ba_wide.hip
    baw = _ba_wide()
    rc, qw1, tw1, Xw1, iw1, rep_w1 = estimators.ba_solve(baw, ctx=gsfm_ctx)
    assert rc == 0 and iw1.shape[1] == 16
    for r in ranks:
        rc, q_, t_, intr_, rep = res[r]["ba_wide"]
        assert rc == 0 and intr_.shape == iw1.shape
        assert rep["iterations"] == rep_w1["iterations"]
        assert abs(rep["final_cost"] - rep_w1["final_cost"]) <= 1e-8 * rep_w1["final_cost"]
        ang = np.radians(so3.rotation_angle_deg(so3.quat_to_rotmat(q_), so3.quat_to_rotmat(qw1)))
        assert ang.max() < 1e-6
        assert np.abs(t_ - tw1).max() < 1e-6 * (1 + np.abs(tw1).max())
        assert np.abs(intr_ - iw1).max() < 1e-4 * (1 + np.abs(iw1).max())
    assert all(np.array_equal(res[0]["ba_wide"][1], res[r]["ba_wide"][1]) for r in ranks)
    assert all(np.array_equal(res[0]["ba_wide"][3], res[r]["ba_wide"][3]) for r in ranks)


def _big_problems():
    """Above the single-workgroup PCG size (1 024 cameras), one intrinsics block per image: the code configs[3] runs at N > 1."""
    gp = synthetic.make_gp_problem(1200, 40_000, seed=5)
    ba = synthetic.make_ba_problem(num_cams=1200, num_pts=40_000, seed=6)
    return gp, ba


def _big_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      GSFM_PEER_TIMEOUT_S="60")
    import torch.distributed as dist

    from glomap_amd import _lib

    dist.init_process_group("gloo", rank=rank, world_size=world)
    ctx = _lib.Context(0)

    def allgather(b):
        out = [None] * world
        dist.all_gather_object(out, b)
        return out

    ctx.comm_init_peer(allgather, rank, world, 1 << 18)
    assert ctx.comm_peer_selftest() == world
    gp, ba = _big_problems()
    out = {}
    s, _ = sharding.shard_gp_problem(gp, rank, world)
    ctx.stats(reset=True)
    rc, cen, xyz, rep = estimators.gp_solve(s, ctx=ctx)
    rep["lm_trace"] = ctx.lm_trace()
    out["gp"] = (rc, cen, rep, ctx.stats(reset=True))
    s, (lo, hi) = sharding.shard_ba_problem(ba, rank, world)
    rc, q_, t_, X_, intr_, rep = estimators.ba_solve(s, ctx=ctx)
    out["ba"] = (rc, q_, t_, intr_, rep, ctx.stats(reset=True))
    q.put((rank, out))
    dist.barrier()
    ctx.close()
    dist.destroy_process_group()


@pytest.mark.timeout(900)
@pytest.mark.parametrize("world,chunked,recycle", [(2, 0, 1), (4, 0, 0), (2, 1, 1)])
def test_ranks_reproduce_single_rank_above_the_single_workgroup_size(gsfm_ctx, world, chunked, recycle, monkeypatch):
    """1 200 cameras / 40 k tracks, one intrinsics block per image, peer transport: GP takes the multi-block k_cg_update<3>
    with the gauge modes deflated and the all-reduced closed-form k_gp_aw_modes products, BA the joint 14 x 14 blocks with
    k_ba_aw_modes — the paths bench.py --gpus N runs on configs[3] (chunked = 1: with GP's camera-side sweep in the chunked,
    XCD-partitioned order, which every rank builds over its own tracks).  The counters of gsfm_ctx_stats prove they ran."""
    import multiprocessing as mp

    gp, ba = _big_problems()
    gsfm_ctx.stats(reset=True)
    knobs = {}
    if chunked:  # GP's camera-side sweep in the chunked order (forced: the library picks it only above ~130 k tracks per rank)
        knobs["chunked_sweeps"] = 1
    if recycle:
        # Ritz vectors harvested from solves of 8 iterations on, Ritz values below 0.9 (defaults 25 / 0.3: this problem's solves
        # take ~12 and its spectrum has no tail), so that the recycled-vector preconditioner and its all-reduced dot products
        # run — with the chunked sweep (they ride on k_gp_wsum) and without it (k_cgr_dots_w)
        knobs["gp_recycle_min_iters"] = 8
        knobs["gp_recycle_cut_percent"] = 90
    for k, v in knobs.items():
        gsfm_ctx.set_knob(k, v)
    if knobs:  # the spawned ranks read it when they create their context
        monkeypatch.setenv("GSFM_KNOBS", ",".join(f"{k}={v}" for k, v in knobs.items()))
    try:
        rc, cen1, xyz1, rep_gp1 = estimators.gp_solve(gp, ctx=gsfm_ctx)
        rep_gp1["lm_trace"] = gsfm_ctx.lm_trace()
    finally:
        for k in knobs:
            gsfm_ctx.set_knob(k, 0)
    assert rc == 0
    st_gp1 = gsfm_ctx.stats(reset=True)
    assert (st_gp1["pcg_chunked_sweeps"] > 0) == bool(chunked)
    rc, q1, t1, X1, intr1, rep_ba1 = estimators.ba_solve(ba, ctx=gsfm_ctx)
    assert rc == 0
    st_ba1 = gsfm_ctx.stats(reset=True)
    assert st_gp1["pcg_deflated"] > 0 and st_gp1["pcg_closed_form_aw"] == st_gp1["pcg_deflated"]
    assert st_ba1["pcg_joint_blocks"] == st_ba1["pcg_solves"] and st_ba1["pcg_closed_form_aw"] > 0

    mpc = mp.get_context("spawn")
    queue = mpc.Queue()
    port = _free_port()
    procs = [mpc.Process(target=_big_worker, args=(r, world, port, queue)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict(queue.get(timeout=800) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    ranks = range(world)
    for r in ranks:
        rc, cen, rep, st = res[r]["gp"]
        assert rc == 0
        # the paths: no single-workgroup solve, deflated solves with closed-form products, collectives issued
        assert st["pcg_single_workgroup"] == 0 and abs(st["pcg_solves"] - st_gp1["pcg_solves"]) <= 3
        assert abs(st["pcg_deflated"] - st_gp1["pcg_deflated"]) <= 3 and st["pcg_deflated"] > 0 and st_gp1["pcg_deflated"] > 0
        assert st["pcg_closed_form_aw"] == st["pcg_deflated"]
        assert st["allreduces"] > st["pcg_iterations"]
        assert (st["pcg_chunked_sweeps"] == st["pcg_solves"]) if chunked else st["pcg_chunked_sweeps"] == 0
        # recycled Ritz vectors: the dot products with them travel with w through the all-reduce, the
        # harvest is a function of replicated scalars — every rank harvests what one rank harvests
        assert st["pcg_recycled"] == res[0]["gp"][3]["pcg_recycled"] and st["ritz_harvested"] == res[0]["gp"][3]["ritz_harvested"]
        assert (st["pcg_recycled"] >= 10) == bool(recycle) == (st_gp1["pcg_recycled"] >= 10)
        _gp_follows(rep, cen, rep_gp1, cen1)
        assert abs(rep["linear_iterations"] - rep_gp1["linear_iterations"]) <= 0.15 * rep_gp1["linear_iterations"] + 2
    assert all(np.array_equal(res[0]["gp"][1], res[r]["gp"][1]) for r in ranks)  # replicated state is bit-identical
    for r in ranks:
        rc, q_, t_, intr_, rep, st = res[r]["ba"]
        assert rc == 0
        assert st["pcg_joint_blocks"] == st["pcg_solves"] == st_ba1["pcg_solves"]
        assert st["pcg_deflated"] == st_ba1["pcg_deflated"] > 0
        assert st["pcg_closed_form_aw"] == st_ba1["pcg_closed_form_aw"] > 0
        assert st["allreduces"] > st["pcg_iterations"]
        assert rep["iterations"] == rep_ba1["iterations"] and rep["successful_steps"] == rep_ba1["successful_steps"]
        assert abs(rep["linear_iterations"] - rep_ba1["linear_iterations"]) <= 0.02 * rep_ba1["linear_iterations"] + 2
        assert abs(rep["final_cost"] - rep_ba1["final_cost"]) <= 1e-9 * rep_ba1["final_cost"]
        ang = np.radians(so3.rotation_angle_deg(so3.quat_to_rotmat(q_), so3.quat_to_rotmat(q1)))
        assert ang.max() < 1e-6
        assert np.abs(t_ - t1).max() < 1e-7 * (1 + np.abs(t1).max())
        assert np.abs(intr_ - intr1).max() < 1e-6 * (1 + np.abs(intr1).max())
    assert all(np.array_equal(res[0]["ba"][1], res[r]["ba"][1]) for r in ranks)
    assert all(np.array_equal(res[0]["ba"][2], res[r]["ba"][2]) for r in ranks)


@pytest.mark.gpu
def test_rccl_transport_in_a_process_that_also_imports_torch():
    """bench.py imports torch (rendezvous) and libgsfm (RCCL inside the library) into one process.  PyTorch wheels
    bundle their own HIP runtime and RCCL; this checks that libgsfm's communicator, stream and device memory still
    belong together there: a world-size-1 communicator is created from a fresh unique id and one checked all-reduce
    runs on the context's stream, followed by a sharded-API solve with the communicator attached."""
    import numpy as np
    import torch  # noqa: F401  (the point of the test)

    from glomap_amd import _lib, estimators, synthetic

    ctx = _lib.Context(-1)
    try:
        ctx.comm_init(_lib.comm_unique_id(), 0, 1)
        assert ctx.comm_selftest() == 1.0
        p = synthetic.make_gp_problem(12, 150, seed=3)
        rc, cen, X, rep = estimators.gp_solve(p, ctx=ctx)
        assert rc == 0 and np.isfinite(cen).all()
    finally:
        ctx.close()


def _rig_problems():
    """Known rigs (GP image offsets, BA constant cam_from_rig), unknown cam_from_rig centres (GP sensor blocks) and
    optimize_rig_poses (BA sensor blocks) — the rig branches of gp.cc:318-368 and ba.cc:147-179 — on one scene."""
    import copy

    gp, ba, info = synthetic.make_rig_problems(30, 3, 3000, seed=4, dir_noise=1e-3, pixel_noise=0.5, outlier_ratio=0.01)
    gp_unk = synthetic.forget_rig_translations(gp, info)
    rng = np.random.default_rng(12)
    s0 = info["sensor_cam_from_rig"].copy()
    s0[:, :4] = so3.rotmat_to_quat(so3.aa_to_rotmat(rng.normal(0, np.radians(0.5), (s0.shape[0], 3))) @ so3.quat_to_rotmat(s0[:, :4]))
    s0[:, 4:] += rng.normal(0, 0.03, (s0.shape[0], 3))
    ba_sens = copy.copy(ba)
    ba_sens.image_sensor = info["sensor_block"].copy()
    ba_sens.sensor_cam_from_rig = s0
    return gp, gp_unk, ba, ba_sens


def _rig_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      GSFM_PEER_TIMEOUT_S="60")
    import torch.distributed as dist

    from glomap_amd import _lib

    dist.init_process_group("gloo", rank=rank, world_size=world)
    ctx = _lib.Context(0)

    def allgather(b):
        out = [None] * world
        dist.all_gather_object(out, b)
        return out

    ctx.comm_init_peer(allgather, rank, world, 1 << 16)
    gp, gp_unk, ba, ba_sens = _rig_problems()
    out = {}
    opt = estimators.GlobalPositionerOptions()
    opt.solver_options.pcg_relative_tolerance = 1e-10
    for name, prob in (("gp_known", gp), ("gp_unknown", gp_unk)):
        s, _ = sharding.shard_gp_problem(prob, rank, world)
        rc, cen, xyz, rep = estimators.gp_solve(s, opt, ctx=ctx)
        rep["lm_trace"] = ctx.lm_trace()
        out[name] = (rc, cen, rep)
    for name, prob, o in (("ba_known", ba, estimators.BundleAdjusterOptions()),
                          ("ba_sensors", ba_sens, estimators.BundleAdjusterOptions(optimize_rig_poses=True))):
        o.solver_options.pcg_relative_tolerance = 1e-10
        s, _ = sharding.shard_ba_problem(prob, rank, world)
        rc, q_, t_, X_, intr_, rep = estimators.ba_solve(s, o, ctx=ctx)
        out[name] = (rc, q_, t_, intr_, rep)
    q.put((rank, out))
    dist.barrier()
    ctx.close()
    dist.destroy_process_group()


@pytest.mark.gpu
@pytest.mark.timeout(900)
@pytest.mark.parametrize("world", [2, 3])
def test_ranks_reproduce_single_rank_on_rig_problems(gsfm_ctx, world):
    """Multi-camera rigs with the tracks sharded over the ranks: the image-space quantities of the rig kernels enter frame /
    sensor space through linear maps in front of the all-reduces the solvers do anyway, so known rigs, unknown cam_from_rig
    centres (GP sensor blocks, their random start drawn after every rank's points) and optimize_rig_poses (BA sensor blocks)
    reproduce the single-rank solves."""
    import multiprocessing as mp

    gp, gp_unk, ba, ba_sens = _rig_problems()
    opt = estimators.GlobalPositionerOptions()
    opt.solver_options.pcg_relative_tolerance = 1e-10
    ref = {}
    for name, prob in (("gp_known", gp), ("gp_unknown", gp_unk)):
        rc, cen, xyz, rep = estimators.gp_solve(prob, opt, ctx=gsfm_ctx)
        assert rc == 0
        rep["lm_trace"] = gsfm_ctx.lm_trace()
        ref[name] = (cen, rep)
    for name, prob, o in (("ba_known", ba, estimators.BundleAdjusterOptions()),
                          ("ba_sensors", ba_sens, estimators.BundleAdjusterOptions(optimize_rig_poses=True))):
        o.solver_options.pcg_relative_tolerance = 1e-10
        rc, q_, t_, X_, intr_, rep = estimators.ba_solve(prob, o, ctx=gsfm_ctx)
        assert rc == 0
        ref[name] = (q_, t_, intr_, rep)
    mpc = mp.get_context("spawn")
    queue = mpc.Queue()
    port = _free_port()
    procs = [mpc.Process(target=_rig_worker, args=(r, world, port, queue)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict(queue.get(timeout=800) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for r in range(world):
        for name in ("gp_known", "gp_unknown"):
            rc, cen, rep = res[r][name]
            cen1, rep1 = ref[name]
            assert rc == 0
            assert abs(rep["initial_cost"] - rep1["initial_cost"]) <= 1e-12 * rep1["initial_cost"]  # the same random start
            assert abs(rep["iterations"] - rep1["iterations"]) <= 3
            assert abs(rep["final_cost"] - rep1["final_cost"]) <= 1e-3 * rep1["final_cost"]
            assert synthetic.center_errors_after_sim3(cen, cen1).max() < 1e-3
            if name == "gp_unknown":
                sc, _, _ = synthetic.align_sim3(cen, cen1)
                assert np.abs(rep["sensor_center"] * sc - rep1["sensor_center"]).max() < 1e-3 * np.abs(rep1["sensor_center"]).max()
            print(f"[parity] {name} on {world} ranks: LM {rep['iterations']} / {rep1['iterations']}, final cost {rep['final_cost']:.9g} / "
                  f"{rep1['final_cost']:.9g}, centres {synthetic.center_errors_after_sim3(cen, cen1).max():.2e}")
        for name in ("ba_known", "ba_sensors"):
            rc, q_, t_, intr_, rep = res[r][name]
            q1, t1, intr1, rep1 = ref[name]
            assert rc == 0
            assert abs(rep["initial_cost"] - rep1["initial_cost"]) <= 1e-12 * rep1["initial_cost"]
            assert rep["iterations"] == rep1["iterations"]
            assert abs(rep["final_cost"] - rep1["final_cost"]) <= 1e-8 * rep1["final_cost"]
            ang = np.radians(so3.rotation_angle_deg(so3.quat_to_rotmat(q_), so3.quat_to_rotmat(q1)))
            assert ang.max() < 1e-6 and np.abs(t_ - t1).max() < 1e-6 * (1 + np.abs(t1).max())
            if name == "ba_sensors":
                sc, sc1 = rep["sensor_cam_from_rig"], rep1["sensor_cam_from_rig"]
                assert np.abs(sc - sc1).max() < 1e-6
            print(f"[parity] {name} on {world} ranks: LM {rep['iterations']} / {rep1['iterations']}, final cost {rep['final_cost']:.12g} / "
                  f"{rep1['final_cost']:.12g}, rotations {ang.max():.2e} rad")
    for name in ("gp_known", "gp_unknown", "ba_known", "ba_sensors"):
        assert all(np.array_equal(res[0][name][1], res[r][name][1]) for r in range(world))  # replicated state bit-identical
