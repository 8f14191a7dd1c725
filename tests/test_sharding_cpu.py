"""N > 1 path on CPU (gloo, world_size 2): the track sharding that the multi-GPU solvers use is
exact — per-shard reduced camera systems (point blocks eliminated locally) add up to the reduced
system of the whole problem — and the shard helpers partition the work correctly."""
import os
import socket

import numpy as np
import pytest

from glomap_amd import sharding, synthetic


def test_shard_tracks_partitions_everything():
    rng = np.random.default_rng(0)
    lens = rng.integers(2, 30, size=1000)
    off = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    for world in (1, 2, 3, 8):
        ranges = [sharding.shard_tracks(off, r, world) for r in range(world)]
        assert ranges[0][0] == 0 and ranges[-1][1] == 1000
        for (a, b), (c, d) in zip(ranges[:-1], ranges[1:]):
            assert b == c and a <= b
        obs = [off[hi] - off[lo] for lo, hi in ranges]
        assert max(obs) - min(obs) <= 2 * lens.max()  # observation-balanced
    # more ranks than tracks: empty shards are legal
    tiny = np.array([0, 3, 6], dtype=np.int64)
    rr = [sharding.shard_tracks(tiny, r, 4) for r in range(4)]
    assert sum(hi - lo for lo, hi in rr) == 2


def test_shard_edges_partitions_everything():
    for E, world in ((10, 3), (50_000, 8), (3, 4)):
        rr = [sharding.shard_edges(E, r, world) for r in range(world)]
        assert rr[0][0] == 0 and rr[-1][1] == E
        assert all(b == c for (_, b), (c, _) in zip(rr[:-1], rr[1:]))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _reduced_system(p, opt_kwargs, D=1e-3):
    """Reduced camera system (S, g) of one BA problem/shard at its initial point: points eliminated
    exactly, LM damping D on every diagonal (oracle formulas, oracle/ba.py + oracle/lm.py)."""
    import scipy.sparse as sp

    from oracle import ba as oba

    opt = oba.BundleAdjusterOptions(**opt_kwargs)
    lens = np.diff(p.pt_offset)
    pt = np.repeat(np.arange(p.num_pts), lens)
    prob = oba._BaProblem(p.num_cams, p.obs_cam, pt, p.obs_xy, p.cam_intr, p.intr_model, p.fixed_cam, p.num_pts, opt)
    x0 = prob.pack(p.cam_q, p.cam_t, p.pt_xyz, p.intr_params)
    cost, r, J = prob.evaluate(x0)
    nc = prob.pt_col0
    Jc, Jp = J[:, :nc], J[:, nc:]
    Hpp = (Jp.T @ Jp + D * sp.identity(Jp.shape[1])).tocsr()
    from oracle.lm import _block_diag_inverse

    Hinv = _block_diag_inverse(Hpp, 3)
    Hcp = (Jc.T @ Jp).tocsr()
    S = (Jc.T @ Jc - Hcp @ Hinv @ Hcp.T).toarray()
    g = Jc.T @ r - Hcp @ (Hinv @ (Jp.T @ r))
    return cost, S, g


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch
    import torch.distributed as dist

    dist.init_process_group("gloo", rank=rank, world_size=world)
    p = synthetic.make_ba_problem(num_cams=12, num_pts=300, seed=11, shared_intrinsics=True)
    shard, _ = sharding.shard_ba_problem(p, rank, world)
    cost, S, g = _reduced_system(shard, {})
    allreduce = sharding.host_allreduce(dist)
    buf = np.concatenate([[cost], S.ravel(), g])
    allreduce(buf, 0)
    mx = np.array([float(rank)])
    allreduce(mx, 1)
    if rank == 0:
        q.put((buf, mx[0]))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_sharded_reduced_system_equals_whole_gloo_world2():
    import multiprocessing as mp

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for pr in procs:
        pr.start()
    buf, mx = q.get(timeout=240)
    for pr in procs:
        pr.join(timeout=60)
        assert pr.exitcode == 0
    assert mx == 1.0  # max-reduce path of the callback
    p = synthetic.make_ba_problem(num_cams=12, num_pts=300, seed=11, shared_intrinsics=True)
    cost, S, g = _reduced_system(p, {})
    n = g.shape[0]
    assert abs(buf[0] - cost) <= 1e-12 * cost
    S2 = buf[1 : 1 + n * n].reshape(n, n)
    g2 = buf[1 + n * n :]
    assert np.abs(S2 - S).max() <= 1e-9 * np.abs(S).max()
    assert np.abs(g2 - g).max() <= 1e-9 * np.abs(g).max()


def test_gp_shards_carry_every_pair_and_their_tracks_only():
    """Camera-to-camera constraints live in camera space, which is replicated: shard_gp_problem hands EVERY rank the whole
    pair list (rank 0 adds their terms inside the library) next to its own contiguous range of tracks; together the shards
    hold every observation exactly once."""
    p = synthetic.make_gp_problem(num_cams=30, num_pts=500, seed=3)
    rng = np.random.default_rng(0)
    p.pair_i = rng.integers(0, 30, 80).astype(np.int32)
    p.pair_j = ((p.pair_i + rng.integers(1, 5, 80)) % 30).astype(np.int32)
    p.pair_dir = rng.normal(size=(80, 3))
    for world in (2, 3):
        shards = [sharding.shard_gp_problem(p, r, world) for r in range(world)]
        assert sum(s.num_obs for s, _ in shards) == p.num_obs
        assert np.array_equal(np.concatenate([s.obs_cam for s, _ in shards]), p.obs_cam)
        assert np.array_equal(np.concatenate([s.obs_dir for s, _ in shards]), p.obs_dir)
        for s, (lo, hi) in shards:
            assert s.num_pts == hi - lo and s.pt_offset[0] == 0 and s.pt_offset[-1] == s.num_obs
            assert np.array_equal(s.pair_i, p.pair_i) and np.array_equal(s.pair_j, p.pair_j) and np.array_equal(s.pair_dir, p.pair_dir)
            assert s.num_cams == p.num_cams
