"""Cost of one all-reduce through the peer-mailbox transport (csrc/peer.hpp) as a PCG loop pays it: R processes, one
libgsfm context each, all on GPU 0 (the only GPU of the test box — between GPUs the same kernels write over xGMI instead of
into local HBM, so this measures the launch / flag / ordered-sum part of the cost, not the link transfer).
Usage: python tools/exp_peer_allreduce.py [R ...]      (default 2 4 8)"""
import os
import socket
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

SIZES = [(30_001, "GP configs[3]: 3 N + 1"), (140_001, "BA configs[3]: 6 N + 8 K + 1"), (1_000_000, "8 MB")]


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      GSFM_PEER_TIMEOUT_S="30")
    import torch.distributed as dist

    from glomap_amd import _lib

    dist.init_process_group("gloo", rank=rank, world_size=world)
    ctx = _lib.Context(0)

    def allgather(b):
        out = [None] * world
        dist.all_gather_object(out, b)
        return out

    ctx.comm_init_peer(allgather, rank, world, 1 << 18)
    ctx.comm_peer_selftest()
    res = []
    for n, _ in SIZES:
        dist.barrier()
        res.append(ctx.comm_allreduce_bench(n, 300))
    q.put((rank, res))
    dist.barrier()
    ctx.close()
    dist.destroy_process_group()


def main():
    import multiprocessing as mp

    worlds = [int(a) for a in sys.argv[1:]] or [2, 4, 8]
    mpc = mp.get_context("spawn")
    for world in worlds:
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            port = s.getsockname()[1]
        q = mpc.Queue()
        procs = [mpc.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
        for p in procs:
            p.start()
        res = dict(q.get(timeout=600) for _ in range(world))
        for p in procs:
            p.join(timeout=60)
        for i, (n, what) in enumerate(SIZES):
            us = [res[r][i] for r in range(world)]
            print(f"world {world}: all-reduce of {n} doubles ({what}): {max(us):.1f} us per collective (slowest rank; fastest {min(us):.1f})")


if __name__ == "__main__":
    main()
