"""Rotation averaging at 10 k cameras on the ring of configs[3] and on the three non-ring graphs of bench.py's extras
(k-NN, hubs, loop-closure chords; node ids shuffled): time, iteration counts, error against ground truth.
Usage: python tools/ab_ra_graphs.py"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

import bench  # noqa: E402
from glomap_amd import _lib, estimators, so3, synthetic  # noqa: E402

ctx = _lib.Context(0)
for item in filter(None, os.environ.get("AB_KNOBS", "").split(",")):
    k, _, v = item.partition("=")
    ctx.set_knob(k, int(v))
N = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000
for kind in ("ring", "geometric", "hub", "chords"):
    p = synthetic.make_ring_view_graph(N, 50, seed=0) if kind == "ring" else synthetic.make_view_graph(kind, N, 100, seed=0)
    pd = bench._dev_ra(ctx, p)
    rot = pd.node_aa0.clone()
    times = []
    for i in range(3):
        rot.copy_from(pd.node_aa0)
        ctx.synchronize()
        t0 = time.perf_counter()
        rc, _, rep = estimators.ra_solve(pd, estimators.RotationEstimatorOptions(), ctx=ctx, rot_inout=rot)
        ctx.synchronize()
        assert rc == 0, rc
        if i:
            times.append(time.perf_counter() - t0)
    err = synthetic.rotation_errors_deg(so3.aa_to_rotmat(rot.numpy()), p.gt_R)
    print("%-10s %6d cameras %8d edges: %7.1f ms  L1 %d IRLS %d PCG %d  median error %.4f deg" % (
        kind, N, p.num_edges, np.median(times) * 1e3, rep["iterations_l1"], rep["iterations_irls"], rep["linear_iterations"],
        np.median(err)), flush=True)
