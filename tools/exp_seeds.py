"""Other inputs than seed 0 at configs[3] size: GP with the chunked camera-side sweep on / off (LM and PCG counts, final costs,
centre differences after Sim(3)), and the full RA + GP + BA step — a guard against anything that only works on the benchmark's
own seed.  Usage: python tools/exp_seeds.py [seeds ...]   (default 1 2)"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

from glomap_amd import _lib, estimators, so3, synthetic  # noqa: E402

ctx = _lib.Context(0)
for seed in [int(a) for a in sys.argv[1:]] or [1, 2]:
    p = synthetic.make_gp_problem(10_000, 1_000_000, seed=seed, uncalibrated_ratio=0.1 if seed % 2 else 0.0)
    out = {}
    for knob in (2, 0):
        ctx.set_knob("chunked_sweeps", knob)
        ctx.stats(reset=True)
        t0 = time.perf_counter()
        rc, cen, xyz, rep = estimators.gp_solve(p, ctx=ctx)
        dt = time.perf_counter() - t0
        assert rc == 0
        out[knob] = (cen, rep, dt, ctx.stats()["pcg_chunked_sweeps"])
    ctx.set_knob("chunked_sweeps", 0)
    (c0, r0, t0_, n0), (c1, r1, t1_, n1) = out[2], out[0]
    ext = np.linalg.norm(p.gt_center - p.gt_center.mean(0), axis=1).max()
    print("seed %d GP: camera-major LM %d PCG %d cost %.6f (%.0f ms) | chunked (%d solves) LM %d PCG %d cost %.6f (%.0f ms) | "
          "centres apart %.2e of the extent, median error vs ground truth %.2e / %.2e" % (
              seed, r0["iterations"], r0["linear_iterations"], r0["final_cost"], t0_ * 1e3, n1, r1["iterations"],
              r1["linear_iterations"], r1["final_cost"], t1_ * 1e3, synthetic.center_errors_after_sim3(c1, c0).max(),
              np.median(synthetic.center_errors_after_sim3(c0, p.gt_center)), np.median(synthetic.center_errors_after_sim3(c1, p.gt_center))),
          flush=True)
    b = synthetic.make_ba_problem(10_000, 1_000_000, seed=seed)
    t0 = time.perf_counter()
    rc, q, t, X, intr, rep = estimators.ba_solve(b, ctx=ctx)
    dt = time.perf_counter() - t0
    assert rc == 0
    err = synthetic.rotation_errors_deg(so3.quat_to_rotmat(q), so3.quat_to_rotmat(b.gt_q))
    print("seed %d BA: LM %d (accepted %d) PCG %d cost %.3f (%.0f ms), median rotation error vs ground truth %.4f deg" % (
        seed, rep["iterations"], rep["successful_steps"], rep["linear_iterations"], rep["final_cost"], dt * 1e3, np.median(err)), flush=True)
