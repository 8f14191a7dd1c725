"""GP at configs[3] size (10 k cameras / 1 M tracks / 6.0 M observations) under several settings of the chunked-sweep knob:
solve time, LM / PCG counts, final cost and the HIP-event averages of the two PCG sweeps (k_gp_phaseA, k_gp_phaseB[_x]).
Usage: python tools/ab_gp_sweeps.py [knob values ...]      (default: 2 0 16 32 48)"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

from glomap_amd import _lib, estimators, synthetic  # noqa: E402

vals = [int(a) for a in sys.argv[1:]] or [2, 0, 16, 32, 48]
ctx = _lib.Context(0)
p = synthetic.make_gp_problem(10_000, 1_000_000, seed=0)
if os.environ.get("AB_SORT_TRACKS"):  # what a library-side track ordering would give: tracks by the smallest camera that sees them
    key = os.environ["AB_SORT_TRACKS"]
    lens = np.diff(p.pt_offset)
    if key == "min":
        k = np.minimum.reduceat(p.obs_cam, p.pt_offset[:-1])
    else:  # circular mean of the cameras on the ring
        ang = 2 * np.pi * p.obs_cam / p.num_cams
        sx = np.add.reduceat(np.cos(ang), p.pt_offset[:-1]); sy = np.add.reduceat(np.sin(ang), p.pt_offset[:-1])
        k = np.arctan2(sy, sx)
    perm = np.argsort(k, kind="stable")
    new_off = np.zeros(p.num_pts + 1, np.int64); np.cumsum(lens[perm], out=new_off[1:])
    idx = np.repeat(p.pt_offset[:-1][perm] - new_off[:-1], lens[perm]) + np.arange(new_off[-1])
    p.pt_offset, p.obs_cam, p.obs_dir, p.obs_calibrated = new_off, p.obs_cam[idx], np.ascontiguousarray(p.obs_dir[idx]), p.obs_calibrated[idx]
    print("tracks sorted by", key)
print("problem:", p.num_cams, "cameras", p.num_pts, "tracks", p.num_obs, "observations", flush=True)
exps = [int(a) for a in os.environ.get("AB_EXPERIMENT", "0").split(",")]
if os.environ.get("AB_BA"):  # the same A/B on bundle adjustment at configs[3] size: sweeps k_ba_phaseA (id 2) / k_ba_phaseB (id 4)
    b = synthetic.make_ba_problem(10_000, 1_000_000, seed=0)
    for e in exps:
        ctx.set_knob("experiment", e)
        best = None
        for rep_i in range(3):
            ctx.profile_enable(True)
            ctx.profile_read(2)
            ctx.profile_read(4)
            t0 = time.perf_counter()
            rc, q, t, X, intr, rep = estimators.ba_solve(b, ctx=ctx)
            ctx.synchronize()
            dt = time.perf_counter() - t0
            nA, msA = ctx.profile_read(2)
            nB, msB = ctx.profile_read(4)
            ctx.profile_enable(False)
            if best is None or dt < best[0]:
                best = (dt, rep, nA, msA, nB, msB)
        dt, rep, nA, msA, nB, msB = best
        print("BA experiment=%d solve %.1f ms  LM %d  PCG %d  cost %.3f | phaseA %.1f us (%d)  phaseB %.1f us (%d)" % (
            e, dt * 1e3, rep["iterations"], rep["linear_iterations"], rep["final_cost"], 1e3 * msA / max(1, nA), nA,
            1e3 * msB / max(1, nB), nB), flush=True)
    sys.exit(0)
for v in [(a, e) for a in vals for e in exps]:
    v, e = v
    ctx.set_knob("experiment", e)
    ctx.set_knob("chunked_sweeps", v)
    best = None
    for rep_i in range(3):
        ctx.profile_enable(True)
        ctx.profile_read(1)
        ctx.profile_read(3)
        t0 = time.perf_counter()
        rc, cen, xyz, rep = estimators.gp_solve(p, ctx=ctx)
        ctx.synchronize()
        dt = time.perf_counter() - t0
        nA, msA = ctx.profile_read(1)
        nB, msB = ctx.profile_read(3)
        ctx.profile_enable(False)
        if best is None or dt < best[0]:
            best = (dt, rep, nA, msA, nB, msB)
    dt, rep, nA, msA, nB, msB = best
    print("experiment=%d chunked_sweeps=%-3d solve %.1f ms  LM %d  PCG %d  cost %.6f | phaseA %.1f us (%d)  phaseB %.1f us (%d)" % (
        e, v, dt * 1e3, rep["iterations"], rep["linear_iterations"], rep["final_cost"], 1e3 * msA / max(1, nA), nA,
        1e3 * msB / max(1, nB), nB), flush=True)
