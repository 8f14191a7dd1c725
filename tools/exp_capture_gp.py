"""GPU: global positioning on a capture-like scene AS THE REFERENCE'S MAPPER POSES IT — the ring that looks outward
(synthetic.make_pipeline_scene(layout="outward")) run through the reference's GlobalMapper::Solve on the adapter classes with
GSFM_DUMP_DIR set, and the dumped GP problem replayed with the library's counters and LM trace: LM iterations, PCG iterations per
solve, which PCG paths ran, milliseconds.  Usage: python tools/exp_capture_gp.py [n_images n_points num_succ]"""
import glob
import json
import os
import sys
import tempfile
import time

root = os.environ.get("GRAFT_REPO_ROOT", ".")
sys.path.insert(0, root)
sys.path.insert(0, os.path.join(root, "tests"))
dump = tempfile.mkdtemp(prefix="gsfm_dump_")
os.environ["GSFM_DUMP_DIR"] = dump
import numpy as np

from glomap_amd import _lib, estimators, flatio, synthetic
import test_dropin_reference_mapper as T

N, P, succ = (int(a) for a in sys.argv[1:4]) if len(sys.argv) > 3 else (300, 20000, 10)
s = synthetic.make_pipeline_scene(n_images=N, n_points=P, seed=0, pixel_noise=0.5, num_succ=succ, layout="outward")
r = T._solve(1, s)
print(f"mapper: ok={r['ok']} {r['seconds']:.2f} s", {k: (v['calls'], round(v['call'] * 1e3, 1), v['iterations'], v['linear_iterations']) for k, v in r["adapter_timings"].items()}, flush=True)
files = sorted(glob.glob(os.path.join(dump, "*.gsfm")))
print("dumps:", [os.path.basename(f) for f in files])
del os.environ["GSFM_DUMP_DIR"]
ctx = _lib.Context(-1)
for path in files:
    rec = flatio.load(path)
    if rec.kind == "ba":  # the two bundle adjustments of the mapper's first round: positions only, then everything
        p, opt = flatio.to_problem(rec)
        res = {}
        for knob in (1, 2):
            ctx.set_knob("gp_dense", knob)
            ctx.stats(reset=True)
            t0 = time.perf_counter()
            rc, q, t, X, intr, rep = estimators.ba_solve(p, opt, ctx=ctx)
            ms = (time.perf_counter() - t0) * 1e3
            tr = ctx.lm_trace()
            res[knob] = (q, t, intr, rep, tr)
            print(json.dumps(dict(file=os.path.basename(path), cams=p.num_cams, pts=p.num_pts, obs=p.num_obs, gp_dense=knob, rc=rc, lm=rep["iterations"],
                                  pcg=rep["linear_iterations"], ms=round(ms, 1), final_cost=rep["final_cost"], stats=ctx.stats(),
                                  costs=[float(x[0]) for x in tr], accepted=[int(x[5]) for x in tr])), flush=True)
        ctx.set_knob("gp_dense", 0)
        print("   never vs always: |dq| %.3e |dt| %.3e |dintr| %.3e" % (np.abs(res[1][0] - res[2][0]).max(), np.abs(res[1][1] - res[2][1]).max(),
                                                                         np.abs(res[1][2] - res[2][2]).max()), flush=True)
        continue
    if rec.kind != "gp":
        continue
    p, opt = flatio.to_problem(rec)
    for knobs in ({},) + tuple({k: 1} for k in sys.argv[4:]):
        for k, v in knobs.items():
            ctx.set_knob(k, v)
        for rep_i in range(2):
            ctx.stats(reset=True)
            t0 = time.perf_counter()
            rc, cen, xyz, rep = estimators.gp_solve(p, opt, ctx=ctx)
            ms = (time.perf_counter() - t0) * 1e3
        tr = ctx.lm_trace()
        print(json.dumps(dict(file=os.path.basename(path), cams=p.num_cams, pts=p.num_pts, obs=p.num_obs, knobs=knobs, rc=rc, lm=rep["iterations"], pcg=rep["linear_iterations"],
                              ms=round(ms, 1), final_cost=rep["final_cost"], stats=ctx.stats(), pcg_per_solve=[int(x[-1]) for x in tr])), flush=True)
        for k in knobs:
            ctx.set_knob(k, 0)
