"""The A/B measurements prepared at the end of round 2 (no GPU clock was left to run them) — ONE gpurun call:

    gpurun --timeout 900 -- 'python tools/ab_round3.py > gpurun_out/ab_round3.log 2>&1'

1. `k_gp_phaseA_chain` (GSFM_GP_PHASEA_CHAIN=1; gp.hip) against `k_gp_phaseA`: bit-identical centres required, time per
   launch from the library's own HIP events (DESIGN.md section 7 item 4 expects 105 -> 60-65 us at configs[3]).
2. BA with the reduced solves stopped at 1e-6 instead of 1e-8: same LM trajectory expected,
   rotations within 1e-6 rad of the 1e-8 run, about a quarter fewer PCG iterations.
3. GSFM_DEFLATE=1 (CgDeflation in cg.hpp): the similarity gauge deflated from the PCG of GP
   and BA.  Expected from the CPU oracle (tools/exp_deflation.py): same LM iteration counts, BA linear iterations
   688 -> ~330 (the count includes the k applications that form A W), GP at 10k cameras 1254 -> ~880; BA result within
   1e-7 rad / 1e-4 of the default run, GP within 1e-3 relative after Sim(3) alignment.  THIS CODE HAS NEVER RUN: if a
   variant fails, the default path is untouched by it (every default kernel was diffed against its pre-change ISA).
Each variant runs in its own process because the switches are read once per process."""
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

WORKER = r"""
import json, sys, time
import numpy as np
sys.path.insert(0, {root!r})
from glomap_amd import _lib, estimators, so3, synthetic
what = sys.argv[1]
ctx = _lib.Context(0)
out = dict(what=what)
if what.startswith("gp"):
    p = synthetic.make_gp_problem(10_000, 1_000_000, seed=0)
    opt = estimators.GlobalPositionerOptions()
    best = None
    for rep in range(3):
        t0 = time.perf_counter()
        rc, c, X, r = estimators.gp_solve(p, opt, ctx=ctx)
        ctx.synchronize()
        dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
    ctx.profile_enable(True)  # HIP events around every launch of the phase-A kernel (GSFM_KERNEL_GP_SCHUR = 1, gsfm.h)
    ctx.profile_read(1)
    estimators.gp_solve(p, opt, ctx=ctx)
    ctx.profile_enable(False)
    launches, total_ms = ctx.profile_read(1)
    prof = dict(launches=launches, avg_us=1e3 * total_ms / max(launches, 1))
    out.update(rc=rc, ms=best * 1e3, lm=r["iterations"], pcg=r.get("linear_iterations"), final_cost=r["final_cost"], phaseA=prof)
    np.save(sys.argv[2], c)
else:
    tol = float(sys.argv[3])
    p = synthetic.make_ba_problem(10_000, 1_000_000, seed=0)
    opt = estimators.BundleAdjusterOptions()
    opt.solver_options.pcg_relative_tolerance = tol
    best = None
    for rep in range(3):
        t0 = time.perf_counter()
        rc, q, t, X, intr, r = estimators.ba_solve(p, opt, ctx=ctx)
        ctx.synchronize()
        dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
    out.update(rc=rc, tol=tol, ms=best * 1e3, lm=r["iterations"], pcg=r.get("linear_iterations"), final_cost=r["final_cost"])
    np.save(sys.argv[2], np.concatenate([q.ravel(), t.ravel()]))
print("RESULT " + json.dumps(out, default=str))
"""


def run(args, env_extra=None):
    env = dict(os.environ)
    env.update(env_extra or {})
    p = subprocess.run([sys.executable, "-c", WORKER.format(root=ROOT), *args], env=env, capture_output=True, text=True, timeout=1500)
    for line in p.stdout.splitlines():
        if line.startswith("RESULT "):
            return json.loads(line[7:])
    print(p.stdout[-2000:], p.stderr[-4000:])
    raise SystemExit("worker failed: " + " ".join(args))


def main():
    import numpy as np

    from glomap_amd import build, so3

    build.build_lib(verbose=False)
    tmp = os.path.join(ROOT, "gpurun_out")
    os.makedirs(tmp, exist_ok=True)
    a = run(["gp", os.path.join(tmp, "ab_gp_a.npy")])
    b = run(["gp", os.path.join(tmp, "ab_gp_b.npy")], {"GSFM_GP_PHASEA_CHAIN": "1"})
    ca, cb = np.load(os.path.join(tmp, "ab_gp_a.npy")), np.load(os.path.join(tmp, "ab_gp_b.npy"))
    print("GP  k_gp_phaseA      :", a)
    print("GP  k_gp_phaseA_chain:", b)
    print("GP  centres bit-identical:", bool(np.array_equal(ca, cb)), " max |diff| =", float(np.abs(ca - cb).max()))
    r8 = run(["ba", os.path.join(tmp, "ab_ba_8.npy"), "1e-8"])
    r6 = run(["ba", os.path.join(tmp, "ab_ba_6.npy"), "1e-6"])
    x8, x6 = np.load(os.path.join(tmp, "ab_ba_8.npy")), np.load(os.path.join(tmp, "ab_ba_6.npy"))
    n = x8.shape[0] // 7
    ang = np.radians(so3.rotation_angle_deg(so3.quat_to_rotmat(x8[: 4 * n].reshape(n, 4)), so3.quat_to_rotmat(x6[: 4 * n].reshape(n, 4))))
    print("BA  pcg tol 1e-8:", r8)
    print("BA  pcg tol 1e-6:", r6)
    print("BA  max rotation difference (rad):", float(ang.max()), " max |t| difference:", float(np.abs(x8[4 * n :] - x6[4 * n :]).max()))
    from glomap_amd import synthetic

    gd = run(["gp", os.path.join(tmp, "ab_gp_d.npy")], {"GSFM_DEFLATE": "1"})
    cd = np.load(os.path.join(tmp, "ab_gp_d.npy"))
    ext = np.linalg.norm(ca - ca.mean(0), axis=1).max()
    print("GP  GSFM_DEFLATE=1:", gd)
    print("GP  deflated vs default, max centre distance after Sim(3) / extent:", float(synthetic.center_errors_after_sim3(cd, ca).max()))
    bd = run(["ba", os.path.join(tmp, "ab_ba_d.npy"), "1e-8"], {"GSFM_DEFLATE": "1"})
    xd = np.load(os.path.join(tmp, "ab_ba_d.npy"))
    ang = np.radians(so3.rotation_angle_deg(so3.quat_to_rotmat(x8[: 4 * n].reshape(n, 4)), so3.quat_to_rotmat(xd[: 4 * n].reshape(n, 4))))
    print("BA  GSFM_DEFLATE=1:", bd)
    print("BA  deflated vs default: max rotation difference (rad):", float(ang.max()), " max |t| difference:", float(np.abs(x8[4 * n :] - xd[4 * n :]).max()))


if __name__ == "__main__":
    main()
