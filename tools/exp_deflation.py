"""Deflating the gauge modes from the reduced-system PCG of BA and GP  (CPU study with the C++ oracle, no GPU).

The block-Jacobi-preconditioned reduced camera systems of the synthetic scenes have a tight spectrum (bulk of the
eigenvalues within [0.8, 1.2]) plus a handful of tiny ones: the similarity gauge of the scene.  BA: world translation,
world rotation, scale (7 modes; one constant frame anchors six of them with a stiffness of O(1/N), the LM damping the
seventh).  GP: world translation and scale (4 modes; no frame is constant, only the damping holds them).  These modes
are known in closed form, so the PCG can be deflated on them (Saad et al. 2000): same system, same tolerance and
stopping rule, k extra operator applications per solve to form A W.

    python tools/exp_deflation.py ba [num_cams num_pts]     # configs[3]: 10000 1000000
    python tools/exp_deflation.py gp [num_cams num_pts]     # configs[2]:  5000  500000
    python tools/exp_deflation.py spectrum                   # dense eigenvalues of a 400-camera BA / 500-camera GP system
    python tools/exp_deflation.py variant                    # the algorithm exactly as cg.hpp runs it (numpy, dense 500-camera GP system)

`ba` / `gp` run the C++ oracle at the GPU's PCG tolerance (1e-8) without and with ORC_DEFLATE (two subprocesses; the
switch is read once per process), print the PCG count of every LM iteration (the k applications for A W are INCLUDED
in the deflated counts) and the distance between the two results."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

WORKER = r"""
import json, sys, time
import numpy as np
sys.path.insert(0, {root!r})
from glomap_amd import synthetic
from oracle import cpu, gp as ogp
stage, N, P, out = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
t0 = time.time()
if stage == "ba":
    p = synthetic.make_ba_problem(N, P, seed=0)
    ok, q, t, X, intr, s = cpu.ba_solve(p.num_cams, p.pt_offset, p.obs_cam, p.obs_xy, p.cam_intr, p.intr_model, p.fixed_cam,
                                        p.cam_q, p.cam_t, p.pt_xyz, p.intr_params, pcg_tol=1e-8, verbose=True)
    np.save(out, np.concatenate([q.ravel(), t.ravel()]))
else:
    p = synthetic.make_gp_problem(num_cams=N, num_pts=P, seed=0)
    ok, c, X, s = cpu.gp_solve(p.num_cams, p.pt_offset, p.obs_cam, p.obs_dir, p.obs_calibrated, p.cam_center, p.pt_xyz,
                               ogp.GlobalPositionerOptions(), pcg_tol=1e-8, verbose=True)
    np.save(out, c)
print("RESULT " + json.dumps(dict(ok=bool(ok), lm=int(s.iterations), accepted=int(s.successful_steps), pcg=int(s.linear_iterations),
                                   final_cost=float(s.final_cost), seconds=round(time.time() - t0, 1))))
"""


def run(stage, N, P, out, deflate):
    env = dict(os.environ)
    env.pop("ORC_DEFLATE", None)
    if deflate:
        env["ORC_DEFLATE"] = "7"
    p = subprocess.run([sys.executable, "-c", WORKER.format(root=ROOT), stage, str(N), str(P), out], env=env, capture_output=True, text=True)
    per_step = [ln.split()[7] for ln in p.stderr.splitlines() if ln.startswith("[orc lm] it") and " pcg " in ln]
    for line in p.stdout.splitlines():
        if line.startswith("RESULT "):
            return json.loads(line[7:]), per_step
    raise SystemExit(p.stdout[-2000:] + p.stderr[-4000:])


def spectrum():
    import numpy as np
    import scipy.linalg as sl

    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import exp_precond as G
    import exp_precond_ba as B
    from glomap_amd import synthetic
    from oracle import ba as oba
    from oracle import gp as ogp

    N, P = 400, 40_000
    p = synthetic.make_ba_problem(N, P, seed=0)
    opt = oba.BundleAdjusterOptions()
    lens = np.diff(p.pt_offset)
    used = lens >= opt.min_num_view_per_track
    obs_pt = np.repeat(np.arange(P), lens)
    keep = used[obs_pt]
    remap = -np.ones(P, dtype=np.int64)
    remap[used] = np.arange(int(used.sum()))
    prob = oba._BaProblem(N, p.obs_cam[keep].astype(np.int64), remap[obs_pt[keep]], p.obs_xy[keep], p.cam_intr.astype(np.int64),
                          p.intr_model.astype(np.int64), int(p.fixed_cam), int(used.sum()), opt)
    x0 = prob.pack(p.cam_q, p.cam_t, p.pt_xyz[used], p.intr_params)
    for radius in (1e4, 1e6):
        S, b, idx = B.reduced_system(prob, x0, radius)
        Sd = S.toarray()
        pos = -np.ones(prob.pt_col0, dtype=np.int64)
        pos[idx] = np.arange(idx.shape[0])
        M = np.zeros_like(Sd)
        for n in range(N):
            c = [6 * n + j for j in range(6)] + [int(v) for v in prob.intr_col[p.cam_intr[n]] if v >= 0]
            c = pos[np.array(c)]
            c = c[c >= 0]
            if c.size:
                M[np.ix_(c, c)] = Sd[np.ix_(c, c)]
        w = sl.eigh(Sd, M, eigvals_only=True)
        print(f"BA {N} cameras, radius {radius:.0e}: {w.size} eigenvalues of M^-1 S; smallest 9: {' '.join(f'{v:.1e}' for v in w[:9])}; "
              f"1 % / 50 % / 99 % quantiles {np.quantile(w, 0.01):.2f} / {np.quantile(w, 0.5):.2f} / {np.quantile(w, 0.99):.2f}; largest {w[-1]:.2f}")
    N, P = 500, 50_000
    p = synthetic.make_gp_problem(num_cams=N, num_pts=P, seed=0)
    opt = ogp.GlobalPositionerOptions()
    lens = np.diff(p.pt_offset)
    used = lens >= opt.min_num_view_per_track
    obs_pt = np.repeat(np.arange(P), lens)
    keep = used[obs_pt]
    remap = -np.ones(P, dtype=np.int64)
    remap[used] = np.arange(int(used.sum()))
    prob = ogp._GpProblem(N, p.obs_cam[keep].astype(np.int64), remap[obs_pt[keep]], p.obs_dir[keep], p.obs_calibrated[keep], opt, int(used.sum()))
    rng = np.random.default_rng(0)
    x = np.concatenate([100 * rng.uniform(-1, 1, 3 * N), 100 * rng.uniform(-1, 1, 3 * prob.P), np.ones(prob.M)])
    for lam in (1e-4, 1e-6):
        S, b = G.schur_system(prob, x, lam)
        Sd = S.toarray()
        M = np.zeros_like(Sd)
        for n in range(N):
            M[3 * n : 3 * n + 3, 3 * n : 3 * n + 3] = Sd[3 * n : 3 * n + 3, 3 * n : 3 * n + 3]
        w = sl.eigh(Sd, M, eigvals_only=True)
        print(f"GP {N} cameras, damping {lam:.0e}: smallest 6: {' '.join(f'{v:.1e}' for v in w[:6])}; "
              f"1 % / 50 % / 99 % quantiles {np.quantile(w, 0.01):.2f} / {np.quantile(w, 0.5):.2f} / {np.quantile(w, 0.99):.2f}; largest {w[-1]:.2f}")


def variant():
    """Chronopoulos-Gear PCG as cg.hpp runs it (gamma = r.z and delta = z.Az per iteration, p = z + beta p, s = w + beta s,
    alpha = gamma / (delta - beta gamma / alpha_prev)) with the k_cgd_* steps: deflated right-hand side b2 = b - AW E^-1 W^T b,
    start from zero, z <- z - W E^-1 (AW)^T z after every preconditioner application with r.z patched by -(W^T r).(E^-1 (AW)^T z),
    x += W E^-1 W^T b at the end.  Checks iterations and the error against a direct solve."""
    import numpy as np

    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import exp_precond as G
    from glomap_amd import synthetic
    from oracle import gp as ogp

    N, P = 500, 50_000
    p = synthetic.make_gp_problem(num_cams=N, num_pts=P, seed=0)
    opt = ogp.GlobalPositionerOptions()
    lens = np.diff(p.pt_offset)
    used = lens >= opt.min_num_view_per_track
    obs_pt = np.repeat(np.arange(P), lens)
    keep = used[obs_pt]
    remap = -np.ones(P, dtype=np.int64)
    remap[used] = np.arange(int(used.sum()))
    prob = ogp._GpProblem(N, p.obs_cam[keep].astype(np.int64), remap[obs_pt[keep]], p.obs_dir[keep], p.obs_calibrated[keep], opt, int(used.sum()))
    rng = np.random.default_rng(0)
    x = np.concatenate([100 * rng.uniform(-1, 1, 3 * N), 100 * rng.uniform(-1, 1, 3 * prob.P), np.ones(prob.M)])
    S, b = G.schur_system(prob, x, 1e-6)
    A = S.toarray()
    n = A.shape[0]
    Minv = np.zeros_like(A)
    for c in range(N):
        sl = slice(3 * c, 3 * c + 3)
        Minv[sl, sl] = np.linalg.inv(A[sl, sl])
    W = np.zeros((4, n))
    for a in range(3):
        W[a, a::3] = 1.0
    W[3] = x[: 3 * N]

    def cg_gear(W, tol=1e-8, max_it=500):
        k = 0 if W is None else W.shape[0]
        y0 = None
        b2 = b
        if k:
            AW = (A @ W.T).T
            E = W @ AW.T
            Einv = np.linalg.inv(0.5 * (E + E.T))
            y0 = Einv @ (W @ b)
            b2 = b - AW.T @ y0

        def project(z, gamma, r):
            y = Einv @ (AW @ z)
            return z - W.T @ y, gamma - y @ (W @ r)

        xx, r, pv, sv = np.zeros(n), b2.copy(), np.zeros(n), np.zeros(n)
        z = Minv @ r
        gamma, rr = r @ z, r @ r
        if k:
            z, gamma = project(z, gamma, r)
        bb, pg, pa = rr, 0.0, 0.0
        for it in range(max_it):
            if rr <= tol * tol * bb:  # cg_converged at the top of the apply
                break
            w = A @ z
            delta = z @ w
            beta = gamma / pg if it else 0.0
            alpha = gamma / (delta - beta * gamma / pa if it else delta)
            pv, sv = z + beta * pv, w + beta * sv
            xx, r = xx + alpha * pv, r - alpha * sv
            rr, pg, pa = r @ r, gamma, alpha
            z = Minv @ r
            gamma = r @ z
            if k:
                z, gamma = project(z, gamma, r)
        return (xx + W.T @ y0 if k else xx), it

    xs = np.linalg.solve(A, b)
    for Wk, name in ((None, "plain"), (W, "deflated")):
        xr, it = cg_gear(Wk)
        print(f"{name:9s}: {it:3d} iterations, |A x - b| / |b| = {np.linalg.norm(A @ xr - b) / np.linalg.norm(b):.1e}, "
              f"|x - x*| / |x*| = {np.linalg.norm(xr - xs) / np.linalg.norm(xs):.1e}")


def main():
    import numpy as np

    from glomap_amd import synthetic

    stage = sys.argv[1] if len(sys.argv) > 1 else "ba"
    if stage == "spectrum":
        return spectrum()
    if stage == "variant":
        return variant()
    N = int(sys.argv[2]) if len(sys.argv) > 2 else (10_000 if stage == "ba" else 5_000)
    P = int(sys.argv[3]) if len(sys.argv) > 3 else (1_000_000 if stage == "ba" else 500_000)
    res = {}
    for deflate in (False, True):
        out = f"/tmp/exp_deflation_{stage}_{int(deflate)}.npy"
        r, per_step = run(stage, N, P, out, deflate)
        res[deflate] = np.load(out)
        print(("deflated " if deflate else "plain    "), r, "\n    PCG per LM iteration:", " ".join(per_step), flush=True)
    a, b = res[False], res[True]
    if stage == "ba":
        n = a.shape[0] // 7
        print(f"max |dq| {np.abs(a[:4 * n] - b[:4 * n]).max():.2e}, max |dt| {np.abs(a[4 * n:] - b[4 * n:]).max():.2e}")
    else:
        ext = np.linalg.norm(a - a.mean(0), axis=1).max()
        print(f"max centre distance after Sim(3) alignment / extent: {synthetic.center_errors_after_sim3(b, a).max():.2e}")


if __name__ == "__main__":
    main()
