"""Does warm-starting the reduced solve after a REJECTED LM step pay?  (CPU study with the C++ oracle, no GPU.)

After a rejected step Levenberg-Marquardt keeps the linearisation and only shrinks the trust region, so the next
reduced system differs from the last one by its damping alone and the rejected step (optimally rescaled along itself)
is a starting guess for the PCG.  The tolerance and the stopping rule |r| <= tol |b| stay, so the LM trajectory is the
same up to the solves' accuracy.  Runs bundle adjustment (configs[3] by default) twice — cold starts, warm starts
(ORC_WARM_START=1, read once per process: two subprocesses) — at the GPU's PCG tolerance and prints LM / PCG iteration
counts and the distance between the two results.
Usage: python tools/exp_lm_warm_start.py [num_cams] [num_pts] [pcg_tol]"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r"""
import json, sys, time
import numpy as np
sys.path.insert(0, {root!r})
from glomap_amd import synthetic
from oracle import cpu
N, P, tol, out = int(sys.argv[1]), int(sys.argv[2]), float(sys.argv[3]), sys.argv[4]
p = synthetic.make_ba_problem(N, P, seed=0)
t0 = time.time()
ok, q, t, X, intr, s = cpu.ba_solve(p.num_cams, p.pt_offset, p.obs_cam, p.obs_xy, p.cam_intr, p.intr_model, p.fixed_cam,
                                    p.cam_q, p.cam_t, p.pt_xyz, p.intr_params, pcg_tol=tol)
np.save(out, np.concatenate([q.ravel(), t.ravel()]))
print("RESULT " + json.dumps(dict(ok=bool(ok), lm=int(s.iterations), accepted=int(s.successful_steps), pcg=int(s.linear_iterations),
                                   final_cost=float(s.final_cost), seconds=round(time.time() - t0, 1))))
"""


def run(args, warm):
    env = dict(os.environ)
    env.pop("ORC_WARM_START", None)
    if warm:
        env["ORC_WARM_START"] = "1"
    p = subprocess.run([sys.executable, "-c", WORKER.format(root=ROOT), *args], env=env, capture_output=True, text=True)
    for line in p.stdout.splitlines():
        if line.startswith("RESULT "):
            return json.loads(line[7:])
    raise SystemExit(p.stdout[-2000:] + p.stderr[-4000:])


def main():
    import numpy as np

    sys.path.insert(0, ROOT)
    from glomap_amd import so3

    N = sys.argv[1] if len(sys.argv) > 1 else "10000"
    P = sys.argv[2] if len(sys.argv) > 2 else "1000000"
    tol = sys.argv[3] if len(sys.argv) > 3 else "1e-8"
    res = {}
    for warm in (False, True):
        out = f"/tmp/exp_lm_warm_{int(warm)}.npy"
        res[warm] = (run([N, P, tol, out], warm), np.load(out))
        print("warm start" if warm else "cold start", res[warm][0], flush=True)
    a, b = res[False][1], res[True][1]
    n = a.shape[0] // 7
    ang = np.radians(so3.rotation_angle_deg(so3.quat_to_rotmat(a[: 4 * n].reshape(n, 4)), so3.quat_to_rotmat(b[: 4 * n].reshape(n, 4))))
    print(f"max rotation difference {ang.max():.2e} rad, max |t| difference {np.abs(a[4 * n:] - b[4 * n:]).max():.2e}")


if __name__ == "__main__":
    main()
