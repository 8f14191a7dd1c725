"""Host-core probe for the CPU oracle: cgroup limits of the box and wall time of one small GP solve per thread count."""
import os, sys, time, subprocess
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for f in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "/sys/fs/cgroup/cpu/cpu.cfs_period_us", "/proc/loadavg"):
    try:
        print(f, open(f).read().strip())
    except Exception as e:
        print(f, "n/a")
print("cpu_count", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)), "OMP env", {k: v for k, v in os.environ.items() if k.startswith(("OMP", "GOMP"))}, flush=True)
if len(sys.argv) > 1:
    import numpy as np
    from glomap_amd import synthetic
    from oracle import cpu
    p = synthetic.make_gp_problem(300, 20000, seed=5)
    t0 = time.time()
    ok, c, X, s = cpu.gp_solve(p.num_cams, p.pt_offset, p.obs_cam, p.obs_dir, p.obs_calibrated, p.cam_center, p.pt_xyz, threads=int(sys.argv[1]))
    print("threads", sys.argv[1], "->", s.threads, "GP 300/20k: %.2f s" % (time.time() - t0), s.iterations, s.linear_iterations, flush=True)
else:
    for t in (8, 32, 256):
        for env in ({},):
            try:
                out = subprocess.run([sys.executable, __file__, str(t)], capture_output=True, text=True, timeout=40, env={**os.environ, **env})
                print(env, out.stdout.strip().splitlines()[-1], flush=True)
            except subprocess.TimeoutExpired:
                print(env, "threads", t, "TIMEOUT 40 s", flush=True)
