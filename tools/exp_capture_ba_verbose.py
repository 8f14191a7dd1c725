"""GPU, diagnostic: the mapper's second bundle adjustment on the outward ring replayed with the dense path forced (knob gp_dense = 2)
and GSFM_VERBOSE set: LM iterations and the residual of the dense solves.  Usage: python tools/exp_capture_ba_verbose.py"""
import glob, os, sys, tempfile
root = os.environ.get("GRAFT_REPO_ROOT", ".")
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
dump = tempfile.mkdtemp(prefix="gsfm_dump_")
os.environ["GSFM_DUMP_DIR"] = dump
from glomap_amd import _lib, estimators, flatio, synthetic
import test_dropin_reference_mapper as T
s = synthetic.make_pipeline_scene(n_images=300, n_points=20000, seed=0, pixel_noise=0.5, num_succ=10, layout="outward")
ctx0 = _lib.Context(-1)
ctx0.set_knob("gp_dense", 1)
r = T._solve(1, s)
del os.environ["GSFM_DUMP_DIR"]
files = sorted(glob.glob(os.path.join(dump, "ba_*.gsfm")))
os.environ["GSFM_VERBOSE"] = "1"
ctx = _lib.Context(-1)
rec = flatio.load(files[1])
p, opt = flatio.to_problem(rec)
ctx.set_knob("gp_dense", int(sys.argv[1]) if len(sys.argv) > 1 else 2)
dd = tempfile.mkdtemp(prefix="gsfm_dense_")
os.environ["GSFM_DUMP_DENSE"] = dd
print(estimators.ba_solve(p, opt, ctx=ctx)[-1])
import numpy as np
for f in sorted(glob.glob(os.path.join(dd, "dense_*.bin")), key=lambda x: int(x.rsplit("_", 1)[1][:-4])):
    n, ld = (int(v) for v in os.path.basename(f).split("_")[1:3])
    a = np.fromfile(f)
    S, b = a[:ld * ld].reshape(ld, ld)[:n, :n], a[ld * ld:ld * ld + n]
    asym = np.abs(S - S.T).max() / np.abs(S).max()
    d = 1.0 / np.sqrt(np.where(np.diag(S) > 0, np.diag(S), 1.0))
    E = (S + S.T) * 0.5 * d[:, None] * d[None, :]
    w = np.linalg.eigvalsh(E)
    x = np.linalg.solve(S, b)
    print(os.path.basename(f), "asymmetry %.1e" % asym, "equilibrated eigenvalues min %.3e max %.3e (negative: %d), condition %.2e" % (w[0], w[-1], int((w < 0).sum()), w[-1] / abs(w[0])),
          "numpy LU residual %.1e" % (np.linalg.norm(b - S @ x) / np.linalg.norm(b)), flush=True)
