"""GPU: the reference's GlobalMapper::Solve (oracle/_ref/libref_dropin_mapper.so: global_mapper.cc compiled unmodified on the adapter
classes) at sizes where the adapter's own work shows: a ring that looks outward, every image seeing its sector only
(synthetic.make_pipeline_scene(layout="outward")).  Prints, per size and drop-in build, the wall time of Solve and where it went
inside include/gsfm_glomap_adapter.hpp — packing the reference's containers, inside libgsfm, writing back — per entry point
(gsfm_glomap::AdapterTimings).  With `ref` as last argument the all-reference build (which = 0, CPU, Ceres stand-in) runs on the
smallest size as well and the final poses are compared.
Usage: python tools/exp_dropin_mapper_scale.py [ref]"""
import os
import sys
import time

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "."), "tests"))
import numpy as np

from glomap_amd import so3, synthetic
import test_dropin_reference_mapper as T

with_ref = len(sys.argv) > 1 and sys.argv[1] == "ref"
sizes = [(300, 20000, 10), (1000, 60000, 12)]
for N, P, succ in sizes:
    t0 = time.perf_counter()
    s = synthetic.make_pipeline_scene(n_images=N, n_points=P, seed=0, pixel_noise=0.5, num_succ=succ, layout="outward")
    print(f"scene: {N} images, {P} points, {int(s['feat_offset'][-1])} features, {len(s['pair_image1'])} pairs, {int(s['pair_offset'][-1])} matches "
          f"(generated in {time.perf_counter() - t0:.1f} s)", flush=True)
    res = {}
    for which in ([0] if with_ref and N <= 300 else []) + [1, 2, 3, 1, 2, 3]:
        r = T._solve(which, s)
        res[which] = r
        R, c = T._poses(r)
        reg = r["frame_registered"]
        rot = synthetic.rotation_errors_deg(R[reg], s["gt_R"][reg]).max()
        cen = synthetic.center_errors_after_sim3(c[reg], s["gt_center"][reg]).max() / synthetic.scene_extent(s["gt_center"][reg])
        print(f"  which={which}: ok={r['ok']} Solve {r['seconds']:.3f} s; {int(reg.sum())} images, {r['num_tracks']} tracks / {r['num_observations']} observations; "
              f"vs ground truth {rot:.2e} deg / {cen:.2e}", flush=True)
        tot = dict(pack=0.0, call=0.0, unpack=0.0)
        for name, t in r["adapter_timings"].items():
            print(f"      {name:45s} x{t['calls']:<3d} pack {t['pack']*1e3:9.2f} ms   libgsfm {t['call']*1e3:9.2f} ms   unpack {t['unpack']*1e3:8.2f} ms" + (f"   iterations {t['iterations']} / PCG {t['linear_iterations']}" if t['iterations'] else ""))
            for k in tot:
                tot[k] += t[k]
        if r["adapter_timings"]:
            rest = r["seconds"] - sum(tot.values())
            print(f"      {'adapter total':45s}      pack {tot['pack']*1e3:9.2f} ms   libgsfm {tot['call']*1e3:9.2f} ms   unpack {tot['unpack']*1e3:8.2f} ms   "
                  f"| rest of Solve (reference CPU code: track establishment, controller, unswitched processors) {rest*1e3:.1f} ms")
    if 0 in res:
        a = res[0]
        for which in (1, 2):
            b = res[which]
            reg = a["frame_registered"]
            Ra, ca = T._poses(a)
            Rb, cb = T._poses(b)
            rot = np.linalg.norm(so3.quat_to_aa(so3.rotmat_to_quat(np.einsum("nij,nkj->nik", Ra[reg], Rb[reg]))), axis=1).max()
            cen = np.linalg.norm(ca[reg] - cb[reg], axis=1).max() / synthetic.scene_extent(ca[reg])
            same = np.array_equal(a["track_id"], b["track_id"]) and np.array_equal(a["track_len"], b["track_len"]) and np.array_equal(a["pair_valid"], b["pair_valid"])
            print(f"[parity] DROP-IN GlobalMapper::Solve {N} images / {a['num_observations']} observations, which={which} vs the reference's own: "
                  f"rotations {rot:.2e} rad, centres {cen:.2e} of the extent, same tracks / pairs: {same}; reference code {a['seconds']:.1f} s (CPU, Ceres stand-in) "
                  f"vs {b['seconds']:.2f} s")
