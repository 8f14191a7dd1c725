#!/usr/bin/env python
"""Replays flat problem files (`*.gsfm`, written by libgsfm with GSFM_DUMP_DIR set — e.g. from inside a GLOMAP build
linked against it) through the C ABI and compares with the results stored in the file.

    python tools/replay.py dump/*.gsfm
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from glomap_amd import _lib, estimators, flatio, so3, synthetic


def main(paths):
    ctx = _lib.Context(-1)
    for path in paths:
        rec = flatio.load(path)
        p, opt = flatio.to_problem(rec)
        t0 = time.time()
        if rec.kind == "ra":
            rc, rot, rep = estimators.ra_solve(p, opt, ctx=ctx)
            ref = rec.arrays.get("out_rot_aa")
            diff = "" if ref is None else "max rotation difference vs stored result %.3e rad" % np.radians(
                so3.rotation_angle_deg(so3.aa_to_rotmat(rot), so3.aa_to_rotmat(ref))).max()
            size = f"{p.num_nodes} nodes / {p.num_edges} edges"
        elif rec.kind == "gp":
            rc, cen, X, rep = estimators.gp_solve(p, opt, ctx=ctx)
            ref = rec.arrays.get("out_cam_center")
            diff = "" if ref is None else "max centre difference vs stored result (after Sim3) %.3e" % synthetic.center_errors_after_sim3(cen, ref).max()
            size = f"{p.num_cams} cameras / {p.num_pts} tracks / {p.num_obs} observations"
        else:
            rc, q, t, X, intr, rep = estimators.ba_solve(p, opt, ctx=ctx)
            ref = rec.arrays.get("out_cam_q")
            diff = "" if ref is None else "max rotation difference vs stored result %.3e rad" % np.radians(
                so3.rotation_angle_deg(so3.quat_to_rotmat(q), so3.quat_to_rotmat(ref))).max()
            size = f"{p.num_cams} cameras / {p.num_pts} tracks / {p.num_obs} observations"
        print(f"{path}: {rec.kind} {size}: status {rc} (stored {rec.status}), {rep['iterations']} iterations "
              f"(stored {rec.report.get('iterations')}), final cost {rep['final_cost']:.6e}, {1e3 * (time.time() - t0):.1f} ms; {diff}")
    ctx.close()


if __name__ == "__main__":
    main(sys.argv[1:])
