#!/usr/bin/env python
"""Generates tests/golden/ba_c4_shared_oracle.npz: the CPU oracle's result for bundle adjustment at configs[3] size with ONE
camera shared by all images (synthetic.make_ba_problem(10_000, 1_000_000, seed=0, shared_intrinsics=True)).

Why a fixture instead of running the oracle inside the GPU test: on this input the oracle needs 53 LM iterations with
reduced solves to 1e-14 — ten minutes on 8 cores — and the result does not depend on the thread count (every reduction of
oracle/csrc runs in a fixed order), so the numbers generated here ARE what the GPU box would compute.  The input is
regenerated from the seed by the test; only the oracle's poses travel.

How far the oracle can be trusted on this input: the problem has a free scale gauge (one constant frame, DESIGN.md
section 2.1) and in a few LM steps the block-Jacobi PCG of the oracle stops at a true relative residual of 0.08; the same
solve with the gauge modes and the shared-intrinsics border deflated (ORC_DEFLATE=9, 52 instead of 53 LM iterations) ends
7e-7 rad / 1.5e-6 (centres, relative) away — two orders below the parity bar, so the poses are a usable reference.

Usage: python tests/golden/make_ba_shared_golden.py     (about 10 minutes)"""
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))

from glomap_amd import synthetic  # noqa: E402
from oracle import cpu  # noqa: E402


def main():
    cap = None
    if "--max-iterations" in sys.argv:
        cap = int(sys.argv[sys.argv.index("--max-iterations") + 1])
    p = synthetic.make_ba_problem(10_000, 1_000_000, seed=0, shared_intrinsics=True)
    opt = None
    if cap is not None:
        from oracle import ba as oba

        opt = oba.BundleAdjusterOptions()
        opt.lm.max_num_iterations = cap
    r = cpu.ba_solve(p.num_cams, p.pt_offset, p.obs_cam, p.obs_xy, p.cam_intr, p.intr_model, p.fixed_cam, p.cam_q, p.cam_t,
                     p.pt_xyz, p.intr_params, options=opt, verbose=True)
    assert r[0]
    s = r[5]
    print("LM", s.iterations, "final cost", s.final_cost, "max true relative residual of a reduced solve", s.max_linear_residual)
    name = "ba_c4_shared_oracle.npz" if cap is None else "ba_c4_shared_oracle_it%d.npz" % cap
    np.savez_compressed(Path(__file__).resolve().parent / name, out_q=r[1], out_t=r[2], out_intr=r[4],
                        out_final_cost=s.final_cost, out_initial_cost=s.initial_cost, out_iterations=s.iterations,
                        out_max_linear_residual=s.max_linear_residual,
                        num_obs=p.num_obs, obs_xy_checksum=float(np.sum(p.obs_xy)), cam_t_checksum=float(np.sum(p.cam_t)))


if __name__ == "__main__":
    main()
