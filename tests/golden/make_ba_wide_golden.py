#!/usr/bin/env python
"""Generates tests/golden/ba_c4_full_opencv_oracle.npz: the C++ CPU oracle's result (oracle/csrc/orc_ba_wide.cc, exact
elimination) for bundle adjustment at configs[3] size through a 12-parameter camera model —
synthetic.make_ba_problem_wide(10_000, 1_000_000, "full_opencv", seed=0, num_intr_groups=100): 100 FULL_OPENCV cameras shared
round-robin by the 10 000 images (one camera per IMAGE leaves twelve parameters to a few hundred observations: the rational
distortion terms k4 .. k6 are then nearly degenerate with k1 .. k3, LM creeps along that valley for as long as it is allowed to
and "the end point" is not defined to 1e-4 rad — tools/exp_ba_wide_fullsize.py 2000 200000 0).

Why a fixture: 26 LM iterations with reduced solves to 1e-14 take four minutes on 16 cores, and the result does not depend on
the thread count (every reduction of oracle/csrc runs in a fixed order).  The input is regenerated from the seed by the test
(checksums below); only the oracle's poses and intrinsics travel.

Usage: python tests/golden/make_ba_wide_golden.py     (about 8 minutes on 8 cores)"""
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))

from glomap_amd import synthetic  # noqa: E402
from oracle import cpu  # noqa: E402


def main():
    p = synthetic.make_ba_problem_wide(10_000, 1_000_000, "full_opencv", seed=0, num_intr_groups=100)
    r = cpu.ba_solve(p.num_cams, p.pt_offset, p.obs_cam, p.obs_xy, p.cam_intr, p.intr_model, p.fixed_cam, p.cam_q, p.cam_t,
                     p.pt_xyz, p.intr_params, verbose=True)
    assert r[0]
    s = r[5]
    print("LM", s.iterations, "final cost", s.final_cost, "max true relative residual of a reduced solve", s.max_linear_residual)
    np.savez_compressed(Path(__file__).resolve().parent / "ba_c4_full_opencv_oracle.npz", out_q=r[1], out_t=r[2], out_intr=r[4],
                        out_final_cost=s.final_cost, out_initial_cost=s.initial_cost, out_iterations=s.iterations,
                        out_max_linear_residual=s.max_linear_residual,
                        num_obs=p.num_obs, obs_xy_checksum=float(np.sum(p.obs_xy)), cam_t_checksum=float(np.sum(p.cam_t)))


if __name__ == "__main__":
    main()
