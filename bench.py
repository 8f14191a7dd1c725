#!/usr/bin/env python
"""bench.py — headline benchmark of the RA -> GP -> BA hot path on MI355X.

Contract: `python bench.py --gpus N --steps K --warmup W` prints ONE JSON line on rank 0.
A "step" is one full pass of the hot path over one batch of synthetic input:

  workload "ra_c2"  (default; BASELINE.json configs[1]): one complete rotation-averaging solve
                    (MST init + L1-ADMM + IRLS, reference defaults) of the synthetic ring view graph
                    with 1k cameras / 50k relative-pose edges.          metric: view-graph edges/s
  workload "gp_c3"  (configs[2]) global positioning, 5k cameras / 500k tracks / ~3M observations
                    metric: track-obs/s per LM iteration
  workload "ba_c4"  (configs[3] on ONE GPU) bundle adjustment, 10k cameras / 1M tracks / ~5M
                    observations                                        metric: track-obs/s per BA (LM) iteration

The default run times ra_c2 (the configuration the metric is quoted on that fits one GPU) and adds
one measured solve each of gp_c3 and ba_c4 under "extra" (skip with --no-extra).

Inputs are resident in HBM (glomap_amd DeviceArrays) before the timed region starts.  With
--gpus N > 1 (launched by torch.distributed.run, one rank per GPU) the track set grows with N (weak
scaling): every rank owns an equal shard of the tracks, camera vectors are replicated and the
reduced-system vectors are all-reduced over RCCL inside libgsfm.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec (/opt/skills/guides/MI355X_MICROARCH.md)
KERNEL_RA, KERNEL_GP, KERNEL_BA = 0, 1, 2


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="ra_c2", choices=["ra_c2", "gp_c3", "ba_c4"])
    ap.add_argument("--scale", type=float, default=1.0, help="scale the GP/BA problem size (cameras and tracks)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra", action="store_true", help="skip the GP/BA side measurements of the default run")
    return ap.parse_args()


def main():
    args = parse()
    from glomap_amd import _lib, build

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world == 1 and args.gpus > 1:
        raise SystemExit("--gpus N > 1 must be launched with torch.distributed.run (one rank per GPU)")
    if rank == 0:
        build.build_lib(verbose=False)
    # One HIP runtime per process: libgsfm's (ROCm, the one hipcc/rocprofv3 belong to).  PyTorch
    # wheels bundle a second copy of the runtime, so torch is used for the control plane only
    # (torch.distributed rendezvous / barrier / max-over-ranks on the gloo backend); device
    # memory, streams and the RCCL communicator are libgsfm's own (gsfm_device_*, gsfm_comm_*).
    # ctx.synchronize() below is the hipStreamSynchronize that torch.cuda.synchronize() would be.
    dist = None
    if world > 1:
        import torch.distributed as dist

        dist.init_process_group("gloo")
        dist.barrier()  # rank 0 finished building
    ctx = _lib.Context(local_rank)  # raises GSFM_ERR_NO_DEVICE without an MI355X: no CPU fallback
    if world > 1:
        uid = [_lib.comm_unique_id() if rank == 0 else None]
        dist.broadcast_object_list(uid, src=0)
        ctx.comm_init(uid[0], rank, world)

    def barrier():
        ctx.synchronize()
        if dist is not None:
            dist.barrier()
        ctx.synchronize()

    env = dict(args=args, ctx=ctx, rank=rank, world=world, barrier=barrier, dist=dist)
    if args.workload == "ra_c2":
        out = bench_ra(**env)
        if not args.no_extra and world == 1:
            extra = {}
            sub_args = argparse.Namespace(**{**vars(args), "steps": 1, "warmup": 0})
            for name, fn in (("gp_c3", bench_gp), ("ba_c4", bench_ba)):
                try:
                    sub = fn(**{**env, "args": sub_args})
                    extra[name] = {k: sub[k] for k in ("metric", "value", "unit", "ms_per_step", "config", "roofline",
                                                       "cpu_baseline")}
                except Exception as e:  # report, never hide
                    extra[name] = {"error": repr(e)}
            out["extra"] = extra
    elif args.workload == "gp_c3":
        out = bench_gp(**env)
    else:
        out = bench_ba(**env)
    if rank == 0:
        print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()
    ctx.close()


def timed_steps(step_fn, steps, warmup, barrier, dist):
    for _ in range(warmup):
        step_fn()
    barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        step_fn()
    barrier()
    dt = time.perf_counter() - t0
    if dist is not None:
        import torch

        t = torch.tensor([dt], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    return dt


def profiled_step(ctx, kernel_id, step_fn):
    """One extra, event-instrumented step: every launch of the dominant kernel is bracketed by a HIP
    event pair on the ctx stream (gsfm_ctx_profile_*).  Returns (launches, avg_ms)."""
    ctx.profile_enable(True)
    ctx.profile_read(kernel_id)
    step_fn()
    launches, total_ms = ctx.profile_read(kernel_id)
    ctx.profile_enable(False)
    return launches, (total_ms / launches if launches else None)


def roofline(kernel, bytes_per_launch, launches, avg_ms, note):
    achieved = bytes_per_launch / (avg_ms * 1e-3) / 1e9 if avg_ms else None
    return {
        "bound": "hbm",
        "kernel": kernel,
        "achieved": achieved,
        "peak": HBM_PEAK_GBS,
        "unit": "GB/s",
        "frac": (achieved / HBM_PEAK_GBS) if achieved else None,
        "traffic": None,
        "bytes_per_launch": bytes_per_launch,
        "avg_kernel_us": avg_ms * 1e3 if avg_ms else None,
        "launches_in_profiled_step": launches,
        "note": note,
    }


def base_line(metric, value, unit, world, args, dt, config, roof, cpu, ctx):
    return {
        "metric": metric,
        "value": value,
        "unit": unit,
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": dt / args.steps * 1e3,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f64",
        "data": "synthetic",
        "config": config,
        "roofline": roof,
        "cpu_baseline": cpu,
        "device": ctx.device_name(),
    }


# ----------------------------------------------------------------------------------------------
# rotation averaging, configs[1]
# ----------------------------------------------------------------------------------------------
def bench_ra(args, ctx, rank, world, barrier, dist):
    import numpy as np

    from glomap_amd import estimators, so3, synthetic

    if world > 1:
        raise SystemExit("ra_c2 is a single-GPU workload (1.2 MB working set); use --workload ba_c4 / gp_c3 with --gpus N")
    N, succ = 1000, 50
    p = synthetic.make_ring_view_graph(N, succ, seed=0)
    E = p.num_edges
    opt = estimators.RotationEstimatorOptions()
    pd = type(p)(
        num_nodes=p.num_nodes,
        edge_i=ctx.to_device(p.edge_i),
        edge_j=ctx.to_device(p.edge_j),
        edge_q=ctx.to_device(p.edge_q),
        edge_weight=ctx.to_device(p.edge_weight),
        edge_ninl=ctx.to_device(p.edge_ninl),
        node_aa0=ctx.to_device(p.node_aa0),
        fixed_node=0,
    )
    rot = pd.node_aa0.clone()
    last = {}

    def step():
        rot.copy_from(pd.node_aa0)
        rc, _, rep = estimators.ra_solve(pd, opt, ctx=ctx, rot_inout=rot)
        if rc != 0:
            raise RuntimeError(f"gsfm_ra_solve failed: {rc}")
        last.update(rep)

    dt = timed_steps(step, args.steps, args.warmup, barrier, dist)
    value = E * args.steps / dt
    launches, avg_ms = profiled_step(ctx, KERNEL_RA, step)
    roof = roofline(
        "k_pcg_dir_fused (RA weighted-Laplacian SpMV, 3 RHS)",
        16.0 * E + 48.0 * N,  # 16 E + 48 N per SpMV
        launches,
        avg_ms,
        "configs[1] working set (1.2 MB) is L2-resident: launch/latency-bound at this size, see DESIGN.md",
    )
    err = synthetic.rotation_errors_deg(so3.aa_to_rotmat(rot.numpy()), p.gt_R)
    cpu = None if (args.no_cpu_baseline or rank != 0) else cpu_baseline_ra(p)
    config = {
        "workload": "configs[1]: synthetic ring view graph, 1k cameras / 50k relative-pose edges, full rotation "
        "averaging (MST init + L1-ADMM + IRLS, reference defaults)",
        "cameras": N,
        "edges": E,
        "parallelism": "single GPU",
        "l1_iterations": last.get("iterations_l1"),
        "irls_iterations": last.get("iterations_irls"),
        "pcg_iterations_per_step": last.get("linear_iterations"),
        "median_rot_err_deg_vs_gt": float(np.median(err)),
    }
    return base_line("view-graph edges/sec (RA)", value, "edges/s", world, args, dt, config, roof, cpu, ctx)


def cpu_baseline_ra(p):
    """Restated CPU oracle (numpy + scipy SuperLU — NOT Ceres/CHOLMOD) on the same view graph."""
    from oracle import ra as ora

    t0 = time.perf_counter()
    n = 0
    while True:
        ora.estimate_rotations(p.num_nodes, p.edge_i, p.edge_j, p.edge_q, p.edge_weight, p.edge_ninl, p.node_aa0,
                               p.fixed_node)
        n += 1
        dt = time.perf_counter() - t0
        if dt > 10.0 or n >= 8:
            break
    return {
        "value": p.num_edges * n / dt,
        "unit": "edges/s",
        "cores": 1,
        "host_cores_available": os.cpu_count(),
        "kind": "port",
        "sample": f"{n} full RA solves of the same 1k-camera / 50k-edge view graph (restated CPU oracle, numpy + "
        "scipy SuperLU; not Ceres/CHOLMOD)",
    }


# ----------------------------------------------------------------------------------------------
# global positioning, configs[2]
# ----------------------------------------------------------------------------------------------
def shard_tracks(pt_offset, rank, world):
    """Contiguous, observation-balanced track range of this rank."""
    import numpy as np

    M = int(pt_offset[-1])
    lo = int(np.searchsorted(pt_offset, (M * rank) // world, side="left"))
    hi = int(np.searchsorted(pt_offset, (M * (rank + 1)) // world, side="left")) if rank + 1 < world else len(pt_offset) - 1
    return lo, hi


def bench_gp(args, ctx, rank, world, barrier, dist):
    import numpy as np

    from glomap_amd import estimators, synthetic
    from glomap_amd.flat import GpProblem

    ncam = int(5000 * args.scale)
    npts = int(500_000 * args.scale) * world  # weak scaling: tracks per GPU fixed
    p = synthetic.make_gp_problem(ncam, npts, seed=0)
    lo, hi = shard_tracks(p.pt_offset, rank, world)
    o0, o1 = int(p.pt_offset[lo]), int(p.pt_offset[hi])
    M_total = p.num_obs
    pd = GpProblem(
        num_cams=ncam,
        num_pts=hi - lo,
        pt_offset=ctx.to_device((p.pt_offset[lo : hi + 1] - o0).astype(np.int64)),
        obs_cam=ctx.to_device(p.obs_cam[o0:o1]),
        obs_dir=ctx.to_device(p.obs_dir[o0:o1]),
        obs_calibrated=ctx.to_device(p.obs_calibrated[o0:o1]),
        cam_center=ctx.to_device(p.cam_center),
        pt_xyz=ctx.to_device(p.pt_xyz[lo:hi]),
    )
    opt = estimators.GlobalPositionerOptions()
    last = {}
    res = {}

    def step():
        rc, cen, xyz, rep = estimators.gp_solve(pd, opt, ctx=ctx)
        if rc != 0:
            raise RuntimeError(f"gsfm_gp_solve failed: {rc}")
        last.update(rep)
        res["cen"] = cen

    dt = timed_steps(step, args.steps, args.warmup, barrier, dist)
    iters = max(1, last["iterations"])
    value = M_total * iters * args.steps / dt  # observations swept per second, per LM iteration
    launches, avg_ms = profiled_step(ctx, KERNEL_GP, step)
    M_loc, P_loc = o1 - o0, hi - lo
    roof = roofline(
        "k_gp_schur_matvec (implicit Schur product over BATA observations)",
        41.0 * M_loc + 48.0 * P_loc + 48.0 * ncam,  # SURVEY.md §8d K-GP-res
        launches,
        avg_ms,
        "one launch = one product of the 3N reduced camera system with a vector",
    )
    err = synthetic.center_errors_after_sim3(res["cen"].numpy(), p.gt_center)
    cpu = None if (args.no_cpu_baseline or rank != 0) else cpu_baseline_gp()
    config = {
        "workload": "configs[2]: synthetic 5k cameras / 500k tracks per GPU, global positioning (BATA, Huber 0.1, "
        "random init seed 1, reference defaults)",
        "cameras": ncam,
        "tracks": npts,
        "observations": M_total,
        "parallelism": f"track-shard x{world}",
        "lm_iterations": last["iterations"],
        "successful_steps": last["successful_steps"],
        "pcg_iterations_per_step": last["linear_iterations"],
        "final_cost": last["final_cost"],
        "solves_per_s_in_obs": M_total * args.steps / dt,
        "median_center_err_vs_gt": float(np.median(err)),
    }
    return base_line("track-obs/sec per LM iteration (GP)", value, "obs/s", world, args, dt, config, roof, cpu, ctx)


def cpu_baseline_gp():
    from glomap_amd import synthetic
    from oracle import gp as ogp

    p = synthetic.make_gp_problem(150, 8000, seed=0)
    t0 = time.perf_counter()
    ok, c, X, s = ogp.solve(p.num_cams, p.pt_offset, p.obs_cam, p.obs_dir, p.obs_calibrated, p.cam_center, p.pt_xyz)
    dt = time.perf_counter() - t0
    return {
        "value": p.num_obs * max(1, s.iterations) / dt,
        "unit": "obs/s",
        "cores": 1,
        "host_cores_available": os.cpu_count(),
        "kind": "port",
        "sample": f"one GP solve of a 150-camera / 8k-track / {p.num_obs}-observation sample of the same generator "
        f"({s.iterations} LM iterations; restated CPU oracle: numpy + exact Schur elimination, not Ceres)",
    }


# ----------------------------------------------------------------------------------------------
# bundle adjustment, configs[3] on one GPU (and track-sharded over N GPUs)
# ----------------------------------------------------------------------------------------------
def bench_ba(args, ctx, rank, world, barrier, dist):
    import numpy as np

    from glomap_amd import estimators, so3, synthetic
    from glomap_amd.flat import BaProblem

    ncam = int(10_000 * args.scale)
    npts = int(1_000_000 * args.scale) * world  # weak scaling: tracks per GPU fixed
    p = synthetic.make_ba_problem(ncam, npts, seed=0, shared_intrinsics=False)
    lo, hi = shard_tracks(p.pt_offset, rank, world)
    o0, o1 = int(p.pt_offset[lo]), int(p.pt_offset[hi])
    M_total = p.num_obs
    pd = BaProblem(
        num_cams=ncam,
        num_pts=hi - lo,
        num_intr=p.num_intr,
        pt_offset=ctx.to_device((p.pt_offset[lo : hi + 1] - o0).astype(np.int64)),
        obs_cam=ctx.to_device(p.obs_cam[o0:o1]),
        obs_xy=ctx.to_device(p.obs_xy[o0:o1]),
        cam_intr=ctx.to_device(p.cam_intr),
        cam_q=ctx.to_device(p.cam_q),
        cam_t=ctx.to_device(p.cam_t),
        pt_xyz=ctx.to_device(p.pt_xyz[lo:hi]),
        intr_model=ctx.to_device(p.intr_model),
        intr_params=ctx.to_device(p.intr_params),
        fixed_cam=0,
    )
    opt = estimators.BundleAdjusterOptions()
    last = {}
    res = {}

    def step():
        rc, q, t, X, intr, rep = estimators.ba_solve(pd, opt, ctx=ctx)
        if rc != 0:
            raise RuntimeError(f"gsfm_ba_solve failed: {rc}")
        last.update(rep)
        res["q"], res["t"] = q, t

    dt = timed_steps(step, args.steps, args.warmup, barrier, dist)
    iters = max(1, last["iterations"])
    value = M_total * iters * args.steps / dt
    launches, avg_ms = profiled_step(ctx, KERNEL_BA, step)
    M_loc, P_loc = o1 - o0, hi - lo
    roof = roofline(
        "k_ba_schur_matvec (matrix-free implicit Schur product over reprojection observations)",
        28.0 * M_loc + 392.0 * ncam + 120.0 * P_loc + 32.0 * p.num_intr,  # SURVEY.md §8d K-BA-res
        launches,
        avg_ms,
        "one launch = one product of the reduced camera system (6N + 8K) with a vector",
    )
    R = so3.quat_to_rotmat(res["q"].numpy())
    Rg = so3.quat_to_rotmat(p.gt_q)
    rot_err = synthetic.rotation_errors_deg(R, Rg)
    cpu = None if (args.no_cpu_baseline or rank != 0) else cpu_baseline_ba()
    config = {
        "workload": "configs[3] on one GPU per rank: synthetic 10k cameras / 1M tracks / ~5M observations per GPU, "
        "bundle adjustment (SIMPLE_RADIAL per image, Huber 1 px, reference defaults), start = GT + noise",
        "cameras": ncam,
        "tracks": npts,
        "observations": M_total,
        "parallelism": f"track-shard x{world}",
        "lm_iterations": last["iterations"],
        "successful_steps": last["successful_steps"],
        "pcg_iterations_per_step": last["linear_iterations"],
        "initial_cost": last["initial_cost"],
        "final_cost": last["final_cost"],
        "median_rot_err_deg_vs_gt": float(np.median(rot_err)),
    }
    return base_line("track-obs/sec per BA iteration", value, "obs/s", world, args, dt, config, roof, cpu, ctx)


def cpu_baseline_ba():
    from glomap_amd import synthetic
    from oracle import ba as oba

    p = synthetic.make_ba_problem(200, 10_000, seed=0)
    t0 = time.perf_counter()
    ok, q, t, X, intr, s = oba.solve(p.num_cams, p.pt_offset, p.obs_cam, p.obs_xy, p.cam_intr, p.intr_model,
                                     p.fixed_cam, p.cam_q, p.cam_t, p.pt_xyz, p.intr_params)
    dt = time.perf_counter() - t0
    return {
        "value": p.num_obs * max(1, s.iterations) / dt,
        "unit": "obs/s",
        "cores": 1,
        "host_cores_available": os.cpu_count(),
        "kind": "port",
        "sample": f"one BA solve of a 200-camera / 10k-track / {p.num_obs}-observation sample of the same generator "
        f"({s.iterations} LM iterations; restated CPU oracle: numpy + exact Schur elimination, not Ceres)",
    }


if __name__ == "__main__":
    main()
