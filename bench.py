#!/usr/bin/env python
"""bench.py — headline benchmark of the RA -> GP -> BA hot path on MI355X.

Contract: `python bench.py --gpus N --steps K --warmup W` prints ONE JSON line on rank 0.
A "step" is one full pass of the hot path over one batch of synthetic input.

  workload "pipeline_c4" (default; BASELINE.json configs[3], the north_star problem, which fits one GPU at ~2 GB):
                    one rotation-averaging solve (10k cameras / 500k relative-pose edges, MST init + L1-ADMM + IRLS)
                    + one global-positioning solve (10k cameras / 1M tracks / ~6M observations, random start)
                    + one bundle adjustment (10k cameras / 1M tracks / ~5M observations, SIMPLE_RADIAL per image),
                    every stage with the reference's default options on the synthetic inputs SURVEY.md section 8d
                    defines.                                 value: track observations through RA+GP+BA per second
                    BASELINE.json's two sub-metrics — view-graph edges/s (RA+GP) and track-obs/s per BA iteration —
                    are reported beside it in "submetrics".
  workload "ra_c2"  (configs[1]) one RA solve of the 1k-camera / 50k-edge ring graph      metric: view-graph edges/s
  workload "gp_c3"  (configs[2]) global positioning, 5k cameras / 500k tracks / ~3M observations
  workload "ba_c4"  bundle adjustment of configs[3] alone

Inputs are resident in HBM (glomap_amd DeviceArrays) before the timed region starts.  With --gpus N > 1 (launched by
torch.distributed.run, one rank per GPU) the SAME fixed problem is split N ways (strong scaling): GP and BA shard by
track — every rank owns an observation-balanced range of tracks, camera vectors are replicated and the reduced-system
vectors are all-reduced over RCCL inside libgsfm on every PCG iteration; RA (154 ms, 3.6 MB of node state) runs
replicated on every rank without a collective.
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
KERNEL_RA, KERNEL_GP, KERNEL_BA, KERNEL_GP_B, KERNEL_BA_B, KERNEL_RA_GJ, KERNEL_FILTER, KERNEL_TRACK_HOOK, KERNEL_GP_WSUM = 0, 1, 2, 3, 4, 5, 6, 7, 8
F64_MFMA_PEAK_TFLOPS = 78.6  # MI355X FP64 matrix, AMD datasheet (the local guide lists no f64 MFMA figure)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="pipeline_c4", choices=["pipeline_c4", "ra_c2", "gp_c3", "ba_c4"])
    ap.add_argument("--scale", type=float, default=1.0, help="scale the GP/BA problem size (cameras and tracks)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra", action="store_true", help="skip the GP/BA side measurements of the default run")
    return ap.parse_args()


_RESULT_FD = None


def emit_result(line: dict):
    """The ONE JSON line of the contract, on the process's original stdout."""
    data = (json.dumps(line) + "\n").encode()
    if _RESULT_FD is None:
        sys.stdout.write(data.decode())
        sys.stdout.flush()
    else:
        os.write(_RESULT_FD, data)


def main():
    global _RESULT_FD
    args = parse()
    # stdout carries exactly one JSON line: whatever libraries print from here on (RCCL prints a version banner to
    # stdout when a communicator is created) goes to stderr, the result goes to the saved descriptor.
    sys.stdout.flush()
    _RESULT_FD = os.dup(1)
    os.dup2(2, 1)
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
    # GSFM_BENCH_SINGLE_DEVICE=1 (validation only): every rank uses GPU 0 — lets the N > 1 control flow run
    # on a one-GPU box together with GSFM_BENCH_HOST_COMM=1 (host-staged all-reduce instead of RCCL).
    dev = 0 if os.environ.get("GSFM_BENCH_SINGLE_DEVICE") else local_rank
    ctx = _lib.Context(dev)  # raises GSFM_ERR_NO_DEVICE without an MI355X: no CPU fallback

    def barrier():
        ctx.synchronize()
        if dist is not None:
            dist.barrier()
        ctx.synchronize()

    env = dict(args=args, ctx=ctx, rank=rank, world=world, barrier=barrier, dist=dist)
    if args.workload == "pipeline_c4":
        out = bench_pipeline(**env)
        if not args.no_extra and world == 1:
            out["extra"] = run_extras(env, args, world, rank, out)
    elif args.workload == "ra_c2":
        out = bench_ra(**env)
    elif args.workload == "gp_c3":
        comm_init(ctx, dist, rank, world)
        out = bench_gp(**env)
    else:
        comm_init(ctx, dist, rank, world)
        out = bench_ba(**env)
    if world > 1 and isinstance(out.get("config"), dict):
        out["config"].setdefault("transport", getattr(ctx, "_transport", None))
    if rank == 0:
        emit_result(out)
    if dist is not None:
        dist.destroy_process_group()
    ctx.close()


def comm_init(ctx, dist, rank, world):
    """Transport of the sharded solves (one rank per GPU).  Default: the peer-mailbox all-reduce of libgsfm (csrc/peer.hpp; gloo
    carries the memory handles), checked by its collective self test on every rank; if any rank fails to map a peer or the
    self test fails, all ranks fall back to an RCCL communicator (rank 0 creates the unique id).  GSFM_BENCH_TRANSPORT =
    peer | rccl | host (host = the validation transport, never the measured configuration) overrides."""
    if world <= 1 or getattr(ctx, "_comm_ready", False):
        return
    import torch

    from glomap_amd import _lib, sharding

    want = os.environ.get("GSFM_BENCH_TRANSPORT", "host" if os.environ.get("GSFM_BENCH_HOST_COMM") else "peer")
    ctx._transport = None
    if want == "host":
        ctx.comm_init_host(sharding.host_allreduce(dist), rank, world)
        ctx._transport = "host-staged"
    if want == "peer":
        ok = 1
        try:
            os.environ.setdefault("GSFM_PEER_TIMEOUT_S", "20")

            def allgather(b):
                out = [None] * world
                dist.all_gather_object(out, b)
                return out

            ctx.comm_init_peer(allgather, rank, world, 1 << 18)
            ctx.comm_peer_selftest()
        except Exception as e:  # noqa: BLE001 (any failure means: not this transport)
            print(f"[bench] rank {rank}: peer transport unavailable ({e}); RCCL instead", file=sys.stderr)
            ok = 0
        t = torch.tensor([ok])
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        if int(t.item()) == 1:
            ctx._transport = "peer mailboxes (one-shot all-reduce)"
        else:
            ctx.comm_destroy()
    if ctx._transport is None:
        uid = [_lib.comm_unique_id() if rank == 0 else None]
        dist.broadcast_object_list(uid, src=0)
        ctx.comm_init(uid[0], rank, world)
        ctx._transport = "RCCL"
    ctx._comm_ready = True


def run_extras(env, args, world, rank, main_line):
    """Side measurements of a single-GPU default run (world == 1): the other BASELINE.json configurations and the
    sweep kernels at sizes that do not fit the caches.  Every timing is the median of >= 3 warm repeats.  They run
    under a watchdog: whatever happens in there, the main JSON line is still printed."""
    import threading

    extra = {}
    sub = lambda **kw: argparse.Namespace(**{**vars(args), "steps": 3, "warmup": 1, "no_cpu_baseline": True, **kw})
    keys = ("metric", "value", "unit", "ms_per_step", "config", "roofline")

    def work():
        ctx = env["ctx"]
        jobs = (
            ("ra_c2", lambda: {k: v for k, v in bench_ra(**{**env, "args": sub(steps=10, warmup=3)}).items() if k in keys}),
            ("gp_c3", lambda: {k: v for k, v in bench_gp(**{**env, "args": sub()}).items() if k in keys}),
            ("ba_c4_shared_intrinsics", lambda: {k: v for k, v in bench_ba(**{**env, "args": sub(shared_intrinsics=True)}).items()
                                                 if k in keys}),
            # the same solve through a 12-parameter camera model: the 16-wide unit (bundle_adjustment.cc:136-139 dispatches any model)
            ("ba_c4_full_opencv", lambda: {k: v for k, v in bench_ba(**{**env, "args": sub(wide_model="full_opencv")}).items() if k in keys}),
            ("gp_c3_skewed_visibility", lambda: {k: v for k, v in bench_gp(**{**env, "args": sub(zipf=0.8)}).items() if k in keys}),
            # the same two solves on scenes with the locality of a walk-around capture (runs of consecutive cameras per
            # point, tracks in capture order): what the camera-major gathers cost when co-visible points share cache lines
            ("gp_c3_sequential_capture", lambda: {k: v for k, v in bench_gp(**{**env, "args": sub(capture="sequential")}).items() if k in keys}),
            ("ba_c4_sequential_capture", lambda: {k: v for k, v in bench_ba(**{**env, "args": sub(capture="sequential")}).items() if k in keys}),
            # small captures (800 / 400 cameras): where the reduced systems are assembled densely and factorised on the matrix cores
            # once a PCG solve runs past 100 iterations — the sizes of the reference's real-data configurations
            ("gp_small_sequential_capture", lambda: {k: v for k, v in bench_gp(**{**env, "args": sub(capture="sequential", scale=0.16)}).items() if k in keys}),
            ("ba_small_sequential_capture", lambda: {k: v for k, v in bench_ba(**{**env, "args": sub(capture="sequential", scale=0.04, shared_intrinsics=True)}).items()
                                                     if k in keys}),
            ("ra_large", lambda: bench_ra_large(ctx)),
            ("ra_c3", lambda: bench_ra_sized(ctx, 5000, 50)),
            ("ra_c4_non_ring_graphs", lambda: bench_ra_nonring(ctx)),
            ("track_filters_c3", lambda: bench_filters(ctx)),
            ("track_establishment_c3", lambda: bench_tracks(ctx, no_cpu=True)),
            ("chain_c4", lambda: bench_chain(ctx)),
        )
        for name, fn in jobs:
            try:
                extra[name] = fn()
            except Exception as e:  # report, never hide
                extra[name] = {"error": repr(e)}

    t = threading.Thread(target=work, daemon=True)
    t.start()
    t.join(timeout=float(os.environ.get("GSFM_BENCH_EXTRA_TIMEOUT", "300")))
    if t.is_alive():
        extra["watchdog"] = {"error": "extra measurements did not finish in time; main line printed without the rest"}
        if rank == 0:
            main_line["extra"] = extra
            emit_result(main_line)
        os._exit(0)
    return extra


def bench_chain(ctx, ncam=10_000, npts=1_000_000):
    """What GlobalMapper::Solve actually runs between its first rotation averaging and the end of its first bundle-adjustment
    round (global_mapper.cc:92-223), chained on ONE configs[3]-size scene, every stage started from the previous stage's result:
    RA -> GP (bearings oriented by RA's rotations, random start) -> FilterTracksByAngle / FilterTrackTriangulationAngle /
    FilterTracksByReprojection(10 x) -> NormalizeReconstruction -> BA positions-only -> BA with rotations (driver
    tests/chain_util.py, the one the parity tests use).  The headline step times the three estimators on three independent
    inputs; this is the same work with the processors inside the timed region and BA starting where GP ended.  Stage times
    are wall times of the calls through the C ABI with HOST arrays (one H2D / D2H per stage, the drop-in case); `solve_ms` is
    the library's own time of the solve proper.  Second of two runs (the first grows the workspaces)."""
    sys.path.insert(0, str(ROOT / "tests"))
    import chain_util
    from glomap_amd import synthetic

    sc = synthetic.make_chained_scene(ncam, npts, seed=0)
    stages = {}

    class Timed(chain_util.GpuBackend):
        pass

    def wrap(name):
        fn = getattr(chain_util.GpuBackend, name)

        def timed(self, *a, **k):
            ctx.synchronize()
            t0 = time.perf_counter()
            out = fn(self, *a, **k)
            ctx.synchronize()
            stages.setdefault(name, []).append(1e3 * (time.perf_counter() - t0))
            return out

        setattr(Timed, name, timed)

    for name in ("ra", "gp", "filter_angle", "filter_triangulation", "filter_reprojection", "normalize", "ba"):
        wrap(name)
    res = None
    for _ in range(2):
        stages.clear()
        t0 = time.perf_counter()
        res = chain_util.run_chain(sc, Timed(ctx))
        wall = 1e3 * (time.perf_counter() - t0)
    est = sum(sum(v) for k, v in stages.items())
    return {"cameras": ncam, "tracks": npts, "observations": int(sc.obs_cam.shape[0]),
            "ms_stage_calls": {k: [round(x, 2) for x in v] for k, v in stages.items()},
            "ms_estimators_and_processors": round(est, 1), "ms_wall_including_host_glue": round(wall, 1),
            "observations_kept": res["observations_kept"],
            "iterations": {"ra": res["rep_ra"], "gp_lm": res["rep_gp"]["iterations"], "gp_pcg": res["rep_gp"]["linear_iterations"],
                           "ba1_lm": res["rep_ba1"]["iterations"], "ba1_pcg": res["rep_ba1"]["linear_iterations"],
                           "ba2_lm": res["rep_ba2"]["iterations"], "ba2_pcg": res["rep_ba2"]["linear_iterations"]},
            "note": "host glue between the stages (numpy compaction of the filtered observations, bearing rotation) is in the wall time, "
                    "not in the stage sum; the device-resident variant of the BA outer loop is tests/test_pipeline_gpu.py::test_ba_outer_loop_device_resident"}


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


def profiled_step(ctx, kernel_id, step_fn, also=(), read=True):
    """One extra, event-instrumented step: every launch of the timed kernels is bracketed by a HIP
    event pair on the ctx stream (gsfm_ctx_profile_*).  Returns (launches, avg_ms) of `kernel_id`;
    the ids in `also` (and kernel_id itself with read=False) stay readable through kernel_line()."""
    ctx.profile_enable(True)
    for k in (kernel_id, *also):
        ctx.profile_read(k)
    step_fn()
    ctx.profile_enable(False)
    if not read:
        return None, None
    launches, total_ms = ctx.profile_read(kernel_id)
    return launches, (total_ms / launches if launches else None)


def pmc_traffic(kernel, field="bytes_per_launch"):
    """HBM bytes per launch from the rocprofv3 PMC passes committed under profiles/ (collected by
    tools/pmc_traffic.py; FETCH_SIZE doubled per the gfx950 note of MI355X_MICROARCH.md: right for the coalesced streams, an
    upper bound where record gathers are mixed in — `bytes_per_launch_raw` in the same file is the lower bound, see that
    tool's header).  None when no PMC pass exists for this kernel."""
    try:
        d = json.loads((ROOT / "profiles" / "pmc_traffic.json").read_text())
        for name, rec in d.items():  # template instances carry their arguments: "k_ba_phaseA<2>"
            if name == kernel or name.startswith(kernel + "<"):
                return rec.get(field)
        return None
    except Exception:
        return None


def roofline(kernel, bytes_per_launch, launches, avg_ms, note, others=None):
    achieved = bytes_per_launch / (avg_ms * 1e-3) / 1e9 if avg_ms else None
    r = {
        "bound": "hbm",
        "kernel": kernel,
        "achieved": achieved,
        "peak": HBM_PEAK_GBS,
        "unit": "GB/s",
        "frac": (achieved / HBM_PEAK_GBS) if achieved else None,
        "traffic": pmc_traffic(kernel.split(" ")[0]),
        "bytes_per_launch": bytes_per_launch,
        "avg_kernel_us": avg_ms * 1e3 if avg_ms else None,
        "launches_in_profiled_step": launches,
        "note": note,
    }
    if others:
        r["other_kernels"] = others
    return r


def kernel_line(ctx, kid, name, bytes_per_launch):
    launches, total_ms = ctx.profile_read(kid)
    avg_ms = total_ms / launches if launches else None
    gbs = bytes_per_launch / (avg_ms * 1e-3) / 1e9 if avg_ms else None
    line = {"kernel": name, "avg_kernel_us": avg_ms * 1e3 if avg_ms else None, "bytes_per_launch": bytes_per_launch,
            "achieved": gbs, "frac": gbs / HBM_PEAK_GBS if gbs else None, "launches": launches}
    # the same launch against the HBM bytes the committed PMC passes saw (profiles/pmc_traffic.json): a sweep whose gathers are
    # meant to hit the L2 moves fewer bytes than its algorithmic count, and `frac` alone would flatter it
    k = name.split(" ")[0]
    lo, hi = pmc_traffic(k, "bytes_per_launch_raw"), pmc_traffic(k)
    if avg_ms and lo and hi:
        line["traffic"] = hi
        line["traffic_raw"] = lo
        line["frac_of_hbm_on_pmc_bytes"] = [lo / (avg_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, hi / (avg_ms * 1e-3) / 1e9 / HBM_PEAK_GBS]
    return line


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
# configs[3]: the whole hot path on 10k cameras (headline)
# ----------------------------------------------------------------------------------------------
def _dev_ra(ctx, p):
    return type(p)(num_nodes=p.num_nodes, edge_i=ctx.to_device(p.edge_i), edge_j=ctx.to_device(p.edge_j),
                   edge_q=ctx.to_device(p.edge_q), edge_weight=ctx.to_device(p.edge_weight),
                   edge_ninl=ctx.to_device(p.edge_ninl), node_aa0=ctx.to_device(p.node_aa0), fixed_node=p.fixed_node)


def _dev_gp(ctx, p):
    from glomap_amd.flat import GpProblem

    return GpProblem(num_cams=p.num_cams, num_pts=p.num_pts, pt_offset=ctx.to_device(p.pt_offset.astype("int64")),
                     obs_cam=ctx.to_device(p.obs_cam), obs_dir=ctx.to_device(p.obs_dir),
                     obs_calibrated=ctx.to_device(p.obs_calibrated), cam_center=ctx.to_device(p.cam_center),
                     pt_xyz=ctx.to_device(p.pt_xyz))


def _dev_ba(ctx, p):
    from glomap_amd.flat import BaProblem

    return BaProblem(num_cams=p.num_cams, num_pts=p.num_pts, num_intr=p.num_intr,
                     pt_offset=ctx.to_device(p.pt_offset.astype("int64")), obs_cam=ctx.to_device(p.obs_cam),
                     obs_xy=ctx.to_device(p.obs_xy), cam_intr=ctx.to_device(p.cam_intr), cam_q=ctx.to_device(p.cam_q),
                     cam_t=ctx.to_device(p.cam_t), pt_xyz=ctx.to_device(p.pt_xyz), intr_model=ctx.to_device(p.intr_model),
                     intr_params=ctx.to_device(p.intr_params), fixed_cam=p.fixed_cam)


def bench_pipeline(args, ctx, rank, world, barrier, dist):
    """One step = gsfm_ra_solve + gsfm_gp_solve + gsfm_ba_solve on the configs[3] inputs."""
    import numpy as np

    from glomap_amd import _lib, estimators, sharding, so3, synthetic

    ncam = max(50, int(10_000 * args.scale))
    npts = max(500, int(1_000_000 * args.scale))
    succ = min(50, max(2, ncam // 4))
    p_ra = synthetic.make_ring_view_graph(ncam, succ, seed=0)
    p_gp = synthetic.make_gp_problem(ncam, npts, seed=0)
    p_ba = synthetic.make_ba_problem(ncam, npts, seed=0)
    E, M_gp, M_ba = p_ra.num_edges, p_gp.num_obs, p_ba.num_obs
    # strong scaling: the SAME problem, tracks split over the ranks; RA replicated on a communicator-free context
    g_loc, b_loc = p_gp, p_ba
    ctx_ra = ctx
    if world > 1:
        g_loc, _ = sharding.shard_gp_problem(p_gp, rank, world)
        b_loc, _ = sharding.shard_ba_problem(p_ba, rank, world)
        ctx_ra = _lib.Context(0 if os.environ.get("GSFM_BENCH_SINGLE_DEVICE") else int(os.environ.get("LOCAL_RANK", "0")))
        comm_init(ctx, dist, rank, world)
    d_ra, d_gp, d_ba = _dev_ra(ctx_ra, p_ra), _dev_gp(ctx, g_loc), _dev_ba(ctx, b_loc)
    rot = d_ra.node_aa0.clone()
    o_ra, o_gp, o_ba = (estimators.RotationEstimatorOptions(), estimators.GlobalPositionerOptions(),
                        estimators.BundleAdjusterOptions())
    rep, res, stage_ms = {}, {}, {"ra": [], "gp": [], "ba": [], "total": []}

    def step():
        t0 = time.perf_counter()
        rot.copy_from(d_ra.node_aa0)
        rc, _, rep["ra"] = estimators.ra_solve(d_ra, o_ra, ctx=ctx_ra, rot_inout=rot)
        if rc != 0:
            raise RuntimeError(f"gsfm_ra_solve failed: {rc}")
        t1 = time.perf_counter()
        rc, res["cen"], _, rep["gp"] = estimators.gp_solve(d_gp, o_gp, ctx=ctx)
        if rc != 0:
            raise RuntimeError(f"gsfm_gp_solve failed: {rc}")
        t2 = time.perf_counter()
        rc, res["q"], res["t"], _, _, rep["ba"] = estimators.ba_solve(d_ba, o_ba, ctx=ctx)
        if rc != 0:
            raise RuntimeError(f"gsfm_ba_solve failed: {rc}")
        t3 = time.perf_counter()
        for k, v in (("ra", t1 - t0), ("gp", t2 - t1), ("ba", t3 - t2), ("total", t3 - t0)):
            stage_ms[k].append(v * 1e3)

    if world > 1 and getattr(ctx, "_transport", "").startswith("peer"):
        # The peer-mailbox transport has only ever run with all ranks on ONE device (the harness has one GPU); the first time
        # it crosses xGMI is here.  One untimed step is its acceptance test: it must complete on every rank and leave the
        # replicated state bit-identical on all of them (that is what the transport guarantees by construction); otherwise
        # every rank detaches it and the run goes through RCCL.
        import torch

        ok, sums = 1, torch.zeros(2, dtype=torch.float64)
        try:
            step()
            sig = np.concatenate([res["cen"].numpy().ravel(), res["q"].numpy().ravel(), res["t"].numpy().ravel()])
            ok = int(np.isfinite(sig).all())
            sums = torch.tensor([float(sig.sum()), float(np.abs(sig).sum())], dtype=torch.float64)
        except Exception as e:  # noqa: BLE001
            print(f"[bench] rank {rank}: sharded step failed on the peer transport ({e}); RCCL instead", file=sys.stderr)
            ok = 0
        lo, hi = sums.clone(), sums.clone()  # (every rank takes part in every rendezvous collective, failed step or not)
        dist.all_reduce(lo, op=dist.ReduceOp.MIN)
        dist.all_reduce(hi, op=dist.ReduceOp.MAX)
        ok = int(ok and bool((lo == hi).all()))
        t = torch.tensor([ok])
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        for v in stage_ms.values():
            v.clear()
        if int(t.item()) != 1:
            ctx.comm_destroy()
            ctx._comm_ready = False
            os.environ["GSFM_BENCH_TRANSPORT"] = "rccl"
            comm_init(ctx, dist, rank, world)
    dt = timed_steps(step, args.steps, args.warmup, barrier, dist)
    timed = {k: v[args.warmup:] for k, v in stage_ms.items()}  # the warm-up steps are not part of the statistics
    med = {k: float(np.median(v)) for k, v in timed.items()}
    value = M_ba * args.steps / dt
    # ---- roofline: one extra, event-instrumented step; the four sweep kernels of the PCG iterations, the one with the
    # most time per step in front
    ctx.stats(reset=True)  # the solver-path counters of ONE step (config.solver_paths)
    profiled_step(ctx, KERNEL_BA, step, also=(KERNEL_BA_B, KERNEL_GP, KERNEL_GP_B, KERNEL_GP_WSUM), read=False)
    paths = ctx.stats()
    Mg, Pg, Mb, Pb = g_loc.num_obs, g_loc.num_pts, b_loc.num_obs, b_loc.num_pts
    F = 2  # free intrinsics columns stored per observation (SIMPLE_RADIAL: f, k)
    lines = [
        kernel_line(ctx, KERNEL_BA, "k_ba_phaseA (BA implicit Schur product, track-major half over the stored Jacobian planes)",
                    ba_phaseA_bytes(Mb, Pb, ncam, F)),
        kernel_line(ctx, KERNEL_BA_B, "k_ba_phaseB (BA, camera-major half)", ba_phaseB_bytes(Mb, ncam, p_ba.num_intr)),
        kernel_line(ctx, KERNEL_GP, "k_gp_phaseA (GP, track-major half)", gp_phaseA_bytes(Mg, Pg, ncam)),
        kernel_line(ctx, KERNEL_GP_B, gp_phaseB_name(ctx), gp_phaseB_bytes(Mg, ncam)),
    ]
    for o in lines:
        o["ms_per_step"] = (o["avg_kernel_us"] or 0.0) * (o["launches"] or 0) * 1e-3
    # Second bound of the three sweeps whose lanes each fetch one 64-byte record from a table (tools/exp_gather_calib.hip,
    # profiles/r04_gather_calibration_times.txt): a CU's vector memory path sustains a fixed number of outstanding line
    # requests, which caps record gathers at ~103 G requests/s chip-wide when the table sits in the L2 (64 KB ... 1 MB windows:
    # 106.7 / 101.5 G/s) and ~54 G/s when every gather misses it — whatever the record size.  A sweep of M observations cannot
    # run faster than M / that rate, however few bytes it moves; `frac` (of HBM) alone hides that.
    for o, req, peak, why in ((lines[2], Mg, 103.0, "(c | z) records, 640 KB table: L2-resident"),
                              (lines[3], Mg, 103.0 if "phaseB_x" in lines[3]["kernel"] else 54.6,
                               "point records, chunk windows <= 2.75 MB per XCD" if "phaseB_x" in lines[3]["kernel"] else "point records, no locality: L2 misses"),
                              (lines[1], Mb, 54.6, "point records, no locality: L2 misses")):
        if o["avg_kernel_us"]:
            g = req / (o["avg_kernel_us"] * 1e-6) / 1e9
            o["gather_bound"] = {"record_gathers_per_launch": req, "achieved_G_per_s": g, "calibrated_peak_G_per_s": peak,
                                 "frac": g / peak, "table": why, "source": "profiles/r04_gather_calibration_times.txt"}
    # the camera side of a GP iteration in the chunked order is two launches: the sweep and the per-camera sum of its pieces
    wl, wms = ctx.profile_read(KERNEL_GP_WSUM)
    if wl:
        gb = lines[3]
        gb["k_gp_wsum_avg_us"] = wms / wl * 1e3
        both_us = (gb["avg_kernel_us"] or 0.0) + gb["k_gp_wsum_avg_us"]
        gb["with_k_gp_wsum"] = {"avg_us": both_us, "frac": gb["bytes_per_launch"] / (both_us * 1e-6) / 1e9 / HBM_PEAK_GBS,
                                "ms_per_step": both_us * (gb["launches"] or 0) * 1e-3}
    lines.sort(key=lambda o: -o["ms_per_step"])
    top = lines[0]
    roof = roofline(
        top["kernel"], top["bytes_per_launch"], top["launches"], (top["avg_kernel_us"] or 0.0) * 1e-3 or None,
        "the sweep kernel with the most time per step (launches x avg = %.0f ms of the %.0f ms step); averages are over the "
        "launches that ran the sweep (HIP events on the library's stream; the one launch per solve that finds it converged and "
        "returns is not counted); one BA PCG iteration = k_ba_phaseA + k_ba_phaseB + k_ba_phaseI + k_cg_update, one GP PCG "
        "iteration = k_gp_phaseA + k_gp_phaseB[_x + k_gp_wsum] + k_cg_update" % (top["ms_per_step"], med["total"]),
        others=lines[1:])
    roof["ms_per_step"] = top["ms_per_step"]
    for k in ("gather_bound", "traffic_raw", "frac_of_hbm_on_pmc_bytes"):
        if k in top:
            roof[k] = top[k]
    err_ra = synthetic.rotation_errors_deg(so3.aa_to_rotmat(rot.numpy()), p_ra.gt_R)
    err_gp = synthetic.center_errors_after_sim3(res["cen"].numpy(), p_gp.gt_center)
    err_ba = synthetic.rotation_errors_deg(so3.quat_to_rotmat(res["q"].numpy()), so3.quat_to_rotmat(p_ba.gt_q))
    cpu = None
    if not args.no_cpu_baseline and rank == 0 and world == 1:
        cpu = cpu_baseline_pipeline(p_ra, p_gp, p_ba, rep)
        cpu["gpu_speedup_same_inputs"] = cpu["seconds"]["total"] * 1e3 / med["total"] if cpu.get("seconds") else None
        if cpu.get("exact_solves"):
            cpu["gpu_speedup_vs_exact_solves"] = cpu["exact_solves"]["seconds"]["total"] * 1e3 / med["total"]
    config = {
        "workload": "configs[3]: synthetic 10k cameras — full hot path RA (ring view graph, %d relative-pose edges) + GP (%d tracks / %d "
        "observations, random start) + BA (%d tracks / %d observations, one SIMPLE_RADIAL camera per image, start = GT + "
        "noise), reference default options; inputs per SURVEY.md section 8d" % (E, npts, M_gp, npts, M_ba),
        "cameras": ncam, "edges": E, "tracks": npts, "observations_gp": M_gp, "observations_ba": M_ba,
        "parallelism": "single GPU" if world == 1 else f"strong scaling: GP/BA track-shard x{world} (one all-reduce per PCG iteration), RA replicated",
        "rccl_ranks": world, "transport": getattr(ctx, "_transport", None),
        "ms_per_stage_median": med,
        "ms_per_step_each": timed["total"],
        "iterations": {"ra_l1": rep["ra"]["iterations_l1"], "ra_irls": rep["ra"]["iterations_irls"],
                       "ra_pcg": rep["ra"]["linear_iterations"], "gp_lm": rep["gp"]["iterations"],
                       "gp_pcg": rep["gp"]["linear_iterations"], "ba_lm": rep["ba"]["iterations"],
                       "ba_lm_accepted": rep["ba"]["successful_steps"], "ba_pcg": rep["ba"]["linear_iterations"]},
        # which linear-solver paths one step took (gsfm_ctx_stats): reduced solves, how many were deflated / ran with recycled
        # Ritz vectors in the preconditioner, Ritz vectors harvested, chunked sweeps
        "solver_paths": paths,
        "final_cost": {"gp": rep["gp"]["final_cost"], "ba": rep["ba"]["final_cost"]},
        "vs_ground_truth": {"ra_median_rot_err_deg": float(np.median(err_ra)),
                            "gp_median_center_err_rel": float(np.median(err_gp)),  # relative to the GT extent (helper divides ONCE)
                            "ba_median_rot_err_deg": float(np.median(err_ba))},
    }
    line = base_line("track-obs/sec through RA+GP+BA (configs[3] hot path)", value, "obs/s", world, args, dt, config, roof, cpu, ctx)
    line["scaling"] = "strong"  # the fixed configs[3] problem, tracks split over the ranks
    line["submetrics"] = {  # BASELINE.json's metric, part by part (medians of the timed steps)
        "view_graph_edges_per_s_RA+GP": E / ((med["ra"] + med["gp"]) * 1e-3),
        "view_graph_edges_per_s_RA": E / (med["ra"] * 1e-3),
        "track_obs_per_s_per_GP_iter": M_gp * max(1, rep["gp"]["iterations"]) / (med["gp"] * 1e-3),
        "track_obs_per_s_per_BA_iter": M_ba * max(1, rep["ba"]["iterations"]) / (med["ba"] * 1e-3),
    }
    return line


# algorithmic bytes per launch of the four sweep kernels
def gp_phaseA_bytes(M, P, N):
    return 24.0 * M + 96.0 * P + 48.0 * N


def gp_phaseB_bytes(M, N):
    # per observation: point index 4 + (a, beta) 16 + the (X_p, t_p) record 48.  The chunked order (k_gp_phaseB_x) streams a
    # camera index per observation as well and writes 24-byte piece partials; those extra bytes are NOT counted as useful.
    return 68.0 * M + 96.0 * N


def gp_phaseB_name(ctx):
    """Which camera-side sweep the library ran (it picks the chunked order when the point records overflow an XCD's L2)."""
    if ctx.stats().get("pcg_chunked_sweeps", 0) > 0:
        return "k_gp_phaseB_x (GP, camera-side half in the chunked, XCD-partitioned order; + k_gp_wsum, not in this time)"
    return "k_gp_phaseB (GP, camera-major half)"


def ba_phaseA_bytes(M, P, N, F):
    return (16.0 * (9 + F) + 12.0) * M + 96.0 * P + 48.0 * N


def ba_phaseB_bytes(M, N, K):
    return 60.0 * M + 304.0 * N + 64.0 * K


def cpu_baseline_pipeline(p_ra, p_gp, p_ba, gpu_rep):
    """The multithreaded C++ restatement (oracle/cpu.py: same LM decisions, exact block elimination; RA with direct
    skyline-Cholesky solves) on the SAME three inputs, on all host cores of this box — restated CPU oracle, NOT Ceres /
    CHOLMOD (the reference cannot be built here).  Two legs:
      like-for-like (`value`): the reduced solves of GP / BA exactly as libgsfm runs them — PCG to 1e-10 (GP) / 1e-6 (BA)
                               with the gauge modes deflated — i.e. the same linear-solver work on the CPU;
      exact_solves           : PCG to 1e-14 without deflation, what "SPARSE_SCHUR is exact" means for the parity tests
                               (the oracle configuration tests/test_fullsize_gpu.py compares against)."""
    from oracle import cpu

    cpu.load_native()  # -march=native build of the restatement, made on this box (oracle/Makefile `native`)

    out = {"unit": "obs/s", "cores": cpu.num_threads(), "host_hw_threads": os.cpu_count(), "kind": "port", "build": cpu.BUILD_FLAGS,
           "cores_note": "cores = the CPUs this process may use (scheduler affinity capped by the cgroup CPU quota of the box)"}

    def leg(gp_tol, ba_tol, deflate):
        t0 = time.perf_counter()
        rr = {}
        cpu.ra_estimate_rotations(p_ra.num_nodes, p_ra.edge_i, p_ra.edge_j, p_ra.edge_q, p_ra.edge_weight, p_ra.edge_ninl,
                                  p_ra.node_aa0, p_ra.fixed_node, report=rr)
        t1 = time.perf_counter()
        _, _, _, sg = cpu.gp_solve(p_gp.num_cams, p_gp.pt_offset, p_gp.obs_cam, p_gp.obs_dir, p_gp.obs_calibrated,
                                   p_gp.cam_center, p_gp.pt_xyz, pcg_tol=gp_tol, deflate=deflate)
        t2 = time.perf_counter()
        rb = cpu.ba_solve(p_ba.num_cams, p_ba.pt_offset, p_ba.obs_cam, p_ba.obs_xy, p_ba.cam_intr, p_ba.intr_model, p_ba.fixed_cam,
                          p_ba.cam_q, p_ba.cam_t, p_ba.pt_xyz, p_ba.intr_params, pcg_tol=ba_tol, deflate=deflate)
        t3 = time.perf_counter()
        return {"seconds": {"ra": t1 - t0, "gp": t2 - t1, "ba": t3 - t2, "total": t3 - t0},
                "value": p_ba.num_obs / (t3 - t0),
                "pcg_relative_tolerance": {"gp": gp_tol, "ba": ba_tol}, "deflated": bool(deflate),
                "iterations": {"ra_l1": rr.get("l1_iterations"), "ra_irls": rr.get("irls_iterations"), "gp_lm": sg.iterations,
                               "gp_pcg": sg.linear_iterations, "ba_lm": rb[5].iterations, "ba_pcg": rb[5].linear_iterations}}

    like = leg(1e-10, 1e-6, 1)
    out.update(like)
    out["reference_code_ra"] = reference_code_ra(p_ra)
    # the second leg doubles the CPU time of the run: on a slow box it is left out so that the default run stays within minutes
    if os.environ.get("GSFM_BENCH_NO_EXACT_CPU_LEG"):
        out["exact_solves_skipped"] = "GSFM_BENCH_NO_EXACT_CPU_LEG"
    elif like["seconds"]["total"] > 90.0:
        out["exact_solves_skipped"] = "the like-for-like leg took %.0f s on this box" % like["seconds"]["total"]
    else:
        out["exact_solves"] = leg(1e-14, 1e-14, 0)
    out["sample"] = ("ONE pass of the same configs[3] inputs the GPU line is timed on (RA %d edges + GP %d obs + BA %d obs), "
                     "restated C++/OpenMP CPU oracle on %d threads (RA factorisation single-threaded) with the GPU's linear-solver "
                     "settings (PCG 1e-10 / 1e-6, gauge modes deflated, Ceres line search in GP) — not Ceres; `exact_solves` = the same pass with PCG to 1e-14; "
                     "compiler flags: %s"
                     % (p_ra.num_edges, p_gp.num_obs, p_ba.num_obs, out["cores"], cpu.BUILD_FLAGS))
    return out


# ----------------------------------------------------------------------------------------------
# rotation averaging, configs[1]
# ----------------------------------------------------------------------------------------------
def bench_ra(args, ctx, rank, world, barrier, dist):
    import numpy as np

    from glomap_amd import estimators, so3, synthetic

    # configs[1] is a 3.6 MB problem: one PCG iteration (~10 us) is shorter than one xGMI all-reduce, so
    # sharding it can only slow it down.  With --gpus N every rank solves its own view graph of this
    # size (replicas only, no collective) — the sharded RA path exists for large
    # graphs and is exercised by tests/test_multirank_gpu.py.
    N, succ = 1000, 50
    p = synthetic.make_ring_view_graph(N, succ, seed=rank)
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
    value = world * E * args.steps / dt
    launches, avg_ms = profiled_step(ctx, KERNEL_RA_GJ, step, also=(KERNEL_RA,))
    T = (N + 31) // 32
    gj_flops = (T - 1) * (T - 1) * 2 * 2.0 * 32**3 + 2 * (T - 1) * 2.0 * 32**3  # tile products of one step
    achieved = gj_flops / (avg_ms * 1e-3) / 1e12 if avg_ms else None
    roof = {
        "bound": "mfma",
        "kernel": "k_dense_gj_step (one block Gauss-Jordan step of the dense Laplacian inverse, v_mfma_f64_16x16x4_f64)",
        "achieved": achieved,
        "peak": F64_MFMA_PEAK_TFLOPS,
        "unit": "TFLOP/s",
        "frac": achieved / F64_MFMA_PEAK_TFLOPS if achieved else None,
        "traffic": pmc_traffic("k_dense_gj_step"),
        "flops_per_launch": gj_flops,
        "avg_kernel_us": avg_ms * 1e3 if avg_ms else None,
        "launches_in_profiled_step": launches,
        "note": "configs[1] is a 3.6 MB problem (8 MB dense Laplacian): every kernel of it is latency-bound — the step "
        "time is set by the serial 32x32 pivot inversion of ONE workgroup, not by the tile products; "
        "extra.ra_large reports the HBM-bound RA sweep kernels on a graph that does not fit the caches",
    }
    err = synthetic.rotation_errors_deg(so3.aa_to_rotmat(rot.numpy()), p.gt_R)
    cpu = None if (args.no_cpu_baseline or rank != 0) else cpu_baseline_ra(p)
    config = {
        "workload": "configs[1]: synthetic ring view graph, 1k cameras / 50k relative-pose edges, full rotation "
        "averaging (MST init + L1-ADMM + IRLS, reference defaults)",
        "cameras": N,
        "edges": E,
        "parallelism": "single GPU" if world == 1 else f"replicas x{world} (no collective)",
        "seconds_setup_last": last.get("seconds_total", 0.0) - last.get("seconds_solve", 0.0),
        "l1_iterations": last.get("iterations_l1"),
        "irls_iterations": last.get("iterations_irls"),
        "pcg_iterations_per_step": last.get("linear_iterations"),
        "median_rot_err_deg_vs_gt": float(np.median(err)),
    }
    return base_line("view-graph edges/sec (RA)", value, "edges/s", world, args, dt, config, roof, cpu, ctx)


def bench_filters(ctx):
    """Track filters on the observation lists of configs[2] (5k cameras / 500k tracks /
    ~3M observations), arrays resident in HBM: FilterTracksByAngle + FilterTracksByReprojection sweeps."""
    import numpy as np

    from glomap_amd import processors as pr
    from glomap_amd import so3, synthetic

    p = synthetic.make_gp_problem(5000, 500_000, seed=0)
    q = so3.rotmat_to_quat(p.cam_R)
    t = -np.einsum("nij,nj->ni", p.cam_R, p.gt_center)
    undist = np.einsum("mij,mj->mi", p.cam_R[p.obs_cam], p.obs_dir)
    dv = pr.SceneView(p.num_cams, ctx.to_device(p.pt_offset), ctx.to_device(p.obs_cam), ctx.to_device(q), ctx.to_device(t),
                      ctx.to_device(p.gt_xyz), obs_undist=ctx.to_device(undist))
    M = p.num_obs
    out = {}
    for name, fn in (("FilterTracksByAngle", lambda: pr.TrackFilter.FilterTracksByAngle(dv, 1.0, ctx=ctx)),
                     ("FilterTracksByReprojection", lambda: pr.TrackFilter.FilterTracksByReprojection(dv, 1e-2, True, ctx=ctx))):
        fn()
        ctx.profile_enable(True)
        ctx.profile_read(KERNEL_FILTER)
        ctx.synchronize()
        t0 = time.perf_counter()
        reps = 10
        for _ in range(reps):
            keep, changed = fn()
        ctx.synchronize()
        dt = (time.perf_counter() - t0) / reps
        ctx.profile_enable(False)
        n, ms = ctx.profile_read(KERNEL_FILTER)
        kbytes = 33.0 * M + 32.0 * p.num_pts  # ray 24 + cam 4 + obs_pt 4 + keep 1 per observation; X_p 24 + off 8 per track
        out[name] = {"observations": M, "tracks_changed": int(changed), "call_ms": dt * 1e3, "value": M / dt, "unit": "obs/s",
                     "k_filter_obs_avg_us": ms / n * 1e3 if n else None,
                     "k_filter_obs_GBps": kbytes / (ms / n * 1e-3) / 1e9 if n else None,
                     "frac_of_hbm_peak": kbytes / (ms / n * 1e-3) / 1e9 / HBM_PEAK_GBS if n else None}
    return out


def bench_tracks(ctx, no_cpu=False):
    """Track establishment + selection on a match graph of the size of configs[2]
    (5k images, 500k ground-truth tracks -> ~5.7M inlier matches over ~250k image pairs, 6M features), inputs
    resident in HBM; results stay in HBM (counts only come back)."""
    import numpy as np

    from glomap_amd import synthetic
    from glomap_amd.tracks import MatchGraph, TrackEngine, TrackEstablishmentOptions

    g = synthetic.make_match_graph(5000, 500_000, seed=0)
    NM, F = len(g["match_feat1"]), int(g["feat_offset"][-1])
    eng = TrackEngine(MatchGraph.from_dict(g).to_device(ctx), ctx=ctx)
    reg = ctx.to_device(np.ones(5000, np.uint8))
    eng.EstablishFullTracks(fetch=False)
    ctx.profile_enable(True)
    ctx.profile_read(KERNEL_TRACK_HOOK)
    reps = 10
    ctx.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        nt = eng.EstablishFullTracks(fetch=False)
    ctx.synchronize()
    t_est = (time.perf_counter() - t0) / reps
    ctx.profile_enable(False)
    n, ms = ctx.profile_read(KERNEL_TRACK_HOOK)
    out = {"images": 5000, "features": F, "image_pairs": len(g["pair_image1"]), "inlier_matches": NM,
           "tracks": int(nt), "tracks_discarded": int(eng.num_discarded),
           "establish_ms": t_est * 1e3, "value": NM / t_est, "unit": "matches/s"}
    if n:
        # 8 B per match streamed + the two parent entries it must at least read (4 B each)
        kbytes = 16.0 * NM
        out["k_uf_hook"] = {"avg_kernel_us": ms / n * 1e3, "bytes_per_launch": kbytes, "achieved_GBps": kbytes / (ms / n * 1e-3) / 1e9,
                            "frac_of_hbm_peak": kbytes / (ms / n * 1e-3) / 1e9 / HBM_PEAK_GBS,
                            "note": "bound by the device's returning-atomic (CAS) rate, not by HBM: the same sweep with the CAS "
                                    "replaced by a plain store takes 108 us (DESIGN.md section 4.7)"}
    for name, opts in (("select_default", TrackEstablishmentOptions()),
                       ("select_cap_200_per_view", TrackEstablishmentOptions(min_num_tracks_per_view=200))):
        eng.options = opts
        eng.FindTracksForProblem(reg, fetch=False)
        ctx.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            ns = eng.FindTracksForProblem(reg, fetch=False)
        ctx.synchronize()
        out[name] = {"ms": (time.perf_counter() - t0) / reps * 1e3, "tracks_selected": int(ns)}
    if not no_cpu:
        from oracle import tracks as ot

        a = (g["pair_image1"], g["pair_image2"], g["pair_valid"], g["pair_offset"], g["match_feat1"], g["match_feat2"],
             g["feat_offset"], g["feat_xy"])
        t0 = time.perf_counter()
        ref = ot.establish_full_tracks(*a)
        t1 = time.perf_counter()
        ot.find_tracks_for_problem(*ref[:4], np.ones(5000, bool))
        t2 = time.perf_counter()
        out["cpu_baseline"] = {"establish_ms": (t1 - t0) * 1e3, "select_ms": (t2 - t1) * 1e3, "cores": 1, "kind": "port",
                               "sample": "the same match graph, vectorised numpy / scipy.sparse.csgraph restatement (oracle/tracks.py)",
                               "tracks": int(len(ref[0]))}
    return out


def bench_ra_sized(ctx, N, succ):
    """One full RA solve (reference defaults) of a ring view graph with N cameras, inputs resident in HBM."""
    import numpy as np

    from glomap_amd import estimators, so3, synthetic

    p = synthetic.make_ring_view_graph(N, succ, seed=0)
    pd = type(p)(num_nodes=p.num_nodes, edge_i=ctx.to_device(p.edge_i), edge_j=ctx.to_device(p.edge_j),
                 edge_q=ctx.to_device(p.edge_q), edge_weight=ctx.to_device(p.edge_weight),
                 edge_ninl=ctx.to_device(p.edge_ninl), node_aa0=ctx.to_device(p.node_aa0), fixed_node=0)
    rot = pd.node_aa0.clone()
    times, rep = [], None
    for i in range(4):  # one warm-up + three timed solves, median
        rot.copy_from(pd.node_aa0)
        ctx.synchronize()
        t0 = time.perf_counter()
        rc, _, rep = estimators.ra_solve(pd, estimators.RotationEstimatorOptions(), ctx=ctx, rot_inout=rot)
        ctx.synchronize()
        if rc != 0:
            raise RuntimeError(f"gsfm_ra_solve failed: {rc}")
        if i:
            times.append(time.perf_counter() - t0)
    best = float(np.median(times))
    err = synthetic.rotation_errors_deg(so3.aa_to_rotmat(rot.numpy()), p.gt_R)
    return {"cameras": N, "edges": p.num_edges, "ms_per_solve": best * 1e3, "value": p.num_edges / best, "unit": "edges/s",
            "linear_solver": "dense direct (f64 MFMA)" if N <= 2048 else ("PCG, dense diagonal-block preconditioner (f64 MFMA block inverses)" if N <= 32768 else "3-RHS Jacobi-PCG"),
            "l1_iterations": rep["iterations_l1"], "irls_iterations": rep["iterations_irls"],
            "pcg_iterations": rep["linear_iterations"], "median_rot_err_deg_vs_gt": float(np.median(err))}


def bench_ra_nonring(ctx, N=10_000, degree=100):
    """Full RA solves on view graphs that are not banded rings, at the camera count of configs[3]: k-nearest-neighbour
    graph, the same plus 8 hub images linked to a quarter of all images, and a thin ring with random long-range loop
    closures — node ids shuffled.  Reports time and the PCG iteration counts of the block-preconditioned path."""
    import numpy as np

    from glomap_amd import estimators, so3, synthetic

    out = {}
    for kind in ("geometric", "hub", "chords"):
        p = synthetic.make_view_graph(kind, N, degree, seed=0)
        pd = _dev_ra(ctx, p)
        rot = pd.node_aa0.clone()
        times, rep = [], None
        for i in range(3):  # one warm-up + two timed solves
            rot.copy_from(pd.node_aa0)
            ctx.synchronize()
            t0 = time.perf_counter()
            rc, _, rep = estimators.ra_solve(pd, estimators.RotationEstimatorOptions(), ctx=ctx, rot_inout=rot)
            ctx.synchronize()
            if rc != 0:
                raise RuntimeError(f"gsfm_ra_solve failed on the {kind} graph: {rc}")
            if i:
                times.append(time.perf_counter() - t0)
        err = synthetic.rotation_errors_deg(so3.aa_to_rotmat(rot.numpy()), p.gt_R)
        deg = np.bincount(np.concatenate([p.edge_i, p.edge_j]), minlength=N)
        out[kind] = {"cameras": N, "edges": p.num_edges, "max_degree": int(deg.max()), "ms_per_solve": float(np.median(times)) * 1e3,
                     "l1_iterations": rep["iterations_l1"], "irls_iterations": rep["iterations_irls"],
                     "pcg_iterations": rep["linear_iterations"], "median_rot_err_deg_vs_gt": float(np.median(err))}
    return out


def bench_ra_large(ctx):
    """The RA sweep kernels on a view graph that does not fit the caches (200k cameras / 10M edges):
    per-edge residual + IRLS weight sweep and the weighted-Laplacian SpMV with 3 right-hand sides."""
    import numpy as np

    from glomap_amd import estimators, synthetic

    N, succ = 200_000, 50  # E = 10^7 (a scaled-up run for a true HBM number)
    p = synthetic.make_ring_view_graph(N, succ, seed=0)
    E = p.num_edges
    w = np.ones(E)
    x = np.random.default_rng(0).normal(size=(N, 3))
    y, ms = estimators.ra_laplacian_apply(p, w, x, repeat=20, ctx=ctx)
    spmv_bytes = 24.0 * E + 60.0 * N  # 2E incidences x (4 B nbr + 8 B w) + node vectors / diag / rowptr
    spmv_min = 16.0 * E + 48.0 * N  # SURVEY.md section 8d's compulsory traffic: every edge ONCE
    res_ms = estimators.ra_residuals_timed(p, np.zeros((N, 3)), repeat=20, ctx=ctx)
    res_bytes = 72.0 * E + 24.0 * N  # SURVEY.md section 8d K-RA-res: q 32 + idx 8 + residual 24 + weight 8 per edge
    out = {
        "workload": f"ring view graph, {N} cameras / {E} edges (does not fit L2 / Infinity Cache residency of C2)",
        "k_edge_residual": {"avg_kernel_us": res_ms * 1e3, "bytes_per_launch": res_bytes,
                            "achieved_GBps": res_bytes / (res_ms * 1e-3) / 1e9,
                            "frac_of_hbm_peak": res_bytes / (res_ms * 1e-3) / 1e9 / HBM_PEAK_GBS},
        "k_spmv": {"avg_kernel_us": ms * 1e3, "bytes_per_launch_layout": spmv_bytes, "bytes_per_launch": spmv_min,
                   "achieved_GBps": spmv_min / (ms * 1e-3) / 1e9, "frac_of_hbm_peak": spmv_min / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                   "note": "priced against the compulsory 16E + 48N bytes; the CSR-by-node layout actually streams 24E + 60N "
                           "(every edge from both endpoints)"},
    }
    return out


def reference_code_ra(p):
    """The REFERENCE'S OWN rotation averaging on the same view graph: glomap/estimators/global_rotation_averaging.cc (+
    rotation_initializer.cc, tree.cc, rigid3d.cc) compiled unmodified into oracle/_ref/libref_glomap_ra.so (built in the build
    container, travels with the snapshot) — on stand-ins for Eigen / CHOLMOD (an envelope Cholesky after reverse Cuthill-McKee) /
    COLMAP's LAD solver / Boost, one thread.  Reported next to the restated oracle's time: reference CODE, not the reference
    BUILD (no SuiteSparse, no Eigen vectorisation)."""
    try:
        import numpy as np

        from glomap_amd import so3, synthetic
        from oracle import ref

        if ref.load_ra() is None:
            return {"skipped": "oracle/_ref/libref_glomap_ra.so not present"}
        N = p.num_nodes
        t0 = time.perf_counter()
        r = ref.ra_estimate([0], np.zeros(N), np.arange(N), np.zeros(N), p.edge_i, p.edge_j, p.edge_q, pair_weight=p.edge_weight,
                            pair_ninl=synthetic.break_inlier_ties_by_index(p.edge_ninl), frame_q=so3.aa_to_quat(p.node_aa0))
        dt = time.perf_counter() - t0
        return {"seconds": dt, "value": p.num_edges / dt, "unit": "edges/s", "cores": 1, "kind": "reference", "ok": bool(r["ok"]),
                "iterations": {"l1": r["l1_iterations"], "irls": r["irls_iterations"], "admm": r["admm_iterations"]},
                "sample": "ONE RotationEstimator::EstimateRotations of the reference's own source on the pipeline's view graph, stand-in "
                          "linear algebra (envelope Cholesky for CHOLMOD), single thread"}
    except Exception as e:  # noqa: BLE001 — a checker-side problem must not take the bench line down
        return {"skipped": f"{type(e).__name__}: {e}"}


def cpu_baseline_ra(p):
    """Restated C++ CPU oracle (direct skyline-Cholesky solves, per-edge sweeps on all host cores — NOT CHOLMOD) on the
    same view graph."""
    from oracle import cpu

    cpu.load_native()  # -march=native build of the restatement, made on this box (oracle/Makefile `native`)

    t0 = time.perf_counter()
    n = 0
    while True:
        cpu.ra_estimate_rotations(p.num_nodes, p.edge_i, p.edge_j, p.edge_q, p.edge_weight, p.edge_ninl, p.node_aa0, p.fixed_node)
        n += 1
        dt = time.perf_counter() - t0
        if dt > 10.0 or n >= 8:
            break
    return {"value": p.num_edges * n / dt, "unit": "edges/s", "cores": cpu.num_threads(), "host_hw_threads": os.cpu_count(),
            "kind": "port", "build": cpu.BUILD_FLAGS,
            "sample": f"{n} full RA solves of the same view graph (restated C++/OpenMP CPU oracle, single-threaded skyline "
                      "Cholesky; not Ceres/CHOLMOD)"}


# ----------------------------------------------------------------------------------------------
# global positioning, configs[2]
# ----------------------------------------------------------------------------------------------
def shard_tracks(pt_offset, rank, world):
    from glomap_amd import sharding

    return sharding.shard_tracks(pt_offset, rank, world)


def bench_gp(args, ctx, rank, world, barrier, dist):
    import numpy as np

    from glomap_amd import estimators, synthetic
    from glomap_amd.flat import GpProblem

    ncam = int(5000 * args.scale)
    npts_rank = int(500_000 * args.scale)  # weak scaling: tracks per GPU fixed
    npts = npts_rank * world
    zipf = float(getattr(args, "zipf", 0.0))  # > 0: Zipf-distributed per-camera observation counts (skewed visibility)
    capture = getattr(args, "capture", "random")
    if world == 1:
        p = synthetic.make_gp_problem(ncam, npts, seed=0, zipf=zipf, capture=capture)
    else:  # every rank generates only its own shard (cameras identical everywhere)
        p = synthetic.make_gp_problem(ncam, npts_rank, seed=0, shard=(rank, world), zipf=zipf, capture=capture)
    lo, hi = 0, p.num_pts
    o0, o1 = 0, p.num_obs
    M_total = p.num_obs
    if dist is not None:
        import torch

        tm = torch.tensor([p.num_obs], dtype=torch.int64)
        dist.all_reduce(tm)
        M_total = int(tm.item())
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
    launches, avg_ms = profiled_step(ctx, KERNEL_GP, step, also=(KERNEL_GP_B,))
    M_loc, P_loc = o1 - o0, hi - lo
    roof = roofline(
        "k_gp_phaseA (implicit Schur product, track-major half: t_p = Hpp^-1 sum_k Q_k z_c)",
        24.0 * M_loc + 96.0 * P_loc + 48.0 * ncam,
        launches,
        avg_ms,
        "one PCG iteration = k_gp_phaseA + k_gp_phaseB + k_cg_update",
        others=[kernel_line(ctx, KERNEL_GP_B, gp_phaseB_name(ctx), 68.0 * M_loc + 96.0 * ncam)],
    )
    err = synthetic.center_errors_after_sim3(res["cen"].numpy(), p.gt_center)
    cpu = None if (args.no_cpu_baseline or rank != 0 or world > 1) else cpu_baseline_gp(p)
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
    if zipf:
        per_cam = np.bincount(p.obs_cam, minlength=ncam)
        config["skewed_visibility"] = {"zipf_exponent": zipf, "obs_per_camera_median": int(np.median(per_cam)),
                                       "obs_per_camera_max": int(per_cam.max())}
    return base_line("track-obs/sec per LM iteration (GP)", value, "obs/s", world, args, dt, config, roof, cpu, ctx)


def cpu_baseline_gp(p):
    from oracle import cpu

    cpu.load_native()  # -march=native build of the restatement, made on this box (oracle/Makefile `native`)

    t0 = time.perf_counter()
    ok, c, X, s = cpu.gp_solve(p.num_cams, p.pt_offset, p.obs_cam, p.obs_dir, p.obs_calibrated, p.cam_center, p.pt_xyz)
    dt = time.perf_counter() - t0
    return {"value": p.num_obs * max(1, s.iterations) / dt, "unit": "obs/s", "cores": cpu.num_threads(),
            "host_hw_threads": os.cpu_count(), "kind": "port", "seconds": dt, "build": cpu.BUILD_FLAGS,
            "sample": f"one GP solve of the SAME input ({p.num_obs} observations, {s.iterations} LM iterations; restated "
                      "C++/OpenMP CPU oracle with exact elimination, not Ceres)"}


# ----------------------------------------------------------------------------------------------
# bundle adjustment, configs[3] on one GPU (and track-sharded over N GPUs)
# ----------------------------------------------------------------------------------------------
def bench_ba(args, ctx, rank, world, barrier, dist):
    import numpy as np

    from glomap_amd import estimators, so3, synthetic
    from glomap_amd.flat import BaProblem

    ncam = int(10_000 * args.scale)
    npts_rank = int(1_000_000 * args.scale)  # weak scaling: tracks per GPU fixed
    npts = npts_rank * world
    shared = bool(getattr(args, "shared_intrinsics", False))  # configs[3] has both variants
    wide = getattr(args, "wide_model", None)  # a camera model with more than 8 parameters: the 16-wide unit (csrc/ba_wide.hip)
    if world == 1 and wide:
        # 100 physical cameras shared round-robin (one 12-parameter camera per IMAGE is not a well-posed problem: DESIGN.md 3.1)
        p = synthetic.make_ba_problem_wide(ncam, npts, wide, seed=0, shared_intrinsics=shared, num_intr_groups=max(1, ncam // 100))
    elif world == 1:
        p = synthetic.make_ba_problem(ncam, npts, seed=0, shared_intrinsics=shared, capture=getattr(args, "capture", "random"))
    else:  # every rank generates only its own shard (cameras / intrinsics / start identical everywhere)
        p = synthetic.make_ba_problem(ncam, npts_rank, seed=0, shared_intrinsics=shared, shard=(rank, world),
                                      capture=getattr(args, "capture", "random"))
    lo, hi = 0, p.num_pts
    o0, o1 = 0, p.num_obs
    M_total = p.num_obs
    if dist is not None:
        import torch

        tm = torch.tensor([p.num_obs], dtype=torch.int64)
        dist.all_reduce(tm)
        M_total = int(tm.item())
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
    launches, avg_ms = profiled_step(ctx, KERNEL_BA, step, also=(KERNEL_BA_B,))
    M_loc, P_loc = o1 - o0, hi - lo
    F = 10 if wide else 2  # free intrinsics columns stored per observation (SIMPLE_RADIAL: f, k; FULL_OPENCV: all but cx, cy)
    roof = roofline(
        "k_ba_phaseA (implicit Schur product, track-major half over the stored Jacobian planes)",
        (16.0 * (9 + F) + 12.0) * M_loc + 96.0 * P_loc + 48.0 * ncam,
        launches,
        avg_ms,
        "one PCG iteration = k_ba_phaseA + k_ba_phaseB + k_ba_phaseI + k_cg_update",
        others=[kernel_line(ctx, KERNEL_BA_B, "k_ba_phaseB (camera-major half, Jacobians recomputed, 64-byte point-record gathers)",
                            60.0 * M_loc + 304.0 * ncam + 64.0 * p.num_intr)],
    )
    # drop-in case: the same solve with every array handed over in HOST memory (H2D at entry, D2H at exit)
    host_ms = None
    if world == 1 and not shared:
        t0 = time.perf_counter()
        rc, *_ = estimators.ba_solve(p, opt, ctx=ctx)
        host_ms = (time.perf_counter() - t0) * 1e3 if rc == 0 else None
    R = so3.quat_to_rotmat(res["q"].numpy())
    Rg = so3.quat_to_rotmat(p.gt_q)
    rot_err = synthetic.rotation_errors_deg(R, Rg)
    cpu = None if (args.no_cpu_baseline or rank != 0 or world > 1) else cpu_baseline_ba(p)
    config = {
        "workload": "configs[3] on one GPU per rank: synthetic 10k cameras / 1M tracks / ~5M observations per GPU, "
        f"bundle adjustment ({'FULL_OPENCV (12 parameters, 16-wide intrinsics blocks)' if wide else 'SIMPLE_RADIAL'}, "
        f"{'ONE camera shared by all images' if shared else ('%d cameras shared round-robin' % p.num_intr if wide else 'one camera per image')}, Huber 1 px, "
        "reference defaults), start = GT + noise",
        "intrinsics_blocks": int(p.num_intr),
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
        "ms_per_solve_host_arrays_pcie_inclusive": host_ms,
    }
    return base_line("track-obs/sec per BA iteration", value, "obs/s", world, args, dt, config, roof, cpu, ctx)


def cpu_baseline_ba(p):
    from oracle import cpu

    cpu.load_native()  # -march=native build of the restatement, made on this box (oracle/Makefile `native`)

    t0 = time.perf_counter()
    r = cpu.ba_solve(p.num_cams, p.pt_offset, p.obs_cam, p.obs_xy, p.cam_intr, p.intr_model, p.fixed_cam, p.cam_q, p.cam_t,
                     p.pt_xyz, p.intr_params)
    dt = time.perf_counter() - t0
    return {"value": p.num_obs * max(1, r[5].iterations) / dt, "unit": "obs/s", "cores": cpu.num_threads(),
            "host_hw_threads": os.cpu_count(), "kind": "port", "seconds": dt, "build": cpu.BUILD_FLAGS,
            "sample": f"one BA solve of the SAME input ({p.num_obs} observations, {r[5].iterations} LM iterations; restated "
                      "C++/OpenMP CPU oracle with exact elimination, not Ceres)"}


if __name__ == "__main__":
    main()
