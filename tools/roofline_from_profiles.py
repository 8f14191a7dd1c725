"""Recomputes the roofline fractions of the sweep kernels from the committed evidence alone:

  algorithmic bytes per launch (the formulas of DESIGN.md section 4.3 = bench.py's *_bytes functions, at the configs[3]
  sizes recorded in profiles/<tag>_bench_default.json)  /  average duration of the WORKING launches
  (profiles/<tag>_pipeline_c4_kernel_stats.csv, rocprofv3 --kernel-trace; `work_avg_us` of tools/rocpd_stats.py: launches
  that only found the solve converged and returned are left out)  /  8 TB/s,   next to the PMC traffic per working launch
  of profiles/<tag>_pipeline_c4_pmc.csv.

Usage: python tools/roofline_from_profiles.py r03 [> profiles/r03_rooflines.md]"""
import csv
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402  (only its byte formulas)

PEAK = 8000.0  # GB/s, /opt/skills/guides/MI355X_MICROARCH.md


def main(tag):
    line = json.load(open(os.path.join(ROOT, "profiles", f"{tag}_bench_default.json")))
    cfg = line["config"]
    N, Mg, Mb = cfg["cameras"], cfg["observations_gp"], cfg["observations_ba"]
    P = cfg["tracks"]
    K, F = N, 2  # one SIMPLE_RADIAL camera per image: free columns f, k
    formulas = {
        "k_ba_phaseA": ("(16 (9+F) + 12) M + 96 P + 48 N", bench.ba_phaseA_bytes(Mb, P, N, F)),
        "k_ba_phaseB": ("60 M + 304 N + 64 K", bench.ba_phaseB_bytes(Mb, N, K)),
        "k_gp_phaseA": ("24 M + 96 P + 48 N", bench.gp_phaseA_bytes(Mg, P, N)),
        "k_gp_phaseB": ("68 M + 96 N", bench.gp_phaseB_bytes(Mg, N)),
    }
    dur = {}
    with open(os.path.join(ROOT, "profiles", f"{tag}_pipeline_c4_kernel_stats.csv")) as f:
        for row in csv.DictReader(f):
            for k in formulas:
                if k in row["kernel"] and k not in dur:
                    dur[k] = (float(row["work_avg_us"]), int(row["work_calls"]), int(row["calls"]), float(row["avg_us"]), int(row["vgpr"]))
    traffic = {}
    with open(os.path.join(ROOT, "profiles", f"{tag}_pipeline_c4_pmc.csv")) as f:
        for row in csv.reader(f):
            for k in formulas:
                if row and row[0].startswith(k) and k not in traffic:
                    hi = float(row[6])                       # 2 x raw FETCH_SIZE + writes
                    lo = 0.5 * float(row[4]) + float(row[5])  # raw FETCH_SIZE + writes (tools/pmc_traffic.py: gathers)
                    traffic[k] = (hi, lo)
    print("| kernel | algorithmic bytes / launch | formula | avg us, working launches (rocprofv3) | working / all launches | avg us, all | VGPRs | GB/s | frac of 8 TB/s | PMC traffic / working launch: raw ... 2 x raw FETCH_SIZE (+ writes) | traffic / algorithmic |")
    print("|---|---|---|---|---|---|---|---|---|---|---|")
    for k, (form, nbytes) in formulas.items():
        us, wcalls, calls, us_all, vgpr = dur[k]
        gbps = nbytes / us / 1e3
        tr = traffic.get(k)
        print(f"| `{k}` | {nbytes / 1e6:.1f} MB | `{form}` | {us:.1f} | {wcalls} / {calls} | {us_all:.1f} | {vgpr} | {gbps:.0f} | {gbps / PEAK:.3f} | "
              f"{'' if tr is None else f'{tr[1] / 1e6:.0f} ... {tr[0] / 1e6:.0f} MB'} | {'' if tr is None else f'{tr[1] / nbytes:.2f} ... {tr[0] / nbytes:.2f}'} |")
    print("\nTraffic: the upper value doubles the raw FETCH_SIZE (right for coalesced 16 B/lane streams: `k_ba_phaseA`), the lower "
          "one takes it as it is (right for record gathers, one 64-byte request each: `k_ba_phaseB`, the old `k_gp_phaseB`); the sweeps "
          "that mix both lie in between (tools/pmc_traffic.py, tools/exp_gather_calib.hip).")
    r = line["roofline"]
    print(f"\nbench.py's own line (HIP events over the working launches, other box): `{r['kernel'].split(' ')[0]}` {r['avg_kernel_us']:.1f} us -> "
          f"frac {r['frac']:.3f}; step {line['ms_per_step']:.1f} ms, value {line['value'] / 1e6:.2f} M obs/s.")
    for o in r.get("other_kernels", []):
        print(f"  `{o['kernel'].split(' ')[0]}` {o['avg_kernel_us']:.1f} us -> frac {o['frac']:.3f} ({o['launches']} launches)")


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "r03")
