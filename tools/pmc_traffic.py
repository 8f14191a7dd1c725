#!/usr/bin/env python
"""HBM traffic per kernel launch from rocprofv3 PMC passes (separate --pmc FETCH_SIZE and --pmc
WRITE_SIZE runs, as /opt/skills/guides/MI355X_MICROARCH.md prescribes: the two do not fit one pass).

    rocprofv3 --pmc FETCH_SIZE --kernel-trace -d out -o <name>_FETCH_SIZE -- python bench.py ...
    rocprofv3 --pmc WRITE_SIZE --kernel-trace -d out -o <name>_WRITE_SIZE -- python bench.py ...
    python tools/pmc_traffic.py out/<name>  > profiles/<round>_<name>_pmc.csv   (and merges pmc_traffic.json)

Units and correction: both counters are in KiB.  On gfx950 FETCH_SIZE tallies the 128-byte requests
of wide (16 B/lane) coalesced streams at 64 B, i.e. reports exactly half of their bytes (guide, HBM
section).  A record GATHER is different — calibrated in round 4 on known byte counts in this access
pattern (tools/exp_gather_calib.hip, profiles/r04_gather_calibration_pmc.csv): every missing gather
is ONE request tallied at 64 B whatever the record size (32 / 64 / 128 bytes), so for the 64-byte
point records of the camera-major sweeps the RAW value is already the truth and doubling it counts
every record twice.  The read side is therefore reported both ways:
  bytes_per_launch      = 2 x raw + writes: right for stream kernels (k_ba_phaseA, k_gp_phaseA's tile
                          streams), an UPPER bound for gather kernels — what bench.py quotes as `traffic`;
  bytes_per_launch_raw  = raw + writes: the LOWER bound, and within a few per cent of the truth for the
                          gather-dominated sweeps (k_ba_phaseB: 5.0 M records x 64 B = 320 MB of its
                          327 MB raw; round 3's "2.16 x its algorithmic bytes" was the doubling, the
                          kernel moves ~1.1 - 1.2 x).
Infinity-Cache hits are counted by these counters, so this is fabric traffic, not DRAM-only traffic.
"""
import json
import sqlite3
import sys
from pathlib import Path


def per_kernel(db, counter):
    """kernel -> (launches, avg, min, max, working launches, avg over the working launches).  A launch of a PCG kernel
    that finds the solve converged returns at once and moves (almost) nothing; "working" = at least a quarter of the
    kernel's median counter value (same rule as tools/rocpd_stats.py applies to durations)."""
    c = sqlite3.connect(db)
    vals = {}
    for name, v in c.execute("select kernel_name, value from counters_collection where counter_name = ?", (counter,)):
        vals.setdefault(name, []).append(v)
    out = {}
    for name, v in vals.items():
        v.sort()
        med = v[len(v) // 2]
        work = [x for x in v if x >= 0.25 * med] or v
        out[name] = (len(v), sum(v) / len(v), v[0], v[-1], len(work), sum(work) / len(work))
    return out


def short(name):
    name = name.replace("gsfm::(anonymous namespace)::", "").replace("gsfm::", "")
    return name.split("(")[0].replace("void ", "").strip()


def main(prefix):
    fetch = per_kernel(f"{prefix}_FETCH_SIZE_results.db", "FETCH_SIZE")
    write = per_kernel(f"{prefix}_WRITE_SIZE_results.db", "WRITE_SIZE")
    out = {}
    print("kernel,launches,working_launches,fetch_KiB_raw_avg,fetch_bytes_x2_avg,write_bytes_avg,bytes_per_launch,bytes_per_launch_raw")
    for k in sorted(fetch, key=lambda k: -fetch[k][0] * fetch[k][1]):
        n, nwork, favg = fetch[k][0], fetch[k][4], fetch[k][5]  # averages over the working launches
        wavg = write.get(k, (0, 0.0, 0, 0, 0, 0.0))[5]
        fb2 = 2.0 * favg * 1024.0
        wb = wavg * 1024.0
        name = short(k)
        print(f'"{name}",{n},{nwork},{favg:.1f},{fb2:.0f},{wb:.0f},{fb2 + wb:.0f},{0.5 * fb2 + wb:.0f}')
        out[name] = {"launches": n, "working_launches": nwork, "fetch_bytes_raw": favg * 1024.0, "fetch_bytes_x2": fb2,
                     "write_bytes": wb, "bytes_per_launch": fb2 + wb, "bytes_per_launch_raw": 0.5 * fb2 + wb}
    dst = Path(__file__).resolve().parent.parent / "profiles" / "pmc_traffic.json"
    merged = json.loads(dst.read_text()) if dst.exists() else {}
    merged.update(out)
    dst.write_text(json.dumps(merged, indent=1, sort_keys=True))


if __name__ == "__main__":
    main(sys.argv[1])
