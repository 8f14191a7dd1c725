#!/usr/bin/env python
"""Dump the per-kernel summary (the `--stats` view) of a rocprofv3 rocpd SQLite database as CSV.

rocprofv3 (ROCm 7.2) writes `<name>_results.db`; its `top_kernels` view is the kernel-trace
summary.  Usage: python tools/rocpd_stats.py gpurun_out/x/prof/ra_results.db > profiles/x.csv
Durations are in nanoseconds in the database; the CSV reports microseconds.

`work_calls` / `work_avg_us`: the launches that did the kernel's work.  The PCG kernels are enqueued ahead of the device;
the launch that finds the solve converged (and the ones behind it until the host has read the status back) return at
once after a few microseconds.  Those are real launches and stay in `calls` / `avg_us`, but a roofline figure is about the
sweeps, so the averages are also given over the launches that last at least a quarter of the kernel's median duration.
"""
import sqlite3
import sys


def main(path):
    c = sqlite3.connect(path)
    rows = c.execute(
        "select name, count(*), sum(duration), avg(duration), min(duration), max(duration), "
        "max(grid_x), max(workgroup_x), max(vgpr_count), max(accum_vgpr_count), max(sgpr_count), max(lds_size), "
        "max(scratch_size) from kernels group by name order by sum(duration) desc"
    ).fetchall()
    total = sum(r[2] for r in rows) or 1
    print("kernel,calls,total_us,avg_us,min_us,max_us,pct,grid_x,wg_x,vgpr,agpr,sgpr,lds,scratch,work_calls,work_avg_us")
    for r in rows:
        durs = sorted(d[0] for d in c.execute("select duration from kernels where name = ?", (r[0],)))
        med = durs[len(durs) // 2]
        work = [d for d in durs if d >= 0.25 * med]
        name = r[0].replace("gsfm::(anonymous namespace)::", "").replace(",", ";")
        if len(name) > 90:
            name = name[:87] + "..."
        print(f'"{name}",{r[1]},{r[2]/1e3:.1f},{r[3]/1e3:.3f},{r[4]/1e3:.3f},{r[5]/1e3:.3f},{100*r[2]/total:.2f},'
              f"{r[6]},{r[7]},{r[8]},{r[9]},{r[10]},{r[11]},{r[12]},{len(work)},{sum(work)/len(work)/1e3:.3f}")


if __name__ == "__main__":
    main(sys.argv[1])
