#!/usr/bin/env python
"""Dump the per-kernel summary (the `--stats` view) of a rocprofv3 rocpd SQLite database as CSV.

rocprofv3 (ROCm 7.2) writes `<name>_results.db`; its `top_kernels` view is the kernel-trace
summary.  Usage: python tools/rocpd_stats.py gpurun_out/x/prof/ra_results.db > profiles/x.csv
Durations are in nanoseconds in the database; the CSV reports microseconds.
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
    print("kernel,calls,total_us,avg_us,min_us,max_us,pct,grid_x,wg_x,vgpr,agpr,sgpr,lds,scratch")
    for r in rows:
        name = r[0].replace("gsfm::(anonymous namespace)::", "").replace(",", ";")
        if len(name) > 90:
            name = name[:87] + "..."
        print(f'"{name}",{r[1]},{r[2]/1e3:.1f},{r[3]/1e3:.3f},{r[4]/1e3:.3f},{r[5]/1e3:.3f},{100*r[2]/total:.2f},'
              f"{r[6]},{r[7]},{r[8]},{r[9]},{r[10]},{r[11]},{r[12]}")


if __name__ == "__main__":
    main(sys.argv[1])
