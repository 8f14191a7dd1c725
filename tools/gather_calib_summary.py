#!/usr/bin/env python
"""Per-kernel FETCH_SIZE / WRITE_SIZE of tools/exp_gather_calib (rocprofv3 --pmc pass) -> raw counter bytes per gather.

    python tools/gather_calib_summary.py out/calib_FETCH_SIZE_results.db [out/calib_WRITE_SIZE_results.db]
"""
import sqlite3
import sys

NG = 6_000_000


def per_kernel(db, counter):
    c = sqlite3.connect(db)
    vals = {}
    for name, v in c.execute("select kernel_name, value from counters_collection where counter_name = ?", (counter,)):
        vals.setdefault(name.split("(")[0].replace("void ", "").strip(), []).append(v)
    return {k: sum(v) / len(v) for k, v in vals.items()}


def main():
    f = per_kernel(sys.argv[1], "FETCH_SIZE")
    w = per_kernel(sys.argv[2], "WRITE_SIZE") if len(sys.argv) > 2 else {}
    print("kernel,fetch_KiB_raw,fetch_bytes_raw_per_gather,write_bytes_raw_per_gather")
    for k in sorted(f):
        print(f'"{k}",{f[k]:.0f},{f[k] * 1024 / NG:.1f},{w.get(k, 0.0) * 1024 / NG:.1f}')


if __name__ == "__main__":
    main()
