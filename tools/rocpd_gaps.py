#!/usr/bin/env python
"""Idle time between consecutive kernels of a rocprofv3 kernel trace (rocpd SQLite database): for every pair of kernels that
follow each other on the device, gap = start(next) - end(previous).  Printed: the busy time, the sum of the short gaps
(< 50 us: dispatch latency between dependent kernels of one stream) and of the long ones (host round trips: status
read-backs, set-up), and the short gaps by the kernel that FOLLOWS them.
Usage: python tools/rocpd_gaps.py <name>_results.db"""
import collections
import sqlite3
import sys


def main(path):
    c = sqlite3.connect(path)
    rows = c.execute("select name, start, end from kernels order by start").fetchall()
    busy = sum(e - s for _, s, e in rows)
    short = collections.Counter()
    nshort = collections.Counter()
    s_short = s_long = 0
    n_long = 0
    for (n0, s0, e0), (n1, s1, e1) in zip(rows, rows[1:]):
        g = s1 - e0
        if g <= 0:
            continue
        if g < 50_000:
            s_short += g
            key = n1.replace("gsfm::(anonymous namespace)::", "").split("(")[0][:50]
            short[key] += g
            nshort[key] += 1
        else:
            s_long += g
            n_long += 1
    print(f"kernels {len(rows)}  busy {busy/1e6:.1f} ms  short gaps {s_short/1e6:.1f} ms  long gaps {s_long/1e6:.1f} ms ({n_long})")
    for k, v in short.most_common(12):
        print(f"  before {k:52s} {v/1e6:7.2f} ms in {nshort[k]:6d} gaps ({v/nshort[k]/1e3:.2f} us each)")


if __name__ == "__main__":
    main(sys.argv[1])
