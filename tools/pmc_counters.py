#!/usr/bin/env python
"""Per-kernel averages of arbitrary PMC counters from the rocpd databases of tools/profile_counters.sh:

    python tools/pmc_counters.py gpurun_out/<tag>  > <tag>_counters.csv

One row per kernel: launches, then the per-launch average of every counter found in the directory's *_results.db files
(each database holds the counters of one --pmc pass), plus the stall split the guide defines
(WAIT_ANY + WAIT_INST_ANY + ACTIVE_INST_ANY ~ WAVE_CYCLES, /opt/skills/guides/MI355X_MICROARCH.md) and the L2 hit rate."""
import glob
import sqlite3
import sys


def short(name):
    name = name.replace("gsfm::(anonymous namespace)::", "").replace("gsfm::", "")
    return name.split("(")[0].replace("void ", "").strip()


def main(directory):
    data, launches, counters = {}, {}, []
    for db in sorted(glob.glob(f"{directory}/*_results.db")):
        c = sqlite3.connect(db)
        try:
            rows = c.execute("select kernel_name, counter_name, count(*), avg(value) from counters_collection "
                             "group by kernel_name, counter_name").fetchall()
        except sqlite3.Error:
            continue
        for kname, cname, n, avg in rows:
            k = short(kname)
            data.setdefault(k, {})[cname] = avg
            launches[k] = n
            if cname not in counters:
                counters.append(cname)
    derived = ["wait_any_frac", "wait_inst_frac", "active_inst_frac", "l2_hit_rate"]
    print(",".join(["kernel", "launches"] + counters + derived))
    order = sorted(data, key=lambda k: -data[k].get("SQ_WAVE_CYCLES", 0.0) * launches[k])
    for k in order:
        d = data[k]
        wc = d.get("SQ_WAVE_CYCLES", 0.0)
        fr = [d.get(n, 0.0) / wc if wc else float("nan") for n in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY")]
        hit, miss = d.get("TCC_HIT_sum", 0.0), d.get("TCC_MISS_sum", 0.0)
        hr = hit / (hit + miss) if hit + miss else float("nan")
        print(",".join([f'"{k}"', str(launches[k])] + [f"{d.get(c, float('nan')):.1f}" for c in counters] +
                       [f"{v:.3f}" for v in fr + [hr]]))


if __name__ == "__main__":
    main(sys.argv[1])
