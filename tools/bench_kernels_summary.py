#!/usr/bin/env python
"""One line per bench.py JSON file: step time, per-stage medians and the four sweep kernels' average launch times."""
import json
import sys

for f in sys.argv[1:]:
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e:  # noqa: BLE001
        print(f, "ERR", e)
        continue
    r = d["roofline"]
    ks = [r] + r.get("other_kernels", [])
    kern = "  ".join("%s %.1fus(%.2f)" % (k["kernel"].split(" ")[0], k["avg_kernel_us"] or 0, k["frac"] or 0) for k in ks)
    c = d["config"]
    med = c.get("ms_per_stage_median", {})
    print("%-40s %.1f ms/step  %s  it=%s\n    %s" % (f.split("/")[-1], d["ms_per_step"], {k: round(v, 1) for k, v in med.items()},
                                                     c.get("iterations"), kern))
