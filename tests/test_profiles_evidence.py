"""The committed evidence of the current round is self-consistent (no GPU needed): the roofline object of the default bench
line follows from its own fields, the kernel it names is in the committed rocprofv3 kernel summary with an average duration
that agrees with the HIP-event figure (profiler overhead and box-to-box spread: 15 %), and the PMC file has a traffic
figure for it."""
import csv
import json
import pathlib

ROOT = pathlib.Path(__file__).resolve().parent.parent
# the newest round that has committed its default bench line AND the rocprofv3 summaries that go with it
TAG = max(p.name[:3] for p in (ROOT / "profiles").glob("r??_bench_default.json")
          if (ROOT / "profiles" / f"{p.name[:3]}_pipeline_c4_kernel_stats.csv").exists()
          and (ROOT / "profiles" / f"{p.name[:3]}_pipeline_c4_pmc.csv").exists())


def test_bench_line_roofline_follows_from_its_fields():
    line = json.loads((ROOT / "profiles" / f"{TAG}_bench_default.json").read_text())
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "dtype", "data",
                "config", "roofline", "cpu_baseline"):
        assert key in line, key
    assert line["n_gpus"] == 1 and line["dtype"] == "f64" and line["vs_baseline"] is None and "workload" in line["config"]
    r = line["roofline"]
    achieved = r["bytes_per_launch"] / (r["avg_kernel_us"] * 1e-6) / 1e9
    assert abs(achieved - r["achieved"]) <= 1e-6 * achieved and r["peak"] == 8000.0 and r["unit"] == "GB/s"
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12
    assert r["ms_per_step"] <= line["ms_per_step"]  # the dominant kernel fits inside the step
    # value = observations of the BA problem per second of a step
    assert abs(line["value"] - line["config"]["observations_ba"] / (line["ms_per_step"] * 1e-3)) <= 1e-6 * line["value"]
    cb = line["cpu_baseline"]
    assert cb["kind"] == "port" and cb["cores"] >= 1 and "march=native" in cb["build"] and cb["value"] > 0


def test_named_kernel_is_in_the_committed_trace_and_pmc_files():
    line = json.loads((ROOT / "profiles" / f"{TAG}_bench_default.json").read_text())
    kernels = [line["roofline"]] + line["roofline"]["other_kernels"]
    with open(ROOT / "profiles" / f"{TAG}_pipeline_c4_kernel_stats.csv") as f:
        rows = list(csv.DictReader(f))
    pmc = (ROOT / "profiles" / f"{TAG}_pipeline_c4_pmc.csv").read_text()
    for k in kernels:
        name = k["kernel"].split(" ")[0]
        hit = [r for r in rows if name in r["kernel"]]
        assert hit, name
        trace_us = float(hit[0]["work_avg_us"])
        assert abs(trace_us - k["avg_kernel_us"]) <= 0.15 * k["avg_kernel_us"], (name, trace_us, k["avg_kernel_us"])
        assert name in pmc, name
