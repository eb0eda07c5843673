#!/usr/bin/env python3
"""tools/summarize_profile.py <gpurun_out/prof_TAG> <profiles/rN/NAME>

Turns the raw output of tools/profile.sh (rocprofv3 --kernel-trace --stats, and one --pmc pass per
counter group) into the committed summaries:
  <NAME>_kernel_stats.csv   the rocprofv3 kernel statistics as they came
  <NAME>_pmc.json           per kernel: mean counter value per dispatch, dispatch count
and refreshes profiles/latest_pmc.json (the HBM bytes per traceq_kernel launch that bench.py reports
as roofline.traffic: (2 * FETCH_SIZE + WRITE_SIZE) KB, MI355X_MICROARCH.md "HBM" correction)."""
import collections
import csv
import glob
import json
import os
import shutil
import sys


def short(name):
    n = name.split("ezd::", 1)[1] if "ezd::" in name else name
    return n.split("(")[0]


def main():
    src, dst = sys.argv[1], sys.argv[2]
    os.makedirs(os.path.dirname(dst), exist_ok=True)
    stats = glob.glob(os.path.join(src, "stats", "*kernel_stats.csv"))
    if stats:
        shutil.copy(stats[0], dst + "_kernel_stats.csv")
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in sorted(glob.glob(os.path.join(src, "pmc*", "*counter_collection.csv"))):
        for r in csv.DictReader(open(f)):
            agg[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    out = {
        "workload": "bench.py default (C2), rocprofv3 --pmc passes of tools/profile.sh, one counter group per run",
        "units": "FETCH_SIZE / WRITE_SIZE in KB per dispatch; SQ_* as reported (quad-cycles for *_CYCLES)",
        "hbm_correction": "bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024: MI355X_MICROARCH.md says gfx950 FETCH_SIZE tallies "
                          "128-B requests at 64 B (calibrated there for wide streaming reads; divergent 16-B gathers are "
                          "uncalibrated, so this is an upper bound of the read side)",
        "kernels": {k: {c: {"mean_per_dispatch": sum(v) / len(v), "dispatches": len(v)} for c, v in cs.items()}
                    for k, cs in agg.items() if k.startswith(("traceq", "shade", "raygen", "accumulate", "trace_kernel"))},
    }
    json.dump(out, open(dst + "_pmc.json", "w"), indent=1)
    tq = [k for k in out["kernels"] if k.startswith("traceq_kernel<false")]
    if tq and "FETCH_SIZE" in out["kernels"][tq[0]] and "WRITE_SIZE" in out["kernels"][tq[0]]:
        fk = out["kernels"][tq[0]]["FETCH_SIZE"]["mean_per_dispatch"]
        wk = out["kernels"][tq[0]]["WRITE_SIZE"]["mean_per_dispatch"]
        latest = {
            "traceq_hbm_bytes_per_launch": int((2 * fk + wk) * 1024),
            "source": dst + "_pmc.json",
            "fetch_kb": fk,
            "write_kb": wk,
            "note": "mean over all traceq_kernel dispatches of a step (5 stage launches + 5 normally-empty redo launches)",
        }
        json.dump(latest, open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(dst))), "latest_pmc.json"), "w"), indent=1)
        print("traceq HBM bytes/launch", latest["traceq_hbm_bytes_per_launch"])
    for k, cs in out["kernels"].items():
        print(k, {c: round(v["mean_per_dispatch"]) for c, v in list(cs.items())[:4]})


if __name__ == "__main__":
    main()
