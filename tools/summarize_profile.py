#!/usr/bin/env python3
"""tools/summarize_profile.py <gpurun_out/prof_TAG> <profiles/rN/NAME>

Turns the raw output of tools/profile.sh (rocprofv3 --kernel-trace --stats, and one --pmc pass per counter group) into
the committed summaries:
  <NAME>_kernel_stats.csv   the rocprofv3 kernel statistics as they came
  <NAME>_pmc.json           per kernel: mean counter value per dispatch, dispatch count
  pmc_summary.json (next to them)   what bench.py quotes: per-STEP sums of the dominant kernel's counters (all
                            traceq4_kernel instances; a step = one render call = 1 + max_bounce of its launches), HBM bytes
                            = (2 * FETCH_SIZE + WRITE_SIZE) KB as MI355X_MICROARCH.md "HBM" prescribes for gfx950, L2 bytes
                            = TCP_TCC_READ_REQ * 64 B (L1-miss read requests), LDS bytes <= SQ_INSTS_LDS * 64 lanes * 16 B,
                            and `source_sha` = the hash of the GPU sources the profile was taken from."""
import collections
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ezrt_amd.srchash import gpu_source_hash  # noqa: E402


def short(name):
    n = name.split("ezd::", 1)[1] if "ezd::" in name else name
    return n.split("(")[0]


def all_kernels_per_step(kernels):
    """kernels: {short name: {counter: {"mean_per_dispatch", "dispatches"}}} (the *_pmc.json form) -> VALU wave-instructions of ONE timed step
    summed over every kernel of the timed pipeline (the launches of the one instrumented step -- FULLCTR variants, raygen_kernel,
    inner_rel_kernel, the binary traceq_kernel<true,..> -- are left out); steps profiled = dispatches of the primary trace kernel."""
    steps = 0
    for k, v in kernels.items():
        if k.startswith("traceq4_kernel") and "SQ_INSTS_VALU" in v and ", true, false" in k.split("<", 1)[1][:14]:
            steps = max(steps, v["SQ_INSTS_VALU"]["dispatches"])
    if not steps:
        return None
    per, total = {}, 0.0
    for k, v in kernels.items():
        c = v.get("SQ_INSTS_VALU")
        if not c:
            continue
        args = k.split("<", 1)[1] if "<" in k else ""
        instrumented = k.startswith(("raygen", "inner_rel", "inner4_rel")) or (k.startswith("traceq_kernel") and args.startswith("true")) or \
            (k.startswith("shade") and ", true," in args)
        if instrumented:
            continue
        x = c["mean_per_dispatch"] * c["dispatches"] / steps
        if x > 0:
            per[k] = int(x)
            total += x
    return {"valu_wave_instr_per_step": int(total), "steps_profiled": steps, "per_kernel": per,
            "note": "sum over every kernel of a timed step (dispatch counts / render calls profiled); the instrumented step's kernels are left out"}


def main():
    if len(sys.argv) > 2 and sys.argv[1] == "--add-all-kernels":   # derive the field from an already collected <dir>/final_pmc.json
        d = sys.argv[2]
        pm = json.load(open(os.path.join(d, "final_pmc.json")))
        sm = json.load(open(os.path.join(d, "pmc_summary.json")))
        sm["all_kernels"] = all_kernels_per_step(pm["kernels"])
        json.dump(sm, open(os.path.join(d, "pmc_summary.json"), "w"), indent=1)
        print(sm["all_kernels"]["valu_wave_instr_per_step"])
        return
    src, dst = sys.argv[1], sys.argv[2]
    launches_per_step = int(sys.argv[3]) if len(sys.argv) > 3 else 5
    os.makedirs(os.path.dirname(dst), exist_ok=True)
    stats = glob.glob(os.path.join(src, "stats", "*kernel_stats.csv"))
    if stats:
        shutil.copy(stats[0], dst + "_kernel_stats.csv")
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in sorted(glob.glob(os.path.join(src, "pmc*", "*counter_collection.csv"))):
        for r in csv.DictReader(open(f)):
            agg[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    keep = ("traceq", "shade", "raygen", "accumulate", "trace_kernel", "tail_kernel", "inner4_rel", "inner_rel")
    out = {
        "workload": "bench.py default (C2) --steps 2 --warmup 1 --extras 0, rocprofv3 --pmc passes of tools/profile.sh, one counter group per run",
        "units": "FETCH_SIZE / WRITE_SIZE in KB per dispatch; SQ_* as reported",
        "hbm_correction": "bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024: MI355X_MICROARCH.md says gfx950 FETCH_SIZE tallies "
                          "128-B requests at 64 B (calibrated there for wide streaming reads; divergent 16-B gathers are "
                          "uncalibrated, so this is an upper bound of the read side)",
        "source_sha": gpu_source_hash(),
        "kernels": {k: {c: {"mean_per_dispatch": sum(v) / len(v), "dispatches": len(v)} for c, v in cs.items()}
                    for k, cs in agg.items() if k.startswith(keep)},
    }
    json.dump(out, open(dst + "_pmc.json", "w"), indent=1)
    dom = [k for k in agg if k.startswith("traceq4_kernel")]
    if not dom:
        dom = [k for k in agg if k.startswith("traceq_kernel<false")]
    if dom:
        tot = collections.defaultdict(float)
        n_disp = 0
        for k in dom:
            for c, v in agg[k].items():
                tot[c] += sum(v)
            n_disp += len(next(iter(agg[k].values())))
        # the redo launches of the binary kernel are (nearly) empty; a step = launches_per_step launches of the dominant kernel
        steps = n_disp / float(launches_per_step)
        per_step = {c: v / steps for c, v in tot.items()}
        per_step["names"] = dom
        per_step["launches_profiled"] = n_disp
        per_step["hbm_bytes"] = (2.0 * per_step.get("FETCH_SIZE", 0.0) + per_step.get("WRITE_SIZE", 0.0)) * 1024.0
        per_step["l2_bytes"] = per_step.get("TCP_TCC_READ_REQ_sum", 0.0) * 64.0
        per_step["lds_bytes"] = per_step.get("SQ_INSTS_LDS", 0.0) * 64.0 * 16.0
        summary = {"source_sha": out["source_sha"], "source": os.path.basename(dst) + "_pmc.json", "dominant": per_step, "all_kernels": all_kernels_per_step(out["kernels"]),
                   "note": "per-step sums over the dominant kernel's launches (1 + max_bounce per step); see tools/summarize_profile.py"}
        json.dump(summary, open(os.path.join(os.path.dirname(os.path.abspath(dst)), "pmc_summary.json"), "w"), indent=1)
        print("dominant", dom, "per step: VALU %.4g SALU %.4g HBM %.4g B L2 %.4g B" % (
            per_step.get("SQ_INSTS_VALU", 0), per_step.get("SQ_INSTS_SALU", 0), per_step["hbm_bytes"], per_step["l2_bytes"]))
    for k, cs in out["kernels"].items():
        print(k, {c: round(v["mean_per_dispatch"]) for c, v in list(cs.items())[:4]})


if __name__ == "__main__":
    main()
