#!/usr/bin/env python3
"""tools/summarize_config_profile.py CFG <gpurun_out/profcfg_CFG> <out dir>: the raw rocprofv3 output of tools/profile_configs.sh ->
<out dir>/<cfg>_kernel_stats.csv (as rocprofv3 wrote it) and <cfg>_pmc_summary.json: per kernel -- calls, average and total time,
share of the GPU time, VALU wave-instructions per dispatch, issue rate (T wave-instr/s) and its fraction of the measured 1.086 T /
nominal 1.229 T peaks, lane fill, share of wave-cycles waiting, HBM bytes ((2 FETCH_SIZE + WRITE_SIZE) KB, gfx950 correction of
MI355X_MICROARCH.md) and HBM GB/s, L1 -> L2 read requests; stamped with the hash of the GPU sources."""
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


def main():
    cfg, src, dst = sys.argv[1], sys.argv[2], sys.argv[3]
    # (round 5) the spp of the kernel-trace pass and of the counter passes, so that a reader need not redo the ratio
    spp_trace = int(sys.argv[4]) if len(sys.argv) > 4 else None
    spp_pmc = int(sys.argv[5]) if len(sys.argv) > 5 else None
    os.makedirs(dst, exist_ok=True)
    stats = glob.glob(os.path.join(src, "stats", "**", "*kernel_stats.csv"), recursive=True)
    dur = {}
    total = 0.0
    if stats:
        shutil.copy(stats[0], os.path.join(dst, cfg.lower() + "_kernel_stats.csv"))
        for r in csv.DictReader(open(stats[0])):
            if "ezd::" not in r["Name"]:
                continue
            dur[short(r["Name"])] = (int(r["Calls"]), float(r["AverageNs"]), float(r["TotalDurationNs"]))
            total += float(r["TotalDurationNs"])
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    pdur = collections.defaultdict(dict)     # kernel -> counter -> average duration (ns) of the kernel IN THE PASS that collected the counter
    for d in sorted(glob.glob(os.path.join(src, "pmc*"))):
        if not os.path.isdir(d):
            continue
        names = set()
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            for r in csv.DictReader(open(f)):
                agg[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
                names.add(r["Counter_Name"])
        kd = collections.defaultdict(list)
        for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
            for r in csv.DictReader(open(f)):
                kd[short(r["Kernel_Name"])].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
        for k, v in kd.items():
            for n in names:
                pdur[k][n] = sum(v) / len(v)
    kernels = {}
    for k, (calls, avg_ns, tot_ns) in sorted(dur.items(), key=lambda kv: -kv[1][2]):
        c = {n: sum(v) / len(v) for n, v in agg.get(k, {}).items()}
        e = {"calls": calls, "avg_us": round(avg_ns / 1e3, 2), "total_ms": round(tot_ns / 1e6, 3), "share_of_gpu_time": round(tot_ns / total, 4) if total else None}
        pd = pdur.get(k, {})
        if c.get("SQ_INSTS_VALU", 0) > 0 and pd.get("SQ_INSTS_VALU", 0) > 0:
            e["avg_us_in_counter_pass"] = round(pd["SQ_INSTS_VALU"] / 1e3, 2)
            rate = c["SQ_INSTS_VALU"] / (pd["SQ_INSTS_VALU"] * 1e-9) / 1e12
            e.update({"valu_wave_instr_per_dispatch": int(c["SQ_INSTS_VALU"]), "issue_rate_T": round(rate, 4),
                      "issue_frac_of_measured_peak_1.086": round(rate / 1.086, 4), "issue_frac_of_nominal_peak_1.229": round(rate / 1.2288, 4)})
            if "SQ_INSTS_SALU" in c:
                e["salu_per_valu"] = round(c["SQ_INSTS_SALU"] / c["SQ_INSTS_VALU"], 3)
            if "SQ_THREAD_CYCLES_VALU" in c and c["SQ_INSTS_VALU"] > 0:
                e["lane_fill"] = round(c["SQ_THREAD_CYCLES_VALU"] / (64.0 * c["SQ_INSTS_VALU"]), 4)
        if c.get("SQ_WAVE_CYCLES"):
            wc = c["SQ_WAVE_CYCLES"]
            e["wave_cycles_waiting"] = round(c.get("SQ_WAIT_ANY", 0.0) / wc, 4)
            # (MI355X_MICROARCH.md "SQ": WAIT_ANY + WAIT_INST_ANY + ACTIVE_INST_ANY ~ WAVE_CYCLES, disjoint; WAIT_INST_LDS is a sub-bucket of
            # WAIT_INST_ANY and comes from a pass of its own: scaled by that pass's wave cycles is not possible, so it is quoted against this one's)
            e["stall_split"] = {"parked_on_s_waitcnt_or_barrier": round(c.get("SQ_WAIT_ANY", 0.0) / wc, 4),
                                "stalled_at_issue": round(c.get("SQ_WAIT_INST_ANY", 0.0) / wc, 4),
                                "stalled_at_issue_lds_pipe": round(c["SQ_WAIT_INST_LDS"] / wc, 4) if "SQ_WAIT_INST_LDS" in c else None,
                                "issuing": round(c.get("SQ_ACTIVE_INST_ANY", 0.0) / wc, 4),
                                "issuing_valu": round(c.get("SQ_ACTIVE_INST_VALU", 0.0) / wc, 4)}
        if "FETCH_SIZE" in c and "WRITE_SIZE" in c and pd.get("FETCH_SIZE", 0) > 0:
            hbm = (2.0 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024.0
            sec = 0.5 * (pd["FETCH_SIZE"] + pd.get("WRITE_SIZE", pd["FETCH_SIZE"])) * 1e-9   # (the two counters come from two passes)
            e.update({"hbm_bytes_per_dispatch_in_counter_pass": int(hbm), "hbm_GBs": round(hbm / sec / 1e9, 1), "hbm_frac_of_8TBs": round(hbm / sec / 8e12, 4)})
        if "TCP_TCC_READ_REQ_sum" in c:
            e["l1_to_l2_read_requests"] = int(c["TCP_TCC_READ_REQ_sum"])
            e["l1_accesses"] = int(c.get("TCP_TOTAL_CACHE_ACCESSES_sum", 0))
        kernels[k] = e
    trace_share = sum(v["share_of_gpu_time"] or 0 for k, v in kernels.items() if k.startswith("traceq"))
    # HBM bytes of ALL kernels per pixel-sample (VERDICT r4 #3 asks for this figure): the counter pass renders tools/config_one.py's five
    # calls (1 warm-up + 3 timed + 1 with launch events) of W x H x spp_pmc pixel-samples each
    per_sample = None
    if spp_pmc:
        from ezrt_amd import scenes
        c0 = scenes.CONFIGS[cfg]
        samples = 5.0 * c0["width"] * c0["height"] * spp_pmc
        tot_hbm = 0.0
        by_kernel = {}
        for k, e in kernels.items():
            n_disp = len(agg.get(k, {}).get("FETCH_SIZE", []))
            if "hbm_bytes_per_dispatch_in_counter_pass" in e and n_disp:
                by_kernel[k] = round(e["hbm_bytes_per_dispatch_in_counter_pass"] * n_disp / samples, 1)
                tot_hbm += e["hbm_bytes_per_dispatch_in_counter_pass"] * n_disp
        per_sample = {"all_kernels": round(tot_hbm / samples, 1), "by_kernel": {k: v for k, v in by_kernel.items() if v >= 1.0}}
    out = {"config": cfg, "command": "EZRT_PIPELINE_CALLS=0 tools/config_one.py %s at the BASELINE spp (tools/profile_configs.sh): chunks NOT overlapped, every kernel measured alone" % cfg, "source_sha": gpu_source_hash(),
           "note": "kernel times and shares: rocprofv3 --kernel-trace --stats of the config at its BASELINE spp; counters: separate --pmc passes at one "
                   "chunk's worth of frames (same kernels, per-dispatch work smaller by the spp ratio for the per-chunk stages) -- the issue rate of a "
                   "kernel is its counters' VALU instructions / its average duration IN THE SAME pass",
           "spp_kernel_trace_pass": spp_trace, "spp_counter_passes": spp_pmc,
           "per_dispatch_scale_counter_pass_over_kernel_trace_pass": round(spp_pmc / float(spp_trace), 4) if (spp_trace and spp_pmc) else None,
           "hbm_bytes_per_pixel_sample": per_sample,
           "gpu_time_ms_all_ezd_kernels": round(total / 1e6, 3), "trace_kernels_share_of_gpu_time": round(trace_share, 4),
           "hbm_correction": "bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 (MI355X_MICROARCH.md: gfx950 FETCH_SIZE tallies 128-B requests at 64 B)",
           "kernels": kernels}
    json.dump(out, open(os.path.join(dst, cfg.lower() + "_pmc_summary.json"), "w"), indent=1)
    print(cfg, "GPU time %.1f ms, trace share %.2f;" % (total / 1e6, trace_share),
          "; ".join("%s %.0f us x%d" % (k[:40], v["avg_us"], v["calls"]) for k, v in list(kernels.items())[:5]))


if __name__ == "__main__":
    main()
