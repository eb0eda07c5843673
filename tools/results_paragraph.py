#!/usr/bin/env python3
"""tools/results_paragraph.py [rN]: the numbers of DESIGN.md section 6 "Results on MI355X, round N", from profiles/rN/ (so that the text is derived, not typed)."""
import csv, json, os, sys
R = sys.argv[1] if len(sys.argv) > 1 else "r6"
P = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", R)
L = lambda f: json.loads(open(os.path.join(P, f)).read().strip().splitlines()[-1])
d, f = L("final_bench.json"), L("final_bench_first_process.json")
rf, ph = d["roofline"], d["roofline"]["physical"]
ks = {}
for r in csv.DictReader(open(os.path.join(P, "final_kernel_stats.csv"))):
    if "ezd::" in r["Name"]:
        ks[r["Name"].split("ezd::")[1].split("(")[0]] = float(r["AverageNs"]) / 1e3
g = lambda pre: next(v for k, v in ks.items() if k.startswith(pre))
mb = open(os.path.join(P, "valu_issue_microbench.txt")).read().splitlines()
best = lambda name: max(float(l.split(" chip ")[1].split()[0]) for l in mb if l.startswith(name))
sm = d["scaling_model"]
print("hash", ph["source"].split("@")[1].strip())
print("C2 first/second %.2f / %.2f Grays/s, %.3f / %.3f ms, windows %.1f-%.1f ms; lone %.2f / %.2f Grays/s (%.3f / %.3f ms)" % (
    f["value"] / 1e3, d["value"] / 1e3, f["ms_per_step"], d["ms_per_step"], min(d["timing"]["window_ms"]), max(f["timing"]["window_ms"] + d["timing"]["window_ms"]),
    f["value_lone_call"] / 1e3, d["value_lone_call"] / 1e3, f["ms_per_step_lone_call"], d["ms_per_step_lone_call"]))
print("trace ms/step %.3f: primary %.0f us, bounce %.0f us; shade_hit<50,0> %.0f, shade_miss<50,1> %.0f, shade_miss<50,0> %.0f, shade_hit<50,1> %.0f, accumulate %.0f" % (
    rf["trace_ms_per_step"], g("traceq4_kernel<7, true"), g("traceq4_kernel<6, false"), g("shade_hit_kernel<50, false, 0"), g("shade_miss_kernel<50, false, 1"),
    g("shade_miss_kernel<50, false, 0"), g("shade_hit_kernel<50, false, 1"), g("accumulate")))
print("VALU %.1f M/step at %.3f T/s = %.4f nominal, %.3f of measured %.3f (box mov %.3f, mix %.3f), lane fill %.3f, useful %.3f; hbm %.3f (%d MB/launch) l2 %.3f lds %.3f" % (
    ph["valu_wave_instr_per_step"] / 1e6, ph["issue_rate_T"], rf["frac"], rf["frac_of_measured_peak"], rf["peak_measured"], best("v_mov_b32"), best("traceq4 opcode mix"),
    ph["lane_fill"], rf["useful_lane_frac"], ph["hbm_frac"], ph["traffic_per_launch"] / 1e6, ph["l2_frac"], ph["lds_frac"]))
ss = rf["stall_split"]
print("stall: parked %.3f issue-stalled %.3f (lds %.3f) issuing %.3f (valu %.3f); whole step %.3f windows / %.3f alone (%d M)" % (
    ss["parked_on_s_waitcnt"], ss["stalled_at_issue"], ss["stalled_at_issue_lds_pipe"], ss["issuing"], ss["issuing_valu"],
    rf["whole_step"]["issue_frac_in_the_timed_windows"], rf["whole_step"]["issue_frac_one_call_at_a_time"], rf["whole_step"]["valu_wave_instr_all_kernels"] / 1e6))
for k, v in d["configs"].items():
    r = v["roofline"]
    print("%s %.2f / %.2f Grays/s, %.1f ms, trace %.1f, nonfinite %d/%d | %s share %.2f hbm %.2f (%s) issue %.2f lanefill %.2f parked %.2f issue-stalled %.2f B/px-sample %.0f of %.0f" % (
        k, f["configs"][k]["Mrays_s"] / 1e3, v["Mrays_s"] / 1e3, v["ms_per_frame"], v["trace_ms"], v["non_finite_pixels"], v["non_finite_pixels_ezrt_frame_nonfinite"],
        r["kernel"].split("ezd::")[1], r["share_of_gpu_time"], r["ceilings"]["hbm"], r["achieved"] if r["bound"] == "hbm" else "-", r["ceilings"]["valu_issue"], r["lane_fill"],
        r["stall_split"]["parked_on_s_waitcnt_or_barrier"], r["stall_split"]["stalled_at_issue"], r["hbm_bytes_per_pixel_sample"], r["hbm_bytes_per_pixel_sample_all_kernels"]))
for k in ("C2", "C4"):
    print(k, "one gpu %.2f ms; shards" % sm[k]["one_gpu_ms"], {n: (round(v["critical_path_ms"], 2), v["predicted_speedup"]) for n, v in sm[k]["shards"].items()}, "floor", sm[k]["shards"]["8"]["rccl_floor_ms"])
cb = d["cpu_baseline"]
print("cpu %.1f on %d cores, one thread %.2f, ref shader %.2f, linf %s" % (cb["value"], cb["cores"], cb.get("one_thread_Mrays_s", 0), cb.get("reference_shader_one_thread_Mrays_s", 0), cb.get("linf_vs_gpu")))
