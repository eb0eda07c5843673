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

# ---- the paragraph itself, ready to paste
c3, c4, c5 = (d["configs"][k] for k in ("C3", "C4", "C5"))
f3, f4, f5 = (f["configs"][k] for k in ("C3", "C4", "C5"))
r3, r4, r5 = c3["roofline"], c4["roofline"], c5["roofline"]
lohi = lambda a, b: ("%.2f" % a) if abs(a - b) < 0.005 else "%.2f-%.2f" % (min(a, b), max(a, b))
s2, s4 = sm["C2"]["shards"], sm["C4"]["shards"]
print()
print("""Results on MI355X, round 6 (`profiles/r6/`: `final_bench.json` + `final_bench_first_process.json`, `final_kernel_stats.csv`, `final_pmc.json`, `pmc_summary.json` and
`c{3,4,5}_kernel_stats.csv` / `c{3,4,5}_pmc_summary.json` all stamped `%s`, `valu_issue_microbench.txt`, `rccl_floor.json`; `tools/results_paragraph.py r6` prints every number below from
those files): C2 **%.2f / %.2f Grays/s** (first / second process of the kept run: %.3f / %.3f ms per 64-spp frame, windows %.1f-%.1f ms; other boxes of the round 14.6-15.1), a lone call
**%.2f / %.2f Grays/s** (%.3f / %.3f ms).  Trace launches per step in the sequential pass %.3f ms (primary %.0f µs, bounce stages %.0f µs on average; r5's box: 575 / 175);
`shade_hit<50,0>` %.0f, `shade_miss<50,1>` %.0f, `shade_miss<50,0>` %.0f, `shade_hit<50,1>` %.0f, `accumulate` %.0f µs.  The dominant kernel issues %.1f M VALU wave-instructions per step at
%.3f T/s = **%.4f of the nominal 1.2288 T** (`roofline.frac`), %.3f of the best measured ceiling (%.3f T; this box measured %.3f T, the kernel's own opcode mix %.3f T), lane fill %.3f
→ useful lane-instructions %.3f of the chip's peak; HBM %.3f (%d MB per launch), L2 %.3f, LDS %.3f; wave-cycles: parked on `s_waitcnt` %.3f, stalled at issue %.3f (the LDS pipe: %.3f),
issuing %.3f (VALU %.3f).  The chip over a whole step (%d M wave-instructions of every kernel): %.3f of nominal in the timed windows, %.3f for a call alone.  Configs at their
BASELINE spp, crops of the timed frames bit-identical to the oracle: **C3 %s Grays/s** (%.1f ms per 128 spp, trace %.1f), **C4 %s** (%.1f ms per 256 spp, trace %.1f; %d
non-finite pixels, also by `ezrt_frame_nonfinite`), **C5 %s** (%.3f s per 512 spp, trace %.3f; %d non-finite pixels) — r5: 7.61 / 16.30 / 5.03; the ring stack is the C5 and C3 difference, C4 is
box noise.  Their dominant kernels (`configs.*.roofline`): C3 bounce-stage trace `<6,F,F,2,F,F,GS>` %.0f %% of the GPU time, HBM **%.2f** (%.2f TB/s, %.0f B per pixel-sample of %.0f), issue %.2f, lane fill
%.2f, parked %.2f; C4 `shade_hit<51,0>` %.0f %%, HBM **%.2f** (%.2f TB/s, %.0f B per pixel-sample of %.0f), issue %.2f, stalled at issue %.2f; C5 bounce-stage trace (SEMI) %.0f %%, issue **%.2f**, HBM %.2f
(%.0f B per pixel-sample of %.0f), lane fill %.2f, parked %.2f.  Scaling model with the RCCL floor (%.4f ms): C2 %.2f / %.2f / %.2f× on 2 / 4 / 8 shards (critical shards %.2f / %.2f / %.2f ms of
a %.2f-ms frame), C4 %.2f / %.2f / **%.2f×** (%.1f / %.1f / %.2f ms of %.1f).  CPU: the oracle %.1f Mrays/s on %d cores (%.2f on one thread), the reference's compiled shader %.2f on one thread;
`linf_vs_gpu` %s for the complete frame of the timed loop.""" % (
    ph["source"].split("@")[1].strip(), f["value"] / 1e3, d["value"] / 1e3, f["ms_per_step"], d["ms_per_step"], min(f["timing"]["window_ms"] + d["timing"]["window_ms"]),
    max(f["timing"]["window_ms"] + d["timing"]["window_ms"]), f["value_lone_call"] / 1e3, d["value_lone_call"] / 1e3, f["ms_per_step_lone_call"], d["ms_per_step_lone_call"],
    rf["trace_ms_per_step"], g("traceq4_kernel<7, true"), g("traceq4_kernel<6, false"), g("shade_hit_kernel<50, false, 0"), g("shade_miss_kernel<50, false, 1"),
    g("shade_miss_kernel<50, false, 0"), g("shade_hit_kernel<50, false, 1"), g("accumulate"), ph["valu_wave_instr_per_step"] / 1e6, ph["issue_rate_T"], rf["frac"],
    rf["frac_of_measured_peak"], rf["peak_measured"], best("v_mov_b32"), best("traceq4 opcode mix"), ph["lane_fill"], rf["useful_lane_frac"], ph["hbm_frac"],
    ph["traffic_per_launch"] / 1e6, ph["l2_frac"], ph["lds_frac"], ss["parked_on_s_waitcnt"], ss["stalled_at_issue"], ss["stalled_at_issue_lds_pipe"], ss["issuing"], ss["issuing_valu"],
    rf["whole_step"]["valu_wave_instr_all_kernels"] / 1e6, rf["whole_step"]["issue_frac_in_the_timed_windows"], rf["whole_step"]["issue_frac_one_call_at_a_time"],
    lohi(f3["Mrays_s"] / 1e3, c3["Mrays_s"] / 1e3), c3["ms_per_frame"], c3["trace_ms"], lohi(f4["Mrays_s"] / 1e3, c4["Mrays_s"] / 1e3), c4["ms_per_frame"], c4["trace_ms"], c4["non_finite_pixels"],
    lohi(f5["Mrays_s"] / 1e3, c5["Mrays_s"] / 1e3), c5["ms_per_frame"] / 1e3, c5["trace_ms"] / 1e3, c5["non_finite_pixels"],
    100 * r3["share_of_gpu_time"], r3["ceilings"]["hbm"], r3["achieved"] / 1e3, r3["hbm_bytes_per_pixel_sample"], r3["hbm_bytes_per_pixel_sample_all_kernels"], r3["ceilings"]["valu_issue"],
    r3["lane_fill"], r3["stall_split"]["parked_on_s_waitcnt_or_barrier"],
    100 * r4["share_of_gpu_time"], r4["ceilings"]["hbm"], r4["achieved"] / 1e3, r4["hbm_bytes_per_pixel_sample"], r4["hbm_bytes_per_pixel_sample_all_kernels"], r4["ceilings"]["valu_issue"],
    r4["stall_split"]["stalled_at_issue"],
    100 * r5["share_of_gpu_time"], r5["ceilings"]["valu_issue"], r5["ceilings"]["hbm"], r5["hbm_bytes_per_pixel_sample"], r5["hbm_bytes_per_pixel_sample_all_kernels"], r5["lane_fill"],
    r5["stall_split"]["parked_on_s_waitcnt_or_barrier"],
    s2["8"]["rccl_floor_ms"], s2["2"]["predicted_speedup"], s2["4"]["predicted_speedup"], s2["8"]["predicted_speedup"], s2["2"]["critical_path_ms"], s2["4"]["critical_path_ms"],
    s2["8"]["critical_path_ms"], sm["C2"]["one_gpu_ms"], s4["2"]["predicted_speedup"], s4["4"]["predicted_speedup"], s4["8"]["predicted_speedup"], s4["2"]["critical_path_ms"],
    s4["4"]["critical_path_ms"], s4["8"]["critical_path_ms"], sm["C4"]["one_gpu_ms"], cb["value"], cb["cores"], cb.get("one_thread_Mrays_s", 0), cb.get("reference_shader_one_thread_Mrays_s", 0),
    cb.get("linf_vs_gpu")))
