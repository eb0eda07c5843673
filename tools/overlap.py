#!/usr/bin/env python3
"""tools/overlap.py <kernel_trace.csv> [skip_steps]: how the kernels of consecutive render calls share the GPU in steady state (pipeline_calls): wall span, time with
no kernel / one / two or more kernels running, and each kernel's mean duration there (to compare with the profile of kernels running ALONE)."""
import collections, csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "ezd::" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
starts = [i for i, r in enumerate(rows) if "chunk_prologue" in r["Kernel_Name"]]
skip = int(sys.argv[2]) if len(sys.argv) > 2 else 3
i0, i1 = starts[skip], starts[-2]
sel = rows[i0:i1]
t0, t1 = int(sel[0]["Start_Timestamp"]), max(int(r["End_Timestamp"]) for r in sel)
ev = []
for r in sel:
    ev.append((int(r["Start_Timestamp"]), 1))
    ev.append((int(r["End_Timestamp"]), -1))
ev.sort()
busy = collections.Counter()
n, last = 0, t0
for t, d in ev:
    busy[min(n, 2)] += t - last
    last = t
    n += d
steps = len([i for i in starts if i0 <= i < i1])
span = (t1 - t0) / 1e3
print("steps %d  span %.1f us  per step %.1f us | no kernel %.1f %%  one kernel %.1f %%  two or more %.1f %%" % (
    steps, span, span / steps, 100.0 * busy[0] / (t1 - t0), 100.0 * busy[1] / (t1 - t0), 100.0 * busy[2] / (t1 - t0)))
per = collections.defaultdict(list)
for r in sel:
    per[r["Kernel_Name"].split("ezd::")[1].split("(")[0][:60]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
tot = sum(sum(v) for v in per.values())
print("sum of kernel durations per step %.1f us" % (tot / steps))
for k, v in sorted(per.items(), key=lambda kv: -sum(kv[1])):
    print("  %-62s x%5.1f/step  mean %8.1f us  per step %8.1f us" % (k, len(v) / steps, sum(v) / len(v), sum(v) / steps))
