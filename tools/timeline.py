#!/usr/bin/env python3
"""Print the kernel timeline of one bench step from a rocprofv3 kernel-trace CSV."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
seq = []
for r in rows:
    n = r["Kernel_Name"]
    if "ezd::" not in n:
        continue
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    seq.append((n.split("ezd::")[1][:34], (e - s) / 1e3, s, e))
start = [i for i, o in enumerate(seq) if o[0].startswith("raygen") or o[0].startswith("chunk_prologue")]
which = int(sys.argv[2]) if len(sys.argv) > 2 else 1
i0, i1 = start[which], start[which + 1]
prev = None
for o in seq[i0:i1]:
    gap = (o[2] - prev) / 1e3 if prev else 0
    print("%-36s %9.1f us   gap_before %6.1f us" % (o[0], o[1], gap))
    prev = o[3]
print("step span us %.1f" % ((seq[i1][2] - seq[i0][2]) / 1e3))
