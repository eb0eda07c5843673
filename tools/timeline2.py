#!/usr/bin/env python3
"""Kernel timeline with absolute start/end (us from the first raygen of the chosen step) -- shows overlap across streams."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
seq = [(r["Kernel_Name"].split("ezd::")[1][:30], int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Queue_Id", r.get("Stream_Id", "?")))
       for r in rows if "ezd::" in r["Kernel_Name"]]
acc = [i for i, o in enumerate(seq) if o[0].startswith("accumulate")]
which = int(sys.argv[2]) if len(sys.argv) > 2 else 2
i0 = acc[which - 1] + 1
i1 = acc[which + 1] + 1 if which + 1 < len(acc) else len(seq)
t0 = seq[i0][1]
for o in seq[i0:i1]:
    print("%-32s q%-3s start %8.1f end %8.1f dur %7.1f" % (o[0], o[3], (o[1] - t0) / 1e3, (o[2] - t0) / 1e3, (o[2] - o[1]) / 1e3))
