"""tools/timeline3.py <kernel_trace.csv> [step]: kernels of one render call in start order -- offset, duration, gap to the
previous kernel's end (us), queue -- from a rocprofv3 --kernel-trace CSV of bench.py."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'raygen' in r['Kernel_Name'] or 'chunk_prologue' in r['Kernel_Name']]
k = int(sys.argv[2]) if len(sys.argv) > 2 else -4
i0, i1 = idx[k], idx[k + 2]
t0 = int(rows[i0]['Start_Timestamp']); prev_end = t0
for r in rows[i0:i1 + 1]:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    n = r['Kernel_Name'].replace('void ezd::', '').replace('ezd::', '').split('(')[0]
    print("%8.1f  dur %7.1f  gap %6.1f  q%s %s" % ((s - t0) / 1e3, (e - s) / 1e3, (s - prev_end) / 1e3, r.get('Queue_Id', '?'), n[:60]))
    prev_end = max(prev_end, e)
