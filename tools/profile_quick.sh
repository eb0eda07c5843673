#!/bin/bash
# tools/profile_quick.sh <tag> [bench args...]: kernel-trace stats only
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o s -- python $R/bench.py --steps 3 --warmup 1 --cpu-seconds 0 "$@" > $OUT/stats.log 2>&1
python3 - <<PY
import csv
rows = list(csv.DictReader(open("$OUT/stats/s_kernel_stats.csv")))
for r in rows[:14]:
    print("%-70s calls %5s avg_us %10.1f total_ms %9.3f  %5s%%" % (r["Name"][:70], r["Calls"], float(r["AverageNs"])/1e3, float(r["TotalDurationNs"])/1e6, r["Percentage"]))
PY
