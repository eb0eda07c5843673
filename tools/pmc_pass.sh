#!/bin/bash
# tools/pmc_pass.sh <tag> "<counters>" : one rocprofv3 --pmc pass (+ --kernel-trace) of the default bench
TAG=$1; PMC=$2
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 150 rocprofv3 --pmc $PMC --kernel-trace --output-format csv -d $OUT -o p -- python $R/bench.py --steps 2 --warmup 1 --cpu-seconds 0 > $OUT/log.txt 2>&1 || tail -3 $OUT/log.txt
python3 - <<PY
import csv, collections
rows = list(csv.DictReader(open("$OUT/p_counter_collection.csv")))
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    k = r['Kernel_Name']
    if 'traceq_kernel<false' in k: k='traceq'
    elif 'shade_kernel<50, false>' in k: k='shade'
    else: continue
    agg[k][r['Counter_Name']].append(float(r['Counter_Value']))
for k, cs in agg.items():
    for c, v in cs.items():
        per_stage = [sum(v[i::5])/len(v[i::5]) for i in range(5)]
        print(k, c, ["%.3g" % x for x in per_stage])
PY
