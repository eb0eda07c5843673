#!/bin/bash
# tools/gpu_pmc_cfg.sh <config> <tag> "<counters>": one rocprofv3 --pmc pass (+ --kernel-trace) of tools/config_one.py, per-kernel means
R=${GRAFT_REPO_ROOT:-/root/repo}; CFG=$1; TAG=$2; PMC=$3; O=$R/gpurun_out/pmcc_$TAG; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --pmc $PMC --kernel-trace --output-format csv -d $O -o p -- python $R/tools/config_one.py $CFG 4 > $O/log.txt 2>&1 || tail -3 $O/log.txt
python3 - "$O/p_counter_collection.csv" <<'PY'
import csv, collections, sys
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    k = r['Kernel_Name'].replace('void ezd::', '').split('(')[0]
    agg[k][r['Counter_Name']].append(float(r['Counter_Value']))
for k, cs in sorted(agg.items()):
    if not any(x in k for x in ('shade', 'traceq4')): continue
    print(k, {c: "%.4g" % (sum(v) / len(v)) for c, v in cs.items()}, "dispatches", len(next(iter(cs.values()))))
PY
