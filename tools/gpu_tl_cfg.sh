#!/bin/bash
# tools/gpu_tl_cfg.sh <config> <tag> [ENV=VAL ...]: kernel stats of one BASELINE config (tools/config_one.py) under rocprofv3
R=${GRAFT_REPO_ROOT:-/root/repo}; CFG=$1; TAG=$2; shift; shift; O=$R/gpurun_out/tlc_$TAG; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
env "$@" timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O -o t -- python $R/tools/config_one.py $CFG > $O/log.txt 2>&1
cd $R; tail -2 $O/log.txt; python - "$O/t_kernel_stats.csv" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: -float(r['TotalDurationNs']))
tot = sum(float(r['TotalDurationNs']) for r in rows)
for r in rows[:14]:
    print("%6.2f%%  calls %5s  avg %9.1f us  %s" % (100*float(r['TotalDurationNs'])/tot, r['Calls'], float(r['AverageNs'])/1e3, r['Name'].replace('void ezd::','')[:90]))
PY
