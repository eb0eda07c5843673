#!/bin/bash
# tools/gpu_timeline.sh TAG [shard args]: kernel timeline (durations + gaps) of one C2 step of the bench command -> gpurun_out/TAG/timeline.txt
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/$TAG; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --output-format csv -d $O/kt -o k -- python $R/bench.py --steps 4 --warmup 2 --windows 1 --cpu-seconds 0 --extras 0 --configs none --model 0 "$@" > $O/kt.log 2>&1
f=$(find $O/kt -name "*kernel_trace.csv" | head -1)
python3 $R/tools/timeline.py $f 5 > $O/timeline.txt 2>&1
cat $O/timeline.txt
