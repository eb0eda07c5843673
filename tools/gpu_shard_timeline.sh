#!/bin/bash
# tools/gpu_shard_timeline.sh CFG R N: kernel durations of one call of shard R of N (rocprofv3 kernel trace) -> gpurun_out/shard_CFG_N/timeline.txt
CFG=$1; RR=$2; N=$3
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/shard_${CFG}_$N; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --output-format csv -d $O/kt -o k -- python $R/tools/shard_one.py $CFG $RR $N > $O/kt.log 2>&1
grep "ms/call" $O/kt.log
f=$(find $O/kt -name "*kernel_trace.csv" | head -1)
python3 $R/tools/timeline.py $f 4 > $O/timeline.txt 2>&1
cat $O/timeline.txt
