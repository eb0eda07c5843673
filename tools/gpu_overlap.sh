#!/bin/bash
# tools/gpu_overlap.sh TAG [env...]: kernel trace of 16 back-to-back C2 calls (bench windows) -> tools/overlap.py
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/$TAG; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
env "$@" timeout 200 rocprofv3 --kernel-trace --output-format csv -d $O/kt -o k -- python $R/bench.py --steps 16 --warmup 2 --windows 1 --cpu-seconds 0 --extras 0 --configs none --model 0 > $O/kt.log 2>&1
f=$(find $O/kt -name "*kernel_trace.csv" | head -1)
python3 $R/tools/overlap.py $f 3 | tee $O/overlap.txt
rm -rf $O/kt
