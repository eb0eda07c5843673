#!/bin/bash
# tools/gpu_tl.sh <tag> [ENV=VAL ...]: kernel timeline of one C2 step under rocprofv3 --kernel-trace
R=${GRAFT_REPO_ROOT:-/root/repo}; TAG=$1; shift; O=$R/gpurun_out/tl_$TAG; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
env "$@" timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O -o t -- python $R/bench.py --steps 10 --warmup 3 --windows 2 --extras 0 --cpu-seconds 0 > $O/log.txt 2>&1
cd $R; python tools/timeline3.py $O/t_kernel_trace.csv | head -24
