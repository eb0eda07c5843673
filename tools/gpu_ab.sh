#!/bin/bash
# tools/gpu_ab.sh TAG "CFG ..." ROUNDS "ENV=V ..." ...: tools/ab_env.sh under gpurun with its output kept in gpurun_out/TAG/ab.txt
# (SUITE=1 first runs the -m gpu suite with the LAST environment set)
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/$TAG; mkdir -p $O; cd $R
if [ -n "$SUITE" ]; then
  LAST="${@: -1}"
  (time env $LAST timeout 1500 python -m pytest tests -m gpu -x -q) > $O/pytest_gpu.log 2>&1; grep -E "passed|failed|error" $O/pytest_gpu.log | tail -3
fi
bash tools/ab_env.sh "$@" 2>&1 | tee $O/ab.txt
