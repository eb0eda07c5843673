#!/bin/bash
# tools/gpu_ab_base.sh TAG "CFG ..." ROUNDS: the working tree's library against ezrt_amd/lib/ab/libezrt_hip_base.so (built from the last
# commit by tools/build_variant.sh base), config_one.py interleaved; SUITE=1 first runs the -m gpu suite on the working tree's library
TAG=$1; CFGS=$2; ROUNDS=${3:-2}
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/$TAG; mkdir -p $O; cd $R
if [ -n "$SUITE" ]; then (time timeout 1500 python -m pytest tests -m gpu -x -q) > $O/pytest_gpu.log 2>&1; grep -E "passed|failed|error" $O/pytest_gpu.log | tail -3; fi
for r in $(seq $ROUNDS); do for c in $CFGS; do
  EZRT_HIP_LIB=$PWD/ezrt_amd/lib/ab/libezrt_hip_base.so timeout 120 python tools/config_one.py $c 2>&1 | grep -v amdgpu | sed "s/^/base  /"
  timeout 120 python tools/config_one.py $c 2>&1 | grep -v amdgpu | sed "s/^/new   /"
done; done 2>&1 | tee $O/ab.txt
