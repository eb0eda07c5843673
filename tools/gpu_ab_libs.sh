#!/bin/bash
# tools/gpu_ab_libs.sh "CFG ..." NAME ...: tools/config_one.py of each config under the default library and under each A/B build
# (ezrt_amd/lib/ab/libezrt_hip_NAME.so), interleaved, two rounds.
cfgs=$1; shift
for round in 1 2; do
  for cfg in $cfgs; do
    python tools/config_one.py $cfg 2>&1 | grep -v amdgpu | sed "s/^/default  /"
    for n in "$@"; do
      EZRT_HIP_LIB=$PWD/ezrt_amd/lib/ab/libezrt_hip_$n.so python tools/config_one.py $cfg 2>&1 | grep -v amdgpu | sed "s/^/$n  /"
    done
  done
done
