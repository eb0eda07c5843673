#!/bin/bash
# tools/gpu_split.sh: EZRT_SPLIT_LEAF A/B (0 = leaves as they are) on the BASELINE configs + the parity tests that stress the traversal
cfgs=${1:-"C2 C3 C5"}; vals=${2:-"0 2 1 3"}
for cfg in $cfgs; do
  for v in $vals; do
    EZRT_SPLIT_LEAF=$v python tools/config_one.py $cfg 2>&1 | grep -v amdgpu | sed "s/^/split=$v  /"
  done
done
