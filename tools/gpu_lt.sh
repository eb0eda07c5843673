#!/bin/bash
for cfg in C2 C3 C4 C5; do
  for lt in 8 12 16 20 24; do
    EZRT_LEAF_THRESHOLD=$lt python tools/config_one.py $cfg 2>&1 | grep -v amdgpu | sed "s/^/leaf_threshold=$lt  /"
  done
done
