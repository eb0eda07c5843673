#!/bin/bash
for cfg in C2 C5; do
  for v in 0 2; do
    for lt in 12 24 33 48; do
      EZRT_SPLIT_LEAF=$v EZRT_LEAF_THRESHOLD=$lt python tools/config_one.py $cfg 2>&1 | grep -v amdgpu | sed "s/^/split=$v leaf_threshold=$lt  /"
    done
  done
done
