#!/bin/bash
run() { env "$@" python tools/config_one.py $cfg 2>&1 | grep -v amdgpu | sed "s/^/$* /"; }
for cfg in C2 C4; do
  run X=0
  for v in 2048 4096 6144 8192; do run EZRT_SHADE_WGS=$v; done
  run X=0
  for v in 1 2; do run EZRT_SPLIT_SHADE=$v; done
  for v in 20 28; do run EZRT_LEAF_THRESHOLD=$v EZRT_REFILL_MIN=$v; done
done
