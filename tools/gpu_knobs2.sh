#!/bin/bash
run() { env "$@" python tools/config_one.py $cfg 2>&1 | grep -v amdgpu | sed "s/^/$* /"; }
for cfg in C2 C3 C5 C4; do
  run X=0
  for v in 32 40 48 56; do run EZRT_REFILL_MIN_REL=$v; done
  run X=0
  for v in 28 32; do run EZRT_REFILL_MIN=$v EZRT_REFILL_MIN_REL=40; done
done
