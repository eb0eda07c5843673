#!/bin/bash
run() { env "$@" python tools/config_one.py $cfg 2>&1 | grep -v amdgpu | sed "s/^/$* /"; }
for cfg in C2 C3 C5 C4; do
  for v in 0 1 2 0 1 2; do run EZRT_COLLAPSE_RULE=$v; done
done
