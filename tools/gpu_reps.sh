#!/bin/bash
run() { env "$@" python tools/config_one.py $cfg 2>&1 | grep -v amdgpu | sed "s/^/$* /"; }
for cfg in C2 C5 C3 C4; do
  for g in 1 2 3 1 2; do run EZRT_INNER_REPS=$g; done
done
