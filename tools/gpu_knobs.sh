#!/bin/bash
# tools/gpu_knobs.sh "CFG ..." : one-at-a-time sweep of the trace schedule knobs (tools/config_one.py, EZRT_<NAME> at scene creation)
cfgs=${1:-"C2 C5"}
run() { env "$@" python tools/config_one.py $cfg 2>&1 | grep -v amdgpu | sed "s/^/$* /"; }
for cfg in $cfgs; do
  run X=0; run X=0
  for v in 8 16 32 40; do run EZRT_REFILL_MIN=$v; done
  for v in 64 256; do run EZRT_POOL_MAX=$v; done
  for v in 25 75; do run EZRT_STATIC_PCT=$v; done
  for v in 5 7; do run EZRT_TRACE_WPS=$v; done
  for v in 6; do run EZRT_TRACE_WPS_REL=$v; done
  for v in 8 32 64; do run EZRT_MIN_STAGED=$v; done
  for v in 0; do run EZRT_STEAL=$v; done
  for v in 8 16; do run EZRT_LEAF_THRESHOLD=$v EZRT_REFILL_MIN=16; done
done
