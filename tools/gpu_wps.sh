#!/bin/bash
run() { env "$@" python tools/config_one.py $cfg 2>&1 | grep -v amdgpu | sed "s/^/$* /"; }
for cfg in C2 C3 C4 C5; do
  for r in 1 2; do
  run X=0
  run EZRT_TRACE_WPS=7
  run EZRT_TRACE_WPS_REL=6
  run EZRT_TRACE_WPS=7 EZRT_TRACE_WPS_REL=6
  done
done
