#!/bin/bash
run() { env "$@" python tools/config_one.py $cfg 2>&1 | grep -v amdgpu | sed "s/^/$* /"; }
for cfg in C2 C4; do
  run X=0
  run EZRT_PIPES=2
  run EZRT_PIPES=2 EZRT_TRACE_WPS=5 EZRT_TRACE_WPS_REL=5
  run EZRT_PIPES=2 EZRT_TRACE_WPS=5 EZRT_TRACE_WPS_REL=6
  run EZRT_PIPES=2 EZRT_TRACE_WPS=4 EZRT_TRACE_WPS_REL=4
  run EZRT_PIPES=2 EZRT_SUB_FRAMES=16
  run X=0
done
