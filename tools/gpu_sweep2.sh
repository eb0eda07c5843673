#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
for lt in 16 24 32 40 48; do EZRT_LEAF_THRESHOLD=$lt python tools/config_one.py C2 2>&1 | grep -v amdgpu | sed "s/^/leaf_threshold=$lt /"; done
for rm in 12 16 24 32 40; do EZRT_REFILL_MIN=$rm python tools/config_one.py C2 2>&1 | grep -v amdgpu | sed "s/^/refill_min=$rm /"; done
for lt in 16 24 32 40; do EZRT_LEAF_THRESHOLD=$lt python tools/config_one.py C5 2>&1 | grep -v amdgpu | sed "s/^/leaf_threshold=$lt /"; done
for rm in 16 24 32; do EZRT_REFILL_MIN=$rm python tools/config_one.py C5 2>&1 | grep -v amdgpu | sed "s/^/refill_min=$rm /"; done
for w in 5 6 7; do EZRT_TRACE_WPS=$w python tools/config_one.py C5 2>&1 | grep -v amdgpu | sed "s/^/trace_wps=$w /"; done
