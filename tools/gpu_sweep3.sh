#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
for c in C5 C3 C2 C4; do for k in 0 16 13 11 9 7; do EZRT_STACK_CAP=$k python tools/config_one.py $c 2>&1 | grep -v amdgpu | sed "s/^/stack_cap=$k /"; done; done
EZRT_STACK_CAP=9 EZRT_DEBUG_STAGES=1 python tools/config_one.py C5 4 2>&1 | grep -E "re-traced" | head -12
EZRT_STACK_CAP=9 EZRT_DEBUG_STAGES=1 python tools/config_one.py C2 4 2>&1 | grep -E "re-traced" | head -6
