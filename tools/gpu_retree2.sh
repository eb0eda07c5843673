#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
for cfg in "16 0" "32 0" "64 0" "16 64" "16 1024" "32 100000000"; do set -- $cfg; for c in C5 C3 C2; do EZRT_RETREE_BINS=$1 EZRT_RETREE_SWEEP=$2 timeout 600 python tools/config_one.py $c 2>&1 | grep -v amdgpu | sed "s/^/bins=$1 sweep=$2 /"; done; done
