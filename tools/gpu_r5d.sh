#!/bin/bash
O=gpurun_out/r5d; mkdir -p $O
EZRT_XSTEAL=1 EZRT_DEBUG_STAGES=2 timeout 60 python tools/debug_stages.py > $O/stages_full_xs.txt 2>&1
grep -v amdgpu.ids $O/stages_full_xs.txt | cut -c1-260
(echo -n "XSTEAL=1 SPIN=0  "; EZRT_XSTEAL=1 EZRT_XSTEAL_SPIN=0 timeout 60 python tools/config_one.py C2 2>&1 | grep -v amdgpu.ids
echo -n "XSTEAL=0  "; EZRT_XSTEAL=0 timeout 60 python tools/config_one.py C2 2>&1 | grep -v amdgpu.ids) | tee $O/ab.txt
