#!/bin/bash
O=gpurun_out/r5e; mkdir -p $O
EZRT_XSTEAL=1 EZRT_DEBUG_STAGES=2 timeout 60 python tools/debug_stages.py > $O/stages_full_xs.txt 2>&1
grep -v amdgpu.ids $O/stages_full_xs.txt | grep -v "paths_in\|re-traced\|iterations\|refill block" | cut -c1-250
(for v in 1 0 1 0; do echo -n "XSTEAL=$v  "; EZRT_XSTEAL=$v timeout 60 python tools/config_one.py C2 2>&1 | grep -v amdgpu.ids; done
for v in 1 0; do echo -n "XSTEAL=$v SYNC_EACH "; SYNC_EACH=1 EZRT_XSTEAL=$v timeout 60 python tools/config_one.py C2 2>&1 | grep -v amdgpu.ids; done
for v in 1 0; do echo -n "XSTEAL=$v  "; EZRT_XSTEAL=$v timeout 60 python tools/shard_one.py C2 0 8 2>&1 | grep -v amdgpu.ids; done
for v in 1 0; do echo -n "XSTEAL=$v PIPELINE_CALLS=0 "; EZRT_PIPELINE_CALLS=0 EZRT_XSTEAL=$v timeout 60 python tools/shard_one.py C2 0 8 2>&1 | grep -v amdgpu.ids; done) | tee $O/ab.txt
