#!/bin/bash
O=gpurun_out/r5f; mkdir -p $O
run() { echo -n "$* | "; env "$@" timeout 60 python tools/config_one.py C2 2>&1 | grep -v amdgpu.ids; }
shard() { echo -n "$* | "; env "$@" timeout 60 python tools/shard_one.py C2 0 8 2>&1 | grep -v amdgpu.ids; }
(run EZRT_XSTEAL=0
for mi in 64 48 32; do for st in 8 32; do run EZRT_XSTEAL=1 EZRT_XSTEAL_MIN_IDLE=$mi EZRT_XSTEAL_STOCK=$st; done; done
run EZRT_XSTEAL=1 EZRT_XSTEAL_MIN_IDLE=64 EZRT_XSTEAL_STOCK=16 EZRT_XSTEAL_GROUPS=64
run EZRT_XSTEAL=0
shard EZRT_XSTEAL=0
for mi in 64 48 32; do for st in 8 32; do shard EZRT_XSTEAL=1 EZRT_XSTEAL_MIN_IDLE=$mi EZRT_XSTEAL_STOCK=$st; done; done
shard EZRT_XSTEAL=1 EZRT_XSTEAL_MIN_IDLE=64 EZRT_XSTEAL_STOCK=16 EZRT_XSTEAL_GROUPS=64
shard EZRT_XSTEAL=0 EZRT_PIPELINE_CALLS=0
shard EZRT_XSTEAL=1 EZRT_XSTEAL_MIN_IDLE=64 EZRT_XSTEAL_STOCK=16 EZRT_PIPELINE_CALLS=0
shard EZRT_XSTEAL=1 EZRT_XSTEAL_MIN_IDLE=48 EZRT_XSTEAL_STOCK=16 EZRT_PIPELINE_CALLS=0
) 2>&1 | tee $O/ab.txt
EZRT_XSTEAL=1 EZRT_DEBUG_STAGES=2 timeout 60 python tools/debug_stages.py > $O/stages_full_xs.txt 2>&1
EZRT_XSTEAL=1 EZRT_DEBUG_STAGES=2 timeout 60 python tools/debug_stages.py 0 8 > $O/stages_shard_xs.txt 2>&1
