#!/bin/bash
O=gpurun_out/r5g; mkdir -p $O
run() { c=$1; shift; echo -n "$c $* | "; env "$@" timeout 100 python tools/config_one.py $c 2>&1 | grep -v amdgpu.ids; }
shard() { echo -n "$* | "; env "$@" timeout 60 python tools/shard_one.py C2 0 8 2>&1 | grep -v amdgpu.ids; }
(for r in 1 2; do for c in C2 C4 C3; do for v in 0 1; do run $c EZRT_HANDOVER=$v; done; done; done
for v in 0 1; do run C5 EZRT_HANDOVER=$v; done
for v in 0 1; do run C2 EZRT_HANDOVER=$v SYNC_EACH=1; done
for v in 0 1 0 1; do shard EZRT_HANDOVER=$v; done
for v in 0 1; do shard EZRT_HANDOVER=$v EZRT_PIPELINE_CALLS=0; done) 2>&1 | tee $O/ab.txt
EZRT_DEBUG_STAGES=2 timeout 60 python tools/debug_stages.py > $O/stages_full.txt 2>&1
timeout 400 python -m pytest tests -m gpu -x -q > $O/suite.txt 2>&1; tail -4 $O/suite.txt
