#!/bin/bash
O=gpurun_out/r5m; mkdir -p $O
run() { c=$1; shift; echo -n "$c $* | "; env "$@" timeout 100 python tools/config_one.py $c 2>&1 | grep -v amdgpu.ids; }
(for r in 1 2; do for v in 24 23 22; do run C2 EZRT_CHUNK_LOG2=$v SYNC_EACH=1; done; done
for v in 24 23 22; do run C2 EZRT_CHUNK_LOG2=$v; done
for v in 24 23; do run C2 EZRT_CHUNK_LOG2=$v EZRT_PIPELINE_DEPTH=3 SYNC_EACH=1; done
) 2>&1 | tee $O/ab.txt
