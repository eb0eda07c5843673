#!/bin/bash
O=gpurun_out/r5i; mkdir -p $O
run() { c=$1; shift; echo -n "$c $* | "; env "$@" timeout 100 python tools/config_one.py $c 2>&1 | grep -v amdgpu.ids; }
shard() { n=$1; shift; echo -n "1/$n $* | "; env "$@" timeout 60 python tools/shard_one.py C2 0 $n 2>&1 | grep -v amdgpu.ids; }
(for d in 2 3 4 2 3 4; do run C2 EZRT_PIPELINE_DEPTH=$d; done
for d in 2 3 4; do run C4 EZRT_PIPELINE_DEPTH=$d; done
for d in 2 3; do run C3 EZRT_PIPELINE_DEPTH=$d; done
for d in 2 3 4 2 3 4; do shard 8 EZRT_PIPELINE_DEPTH=$d; done
for d in 2 3 4; do shard 4 EZRT_PIPELINE_DEPTH=$d; done
for d in 2 3 4; do shard 2 EZRT_PIPELINE_DEPTH=$d; done) 2>&1 | tee $O/ab.txt
