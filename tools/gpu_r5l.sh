#!/bin/bash
O=gpurun_out/r5l; mkdir -p $O
run() { c=$1; shift; echo -n "$c $* | "; env "$@" timeout 100 python tools/config_one.py $c 2>&1 | grep -v amdgpu.ids; }
(for r in 1 2; do for c in C2 C3 C4 C5; do for v in 0 1; do run $c EZRT_STEAL_BOUND=$v; done; done; done
for v in 0 1; do echo -n "shard EZRT_STEAL_BOUND=$v | "; EZRT_STEAL_BOUND=$v timeout 60 python tools/shard_one.py C2 0 8 2>&1 | grep -v amdgpu.ids; done
) 2>&1 | tee $O/ab.txt
