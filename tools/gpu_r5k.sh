#!/bin/bash
O=gpurun_out/r5k; mkdir -p $O
run() { c=$1; shift; echo -n "$c $* | "; env "$@" timeout 100 python tools/config_one.py $c 2>&1 | grep -v amdgpu.ids; }
(for r in 1 2; do
for v in 24 12 16 20 32; do run C2 EZRT_REFILL_MIN=$v; done
for v in 12 8 16 20; do run C2 EZRT_LEAF_THRESHOLD=$v; done
for v in 40 32 48 56; do run C2 EZRT_REFILL_MIN_REL=$v; done
done
for v in 24 16 32; do run C3 EZRT_REFILL_MIN=$v; done
for v in 12 8 16; do run C3 EZRT_LEAF_THRESHOLD=$v; done
for v in 24 16 32; do run C4 EZRT_REFILL_MIN=$v; done
) 2>&1 | tee $O/ab.txt
