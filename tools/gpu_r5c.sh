#!/bin/bash
# round 5, call C: cross-wave stealing after the ring-overflow fix.  Every step under a tight timeout; stop at the first failure.
O=gpurun_out/r5c; mkdir -p $O
echo "== smoke C2 xsteal=1"; EZRT_XSTEAL=1 timeout 90 python tools/config_one.py C2 2>&1 | grep -v amdgpu.ids | tee $O/smoke.txt
grep -q "Mrays/s" $O/smoke.txt || { echo "SMOKE FAILED"; exit 1; }
echo "== suite"; timeout 420 python -m pytest tests -m gpu -x -q > $O/suite.txt 2>&1; tail -5 $O/suite.txt
grep -q " passed" $O/suite.txt && ! grep -q "failed" $O/suite.txt || { echo "SUITE FAILED"; tail -40 $O/suite.txt; exit 1; }
echo "== A/B configs"
(for r in 1 2; do for c in C2 C4 C3; do for v in 0 1; do echo -n "XSTEAL=$v  "; EZRT_XSTEAL=$v timeout 120 python tools/config_one.py $c 2>&1 | grep -v amdgpu.ids; done; done; done) | tee $O/ab_configs.txt
(for v in 0 1; do echo -n "XSTEAL=$v SYNC_EACH  "; EZRT_XSTEAL=$v SYNC_EACH=1 timeout 90 python tools/config_one.py C2 2>&1 | grep -v amdgpu.ids; done) | tee -a $O/ab_configs.txt
(for v in 0 1; do echo -n "XSTEAL=$v  "; EZRT_XSTEAL=$v timeout 90 python tools/shard_one.py C2 0 8 2>&1 | grep -v amdgpu.ids; done) | tee $O/ab_shard.txt
(for v in 0 1; do echo -n "XSTEAL=$v PIPELINE_CALLS=0 "; EZRT_PIPELINE_CALLS=0 EZRT_XSTEAL=$v timeout 90 python tools/shard_one.py C2 0 8 2>&1 | grep -v amdgpu.ids; done) | tee -a $O/ab_shard.txt
EZRT_XSTEAL=1 EZRT_DEBUG_STAGES=2 timeout 90 python tools/debug_stages.py > $O/stages_full_xs.txt 2>&1
EZRT_XSTEAL=1 EZRT_DEBUG_STAGES=2 timeout 90 python tools/debug_stages.py 0 8 > $O/stages_shard8_xs.txt 2>&1
echo done
