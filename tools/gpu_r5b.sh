#!/bin/bash
# round 5, call B: cross-wave stealing (knob xsteal): the suite with it on, then A/B on the configs, the 1/8 shard, the per-wave log
O=gpurun_out/r5b; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q > $O/suite.txt 2>&1; tail -5 $O/suite.txt
(timeout 900 bash tools/ab_knob.sh XSTEAL "0 1" "C2 C4 C3 C5" 2) > $O/ab_configs.txt 2>&1
(for v in 0 1; do echo -n "XSTEAL=$v SYNC_EACH  "; EZRT_XSTEAL=$v SYNC_EACH=1 timeout 200 python tools/config_one.py C2 2>&1 | grep -v amdgpu.ids; done) >> $O/ab_configs.txt 2>&1
(timeout 600 bash tools/ab_shard.sh XSTEAL "0 1" C2 8) > $O/ab_shard.txt 2>&1
EZRT_XSTEAL=1 EZRT_DEBUG_STAGES=2 timeout 120 python tools/debug_stages.py > $O/stages_full_xs.txt 2>&1
EZRT_XSTEAL=1 EZRT_DEBUG_STAGES=2 timeout 120 python tools/debug_stages.py 0 8 > $O/stages_shard8_xs.txt 2>&1
cat $O/ab_configs.txt $O/ab_shard.txt
