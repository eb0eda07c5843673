#!/bin/bash
# tools/ab_env.sh "CFG ..." ROUNDS "ENV=V ENV2=W" "ENV=V2" ...: tools/config_one.py of each config under each environment set, interleaved
# rounds (round 5's A/Bs: handover, steal_bound, xsteal parameter sweeps, pipeline_depth, the knob re-sweep; on the GPU box, every run under a timeout).
# SHARD=N adds tools/shard_one.py C2 0 N under each set.
CFGS=$1; ROUNDS=$2; shift 2
for r in $(seq $ROUNDS); do
  for c in $CFGS; do
    for e in "$@"; do echo -n "$c $e | "; env $e timeout 120 python tools/config_one.py $c 2>&1 | grep -v amdgpu.ids; done
  done
done
if [ -n "$SHARD" ]; then
  for e in "$@"; do echo -n "1/$SHARD $e | "; env $e timeout 60 python tools/shard_one.py C2 0 $SHARD 2>&1 | grep -v amdgpu.ids; done
fi
