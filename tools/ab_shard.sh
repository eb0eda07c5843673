#!/bin/bash
# tools/ab_shard.sh KNOB "V1 V2 ..." CFG N: tools/shard_one.py CFG 0 N under EZRT_<KNOB>=V (and the whole frame, N = 1, for comparison)
KNOB=$1; VALS=$2; CFG=$3; N=$4
for v in $VALS; do
  echo -n "$KNOB=$v  "; env EZRT_$KNOB=$v python tools/shard_one.py $CFG 0 $N 2>&1 | grep -v amdgpu.ids
  echo -n "$KNOB=$v  "; env EZRT_$KNOB=$v python tools/shard_one.py $CFG 0 1 2>&1 | grep -v amdgpu.ids
done
