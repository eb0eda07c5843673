#!/bin/bash
# tools/ab_knob.sh KNOB "V1 V2 ..." "CFG ..." [rounds]: tools/config_one.py of each config under EZRT_<KNOB>=V, interleaved rounds
# (a knob A/B inside one library build; on the GPU box)
KNOB=$1; VALS=$2; CFGS=$3; ROUNDS=${4:-2}
for r in $(seq $ROUNDS); do
  for c in $CFGS; do
    for v in $VALS; do
      echo -n "$KNOB=$v  "; env EZRT_$KNOB=$v python tools/config_one.py $c 2>&1 | grep -v amdgpu.ids
    done
  done
done
