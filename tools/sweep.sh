#!/bin/bash
# tools/sweep.sh VAR "v1 v2 ..." [extra env assignments]: bench A/B over one env knob
VAR=$1; VALS=$2; shift 2
for v in $VALS; do
  echo -n "$VAR=$v $@ : "
  env $VAR=$v "$@" python bench.py --cpu-seconds 0 --extras 0 --steps 12 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print(d['value'], d['ms_per_step'], d['config']['median_gpu_ms_per_step'], d['roofline']['trace_ms_per_step'])"
done
