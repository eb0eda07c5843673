#!/bin/bash
# tools/ab_bench.sh "ENV1" "ENV2" ...: bench.py headline (C2, 7 windows of 20 steps, no extras) under each environment, two interleaved rounds
for r in 1 2; do
  for e in "$@"; do
    echo -n "[$e]  "; env $e python bench.py --steps 20 --warmup 5 --cpu-seconds 0 --extras 0 --configs none --model 0 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print(d['value'], d['ms_per_step'], d['roofline']['trace_ms_per_step'])"
  done
done
