#!/bin/bash
# tools/ab_bench_configs.sh "ENV1" "ENV2" ...: bench.py's `configs` block (C3 / C4 / C5 at their BASELINE spp, calls bracketed by synchronisations) under each environment, two interleaved rounds
for r in 1 2; do
  for e in "$@"; do
    echo -n "[$e]  "; env $e python bench.py --steps 5 --warmup 2 --windows 3 --cpu-seconds 0 --extras 0 --model 0 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print(d['value'], {k:(v['Mrays_s'], v['ms_per_frame']) for k,v in d['configs'].items()})"
  done
done
