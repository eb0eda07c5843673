#!/bin/bash
# tools/ab.sh lib1 lib2 ...: bench A/B over prebuilt library variants (EZRT_HIP_LIB override)
for rep in 1 2; do
  for l in "$@"; do
    echo -n "$l : "
    EZRT_HIP_LIB=$PWD/$l python bench.py --cpu-seconds 0 --extras 0 --steps 12 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print(d['value'], d['ms_per_step'], d['config']['median_gpu_ms_per_step'], d['roofline']['trace_ms_per_step'])"
  done
done
