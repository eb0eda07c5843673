#!/bin/bash
# tools/ab_libs_bench.sh NAME ...: bench.py headline under the default library and each A/B build (ezrt_amd/lib/ab/libezrt_hip_NAME.so), interleaved, two rounds; then config_one of C3 C4 C5
for r in 1 2; do
  for n in default "$@"; do
    L=""; [ "$n" != default ] && L="EZRT_HIP_LIB=$PWD/ezrt_amd/lib/ab/libezrt_hip_$n.so"
    echo -n "[$n]  "; env $L python bench.py --steps 20 --warmup 5 --cpu-seconds 0 --extras 0 --configs none --model 0 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print(d['value'], d['ms_per_step'], d['roofline']['trace_ms_per_step'])"
  done
done
for c in C3 C4 C5; do
  for n in default "$@"; do
    L=""; [ "$n" != default ] && L="EZRT_HIP_LIB=$PWD/ezrt_amd/lib/ab/libezrt_hip_$n.so"
    echo -n "[$n]  "; env $L python tools/config_one.py $c 2>&1 | grep -v amdgpu
  done
done
