#!/bin/bash
run() { env "$@" python tools/config_one.py $cfg 2>&1 | grep -v amdgpu | sed "s/^/$* /"; }
for cfg in C2 C4 C3 C5; do
  for r in 24 40; do for g in 0 1 2 4; do run EZRT_GEN_TRIES=$g EZRT_REFILL_MIN_REL=$r; done; done
done
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py tests/test_gpu_timed_kernel_audit.py -x -q -m gpu 2>&1 | tail -3
