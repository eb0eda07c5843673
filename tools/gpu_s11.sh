#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/s11; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_timed_kernel_audit.py tests/test_fsh_golden.py tests/test_gpu_prune.py -m gpu -x -q > $O/pytest1.log 2>&1; tail -8 $O/pytest1.log
for g in 0 2 0 2 3; do EZRT_PATH_STAGE=$g python bench.py --steps 20 --warmup 5 --windows 5 --extras 0 --cpu-seconds 0 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('path_stage=$g', d['value'], d['ms_per_step'], d['config']['median_gpu_ms_per_step'], d['roofline']['trace_ms_per_step'])"; done
for g in 0 2; do EZRT_PATH_STAGE=$g timeout 300 python tools/config_one.py C3;  done 2>&1 | grep -v amdgpu.ids
bash tools/gpu_tl.sh path2 EZRT_PATH_STAGE=2
