#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/s10; mkdir -p $O; cd $R
for g in 0 1; do for rep in 1 2; do EZRT_GEN_PRIMARY=$g python bench.py --steps 20 --warmup 5 --windows 5 --extras 0 --cpu-seconds 0 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('gen_primary=$g', d['value'], d['ms_per_step'], d['config']['median_gpu_ms_per_step'], d['roofline']['trace_ms_per_step'])"; done; done
for g in 0 1; do EZRT_GEN_PRIMARY=$g timeout 300 python tools/config_one.py C4; EZRT_GEN_PRIMARY=$g timeout 300 python tools/config_one.py C3;  done 2>&1 | grep -v amdgpu.ids
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; tail -5 $O/pytest.log
