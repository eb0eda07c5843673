#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/retree; mkdir -p $O; cd $R
for c in C2 C4 C3 C5; do for r in 0 1; do EZRT_RETREE=$r timeout 600 python tools/config_one.py $c 2>&1 | grep -v amdgpu | sed "s/^/retree=$r /"; done; done
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_timed_kernel_audit.py tests/test_gpu_configs.py tests/test_fsh_golden.py tests/test_gpu_prune.py -m gpu -x -q > $O/pytest.log 2>&1; tail -5 $O/pytest.log
