#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/s2; mkdir -p $O; cd $R
for c in C2 C4 C3 C5; do timeout 600 python tools/prune_ab.py $c > $O/prune_$c.jsonl 2> $O/prune_$c.err; tail -4 $O/prune_$c.jsonl | cut -c1-330; tail -2 $O/prune_$c.err; done
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_timed_kernel_audit.py tests/test_gpu_configs.py tests/test_fsh_golden.py -m gpu -x -q > $O/pytest.log 2>&1; tail -5 $O/pytest.log
