#!/bin/bash
# negative control of tests/test_gpu_prune.py: a library built with -DEZRT_PRUNE_NEGATIVE_CONTROL (a pruning margin that is
# negative by 1e-3 of the distance) must FAIL the adversarial tests; the product library passes them.
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/negctl; mkdir -p $O; cd $R
EZRT_HIP_LIB=$R/build_ab/libezrt_hip_negctl.so timeout 600 python -m pytest tests/test_gpu_prune.py -m gpu -q > $O/negctl.log 2>&1; echo "negative control rc $? (expected: failures)"; grep -E "passed|failed|differ" $O/negctl.log | tail -8
timeout 600 python -m pytest tests/test_gpu_prune.py -m gpu -q > $O/product.log 2>&1; echo "product rc $?"; tail -2 $O/product.log
