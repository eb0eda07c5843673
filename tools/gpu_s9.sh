#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/s9; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_gpu_prune.py -m gpu -q > $O/pytest_prune.log 2>&1; tail -15 $O/pytest_prune.log
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; tail -5 $O/pytest.log
