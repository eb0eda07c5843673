#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/s8; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_gpu_prune.py -m gpu -x -q > $O/pytest_prune.log 2>&1; tail -15 $O/pytest_prune.log
for c in C2 C4 C3 C5; do timeout 600 python tools/prune_ab.py $c --modes 0,1,2 > $O/prune_$c.jsonl 2> $O/prune_$c.err; cut -c1-130,290-420 $O/prune_$c.jsonl; done
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; tail -5 $O/pytest.log
