#!/bin/bash
# round 3, GPU session 1: the new tests, then the driver's bench case in fresh processes (stall hunt)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/s1
mkdir -p $O
cd $R
# 1. the driver's exact command FIRST: first GPU work of the lease, fresh process
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_a.json 2> $O/bench_a.err
EZRT_REDO_OVERLAP=0 python bench.py --gpus 1 --steps 20 --warmup 5 --extras 0 --cpu-seconds 0 > $O/bench_b_nooverlap.json 2> $O/bench_b.err
python bench.py --gpus 1 --steps 20 --warmup 5 --extras 0 --cpu-seconds 0 > $O/bench_c.json 2> $O/bench_c.err
# 2. tests
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
tail -5 $O/pytest.log
# 3. where a fresh process' first window goes: kernel + hip api trace (no counters)
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --hip-trace --output-format csv -d $O/trace -o t -- python $R/bench.py --gpus 1 --steps 20 --warmup 5 --windows 2 --extras 0 --cpu-seconds 0 > $O/trace.log 2>&1
ls -la $O/trace | head
for f in $O/bench_a.json $O/bench_b_nooverlap.json $O/bench_c.json; do python - "$f" <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1]))
    t = d["timing"]
    print(sys.argv[1].split("/")[-1], "value", d["value"], "ms/step", d["ms_per_step"], "windows", t["window_ms"], "wall/gpu", t["wall_over_gpu"])
    print("   first window gpu ms/step", t["first_window_gpu_ms_per_step"])
except Exception as e:
    print(sys.argv[1], "unreadable:", e)
PY
done
