#!/bin/bash
# tools/gpu_suite_bench.sh TAG [bench args]: the whole -m gpu suite + the driver's bench command, outputs under gpurun_out/TAG/
TAG=${1:-run}; shift
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/$TAG; mkdir -p $O; cd $R
(time timeout 1500 python -m pytest tests -m gpu -x -q) > $O/pytest_gpu.log 2>&1; tail -5 $O/pytest_gpu.log
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 "$@" > $O/bench.json 2> $O/bench.err; tail -2 $O/bench.err
python - <<PY
import json
try:
    d = json.loads(open("$O/bench.json").read().strip().splitlines()[-1])
    print("value", d["value"], "ms", d["ms_per_step"], "lone", d.get("value_lone_call"), d.get("ms_per_step_lone_call"), "trace_ms", d.get("roofline", {}).get("trace_ms_per_step"))
    print({k: (v.get("Mrays_s"), v.get("first_call_s"), v.get("crop_vs_oracle", {}).get("bit_identical")) for k, v in d.get("configs", {}).items()})
    print("parity", d.get("parity"))
except Exception as e:
    print("bench unreadable", e)
PY
