#!/bin/bash
O=gpurun_out/r5j; mkdir -p $O
for d in 3 2 3 2; do EZRT_PIPELINE_DEPTH=$d timeout 200 python bench.py --extras 0 --model 0 --configs "" --cpu-seconds 0 > $O/bench_d$d.json 2>>$O/bench.err
python - <<P
import json
d=json.loads(open('$O/bench_d$d.json').read().strip().splitlines()[-1])
print('depth $d', d['value'], d['ms_per_step'], d.get('value_lone_call'), d['timing']['window_ms'])
P
done
timeout 500 python -m pytest tests -m gpu -x -q > $O/suite.txt 2>&1; tail -4 $O/suite.txt
