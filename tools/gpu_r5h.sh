#!/bin/bash
O=gpurun_out/r5h; mkdir -p $O
timeout 120 ./build/exp_valu quick > $O/valu_issue_microbench.txt 2>&1; cat $O/valu_issue_microbench.txt | cut -c1-120
timeout 600 python -m pytest tests -m gpu -x -q > $O/suite.txt 2>&1; tail -15 $O/suite.txt
mkdir -p profiles/r5; cp $O/valu_issue_microbench.txt profiles/r5/valu_issue_microbench.txt
timeout 300 python bench.py > $O/bench.json 2> $O/bench.err; python - <<'P'
import json
d=json.loads(open('gpurun_out/r5h/bench.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d.get('value_lone_call'), d.get('ms_per_step_lone_call'), d['roofline'].get('frac'), d['roofline'].get('peak'), d['roofline'].get('peak_kernel_opcode_mix'))
print(d.get('scaling_model',{}).get('predicted_speedup'))
print({k:(v.get('rays_s') if isinstance(v,dict) else v) for k,v in d.get('configs',{}).items()})
P
