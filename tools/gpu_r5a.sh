#!/bin/bash
# round 5, call A: suite on the ADVICE fixes + the tails of C2's trace launches (per-wave log), full frame and 1/8 shard + a bench line
O=gpurun_out/r5a; mkdir -p $O
timeout 600 python -m pytest tests -m gpu -x -q > $O/suite.txt 2>&1; tail -3 $O/suite.txt
EZRT_DEBUG_STAGES=2 timeout 120 python tools/debug_stages.py > $O/stages_full.txt 2>&1
EZRT_DEBUG_STAGES=2 timeout 120 python tools/debug_stages.py 0 8 > $O/stages_shard8.txt 2>&1
timeout 300 python bench.py > $O/bench.json 2> $O/bench.err; tail -c 600 $O/bench.json
