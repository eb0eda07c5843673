#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/s3; mkdir -p $O; cd $R
for c in C2 C3 C5; do timeout 600 python tools/prune_ab.py $c --modes 0,1,2 --scales 100,1 > $O/prune_$c.jsonl 2> $O/prune_$c.err; cut -c1-200 $O/prune_$c.jsonl; done
for m in 0 2; do EZRT_PRUNE=$m EZRT_DEBUG_STAGES=2 timeout 600 python tools/config_one.py C5 4 > $O/stages_c5_p$m.log 2>&1; done
grep -E "stage [0-9]: paths|iterations|waves" $O/stages_c5_p0.log | head -40
echo ======
grep -E "stage [0-9]: paths|iterations|waves" $O/stages_c5_p2.log | head -40
