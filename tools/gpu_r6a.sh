#!/bin/bash
# round 6, GPU call A: RCCL floor on one GPU; the path kernel on the 1/8 shards of C2
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r6a; mkdir -p $O; cd $R
timeout 300 python tools/exp_rccl_floor.py > $O/rccl_floor.json 2> $O/rccl_floor.err; tail -c 1500 $O/rccl_floor.json; tail -3 $O/rccl_floor.err
timeout 400 python tools/exp_shard_path_stage.py 8 > $O/shard8_path_stage.txt 2>&1; cat $O/shard8_path_stage.txt | grep -v amdgpu.ids
timeout 300 python tools/exp_shard_path_stage.py 4 > $O/shard4_path_stage.txt 2>&1; cat $O/shard4_path_stage.txt | grep -v amdgpu.ids
