#!/bin/bash
# tools/profile.sh <tag> [bench args...] -- run on the GPU box (via gpurun).  Collects, under
# gpurun_out/prof_<tag>/: kernel-trace stats of the default bench, and PMC passes (each in its own
# run, --kernel-trace only, as MI355X_MICROARCH.md "HBM" / the pool's rocprofv3 rules require).
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
# (EZRT_PIPELINE_CALLS=0: the kernels are profiled running ALONE -- consecutive steps otherwise overlap (DESIGN.md 5 "pipeline_calls") and a
# kernel's duration in the trace would include waiting for wave slots held by the other chunk's launches; bench.py's own trace_ms_per_step
# comes from a sequential pass too, and the counter passes serialise every dispatch anyway)
export EZRT_PIPELINE_CALLS=0
BENCH="python $R/bench.py --steps 2 --warmup 1 --cpu-seconds 0 --extras 0 --configs none --model 0 $@"
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o s -- $BENCH > $OUT/stats.log 2>&1
i=0
for PMC in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" \
           "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" \
           "SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_THREAD_CYCLES_VALU SQ_LDS_BANK_CONFLICT SQ_INST_CYCLES_VMEM" \
           "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum" "GRBM_GUI_ACTIVE" "SQ_WAIT_INST_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_WR SQ_INSTS_FLAT" "SQ_WAIT_INST_LDS"; do
  i=$((i+1))
  timeout 150 rocprofv3 --pmc $PMC --kernel-trace --output-format csv -d $OUT/pmc$i -o p -- $BENCH > $OUT/pmc$i.log 2>&1 || echo "pmc pass $i failed: $PMC"
done
find $OUT -name "*.csv" | head -40
