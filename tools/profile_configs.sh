#!/bin/bash
# tools/profile_configs.sh [CFG ...]: rocprofv3 evidence for BASELINE configs[2..4] at their stated spp (VERDICT r3 #1c) -- one
# --kernel-trace --stats run and four --pmc passes (FETCH_SIZE and WRITE_SIZE cannot share a pass) (each its own run, --kernel-trace only) of tools/config_one.py per config;
# tools/summarize_config_profile.py turns them into profiles/r5/<cfg>_kernel_stats.csv and <cfg>_pmc_summary.json.
R=${GRAFT_REPO_ROOT:-/root/repo}
CFGS=${@:-C3 C4 C5}
declare -A SPP=([C2]=64 [C3]=128 [C4]=256 [C5]=512)
# the counter passes serialise every dispatch: they run one chunk's worth of frames (the kernels and their per-dispatch rates
# are the same; the kernel-trace statistics above them are taken at the full BASELINE spp)
declare -A SPP_PMC=([C2]=64 [C3]=32 [C4]=32 [C5]=8)
cd /tmp && export TMPDIR=/tmp
# (as tools/profile.sh: the kernels are profiled running ALONE; consecutive chunks otherwise overlap -- pipeline_calls -- and a kernel's
# duration in the trace would include its wait for wave slots held by the other chunk's launches)
export EZRT_PIPELINE_CALLS=0
for c in $CFGS; do
  O=$R/gpurun_out/profcfg_$c; mkdir -p $O
  CMD="python $R/tools/config_one.py $c ${SPP[$c]}"
  timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o s -- $CMD > $O/stats.log 2>&1
  CMD="python $R/tools/config_one.py $c ${SPP_PMC[$c]}"
  i=0
  for PMC in "FETCH_SIZE" "WRITE_SIZE" \
             "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" \
             "SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_LDS SQ_THREAD_CYCLES_VALU SQ_LDS_BANK_CONFLICT TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" \
             "SQ_WAIT_INST_LDS"; do
    i=$((i+1))
    timeout 150 rocprofv3 --pmc $PMC --kernel-trace --output-format csv -d $O/pmc$i -o p -- $CMD > $O/pmc$i.log 2>&1 || echo "$c: pmc pass $i failed: $PMC"
  done
  grep -h "Mrays/s" $O/stats.log | tail -1
  python3 $R/tools/summarize_config_profile.py $c $O $R/gpurun_out/profcfg_summary ${SPP[$c]} ${SPP_PMC[$c]}
  du -sh $O | tail -1
  rm -rf $O   # (raw traces stay on the box: only the summaries travel back)
done
