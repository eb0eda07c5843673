#!/bin/bash
# tools/profile_configs.sh [CFG ...]: rocprofv3 evidence for BASELINE configs[2..4] at their stated spp (VERDICT r3 #1c) -- one
# --kernel-trace --stats run and four --pmc passes (each its own run, --kernel-trace only) of tools/config_one.py per config;
# tools/summarize_config_profile.py turns them into profiles/r4/<cfg>_kernel_stats.csv and <cfg>_pmc_summary.json.
R=${GRAFT_REPO_ROOT:-/root/repo}
CFGS=${@:-C3 C4 C5}
declare -A SPP=([C2]=64 [C3]=128 [C4]=256 [C5]=512)
cd /tmp && export TMPDIR=/tmp
for c in $CFGS; do
  O=$R/gpurun_out/profcfg_$c; mkdir -p $O
  CMD="python $R/tools/config_one.py $c ${SPP[$c]}"
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o s -- $CMD > $O/stats.log 2>&1
  i=0
  for PMC in "FETCH_SIZE WRITE_SIZE" \
             "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" \
             "SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_LDS SQ_THREAD_CYCLES_VALU SQ_LDS_BANK_CONFLICT" \
             "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum"; do
    i=$((i+1))
    timeout 300 rocprofv3 --pmc $PMC --kernel-trace --output-format csv -d $O/pmc$i -o p -- $CMD > $O/pmc$i.log 2>&1 || echo "$c: pmc pass $i failed: $PMC"
  done
  grep -h "Mrays/s" $O/stats.log | tail -1
  python3 $R/tools/summarize_config_profile.py $c $O $R/gpurun_out/profcfg_summary
done
