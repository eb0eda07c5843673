#!/bin/bash
# tools/pmc_quick.sh <tag> [env assignments...] : two rocprofv3 --pmc passes (SQ instruction mix, SQ cycles) of the
# default bench (BENCH_ARGS="--workload c4 ..." for another), per-stage means for the trace kernels.  Run on the GPU box via gpurun.
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/pmcq_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for PMC in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU" \
           "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT"; do
  i=$((i+1))
  env "$@" timeout 150 rocprofv3 --pmc $PMC --kernel-trace --output-format csv -d $OUT/p$i -o p -- python $R/bench.py --steps 2 --warmup 1 --cpu-seconds 0 $BENCH_ARGS > $OUT/log$i.txt 2>&1 || tail -3 $OUT/log$i.txt
done
python3 - <<PY
import csv, collections, glob
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob("$OUT/p*/*counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name']
        if 'traceq4_kernel' in k: k = 'traceq4<' + k.split('traceq4_kernel<')[1].split('>')[0] + '>'
        elif 'traceq_kernel<false' in k: k = 'traceq'
        elif 'shade_hit' in k or 'shade_miss' in k: k = k.split('ezd::')[1].split('(')[0]
        else: continue
        agg[k][r['Counter_Name']].append(float(r['Counter_Value']))
for k, cs in sorted(agg.items()):
    print(k, "dispatches", len(next(iter(cs.values()))))
    for c, v in cs.items():
        print("   %-26s total/step %.4g   per dispatch: %s" % (c, sum(v) / 3.0, " ".join("%.3g" % x for x in v[:12])))
PY
