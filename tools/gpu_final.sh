#!/bin/bash
# tools/gpu_final.sh [ROUND]: the round's kept evidence in one call -- the GPU test suite, the driver's bench command (with the
# per-config timings and the scaling model), kernel stats + PMC passes of the bench command (tools/profile.sh), the 2- and 8-rank
# bench lines over gloo on one GPU.  Everything lands under gpurun_out/final/ and is copied to profiles/<ROUND>/ by hand.
RND=${1:-r6}
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/final; mkdir -p $O; cd $R
(time python -m pytest tests -m gpu -x -q) > $O/pytest_gpu.log 2>&1; tail -3 $O/pytest_gpu.log
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_n1_before_profile.json 2> $O/bench_n1.err
bash tools/profile.sh $RND > $O/profile.log 2>&1
python tools/summarize_profile.py gpurun_out/prof_$RND $O/final > $O/summarize.log 2>&1; tail -3 $O/summarize.log
mkdir -p profiles/$RND; cp $O/final_kernel_stats.csv $O/final_pmc.json $O/pmc_summary.json profiles/$RND/ 2>/dev/null
# per-config kernel stats + counters (C3 / C4 / C5 at their BASELINE spp) and the VALU issue microbenchmark of this box: BEFORE the kept bench
# line, so that its per-config roofline blocks find summaries stamped with this tree's source hash
bash tools/profile_configs.sh C3 C4 C5 > $O/profile_configs.log 2>&1; tail -6 $O/profile_configs.log
cp gpurun_out/profcfg_summary/c?_kernel_stats.csv gpurun_out/profcfg_summary/c?_pmc_summary.json profiles/$RND/ 2>/dev/null
(hipcc --offload-arch=gfx950 -O2 -o /tmp/exp_valu tools/exp_valu_issue.hip && timeout 300 /tmp/exp_valu quick) > $O/valu_issue_microbench.txt 2>&1; cp $O/valu_issue_microbench.txt profiles/$RND/ 2>/dev/null
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_n1.json 2> $O/bench_n1b.err
mkdir -p gpurun_out/final/profiles_$RND; cp profiles/$RND/*pmc* profiles/$RND/*kernel_stats.csv profiles/$RND/valu_issue_microbench.txt gpurun_out/final/profiles_$RND/ 2>/dev/null
for n in 2 8; do
  EZRT_BENCH_BACKEND=gloo MASTER_ADDR=127.0.0.1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port $((29650 + n)) bench.py --gpus $n --steps 5 --warmup 2 --windows 5 --cpu-seconds 0 > $O/bench_n${n}_gloo_one_gpu.json 2> $O/bench_n$n.err
done
rm -rf gpurun_out/prof_$RND/*/  # (raw traces stay on the box)
python - <<'PY'
import json
for f in ("bench_n1_before_profile.json", "bench_n1.json", "bench_n2_gloo_one_gpu.json", "bench_n8_gloo_one_gpu.json"):
    try:
        d = json.loads(open("gpurun_out/final/" + f).read().strip().splitlines()[-1])
        print(f, d["value"], d["ms_per_step"], d["scaling"], d.get("roofline", {}).get("frac"), d["timing"]["window_ms"][:3])
    except Exception as e:
        print(f, "unreadable", e)
PY
