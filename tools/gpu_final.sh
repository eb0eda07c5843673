#!/bin/bash
# tools/gpu_final.sh: the round's kept evidence -- bench lines, kernel stats + PMC passes of the bench command, per-config rates,
# build timings, the 2-rank bench line over gloo.  Everything lands under gpurun_out/final/ and is copied to profiles/r3/ by hand.
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/final; mkdir -p $O; cd $R
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_n1.json 2> $O/bench_n1.err
python tools/config_rates.py > $O/config_rates.json 2> $O/config_rates.err
python tools/build_times.py > $O/build_times.json 2> $O/build_times.err
EZRT_BENCH_BACKEND=gloo MASTER_ADDR=127.0.0.1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29655 bench.py --gpus 2 --steps 5 --warmup 2 --windows 5 --cpu-seconds 0 > $O/bench_n2_gloo_one_gpu.json 2> $O/bench_n2.err
bash tools/profile.sh r3 > $O/profile.log 2>&1
python tools/summarize_profile.py gpurun_out/prof_r3 profiles/r3/final > $O/summarize.log 2>&1; cp profiles/r3/final_kernel_stats.csv profiles/r3/final_pmc.json profiles/r3/pmc_summary.json $O/
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_n1_with_profile.json 2> $O/bench_n1b.err
tail -3 $O/summarize.log; python - <<'PY'
import json
for f in ("bench_n1.json", "bench_n1_with_profile.json", "bench_n2_gloo_one_gpu.json"):
    try:
        d = json.load(open("gpurun_out/final/" + f))
        print(f, d["value"], d["ms_per_step"], d["scaling"], d.get("roofline", {}).get("frac"), d["timing"]["window_ms"])
    except Exception as e:
        print(f, "unreadable", e)
PY
