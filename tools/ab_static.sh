bash tools/ab_bench.sh "EZRT_STATIC_PCT=50" "EZRT_STATIC_PCT=25" "EZRT_STATIC_PCT=0"
for r in 1 2; do
for e in "EZRT_PIPELINE_CALLS=0" "EZRT_PIPELINE_CALLS=2" "EZRT_PIPELINE_CALLS=2 EZRT_STATIC_PCT=0" "EZRT_PIPELINE_CALLS=2 EZRT_STATIC_PCT=25"; do
  echo -n "[$e]  "; env $e python tools/config_one.py C4 256 2>&1 | grep -v amdgpu
done
done
