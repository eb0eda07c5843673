for r in 1 2; do
for c in "C3 128" "C5 64" "C4 256"; do
for e in "EZRT_PIPELINE_CALLS=0" "EZRT_PIPELINE_CALLS=2 EZRT_STATIC_PCT=0" "EZRT_PIPELINE_CALLS=0 EZRT_STATIC_PCT=0"; do
  echo -n "[$e]  "; env $e python tools/config_one.py $c 2>&1 | grep -v amdgpu
done
done
done
