for rep in 1 2; do
for lib in "" "ezrt_amd/lib/ab/libezrt_hip_heads64.so"; do
  for pm in 128 64 32 16; do
    L=""; [ -n "$lib" ] && L="EZRT_HIP_LIB=$PWD/$lib"
    echo -n "heads=$([ -n "$lib" ] && echo 64 || echo 8) pool_max=$pm  "; env $L EZRT_POOL_MAX=$pm python tools/shard_one.py C2 0 8 2>&1 | grep -v amdgpu | tr '\n' ' '; env $L EZRT_POOL_MAX=$pm python tools/shard_one.py C2 0 1 2>&1 | grep -v amdgpu
  done
done
done
