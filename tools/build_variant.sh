#!/bin/bash
# tools/build_variant.sh NAME [-DFLAG ...]: an A/B build of the HIP library into ezrt_amd/lib/ab/libezrt_hip_NAME.so
# (EZRT_HIP_LIB selects it for a run; never loaded by default).
set -e
cd "$(dirname "$0")/.."
name=$1; shift
mkdir -p ezrt_amd/lib/ab
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -fno-gpu-flush-denormals-to-zero \
  -fno-slp-vectorize -Wall -Wno-unused-function -Iinclude -Iezrt_amd/csrc/hip "$@" -shared -o ezrt_amd/lib/ab/libezrt_hip_$name.so \
  ezrt_amd/csrc/hip/ezrt_hip.hip ezrt_amd/csrc/hip/ezrt_scene_build.hip ezrt_amd/csrc/hip/ezrt_launch.hip ezrt_amd/csrc/hip/ezrt_lbvh.hip ezrt_amd/csrc/hip/ezrt_sahbvh.hip ezrt_amd/csrc/hip/ezrt_mgpu.hip -ldl -pthread
