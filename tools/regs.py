#!/usr/bin/env python3
"""tools/regs.py [pattern]: compile ezrt_launch.hip with -Rpass-analysis=kernel-resource-usage and print VGPRs / spills /
occupancy of the kernels whose demangled name contains `pattern` (default: traceq4)."""
import re, subprocess, sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pat = sys.argv[1] if len(sys.argv) > 1 else "traceq4"
cmd = "/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -fno-gpu-flush-denormals-to-zero " \
      "-fno-slp-vectorize -Iinclude -Iezrt_amd/csrc/hip -c ezrt_amd/csrc/hip/ezrt_launch.hip -o /tmp/regs.o -Rpass-analysis=kernel-resource-usage"
txt = subprocess.run(cmd.split(), cwd=ROOT, capture_output=True, text=True).stderr
for b in re.split(r"remark: [^\n]*Function Name: ", txt)[1:]:
    name = b.split("\n")[0].split()[0].strip()
    dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
    if pat not in dem:
        continue
    g = lambda k: (re.search(k + r": (\d+)", b) or [None, "?"])[1]
    print("%-70s VGPR %s spill %s SGPR %s scratch %s occ %s" % (dem[:70], g("VGPRs"), g("VGPRs Spill"), g("SGPRs"),
                                                              g(r"ScratchSize \[bytes/lane\]"), g(r"Occupancy \[waves/SIMD\]")))
