"""tests/check_config_parity.py C4 [spp]: one BASELINE config at its full resolution, product (libezrt_hip.so) against the CPU
oracle, bit for bit (test infrastructure: loads oracle/libezrt_oracle.so).  Since round 6 tests/test_gpu_configs.py runs this comparison in the -m gpu suite
for C2 (64 spp), C3 (128), C4 (256) -- their full BASELINE spp -- and C5 at 16 spp; this script is the same comparison for any spp, run by hand on the GPU box
(round 6: `C5 512`, the one BASELINE frame too long for the suite -- ~20 min of oracle on 16 cores: profiles/r6/full_frame_parity_c5_512spp.txt)."""
import ctypes, os, sys, time
sys.path.insert(0, '.')
import numpy as np
import torch  # noqa: F401  (before the library: same HIP runtime)
from ezrt_amd import _abi, scene as S, scenes, trace
hip = trace.hip()
oracle = trace.TraceLib(_abi.declare_trace_abi(ctypes.CDLL(os.path.join("oracle", "libezrt_oracle.so"))))
name = sys.argv[1]
spp = int(sys.argv[2]) if len(sys.argv) > 2 else 2
cfg = scenes.CONFIGS[name]
built = {"C2": lambda: scenes.bunny_scene(subdiv=2, hdr="shipped"), "C3": lambda: scenes.disney_grid_scene(subdiv=3, hdr="shipped"),
         "C4": lambda: scenes.p5_scene(subdiv=2, hdr="shipped"), "C5": lambda: scenes.mega_scene(hdr="shipped")}[name]()
eye, cam = S.camera(*cfg["camera"])
p = trace.make_params(cfg["width"], cfg["height"], eye, cam, cfg["integrator"], cfg["max_bounce"], spp=spp)
t0 = time.time(); got = built.upload(hip).render(p); t1 = time.time()
want = built.upload(oracle).render(p); t2 = time.time()
gb, wb = got.view(np.uint32), want.view(np.uint32)
diff = (gb != wb).any(axis=2)
# a NaN's payload and sign are not part of the contract (0/0 is 0xffc00000 on x86 and 0x7fc00000 on gfx950): where both sides
# are NaN in the same components the pixel counts as equal; everything else is compared on the bits
both_nan = np.isnan(got) & np.isnan(want)
same_mod_nan = ((gb == wb) | both_nan).all()
nf = int((~np.isfinite(want[..., :3]).all(axis=2)).sum())
print("%s %dx%d %d spp integrator %d: hip %.2f s, oracle %.1f s, pixels with different bits %d, identical up to NaN payloads %s, "
      "non-finite pixels (oracle) %d" % (name, cfg["width"], cfg["height"], spp, cfg["integrator"], t1 - t0, t2 - t1, int(diff.sum()),
                                         bool(same_mod_nan), nf))
if diff.any():
    ys, xs = np.nonzero(diff)
    for y, x in list(zip(ys, xs))[:5]:
        print("  pixel", x, y, "hip", got[y, x], [hex(v) for v in gb[y, x]], "oracle", want[y, x], [hex(v) for v in wb[y, x]])
sys.exit(0 if same_mod_nan else 1)
