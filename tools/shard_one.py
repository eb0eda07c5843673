"""tools/shard_one.py CFG R N [spp]: shard R of N of a BASELINE config rendered alone, 6 calls (for a rocprofv3 kernel trace of an N-GPU split's critical path)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ezrt_amd import scene as S, scenes, trace
hip = trace.hip()
name, r, n = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
cfg = scenes.CONFIGS[name]
spp = int(sys.argv[4]) if len(sys.argv) > 4 else cfg["spp"]
built = {"C2": lambda: scenes.bunny_scene(subdiv=2, hdr="shipped"), "C4": lambda: scenes.p5_scene(subdiv=2, hdr="shipped")}[name]()
sc = built.upload(hip)
eye, cam = S.camera(*cfg["camera"])
p = trace.make_params(cfg["width"], cfg["height"], eye, cam, cfg["integrator"], cfg["max_bounce"], spp=spp, tile=(16, 16), shard=(r, n))
acc = torch.zeros((cfg["height"], cfg["width"], 4), dtype=torch.float32, device="cuda")
st = torch.cuda.current_stream().cuda_stream
for _ in range(2):
    sc.render_device(p, acc.data_ptr(), st)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(4):
    sc.render_device(p, acc.data_ptr(), st)
torch.cuda.synchronize()
print("%s shard %d/%d spp %d: %.3f ms/call" % (name, r, n, spp, (time.perf_counter() - t0) / 4 * 1e3))
