import os, sys, time
sys.path.insert(0, os.getcwd())
t0 = time.perf_counter()
import torch
torch.zeros(1, device="cuda"); torch.cuda.synchronize()
t1 = time.perf_counter()
from ezrt_amd import scene as S, scenes, trace
hip = trace.hip()
bs = scenes.bunny_scene(subdiv=2, hdr="shipped")
t2 = time.perf_counter()
sc = bs.upload(hip); torch.cuda.synchronize()
t3 = time.perf_counter()
eye, cam = S.camera(0, 0, 4)
p = trace.make_params(512, 512, eye, cam, 50, 4, spp=64)
acc = torch.zeros((512, 512, 4), dtype=torch.float32, device="cuda")
st = torch.cuda.current_stream().cuda_stream
ts = []
for k in range(4):
    a = time.perf_counter(); sc.render_device(p, acc.data_ptr(), st); torch.cuda.synchronize(); ts.append(time.perf_counter() - a)
print("torch init %.2f s | scene build (host) %.3f s | ezrt_scene_create %.3f s | render calls: %s s" % (t1 - t0, t2 - t1, t3 - t2, " ".join("%.4f" % x for x in ts)))
