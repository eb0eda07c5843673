"""tools/exp_stream_pressure.py N_OTHER: does the gain of pipeline_calls survive other scenes (= other HIP streams) alive in the process?
C3 at 128 spp with pipeline_calls 0 / 1 after N_OTHER other scenes were created and rendered once (and kept alive)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ezrt_amd import scene as S, scenes, trace
hip = trace.hip()
n_other = int(sys.argv[1]) if len(sys.argv) > 1 else 0
st = torch.cuda.current_stream().cuda_stream
others = []
small = scenes.bunny_scene(subdiv=0, hdr="shipped")
for k in range(n_other):
    sc = small.upload(hip)
    eye, cam = S.camera(0, 0, 4)
    acc = torch.zeros((128, 128, 4), dtype=torch.float32, device="cuda")
    sc.render_device(trace.make_params(128, 128, eye, cam, 50, 4, spp=2), acc.data_ptr(), st)
    others.append((sc, acc))
torch.cuda.synchronize()
cfg = scenes.CONFIGS["C3"]
built = scenes.disney_grid_scene(subdiv=3, hdr="shipped")
eye, cam = S.camera(*cfg["camera"])
p = trace.make_params(cfg["width"], cfg["height"], eye, cam, cfg["integrator"], cfg["max_bounce"], spp=128, tile=(16, 16))
for mode in (0, 1, 0, 1):
    sc = built.upload(hip)
    sc.set_option("pipeline_calls", mode)
    acc = torch.zeros((cfg["height"], cfg["width"], 4), dtype=torch.float32, device="cuda")
    sc.render_device(p, acc.data_ptr(), st); torch.cuda.synchronize()
    ms = []
    for _ in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        sc.render_device(p, acc.data_ptr(), st)
        torch.cuda.synchronize(); ms.append((time.perf_counter() - t0) * 1e3)
    print("other scenes alive %d | pipeline_calls %d: C3 128 spp %.2f ms/call (calls: %s)" % (n_other, mode, sorted(ms)[1], " ".join("%.2f" % x for x in ms)), flush=True)
    sc.close()
