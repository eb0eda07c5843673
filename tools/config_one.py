"""tools/config_one.py C5 [spp]: one BASELINE config on one GPU, rays/s (knobs through EZRT_* env variables)."""
import sys, time
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ezrt_amd import scene as S, scenes, trace
hip = trace.hip()
name = sys.argv[1]
cfg = scenes.CONFIGS[name]
spp = int(sys.argv[2]) if len(sys.argv) > 2 else {"C2": 64, "C3": 64, "C4": 64, "C5": 16}[name]
built = {"C2": lambda: scenes.bunny_scene(subdiv=2, hdr="shipped"), "C3": lambda: scenes.disney_grid_scene(subdiv=3, hdr="shipped"),
         "C4": lambda: scenes.p5_scene(subdiv=2, hdr="shipped"), "C5": lambda: scenes.mega_scene(hdr="shipped")}[name]()
sc = built.upload(hip)
eye, cam = S.camera(*cfg["camera"])
p = trace.make_params(cfg["width"], cfg["height"], eye, cam, cfg["integrator"], cfg["max_bounce"], spp=spp)
acc = torch.zeros((cfg["height"], cfg["width"], 4), dtype=torch.float32, device="cuda")
st = torch.cuda.current_stream().cuda_stream
sc.render_device(p, acc.data_ptr(), st); torch.cuda.synchronize()
sc.counters_reset()
t0 = time.perf_counter()
for _ in range(3):
    sc.render_device(p, acc.data_ptr(), st)
    if os.environ.get("SYNC_EACH"):   # (a host that synchronises after every call: no overlap across the call boundary)
        torch.cuda.synchronize()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 3
rays = sc.counters()["rays"] / 3
sc.set_option("launch_events", 1); sc.render_device(p, acc.data_ptr(), st)   # (untimed: per-launch events on)
ms = sc.last_render_ms()
print("%s spp %d: %.2f ms/call  %.0f Mrays/s  trace launches %.2f ms of %.2f" % (name, spp, dt * 1e3, rays / dt / 1e6, ms[1], ms[0]))
