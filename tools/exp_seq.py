"""tools/exp_seq.py C4 C4 C2 C4: the same measurement as config_one.py for a sequence of configs in ONE process."""
import sys, time
sys.path.insert(0, '.')
import torch
from ezrt_amd import scene as S, scenes, trace
hip = trace.hip()
B = {"C2": lambda: scenes.bunny_scene(subdiv=2, hdr="shipped"), "C3": lambda: scenes.disney_grid_scene(subdiv=3, hdr="shipped"),
     "C4": lambda: scenes.p5_scene(subdiv=2, hdr="shipped"), "C5": lambda: scenes.mega_scene(hdr="shipped")}
built = {}
keep = []
for name in sys.argv[1:]:
    hold = name.endswith("+")   # "C2+": keep the scene alive instead of closing it
    name = name.rstrip("+")
    cfg = scenes.CONFIGS[name]
    spp = {"C2": 64, "C3": 64, "C4": 64, "C5": 16}[name]
    if name not in built:
        built[name] = B[name]()
    sc = built[name].upload(hip)
    eye, cam = S.camera(*cfg["camera"])
    p = trace.make_params(cfg["width"], cfg["height"], eye, cam, cfg["integrator"], cfg["max_bounce"], spp=spp)
    acc = torch.zeros((cfg["height"], cfg["width"], 4), dtype=torch.float32, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    sc.render_device(p, acc.data_ptr(), st); torch.cuda.synchronize()
    sc.counters_reset()
    t0 = time.perf_counter()
    for _ in range(3):
        sc.render_device(p, acc.data_ptr(), st)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 3
    rays = sc.counters()["rays"] / 3
    sc.set_option("launch_events", 1); sc.render_device(p, acc.data_ptr(), st)
    ms = sc.last_render_ms()
    print("%s: %.2f ms/call  %.0f Mrays/s  trace %.2f of %.2f  free %.1f GB" % (name, dt * 1e3, rays / dt / 1e6, ms[1], ms[0], torch.cuda.mem_get_info()[0] / 1e9))
    if hold: keep.append(sc)
    else: sc.close()
