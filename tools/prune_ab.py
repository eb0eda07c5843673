"""tools/prune_ab.py C2 C3 C5 ...: the distance-pruning modes of traceq4_kernel side by side on one GPU -- rays/s per mode, the
frames compared on the bits (a schedule knob must never change them), the scene's pruning bound."""
import json, sys, time
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ezrt_amd import scene as S, scenes, trace
hip = trace.hip()
SPP = {"C2": 64, "C3": 64, "C4": 64, "C5": 16}
out = []
modes = [int(x) for x in (sys.argv[sys.argv.index("--modes") + 1].split(",") if "--modes" in sys.argv else "0,1,2".split(","))]
scales = [int(x) for x in (sys.argv[sys.argv.index("--scales") + 1].split(",") if "--scales" in sys.argv else ["100"])]
names = [a for a in sys.argv[1:] if a in SPP]
for name in names:
    cfg = scenes.CONFIGS[name]
    t0 = time.perf_counter()
    built = {"C2": lambda: scenes.bunny_scene(subdiv=2, hdr="shipped"), "C3": lambda: scenes.disney_grid_scene(subdiv=3, hdr="shipped"),
             "C4": lambda: scenes.p5_scene(subdiv=2, hdr="shipped"), "C5": lambda: scenes.mega_scene(hdr="shipped")}[name]()
    tb = time.perf_counter() - t0
    sc = built.upload(hip)
    eye, cam = S.camera(*cfg["camera"])
    spp = SPP[name]
    p = trace.make_params(cfg["width"], cfg["height"], eye, cam, cfg["integrator"], cfg["max_bounce"], spp=spp)
    st = torch.cuda.current_stream().cuda_stream
    ref = None
    for mode in modes:
        for scale in (scales if mode else [100]):
            sc.set_option("prune", mode)
            acc = torch.zeros((cfg["height"], cfg["width"], 4), dtype=torch.float32, device="cuda")
            sc.render_device(p, acc.data_ptr(), st); torch.cuda.synchronize()
            sc.counters_reset()
            t0 = time.perf_counter()
            for _ in range(3):
                sc.render_device(p, acc.data_ptr(), st)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / 3
            rays = sc.counters()["rays"] / 3
            sc.set_option("launch_events", 1); sc.render_device(p, acc.data_ptr(), st); ms = sc.last_render_ms(); sc.set_option("launch_events", 0)
            bits = acc.view(torch.int32)
            if ref is None:
                ref = bits.clone()
            nan_ok = torch.isnan(acc) & torch.isnan(ref.view(torch.float32))
            same = bool(((bits == ref) | nan_ok).all())
            r = {"config": name, "prune": mode, "scale_pct": scale, "spp": spp, "ms_per_call": round(dt * 1e3, 3), "Mrays_s": round(rays / dt / 1e6, 1),
                 "trace_ms": round(ms[1], 3), "gpu_ms": round(ms[0], 3), "same_bits_as_first_mode": same, "prune_info": sc.prune_info(), "build_s": round(tb, 1)}
            out.append(r)
            print(json.dumps(r), flush=True)
    sc.close()
