"""Throughput of the other BASELINE.json configs (C3, C4, C5 at reduced spp) on one GPU: rays/s from the
kernels' ray counter and hipEvent time of ezrt_render_device (frame buffer resident)."""
import sys, time
sys.path.insert(0, '.')
import numpy as np, torch
from ezrt_amd import scene as S, scenes, trace
hip = trace.hip()
def run(name, built, cfg, spp):
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
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 3
    rays = sc.counters()["rays"] / 3
    print("%s: %d tris, %dx%d, integrator %d, %d bounces, %d spp: %.1f ms, %.0f Mrays/s (%.1f M rays)" % (
        name, built.tri.shape[0], cfg["width"], cfg["height"], cfg["integrator"], cfg["max_bounce"], spp, dt * 1e3, rays / dt / 1e6, rays / 1e6))
C = scenes.CONFIGS
run("C2", scenes.bunny_scene(subdiv=2), C["C2"], 64)
run("C3", scenes.disney_grid_scene(subdiv=3), C["C3"], 16)
run("C4", scenes.p5_scene(subdiv=2), C["C4"], 16)
run("C5", scenes.mega_scene(), C["C5"], 4)
