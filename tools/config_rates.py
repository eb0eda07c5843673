"""Throughput of the BASELINE.json configs on one GPU (C2 at full size; C3, C4, C5 at their full resolution and a
reduced spp that still fills a whole chunk of 2^26 pixel-samples): rays/s from the kernels' ray counter and hipEvent time of
ezrt_render_device (frame buffer resident), plus the two f4 variants (integrator 52, sixteen Sobol dimensions).
Prints one JSON object; `python tools/config_rates.py > gpurun_out/config_rates.json`, kept under profiles/rN/."""
import json
import sys
import time

sys.path.insert(0, '.')
import numpy as np  # noqa: F401
import torch

from ezrt_amd import scene as S, scenes, trace

hip = trace.hip()
out = {"tool": "tools/config_rates.py", "device": torch.cuda.get_device_name(0), "runs": []}


def run(name, built, cfg, spp, integrator=None, sobol_dims=8, build_s=None):
    sc = built.upload(hip)
    if sobol_dims != 8:
        sc.set_sampler(sobol_dims)
    integ = integrator or cfg["integrator"]
    eye, cam = S.camera(*cfg["camera"])
    p = trace.make_params(cfg["width"], cfg["height"], eye, cam, integ, cfg["max_bounce"], spp=spp)
    acc = torch.zeros((cfg["height"], cfg["width"], 4), dtype=torch.float32, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    sc.render_device(p, acc.data_ptr(), st)
    torch.cuda.synchronize()
    sc.counters_reset()
    reps = 3
    t0 = time.perf_counter()
    for _ in range(reps):
        sc.render_device(p, acc.data_ptr(), st)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    rays = sc.counters()["rays"] / reps
    sc.set_option("launch_events", 1)   # (untimed extra call with a timing-event pair around every trace launch)
    sc.render_device(p, acc.data_ptr(), st)
    ms = sc.last_render_ms()
    st_ = sc.stats()
    r = {"config": name, "triangles": int(built.tri.shape[0]), "nodes": int(built.nodes.shape[0]), "depth": int(st_["depth"]),
         "width": cfg["width"], "height": cfg["height"], "integrator": integ, "max_bounce": cfg["max_bounce"], "spp": spp,
         "full_spp_of_the_config": cfg["spp"], "sobol_dims": sobol_dims, "ms_per_call": round(dt * 1e3, 3),
         "Mrays_s": round(rays / dt / 1e6, 1), "Mrays_per_call": round(rays / 1e6, 2),
         "gpu_ms_last_call": round(ms[0], 3), "trace_launch_ms_last_call": round(ms[1], 3),
         "finite": bool(torch.isfinite(acc[..., :3]).all())}
    if build_s is not None:
        r["scene_build_s"] = round(build_s, 2)
    out["runs"].append(r)
    print(json.dumps(r), file=sys.stderr)
    sc.close()


C = scenes.CONFIGS


def timed(f, *a, **k):
    t = time.perf_counter()
    b = f(*a, **k)
    return b, time.perf_counter() - t


b, s = timed(scenes.bunny_scene, subdiv=2, hdr="shipped")
run("C2", b, C["C2"], 64, build_s=s)
b, s = timed(scenes.disney_grid_scene, subdiv=3, hdr="shipped", )
run("C3", b, C["C3"], 64, build_s=s)
b4, s = timed(scenes.p5_scene, subdiv=2, hdr="shipped")
run("C4", b4, C["C4"], 64, build_s=s)
run("C4 / integrator 52 (anisotropic lobe sampled)", b4, C["C4"], 64, integrator=52)
b5, s = timed(scenes.mega_scene, hdr="shipped")
run("C5 (8 Sobol dims, d & 7)", b5, C["C5"], 16, build_s=s)
run("C5 (16 Sobol dims)", b5, C["C5"], 16, sobol_dims=16)
print(json.dumps(out, indent=1))
