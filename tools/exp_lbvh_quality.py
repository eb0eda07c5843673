"""Trace cost of the GPU linear BVH vs the reference SAH builder on the C2 scene (same rays)."""
import sys, time
sys.path.insert(0, '.')
import numpy as np
from ezrt_amd import scene as S, scenes, trace, build
hip = trace.hip()
bs = scenes.bunny_scene(subdiv=2)
eye, cam = S.camera(0, 0, 4)
p = trace.make_params(512, 512, eye, cam, 50, 4, spp=64)
t0 = time.time(); tri, nodes, ms = build.build_lbvh(bs.tri, 8); t1 = time.time()
print("lbvh build: device %.2f ms, call %.1f ms, %d nodes (SAH: %d nodes)" % (ms, (t1 - t0) * 1e3, nodes.shape[0], bs.nodes.shape[0]))
for name, T, N in (("sah", bs.tri, bs.nodes), ("lbvh", tri, nodes)):
    sc = hip.scene_create(T, N)
    sc.set_env(bs.hdr, bs.cache, bs.env_filter)
    sc.set_instrumentation(1); sc.counters_reset(); sc.render(p); c = sc.counters(); sc.set_instrumentation(0)
    sc.render(p)
    t = []
    for _ in range(5):
        a = time.time(); sc.render(p); t.append(time.time() - a)
    print(name, "render ms (incl. 2 PCIe copies)", round(min(t) * 1e3, 2), "counters", c)
