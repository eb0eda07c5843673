"""tools/build_times.py: scene-build timings on one GPU box -- the host builder (buildBVHwithSAH, C++), the GPU twin with
bit-identical output (ezrt_build_sah), the GPU median builder and the GPU linear BVH -- on the C2 (79 820 triangles) and C5
(10^6 triangles) triangle sets.  One JSON object; kept as profiles/rN/build_times.json."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch  # noqa: F401
from ezrt_amd import build, scene as S, scenes, trace
hip = trace.hip()
out = {"tool": "tools/build_times.py", "device": torch.cuda.get_device_name(0), "host_cores": os.cpu_count(), "sets": []}


def bench(name, make):
    t0 = time.perf_counter()
    built = make(gpu_build=False) if name == "C5" else make()
    host_s = time.perf_counter() - t0          # scene assembly + host SAH + encode + env cache
    tri = built.tri
    r = {"set": name, "triangles": int(tri.shape[0]), "nodes_host": int(built.nodes.shape[0]),
         "scene_function_total_s_with_host_sah": round(host_s, 3), "host_build_stats": {k: int(v) for k, v in built.build_stats.items() if isinstance(v, (int, np.integer))}}
    # host builder alone on the same (already ordered) triangles
    hs = S.HostScene(); hs.addTriangles(tri)
    t0 = time.perf_counter(); hs.buildBVHwithSAH(8); r["host_sah_s"] = round(time.perf_counter() - t0, 3)
    for fn in ("build_sah", "build_median", "build_lbvh"):
        getattr(build, fn)(tri, 8)                       # warm-up (allocations, code objects)
        ms = [getattr(build, fn)(tri, 8)[2] for _ in range(3)]
        t2, n2, _ = getattr(build, fn)(tri, 8)
        r["gpu_%s_ms" % fn[6:]] = [round(x, 3) for x in ms]
        r["gpu_%s_nodes" % fn[6:]] = int(n2.shape[0])
    t2, n2, _ = build.build_sah(tri, 8)
    th, nh = hs.encode()
    r["gpu_sah_equals_host_sah_bits"] = bool(np.array_equal(t2.view(np.uint32), th.view(np.uint32)) and np.array_equal(n2.view(np.uint32), nh.view(np.uint32)))
    # the default path of round 4 (GPU SAH from 10^5 triangles on) and ezrt_scene_create on 1 host thread and on the default number
    t0 = time.perf_counter()
    auto = make(gpu_build=None) if name == "C5" else make()
    r["scene_function_total_s_default_builder"] = round(time.perf_counter() - t0, 3)
    r["default_builder"] = "gpu" if "gpu_build_ms" in auto.build_stats else "host"
    for label, env in (("one_thread", "1"), ("default_threads", None)):
        if env:
            os.environ["EZRT_HOST_THREADS"] = env
        ts = []
        for _ in range(3):
            t0 = time.perf_counter(); sc = auto.upload(hip); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0); sc.close()
        os.environ.pop("EZRT_HOST_THREADS", None)
        r["scene_create_s_" + label] = round(sorted(ts)[1], 3)
    out["sets"].append(r)
    print(json.dumps(r), file=sys.stderr)


bench("C2", lambda: scenes.bunny_scene(subdiv=2, hdr="shipped"))
bench("C5", lambda gpu_build=False: scenes.mega_scene(hdr="shipped", gpu_build=gpu_build))
print(json.dumps(out, indent=1))
