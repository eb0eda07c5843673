"""tools/exp_two_streams.py [CFG]: what could overlapping one call's latency-bound late stages with the next call's bulk work buy?
Two scenes (the same arrays, each with its own scratch) render the same frame on two streams at the same time; the aggregate rate is
compared with one scene on one stream.  An upper bound for any cross-call pipelining inside the library."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ezrt_amd import scene as S, scenes, trace
hip = trace.hip()
name = sys.argv[1] if len(sys.argv) > 1 else "C2"
cfg = scenes.CONFIGS[name]
spp = {"C2": 64, "C3": 64, "C4": 64, "C5": 16}[name]
built = {"C2": lambda: scenes.bunny_scene(subdiv=2, hdr="shipped"), "C3": lambda: scenes.disney_grid_scene(subdiv=3, hdr="shipped"),
         "C4": lambda: scenes.p5_scene(subdiv=2, hdr="shipped"), "C5": lambda: scenes.mega_scene(hdr="shipped")}[name]()
eye, cam = S.camera(*cfg["camera"])
p = trace.make_params(cfg["width"], cfg["height"], eye, cam, cfg["integrator"], cfg["max_bounce"], spp=spp)
scs = [built.upload(hip) for _ in range(2)]
accs = [torch.zeros((cfg["height"], cfg["width"], 4), dtype=torch.float32, device="cuda") for _ in range(2)]
sts = [torch.cuda.Stream() for _ in range(2)]
K = 20
def run(n_streams):
    for k in range(n_streams):
        scs[k].render_device(p, accs[k].data_ptr(), sts[k].cuda_stream)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(K):
        for k in range(n_streams):
            scs[k].render_device(p, accs[k].data_ptr(), sts[k].cuda_stream)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / (K * n_streams) * 1e3
for rep in range(2):
    a, b = run(1), run(2)
    print("%s spp %d: one stream %.3f ms/call; two scenes on two streams %.3f ms/call aggregate (%+.1f %%)" % (name, spp, a, b, (a / b - 1) * 100))
assert torch.equal(accs[0].view(torch.int32), accs[1].view(torch.int32))
