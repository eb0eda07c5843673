"""tools/exp_shard_path_stage.py [N]: the 1/N shards of C2 with the late stages as ONE launch (`path_stage` = 2) and staged (0).

VERDICT r5 #6: the path kernel lost 2x on full frames (profiles/r3/path_kernel_negative.txt) but was never timed where launch floors are
most of the critical path.  Per shard r of N: median of 3 bursts of 4 back-to-back calls (bench.py's scaling_model protocol) and a lone
call, for each setting, same process, same scene; frames compared on the bits."""
import os
import statistics
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from ezrt_amd import scene as S, scenes, trace

hip = trace.hip()
N = int(sys.argv[1]) if len(sys.argv) > 1 else 8
cfg = scenes.CONFIGS["C2"]
sc = scenes.bunny_scene(subdiv=2, hdr="shipped").upload(hip)
eye, cam = S.camera(*cfg["camera"])
W, H = cfg["width"], cfg["height"]
st = torch.cuda.current_stream().cuda_stream
acc = torch.zeros((H, W, 4), dtype=torch.float32, device="cuda")


def timed(p, burst):
    sc.render_device(p, acc.data_ptr(), st)
    torch.cuda.synchronize()
    ts = []
    for _ in range(3):
        t0 = time.perf_counter()
        for _ in range(burst):
            sc.render_device(p, acc.data_ptr(), st)
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) * 1e3 / burst)
    return statistics.median(ts)


SETTINGS = [("staged", 0), ("path_stage=2", 2), ("path_stage=3", 3)]
res = {k: {"burst": [], "lone": []} for k, _ in SETTINGS}
frames = {}
for r in range(N):
    p = trace.make_params(W, H, eye, cam, cfg["integrator"], cfg["max_bounce"], spp=cfg["spp"], tile=(16, 16), shard=(r, N))
    for k, v in SETTINGS:
        sc.set_option("path_stage", v)
        res[k]["burst"].append(timed(p, 4))
        res[k]["lone"].append(timed(p, 1))
        if r == 0:
            acc.zero_()
            sc.render_device(p, acc.data_ptr(), st)
            torch.cuda.synchronize()
            frames[k] = acc.clone()
for k, _ in SETTINGS:
    b, l = res[k]["burst"], res[k]["lone"]
    print("1/%d shards of C2, %-13s burst-of-4 ms/call: %s  critical %.4f | lone: critical %.4f mean %.4f | frame == staged: %s"
          % (N, k, " ".join("%.3f" % x for x in b), max(b), max(l), sum(l) / len(l), bool(torch.equal(frames[k].view(torch.int32), frames["staged"].view(torch.int32)))))
