"""tools/exp_rccl_floor.py: what ONE GPU can measure of the frame close's RCCL cost.

 (a) ezrt_mgpu with EZRT_TRANSPORT_RCCL on one device: pack -> grouped ncclSend/ncclRecv to ITSELF -> un-permute
     (`gather_ms` of ezrt_mgpu_last_ms: events on the root's stream), for payloads of 0.5 / 2 / 8 MiB (= the per-peer
     payloads of C2 / C4 / C5 on 8 GPUs), next to the pack and un-permute kernels timed alone (stream events);
     floor = gather - pack - unpack = launch + protocol latency of one grouped send/receive + a device-local copy.
 (b) torch.distributed (backend nccl = RCCL) at world size 1: dist.gather of the same payloads, stream events, and the
     host-side enqueue time of the call.
Prints one JSON line (copied to profiles/r6/rccl_floor.json; bench.py's scaling_model reads it)."""
import json
import os
import statistics
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
import numpy as np
import torch

from ezrt_amd import mgpu, scene as S, scenes, trace

hip = trace.hip()
dev = torch.device("cuda", 0)
st = torch.cuda.current_stream().cuda_stream
out = {"what": "RCCL floor measurable on one GPU", "payloads": {}}

bs = scenes.bunny_scene(subdiv=0, hdr="synthetic")
eye, cam = S.camera(0.0, 0.0, 4.0)
SIZES = {"0.5MiB": (256, 128), "2MiB": (512, 256), "8MiB": (1024, 512)}


def ev_time(fn, n=20):
    fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(n):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    return statistics.median(ts)


m = mgpu.Mgpu(hip, bs.tri, bs.nodes, devices=[0], transport="rccl")
m.set_env(bs.hdr, None, bs.env_filter)
for tag, (W, H) in SIZES.items():
    p = trace.make_params(W, H, eye, cam, 50, 1, spp=1, tile=(16, 16))
    m.render(p)
    m.gather(to_host=False)
    g = []
    for _ in range(20):
        m.render(trace.make_params(W, H, eye, cam, 50, 1, spp=1, frame0=1, tile=(16, 16)))
        m.gather(to_host=False)
        g.append(m.last_ms()["gather_ms"])
    nbytes = m.last_ms()["gather_bytes"]
    acc = torch.zeros((H, W, 4), dtype=torch.float32, device=dev)
    nfl = hip.lib.ezrt_tiles_packed_floats(W, H, 16, 16, 0, 1)
    packed = torch.zeros(int(nfl), dtype=torch.float32, device=dev)
    a = (W, H, 16, 16)
    pack = ev_time(lambda: hip.lib.ezrt_tiles_pack_device(acc.data_ptr(), *a, 0, 1, packed.data_ptr(), st))
    unpack = ev_time(lambda: hip.lib.ezrt_tiles_unpack_device(packed.data_ptr(), *a, 0, 1, acc.data_ptr(), st))
    both = ev_time(lambda: (hip.lib.ezrt_tiles_pack_device(acc.data_ptr(), *a, 0, 1, packed.data_ptr(), st),
                            hip.lib.ezrt_tiles_unpack_device(packed.data_ptr(), *a, 0, 1, acc.data_ptr(), st)))
    med = statistics.median(g)
    out["payloads"][tag] = {"frame": [W, H], "payload_bytes": int(nbytes),
                            "mgpu_rccl_loopback_gather_ms": {"median": round(med, 4), "min": round(min(g), 4), "max": round(max(g), 4)},
                            "pack_ms": round(pack, 4), "unpack_ms": round(unpack, 4), "pack_plus_unpack_back_to_back_ms": round(both, 4),
                            "rccl_floor_ms": round(med - both, 4)}
m.close()

# (b) torch.distributed at world size 1
import torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29731")
try:
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    for tag, (W, H) in SIZES.items():
        n = W * H * 4
        wire = torch.zeros(n, dtype=torch.float32, device=dev)
        bufs = [torch.zeros_like(wire)]
        host = []

        def call():
            t0 = time.perf_counter()
            dist.gather(wire, gather_list=bufs, dst=0)
            host.append((time.perf_counter() - t0) * 1e3)
        gms = ev_time(call)
        out["payloads"][tag]["torch_dist_gather_world1_ms"] = round(gms, 4)
        out["payloads"][tag]["torch_dist_gather_world1_host_enqueue_ms"] = round(statistics.median(host), 4)
    dist.destroy_process_group()
except Exception as e:  # noqa: BLE001
    out["torch_dist_error"] = "%s: %s" % (type(e).__name__, e)
print(json.dumps(out))
