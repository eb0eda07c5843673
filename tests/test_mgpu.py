"""include/ezrt_mgpu.h: one process, N devices, one frame.

CPU: the tile index rules (include/ezrt_tiles.h) against tiles.TilePlan, and the whole render -> pack -> transport ->
un-permute sequence through the oracle's implementation of the header (host memory as "devices").  GPU: the product's
implementation with several shards on the one GPU of the test box (peer and host transports), bit-identical to the
single-device frame of ezrt_render; RCCL needs distinct devices, so on one GPU only its N = 1 path runs here -- the N > 1
RCCL exchange is exercised by bench.py --gpus N under torch.distributed (same packed layout, same kernels).  What one GPU CAN
exercise of RCCL is run here (round 5): ezrt_mgpu with EZRT_TRANSPORT_RCCL on one device binds librccl, creates the communicator
and sends the frame to itself through a grouped ncclSend / ncclRecv; torch.distributed's nccl backend with world_size 1 runs a
collective and tiles.gather_frame's nccl branch (test_gpu_nccl_backend_world_size_one_*)."""
import ctypes as C

import numpy as np
import pytest
import torch

from ezrt_amd import _abi, mgpu, tiles


def _packed_index_to_pixel(lib, W, H, tw, th, rank, world):
    """Pixel index (y * W + x, -1 outside) of every texel of a packed shard, via the library's own pack kernel run
    on an image whose texels hold their own index."""
    n = lib.ezrt_tiles_packed_floats(W, H, tw, th, rank, world)
    img = np.zeros((H, W, 4), np.float32)
    img[..., 0] = np.arange(H * W, dtype=np.float32).reshape(H, W) + 1.0
    out = np.full(max(n, 4), -7.0, np.float32)
    return n, img, out


@pytest.mark.parametrize("W,H,tw,th,world", [(64, 64, 16, 16, 2), (203, 117, 32, 32, 8), (40, 24, 8, 8, 3), (16, 16, 0, 0, 4),
                                             (33, 17, 16, 8, 5)])
def test_packed_layout_is_tileplans(oracle, W, H, tw, th, world):
    """ezrt_tiles_pack_device / unpack (oracle build = the same include/ezrt_tiles.h on the CPU) == tiles.TilePlan."""
    lib = oracle.lib
    etw, eth = (tw or 32), (th or 32)
    plan = tiles.TilePlan(W, H, etw, eth, world)
    img = np.random.default_rng(1).random((H, W, 4)).astype(np.float32)
    back = np.full((H, W, 4), -1.0, np.float32)
    for r in range(world):
        n = lib.ezrt_tiles_packed_floats(W, H, tw, th, r, world)
        ids, n_real = plan.tile_ids(r)
        assert n == n_real * etw * eth * 4
        packed = np.full(max(n, 1), -3.0, np.float32)
        assert lib.ezrt_tiles_pack_device(img.ctypes.data, W, H, tw, th, r, world, packed.ctypes.data, None) == 0
        want = plan.pack(torch.from_numpy(img), r)[:n_real].numpy().reshape(-1)
        assert np.array_equal(packed[:n], want)
        assert lib.ezrt_tiles_unpack_device(packed.ctypes.data, W, H, tw, th, r, world, back.ctypes.data, None) == 0
    assert np.array_equal(back, img)
    assert lib.ezrt_tiles_packed_floats(W, H, tw, th, world, world) == -1


def _scene_and_params(integ=50, W=48, H=40, T=8, spp=3):
    from ezrt_amd import scene as S, scenes, trace
    bs = scenes.bunny_scene(subdiv=0, want_cache=True)
    eye, cam = S.camera(0, 0, 4)
    return bs, (lambda **kw: trace.make_params(W, H, eye, cam, integ, 3, tile=(T, T), **kw))


@pytest.mark.parametrize("n_dev", [1, 3, 4])
def test_oracle_mgpu_frame_equals_single_frame(oracle, n_dev):
    bs, mk = _scene_and_params()
    m = mgpu.Mgpu(oracle, bs.tri, bs.nodes, devices=[0] * n_dev, transport="host")
    m.set_env(bs.hdr, bs.cache, bs.env_filter)
    so = bs.upload(oracle)
    m.render(mk(spp=2))
    m.render(mk(spp=3, frame0=2))          # the shards continue their running means
    got = m.gather()
    want = so.render(mk(spp=5))
    assert np.array_equal(got, want)
    assert m.counters()["samples"] == 48 * 40 * 5
    ms = m.last_ms()
    assert len(ms["render_ms"]) == n_dev
    # payload: every tile but the root's crosses once
    plan = tiles.TilePlan(48, 40, 8, 8, n_dev)
    assert ms["gather_bytes"] == (plan.n_tiles - plan.tile_ids(0)[1]) * 8 * 8 * 16
    m.close()


def test_mgpu_argument_errors(oracle):
    from ezrt_amd import trace
    bs, mk = _scene_and_params()
    with pytest.raises(trace.TraceError):
        mgpu.Mgpu(oracle, bs.tri, bs.nodes, devices=[], transport="host")
    m = mgpu.Mgpu(oracle, bs.tri, bs.nodes, devices=[0, 0], transport="host")
    with pytest.raises(trace.TraceError):
        m.gather()                         # nothing rendered yet


# ----------------------------------------------------------------------------- GPU
@pytest.mark.gpu
@pytest.mark.parametrize("transport,n_dev", [("peer", 1), ("peer", 4), ("host", 3), ("peer", 8), ("rccl", 1)])
def test_gpu_mgpu_frame_bit_equal_to_ezrt_render(hip, transport, n_dev):
    bs, mk = _scene_and_params(integ=51, W=200, H=120, T=16, spp=4)
    m = mgpu.Mgpu(hip, bs.tri, bs.nodes, devices=[0] * n_dev, transport=transport)
    m.set_env(bs.hdr, bs.cache, bs.env_filter)
    sg = bs.upload(hip)
    m.render(mk(spp=4))
    first = m.gather()
    assert np.array_equal(first, sg.render(mk(spp=4)))
    m.render(mk(spp=3, frame0=4))          # continue after a gather: the root's buffer still holds its own shard
    assert np.array_equal(m.gather(), sg.render(mk(spp=7)))
    assert m.counters()["samples"] == 200 * 120 * 7
    ms = m.last_ms()
    assert all(t > 0 for t in ms["render_ms"]) and (n_dev == 1 or ms["gather_ms"] > 0)
    # the assembled frame is also device-resident on the root
    assert m.frame_device()
    # a new frame size starts over
    bs2, mk2 = _scene_and_params(integ=50, W=64, H=64, T=16)
    m.render(mk2(spp=2))
    assert np.array_equal(m.gather(), sg.render(mk2(spp=2)))
    m.close()


@pytest.mark.gpu
def test_gpu_rccl_transport_rejects_repeated_devices(hip):
    from ezrt_amd import trace
    bs, _ = _scene_and_params()
    with pytest.raises(trace.TraceError, match="distinct"):
        mgpu.Mgpu(hip, bs.tri, bs.nodes, devices=[0, 0], transport="rccl")
    with pytest.raises(trace.TraceError, match="not visible"):
        mgpu.Mgpu(hip, bs.tri, bs.nodes, devices=[0, 99], transport="peer")


_NCCL_WORLD_ONE = r"""
import os, sys, socket
import numpy as np, torch, torch.distributed as dist
sys.path.insert(0, os.getcwd())
from ezrt_amd import scene as S, scenes, trace, tiles
s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
torch.cuda.set_device(0)
dist.init_process_group("nccl", init_method="tcp://127.0.0.1:%d" % port, rank=0, world_size=1)
assert dist.get_backend() == "nccl"
t = torch.arange(1024, dtype=torch.float32, device="cuda")
dist.all_reduce(t)                      # a real RCCL collective: the communicator is created, a kernel runs
torch.cuda.synchronize()
assert torch.equal(t.cpu(), torch.arange(1024, dtype=torch.float32))
hip = trace.hip()
bs = scenes.bunny_scene(subdiv=0, want_cache=True)
sc = bs.upload(hip)
eye, cam = S.camera(0, 0, 4)
W, H, T = 96, 64, 16
p = trace.make_params(W, H, eye, cam, 51, 3, spp=3, tile=(T, T), shard=(0, 1))
want = sc.render(p)
acc = torch.zeros((H, W, 4), dtype=torch.float32, device="cuda")
sc.render_device(p, acc.data_ptr(), torch.cuda.current_stream().cuda_stream)
plan = tiles.TilePlan(W, H, T, T, 1)
out = tiles.gather_frame(acc, plan, 0, dist, dst=0, lib=hip.lib)       # the driver's N-GPU code path, nccl branch, one rank
torch.cuda.synchronize()
assert np.array_equal(out.cpu().numpy().view(np.uint32), np.asarray(want).view(np.uint32))
out2 = tiles.gather_frame(acc.clone(), plan, 0, dist, dst=0)            # torch-indexing variant: pack -> dist.gather -> unpack
torch.cuda.synchronize()
assert np.array_equal(out2.cpu().numpy().view(np.uint32), np.asarray(want).view(np.uint32))
dist.barrier()
dist.destroy_process_group()
print("NCCL_WORLD_ONE_OK")
"""


@pytest.mark.gpu
def test_gpu_nccl_backend_world_size_one_runs_a_collective_and_gather_frame():
    """RCCL has never seen this code on N > 1 GPUs (the environment grants one).  What one GPU allows: the nccl
    backend initialised with world_size 1, a collective on a device tensor, and tiles.gather_frame through its nccl branch --
    in a process of its own under a timeout, so that a transport that hangs fails the test instead of the suite."""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    r = subprocess.run([sys.executable, "-c", _NCCL_WORLD_ONE], cwd=root, env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "NCCL_WORLD_ONE_OK" in r.stdout, (r.stdout[-2000:], r.stderr[-4000:])


@pytest.mark.gpu
def test_gpu_pack_unpack_kernels_match_tileplan(hip):
    W, H, tw, th, world = 203, 117, 32, 16, 5
    plan = tiles.TilePlan(W, H, tw, th, world)
    img = torch.rand(H, W, 4, device="cuda")
    back = torch.full_like(img, -1.0)
    st = torch.cuda.current_stream().cuda_stream
    for r in range(world):
        n = hip.lib.ezrt_tiles_packed_floats(W, H, tw, th, r, world)
        packed = torch.full((max(n, 1),), -3.0, device="cuda")
        assert hip.lib.ezrt_tiles_pack_device(img.data_ptr(), W, H, tw, th, r, world, packed.data_ptr(), st) == 0
        ids, n_real = plan.tile_ids(r, img.device)
        assert torch.equal(packed[:n], plan.pack(img, r)[:n_real].reshape(-1))
        assert hip.lib.ezrt_tiles_unpack_device(packed.data_ptr(), W, H, tw, th, r, world, back.data_ptr(), st) == 0
    torch.cuda.synchronize()
    assert torch.equal(back, img)
