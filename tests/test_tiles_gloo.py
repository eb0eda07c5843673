"""N > 1 path on CPU: TilePlan pack/unpack and the frame-closing gather with world_size 2 over gloo.
The renderer in these tests is the CPU oracle (test infrastructure); on GPUs bench.py drives the same
functions with libezrt_hip.so and the nccl (= RCCL) backend."""
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

from ezrt_amd import tiles  # noqa: E402


@pytest.mark.parametrize("W,H,tw,th,world", [(64, 64, 16, 16, 2), (203, 117, 32, 32, 8), (40, 24, 8, 8, 3), (16, 16, 32, 32, 4)])
def test_tile_plan_pack_unpack_roundtrip(W, H, tw, th, world):
    plan = tiles.TilePlan(W, H, tw, th, world)
    img = torch.arange(H * W * 4, dtype=torch.float32).reshape(H, W, 4)
    # each virtual rank only has its own tiles valid
    packed = []
    for r in range(world):
        own = torch.full_like(img, -1.0)
        ids, n_real = plan.tile_ids(r)
        t_all = plan.to_tiles(img)
        t_own = plan.to_tiles(own).clone()
        t_own[ids[:n_real]] = t_all[ids[:n_real]]
        packed.append(plan.pack(plan.from_tiles(t_own), r))
        assert packed[-1].shape == (plan.per_rank, th, tw, 4)
    assert torch.equal(plan.unpack(packed), img)
    owners = [plan.owner(t) for t in range(plan.n_tiles)]
    assert owners == [t % world for t in range(plan.n_tiles)]


def test_tile_ownership_rule_matches_the_kernels(oracle, bunny_small):
    """TilePlan's tile_id % world rule == EzrtRenderParams.shard_* as the trace applies it."""
    from ezrt_amd import scene as S, trace
    sc = bunny_small.upload(oracle)
    eye, cam = S.camera()
    W, H, tw, th, world = 40, 24, 8, 8, 3
    plan = tiles.TilePlan(W, H, tw, th, world)
    for r in range(world):
        img = np.full((H, W, 4), -5.0, np.float32)
        sc.render(trace.make_params(W, H, eye, cam, 3, 1, spp=1, tile=(tw, th), shard=(r, world)), img)
        touched = torch.from_numpy(img[..., 3] == 1.0)
        t = plan.to_tiles(touched[..., None].float())[..., 0]
        ids, n_real = plan.tile_ids(r)
        mine = torch.zeros(plan.n_tiles, dtype=torch.bool)
        mine[ids[:n_real]] = True
        assert torch.equal(t.reshape(plan.n_tiles, -1).max(1).values.bool(), mine)


def _worker(rank, world, port, out_path):
    import ctypes
    import torch.distributed as dist
    sys.path.insert(0, ROOT)
    from ezrt_amd import _abi, scene as S, scenes, trace
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    ora = trace.TraceLib(_abi.declare_trace_abi(ctypes.CDLL(os.path.join(ROOT, "oracle", "libezrt_oracle.so"))))
    bs = scenes.bunny_scene(subdiv=0)
    sc = bs.upload(ora)
    eye, cam = S.camera(0, 0, 4)
    W, H, T = 48, 40, 8
    plan = tiles.TilePlan(W, H, T, T, world)
    p = trace.make_params(W, H, eye, cam, 50, 3, spp=3, tile=(T, T), shard=(rank, world))
    accum = np.zeros((H, W, 4), np.float32)
    sc.render(p, accum)
    frame = tiles.gather_frame(torch.from_numpy(accum), plan, rank, dist)
    # the same close through the C entry points of include/ezrt_mgpu.h (ezrt_tiles_pack_device / _unpack_device; here the
    # oracle's build of them, host memory as the transport's end points), in place in the rank's frame buffer
    native = tiles.gather_frame(torch.from_numpy(accum.copy()), plan, rank, dist, lib=ora.lib)
    if rank == 0:
        full = sc.render(trace.make_params(W, H, eye, cam, 50, 3, spp=3))
        np.save(out_path, np.stack([frame.numpy(), full, native.numpy()]))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gloo_gather_equals_single_process_frame(tmp_path, oracle):
    import torch.multiprocessing as mp
    out = str(tmp_path / "frames.npy")
    port = 29500 + (os.getpid() % 2000)
    mp.spawn(_worker, args=(2, port, out), nprocs=2, join=True)
    got, want, native = np.load(out)
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
    assert np.array_equal(native.view(np.uint32), want.view(np.uint32))
