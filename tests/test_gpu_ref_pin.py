"""GPU intersector vs the REFERENCE's C++ twins (oracle/_ref/libezrt_ref_p2.so = P2/main.cpp
compiled as it is; the prebuilt library travels to the GPU box, inputs are synthetic or come from the
mesh assets, /root/reference is not needed)."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import ref as R  # noqa: E402

pytestmark = [pytest.mark.gpu,
              pytest.mark.skipif(not R.available("p2"), reason="oracle/_ref not built")]


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def _rays(n, seed, lo=-3, hi=3):
    rng = np.random.default_rng(seed)
    o = rng.uniform(lo, hi, (n, 3))
    tgt = rng.uniform(-1, 1, (n, 3))
    d = tgt - o
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    return np.concatenate([o, d], 1).astype(np.float32)


def test_device_hit_aabb_both_forms_equal_the_reference_cpp_twin(hip):
    """hit_aabb (select form) and hit_aabb_tame (v_min3/v_max3, the one traceq_kernel runs for tame rays)
    against hitAABB of P2/main.cpp:449-463 on 10^6 ray/box pairs incl. origins inside, boxes behind,
    axis-parallel directions (1/0 = inf)."""
    p2 = R.P2()
    rng = np.random.default_rng(6)
    n = 1_000_000
    rays = _rays(n, 7)
    ax = rng.random(n) < 0.15
    k = rng.integers(0, 3, n)
    rays[ax, 3 + k[ax]] = 0.0
    c = rng.uniform(-2, 2, (n, 3))
    e = rng.uniform(0.01, 1.5, (n, 3))
    boxes = np.concatenate([c - e, c + e], 1).astype(np.float32)
    t_ref = p2.hitAABB(rays, boxes)
    t_sel = hip.debug_math(10, rays, boxes, n)
    t_hw = hip.debug_math(12, rays, boxes, n)
    nan = np.isnan(t_ref)
    assert nan.mean() < 0.01
    assert np.array_equal(bits(t_ref)[~nan], bits(t_sel)[~nan]) and np.isnan(t_sel[nan]).all()
    # the hardware form is only used for tame rays (finite origin and 1/direction): NaN marks "not used".
    # It may return -0.0 where the select form returns +0.0: the value only feeds `> 0` and `<` tests.
    tame = ~np.isnan(t_hw)
    assert tame.mean() > 0.8 and not np.isnan(t_ref[tame]).any()
    assert np.array_equal(t_ref[tame], t_hw[tame])
    assert np.array_equal(t_ref[tame] > 0, t_hw[tame] > 0)


def test_device_hit_triangle_equals_the_reference_cpp_twin(hip):
    """hit_triangle_t on the 48-B device record (precomputed unit normal) == hitTriangle of
    P2/main.cpp:212-238: same hit decision, same t, 10^6 pairs."""
    p2 = R.P2()
    rng = np.random.default_rng(3)
    n = 1_000_000
    c = rng.uniform(-1, 1, (n, 1, 3))
    T = (c + rng.uniform(-0.8, 0.8, (n, 3, 3))).astype(np.float32).reshape(n, 9)
    rays = _rays(n, 5, -2, 2)
    aim = rng.random(n) < 0.6
    w = rng.dirichlet((1, 1, 1), n).astype(np.float32)
    tgt = (T.reshape(n, 3, 3) * w[:, :, None]).sum(1)
    d = tgt - rays[:, :3]
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    rays[aim, 3:] = d[aim].astype(np.float32)
    t_ref = p2.hitTriangle(T, rays)
    t_gpu = hip.debug_math(11, rays, T, n)
    assert (t_ref < p2.INF).mean() > 0.3
    assert np.array_equal(bits(t_ref), bits(t_gpu))


def test_gpu_hit_bvh_equals_the_reference_brute_force_on_the_bunny(hip, bunny_small):
    """chapter 2's acceptance check (P2/main.cpp:585) with the reference's own hitTriangleArray as the
    brute force: the GPU traversal returns its winner and its distance bit for bit."""
    p2 = R.P2()
    p2.setTriangles(bunny_small.tri[:, :9])
    rays = _rays(50000, 8)
    i_ref, t_ref = p2.hitTriangleArray(rays)
    sg = bunny_small.upload(hip)
    i_gpu, t_gpu = sg.query_hits(rays)
    hit = i_ref >= 0
    assert hit.sum() > 10000
    assert np.array_equal(i_ref, i_gpu)
    assert np.array_equal(bits(t_ref[hit]), bits(t_gpu[hit]))
