"""GPU linear-BVH builder (include/ezrt_build.h).  Its contract is not "the same tree
as buildBVHwithSAH" but (a) a well-formed tree in the reference's node/triangle layouts, (b) the
reference's own acceptance check for a BVH, "hitBVH == brute force" (P2/main.cpp:581-588), and
(c) GPU trace == CPU oracle on the same arrays, bit for bit."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import np_oracle as NPO  # noqa: E402

from ezrt_amd import build, scenes, trace  # noqa: E402
from ezrt_amd import scene as S  # noqa: E402

pytestmark = pytest.mark.gpu


def _bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def _check_tree(tri_in, tri, nodes, leaf_n):
    n = tri.shape[0]
    assert np.array_equal(nodes[0], np.array([255, 128, 0, 30, 0, 0, 1, 1, 0, 0, 1, 0], np.float32))   # testNode
    # same triangles, reordered
    assert np.array_equal(np.sort(_bits(tri_in).view([("", np.uint32)] * 36), axis=0),
                          np.sort(_bits(tri).view([("", np.uint32)] * 36), axis=0))
    left, right = nodes[:, 0].astype(np.int64), nodes[:, 1].astype(np.int64)
    cnt, first = nodes[:, 3].astype(np.int64), nodes[:, 4].astype(np.int64)
    m = nodes.shape[0]
    leaf = cnt[1:] > 0
    ids = np.arange(1, m)
    assert np.all(left[1:][~leaf] > ids[~leaf]) and np.all(right[1:][~leaf] > ids[~leaf])
    assert np.all(right[1:][~leaf] < m) and np.all(left[1:][leaf] == 0) and np.all(right[1:][leaf] == 0)
    assert cnt[1:][leaf].max() <= leaf_n
    # leaves partition [0, n)
    order = np.argsort(first[1:][leaf])
    f, c = first[1:][leaf][order], cnt[1:][leaf][order]
    assert f[0] == 0 and np.array_equal(f[1:], (f + c)[:-1]) and f[-1] + c[-1] == n
    # every node is referenced exactly once (node 1 = root by nobody)
    refs = np.concatenate([left[1:][~leaf], right[1:][~leaf]])
    assert np.array_equal(np.sort(refs), np.arange(2, m))
    # boxes: exact min/max of the triangles below (bottom-up over ids in reverse: child id > parent id)
    P = tri[:, :9].reshape(n, 3, 3)
    lo = np.zeros((m, 3), np.float32)
    hi = np.zeros((m, 3), np.float32)
    for i in range(m - 1, 0, -1):
        if cnt[i] > 0:
            q = P[first[i]:first[i] + cnt[i]].reshape(-1, 3)
            lo[i], hi[i] = q.min(0), q.max(0)
        else:
            lo[i], hi[i] = np.minimum(lo[left[i]], lo[right[i]]), np.maximum(hi[left[i]], hi[right[i]])
    assert np.array_equal(_bits(nodes[1:, 6:9]), _bits(lo[1:])) and np.array_equal(_bits(nodes[1:, 9:12]), _bits(hi[1:]))


def _rays(n, seed):
    rng = np.random.default_rng(seed)
    o = rng.uniform(-3, 3, (n, 3))
    d = rng.uniform(-1, 1, (n, 3)) - o
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    return np.concatenate([o, d], 1).astype(np.float32)


@pytest.mark.parametrize("leaf_n", [8, 1, 3])
def test_lbvh_is_well_formed_and_matches_brute_force(hip, oracle, bunny_small, leaf_n):
    tri, nodes, ms = build.build_lbvh(bunny_small.tri, leaf_n)
    _check_tree(bunny_small.tri, tri, nodes, leaf_n)
    rays = _rays(4000, 11)
    sg = hip.scene_create(tri, nodes)
    so = oracle.scene_create(tri, nodes)
    tg, dg = sg.query_hits(rays)
    to, do = so.query_hits(rays)
    assert np.array_equal(tg, to) and np.array_equal(_bits(dg), _bits(do))
    # brute force over the whole array (chapter 2's check), through the host pointer-tree API
    _, ib, tb = S.p2Query(tri[:, :9], rays, use_bvh=False)
    hit = tg >= 0
    assert np.array_equal(hit, ib >= 0) and hit.sum() > 500
    assert np.array_equal(_bits(dg[hit]), _bits(tb[hit]))


def test_lbvh_render_parity_with_the_oracle(hip, oracle, bunny_small):
    tri, nodes, _ = build.build_lbvh(bunny_small.tri, 8)
    sg, so = hip.scene_create(tri, nodes), oracle.scene_create(tri, nodes)
    for s in (sg, so):
        s.set_env(bunny_small.hdr, bunny_small.cache, bunny_small.env_filter)
    eye, cam = S.camera(0, 0, 4)
    p = trace.make_params(96, 64, eye, cam, 50, 4, spp=3)
    sg.set_instrumentation(1)
    so.set_instrumentation(1)
    a, b = sg.render(p), so.render(p)
    assert np.array_equal(_bits(a), _bits(b))
    assert sg.counters() == so.counters()
    p = trace.make_params(96, 64, eye, cam, 51, 2, spp=2)
    assert np.array_equal(_bits(sg.render(p)), _bits(so.render(p)))


def test_lbvh_edge_cases(hip, oracle):
    rng = np.random.default_rng(5)
    one = np.zeros((1, 36), np.float32)
    one[0, :9] = [0, 0, 0, 1, 0, 0, 0, 1, 0]
    one[0, 9:18] = np.tile([0, 0, 1], 3)
    tri, nodes, _ = build.build_lbvh(one, 8)
    assert nodes.shape[0] == 2 and nodes[1, 3] == 1 and nodes[1, 4] == 0
    few = np.repeat(one, 5, axis=0)
    few[:, :9] += rng.uniform(-1, 1, (5, 9)).astype(np.float32)
    tri, nodes, _ = build.build_lbvh(few, 8)
    assert nodes.shape[0] == 2 and nodes[1, 3] == 5
    # many triangles with the SAME centroid (identical Morton codes: only the index separates the keys)
    same = np.repeat(one, 300, axis=0)
    same[:, 18:21] = np.arange(300, dtype=np.float32)[:, None]          # tell them apart by emissive
    tri, nodes, _ = build.build_lbvh(same, 4)
    _check_tree(same, tri, nodes, 4)
    rays = np.array([[0.2, 0.2, 1, 0, 0, -1]], np.float32)
    tg, dg = hip.scene_create(tri, nodes).query_hits(rays)
    to, do = oracle.scene_create(tri, nodes).query_hits(rays)
    assert tg[0] == to[0] >= 0 and dg[0] == do[0] == 1.0
    with pytest.raises(RuntimeError):
        build.build_lbvh(one, 0)


def test_lbvh_million_triangles(hip, oracle):
    c5 = scenes.mega_scene()
    tri, nodes, ms = build.build_lbvh(c5.tri, 8)
    print("LBVH of 1e6 triangles: %.2f ms on the device, %d nodes" % (ms, nodes.shape[0]))
    assert ms < 100.0
    cnt = nodes[1:, 3].astype(np.int64)
    assert cnt[cnt > 0].sum() == 1_000_000 and cnt.max() <= 8
    sg, so = hip.scene_create(tri, nodes), oracle.scene_create(tri, nodes)
    rays = _rays(3000, 3) * np.array([3, 1, 3, 1, 1, 1], np.float32)
    tg, dg = sg.query_hits(rays)
    to, do = so.query_hits(rays)
    assert np.array_equal(tg, to) and np.array_equal(_bits(dg), _bits(do)) and (tg >= 0).sum() > 300
    # the same rays through the reference-builder tree: same distances (different triangle numbering)
    s2 = c5.upload(oracle)
    t2, d2 = s2.query_hits(rays)
    assert np.array_equal(t2 >= 0, tg >= 0) and np.array_equal(_bits(d2), _bits(dg))


# ---- buildBVHwithSAH on the GPU: the parity builder, bit-identical to the host builder
def _host_build(tri36, leaf_n=8):
    hs = S.HostScene()
    hs.addTriangles(tri36)
    hs.buildBVHwithSAH(leaf_n)
    return hs.encode()


def _random_tris(n, seed, ties=False):
    rng = np.random.default_rng(seed)
    c = rng.uniform(-2, 2, (n, 1, 3))
    P = (c + rng.uniform(-0.3, 0.3, (n, 3, 3))).astype(np.float32)
    if ties:  # snap to a coarse grid and repeat triangles: many equal centroid keys and equal costs
        P = (np.round(P * 2) / 2).astype(np.float32)
        P[n // 2:] = P[: n - n // 2]
    T = np.zeros((n, 36), np.float32)
    T[:, :9] = P.reshape(n, 9)
    T[:, 9:18] = np.tile([0, 1, 0], 3)
    T[:, 18:21] = np.arange(n, dtype=np.float32)[:, None]   # tell repeated triangles apart
    return T


@pytest.mark.parametrize("n,seed,ties,leaf", [(1, 1, False, 8), (9, 2, False, 8), (300, 3, False, 8), (5000, 4, False, 4),
                                              (4000, 5, True, 8), (777, 6, True, 1)])
def test_gpu_sah_builder_equals_the_host_builder(hip, n, seed, ties, leaf):
    T = _random_tris(n, seed, ties)
    tri_h, nodes_h = _host_build(T, leaf)
    tri_g, nodes_g, ms = build.build_sah(T, leaf)
    assert np.array_equal(_bits(tri_g), _bits(tri_h))
    assert nodes_g.shape == nodes_h.shape and np.array_equal(_bits(nodes_g), _bits(nodes_h))


def test_gpu_sah_builder_on_the_bench_scene_and_the_inf_cap(hip, bunny_small):
    # C2's tree (79 820 triangles): bunny_scene(subdiv=2) went through the host builder
    c2 = scenes.bunny_scene(subdiv=2)
    hs_tri = c2.tri
    # feed the GPU builder the same INPUT order the host builder saw: rebuild the un-sorted triangle list
    hs = S.HostScene()
    bv, bf = scenes.subdivide(*scenes.mesh("bunny"), 2)
    hs.readObjText(scenes.obj_text(bv, bf), S.Material.disney(baseColor=(1, 1, 1)),
                   S.getTransformMatrix((0, 0, 0), (0.3, -1.6, 0), (1.5, 1.5, 1.5)), True)
    qv, qf = scenes.mesh("quad")
    hs.readObjText(scenes.obj_text(qv, qf), S.Material.disney(baseColor=(0.725, 0.71, 0.68)),
                   S.getTransformMatrix((0, 0, 0), (0, -1.4, 0), (18.83, 0.01, 18.83)), False)
    sv, sf = scenes.mesh("sphere")
    hs.readObjText(scenes.obj_text(sv, sf), S.Material.disney(baseColor=(1, 1, 1), emissive=(30, 20, 10)),
                   S.getTransformMatrix((0, 0, 0), (0.0, 0.9, -0.0), (1, 1, 1)), False)
    raw, _ = hs.encode()
    tri_g, nodes_g, ms = build.build_sah(raw, 8)
    print("GPU buildBVHwithSAH of %d triangles: %.1f ms on the device" % (raw.shape[0], ms))
    assert np.array_equal(_bits(tri_g), _bits(hs_tri)) and np.array_equal(_bits(nodes_g), _bits(c2.nodes))
    assert c2.build_stats["inf_cap_nodes"] > 0     # the INF = 114514 fallback is part of this tree


def test_gpu_sah_builder_million_triangles(hip):
    import time
    t0 = time.time()
    a = scenes.mega_scene(gpu_build=False)
    t1 = time.time()
    b = scenes.mega_scene(gpu_build=True)
    t2 = time.time()
    print("C5 scene: host build path %.1f s, GPU build path %.1f s (device part %.0f ms)" %
          (t1 - t0, t2 - t1, b.build_stats["gpu_build_ms"]))
    assert np.array_equal(_bits(a.tri), _bits(b.tri)) and np.array_equal(_bits(a.nodes), _bits(b.nodes))
    assert a.build_stats["inf_cap_nodes"] > 0          # the SAH INF = 114514 cap is live at this size (host builder's statistics)
    # round 4: the GPU builder is the default from 100 000 triangles on when a GPU is visible (scenes._auto_gpu_build)
    assert build.device_count() >= 1
    assert scenes._auto_gpu_build(1_000_000) and not scenes._auto_gpu_build(99_999)


@pytest.mark.parametrize("n,seed,ties,leaf", [(1, 1, False, 8), (300, 3, False, 8), (5000, 4, False, 4), (4000, 5, True, 8)])
def test_gpu_median_builder_equals_the_host_builder(hip, n, seed, ties, leaf):
    T = _random_tris(n, seed, ties)
    hs = S.HostScene()
    hs.addTriangles(T)
    hs.buildBVH(leaf)
    tri_h, nodes_h = hs.encode()
    tri_g, nodes_g, _ = build.build_median(T, leaf)
    assert np.array_equal(_bits(tri_g), _bits(tri_h))
    assert nodes_g.shape == nodes_h.shape and np.array_equal(_bits(nodes_g), _bits(nodes_h))
