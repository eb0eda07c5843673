"""The device's own tree over the REFERENCE'S leaves (ezrt_scene_build.hip retree_leaves, EZRT_RETREE, default on).

For a tame ray and nested boxes the fp32 slab test is monotone, so the reference's hitBVH reaches a leaf iff the leaf's OWN
box is hit -- the inner nodes do not matter.  The library therefore builds its 4-wide records over a binned-SAH tree of the
reference's leaf boxes instead of over the reference's inner nodes (which its builder's INF = 114514 cap degrades to
median-x splits at the top of large scenes).  These tests check the consequence -- identical hit records and frames with
the re-tree on and off, both equal to the oracle -- through the TIMED kernels, on the Bunny, on the adversarial soup of
tests/test_gpu_prune.py, on a deep skewed tree, and on a scene whose reference tree hits the cap."""
import os

import numpy as np
import pytest

from ezrt_amd import scene as S
from ezrt_amd import scenes, trace

pytestmark = pytest.mark.gpu


def _bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


class _Env:
    def __init__(self, **kv):
        self.kv, self.old = kv, {}

    def __enter__(self):
        for k, v in self.kv.items():
            self.old[k] = os.environ.get(k)
            os.environ[k] = str(v)

    def __exit__(self, *a):
        for k, v in self.old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


def _create(hip, tri, nodes, retree):
    with _Env(EZRT_RETREE=retree):          # read at scene creation
        sc = hip.scene_create(tri, nodes)
    assert sc.prune_info()["retreed"] == float(retree)
    return sc


def _rays(n, seed):
    rng = np.random.default_rng(seed)
    o = rng.uniform(-3, 3, (n, 3))
    d = rng.uniform(-2, 2, (n, 3)) - o
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    r = np.concatenate([o, d], 1).astype(np.float32)
    wild = rng.random(n) < 0.01
    r[wild, 3 + rng.integers(0, 3, int(wild.sum()))] = 0.0      # not tame: redo route, reference tree
    return r


def test_retree_changes_no_hit_record_and_no_frame(hip, oracle, bunny_small):
    so = bunny_small.upload(oracle)
    rays = _rays(300000, 5)
    to, do = so.query_hits(rays)
    eye, cam = S.camera(10, 5, 3)
    frames = {}
    for retree in (0, 1):
        sg = _create(hip, bunny_small.tri, bunny_small.nodes, retree)
        sg.set_env(bunny_small.hdr, bunny_small.cache, 1)
        for prune in (0, 2):
            sg.set_option("prune", prune)
            sg.set_option("audit_via_queue", 1)
            tg, dg = sg.query_hits(rays)
            assert np.array_equal(tg, to) and np.array_equal(_bits(dg), _bits(do)), (retree, prune)
            for integ, mb in ((50, 4), (51, 2)):
                p = trace.make_params(160, 120, eye, cam, integ, mb, spp=2)
                frames[(retree, prune, integ)] = sg.render(p)
                pa = trace.make_params(160, 120, eye, cam, integ, mb, frame0=1)
                tg2, dg2, _ = sg.render_paths(pa)
                to2, do2, _ = so.render_paths(pa)
                assert np.array_equal(tg2, to2) and np.array_equal(_bits(dg2), _bits(do2)), (retree, prune, integ)
    for integ, mb in ((50, 4), (51, 2)):
        want = so.render(trace.make_params(160, 120, eye, cam, integ, mb, spp=2))
        for retree in (0, 1):
            for prune in (0, 2):
                assert np.array_equal(_bits(frames[(retree, prune, integ)]), _bits(want)), (retree, prune, integ)


def test_retree_on_a_tree_that_hits_the_builders_cost_cap(hip, oracle):
    """A wide, flat scene of large triangles: area x count >= 114514 at the top, so the reference builder falls back to
    median-x splits there -- the case the re-tree exists for."""
    rng = np.random.default_rng(9)
    n = 20000
    c = rng.uniform(-400, 400, (n, 1, 3)) * np.array([1.0, 0.02, 1.0])
    P = (c + rng.uniform(-6, 6, (n, 3, 3))).astype(np.float32)
    T = np.zeros((n, 36), np.float32)
    T[:, :9] = P.reshape(n, 9)
    T[:, 9:18] = np.tile([0, 1, 0], 3)
    T[:, 18:36] = S.Material.disney(baseColor=(0.7, 0.7, 0.7)).to18()
    hs = S.HostScene()
    hs.addTriangles(T)
    hs.buildBVHwithSAH(8)
    assert hs.buildStats()["inf_cap_nodes"] > 0
    tri, nodes = hs.encode()
    so = oracle.scene_create(tri, nodes)
    o = rng.uniform(-400, 400, (200000, 3)) * np.array([1.0, 0.1, 1.0]) + np.array([0, 30.0, 0])
    d = rng.uniform(-400, 400, (200000, 3)) * np.array([1.0, 0.02, 1.0]) - o
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    rays = np.concatenate([o, d], 1).astype(np.float32)
    to, do = so.query_hits(rays)
    assert 0.1 < (to >= 0).mean()
    for retree in (0, 1):
        sg = _create(hip, tri, nodes, retree)
        sg.set_option("audit_via_queue", 1)
        tg, dg = sg.query_hits(rays)
        assert np.array_equal(tg, to) and np.array_equal(_bits(dg), _bits(do)), retree


def test_retree_keeps_the_counters_of_the_reference_tree(hip, oracle, bunny_small):
    """P / I / T / M are properties of the REFERENCE's tree: the instrumented run traverses it, re-tree or not."""
    so = bunny_small.upload(oracle)
    so.set_instrumentation(1)
    eye, cam = S.camera(0, 0, 4)
    p = trace.make_params(96, 80, eye, cam, 50, 3, spp=2)
    so.render(p)
    for retree in (0, 1):
        sg = _create(hip, bunny_small.tri, bunny_small.nodes, retree)
        sg.set_env(bunny_small.hdr, bunny_small.cache, 1)
        sg.set_instrumentation(1)
        sg.render(p)
        assert sg.counters() == so.counters()


@pytest.mark.gpu
def test_scene_create_does_not_depend_on_the_host_thread_count(hip):
    """ezrt_scene_create builds the device's tree over the leaves, the geometry / shading records and the pruning bounds on up
    to 16 host threads (subtrees of the re-tree spliced behind its top: another numbering of the same tree).  One thread and
    the default must give the same scene: the same pruning facts, the same answers and -- the schedule being a function of
    the records -- the same work counters for the same rays (C3: 89 k leaves, enough for the parallel re-tree)."""
    import os
    bs = scenes.disney_grid_scene(subdiv=3)
    eye, cam = S.camera(0, 15, 8)
    rng = np.random.default_rng(3)
    o = rng.uniform(-3, 3, (20000, 3)).astype(np.float32)
    d = rng.normal(size=(20000, 3)).astype(np.float32)
    rays = np.concatenate([o, d / np.linalg.norm(d, axis=1, keepdims=True)], 1).astype(np.float32)
    got = []
    for threads in ("1", None):
        if threads is None:
            os.environ.pop("EZRT_HOST_THREADS", None)
        else:
            os.environ["EZRT_HOST_THREADS"] = threads
        try:
            sc = bs.upload(hip)
        finally:
            os.environ.pop("EZRT_HOST_THREADS", None)
        sc.set_option("audit_via_queue", 1)
        tri, t = sc.query_hits(rays)
        img = sc.render(trace.make_params(256, 256, eye, cam, 4, 3, spp=2))
        got.append((sc.prune_info(), sc.stats(), tri, t, img))
    a, b = got
    assert a[0] == b[0] and a[1] == b[1]
    assert np.array_equal(a[2], b[2]) and np.array_equal(a[3].view(np.uint32), b[3].view(np.uint32))
    assert np.array_equal(a[4].view(np.uint32), b[4].view(np.uint32))
    assert (a[2] >= 0).mean() > 0.05


@pytest.mark.gpu
def test_a_leaf_with_two_parents_keeps_the_binary_kernel(hip, oracle, bunny_small):
    """Caller arrays may reference a node from several parents (validation only asks parent < child).  The 4-wide records, the
    re-tree and the tie tables hold ONE parent per node -- for leaves as well -- so such arrays must keep the binary
    in-order kernel, which visits the shared leaf once per parent exactly as the reference's hitBVH would; results == oracle."""
    nodes = bunny_small.nodes.copy()
    is_leaf = nodes[:, 3] > 0
    # an inner node q whose LEFT child is a leaf, and a leaf L with a larger id somewhere else: q's left child becomes L
    cand_q = [i for i in range(2, nodes.shape[0]) if not is_leaf[i] and is_leaf[int(nodes[i, 0])]]
    q = cand_q[len(cand_q) // 3]
    leaves_after = [i for i in range(int(nodes[q, 0]) + 50, nodes.shape[0]) if is_leaf[i]]
    L = leaves_after[0]
    nodes[q, 0] = np.float32(L)
    sg, so = hip.scene_create(bunny_small.tri, nodes), oracle.scene_create(bunny_small.tri, nodes)
    assert sg.prune_info()["records4"] == 0          # no 4-wide records: the binary kernel traces this scene
    hdr = scenes.synthetic_hdr(64, 32)
    sg.set_env(hdr, None, 1)
    so.set_env(hdr, None, 1)
    eye, cam = S.camera(0, 0, 4)
    p = trace.make_params(96, 96, eye, cam, 50, 3, spp=2)
    assert np.array_equal(sg.render(p).view(np.uint32), so.render(p).view(np.uint32))
    sg.set_instrumentation(1)
    so.set_instrumentation(1)
    sg.counters_reset()
    so.counters_reset()
    sg.render(p)
    so.render(p)
    assert sg.counters() == so.counters()
